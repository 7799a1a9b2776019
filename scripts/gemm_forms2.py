"""tile form x K split of the fp32 GEMM on the launches of the ELBO step that cannot fill the chip (present-id GEMMs of the table-fed
layers, remainder rows, latent block) through the avae_debug_gemm hook.  usage: gemm_forms2.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argsim_amd import lib
l = lib.load()
cfg = lib.AvaeConfig(32, 16, 8, 1, 1e-4, 1e-3, 2, 1, 0, 0, 1.0, 0.0, 0)
h = C.c_void_p(); assert l.avae_create(C.byref(cfg), 0, C.byref(h)) == 0
dev = torch.device('cuda', 0)
shapes = [('enc demb NN', 0, 1, 3584, 512, 3072), ('dec demb NN', 0, 1, 3328, 512, 1536), ('dec proj NT', 0, 0, 3328, 1536, 512),
          ('enc proj NT', 0, 0, 3584, 3072, 512), ('dx tail NN', 0, 1, 256, 512, 1536), ('enc dx tail NN', 0, 1, 256, 1024, 3072),
          ('dho tail NN', 0, 1, 256, 512, 8192), ('mu NN', 0, 1, 256, 128, 1024), ('gib NT', 0, 0, 256, 1536, 1024), ('dxl NN', 0, 1, 256, 1024, 1536)]
forms = [('128', 1), ('128/2', 2), ('128/3', 3), ('128/4', 4), ('128/6', 6), ('128/8', 8), ('64', 1001), ('64/2', 1002), ('64/3', 1003), ('64/4', 1004), ('32x128', -1), ('skinny', -3)]
for name, a_mc, b_nc, M, N, K in shapes:
    A = torch.randn((M, K), device=dev); B = torch.randn((K, N) if b_nc else (N, K), device=dev); Cm = torch.zeros((M, N), device=dev)
    line = []
    for form, split in forms:
        def run():
            assert l.avae_debug_gemm(h, a_mc, b_nc, A.data_ptr(), B.data_ptr(), Cm.data_ptr(), None, M, N, K, K, N if b_nc else K, N, 1.0, 0, split) == 0
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        line.append('%s %.1f' % (form, e0.elapsed_time(e1) / 20 * 1e3))
    print('%-16s M %5d N %5d K %5d (ideal %.1f us) | %s' % (name, M, N, K, 2.0 * M * N * K / 157.3e6, '  '.join(line)), flush=True)
