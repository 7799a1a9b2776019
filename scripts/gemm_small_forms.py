"""the tile forms of the fp32 GEMM on the narrow mid-size shapes of the ELBO step (out affine, decoder-top dx, present-id projections)
through the avae_debug_gemm hook: 128x128 tiles, 64x64 tiles, 32x128 tiles, skinny."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argsim_amd import lib
l = lib.load()
cfg = lib.AvaeConfig(32, 16, 8, 1, 1e-4, 1e-3, 2, 1, 0, 0, 1.0, 0.0, 0)
h = C.c_void_p(); assert l.avae_create(C.byref(cfg), 0, C.byref(h)) == 0
dev = torch.device('cuda', 0)
if os.environ.get('RAGGED_SHAPES'):       # the compact layout's row count on the RAGGED set (7.4 k real positions of 16.4 k)
    shapes = [('dec gi NT', 0, 0, 7424, 1536, 512), ('dec dx NN', 0, 1, 7424, 512, 1536), ('enc gi L2 NT', 0, 0, 7168, 3072, 1024),
              ('enc dx NN', 0, 1, 7168, 1024, 3072), ('out affine NN', 0, 1, 7424, 512, 512), ('logits NT', 0, 0, 7424, 8192, 512)]
else:
  shapes = [('out affine NN', 0, 1, 16384, 512, 512), ('out affine tail', 0, 1, 256, 512, 512), ('dhc NT', 0, 0, 16384, 512, 512),
          ('table dec NT U=3328', 0, 0, 3328, 1536, 512), ('table enc NT U=3584', 0, 0, 3584, 3072, 512),
          ('logits tail NT', 0, 0, 256, 8192, 512), ('dec tail NT', 0, 0, 256, 1536, 512)]
for name, a_mc, b_nc, M, N, K in shapes:
    A = torch.randn((M, K), device=dev); B = torch.randn((K, N) if b_nc else (N, K), device=dev); Cm = torch.zeros((M, N), device=dev)
    for form, split in (('128x128', 1), ('64x64', 1001), ('32x128', -1), ('skinny', -3)):
        def run():
            assert l.avae_debug_gemm(h, a_mc, b_nc, A.data_ptr(), B.data_ptr(), Cm.data_ptr(), None, M, N, K, K, N if b_nc else K, N, 1.0, 0, split) == 0
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('%-22s %-8s M %6d N %5d K %5d  %7.1f us  %6.1f TFLOP/s' % (name, form, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
