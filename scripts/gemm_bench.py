"""times the fp32 MFMA GEMM on the shapes of the ELBO step (through the avae_debug_gemm hook)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argsim_amd import lib
l = lib.load()
cfg = lib.AvaeConfig(32, 16, 8, 1, 1e-4, 1e-3, 2, 1, 0, 0, 1.0, 0.0, int(os.environ.get('DTYPE', '0')))   # 0 fp32 MFMA, 1 bf16, 2 split bf16x3
h = C.c_void_p(); assert l.avae_create(C.byref(cfg), 0, C.byref(h)) == 0
dev = torch.device('cuda', 0)
shapes = [  # name, a_mc, b_nc, M, N, K, split
    ('big NT 8192^2 x4096', 0, 0, 8192, 8192, 4096, 1),
    ('enc gi L2 NT', 0, 0, 16384, 3072, 1024, 1),
    ('enc gi L1 NT', 0, 0, 16384, 3072, 512, 1),
    ('logits NT', 0, 0, 16640, 8192, 512, 1),
    ('dec gi NT', 0, 0, 16384, 1536, 512, 1),
    ('dho NN K=8192', 0, 1, 16384, 512, 8192, 1),
    ('enc dX NN', 0, 1, 16384, 1024, 3072, 1),
    ('dec dX NN', 0, 1, 16384, 512, 1536, 1),
    ('dE TN', 1, 1, 8192, 512, 16640, 1),
    ('enc dW TN split4', 1, 1, 3072, 1024, 16384, 4),
    ('dec dW TN split16', 1, 1, 1536, 512, 16640, 16),
    ('dec dW TN 64x64 s4', 1, 1, 1536, 512, 16640, 1004),
    ('dec dW TN 64x64 s8', 1, 1, 1536, 512, 16640, 1008),
    ('dE TN 64x64 s1', 1, 1, 8192, 512, 16640, 1001),
    ('enc dW TN 64x64 s1', 1, 1, 3072, 1024, 16384, 1001),
    ('enc dW TN 64x64 s2', 1, 1, 3072, 1024, 16384, 1002),
    ('big TN 4096^2 x8192', 1, 1, 4096, 4096, 8192, 1),
    ('big NN 8192^2 x4096', 0, 1, 8192, 8192, 4096, 1),
]
if os.environ.get('ONLY'): shapes = [x for x in shapes if x[0].startswith(os.environ['ONLY'])]
variants = [int(x) for x in os.environ.get('VARIANTS', '0').split(',')]
import itertools
for (name, a_mc, b_nc, M, N, K, split), var in itertools.product(shapes, variants):
    if var and split != 1: continue
    if var: split = -var; name = name + ' v%d' % var
    A = torch.randn((K, M) if a_mc else (M, K), device=dev)
    B = torch.randn((K, N) if b_nc else (N, K), device=dev)
    Cm = torch.zeros((M, N), device=dev)
    lda = M if a_mc else K; ldb = N if b_nc else K
    def run():
        rc = l.avae_debug_gemm(h, a_mc, b_nc, A.data_ptr(), B.data_ptr(), Cm.data_ptr(), None, M, N, K, lda, ldb, N, 1.0, 0, split)
        assert rc == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print('%-24s M %6d N %5d K %6d  %8.1f us  %6.1f TFLOP/s' % (name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
