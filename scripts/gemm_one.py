"""runs ONE GEMM shape a few times (for rocprofv3 --pmc passes)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argsim_amd import lib
l = lib.load()
cfg = lib.AvaeConfig(32, 16, 8, 1, 1e-4, 1e-3, 2, 1, 0, 0, 1.0, 0.0, int(os.environ.get('DTYPE', '0')))
h = C.c_void_p(); assert l.avae_create(C.byref(cfg), 0, C.byref(h)) == 0
dev = torch.device('cuda', 0)
a_mc, b_nc, M, N, K = [int(x) for x in sys.argv[1:6]]
A = torch.randn((K, M) if a_mc else (M, K), device=dev)
B = torch.randn((K, N) if b_nc else (N, K), device=dev)
Cm = torch.zeros((M, N), device=dev)
for _ in range(5):
    assert l.avae_debug_gemm(h, a_mc, b_nc, A.data_ptr(), B.data_ptr(), Cm.data_ptr(), None, M, N, K, M if a_mc else K, N if b_nc else K, N, 1.0, 0, 1) == 0
torch.cuda.synchronize()
