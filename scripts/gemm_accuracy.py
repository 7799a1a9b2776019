"""accuracy of the three GEMM modes against float64 products of the same fp32 operands (avae_debug_gemm hook):
max and rms error relative to sum_k |a||b| (the natural scale of fp32 accumulation error)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from argsim_amd import lib
l = lib.load()
dev = torch.device('cuda', 0)
rng = np.random.default_rng(0)
for (a_mc, b_nc, M, N, K) in ((0, 0, 256, 384, 512), (0, 1, 384, 256, 8192), (1, 1, 256, 256, 16640), (1, 0, 128, 256, 1024)):
    for dist in ('normal', 'lognormal'):
        A = rng.standard_normal((K, M) if a_mc else (M, K)).astype(np.float32)
        B = rng.standard_normal((K, N) if b_nc else (N, K)).astype(np.float32)
        if dist == 'lognormal':
            A *= np.exp(2 * rng.standard_normal(A.shape)).astype(np.float32); B *= np.exp(2 * rng.standard_normal(B.shape)).astype(np.float32)
        A64 = (A.T if a_mc else A).astype(np.float64); B64 = (B if b_nc else B.T).astype(np.float64)
        ref = A64 @ B64; scale = np.abs(A64) @ np.abs(B64)
        At, Bt = torch.tensor(A, device=dev), torch.tensor(B, device=dev)
        row = []
        for dt in (0, 2, 1):
            cfg = lib.AvaeConfig(32, 16, 8, 1, 1e-4, 1e-3, 2, 1, 0, 0, 1.0, 0.0, dt)
            h = C.c_void_p(); assert l.avae_create(C.byref(cfg), 0, C.byref(h)) == 0
            Ct = torch.zeros((M, N), device=dev)
            rc = l.avae_debug_gemm(h, a_mc, b_nc, At.data_ptr(), Bt.data_ptr(), Ct.data_ptr(), None, M, N, K, M if a_mc else K, N if b_nc else K, N, C.c_float(1.0), 0, 1)
            assert rc == 0, l.avae_last_error(h)
            torch.cuda.synchronize()
            e = (Ct.cpu().numpy().astype(np.float64) - ref) / scale
            row.append('%s max %.2e rms %.2e' % (('f32 ', 'bf16', 'f32s')[dt], np.abs(e).max(), np.sqrt((e * e).mean())))
            l.avae_destroy(h)
        print('a_mc %d b_nc %d M %d N %d K %5d %-9s | ' % (a_mc, b_nc, M, N, K, dist) + ' | '.join(row), flush=True)
