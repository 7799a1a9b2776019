"""bf16 NT GEMM forms on the shapes of BASELINE configs[2] (B 1024 x S 128) through the avae_debug_gemm hook: the register-staged
256x256 kernel (bf16_nt8 = 0) against the phased LDS-DMA kernel (1).  The times include the fp32 -> bf16 conversion passes of the
hook; run under rocprofv3 --kernel-trace --stats for the kernels alone.  usage: gemm_bf16_bench.py [name prefix]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argsim_amd import lib
l = lib.load()
cfg = lib.AvaeConfig(32, 16, 8, 1, 1e-4, 1e-3, 2, 1, 0, 0, 1.0, 0.0, 1)
h = C.c_void_p(); assert l.avae_create(C.byref(cfg), 0, C.byref(h)) == 0
dev = torch.device('cuda', 0)
shapes = [  # name, a_mc, b_nc, M, N, K
    ('cube 8192', 0, 0, 8192, 8192, 8192),
    ('cube 4096', 0, 0, 4096, 4096, 4096),
    ('logits', 0, 0, 132096, 8192, 512),
    ('enc gi L2', 0, 0, 131072, 3072, 1024),
    ('dec gi', 0, 0, 132096, 1536, 512),
    ('dho', 0, 1, 132096, 512, 8192),
    ('enc dX', 0, 1, 131072, 1024, 3072),
    ('dec dX', 0, 1, 132096, 512, 1536),
]
tn_shapes = [  # weight gradients C (M x N) += A^T B over K = tokens rows
    ('tn dE', 8192, 512, 132096),
    ('tn enc dW', 3072, 1024, 131072),
    ('tn dec dW', 1536, 512, 132096),
]
if len(sys.argv) > 1: shapes = [x for x in shapes if x[0].startswith(sys.argv[1])]; tn_shapes = [x for x in tn_shapes if x[0].startswith(sys.argv[1])]
for name, a_mc, b_nc, M, N, K in shapes:
    A = torch.randn((K, M) if a_mc else (M, K), device=dev)
    B = torch.randn((K, N) if b_nc else (N, K), device=dev)
    Cm = torch.zeros((M, N), device=dev)
    lda = M if a_mc else K; ldb = N if b_nc else K
    res = []
    outs = []
    for form in (0, 1):
        assert l.avae_set_option(h, b'bf16_nt8', form) == 0
        def run():
            assert l.avae_debug_gemm(h, a_mc, b_nc, A.data_ptr(), B.data_ptr(), Cm.data_ptr(), None, M, N, K, lda, ldb, N, 1.0, 0, 1) == 0
        for _ in range(2): run()
        torch.cuda.synchronize()
        outs.append(Cm[:4096].clone())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        res.append('%8.1f us %7.1f TFLOP/s' % (ms * 1e3, 2.0 * M * N * K / ms / 1e9))
    d = float((outs[0] - outs[1]).abs().max()); s = float(outs[0].abs().max())
    print('%-12s M %6d N %5d K %5d | nt256 %s | p8 %s | max diff %.2e of %.1f' % (name, M, N, K, res[0], res[1], d, s), flush=True)

for name, M, N, K in tn_shapes:
    A = torch.randn((K, M), device=dev); B = torch.randn((K, N), device=dev)
    res = []; outs = []
    for form in (0, 1):
        assert l.avae_set_option(h, b'bf16_nt8', form) == 0
        Cm = torch.zeros((M, N), device=dev)
        def run():
            assert l.avae_debug_gemm_tn16(h, A.data_ptr(), B.data_ptr(), Cm.data_ptr(), M, N, K, M, N, N, 1.0) == 0
        for _ in range(2): run()
        torch.cuda.synchronize()
        outs.append(Cm.clone())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        res.append('%8.1f us %7.1f TFLOP/s' % (ms * 1e3, 2.0 * M * N * K / ms / 1e9))
    d = float((outs[0] - outs[1]).abs().max()); sc = float(outs[0].abs().max())
    print('%-12s M %6d N %5d K %6d | tn256 %s | p8 %s | max diff %.2e of %.1f' % (name, M, N, K, res[0], res[1], d, sc), flush=True)
