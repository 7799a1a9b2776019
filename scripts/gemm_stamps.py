"""phase stamps of the split-bf16 GEMM main loop (AVAE_F32S_ABLATE=16): mean time per K tile of wave 0, per phase."""
import ctypes as C, os, sys
os.environ.setdefault('AVAE_F32S_ABLATE', '16')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argsim_amd import lib
l = lib.load()
cfg = lib.AvaeConfig(32, 16, 8, 1, 1e-4, 1e-3, 2, 1, 0, 0, 1.0, 0.0, 2)
h = C.c_void_p(); assert l.avae_create(C.byref(cfg), 0, C.byref(h)) == 0
dev = torch.device('cuda', 0)
M, N, K = [int(x) for x in sys.argv[1:4]]
A = torch.randn((M, K), device=dev); B = torch.randn((N, K), device=dev); Cm = torch.zeros((M, N), device=dev)
st = torch.zeros(16, dtype=torch.int64, device=dev)
for it in range(3):
    st.zero_()
    assert l.avae_debug_gemm(h, 0, 0, A.data_ptr(), B.data_ptr(), Cm.data_ptr(), st.data_ptr(), M, N, K, K, K, N, 1.0, 0, 1) == 0
    torch.cuda.synchronize()
v = st.cpu().tolist(); n = max(v[5], 1)
names = ['wait loads (vmcnt 0)', 'split + ds_write', 'barrier 1', 'issue next loads', 'ds_read + mfma + barrier 2']
print('K tiles stamped', n)
for i in range(5): print('  %-28s %7.1f ns per K tile' % (names[i], v[i] * 10.0 / n))
print('  total %.1f ns per K tile' % (sum(v[:5]) * 10.0 / n))
