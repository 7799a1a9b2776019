"""experiment: team GRU kernels with several row blocks per workgroup (B >= 512) -- equality with the stepwise path, timing"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from argsim_amd import synth
from argsim_amd.model import VAE
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '512,1024').split(',')]:
    m = VAE('train', seed=0, dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
    m.step = 20000
    ids_np = synth.batch(B, S, 8192, ragged=True, seed=3)
    ids = torch.as_tensor(ids_np).cuda()
    m.set_option('persistent', 0)
    z_ref = m.encode(ids)
    m.forward_backward(ids, ids, seed=9)
    g_ref = m.grads.clone(); l_ref = m.losses()
    m.set_option('persistent', 1)
    P0 = m.params.clone()
    for turn, stag in ((0, 0), (0, 1), (0, 2), (0, 4), (0, 7)):
        m.params.copy_(P0); m.adam_m.zero_(); m.adam_v.zero_(); m.step = 20000
        m.set_option('gru_turn', turn); m.set_option('gru_stagger', stag)
        ok = np.array_equal(m.encode(ids), z_ref)
        m.forward_backward(ids, ids, seed=9)
        d = float((m.grads - g_ref).norm() / g_ref.norm())
        l = m.losses()
        for i in range(2): m.train_step(ids, ids, seed=i)
        torch.cuda.synchronize()
        m.set_option('timing', 1)
        t0 = time.perf_counter()
        for i in range(3): m.train_step(ids, ids, seed=10 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        tm = m.timing_collect(); m.set_option('timing', 0)
        print('B %d S %d stag %d  z-equal %s grad-diff %.1e loss %.6f/%.6f  step %.2f ms  ' % (B, S, stag, ok, d, l[2], l_ref[2], dt * 1e3) +
              '  '.join('%s %.3f ms %.1f TF' % (k, v[0] / 3, v[2] / (v[0] * 1e-3) / 1e12 if v[0] else 0) for k, v in tm.items()), flush=True)
    m.close()
