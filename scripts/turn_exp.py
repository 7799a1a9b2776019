"""experiment: MFMA turnstile modes of the D = 512 team GRU kernels (timing + equality with the stepwise path)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from argsim_amd import synth
from argsim_amd.model import VAE
m = VAE('train', seed=0, dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
m.step = 20000
ids = torch.as_tensor(synth.batch(256, 64, 8192, seed=0)).cuda()
m.set_option('persistent', 0)
z_ref = m.encode(ids)
m.forward_backward(ids, ids, seed=9)
g_ref = m.grads.clone()
m.set_option('persistent', 1)
for turn, stag in ((0, 0), (1, 0), (2, 0), (1, 3), (2, 3), (0, 3)):
    m.set_option('gru_turn', turn); m.set_option('gru_stagger', stag)
    ok = np.array_equal(m.encode(ids), z_ref)
    m.forward_backward(ids, ids, seed=9)
    d = float((m.grads - g_ref).norm() / g_ref.norm())
    for i in range(2): m.train_step(ids, ids, seed=i)
    torch.cuda.synchronize()
    m.set_option('timing', 1)
    t0 = time.perf_counter()
    for i in range(6): m.train_step(ids, ids, seed=10 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 6
    tm = m.timing_collect(); m.set_option('timing', 0)
    print('turn %d nb %d  z-equal %s grad-diff %.1e  step %.2f ms  ' % (turn, 3 if stag == 3 else 2, ok, d, dt * 1e3) +
          '  '.join('%s %.3f ms' % (k, v[0] / 6) for k, v in tm.items()), flush=True)
    print('   losses', m.losses(), flush=True)
