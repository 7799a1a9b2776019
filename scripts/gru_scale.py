"""GRU class timings vs batch (how much of a step is contention between the two workgroups of a CU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argsim_amd import synth
from argsim_amd.model import VAE
m = VAE('train', seed=0, dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
m.step = 20000
for B in (64, 128, 256, 512):
    ids = torch.as_tensor(synth.batch(B, 64, 8192, seed=0)).cuda()
    for i in range(3): m.train_step(ids, ids, seed=i)
    m.losses(); m.set_option('timing', 1)
    n = 5
    for i in range(n): m.train_step(ids, ids, seed=i)
    t = m.timing_collect(); m.set_option('timing', 0)
    print('B %4d  gru_fwd %.2f ms  gru_bwd %.2f ms  gemm %.2f ms   fwd us/step %.2f  bwd us/step %.2f' % (
        B, t['gru_fwd'][0] / n, t['gru_bwd'][0] / n, t['gemm'][0] / n, t['gru_fwd'][0] / n / 387 * 1e3, t['gru_bwd'][0] / n / 387 * 1e3), flush=True)
