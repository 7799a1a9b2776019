"""per-class GRU / GEMM times for values of one option (HIP events), alternating.  usage: ab_classes.py KEY [B S]; VALS=0,1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argsim_amd import synth
from argsim_amd.model import VAE
key = sys.argv[1]
B, S = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 64)
VALS = [int(x) for x in os.environ.get('VALS', '0,1').split(',')]
m = VAE('train', seed=0, dtype=os.environ.get('DTYPE', 'f32'), dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
m.step = 20000
ids = torch.as_tensor(synth.batch(B, S, 8192, ragged=bool(os.environ.get('RAGGED')), seed=0)).cuda()
for i in range(3): m.train_step(ids, ids, seed=i)
for rep in range(2):
    for v in VALS:
        m.set_option(key, v)
        for i in range(2): m.train_step(ids, ids, seed=i)
        m.set_option('timing', 1)
        for i in range(6): m.train_step(ids, ids, seed=50 + i)
        tm = m.timing_collect(); m.set_option('timing', 0)
        print(key, v, '  '.join('%s %.3f ms' % (k, x[0] / 6) for k, x in tm.items()), 'loss %.4f' % m.losses()[2], flush=True)
