"""a few training steps with library options set from the environment, for rocprofv3 runs of single kernels.
usage: [OPTS=bwd_rs=1,compact=0] [DTYPE=bf16] [RAGGED=1] [NSTEP=6] gru_steps.py [B S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argsim_amd import synth
from argsim_amd.model import VAE
B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 64)
m = VAE('train', seed=0, dtype=os.environ.get('DTYPE', 'f32'), dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
m.step = 20000
for kv in filter(None, os.environ.get('OPTS', '').split(',')):
    k, v = kv.split('=')
    m.set_option(k, int(v))
ids = torch.as_tensor(synth.batch(B, S, 8192, ragged=bool(os.environ.get('RAGGED')), seed=0, len_median=float(os.environ.get('LEN_MEDIAN', '24')),
                                  len_sigma=float(os.environ.get('LEN_SIGMA', '0.5')))).cuda()
for i in range(int(os.environ.get('NSTEP', '6'))):
    m.train_step(ids, ids, seed=i)
torch.cuda.synchronize()
print('losses', m.losses())
