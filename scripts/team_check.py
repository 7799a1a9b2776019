import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from argsim_amd import synth
from argsim_amd.model import VAE
m = VAE('train', seed=3, dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
ids = synth.batch(256, 64, 8192, ragged=True, seed=5)
m.step = 20000
m.set_option('persistent', 0); z0 = m.encode(ids); e0 = m.eval(ids, ids)
m.set_option('persistent', 1)
for item in (1, 2, 2):
    m.set_option('gru_item', item)
    z = m.encode(ids); e = m.eval(ids, ids)
    print('item', item, 'z equal', np.array_equal(z, z0), 'max diff', float(np.abs(z - z0).max()), 'eval equal', all(np.array_equal(a, b) for a, b in zip(e, e0)), flush=True)
m.set_option('gru_item', 2)
m.forward_backward(ids, ids, seed=9); print('losses', m.losses())
