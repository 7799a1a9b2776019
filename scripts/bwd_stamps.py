"""phase stamps of the backward team kernels (diagnostic build: make -C argsim_amd/csrc clean && make DIAG=1)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argsim_amd import synth
from argsim_amd.model import VAE
m = VAE('train', seed=0, dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
m.step = 20000
names = ['top-loads', 'probe+head', 'stream+mfma', 'barrier1', 'write', 'barrier2', 'gates+stores', 'loop-edge']
for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '256').split(',')]:
    ids = torch.as_tensor(synth.batch(B, 64, 8192, seed=0)).cuda()
    for i in range(2): m.train_step(ids, ids, seed=i)
    for sel in (0, 16, 256, 256 | 16):                      # 0: the T = 4 launches (encoder at B = 256), 256: the T = 2 launches (decoder)
        m.set_option('gru_ablate', 128 | sel)
        out = (C.c_uint64 * 32)()
        m._l.avae_debug_stamps(m._h, out)
        for i in range(3): m.forward_backward(ids, ids, seed=5 + i)
        m._l.avae_debug_stamps(m._h, out)
        m.set_option('gru_ablate', 0)
        n, steps = out[16 + 10], out[16 + 8] / max(out[16 + 10], 1)
        per = [out[16 + i] / max(n, 1) / max(steps, 1) * 0.01 for i in range(8)]
        print('B', B, 'T', 2 if sel & 256 else 4, 'nopoll' if sel & 16 else 'poll', 'no-dgi' if sel & 512 else '', 'no-dgh' if sel & 1024 else '', 'coalesced-wrong' if sel & 8192 else '', 'bwd team-launch-teams', n, 'steps %.1f' % steps, ' '.join('%s %.2f' % (a, b) for a, b in zip(names, per)), 'total %.2f us/step' % sum(per), '| per launch-team: weights->LDS %.1f us, rendezvous %.1f us' % (out[16 + 11] / max(n, 1) * 0.01, out[16 + 12] / max(n, 1) * 0.01), flush=True)
