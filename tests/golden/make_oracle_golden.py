"""writes tests/golden/oracle_<case>.npz: inputs and reduced outputs of the float64 oracle for the
parity cases of tests/helpers.py (losses, z, mu, pred, per-variable gradient norms).

    python tests/golden/make_oracle_golden.py

The reference itself cannot be run (TensorFlow absent, CudnnGRU GPU-only): these vectors pin the
ORACLE against drift, they are not reference outputs ("parity unpinned", oracle/__init__.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from helpers import BIG_CASES, CASES, PROD_CASES, big_extras, make_case  # noqa: E402
from oracle import vae_numpy as vn  # noqa: E402
from oracle import vae_torch as vt  # noqa: E402

for name in (CASES if len(sys.argv) == 1 else [n for n in sys.argv[1:] if n in CASES]):
    cfg, P, ids, keep, eps = make_case(name)
    o = vn.forward(P, cfg, ids, ids, 'train', 20000, keep, eps)
    out = dict(ids=ids, keep=keep, eps=eps, loss=o['loss'], loss_gen=o['loss_gen'], loss_kld=o['loss_kld'],
               z=o['z'], mu=o['mu'], lv=o['lv'], pred=o['pred'], loss_gen_samp=o['loss_gen_samp'])
    if name != 'full2':
        _, grads = vt.loss_and_grads(P, cfg, ids, ids, 20000, keep, eps)
        for k, g in grads.items():
            out['gnorm/' + k] = np.linalg.norm(g)
    np.savez_compressed(os.path.join(HERE, 'oracle_%s.npz' % name), **out)
    print(name, float(o['loss']))


def probe(name, shape):
    """a fixed pseudo-random direction per variable: <grad, probe> pins more than the norm does"""
    import zlib
    return np.random.default_rng(zlib.crc32(name.encode())).standard_normal(shape)


# production geometry: only reduced outputs (loss, mu, per-variable gradient norm and one projection each)
# (BIG_CASES, round 3: PIPE kernels at B = 512, R = 1024 / 512 through backward, the bench batch itself -- minutes of
#  float64 autograd each; only written when named on the command line or with --big)
big = [n for n in BIG_CASES if n in sys.argv[1:] or '--big' in sys.argv[1:]]
for name in (list(PROD_CASES) if len(sys.argv) == 1 else [n for n in sys.argv[1:] if n in PROD_CASES]) + big:
    cfg, P, ids, keep, eps = make_case(name)
    extra = big_extras(name) if name in BIG_CASES else {}
    outs, grads = vt.loss_and_grads(P, cfg, ids, ids, 20000, keep, eps, **extra)
    o = vn.forward(P, cfg, ids, ids, 'valid')
    out = dict(ids=ids, keep=keep, eps=eps, loss=outs['loss'], loss_gen=outs['loss_gen'], loss_kld=outs['loss_kld'], mu=o['mu'], lv=o['lv'])
    if name in BIG_CASES:
        out['loss_gen_samp'] = outs['loss_gen_samp'].astype(np.float32)      # per-token CE of the TRAIN forward (dropout + eps on)
        out['z'] = outs['z']
        for k in ('mu', 'lv', 'z'):            # (float32 is 6e-8 relative: far inside the 2e-5 tolerance, half the file)
            out[k] = out[k].astype(np.float32)
    for k, g in grads.items():
        out['gnorm/' + k] = np.linalg.norm(g)
        out['gdot/' + k] = float((g * probe(k, g.shape)).sum())
    np.savez_compressed(os.path.join(HERE, 'oracle_%s.npz' % name), **out)
    print(name, float(outs['loss']), flush=True)
