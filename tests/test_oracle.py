"""CPU tests of the oracle itself: it is pinned by what little the reference publishes (schedule
table, embedding-bound identity), by agreement of two independent restatements, by torch.nn.GRU, and
by the committed golden fixtures (tests/golden/oracle_*.npz, made by make_oracle_golden.py)."""
import os

import numpy as np
import pytest
import torch

from helpers import CASES, make_case
from oracle import vae_numpy as vn
from oracle import vae_torch as vt

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_schedule_table_from_reference_log():
    # reference docs/log.org:21-28: rate -> keepwd %, anneal % (rate = 1e-4 * step, +1 per 10k steps)
    table = {0: (50.00, 0.00), 1: (73.11, 76.16), 2: (88.08, 96.40), 3: (95.26, 99.51), 4: (98.20, 99.93), 5: (99.33, 99.99)}
    for rate, (keep, ann) in table.items():
        k, a, lr = vn.schedule(rate * 10000)
        assert round(100 * k, 2) == keep
        assert round(100 * a, 2) == ann
        # current code / paper: lr = 1e-3 / (sqrt(rate) + 1)  (model.py:80; the log's 'update' column is an older schedule)
        assert abs(lr - 1e-3 / (rate ** 0.5 + 1)) < 1e-15


def test_embedding_bound_is_glorot_for_scaled_tied_logits():
    # reference docs/log.org:86-88 + model.py:109-110,166: U(+-b), b = sqrt(6/(V/D+1)); scaled by D^-1/2 in the
    # logits this is glorot-uniform for a (D, V) kernel: b / sqrt(D) == sqrt(6 / (D + V))
    for V, D in ((8192, 512), (32, 16), (256, 64)):
        b = (6 / (V / D + 1)) ** 0.5
        assert abs(b / D ** 0.5 - (6 / (D + V)) ** 0.5) < 1e-12
    P = vn.init_params(vn.make_cfg(), 0)
    assert np.abs(P['embed/embedding']).max() <= (6 / (8192 / 512 + 1)) ** 0.5


def test_gru_equations_match_torch_nn_gru():
    rng = np.random.default_rng(0)
    D, In, S, B = 16, 12, 5, 3
    W, R = rng.standard_normal((3 * D, In)), rng.standard_normal((3 * D, D))
    bW, bR = rng.standard_normal(3 * D), rng.standard_normal(3 * D)
    x, h0 = rng.standard_normal((S, B, In)), rng.standard_normal((B, D))
    hs, hl = vn.gru(x, W, R, bW, bR, h0)
    g = torch.nn.GRU(In, D).double()
    with torch.no_grad():
        g.weight_ih_l0.copy_(torch.tensor(W)); g.weight_hh_l0.copy_(torch.tensor(R))
        g.bias_ih_l0.copy_(torch.tensor(bW)); g.bias_hh_l0.copy_(torch.tensor(bR))
        y, hn = g(torch.tensor(x), torch.tensor(h0)[None])
    assert np.abs(y.numpy() - hs).max() < 1e-12 and np.abs(hn.numpy()[0] - hl).max() < 1e-12


@pytest.mark.parametrize("name", ['tiny', 'mid'])
def test_numpy_and_torch_restatements_agree(name):
    cfg, P, ids, keep, eps = make_case(name)
    o = vn.forward(P, cfg, ids, ids, 'train', 20000, keep, eps)
    t, grads = vt.loss_and_grads(P, cfg, ids, ids, 20000, keep, eps)
    for k in ('loss', 'loss_gen', 'loss_kld'):
        assert abs(o[k] - t[k]) < 1e-12
    for k in ('mu', 'lv', 'z', 'logits', 'loss_gen_samp', 'loss_kld_samp', 'state_ex'):
        assert np.abs(o[k] - t[k]).max() < 1e-11, k
    assert np.array_equal(o['pred'], t['pred'])
    assert all(np.isfinite(g).all() for g in grads.values())
    assert grads['embed/embedding'].shape == P['embed/embedding'].shape


def test_mask_shapes_and_token_count():
    cfg, P, ids, keep, eps = make_case('tiny')
    o = vn.forward(P, cfg, ids, ids, 'valid')
    lens = (ids != cfg['eos']).sum(1)
    assert o['msk_tgt'].sum() == (lens + 1).sum() == len(o['labels'])      # row b has len_b + 1 targets
    assert (o['lead'][0] == cfg['bos']).all() and (o['gold'][-1] == cfg['eos']).all()
    # padding never reaches z: change the padding region, z must not move
    ids2 = np.concatenate([ids, np.full((len(ids), 3), cfg['eos'], ids.dtype)], 1)
    assert np.array_equal(vn.forward(P, cfg, ids2, ids2, 'valid')['z'], o['z'])


def test_gradient_by_finite_differences():
    cfg, P, ids, keep, eps = make_case('tiny')
    _, grads = vt.loss_and_grads(P, cfg, ids, ids, 20000, keep, eps)
    rng = np.random.default_rng(1)
    for name in ('embed/embedding', 'encode/rnn2/bwd/R', 'latent/lv/kernel', 'decode/rnn/l3/W', 'decode/out/bias'):
        idx = tuple(rng.integers(0, s) for s in P[name].shape)
        h = 1e-5
        Pp = {k: v.copy() for k, v in P.items()}; Pm = {k: v.copy() for k, v in P.items()}
        Pp[name][idx] += h; Pm[name][idx] -= h
        fd = (vn.forward(Pp, cfg, ids, ids, 'train', 20000, keep, eps)['loss'] - vn.forward(Pm, cfg, ids, ids, 'train', 20000, keep, eps)['loss']) / (2 * h)
        assert abs(fd - grads[name][idx]) < 1e-7 + 1e-5 * abs(fd), (name, fd, grads[name][idx])


def test_adam_tf_first_step_is_lr_sized():
    p = {'w': np.array([1.0, -2.0])}; g = {'w': np.array([0.5, -0.25])}
    z = {'w': np.zeros(2)}
    p1, m1, v1 = vn.adam_tf(p, g, z, z, 0, 1e-3)
    # first TF-Adam step: lr_t * m / (sqrt(v) + eps) with m = 0.1 g, v = 0.001 g^2 -> ~ lr * sign(g)
    assert np.allclose(p['w'] - p1['w'], 1e-3 * np.sign(g['w']), rtol=1e-6)


@pytest.mark.parametrize("name", list(CASES))
def test_golden_fixtures(name):
    """committed oracle outputs: guards the oracle (and thereby every parity test) against silent edits"""
    path = os.path.join(GOLD, 'oracle_%s.npz' % name)
    with np.load(path, allow_pickle=False) as f:
        gold = {k: f[k] for k in f.files}
    cfg, P, ids, keep, eps = make_case(name)
    assert np.array_equal(gold['ids'], ids)
    o = vn.forward(P, cfg, ids, ids, 'train', 20000, keep, eps)
    for k in ('loss', 'loss_gen', 'loss_kld'):
        assert abs(float(gold[k]) - o[k]) <= 1e-12 * max(1.0, abs(o[k])), k
    assert np.abs(gold['z'] - o['z']).max() < 1e-12
    assert np.abs(gold['mu'] - o['mu']).max() < 1e-12
    assert np.array_equal(gold['pred'], o['pred'])
    if name != 'full2':
        _, grads = vt.loss_and_grads(P, cfg, ids, ids, 20000, keep, eps)
        for k, g in grads.items():
            assert abs(float(gold['gnorm/' + k]) - np.linalg.norm(g)) <= 1e-9 * max(1.0, np.linalg.norm(g)), k
