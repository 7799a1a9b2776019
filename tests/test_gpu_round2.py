"""GPU tests added in round 2: the embedding drivers, production-geometry kernels against committed oracle fixtures,
the sentinel-alias incident, per-rank RNG keys and side-stream work across a whole backward."""
import os
import time

import numpy as np
import pytest

from helpers import make_case, rel_l2
from oracle import vae_numpy as vn

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ('dim_tgt', 'dim_emb', 'dim_rep', 'rnn_layers', 'accelerate', 'learn_rate', 'bos', 'eos')


def _vae(cfg, P, mode='train', **kw):
    from argsim_amd.model import VAE
    m = VAE(mode, init=False, **{k: cfg[k] for k in KEYS}, **kw)
    m.set_params(P)
    return m


def _probe(name, shape):
    import zlib
    return np.random.default_rng(zlib.crc32(name.encode())).standard_normal(shape)


# ------------------------------------------------------------------------------------------ SURVEY 8(f) row 1
@pytest.fixture(scope='module')
def tiny_vocab(tmp_path_factory):
    from argsim_amd import util_sp
    d = tmp_path_factory.mktemp('spm')
    rng = np.random.default_rng(1)
    words = ['argument', 'stance', 'abortion', 'rights', 'gun', 'control', 'people', 'think', 'because', 'evidence',
             'the', 'a', 'of', 'and', 'is', 'not', 'that', 'should', 'we', 'they', 'law', 'state', 'debate', 'claim']
    lines = [' '.join(' '.join(rng.choice(words, int(rng.integers(3, 9)))) + '.' for _ in range(int(rng.integers(1, 5))))
             for _ in range(300)]
    path = os.path.join(d, 'train.txt')
    with open(path, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    return util_sp.spm(os.path.join(d, 'vocab'), path, size=48), lines, os.path.join(d, 'vocab.model')


def test_embedding_drivers_match_the_oracle(tiny_vocab):
    """eval_embed_reason.py:33-41 (`embed`: encode_capped -> vpack -> z per partition of 128 rows) and :47-51
    (`infer_avg`: mean z over sampled segmentations) on a tiny SentencePiece vocabulary: the arrays the driver writes
    equal the oracle's mu on the id arrays the driver built."""
    from argsim_amd import eval_embed, util_sp
    from argsim_amd.util_np import vpack
    vocab, lines, _ = tiny_vocab
    cfg = vn.make_cfg(dim_tgt=48, dim_emb=16, dim_rep=8, rnn_layers=2)
    P = vn.init_params(cfg, 3, bias_scale=0.1)
    P = {k: v.astype(np.float32).astype(np.float64) for k, v in P.items()}
    m = _vae(cfg, P, mode='infer')
    text = lines[:150]                                       # two partitions: 128 + 22 rows
    z = eval_embed.embed(m, vocab, text, batch=128)
    assert z.shape == (150, 8) and z.dtype == np.float32     # what eval_classification*.py np.load()s
    ids = [util_sp.encode_capped(vocab, t) for t in text]
    ids = vpack(ids, (len(ids), max(map(len, ids))), vocab.eos_id(), np.int32)
    want = np.concatenate([vn.forward(P, cfg, ids[i:j], ids[i:j], 'valid')['mu'] for i, j in ((0, 128), (128, 150))])
    assert np.abs(z - want).max() <= 2e-5
    # infer_avg draws its segmentations from SentencePiece's own RNG: compare on the very batch it encoded
    seen = []
    enc = m.encode
    m.encode = lambda x: (seen.append(np.array(x)), enc(x))[1]
    avg = eval_embed.infer_avg(m, vocab, text[0], samples=16)
    m.encode = enc
    assert avg.shape == (8,) and len(seen) == 1 and seen[0].shape[0] == 16
    want = vn.forward(P, cfg, seen[0], seen[0], 'valid')['mu'].mean(axis=0)
    assert np.abs(avg - want).max() <= 2e-5


def test_embedding_driver_main_writes_the_npy_files(tiny_vocab, tmp_path):
    """`python -m argsim_amd.eval_embed`: checkpoint -> (n, R) float32 .npy + the sampled-average twin"""
    import json
    from argsim_amd import ckpt, eval_embed
    vocab, lines, vocab_path = tiny_vocab
    cfgm = dict(dim_tgt=48, dim_emb=16, dim_rep=8, rnn_layers=2)
    cfg = vn.make_cfg(**cfgm)
    P = vn.init_params(cfg, 4, bias_scale=0.1)
    m = _vae(cfg, P, mode='infer')
    ckpt.save(m, str(tmp_path / 'model'), slots=False)
    with open(tmp_path / 'config.json', 'w') as f:
        json.dump({'model': cfgm}, f)
    np.savez(tmp_path / 'data.npz', posts=np.array(lines[:9]))
    eval_embed.main(['--ckpt', str(tmp_path / 'model'), '--vocab', vocab_path, '--data', str(tmp_path / 'data.npz'),
                     '--out', str(tmp_path / 'emb'), '--config', str(tmp_path / 'config.json'), '--samples', '4'])
    z, zs = np.load(tmp_path / 'emb.npy'), np.load(tmp_path / 'emb_sample.npy')
    assert z.shape == zs.shape == (9, 8) and z.dtype == zs.dtype == np.float32
    assert np.abs(z - eval_embed.embed(m, vocab, lines[:9])).max() == 0.0


# ------------------------------------------------------------------------------------------ production geometry
def _gold(name):
    with np.load(os.path.join(HERE, 'golden', 'oracle_%s.npz' % name), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def test_production_kernels_against_oracle_fixture():
    """B=64, S=64, D=512, V=8192, R=128, ragged: one full 64-row block per GRU workgroup, so the default path runs the
    LDS-weight team kernels (encoder forward and backward), the register-form decoder kernels, the fast-staging and
    split-K GEMMs.  Compared with the committed float64-oracle fixture ONLY (no live oracle at this size)."""
    gold = _gold('prod64')
    cfg, P, ids, keep, eps = make_case('prod64')
    assert np.array_equal(ids, gold['ids']) and np.array_equal(keep, gold['keep'])
    m = _vae(cfg, P)
    m.step = 20000
    z, lv = m.encode(ids, return_lv=True)
    assert np.abs(z - gold['mu']).max() <= 2e-5 and np.abs(lv - gold['lv']).max() <= 2e-5
    m.forward_backward(ids, ids, keep_mask=keep, eps=eps)
    lg, lk, lo = m.losses()
    assert abs(lo - float(gold['loss'])) <= 2e-5 * abs(float(gold['loss']))
    assert abs(lk - float(gold['loss_kld'])) <= 2e-5 * abs(float(gold['loss_kld'])) + 1e-7
    for k, g in m.get_grads().items():
        g = g.astype(np.float64)
        n = float(gold['gnorm/' + k])
        assert abs(np.linalg.norm(g) - n) <= 2e-4 * n, k
        # one fixed random projection per variable: |<g, p> - gold| <= tol * |g| |p|
        p = _probe(k, g.shape)
        assert abs(float((g * p).sum()) - float(gold['gdot/' + k])) <= 2e-4 * n * np.linalg.norm(p), k


def test_production_kernels_seq128_bf16_tolerance():
    """B=64, S=128 (BASELINE configs[2]'s sequence length) against its fixture: the exact-fp32 path at the fp32
    tolerances, the bf16-operand mode at the bf16 tolerances stated in DESIGN.md (z 3e-2, loss 1e-2 rel, gradient
    norms 5e-2 rel)."""
    gold = _gold('prod128')
    cfg, P, ids, keep, eps = make_case('prod128')
    for dtype, tz, tl, tg in (('f32', 2e-5, 2e-5, 2e-4), ('bf16', 3e-2, 1e-2, 5e-2)):
        m = _vae(cfg, P, dtype=dtype)
        m.step = 20000
        assert np.abs(m.encode(ids) - gold['mu']).max() <= tz, dtype
        m.forward_backward(ids, ids, keep_mask=keep, eps=eps)
        lo = m.losses()[2]
        assert abs(lo - float(gold['loss'])) <= tl * abs(float(gold['loss'])), dtype
        for k, g in m.get_grads().items():
            g = g.astype(np.float64)
            n = float(gold['gnorm/' + k])
            assert abs(np.linalg.norm(g) - n) <= tg * n, (dtype, k)
            p = _probe(k, g.shape)              # one fixed random projection per variable (the fixture's gdot/*)
            assert abs(float((g * p).sum()) - float(gold['gdot/' + k])) <= tg * n * np.linalg.norm(p), (dtype, k)
        m.close()


# ------------------------------------------------------------------------------------------ the round-1 time-out
def test_sentinel_bit_pattern_in_a_parameter_cannot_stall_the_exchange():
    """round-1 incident (gpurun_out/gru_stamps.log): a NaN carrying the exchange's "not yet written" pattern
    0xFFFFFFFF reached the exchanged data and every consumer spun to its 2 s bound.  Exchanged stores now replace that
    one pattern by the canonical NaN: a parameter poisoned with it gives NaN losses at once and no time-out, at the
    benchmark geometry (team kernels) and on a small case (generic kernels)."""
    import torch
    from argsim_amd import synth
    from argsim_amd.model import VAE
    bad = np.array([0xFFFFFFFF], np.uint32).view(np.float32)[0]
    for kw, B, S in ((dict(dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3), 256, 64),
                     (dict(dim_tgt=256, dim_emb=64, dim_rep=32, rnn_layers=3), 8, 12)):
        m = VAE('train', seed=0, **kw)
        m.step = 20000
        ids = synth.batch(B, S, kw['dim_tgt'], seed=0)
        m.train_step(ids, ids, seed=1)                        # warm: workspace, code objects
        assert all(np.isfinite(m.losses()))
        for name in ('encode/rnn1/fwd/R', 'decode/rnn/l2/R'):
            R = m.get_tensor(name)
            R[5, 7] = bad
            m.set_tensor(name, R)
            assert m.get_tensor(name).view(np.uint32)[5, 7] == 0xFFFFFFFF     # the payload survives the upload
        before = m.params.clone()
        t0 = time.perf_counter()
        m.train_step(ids, ids, seed=2)
        losses = m.losses()                                   # raises on a time-out
        dt = time.perf_counter() - t0
        assert dt < 1.0, dt                                   # a stalled wait costs 2 s each
        assert all(np.isnan(x) for x in (losses[0], losses[2]))
        torch.cuda.synchronize()
        assert not torch.equal(before, m.params)              # an honest (NaN) update, not a skipped one
        m.close()


def test_diagnostic_ablations_are_not_in_the_production_library():
    from argsim_amd import synth
    from argsim_amd.model import VAE
    m = VAE('train', seed=0, dim_tgt=256, dim_emb=64, dim_rep=32, rnn_layers=2)
    try:
        m.set_option('gru_ablate', 16)
    except RuntimeError as e:
        assert 'diagnostic build' in str(e)
    else:                                                     # a DIAG=1 build: allowed, and reversible
        m.set_option('gru_ablate', 0)
    ids = synth.batch(8, 12, 256, seed=0)
    m.train_step(ids, ids, seed=1)
    assert all(np.isfinite(m.losses()))


# ------------------------------------------------------------------------------------------ data parallel
def test_rank_keys_give_each_shard_its_own_draws(tmp_path):
    """with seed=None every rank derived the same key, so row k of every shard shared its word-dropout mask
    and eps.  DataParallel.train_step now folds the rank in: same ids + same model state, ranks 0 and 1 -> different
    draws (different loss), and the key a rank uses is reproducible."""
    import torch
    import torch.distributed as dist
    from argsim_amd.dist import DataParallel, rank_seed
    cfg, P, ids, keep, eps = make_case('mid')
    a, b, c = _vae(cfg, P), _vae(cfg, P), _vae(cfg, P)
    for m in (a, b, c):
        m.step = 20000
    if not dist.is_initialized():
        dist.init_process_group('nccl', init_method='file://%s' % (tmp_path / 'rdv'), rank=0, world_size=1,
                                device_id=torch.device('cuda', 0))
    try:
        n_glob = float((ids != cfg['eos']).sum() + len(ids))
        key = a.next_seed()
        dp = DataParallel(a)
        dp.train_step(ids, ids, n_glob, float(len(ids)))                       # seed=None on rank 0
        b.train_step(ids, ids, seed=rank_seed(key, 0))
        c.train_step(ids, ids, seed=rank_seed(key, 1))                         # what rank 1 would have drawn
        la, lb, lc = a.losses(), b.losses(), c.losses()
        # (the data-parallel call scales by 1/N_global passed in, the plain call by its own N: last-bit differences)
        assert abs(la[0] - lb[0]) <= 1e-6 * abs(lb[0]) and abs(la[0] - lc[0]) > 1e-4 * abs(lc[0])
        assert abs(la[1] - lc[1]) <= 1e-6 * abs(lc[1])                         # KL depends on mu, lv only: no draw in it
        assert float((a.params - b.params).abs().max()) <= 1e-6
    finally:
        dist.destroy_process_group()


def test_side_stream_traffic_across_backward_is_fenced_from_persistent_launches():
    """VERDICT r1 item 4.  A bandwidth-bound kernel is launched on a side stream for every gradient bucket through the
    data-parallel hook (as the RCCL all-reduce is), at the benchmark geometry where every GRU launch is a persistent
    kernel that needs all CUs.  The library announces a bucket after the NEXT persistent launch is enqueued and asks
    for a fence before each one: the step must equal the plain step, the side-stream work must really overlap backward
    (HIP-event timestamps) and must never be in flight when a persistent launch starts."""
    import torch
    from argsim_amd import synth
    from argsim_amd.model import VAE
    kw = dict(dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3, seed=0)
    ids = torch.as_tensor(synth.batch(256, 64, 8192, seed=2)).cuda()
    a, b = VAE('train', **kw), VAE('train', **kw)
    a.step = b.step = 20000
    side = torch.cuda.Stream()
    src = torch.empty(64 << 20, dtype=torch.float32, device='cuda').normal_()       # 256 MB: ~0.1 ms per copy pass
    dst = torch.empty_like(src)
    log = []

    def hook(bucket, off, cnt):
        cur = torch.cuda.current_stream()
        if bucket < 0:                                   # fence: compute waits for the side stream
            cur.wait_stream(side)
            log.append(('fence', None, None))
            return
        ev = torch.cuda.Event()
        ev.record(cur)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(side)
            for _ in range(4):
                dst.copy_(src)
            b.grads[off:off + cnt].mul_(1.0)              # touches the bucket, as a reduction would
            e.record(side)
        log.append(('bucket', s, e))

    for m in (a, b):
        m.train_step(ids, ids, seed=3)                    # warm-up
    b.set_grad_hook(hook)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.forward_backward(ids, ids, seed=4)
    t0.record()
    b.forward_backward(ids, ids, seed=4)
    t1.record()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    d = float((a.grads - b.grads).norm() / a.grads.norm())
    assert d <= 1e-5, d                                   # float-atomic ordering only
    kinds = [k for k, _, _ in log]
    assert kinds.count('bucket') == len(b.buckets()) and kinds.count('fence') == 6      # 3 decoder + 3 encoder BPTT launches
    assert kinds[0] == 'fence'                            # nothing is announced before the first persistent launch
    span = t0.elapsed_time(t1)                            # forward + backward as enqueued on the compute stream
    inside = [(t0.elapsed_time(s), t0.elapsed_time(e)) for k, s, e in log if k == 'bucket']
    # the side-stream work of all but the last buckets starts before the compute stream has finished backward.  (The last TWO -- the
    # embedding and encode/rnn1 -- are announced within the last 0.3 ms of backward and queue behind the stand-in copies of the buckets
    # before them, 4 x 256 MB each: whether they start inside the span is a matter of tens of microseconds and of the box; round 4 saw
    # 7 of 9 inside on one box.  What is asserted is that the overlap exists: at least six of the nine start inside backward.)
    assert sum(1 for s, e in inside if s < span) >= len(inside) - 3, (span, inside)
    a.adam_step(); b.adam_step()
    assert all(np.isfinite(b.losses()))


# ------------------------------------------------------------------------------------------ several row blocks per workgroup
@pytest.mark.parametrize("B,S", [(512, 20), (1024, 12)])
def test_row_block_pipelined_team_kernels_equal_stepwise(B, S):
    """B >= 512 at D = 512: every GRU launch (encoder AND decoder layers, forward and backward, incl. the decoder's
    h0 / dh0 tail) runs the LDS-weight team kernels with 2-4 row blocks per workgroup, whose next operand is fetched
    behind the current item's MFMAs.  Same arithmetic as one launch per step: z bit for bit, gradients up to
    float-atomic order, on a ragged batch."""
    import torch
    from argsim_amd import synth
    from argsim_amd.model import VAE
    m = VAE('train', seed=1, dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
    m.step = 20000
    ids = synth.batch(B, S, 8192, ragged=True, seed=7)
    out = {}
    for persistent in (1, 0):
        m.set_option('persistent', persistent)
        z = m.encode(ids)
        ev = m.eval(ids, ids)
        m.forward_backward(ids, ids, seed=5)
        out[persistent] = (z, ev, m.grads.clone(), m.losses())
    assert np.array_equal(out[1][0], out[0][0])
    for x, y in zip(out[1][1], out[0][1]):
        assert np.array_equal(x, y)
    d = float((out[1][2] - out[0][2]).norm() / out[0][2].norm())
    assert d < 1e-5, d
    assert all(np.isfinite(out[1][3]))


@pytest.mark.parametrize("B,S", [(256, 16), (512, 20), (1024, 12)])
def test_bf16_mode_team_kernels_track_the_register_form(B, S):
    """bf16-operand mode (configs[2]) at D = 512: the team kernels -- bf16 weights in LDS, v_mfma_f32_16x16x32_bf16, the
    16-bit exchange whose loaded piece is the MFMA operand, h0 seeded into slot 0, fp32 h_prev carried in registers; one row
    block per workgroup (B = 256) and 2-4 pipelined row blocks (B >= 512) -- against one launch per step of the register-form
    kernels, which round the same two operands to bf16 in fp32 registers.  Same products, another summation order, and
    a last-bit difference in h can flip its bf16 rounding at the next step: agreement to bf16 accuracy (z 1e-2 abs,
    losses 2e-3 rel, gradients 3e-2 rel-L2), not bit for bit.  The 16-bit exchange can be switched off (gru_bf16 = 0:
    fp32 recurrence): the two modes of the recurrence agree at the same level."""
    from argsim_amd import synth
    from argsim_amd.model import VAE
    m = VAE('train', seed=1, dtype='bf16', dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
    m.step = 20000
    ids = synth.batch(B, S, 8192, ragged=True, seed=7)
    out = {}
    for key, (persistent, gru_bf16) in {'team': (1, 1), 'stepwise': (0, 1), 'fp32rec': (1, 0)}.items():
        m.set_option('persistent', persistent)
        m.set_option('gru_bf16', gru_bf16)
        z = m.encode(ids)
        m.forward_backward(ids, ids, seed=5)
        out[key] = (z, m.grads.clone(), m.losses())
    for other in ('stepwise', 'fp32rec'):
        assert np.isfinite(out['team'][0]).all() and np.abs(out['team'][0] - out[other][0]).max() <= 1e-2, other
        assert abs(out['team'][2][2] - out[other][2][2]) <= 2e-3 * abs(out[other][2][2]), other
        d = float((out['team'][1] - out[other][1]).norm() / out[other][1].norm())
        assert d < 3e-2, (other, d)
    m.close()


@pytest.mark.parametrize("dtype", ['f32', 'bf16'])
def test_table_fed_layers_equal_the_per_token_projection(dtype):
    """Benchmark geometry (256 x 64 Zipf ids over 8192: more tokens than vocabulary entries): encoder layer 1 and decoder
    layer 1 project only the ids present in the batch and gather / sum by id -- against the same library with
    the per-token GEMMs (table_l1 = 0).  Forward: the same products in the same order per output element, so z and the
    per-token losses agree to the last bits; backward: the same products grouped by id, another summation order."""
    import torch
    from argsim_amd import synth
    from argsim_amd.model import VAE
    m = VAE('train', seed=3, dtype=dtype, dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
    m.step = 20000
    ids = synth.batch(256, 64, 8192, seed=4)
    out = {}
    for table in (1, 0):
        m.set_option('table_l1', table)
        z = m.encode(ids)
        ev = m.eval(ids, ids)
        m.forward_backward(ids, ids, seed=5)
        out[table] = (z, ev, m.grads.clone(), m.losses(), m.get_grads()['embed/embedding'])
    tz, tg = (1e-6, 2e-5) if dtype == 'f32' else (1e-2, 3e-2)      # (bf16: a last-bit difference can flip an operand rounding)
    assert np.isfinite(out[1][0]).all() and np.abs(out[1][0] - out[0][0]).max() <= tz
    assert np.abs(np.asarray(out[1][1][1]) - np.asarray(out[0][1][1])).max() <= (1e-5 if dtype == 'f32' else 5e-2)      # per-token CE
    d = float((out[1][2] - out[0][2]).norm() / out[0][2].norm())
    assert d < tg, d
    # the embedding gradient on its own (the part that changes route: rows added by uid instead of scatter-added by token)
    assert np.linalg.norm(out[1][4] - out[0][4]) < tg * np.linalg.norm(out[0][4])
    m.close()


# ------------------------------------------------------------------------------------------ SURVEY 8(f) row 2
@pytest.mark.parametrize("b", [1, 5, 32])
def test_persistent_greedy_decode_equals_the_per_token_loop(b):
    """model.py:204-219 in ONE persistent launch (decode.hip) against the launch-per-token loop of the same library at
    the production dimensions (weights resident in LDS, grid barrier per phase): same ids, same early stop."""
    from argsim_amd.model import VAE
    m = VAE('infer', seed=2, dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
    rng = np.random.default_rng(b)
    z = rng.standard_normal((b, 128)).astype(np.float32)
    y1 = m.decode(z, steps=40)
    m.set_option('persistent', 0)
    y0 = m.decode(z, steps=40)
    assert y1.shape == y0.shape and np.array_equal(y1, y0)
    # early stop: a bias that makes eos win at once -> the loop ends before appending anything (model.py:217-218)
    bias = m.get_tensor('decode/out/bias')
    E = m.get_tensor('embed/embedding')
    m.set_tensor('decode/out/bias', bias + 200.0 * E[1] / np.linalg.norm(E[1]))
    m.set_option('persistent', 1)
    assert m.decode(z, steps=40).shape == (b, 0)
    m.set_option('persistent', 0)
    assert m.decode(z, steps=40).shape == (b, 0)


# ------------------------------------------------------------------------------------------ --profile / --save-tf
def test_profile_flag_produces_a_kernel_profile_and_save_tf_writes_bundles(tmp_path):
    """train.py --profile (src/train.py:76-82): the reference writes ONE traced run of the validation loss on valid[:32];
    here a child process runs it under rocprofv3 (when present) and the per-kernel summary lands in <log>/<trial>_profile/.
    --save-tf writes TF V2 checkpoint files beside the native one."""
    import glob
    import json
    import shutil
    import subprocess
    import sys
    from argsim_amd import util_sp
    from argsim_amd.util_np import vpack
    rng = np.random.default_rng(0)
    words = ['argument', 'stance', 'abortion', 'rights', 'gun', 'control', 'people', 'think', 'because', 'evidence',
             'the', 'a', 'of', 'and', 'is', 'not', 'that', 'should', 'we', 'they', 'law', 'state', 'debate', 'claim']
    lines = [' '.join(' '.join(rng.choice(words, int(rng.integers(4, 10)))) + '.' for _ in range(int(rng.integers(1, 4)))) for _ in range(300)]
    d = tmp_path
    (d / 'data').mkdir()
    open(d / 'data' / 'train.txt', 'w').write('\n'.join(lines) + '\n')
    vocab = util_sp.spm(str(d / 'data' / 'vocab'), str(d / 'data' / 'train.txt'), size=48)
    val = [util_sp.encode_capped(vocab, t, cap=24) for t in lines[:40]]
    np.save(d / 'data' / 'valid.npy', vpack(val, (len(val), max(map(len, val))), vocab.eos_id(), np.int32))
    cfgj = {"paths": {"log": str(d / 'log'), "vocab": str(d / 'data' / 'vocab.model'), "train": str(d / 'data' / 'train.txt'),
                      "valid": str(d / 'data' / 'valid.npy'), "ckpt": str(d / 'ckpt')},
            "model": {"accelerate": 1e-4, "learn_rate": 1e-3, "dim_tgt": 48, "dim_emb": 64, "dim_rep": 16, "rnn_layers": 2,
                      "bidirectional": True, "bidir_stacked": True, "attentive": False, "logit_use_embed": True},
            "train": {"seed": 0, "max_len": 24, "batch_train": 16, "batch_valid": 12, "total_valid": 40}}
    json.dump(cfgj, open(d / 'config.json', 'w'))
    root = os.path.dirname(HERE)
    # a fresh process: the profiler child must be started before anything has touched the GPU
    res = subprocess.run([sys.executable, '-m', 'argsim_amd.train', '--config', str(d / 'config.json'), '--trial', 'p', '--profile',
                          '--rounds', '1', '--steps-per-round', '10', '--valid-every', '10', '--save-tf'],
                         cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = res.stdout.decode(errors='replace')
    assert res.returncode == 0, out[-3000:]
    rec = [json.loads(l) for l in out.splitlines() if l.startswith('{"valid32_forward_ms"')]
    assert rec and rec[0]['valid32_forward_ms'] > 0 and rec[0]['classes']['gemm']['launches'] > 0
    if shutil.which('rocprofv3'):
        stats = glob.glob(str(d / 'log' / 'p_profile' / '**' / '*kernel_stats.csv'), recursive=True)
        assert stats, out[-2000:]
        assert 'gemm_f32' in open(stats[0]).read()
    assert (d / 'ckpt' / 'p0.npz').exists() and (d / 'ckpt' / 'p0.index').exists() and glob.glob(str(d / 'ckpt' / 'p0.data-*'))


# ------------------------------------------------------------------------------------------ BASELINE configs[2] at full size
def test_configs2_full_size_bf16_properties():
    """BASELINE configs[2]: bf16 GEMM operands, batch 1024 x seq 128 (vocab 8k, latent 128) -- too large for any oracle,
    so size-independent properties: rows are independent, every loss is finite, and the loss falls over a few Adam steps on
    one batch.  Row independence is exact at one geometry (two 64-row blocks swapped: every row gets bit for bit the z it
    had).  Across geometries (the first 64 rows alone run other kernel forms, which sum the same products in another
    order) this mode agrees to bf16 accuracy only: the recurrent operand h is ROUNDED to bf16 at every step, and a last-bit
    difference in h can flip that rounding (2^-9 relative) -- the fp32 mode holds 1e-5 there
    (test_large_batch_rows_are_independent)."""
    import torch
    from argsim_amd import synth
    from argsim_amd.model import VAE
    m = VAE('train', seed=0, dtype='bf16', dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
    m.step = 20000
    ids = synth.batch(1024, 128, 8192, ragged=True, seed=11)
    z_all = m.encode(ids)
    z_64 = m.encode(ids[:64])
    assert z_all.shape == (1024, 128) and np.isfinite(z_all).all()
    assert np.abs(z_all[:64] - z_64).max() <= 1e-2
    swapped = ids.copy()
    swapped[:64], swapped[64:128] = ids[64:128], ids[:64]
    z_sw = m.encode(swapped)
    assert np.array_equal(z_sw[64:128], z_all[:64]) and np.array_equal(z_sw[:64], z_all[64:128]) and np.array_equal(z_sw[128:], z_all[128:])
    dev = torch.as_tensor(ids).cuda()
    first = last = None
    for i in range(4):
        m.train_step(dev, dev, seed=i)
        lg, lk, lo = m.losses()
        assert np.isfinite([lg, lk, lo]).all()
        first = lo if first is None else first
        last = lo
    assert last < first
    m.close()
