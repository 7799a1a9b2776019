"""GPU tests added in round 3: a batch-shape-independent order of data-parallel bucket announcements, oracle fixtures for
what the BIG configurations run (row-block-pipelined kernels at B = 512, R = 1024 / 512 through backward, the bench batch
itself), executed-FLOP accounting of the timing hook."""
import os

import numpy as np
import pytest

from helpers import big_extras, make_case
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


def _gold(name):
    with np.load(os.path.join(HERE, 'golden', 'oracle_%s.npz' % name), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


# ------------------------------------------------------------------------------------------ bucket order
@pytest.mark.parametrize("L", [1, 2, 3])
def test_bucket_announcement_order_does_not_depend_on_the_batch_shape(L):
    """Data-parallel ranks pad their shards to their own longest row, so within ONE step a rank can be table-fed
    (tokens >= vocabulary) while its peer is not; collectives are paired by call order, so the hook must
    announce buckets in one fixed order whatever the shape: 0..2L, then the embedding (2L+2), then encode/rnn1 (2L+1),
    with one fence before each of the 2L persistent-capable BPTT launches."""
    from argsim_amd.model import VAE
    V = 64
    m = VAE('train', dim_tgt=V, dim_emb=64, dim_rep=16, rnn_layers=L, seed=0)
    m.step = 20000
    want = list(range(2 * L + 1)) + [2 * L + 2, 2 * L + 1]
    rng = np.random.default_rng(3)
    seen = {}
    # (B, S): both id sources table-fed | neither | only the decoder's (S*B < V <= (S+1)*B) | one token per row
    for B, S in ((12, 10), (4, 6), (7, 9), (3, 1)):
        ids = rng.integers(3, V, (B, S)).astype(np.int32)
        log = []
        m.set_grad_hook(lambda b, off, cnt: log.append(b))
        m.forward_backward(ids, ids, seed=1)
        m.set_grad_hook(None)
        assert all(np.isfinite(m.losses()))
        assert [b for b in log if b >= 0] == want, (B, S, log)
        assert log.count(-1) == 2 * L and log[0] == -1, (B, S, log)
        seen[(B, S)] = m.present_ids()
    assert seen[(12, 10)][0] > 0 and seen[(12, 10)][1] > 0          # table-fed both sides
    assert seen[(4, 6)] == (-1, -1) and seen[(7, 9)][0] == -1 and seen[(7, 9)][1] > 0
    m.close()


def test_two_ranks_whose_shards_straddle_the_table_threshold(tmp_path):
    """The same, end to end: two real ranks on the test GPU over gloo; rank 0's shard (96 + 104 tokens, V = 64) is
    table-fed, rank 1's (rows of at most 5 ids) is not.  Before the fix the two ranks paired the embedding all-reduce with
    an encoder-layer all-reduce of another size.  Must reproduce the single-process parameters."""
    import subprocess
    import sys
    from argsim_amd.model import VAE
    V, D, R, B, S, steps = 64, 64, 16, 16, 12, 2
    rng = np.random.default_rng(11)
    lens = rng.integers(2, S + 1, B); lens[0] = S
    lens[B // 2:] = rng.integers(1, 6, B - B // 2)
    ids = np.ones((B, S), np.int32)
    for b, n in enumerate(lens):
        ids[b, :n] = rng.integers(3, V, n)
    keep = (rng.random((S, B)) < 0.8).astype(np.uint8)
    eps = rng.standard_normal((B, R)).astype(np.float32)
    case, out = str(tmp_path / 'case.npz'), str(tmp_path / 'dp.npz')
    np.savez(case, V=V, D=D, R=R, ids=ids, keep=keep, eps=eps, steps=steps)
    port = 29500 + (os.getpid() % 400)
    root = os.path.dirname(HERE)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, 'tests', 'dp_worker.py'), case, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    m = VAE('train', dim_tgt=V, dim_emb=D, dim_rep=R, rnn_layers=3, seed=0)
    m.step = 20000
    for i in range(steps):
        m.train_step(ids, ids, keep_mask=keep, eps=eps)
    ref = m.get_params()
    got = {k.replace('|', '/'): v for k, v in np.load(out).items()}
    worst = max(float(np.abs(got[k] - ref[k]).max()) for k in ref)
    assert worst <= 5e-6, worst          # (two Adam steps on gradients summed in another order: 1-2.5e-6 seen; a mispaired collective gives 1e-2)


# ------------------------------------------------------------------------------------------ fixtures of the big configurations
def _check_against_fixture(name, dtype, tz, tl, tg, check_adam=False, **vae_kw):
    gold = _gold(name)
    cfg, P, ids, keep, eps = make_case(name)
    assert np.array_equal(ids, gold['ids']) and np.array_equal(keep, gold['keep'])
    extra = big_extras(name)
    m = _vae(cfg, P, dtype=dtype, **extra, **vae_kw)
    m.step = 20000
    z, lv = m.encode(ids, return_lv=True)
    assert np.abs(z - gold['mu']).max() <= tz and np.abs(lv - gold['lv']).max() <= tz, (name, dtype)
    p0 = m.params.clone()
    m.forward_backward(ids, ids, keep_mask=keep, eps=eps)
    lg, lk, lo = m.losses()
    for got, key in ((lo, 'loss'), (lg, 'loss_gen'), (lk, 'loss_kld')):
        assert abs(got - float(gold[key])) <= tl * abs(float(gold[key])) + 1e-7, (name, dtype, key, got, float(gold[key]))
    grads = m.get_grads()
    for k, g in grads.items():
        g = g.astype(np.float64)
        n = float(gold['gnorm/' + k])
        assert abs(np.linalg.norm(g) - n) <= tg * n, (name, dtype, k)
        p = _probe(k, g.shape)
        assert abs(float((g * p).sum()) - float(gold['gdot/' + k])) <= tg * n * np.linalg.norm(p), (name, dtype, k)
    if check_adam:
        # the update kernel against the oracle's TF-style Adam applied to the SAME (device) gradient: m = v = 0, the
        # beta powers at step + 1 (avae_adam_step), epsilon outside the bias-corrected square root (model.py:189)
        import torch
        g = m.grads.clone()
        lr = vn.schedule(20000, cfg['accelerate'], cfg['learn_rate'])[2]
        m.adam_step()
        newp, _, _ = vn.adam_tf({'p': p0.double().cpu().numpy()}, {'p': g.double().cpu().numpy()},
                                {'p': 0.0}, {'p': 0.0}, 20000, lr)
        d = np.abs(m.params.double().cpu().numpy() - newp['p']).max()
        assert d <= 2e-6, d          # |update| ~ 1.3e-3 per element; fp32 rounding of p - lr_t m / (sqrt(v) + eps)
        assert m.step == 20001
        del torch
    m.close()
    return gold


@pytest.mark.parametrize("dtype,tz,tl,tg", [('f32', 2e-5, 2e-5, 2e-4), ('bf16', 3e-2, 1e-2, 5e-2)])
def test_row_block_pipelined_kernels_against_the_oracle_fixture(dtype, tz, tl, tg):
    """B = 512 (S = 14, ragged, D 512, V 8192): the two-direction encoder launches and the decoder launches of the default path
    are PIPE team instantiations (two row blocks per workgroup), the one-job top encoder layer the 4-team form -- what
    configs[2] (bf16: 16-bit exchange, bf16 gate gradients / saved gates / h, transposing-load GEMMs) and configs[3] (fp32)
    run -- compared with the float64 oracle fixture ONLY: loss, mu, log sigma^2, per-variable gradient norm and one fixed
    projection of every gradient."""
    _check_against_fixture('prod512', dtype, tz, tl, tg)


def test_reference_config_dims_through_backward_and_adam():
    """src/config.json:10-21 dims (R = 1024) at B = 32, S = 64 ragged: the latent block's thin / paired / split-K GEMM
    shaping keyed on R, through backward and one Adam update."""
    _check_against_fixture('cfg0', 'f32', 2e-5, 2e-5, 2e-4, check_adam=True)


def test_configs4_latent512_beta_free_bits_against_the_oracle_fixture():
    """BASELINE configs[4]: R = 512 at D 512 / V 8192 with kl_beta = 0.5 and free_bits = 0.02 live (extensions: identity at
    (1, 0)), through backward and one Adam update, against the extended float64 oracle's fixture."""
    _check_against_fixture('cfg4', 'f32', 2e-5, 2e-5, 2e-4, check_adam=True)


def test_headline_batch_against_the_oracle_fixture():
    """The bench batch itself -- B 256 x S 64 FULL Zipf ids (argsim_amd.synth seed 0), step 20000, table-fed first layers,
    4-team encoder and 2-team decoder kernels, every GEMM form of the headline step -- against the float64 oracle's
    fixture: loss, mu, per-token CE, gradient norms and projections."""
    gold = _check_against_fixture('headline', 'f32', 2e-5, 2e-5, 2e-4)
    cfg, P, ids, keep, eps = make_case('headline')
    m = _vae(cfg, P)
    m.step = 20000
    m.forward_backward(ids, ids, keep_mask=keep, eps=eps)
    assert m.present_ids()[0] > 3000                    # the table-fed path really ran
    m.close()


# ------------------------------------------------------------------------------------------ executed-FLOP accounting
def test_timing_hook_counts_executed_flops_of_device_count_gemms():
    """avae_timing_collect priced the table-fed layers' GEMMs at their static 8192-row bound although the
    kernels exit at the number of ids present.  The class total must now fall with the present-id count: the same batch
    with table_l1 = 0 (per-token GEMMs, no device counts on those layers) executes MORE FLOPs, by what the table saves:
    3 x 2 (tokens - U) D 3D per direction of the two first layers."""
    from argsim_amd import synth
    from argsim_amd.model import VAE
    B, S, D, V = 256, 64, 512, 8192
    ids = synth.batch(B, S, V, seed=0)
    m = VAE('train', seed=0, dim_tgt=V, dim_emb=D, dim_rep=128, rnn_layers=3)
    m.step = 20000
    fl = {}
    for tab in (1, 0):
        m.set_option('table_l1', tab)
        m.train_step(ids, ids, seed=1)
        m.set_option('timing', 1)
        m.train_step(ids, ids, seed=2)
        fl[tab] = m.timing_collect()['gemm'][2]
        m.set_option('timing', 0)
        if tab:
            us, ut = m.present_ids()
    assert 0 < us < V and 0 < ut < V
    T = S + 1
    saved = 3 * 2.0 * D * 3 * D * (2 * (B * S - us) + (B * T - ut))       # fwd + two backward GEMMs; encoder has two directions
    assert abs((fl[0] - fl[1]) - saved) <= 0.02 * saved, (fl, saved, us, ut)
    m.close()


# ------------------------------------------------------------------------------------------ dead steps of the top encoder layer
@pytest.mark.parametrize("case", ['mid', 'tab', 'prod64'])
def test_top_layer_backward_direction_one_step_equals_the_full_form(case):
    """model.py:119-121,135: the top encoder layer's output is consumed at position len_b - 1 only, where the reversed direction
    has seen ONE token; its other steps never reach z or the loss and carry zero gradient (the oracle's gradient of
    encode/rnnL/bwd/R is exactly 0).  The default path runs that one step (option enc_top1); the full form (enc_top1 = 0)
    executes all S steps like the reference's graph: same z, same losses, same gradients -- and dR of that direction exactly
    zero in both."""
    cfg, P, ids, keep, eps = make_case(case)
    m = _vae(cfg, P)
    m.step = 20000
    out = {}
    for top1 in (1, 0):
        m.set_option('enc_top1', top1)
        z = m.encode(ids)
        ev = m.eval(ids, ids)
        m.forward_backward(ids, ids, keep_mask=keep, eps=eps)
        out[top1] = (z, ev, m.get_grads(), m.losses())
    L = cfg['rnn_layers']
    # the forward direction of that layer runs as a one-job launch in the one-step form (another K-split tree): rounding only
    assert np.abs(out[1][0] - out[0][0]).max() <= 2e-6
    for a, b in zip(out[1][1][1:], out[0][1][1:]):
        assert np.abs(a - b).max() <= 2e-5
    assert abs(out[1][3][2] - out[0][3][2]) <= 2e-6 * abs(out[0][3][2])
    for k in out[0][2]:
        a, b = out[1][2][k].astype(np.float64), out[0][2][k].astype(np.float64)
        if k == 'encode/rnn%d/bwd/R' % L:
            assert not a.any() and not b.any(), k
            continue
        assert np.linalg.norm(a - b) <= 2e-5 * np.linalg.norm(b) + 1e-12, k
    m.close()


# ------------------------------------------------------------------------------------------ bf16 mode: transposing-LDS-load GEMM
@pytest.mark.parametrize("M,N,K", [(1536, 512, 4096), (3072, 1024, 1100), (8192, 512, 777), (264, 1032, 130)])
def test_bf16_tn_gemm_reads_row_major_operands(M, N, K):
    """gemm_bf16_tn (compute_dtype 1): C += A^T B with both operands row-major [k][x] bf16, fragments read with
    ds_read_b64_tr_b16 -- against torch on the SAME bf16-rounded operands in float64.  Shapes: a weight gradient with a K
    split, a K that is no multiple of the 64-deep tile, the vocabulary-sized output, edge tiles in M and N."""
    import ctypes as C
    import torch
    from argsim_amd.model import VAE
    m = VAE('train', dtype='bf16', dim_tgt=64, dim_emb=16, dim_rep=8, rnn_layers=1)
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    A = torch.randn((K, M), device='cuda', generator=g) * 0.5
    B = torch.randn((K, N), device='cuda', generator=g) * 0.5
    Cm = torch.full((M, N), 0.25, device='cuda')
    m._stream()
    rc = m._l.avae_debug_gemm_tn16(m._h, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()), M, N, K, M, N, N, 0.5)
    assert rc == 0, m._l.avae_last_error(m._h)
    torch.cuda.synchronize()
    ref = 0.25 + 0.5 * (A.bfloat16().double().t() @ B.bfloat16().double())
    err = float((Cm.double() - ref).abs().max())
    assert err <= 2e-4 * max(1.0, float(ref.abs().max())), err          # fp32 accumulation over K products of O(1/4)
    m.close()


# ------------------------------------------------------------------------------------------ padding skipped in the recurrence
@pytest.mark.parametrize("B,S,dtype", [(256, 64, 'f32'), (64, 40, 'f32'), (512, 24, 'f32'), (1024, 20, 'f32'), (512, 24, 'bf16')])
def test_padding_skipped_in_the_team_kernels_changes_nothing(B, S, dtype):
    """"outputs at t >= len_b ... provably never reach z or the loss -- the build may skip or zero them".
    Default path: rows sorted by length and dealt over the workgroups (row_order), every team runs its row block for the
    steps of its longest row only, the positions behind are zero-filled (option skip_pad) -- against the same library
    running every step of every row (skip_pad = 0), on RAGGED LogNormal batches: one row block per workgroup (B = 256, 64),
    two and four pipelined row blocks (B = 512, 1024), fp32 and the 16-bit exchange.  A row's arithmetic does not depend
    on which team it sits in: z and the per-token losses bit for bit (fp32), gradients up to float-atomic order."""
    from argsim_amd import synth
    from argsim_amd.model import VAE
    m = VAE('train', seed=1, dtype=dtype, dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
    m.set_option('compact', 0)                       # (the padded layout in both runs: the compact one has a test of its own)
    m.step = 20000
    ids = synth.batch(B, S, 8192, ragged=True, seed=7)
    ids[3, 1:] = 1                                   # a one-token row
    out = {}
    for skip in (1, 0):
        m.set_option('skip_pad', skip)
        z = m.encode(ids)
        ev = m.eval(ids, ids)
        m.forward_backward(ids, ids, seed=5)
        out[skip] = (z, ev, m.grads.clone(), m.losses())
    if dtype == 'f32':
        assert np.array_equal(out[1][0], out[0][0])
        for x, y in zip(out[1][1], out[0][1]):
            assert np.array_equal(x, y)
    else:                                            # bf16 gate gradients: the same values, summed by the GEMMs in the same order
        assert np.array_equal(out[1][0], out[0][0])
    d = float((out[1][2] - out[0][2]).norm() / out[0][2].norm())
    assert d < (1e-5 if dtype == 'f32' else 1e-4), d      # (bf16: float atomics over bf16-rounded products -- 1.1e-5 ... 3.5e-5 seen from run to run over rounds 3-4; the mode's gradient tolerance against the oracle is 5e-2)
    assert all(np.isfinite(out[1][3]))
    m.close()


@pytest.mark.parametrize("M,N,K", [(8192, 4096, 512), (4100, 3080, 1024), (4096, 4096, 96)])
def test_bf16_nt_gemm_large_tile_form(M, N, K):
    """compute_dtype 1, C = A B^T + bias on the 256x256 tile kernel (fp32 operands converted to bf16 panels first), against
    torch on the same bf16-rounded operands in float64; edge tiles in M and N, a K that is no multiple of the 64-deep tile."""
    import ctypes as C
    import torch
    from argsim_amd.model import VAE
    m = VAE('train', dtype='bf16', dim_tgt=64, dim_emb=16, dim_rep=8, rnn_layers=1)
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    A = torch.randn((M, K), device='cuda', generator=g) * 0.5
    B = torch.randn((N, K), device='cuda', generator=g) * 0.5
    bias = torch.randn((N,), device='cuda', generator=g)
    Cm = torch.zeros((M, N), device='cuda')
    m._stream()
    rc = m._l.avae_debug_gemm(m._h, 0, 0, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()), C.c_void_p(bias.data_ptr()),
                              M, N, K, K, K, N, 0.5, 0, 1)
    assert rc == 0, m._l.avae_last_error(m._h)
    torch.cuda.synchronize()
    ref = 0.5 * (A.bfloat16().double() @ B.bfloat16().double().t()) + bias.double()
    err = float((Cm.double() - ref).abs().max())
    assert err <= 2e-4 * max(1.0, float(ref.abs().max())), err
    m.close()


# ------------------------------------------------------------------------------------------ default path vs the graph as written
@pytest.mark.parametrize("D,V,R,L,B,S,dtype", [
    (512, 8192, 128, 3, 64, 9, 'f32'),      # one 64-row block, short rows
    (512, 8192, 128, 2, 128, 33, 'f32'),    # two layers: the one-step top layer sits right on the table-fed layer
    (512, 1024, 64, 3, 192, 17, 'f32'),     # B = 192: 2-team blocks of 32 rows; tokens > vocabulary on both sides
    (512, 8192, 128, 1, 64, 12, 'f32'),     # one layer: no one-step form
    (512, 8192, 128, 3, 48, 20, 'f32'),     # B = 48: no team geometry (register-form kernels), nothing to skip
    (64, 256, 32, 3, 40, 14, 'f32'),        # small width: register-form kernels throughout
    (512, 8192, 128, 3, 256, 21, 'bf16'),   # bf16 storage paths at one row block per workgroup
    (512, 8192, 128, 2, 512, 11, 'bf16'),   # bf16, pipelined row blocks, two layers
    # batches WITHOUT a team geometry of their own (the reference's batch_train 100 and batch_valid 200, src/config.json): the
    # default path runs the team kernels on the next row count that has one, the slots beyond B holding phantom rows (GruArgs::Bx)
    (512, 1024, 64, 3, 100, 21, 'f32'),     # 100 -> 128 slots
    (512, 1024, 64, 3, 200, 12, 'f32'),     # 200 -> 256 slots
    (512, 1024, 64, 2, 17, 70, 'f32'),      # 17 -> 32 slots: one team of real rows and one mixed
    (512, 1024, 64, 3, 72, 30, 'f32'),      # 72 -> 96 slots
    (512, 8192, 128, 3, 100, 90, 'bf16'),   # bf16 exchange, h0 seeded through the permutation
    (512, 1024, 64, 3, 1000, 9, 'f32'),     # 1000 -> 1024 slots: pipelined row blocks
    (512, 8192, 128, 3, 100, 30, 'f32'),    # fewer tokens than vocabulary entries: such a batch is table-fed all the same (use_table)
    (512, 8192, 128, 3, 200, 12, 'bf16'),
    (512, 1024, 64, 3, 300, 10, 'f32'),     # 300 -> 384 slots (three 64-row blocks per job pair / 32-row blocks)
    (512, 1024, 64, 2, 576, 6, 'bf16'),     # 576 -> 768 slots: three pipelined row blocks, a quarter of the slots phantom
])
def test_default_path_equals_the_graph_as_written(D, V, R, L, B, S, dtype):
    """Every exact elimination and storage choice of the default path switched off at once (table_l1, enc_top1, skip_pad, compact; bf16
    mode: bf16_tn, bf16_sv, bf16_act as well) must give the same z, losses and gradients on ragged batches, over the launch
    geometries the library distinguishes.  fp32: z to rounding (another K-split tree where a launch changes form), gradients
    to float-atomic order; bf16: to the mode's accuracy."""
    from argsim_amd import synth
    from argsim_amd.model import VAE
    m = VAE('train', seed=3, dtype=dtype, dim_tgt=V, dim_emb=D, dim_rep=R, rnn_layers=L)
    m.step = 20000
    ids = synth.batch(B, S, V, ragged=True, seed=B + S)
    ids[1, 1:] = 1
    ids[0, :] = synth.batch(1, S, V, seed=1)[0]          # one full-length row
    rng = np.random.default_rng(5)
    keep = (rng.random((S, B)) < 0.7).astype(np.uint8)
    eps = rng.standard_normal((B, R)).astype(np.float32)
    opts = ['table_l1', 'enc_top1', 'skip_pad', 'compact'] + (['bf16_tn', 'bf16_sv', 'bf16_act'] if dtype == 'bf16' else [])
    out = {}
    for on in (1, 0):
        for k in opts:
            m.set_option(k, on)
        z = m.encode(ids)
        m.forward_backward(ids, ids, keep_mask=keep, eps=eps)
        out[on] = (z, m.grads.clone(), m.losses())
    tz, tl, tg = (2e-6, 2e-6, 2e-5) if dtype == 'f32' else (2e-2, 5e-3, 3e-2)
    assert np.isfinite(out[1][0]).all() and np.abs(out[1][0] - out[0][0]).max() <= tz
    assert abs(out[1][2][2] - out[0][2][2]) <= tl * abs(out[0][2][2])
    d = float((out[1][1] - out[0][1]).norm() / out[0][1].norm())
    assert d <= tg, d
    m.close()


# ------------------------------------------------------------------------------------------ compact layout
@pytest.mark.parametrize("B,S,dtype", [(256, 64, 'f32'), (64, 40, 'f32'), (512, 24, 'f32'), (1024, 20, 'f32'), (256, 40, 'bf16'), (512, 24, 'bf16')])
def test_compact_layout_changes_nothing(B, S, dtype):
    """Option compact: every array between the GEMMs and the GRU launches of the encoder and of the decoder holds the REAL rows
    only (row_map: the place of (t, b) among the positions a row's steps cover, time-major; GruArgs::rowmap), the GEMMs over them
    take the device-side row count, the pick / the token compaction read through the map.  Against the padded layout with every
    step of every row run (compact = 0, skip_pad = 0), source != target, a one-token row, and target rows with an eos id in the
    MIDDLE (the decoder mask is per position, model.py:161: such a row's steps end behind its LAST real id), an all-eos source row: z and the per-token
    losses bit for bit in fp32 (a GEMM row's products do not depend on which row of the operand it is), gradients to float-atomic
    order."""
    from argsim_amd import synth
    from argsim_amd.model import VAE
    m = VAE('train', seed=1, dtype=dtype, dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
    m.step = 20000
    src = synth.batch(B, S, 8192, ragged=True, seed=9)
    src[5, 1:] = 1
    tgt = synth.batch(B, S, 8192, ragged=True, seed=10)
    tgt[6, 1:] = 1
    for b in (0, 9, 17, 40):                         # eos in the middle of a target row, real ids behind it
        n = int((tgt[b] != 1).sum())
        if n >= 3:
            tgt[b, n // 2] = 1
    src[11, 2] = 1                                   # and of a source row (its length is the COUNT of non-eos ids, model.py:84)
    src[23, :] = 1                                   # a source row without a single id: undefined in the reference (gather_nd at -1); zeros picked, no gradient, in both layouts
    out = {}
    for c in (1, 0):
        m.set_option('compact', c)
        m.set_option('skip_pad', c)
        z = m.encode(src)
        ev = m.eval(src, tgt)
        m.forward_backward(src, tgt, seed=5)
        out[c] = (z, ev, m.grads.clone(), m.losses())
    if dtype == 'f32':
        assert np.array_equal(out[1][0], out[0][0])
        for x, y in zip(out[1][1], out[0][1]):
            assert np.array_equal(x, y)
        tol = 1e-5
    else:
        assert np.abs(out[1][0] - out[0][0]).max() <= 1e-2
        tol = 3e-2
    d = float((out[1][2] - out[0][2]).norm() / out[0][2].norm())
    assert d < tol, d
    assert all(np.isfinite(out[1][3]))
    m.close()


def test_compact_layout_on_a_full_batch():
    """The compact layout forced on a FULL batch (what a batch size without a team geometry of its own always runs): every
    position is real, the device-side counts equal the static bounds, and from the second call on the host's expected count
    shapes the GEMM launches (the double-buffered form where it fills the chip) -- same z and per-token losses bit for bit,
    gradients to float-atomic order."""
    from argsim_amd import synth
    from argsim_amd.model import VAE
    m = VAE('train', seed=1, dtype='f32', dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
    m.step = 20000
    ids = synth.batch(256, 64, 8192, seed=3)
    out = {}
    for c in (1, 0):
        m.set_option('compact', c)
        for rep in range(2):                  # (the second call knows the first one's fill)
            z = m.encode(ids)
            ev = m.eval(ids, ids)
            m.forward_backward(ids, ids, seed=5)
        out[c] = (z, ev, m.grads.clone())
    assert np.array_equal(out[1][0], out[0][0])
    for x, y in zip(out[1][1], out[0][1]):
        assert np.array_equal(x, y)
    d = float((out[1][2] - out[0][2]).norm() / out[0][2].norm())
    assert d < 1e-5, d
    m.close()
