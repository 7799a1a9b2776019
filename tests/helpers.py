"""shared fixtures for the parity tests: seeded ragged batches and oracle parameter sets."""
import numpy as np

from oracle import vae_numpy as vn

CASES = {
    # name: (cfg kwargs, B, S, lens or None(full))
    'tiny':  (dict(dim_tgt=32, dim_emb=16, dim_rep=8, rnn_layers=3), 4, 7, [7, 1, 3, 5]),
    'mid':   (dict(dim_tgt=256, dim_emb=64, dim_rep=32, rnn_layers=3), 8, 16, [16, 2, 9, 16, 1, 5, 12, 7]),
    'wide':  (dict(dim_tgt=512, dim_emb=256, dim_rep=64, rnn_layers=2), 20, 9, None),
    'full2': (dict(dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3), 2, 64, [64, 23]),
    # five word ids for 288 tokens: every id owns several 32-row segments of the grouped embedding-gradient scatter
    'hot':   (dict(dim_tgt=8, dim_emb=16, dim_rep=8, rnn_layers=1), 24, 12, None),
    # more tokens than vocabulary entries: the layers fed by embedding rows project the table and gather / scatter by id
    'tab':   (dict(dim_tgt=64, dim_emb=64, dim_rep=16, rnn_layers=2), 12, 10, [10, 3, 7, 10, 1, 5, 9, 2, 10, 6, 4, 8]),
    # the other widths the register-form GRU kernels are instantiated for (dim_emb 32 and 128)
    'd32':   (dict(dim_tgt=128, dim_emb=32, dim_rep=16, rnn_layers=2), 6, 9, [9, 1, 4, 9, 2, 7]),
    'd128':  (dict(dim_tgt=384, dim_emb=128, dim_rep=48, rnn_layers=3), 10, 11, [11, 2, 6, 11, 1, 9, 4, 8, 3, 10]),
    # a vocabulary beyond the scatter's LDS histogram: the per-element atomic fallback
    'bigv':  (dict(dim_tgt=12800, dim_emb=16, dim_rep=8, rnn_layers=1), 4, 6, [6, 2, 4, 1]),
}
# production geometry (D = 512, V = 8192, B = 64: one full 64-row block per GRU workgroup, so the LDS-weight team
# kernels, the fast-staging and split-K GEMM paths run).  Too large for the live float64 oracle inside a test:
# compared with committed reduced fixtures only (tests/golden/make_oracle_golden.py).
PROD_CASES = {
    'prod64':  (dict(dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3), 64, 64, 'ragged'),
    'prod128': (dict(dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3), 64, 128, 'ragged'),
}
# round 3: what the BIG configurations run, reduced fixtures only.  Fifth field: extras
#   prod512  B = 512: every GRU launch is a row-block-pipelined (PIPE) team instantiation, fp32 and bf16 (configs[2]/[3])
#   cfg0     the reference's own config.json dims (R = 1024, src/config.json:10-21) through backward
#   cfg4     BASELINE configs[4]: R = 512 with the beta / free-bits extension live
#   headline the bench batch itself: B 256 x S 64 FULL Zipf ids of argsim_amd.synth (seed 0), step 20000
BIG_CASES = {
    'prod512':  (dict(dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3), 512, 14, 'ragged', {}),
    'cfg0':     (dict(dim_tgt=8192, dim_emb=512, dim_rep=1024, rnn_layers=3), 32, 64, 'ragged', {}),
    'cfg4':     (dict(dim_tgt=8192, dim_emb=512, dim_rep=512, rnn_layers=3), 64, 32, 'ragged', dict(kl_beta=0.5, free_bits=0.02)),
    'headline': (dict(dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3), 256, 64, 'synth', {}),
    # round 4: a batch WITHOUT a team-kernel geometry of its own -- 100 rows run in the 128-slot geometry with 28 phantom
    # rows, the compact layout whatever the fill, table-fed first layers below the vocabulary size (1 200 tokens >= 1 024)
    'phantom100': (dict(dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3), 100, 12, 'ragged', {}),
    # BASELINE configs[0] as worded: src/config.json dims, batch 32, rows 0..31 of the 1 000 IAC posts encoded with the 8k SentencePiece
    # model trained in the build container and capped to 64 pieces (tests/golden/make_configs0_golden.py)
    'cfg0real': (dict(dim_tgt=8192, dim_emb=512, dim_rep=1024, rnn_layers=3), 32, 64, 'configs0', {}),
}


def big_extras(name):
    return dict(BIG_CASES[name][4])


def make_case(name, seed=0, pad=2, bias_scale=0.1):
    kw, B, S, lens = (CASES.get(name) or PROD_CASES.get(name) or BIG_CASES[name][:4])
    cfg = vn.make_cfg(**kw)
    rng = np.random.default_rng(seed + 17)
    V, R = cfg['dim_tgt'], cfg['dim_rep']
    ids = np.full((B, S + pad), cfg['eos'], np.int32)
    if lens is None:
        lens = [S] * B
    elif lens == 'ragged':             # a few full rows, a length-1 row, the rest uniform in [2, S]
        lens = rng.integers(2, S + 1, B)
        lens[[0, B // 2]] = S
        lens[1] = 1
    if isinstance(lens, str) and lens == 'configs0':   # real text: the committed ids of tests/golden/configs0_ids.npz
        import os
        with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'configs0_ids.npz'), allow_pickle=False) as f:
            real = f['ids'][:B].astype(np.int32)
        ids[:, :S] = real
        lens = [int((r != cfg['eos']).sum()) for r in real]
    elif isinstance(lens, str) and lens == 'synth':      # the bench batch (FULL Zipf rows), eos-padded like every other case
        from argsim_amd import synth
        ids[:, :S] = synth.batch(B, S, V, seed=0)
        lens = [S] * B
    else:
      for b, n in enumerate(lens):
        ids[b, :n] = rng.integers(3, V, n)
    smax = max(lens)
    keep = (rng.random((smax, B)) < 0.6).astype(np.uint8)
    eps = rng.standard_normal((B, R)).astype(np.float32)
    P = vn.init_params(cfg, seed, bias_scale=bias_scale)
    P = {k: v.astype(np.float32).astype(np.float64) for k, v in P.items()}   # exactly representable in fp32
    return cfg, P, ids, keep, eps


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    d = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (d if d > 0 else 1.0)
