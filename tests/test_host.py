"""CPU tests of the host side: batching helpers against goldens captured from the reference, the
tokeniser wrappers, the batch generator / prefetch pipe, checkpoint name mapping."""
import json
import os
from itertools import islice

import numpy as np
import pytest

from argsim_amd import util_np, util_sp

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_util_np_against_reference_goldens():
    g = json.load(open(os.path.join(GOLD, 'util_np_golden.json')))
    for c in g['vpack']:
        assert util_np.vpack(c['arrays'], tuple(c['shape']), c['fill'], np.int32).tolist() == c['result']
    for c in g['partition']:
        assert [list(p) for p in util_np.partition(c['n'], c['m'], c['discard'])] == c['result']
    for c in g['sample']:
        assert list(islice(util_np.sample(c['n'], c['seed']), len(c['result']))) == c['result']


def test_survey_known_answers():
    # SURVEY.md section 8c
    assert list(islice(util_np.sample(5, 0), 12)) == [2, 0, 1, 3, 4, 1, 2, 0, 3, 4, 0, 1]
    assert util_np.vpack([[1, 2, 3], [4]], (2, 4), 1, np.int32).tolist() == [[1, 2, 3, 1], [4, 1, 1, 1]]
    assert list(util_np.partition(10, 4)) == [(0, 4), (4, 8), (8, 10)]


def test_vpack_edge_cases():
    assert util_np.vpack([], (0, 3), 1, np.int32).shape == (0, 3)
    assert util_np.vpack([[], []], (2, 2), 7, np.int32).tolist() == [[7, 7], [7, 7]]
    with pytest.raises(ValueError):
        util_np.vpack([[1, 2, 3]], (1, 2), 0, np.int32)       # row longer than the shape: reference raises too


def test_sample_does_not_touch_global_rng():
    np.random.seed(123)
    a = np.random.rand()
    np.random.seed(123)
    list(islice(util_np.sample(9, 4), 30))
    assert np.random.rand() == a


def test_sentence_splitter():
    s = util_sp.sent_split('Dr. Smith went home. He slept! "Why?" she asked. The U.S. is big (really.) Yes')
    assert s[0] == 'Dr. Smith went home.' and s[1] == 'He slept!' and s[-1] == 'Yes'
    assert util_sp.sent_split('') == [] and util_sp.sent_split('no boundary') == ['no boundary']


@pytest.fixture(scope='module')
def vocab(tmp_path_factory):
    d = tmp_path_factory.mktemp('spm')
    rng = np.random.default_rng(0)
    words = ['argument', 'stance', 'abortion', 'rights', 'gun', 'control', 'people', 'think', 'because', 'evidence',
             'the', 'a', 'of', 'and', 'is', 'not', 'that', 'should', 'we', 'they', 'law', 'state', 'debate', 'claim']
    lines = []
    for _ in range(400):
        n = int(rng.integers(3, 9))
        sents = [' '.join(rng.choice(words, int(rng.integers(4, 12)))) + '.' for _ in range(n)]
        lines.append(' '.join(sents))
    path = os.path.join(d, 'train.txt')
    with open(path, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    v = util_sp.spm(os.path.join(d, 'vocab'), path, size=48)
    return v, path, lines


def test_spm_ids_and_roundtrip(vocab):
    v, path, lines = vocab
    assert (v.unk_id(), v.eos_id(), v.bos_id()) == (0, 1, 2)          # util_sp.py:17
    ids = util_sp.encode(v, lines[:5])
    assert ids.dtype == np.int32 and ids.shape[0] == 5
    assert (ids[:, -1] == v.eos_id()).any() or ids.shape[1] == max(len(v.encode_as_ids(s)) for s in lines[:5])
    back = list(util_sp.decode(v, ids))
    assert back == lines[:5]


def test_encode_capped_ladder(vocab):
    v, path, lines = vocab
    text = lines[0]
    full = v.encode_as_ids(text)
    assert util_sp.encode_capped(v, text, cap=len(full)) == full
    cap = len(full) // 2
    ids = util_sp.encode_capped(v, text, cap=cap)
    assert 0 < len(ids) <= cap
    # what fits is a whole number of leading sentences, as in the reference's fall-back
    sents = util_sp.sent_split(text)
    assert any(ids == v.encode_as_ids(' '.join(sents[:n])) for n in range(1, len(sents) + 1))
    # even the first sentence does not fit: hard truncation
    assert len(util_sp.encode_capped(v, text, cap=2)) == 2
    s = util_sp.encode_capped_sample(v, text, cap=cap)
    assert len(s) <= cap
    a, b = util_sp.encode_capped_sample_pair(v, text, cap=cap)
    assert len(a) <= cap and len(b) <= cap


def test_batch_generator_and_pipe(vocab):
    from argsim_amd.train import batch, pipe
    v, path, lines = vocab
    gen = batch(8, path, v, seed=0, kudo=False, max_len=16)
    src, tgt = next(gen)
    assert src is tgt and src.shape[0] == 8 and src.shape[1] <= 16 and src.dtype == np.int32
    # order follows util_np.sample; rows are eos padded
    order = list(islice(util_np.sample(len(lines), 0), 8))
    for row, i in zip(src, order):
        want = util_sp.encode_capped(v, lines[i], cap=16)
        assert row[:len(want)].tolist() == want and (row[len(want):] == v.eos_id()).all()
    gen2 = batch(4, path, v, seed=0, kudo=True, max_len=16)
    s2, t2 = next(gen2)
    assert s2.shape[0] == t2.shape[0] == 4
    # data parallel: the ranks' shards are the row blocks of the single-process batch, batch after batch
    whole = batch(8, path, v, seed=0, kudo=False, max_len=16)
    shards = [batch(8, path, v, seed=0, kudo=False, max_len=16, rank=r, world=2) for r in range(2)]
    for _ in range(3):
        full = next(whole)[0]
        for r, g in enumerate(shards):
            part = next(g)[0]
            rows = full[4 * r:4 * r + 4]
            assert part.shape[0] == 4 and part.shape[1] <= rows.shape[1]
            assert (rows[:, :part.shape[1]] == part).all() and (rows[:, part.shape[1]:] == v.eos_id()).all()
    got = list(islice(pipe(iter(range(10)), prefetch=3), 10))
    assert got == list(range(10))


def test_rank_seeds_are_distinct_and_stable():
    from argsim_amd.dist import rank_seed
    keys = [rank_seed(12345, r) for r in range(8)]
    assert len(set(keys)) == 8 and all(0 <= k < 2 ** 64 for k in keys)
    assert keys == [rank_seed(12345, r) for r in range(8)] and rank_seed(12346, 0) != keys[0]


def test_checkpoint_tf_name_roundtrip():
    from argsim_amd import ckpt
    from oracle import vae_numpy as vn
    cfg = vn.make_cfg(dim_tgt=32, dim_emb=16, dim_rep=8, rnn_layers=2)
    P = vn.init_params(cfg, 0, bias_scale=0.1)
    tf = ckpt.to_tf_names(P)
    assert 'embed/embedding' in tf and 'latent/mu/kernel' in tf
    k = 'decode/rnn/cudnn_gru/rnn/multi_rnn_cell/cell_1/cudnn_compatible_gru_cell/gates/kernel'
    assert tf[k].shape == (16 + 16, 32)
    e = 'encode/rnn2/bwd/cudnn_gru/rnn/multi_rnn_cell/cell_0/cudnn_compatible_gru_cell/candidate/input_projection/kernel'
    assert tf[e].shape == (32, 16)
    back = ckpt.from_tf_names(tf, list(P))
    for name in P:
        if name.endswith('/bW') or name.endswith('/bR'):
            continue
        assert np.array_equal(back[name], P[name]), name
    # r/u biases come back summed into bW (same function), candidate biases exactly
    D = 16
    for base in ('decode/rnn/l1/', 'encode/rnn1/fwd/'):
        assert np.allclose(back[base + 'bW'][:2 * D] + back[base + 'bR'][:2 * D], P[base + 'bW'][:2 * D] + P[base + 'bR'][:2 * D])
        assert np.array_equal(back[base + 'bW'][2 * D:], P[base + 'bW'][2 * D:])
        assert np.array_equal(back[base + 'bR'][2 * D:], P[base + 'bR'][2 * D:])
    # equivalence of the function: the oracle gives the same z with the converted parameters
    ids = np.array([[3, 4, 5, 1], [6, 7, 1, 1]], np.int32)
    z0 = vn.forward(P, cfg, ids, ids, 'valid')['z']
    z1 = vn.forward({k: back[k] for k in P}, cfg, ids, ids, 'valid')['z']
    assert np.abs(z0 - z1).max() < 1e-12


def test_train_cli_flags_match_reference():
    from argsim_amd.train import parse_args
    A = parse_args(['--rounds', '2', '--sample', '--trial', 'x'])
    assert (A.trial, A.config, A.ckpt, A.gpu, A.seed, A.rounds, A.prefetch, A.sample, A.profile) == \
        ('x', 'config.json', None, '0', 0, 2, 16, True, False)


def test_synthetic_batches():
    from argsim_amd import synth
    a = synth.batch(16, 12, 8192, seed=0)
    assert a.shape == (16, 12) and a.dtype == np.int32 and a.min() >= 3 and a.max() < 8192
    r = synth.batch(64, 64, 8192, ragged=True, seed=0)
    lens = (r != 1).sum(1)
    assert lens.min() >= 2 and lens.max() <= 64 and (lens < 64).any()
    assert np.array_equal(a, synth.batch(16, 12, 8192, seed=0))


def test_data_prep_ibm_and_iac(tmp_path):
    """counterparts of data_ibm.py / data_iac.py on synthetic corpora of the same file shapes"""
    import csv, json
    from argsim_amd import data_prep
    rng = np.random.default_rng(1)
    words = ['claim', 'evidence', 'topic', 'we', 'should', 'ban', 'allow', 'the', 'of', 'because', 'people', 'rights', 'law',
             'government', 'school', 'marriage', 'justify', 'question', 'believe', 'liberty', 'zoning', 'tax', 'vote', 'health', 'market']
    def sent(): return ' '.join(rng.choice(words, int(rng.integers(4, 10)))).capitalize() + '.'
    src = tmp_path / 'ibm'; src.mkdir()
    n = 0
    for split in data_prep.IBM_SPLITS:
        with open(src / split, 'w', newline='') as f:
            w = csv.writer(f); w.writerow(['id', 'topic', 'x', 'claim'])
            for _ in range(30):
                w.writerow([n, 't', 'x', sent()]); n += 1
    out = tmp_path / 'out_ibm'
    def fit(fn):
        # SentencePiece accepts only a narrow vocabulary range on a toy corpus: take the first size that trains
        for size in (64, 60, 56, 52, 50, 48, 46, 44, 42, 40):
            try:
                return fn(size)
            except RuntimeError:
                continue
        raise AssertionError("no trainable vocabulary size")
    v = fit(lambda size: data_prep.prep_ibm(str(src), str(out), valid_size=20, vocab_size=size))
    valid = np.load(out / 'valid.npy')
    train = open(out / 'train.txt').read().splitlines()
    assert valid.shape[0] == 20 and valid.dtype == np.int32 and len(train) == n - 20
    sents = open(out / 'all.txt').read().splitlines()
    np.random.seed(0); ref = list(sents); np.random.shuffle(ref)          # the reference's exact shuffle
    assert train == ref[20:]
    raw = tmp_path / 'iac'; raw.mkdir()
    for i in range(3):
        posts = [[j, 'side', 'author', '  %s\n %s ' % (sent(), sent()), [], None, 'cat', 0] for j in range(10)] + [[99, 's', 'a', '   ', [], None, 'c', 0]]
        json.dump([posts, {}, {}], open(raw / ('%d.json' % i), 'w'))
    open(tmp_path / 'val.txt', 'w').write('\n'.join(sent() + ' ' + sent() for _ in range(7)) + '\n')
    out2 = tmp_path / 'out_iac'
    v2 = fit(lambda size: data_prep.prep_iac(str(raw), str(tmp_path / 'val.txt'), str(out2), cap=16, vocab_size=size))
    valid2 = np.load(out2 / 'valid.npy')
    assert valid2.shape == (7, 16) and (valid2[:, -1] == v2.eos_id()).all()
    lines = open(out2 / 'train.txt').read().splitlines()
    assert len(lines) == 30                                                 # the empty post was dropped
    assert all(len(v2.encode_as_ids(l)) <= 16 for l in lines) and all(l == l.lower() for l in lines)


def test_checkpoint_audit_is_loud():
    """restore refuses, with names, what does not fit the model: wrong shapes always, missing / unknown keys when strict"""
    from argsim_amd import ckpt

    class Stub:
        names = ['a/kernel', 'a/bias']
        shapes = {'a/kernel': (2, 3), 'a/bias': (3,)}
        step = 0

        def __init__(self):
            self.got = {}

        def set_tensor(self, k, v, kind=0):
            self.got[(k, kind)] = np.asarray(v)

    good = {'a/kernel': np.zeros((2, 3), np.float32), 'a/bias': np.ones(3, np.float32), 'global_step': np.asarray(7)}
    m = Stub()
    ckpt.load_state_dict(m, good)
    assert m.step == 7 and ('a/bias', 0) in m.got
    with pytest.raises(ValueError, match='a/kernel'):
        ckpt.load_state_dict(Stub(), dict(good, **{'a/kernel': np.zeros((3, 2), np.float32)}))
    with pytest.raises(KeyError, match='unexpected'):
        ckpt.load_state_dict(Stub(), dict(good, stray=np.zeros(1)))
    with pytest.raises(KeyError, match='missing'):
        ckpt.load_state_dict(Stub(), {'a/kernel': good['a/kernel']})
    m = Stub()
    ckpt.load_state_dict(m, {'a/kernel': good['a/kernel']}, strict=False)      # partial 'infer' restore
    assert list(m.got) == [('a/kernel', 0)]


def test_bench_quotes_committed_counters_only_for_the_sources_they_were_taken_on(tmp_path, monkeypatch):
    """`roofline.traffic` / `mfma_busy` on the bench line come from a COMMITTED rocprofv3 summary, which goes stale
    the moment a kernel changes.  The summary records the digest of the kernel sources it was taken on (argsim_amd.lib.source_digest,
    written by scripts/summarize_profile.py); bench.pmc_summary hands it out only for the same digest and otherwise says why."""
    import importlib
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    from argsim_amd.lib import source_digest
    d = source_digest()
    assert len(d) == 16 and d == source_digest()
    (tmp_path / 'profiles').mkdir()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    pm, why = bench.pmc_summary('rXX')
    assert pm is None and 'no committed counter passes' in why
    json.dump({'source_digest': d, 'gemm_class': {'hbm_bytes_per_dispatch': 1.0}}, open(tmp_path / 'profiles' / 'rXX_pmc_summary.json', 'w'))
    pm, why = bench.pmc_summary('rXX')
    assert pm is not None and why is None and pm['gemm_class']['hbm_bytes_per_dispatch'] == 1.0
    json.dump({'source_digest': '0' * 16, 'gemm_class': {}}, open(tmp_path / 'profiles' / 'rXX_pmc_summary.json', 'w'))
    pm, why = bench.pmc_summary('rXX')
    assert pm is None and 'other kernel sources' in why


REF_POSTS = '/root/reference/docs/results_iac/clustering.csv'


@pytest.mark.skipif(not os.path.exists(REF_POSTS), reason="the reference tree (its docs hold the only real corpus) is not on this machine")
def test_data_prep_iac_on_the_reference_posts(tmp_path):
    """SURVEY 8f item 3 on REAL text (build container only: the GPU box has no reference tree): the 1 901 cleaned IAC posts of the
    reference's docs/results_iac/clustering.csv, laid out as fourforums discussion files (post field 3 = text, data_iac.py:18-27), go
    through prep_iac: an 8192-piece SentencePiece model with the reference's ids (util_sp.py:17-39), train.txt = the DECODE of every
    post capped to 64 pieces at a sentence boundary (data_iac.py:37-41, util_sp.py:42-63), valid.npy = capped validation posts packed
    with eos (data_iac.py:43-46).  Checked: piece ids, every training line re-encodes to at most 64 pieces, the rows of valid.npy are the
    capped encodings, and the committed configs[0] ids (tests/golden/configs0_ids.npz) are what this flow produces for the same posts."""
    import csv, json
    from argsim_amd import data_prep, util_sp
    csv.field_size_limit(1 << 30)
    with open(REF_POSTS, newline='') as f:
        posts = [r['post'].replace('\n', ' ').strip() for r in csv.DictReader(f)]
    posts = [p for p in posts if p]
    raw = tmp_path / 'discussions'; raw.mkdir()
    for i in range(0, len(posts), 100):          # 20 "discussions" of 100 posts: [id, side, author, text, annotations, parent, category, time]
        json.dump([[[j, 's', 'a', posts[j], [], None, 'c', 0] for j in range(i, min(i + 100, len(posts)))], {}, {}], open(raw / ('%04d.json' % i), 'w'))
    val = tmp_path / 'val.txt'
    open(val, 'w').write('\n'.join(posts[:50]) + '\n')
    out = tmp_path / 'data'
    vocab = data_prep.prep_iac(str(raw), str(val), str(out), cap=64)
    assert vocab.get_piece_size() == 8192 and (vocab.unk_id(), vocab.eos_id(), vocab.bos_id()) == (0, 1, 2)
    train = open(out / 'train.txt').read().splitlines()
    assert len(train) == len(posts)
    assert all(0 < len(vocab.encode_as_ids(t)) <= 64 for t in train[:300])
    valid = np.load(out / 'valid.npy')
    assert valid.shape == (50, 64) and valid.dtype == np.int32
    for row, p in zip(valid, posts[:50]):
        ids = util_sp.encode_capped(vocab, util_io_clean(p), cap=64)
        assert list(row[:len(ids)]) == ids and (row[len(ids):] == 1).all()
    # the committed configs[0] ids come from the same trainer flags and the same cap on the same posts (make_configs0_golden.py trains on
    # the posts in csv order, prep_iac on the same order through the sorted discussion files): the same model, the same ids
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'configs0_ids.npz'))['ids']
    mine = [util_sp.encode_capped(vocab, p, cap=64) for p in posts[:40]]
    same = sum(list(g[:len(m)]) == m and (g[len(m):] == 1).all() for g, m in zip(gold[:40], mine))
    assert same >= 38, same          # (clean() of data_iac.py may touch a post the golden script encoded as it stands)


def util_io_clean(x):
    from argsim_amd.util_io import clean
    return clean(x)
