"""BASELINE configs[0] as worded -- "src/config.json defaults, 1k-sentence IAC subset, SentencePiece vocab 8k, seq_len 64, batch 32" --
without shipping reference text or code: run in the BUILD container (the reference tree is read here, never on the GPU box).

    python tests/golden/make_configs0_golden.py

1. the posts of /root/reference/docs/results_iac/clustering.csv (column `post`: 1 901 cleaned IAC posts, the only corpus the
   reference tree holds) train a SentencePiece model with the reference's trainer flags (src/util_sp.py:24-39 through
   argsim_amd.util_sp.spm: vocab 8192, unk 0 / eos 1 / bos 2, unk surface, coverage 0.9995) -> tests/golden/configs0_vocab.model
   (our artefact: a model file, no text);
2. the first 1 000 posts are encoded with encode_capped(cap = 64) (src/util_sp.py:42-63; the sentence splitter of the cap fall-back
   is the NLTK-free stand-in: parity unpinned there) and packed with vpack -> tests/golden/configs0_ids.npz (int16 ids, eos-padded);
   src/data_iac.py:40-41 builds train.txt the same way: as the DECODE of the capped ids, which is what the GPU test does;
3. tests/golden/make_oracle_golden.py cfg0real then writes the float64 oracle's fixture of one batch-32 training step at the
   config.json dimensions over rows 0..31 of those ids."""
import csv
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from argsim_amd.util_np import vpack  # noqa: E402
from argsim_amd.util_sp import encode_capped, spm  # noqa: E402

csv.field_size_limit(1 << 30)
with open('/root/reference/docs/results_iac/clustering.csv', newline='') as f:
    posts = [r['post'].replace('\n', ' ').strip() for r in csv.DictReader(f)]
posts = [p for p in posts if p]
print(len(posts), 'posts,', sum(map(len, posts)), 'characters')
with tempfile.TemporaryDirectory() as tmp:
    txt = os.path.join(tmp, 'iac.txt')
    with open(txt, 'w') as f:
        f.write('\n'.join(posts) + '\n')
    vocab = spm(os.path.join(HERE, 'configs0_vocab'), txt)
os.remove(os.path.join(HERE, 'configs0_vocab.vocab'))          # (the piece list: text, not needed)
assert vocab.get_piece_size() == 8192 and (vocab.unk_id(), vocab.eos_id(), vocab.bos_id()) == (0, 1, 2)
rows = [encode_capped(vocab, p, cap=64) for p in posts[:1000]]
assert all(0 < len(r) <= 64 for r in rows)
ids = vpack(rows, (len(rows), 64), vocab.eos_id(), np.int32)
print('pieces per post: mean %.1f, full rows %d' % (np.mean([len(r) for r in rows]), sum(len(r) == 64 for r in rows)))
np.savez_compressed(os.path.join(HERE, 'configs0_ids.npz'), ids=ids.astype(np.int16))
