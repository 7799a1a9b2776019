"""SentencePiece tokenisation helpers (counterpart of reference src/util_sp.py).

Same functions, arguments and fall-back order as the reference.  The reference splits sentences with
``nltk.tokenize.sent_tokenize`` (util_sp.py:1) which is not installed here; ``sent_split`` below is
a small rule-based splitter used ONLY on the cap fall-back path (text longer than ``cap`` pieces).
Its boundaries can differ from NLTK's punkt model on abbreviations -- "parity unpinned" for that
branch (no NLTK, no fixtures in the reference)."""
import re

import numpy as np

from .util_np import vpack

_ABBR = {'mr', 'mrs', 'ms', 'dr', 'prof', 'sr', 'jr', 'st', 'vs', 'etc', 'e.g', 'i.e', 'u.s', 'inc', 'no', 'fig', 'al'}
_BOUNDARY = re.compile(r'([.!?]+["\')\]]*)\s+')


def sent_split(text):
    """-> list of sentences.  Splits after ., ! or ? (plus closing quotes/brackets) followed by
    whitespace, unless the token before the period is a known abbreviation or a single letter."""
    out, start = [], 0
    for m in _BOUNDARY.finditer(text):
        end = m.end(1)
        head = text[start:end]
        if m.group(1).startswith('.'):
            words = head[:-len(m.group(1))].split()
            last = words[-1].lower().strip('("\'[') if words else ''
            if last in _ABBR or (len(last) == 1 and last.isalpha()):
                continue
        out.append(head.strip())
        start = m.end()
    tail = text[start:].strip()
    if tail:
        out.append(tail)
    return out


def load_spm(path):
    """-> SentencePieceProcessor (util_sp.py:6-14)"""
    from sentencepiece import SentencePieceProcessor
    spm = SentencePieceProcessor()
    spm.load(path)
    return spm


def spm(name, path, size=8192, bos=2, eos=1, unk=0, coverage=0.9995):
    """trains a SentencePiece model on the text file ``path``, saves ``name``.model/.vocab and returns
    the loaded processor (util_sp.py:17-39; same trainer flags, ids unk=0 eos=1 bos=2)."""
    from sentencepiece import SentencePieceTrainer
    SentencePieceTrainer.train(
        "--model_prefix={} --input={} --vocab_size={} --bos_id={} --eos_id={} --unk_id={} "
        "--unk_surface=☹ --character_coverage={}".format(name, path, size, bos, eos, unk, coverage))
    return load_spm(name + ".model")


def _capped(enc, text, cap, last_resort):
    """shared fall-back ladder of the three encode_capped* functions (util_sp.py:42-111):
    whole text -> first n sentences with n = floor(#sents * cap / len) counted down -> last resort."""
    ids = enc(text)
    if _fits(ids, cap):
        return ids
    sents = sent_split(text)
    n = int(len(sents) * cap / _length(ids))
    while 0 < n:
        ids = enc(" ".join(sents[:n]))
        if _fits(ids, cap):
            return ids
        n -= 1
    return last_resort(ids)


def _length(ids):
    return max(len(ids[0]), len(ids[1])) if isinstance(ids, tuple) else len(ids)


def _fits(ids, cap):
    return _length(ids) <= cap


def encode_capped(vocab, text, cap=512):
    """list of ids no longer than ``cap``: the whole text, else the first few sentences, else the
    hard-truncated first sentence (util_sp.py:42-63)"""
    return _capped(vocab.encode_as_ids, text, cap, lambda ids: ids[:cap])


def encode_capped_sample(vocab, text, cap=512):
    """like encode_capped with sampled segmentation (nbest=-1, alpha=0.5); falls back to the
    deterministic encode_capped when nothing fits (util_sp.py:66-87)"""
    enc = lambda x: vocab.sample_encode_as_ids(x, -1, 0.5)  # noqa: E731
    return _capped(enc, text, cap, lambda ids: encode_capped(vocab, text, cap))


def encode_capped_sample_pair(vocab, text, cap=512):
    """two independent sampled segmentations of the same (possibly shortened) text
    (util_sp.py:90-111); falls back to (det, det)"""
    enc = lambda x: (vocab.sample_encode_as_ids(x, -1, 0.5), vocab.sample_encode_as_ids(x, -1, 0.5))  # noqa: E731

    def det(_):
        ids = encode_capped(vocab, text, cap)
        return ids, ids
    return _capped(enc, text, cap, det)


def encode(vocab, sents, length=None, dtype=np.int32):
    """rank-2 id array padded with eos to ``length`` or the longest row (util_sp.py:114-125)"""
    sents = list(map(vocab.encode_as_ids, sents))
    if length is None:
        length = max(map(len, sents))
    return vpack(sents, (len(sents), length), vocab.eos_id(), dtype)


def decode(vocab, array):
    """ids -> text, cut at the first eos; higher ranks yield a generator (util_sp.py:128-140)"""
    array = np.asarray(array)
    if 1 < array.ndim:
        return (decode(vocab, arr) for arr in array)
    ids = list(map(int, array))
    if vocab.eos_id() in ids:
        ids = ids[:ids.index(vocab.eos_id())]
    return vocab.decode_ids(ids)
