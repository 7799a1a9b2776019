"""checkpoint save / restore (counterpart of tf.train.Saver in reference src/train.py:92-96,121 and
src/eval_embed_reason.py:24-27).

Native format: one ``.npz`` keyed by variable name, holding every model variable in its natural
(checkpoint) layout, ``global_step`` and, for training checkpoints, the Adam slots under
``train/<name>/Adam`` and ``train/<name>/Adam_1`` (TF's slot naming, model.py:188-189).  Restore is
name-based and partial like ``Saver().restore`` on an 'infer' graph: missing slots are left
untouched.

``to_tf_names`` / ``from_tf_names`` convert GRU variables to and from the canonical tensors that
TF 1.x's CudnnGRU saveable writes (names quoted from memory of
tf.contrib.cudnn_rnn, NOT verifiable offline):
    <scope>/cudnn_gru/rnn/multi_rnn_cell/cell_<k>/cudnn_compatible_gru_cell/
        gates/kernel (In+D, 2D) [r | u]      gates/bias (2D) = bW + bR
        candidate/input_projection/{kernel (In,D), bias (D)}
        candidate/hidden_projection/{kernel (D,D), bias (D)}
``save_tf`` / ``restore_tf`` write and read the TF V2 checkpoint files themselves (``tf_bundle.py``:
<prefix>.index + <prefix>.data-00000-of-00001 + ``checkpoint``) under those names, with the Adam slots of
the GRU weights in the flat cuDNN "opaque kernel" layout TF keeps them in (all weight matrices
W_r,W_u,W_n,R_r,R_u,R_n layer by layer, then all biases) -- SURVEY 8(f) item 4.  Byte compatibility with a
real TF process is unverified here (no TensorFlow, no reference checkpoint): parity unpinned."""
import re

import numpy as np

from .model import ADAM_M, ADAM_V, PARAM


def state_dict(vae, slots=True):
    out = {'global_step': np.asarray(vae.step, np.int64)}
    for k, v in vae.get_params(PARAM).items():
        out[k] = v
    if slots:
        for k, v in vae.get_params(ADAM_M).items():
            out['train/%s/Adam' % k] = v
        for k, v in vae.get_params(ADAM_V).items():
            out['train/%s/Adam_1' % k] = v
    return out


def audit(vae, sd):
    """(missing, unexpected, misshapen) of a native-named state dict against the model's variable table"""
    known = set(vae.names) | {'global_step'}
    known |= {'train/%s/%s' % (k, s) for k in vae.names for s in ('Adam', 'Adam_1')} | {'train/beta1_power', 'train/beta2_power'}
    missing = [k for k in vae.names if k not in sd]
    unexpected = sorted(k for k in sd if k not in known)
    misshapen = []
    for k in sd:
        base = k[len('train/'):].rsplit('/', 1)[0] if k.startswith('train/') and k.count('/') > 1 else k
        if base in vae.shapes and tuple(np.shape(sd[k])) != tuple(vae.shapes[base]):
            misshapen.append((k, tuple(np.shape(sd[k])), tuple(vae.shapes[base])))
    return missing, unexpected, misshapen


def load_state_dict(vae, sd, strict=True):
    """name-based restore.  Shapes are checked against the model BEFORE anything is written; strict also refuses
    missing model variables and keys the model does not know (a partial 'infer' restore passes strict=False)."""
    missing, unexpected, misshapen = audit(vae, sd)
    if misshapen:
        raise ValueError("checkpoint tensors do not fit the model (name, found, expected): %s" % misshapen[:5])
    if strict and (missing or unexpected):
        raise KeyError("checkpoint does not match the model: missing %s%s; unexpected %s%s"
                       % (missing[:5], ' ...' if len(missing) > 5 else '', unexpected[:5], ' ...' if len(unexpected) > 5 else ''))
    for k in vae.names:
        if k in sd:
            vae.set_tensor(k, sd[k], PARAM)
        if 'train/%s/Adam' % k in sd:
            vae.set_tensor(k, sd['train/%s/Adam' % k], ADAM_M)
        if 'train/%s/Adam_1' % k in sd:
            vae.set_tensor(k, sd['train/%s/Adam_1' % k], ADAM_V)
    if 'global_step' in sd:
        vae.step = int(sd['global_step'])


def save(vae, path, slots=True):
    """saver.save(sess, path, write_meta_graph=False)  (src/train.py:121)"""
    if not path.endswith('.npz'):
        path += '.npz'
    np.savez(path, **state_dict(vae, slots))
    return path


def restore(vae, path, strict=True):
    """saver.restore(sess, path)  (src/train.py:93-94).  A TF V2 checkpoint prefix (``<path>.index`` exists)
    is read through ``restore_tf``; otherwise the native ``.npz``."""
    import os
    if os.path.exists(path + '.index'):
        return restore_tf(vae, path, strict)
    if not path.endswith('.npz'):
        path += '.npz'
    with np.load(path, allow_pickle=False) as f:
        load_state_dict(vae, {k: f[k] for k in f.files}, strict)


# ------------------------------------------------------------------ TF canonical names
_GRU = re.compile(r'^(encode/rnn\d+/(?:fwd|bwd)|decode/rnn/l(\d+))/(W|R|bW|bR)$')


def _tf_scope(name):
    m = _GRU.match(name)
    if m.group(2) is None:          # encoder: one single-layer CudnnGRU per direction (model.py:120-121)
        return m.group(1), 0
    return 'decode/rnn', int(m.group(2)) - 1     # decoder: one L-layer CudnnGRU (model.py:160)


def to_tf_names(sd):
    """native state dict -> dict keyed like a TF1 checkpoint of the reference graph"""
    out, groups = {}, {}
    for k, v in sd.items():
        if _GRU.match(k):
            scope, cell = _tf_scope(k)
            groups.setdefault((scope, cell), {})[k.rsplit('/', 1)[1]] = np.asarray(v)
        else:
            out[k] = v
    for (scope, cell), g in groups.items():
        D = g['R'].shape[1]
        W, R, bW, bR = g['W'], g['R'], g['bW'], g['bR']
        p = '%s/cudnn_gru/rnn/multi_rnn_cell/cell_%d/cudnn_compatible_gru_cell/' % (scope, cell)
        out[p + 'gates/kernel'] = np.concatenate([W[:2 * D].T, R[:2 * D].T], axis=0)
        out[p + 'gates/bias'] = bW[:2 * D] + bR[:2 * D]
        out[p + 'candidate/input_projection/kernel'] = W[2 * D:].T
        out[p + 'candidate/input_projection/bias'] = bW[2 * D:]
        out[p + 'candidate/hidden_projection/kernel'] = R[2 * D:].T
        out[p + 'candidate/hidden_projection/bias'] = bR[2 * D:]
    return out


def from_tf_names(sd, names):
    """inverse of to_tf_names for the variables in ``names``.  The r/u biases are stored summed in
    TF's canonical form; they are restored into bW with bR = 0 (the same function)."""
    out = {k: v for k, v in sd.items() if 'cudnn_compatible_gru_cell' not in k}
    for k in names:
        if not _GRU.match(k) or not k.endswith('/W'):
            continue
        scope, cell = _tf_scope(k)
        p = '%s/cudnn_gru/rnn/multi_rnn_cell/cell_%d/cudnn_compatible_gru_cell/' % (scope, cell)
        gk, gb = np.asarray(sd[p + 'gates/kernel']), np.asarray(sd[p + 'gates/bias'])
        D = gk.shape[1] // 2
        In = gk.shape[0] - D
        base = k[:-1]
        out[base + 'W'] = np.concatenate([gk[:In].T, np.asarray(sd[p + 'candidate/input_projection/kernel']).T], 0)
        out[base + 'R'] = np.concatenate([gk[In:].T, np.asarray(sd[p + 'candidate/hidden_projection/kernel']).T], 0)
        out[base + 'bW'] = np.concatenate([gb, np.asarray(sd[p + 'candidate/input_projection/bias'])])
        out[base + 'bR'] = np.concatenate([np.zeros_like(gb), np.asarray(sd[p + 'candidate/hidden_projection/bias'])])
    return out


# ------------------------------------------------------------------ TF V2 checkpoint files
def _opaque_groups(names):
    """{tf scope: [per layer: native base name 'encode/rnn1/fwd/' ...]} in cuDNN layer order"""
    groups = {}
    for k in names:
        if _GRU.match(k) and k.endswith('/W'):
            scope, cell = _tf_scope(k)
            groups.setdefault(scope, {})[cell] = k[:-1]
    return {sc: [g[c] for c in sorted(g)] for sc, g in groups.items()}


def _to_opaque(get, bases):
    """flat cuDNN GRU parameter buffer of one CudnnGRU op from native (3D, In) gate-stacked tensors"""
    ws, bs = [], []
    for b in bases:
        W, R, bW, bR = (np.asarray(get(b + n)) for n in ('W', 'R', 'bW', 'bR'))
        D = R.shape[1]
        ws += [W[g * D:(g + 1) * D].ravel() for g in range(3)] + [R[g * D:(g + 1) * D].ravel() for g in range(3)]
        bs += [bW[g * D:(g + 1) * D] for g in range(3)] + [bR[g * D:(g + 1) * D] for g in range(3)]
    return np.concatenate(ws + bs).astype(np.float32)


def _from_opaque(flat, bases, shapes):
    out, pos = {}, 0
    flat = np.asarray(flat).ravel()
    for b in bases:
        D, In = shapes[b + 'R'][1], shapes[b + 'W'][1]
        W = flat[pos:pos + 3 * D * In].reshape(3 * D, In); pos += 3 * D * In
        R = flat[pos:pos + 3 * D * D].reshape(3 * D, D); pos += 3 * D * D
        out[b + 'W'], out[b + 'R'] = W, R
    for b in bases:
        D = shapes[b + 'R'][1]
        out[b + 'bW'] = flat[pos:pos + 3 * D]; pos += 3 * D
        out[b + 'bR'] = flat[pos:pos + 3 * D]; pos += 3 * D
    assert pos == flat.size, "opaque kernel size does not match the GRU geometry"
    return out


def save_tf(vae, prefix, slots=True):
    """saver.save(sess, prefix, write_meta_graph=False) as TF V2 checkpoint files (src/train.py:121)"""
    from . import tf_bundle
    sd = state_dict(vae, slots)
    out = to_tf_names({k: v for k, v in sd.items() if not k.startswith('train/')})
    if slots:
        groups = _opaque_groups(vae.names)
        in_gru = {b + n for bases in groups.values() for b in bases for n in ('W', 'R', 'bW', 'bR')}
        for slot in ('Adam', 'Adam_1'):
            for k in vae.names:
                if k not in in_gru:
                    out['train/%s/%s' % (k, slot)] = sd['train/%s/%s' % (k, slot)]
            for scope, bases in groups.items():
                out['train/%s/cudnn_gru/opaque_kernel/%s' % (scope, slot)] = _to_opaque(lambda n: sd['train/%s/%s' % (n, slot)], bases)
        t = float(vae.step) + 1.0
        out['train/beta1_power'] = np.asarray(0.9 ** t, np.float32)
        out['train/beta2_power'] = np.asarray(0.999 ** t, np.float32)
    tf_bundle.write_bundle(prefix, out)
    return prefix


def restore_tf(vae, prefix, strict=True):
    """saver.restore(sess, prefix) from TF V2 checkpoint files (src/train.py:93-94, eval_embed_reason.py:24-27).
    The CudnnGRU leaf names and gate order are quoted from memory of tf.contrib.cudnn_rnn (parity unpinned: no
    TensorFlow and no reference checkpoint here), so a mismatch must be loud: a file whose GRU tensors are absent
    or shaped differently raises with the offending names instead of loading the rest."""
    from . import tf_bundle
    raw = tf_bundle.read_bundle(prefix)
    want = [k for k in to_tf_names({n: np.zeros(vae.shapes[n], np.float32) for n in vae.names})]
    absent = [k for k in want if k not in raw]
    if absent:
        raise KeyError("TF checkpoint %r lacks %d of the %d tensors this model restores, e.g. %s; it holds e.g. %s"
                       % (prefix, len(absent), len(want), absent[:3], sorted(raw)[:3]))
    sd = from_tf_names({k: v for k, v in raw.items() if not k.startswith('train/')}, vae.names)
    groups = _opaque_groups(vae.names)
    in_gru = {b + n for bases in groups.values() for b in bases for n in ('W', 'R', 'bW', 'bR')}
    for slot in ('Adam', 'Adam_1'):
        for k in vae.names:
            if k not in in_gru and 'train/%s/%s' % (k, slot) in raw:
                sd['train/%s/%s' % (k, slot)] = raw['train/%s/%s' % (k, slot)]
        for scope, bases in groups.items():
            key = 'train/%s/cudnn_gru/opaque_kernel/%s' % (scope, slot)
            if key in raw:
                for n, v in _from_opaque(raw[key], bases, vae.shapes).items():
                    sd['train/%s/%s' % (n, slot)] = v
    load_state_dict(vae, sd, strict)
