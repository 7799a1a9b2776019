"""float64 numpy restatement of the reference VAE graph (TEST INFRASTRUCTURE ONLY).

Every function cites the lines of ``/root/reference/src/model.py`` (or
``util_tf.py``) it follows.  Plain loops over time; no torch, no GPU.
See ``oracle/__init__.py`` for the parity status ("parity unpinned").

Parameter naming (neutral, cuDNN gate order r,u,n stacked along rows):
  embed/embedding                      (V, D)      model.py:108-110
  encode/rnn{i}/{fwd,bwd}/{W,R,bW,bR}  W (3D,In) R (3D,D) biases (3D)   model.py:119-121
  latent/{mu,lv}/{kernel,bias}         (2D,R),(R)  model.py:149-150  (tf.layers.dense: y = x @ kernel + bias)
  latent/ex/{kernel,bias}              (R,D),(D)   model.py:156
  decode/rnn/l{i}/{W,R,bW,bR}          W (3D,D)    model.py:160
  decode/out/{kernel,bias}             (D,D),(D)   model.py:162
"""
import numpy as np

F64 = np.float64


# --------------------------------------------------------------------------- config
def make_cfg(dim_tgt=8192, dim_emb=512, dim_rep=1024, rnn_layers=3,
             accelerate=1e-4, learn_rate=1e-3, bos=2, eos=1):
    """defaults of vAe(), model.py:48-66."""
    return dict(dim_tgt=dim_tgt, dim_emb=dim_emb, dim_rep=dim_rep, rnn_layers=rnn_layers,
                accelerate=accelerate, learn_rate=learn_rate, bos=bos, eos=eos)


def param_shapes(cfg):
    V, D, R, L = cfg['dim_tgt'], cfg['dim_emb'], cfg['dim_rep'], cfg['rnn_layers']
    shp = {'embed/embedding': (V, D)}
    for i in range(1, L + 1):
        In = D if i == 1 else 2 * D
        for d in ('fwd', 'bwd'):
            p = 'encode/rnn%d/%s/' % (i, d)
            shp[p + 'W'] = (3 * D, In)
            shp[p + 'R'] = (3 * D, D)
            shp[p + 'bW'] = (3 * D,)
            shp[p + 'bR'] = (3 * D,)
    shp['latent/mu/kernel'] = (2 * D, R)
    shp['latent/mu/bias'] = (R,)
    shp['latent/lv/kernel'] = (2 * D, R)
    shp['latent/lv/bias'] = (R,)
    shp['latent/ex/kernel'] = (R, D)
    shp['latent/ex/bias'] = (D,)
    for i in range(1, L + 1):
        p = 'decode/rnn/l%d/' % i
        shp[p + 'W'] = (3 * D, D)
        shp[p + 'R'] = (3 * D, D)
        shp[p + 'bW'] = (3 * D,)
        shp[p + 'bR'] = (3 * D,)
    shp['decode/out/kernel'] = (D, D)
    shp['decode/out/bias'] = (D,)
    return shp


def init_params(cfg, seed=0, bias_scale=0.0):
    """reference initialisers: model.py:8-10 (glorot-uniform kernels, zero bias) and
    model.py:109-110 (embedding U(+-sqrt(6/(V/D+1)))).  ``bias_scale`` > 0 draws
    non-zero biases so that tests exercise the bias terms (a trained checkpoint has them).
    CudnnGRU initialises each gate matrix separately with the kernel initialiser
    (fan_in = input size, fan_out = num_units)."""
    rng = np.random.default_rng(seed)
    V, D = cfg['dim_tgt'], cfg['dim_emb']
    out = {}
    for name, shp in param_shapes(cfg).items():
        if name == 'embed/embedding':
            b = (6.0 / (V / D + 1.0)) ** 0.5
            out[name] = rng.uniform(-b, b, shp)
        elif len(shp) == 1:
            out[name] = bias_scale * rng.standard_normal(shp)
        elif name.endswith('/W') or name.endswith('/R'):
            fan_out, fan_in = D, shp[1]
            b = (6.0 / (fan_in + fan_out)) ** 0.5
            out[name] = rng.uniform(-b, b, shp)
        else:  # dense kernel (in, out)
            b = (6.0 / (shp[0] + shp[1])) ** 0.5
            out[name] = rng.uniform(-b, b, shp)
        out[name] = out[name].astype(F64)
    return out


# --------------------------------------------------------------------------- pieces
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def schedule(step, accelerate=1e-4, learn_rate=1e-3):
    """model.py:75-80.  returns (rate_keepwd, rate_anneal, rate_update)."""
    rate = accelerate * float(step)
    return sigmoid(rate), np.tanh(rate), learn_rate / (np.sqrt(rate) + 1.0)


def trim(x, eos):
    """util_tf.py:40-57.  x (S', B) time-major -> x[:max_len], not_eos[:max_len], len (B,)."""
    not_eos = x != eos
    len_seq = not_eos.sum(axis=0).astype(np.int32)
    max_len = int(len_seq.max()) if len_seq.size else 0
    return x[:max_len], not_eos[:max_len], len_seq


def reverse_sequence(x, lens):
    """tf.reverse_sequence(seq_axis=0, batch_axis=1): reverses the first len_b steps of column b."""
    y = x.copy()
    for b, n in enumerate(lens):
        y[:n, b] = x[:n, b][::-1]
    return y


def gru(x, W, R, bW, bR, h0=None):
    """one cuDNN GRU layer (reset-after), SURVEY 8a row 6 / model.py:15.
    x (S,B,In) -> hs (S,B,D), h_last (B,D)."""
    S, B, _ = x.shape
    D = R.shape[1]
    h = np.zeros((B, D), F64) if h0 is None else h0.astype(F64)
    hs = np.empty((S, B, D), F64)
    for t in range(S):
        gi = x[t] @ W.T + bW
        gh = h @ R.T + bR
        r = sigmoid(gi[:, :D] + gh[:, :D])
        u = sigmoid(gi[:, D:2 * D] + gh[:, D:2 * D])
        n = np.tanh(gi[:, 2 * D:] + r * gh[:, 2 * D:])
        h = (1.0 - u) * n + u * h
        hs[t] = h
    return hs, h


def prep_decoder_io(tgt_tm, not_eos, cfg, keep_mask=None):
    """model.py:91-95.  tgt_tm (S,B) trimmed time-major ids."""
    S, B = tgt_tm.shape
    msk_tgt = np.concatenate([np.ones((1, B), bool), not_eos], axis=0)
    gold = np.concatenate([tgt_tm, np.full((1, B), cfg['eos'], tgt_tm.dtype)], axis=0)
    lead = tgt_tm.copy()
    if keep_mask is not None:
        lead = lead * keep_mask.astype(lead.dtype)
    lead = np.concatenate([np.full((1, B), cfg['bos'], tgt_tm.dtype), lead], axis=0)
    return lead, gold, msk_tgt


def encoder(P, cfg, src_tm, len_src):
    """model.py:111-122,135.  src_tm (S,B) trimmed -> hs (S,B,2D), h (B,2D)."""
    E = P['embed/embedding']
    x = E[src_tm]
    for i in range(1, cfg['rnn_layers'] + 1):
        pf, pb = 'encode/rnn%d/fwd/' % i, 'encode/rnn%d/bwd/' % i
        fwd, _ = gru(x, P[pf + 'W'], P[pf + 'R'], P[pf + 'bW'], P[pf + 'bR'])
        bwd, _ = gru(reverse_sequence(x, len_src), P[pb + 'W'], P[pb + 'R'], P[pb + 'bW'], P[pb + 'bR'])
        x = np.concatenate([fwd, reverse_sequence(bwd, len_src)], axis=-1)
    hs = x
    B = src_tm.shape[1]
    h = hs[len_src - 1, np.arange(B)]
    return hs, h


def decoder_rnn(P, cfg, emb_tgt, state_in):
    """model.py:160: L-layer unidirectional CudnnGRU with initial_state (L,B,D)."""
    x = emb_tgt
    state_ex = []
    for i in range(1, cfg['rnn_layers'] + 1):
        p = 'decode/rnn/l%d/' % i
        x, hl = gru(x, P[p + 'W'], P[p + 'R'], P[p + 'bW'], P[p + 'bR'], state_in[i - 1])
        state_ex.append(hl)
    return x, np.stack(state_ex)


def forward(P, cfg, src, tgt, mode='train', step=0, keep_mask=None, eps=None):
    """the whole graph, model.py:75-185.

    src, tgt : (B, S') int32 row-major eos-padded (as util_np.vpack makes them)
    keep_mask: (S, B) {0,1} word-dropout keep mask on the TRIMMED time-major tgt (train only;
               stands for ``random_uniform < rate_keepwd``, model.py:94)
    eps      : (B, R) standard normal draw (train only; model.py:154)
    returns a dict named after the Record fields of model.py.
    """
    assert mode in ('train', 'valid', 'infer')
    P = {k: np.asarray(v, F64) for k, v in P.items()}
    D, L = cfg['dim_emb'], cfg['rnn_layers']
    o = {}
    o['rate_keepwd'], o['rate_anneal'], o['rate_update'] = schedule(step, cfg['accelerate'], cfg['learn_rate'])
    src_tm, msk_src, len_src = trim(np.asarray(src).T, cfg['eos'])
    tgt_tm, not_eos_tgt, len_tgt = trim(np.asarray(tgt).T, cfg['eos'])
    if mode == 'train':
        assert keep_mask is not None and keep_mask.shape == tgt_tm.shape
    lead, gold, msk_tgt = prep_decoder_io(tgt_tm, not_eos_tgt, cfg, keep_mask if mode == 'train' else None)
    o.update(len_src=len_src, len_tgt=len_tgt, lead=lead, gold=gold, msk_tgt=msk_tgt)
    E = P['embed/embedding']
    emb_tgt = E[lead]
    hs, h = encoder(P, cfg, src_tm, len_src)
    o['hs'], o['h_enc'] = hs, h
    mu = h @ P['latent/mu/kernel'] + P['latent/mu/bias']
    lv = h @ P['latent/lv/kernel'] + P['latent/lv/bias']
    z = mu.copy()
    if mode == 'train':
        assert eps is not None and eps.shape == lv.shape
        z = z + np.exp(0.5 * lv) * eps
    o.update(mu=mu, lv=lv, z=z)
    h0 = z @ P['latent/ex/kernel'] + P['latent/ex/bias']
    state_in = np.stack([h0] * L)
    hd, state_ex = decoder_rnn(P, cfg, emb_tgt, state_in)
    o.update(state_in=state_in, state_ex=state_ex)
    if mode != 'infer':
        hd = hd[msk_tgt]                       # boolean_mask: time-major compaction, model.py:161
    else:
        hd = hd.reshape(-1, D)
    hd = hd @ P['decode/out/kernel'] + P['decode/out/bias']
    logits = hd @ ((D ** -0.5) * E.T)          # model.py:166
    o['h_out'] = hd
    o['logits'] = logits
    o['pred'] = logits.argmax(-1).astype(np.int32)
    if mode != 'infer':
        labels = gold[msk_tgt]
        o['labels'] = labels
        o['errt_samp'] = (labels != o['pred']).astype(F64)
        o['errt'] = o['errt_samp'].mean()
        mx = logits.max(-1, keepdims=True)
        lse = mx[:, 0] + np.log(np.exp(logits - mx).sum(-1))
        o['loss_gen_samp'] = lse - logits[np.arange(len(labels)), labels]
        o['loss_gen'] = o['loss_gen_samp'].mean()
        o['loss_kld_samp'] = 0.5 * (mu * mu + np.exp(lv) - lv - 1.0)
        o['loss_kld'] = o['loss_kld_samp'].mean()
        o['loss'] = o['rate_anneal'] * o['loss_kld'] + o['loss_gen']
    return o


def adam_tf(params, grads, m, v, n_updates, lr, beta1=0.9, beta2=0.999, epsilon=1e-8):
    """tf.train.AdamOptimizer.apply (model.py:189).  ``n_updates`` = updates applied so far
    (beta powers start at beta^1).  epsilon is OUTSIDE the bias-corrected sqrt."""
    t = n_updates + 1
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    out_p, out_m, out_v = {}, {}, {}
    for k in params:
        g = np.asarray(grads[k], F64)
        out_m[k] = beta1 * m[k] + (1.0 - beta1) * g
        out_v[k] = beta2 * v[k] + (1.0 - beta2) * g * g
        out_p[k] = params[k] - lr_t * out_m[k] / (np.sqrt(out_v[k]) + epsilon)
    return out_p, out_m, out_v


def decode_greedy(P, cfg, z, steps=256):
    """model.py:204-219: greedy loop; stops when ALL rows emit eos; result excludes that step."""
    P = {k: np.asarray(v, F64) for k, v in P.items()}
    D, L = cfg['dim_emb'], cfg['rnn_layers']
    E = P['embed/embedding']
    b = len(z)
    x = np.full((1, b), cfg['bos'], np.int32)
    h0 = np.asarray(z, F64) @ P['latent/ex/kernel'] + P['latent/ex/bias']
    s = np.stack([h0] * L)
    ys = []
    for _ in range(steps):
        hd, s = decoder_rnn(P, cfg, E[x], s)
        hd = hd.reshape(-1, D) @ P['decode/out/kernel'] + P['decode/out/bias']
        logits = hd @ ((D ** -0.5) * E.T)
        x = logits.argmax(-1).astype(np.int32).reshape(1, b)
        if np.all(x == cfg['eos']):
            break
        ys.append(x)
    if not ys:
        return np.zeros((b, 0), np.int32)
    return np.concatenate(ys).T
