"""TEST INFRASTRUCTURE ONLY -- training-step oracle (torch CPU, float64, autograd).

PARITY UNPINNED (see numpy_oracle.py).  Restates nolboSingleObject_modelnet_category_{VAE,AE}.fit
(reference src/module/nolbo.py:1411-1447 / 1230-1258) with the layer graph of
src/net_core/autoencoder3D.py in training mode (BatchNormalization with batch statistics, biased
variance, momentum 0.99), and Keras' Adam (beta1 0.9, beta2 0.999, epsilon 1e-7, no amsgrad).
The forward pass must agree with oracle/numpy_oracle.py (tests/test_oracle.py checks it); gradients
come from autograd, so this file contains no hand-derived backward formulas.

Only tests/ may import this; the product never does.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS, BN_MOM = 1e-3, 0.99
_DTYPE = [torch.float64]      # float64 = the oracle; bench.py's training cpu_baseline leg times the same graph in float32 (dtype= of fit_step)


def _t(a, grad=False):
    t = torch.tensor(np.asarray(a), dtype=_DTYPE[0])
    t.requires_grad_(grad)
    return t


def _act(x, kind):
    if kind == 'elu':
        return F.elu(x)
    if kind == 'relu':
        return F.relu(x)
    if kind == 'lrelu':
        return F.leaky_relu(x, 0.3)
    return x


def _conv_same(x, w, s):
    """x [B,D,H,W,C] channels-last; w Keras [k,k,k,Ci,Co]; TF SAME for k=4: s=2 pads (1,1), s=1 pads (1,2)."""
    xt = x.permute(0, 4, 1, 2, 3)
    wt = w.permute(4, 3, 0, 1, 2)
    pad = (1, 1) * 3 if s == 2 else (1, 2) * 3
    return F.conv3d(F.pad(xt, pad), wt, stride=s).permute(0, 2, 3, 4, 1)


def _convT_same(x, w, s):
    """w Keras [k,k,k,Co,Ci]."""
    xt = x.permute(0, 4, 1, 2, 3)
    wt = w.permute(4, 3, 0, 1, 2)
    n = x.shape[1] * s
    return F.conv_transpose3d(xt, wt, stride=s, padding=1)[:, :, :n, :n, :n].permute(0, 2, 3, 4, 1)


def _bn_train(x, gamma, beta, stats, name):
    axes = tuple(range(x.dim() - 1))
    m = x.mean(dim=axes)
    v = ((x - m) ** 2).mean(dim=axes)
    stats[name] = (m.detach().numpy(), v.detach().numpy())
    return gamma * (x - m) / torch.sqrt(v + BN_EPS) + beta


def forward_train(config, P, x, y, eps, variational=True, drop_mask=None, drop_scale=1.0, latent_fn=None, aux=None):
    """P: dict name -> torch tensor ('enc/...' and 'dec/...').  Returns (loss_kl, loss_shape, probs, stats).
    latent_fn(enc_out) -> (z_input, extra_loss): replaces the built-in latent algebra (the class-conditional prior model,
    reference nolbo.py:1620-1676); `loss_kl` is then extra_loss."""
    enc, dec = config['encoder'], config['decoder']
    L = config['z_category_dim']
    stats = {}
    h = x
    fl, st = enc['filter_num_list'], enc['strides_list']
    for i in range(len(fl) - 1):
        h = _conv_same(h, P['enc/conv%d/kernel' % i], st[i])
        h = _act(_bn_train(h, P['enc/bn%d/gamma' % i], P['enc/bn%d/beta' % i], stats, 'enc/bn%d' % i), enc['activation'])
    i = len(fl) - 1
    e = _conv_same(h, P['enc/conv%d/kernel' % i], st[i]).mean(dim=(1, 2, 3))
    if latent_fn is not None:
        z, kl = latent_fn(e)
    elif variational:
        mu, lv = e[:, :L], torch.clamp(e[:, L:2 * L], -10.0, 10.0)
        z = mu + torch.sqrt(torch.exp(lv)) * eps
        kl = (0.5 * (0.0 - lv) + (torch.exp(lv) + mu ** 2) / 2.0 - 0.5).sum(-1).mean()
    else:
        z, kl = e, torch.zeros((), dtype=_DTYPE[0])
    if drop_mask is not None:
        z = z * drop_mask * drop_scale
    if aux is not None:                      # the encoder output and the decoder input, for the builder-level model(x, training=True) tests
        aux['enc_out'], aux['z'] = e.detach().numpy().copy(), z.detach().numpy().copy()
    side = dec['output_shape'][0] // int(np.prod(dec['strides_list']))
    ch = max(dec['filter_num_list'][0] // 64, 8)
    t = z @ P['dec/dense/kernel'] + P['dec/dense/bias']
    t = _act(_bn_train(t, P['dec/bn_dense/gamma'], P['dec/bn_dense/beta'], stats, 'dec/bn_dense'), dec['activation'])
    t = t.reshape(-1, side, side, side, ch)
    fl, st = dec['filter_num_list'], dec['strides_list']
    for i in range(len(fl) - 1):
        t = _convT_same(t, P['dec/convT%d/kernel' % i], st[i])
        t = _act(_bn_train(t, P['dec/bnT%d/gamma' % i], P['dec/bnT%d/beta' % i], stats, 'dec/bnT%d' % i), dec['activation'])
    i = len(fl) - 1
    logits = _convT_same(t, P['dec/convT%d/kernel' % i], st[i])
    p = torch.sigmoid(logits)
    # binary_loss, function.py:73-82 (gamma 0.6 at nolbo.py:1432).  tf.clip_by_value passes the gradient where
    # min <= x <= max and blocks it outside; the reference evaluates that in float32, where sigmoid saturates to
    # exactly 0.99999988 == 1 - 1e-7 for logits in ~[15.9, 16.6] (gradient PASSES) and to 1.0 above (blocked).  The
    # mask is therefore taken on the float32-rounded probability; values stay float64.
    l32 = logits.detach().to(torch.float32)
    p32 = 1.0 / (1.0 + torch.exp(-l32))            # the reference's float32 sigmoid, evaluated in float32
    inside = (p32 >= torch.tensor(1e-7, dtype=torch.float32)) & (p32 <= torch.tensor(1.0, dtype=torch.float32) - torch.tensor(1e-7, dtype=torch.float32))
    q = torch.clamp(p, 1e-7, 1.0 - 1e-7).detach() + (p - p.detach()) * inside.to(p.dtype)   # clipped value, masked gradient
    B = x.shape[0]
    bce = -(0.6 * y * torch.log(q) + 0.4 * (1.0 - y) * torch.log(1.0 - q)).reshape(B, -1).sum(-1)
    return kl, bce.mean(), p, stats


def trainable_names(P):
    return [k for k in P if not k.endswith(('moving_mean', 'moving_variance'))]


def fit_step(config, enc_p, dec_p, x, y, eps, adam_state=None, lr=1e-4, variational=True, drop_mask=None, drop_scale=1.0, dtype=None):
    """One reference training step.  Returns dict with losses, grads, updated params, updated moving stats, adam state.
    dtype=torch.float32: the same graph in the reference's own precision (timing leg of bench.py --mode train; the tests use float64)."""
    if dtype is not None:
        _DTYPE.insert(0, dtype)
        try:
            return fit_step(config, enc_p, dec_p, x, y, eps, adam_state, lr, variational, drop_mask, drop_scale)
        finally:
            _DTYPE.pop(0)
    P = {}
    for k, v in enc_p.items():
        P['enc/' + k] = _t(v, grad=not k.endswith(('moving_mean', 'moving_variance')))
    for k, v in dec_p.items():
        P['dec/' + k] = _t(v, grad=not k.endswith(('moving_mean', 'moving_variance')))
    aux = {}
    kl, shape, p, stats = forward_train(config, P, _t(x), _t(y), _t(eps), variational,
                                        None if drop_mask is None else _t(drop_mask), drop_scale, aux=aux)
    total = kl + shape if variational else shape       # nolbo.py:1436 / 1247
    names = trainable_names(P)
    if not variational:
        pass
    grads = torch.autograd.grad(total, [P[n] for n in names], allow_unused=True)
    g = {n: (np.zeros(P[n].shape) if gr is None else gr.numpy()) for n, gr in zip(names, grads)}
    st = adam_state or {'t': 0, 'm': {n: np.zeros(P[n].shape) for n in names}, 'v': {n: np.zeros(P[n].shape) for n in names}}
    t = st['t'] + 1
    b1, b2, e = 0.9, 0.999, 1e-7
    new = {k: v.detach().numpy().copy() for k, v in P.items()}
    m2, v2 = {}, {}
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)      # Keras Adam (non-amsgrad) update rule
    for n in names:
        m2[n] = b1 * st['m'][n] + (1 - b1) * g[n]
        v2[n] = b2 * st['v'][n] + (1 - b2) * g[n] ** 2
        new[n] = new[n] - lr_t * m2[n] / (np.sqrt(v2[n]) + e)
    for name, (bm, bv) in stats.items():
        new[name + '/moving_mean'] = new[name + '/moving_mean'] * BN_MOM + bm * (1 - BN_MOM)
        new[name + '/moving_variance'] = new[name + '/moving_variance'] * BN_MOM + bv * (1 - BN_MOM)
    pn = p.detach().numpy()
    yn = np.asarray(y, np.float64)
    B = pn.shape[0]
    yh = (pn >= 0.5).astype(np.float64)
    tp = (yn * yh).reshape(B, -1).sum(-1)
    fp = ((1 - yn) * yh).reshape(B, -1).sum(-1)
    fn = (yn * (1 - yh)).reshape(B, -1).sum(-1)
    return {'loss_kl': float(kl.detach()), 'loss_shape': float(shape.detach()), 'pr': float(np.mean(tp / (tp + fp + 1e-10))),
            'rc': float(np.mean(tp / (tp + fn + 1e-10))), 'grads': g, 'params': new, 'bn_stats': stats,
            'adam': {'t': t, 'm': m2, 'v': v2}, 'probs': pn, 'enc_out': aux['enc_out'], 'z': aux['z']}


def fit_step_category_only(config, enc_p, dec_p, mean_prior, logvar_prior, x, y, eps, eps_prior, noise=None, drop_keep=None,
                           drop_rate=0.0):
    """Float64 restatement of nolboSingleObject_modelnet_category_only.fit (reference nolbo.py:1620-1676) with every random
    draw injected: eps / eps_prior = the two sampling() normals (:1637, :1639), noise [B,L] in {0,1} or None = the
    posterior / prior mixing mask of :1642-1648 (None: the `np.random.rand() > 0.5` branch, z itself), drop_keep / drop_rate =
    the optional Dropout on z_input (:1650-1652).  mean_prior / logvar_prior [B,L] stand for the prior network's outputs
    (its MLP is stock autograd code on both sides): their gradients are returned next to the encoder / decoder ones.
      loss_kl  = mean_b KL(N(mean, exp lv) || N(mean_p, exp lv_p))        function.py:84-98, nolbo.py:1655-1658
      loss_reg = mean_b regulizer_loss(mean_p, lv_p, 2 L)                 function.py:40-71 (no class input), nolbo.py:1663-1666
      total    = loss_kl + loss_shape + 0.01 loss_reg                     nolbo.py:1668"""
    L = config['z_category_dim']
    P = {}
    for k, v in enc_p.items():
        P['enc/' + k] = _t(v, grad=not k.endswith(('moving_mean', 'moving_variance')))
    for k, v in dec_p.items():
        P['dec/' + k] = _t(v, grad=not k.endswith(('moving_mean', 'moving_variance')))
    mp, lp = _t(mean_prior, grad=True), _t(logvar_prior, grad=True)
    e_, ep_ = _t(eps), _t(eps_prior)
    nz = None if noise is None else _t(noise)
    keep = None if drop_keep is None else _t(drop_keep)
    parts = {}

    def latent(e):
        mean, lv = e[:, :L], torch.clamp(e[:, L:2 * L], -10.0, 10.0)
        z = mean + torch.sqrt(torch.exp(lv)) * e_
        z_prior = mp + torch.sqrt(torch.exp(lp)) * ep_
        z_in = z if nz is None else torch.where(nz == 1.0, z, z_prior)
        if keep is not None:
            z_in = z_in * keep / (1.0 - drop_rate)
        kl = (0.5 * (lp - lv) + (torch.exp(lv) + (mean - mp) ** 2) / (2.0 * torch.exp(lp)) - 0.5).sum(-1).mean()
        B = mp.shape[0]
        zm = mp.reshape(B, 1, L).repeat(1, B, 1)                       # [i][j] = mean_p[i]
        zl = lp.reshape(B, 1, L).repeat(1, B, 1)                       # [i][j] = logvar_p[i]
        diff = (torch.abs(zm - zm.transpose(0, 1)) / torch.exp(0.5 * zl)).sum(-1) - 2.0 * L
        reg = torch.where(diff > 0, torch.zeros_like(diff), diff ** 2).sum(-1).mean()
        parts['kl'], parts['reg'] = kl, reg
        return z_in, kl + 0.01 * reg

    extra, shape, p, stats = forward_train(config, P, _t(x), _t(y), None, latent_fn=latent)
    total = extra + shape
    names = trainable_names(P)
    grads = torch.autograd.grad(total, [P[n] for n in names] + [mp, lp], allow_unused=True)
    g = {n: (np.zeros(P[n].shape) if gr is None else gr.numpy()) for n, gr in zip(names, grads[:-2])}
    pn = p.detach().numpy()
    yn = np.asarray(y, np.float64)
    B = pn.shape[0]
    yh = (pn >= 0.5).astype(np.float64)
    tp = (yn * yh).reshape(B, -1).sum(-1)
    fp = ((1 - yn) * yh).reshape(B, -1).sum(-1)
    fn = (yn * (1 - yh)).reshape(B, -1).sum(-1)
    return {'loss_kl': float(parts['kl'].detach()), 'loss_reg': float(parts['reg'].detach()), 'loss_shape': float(shape.detach()),
            'pr': float(np.mean(tp / (tp + fp + 1e-10))), 'rc': float(np.mean(tp / (tp + fn + 1e-10))), 'grads': g,
            'grad_mean_prior': grads[-2].numpy(), 'grad_logvar_prior': grads[-1].numpy(), 'bn_stats': stats}


def eval_forward_f32(config, enc_p, dec_p, x, eps):
    """Inference forward of getEval(missing_prob=0) (nolbo.py:1463-1501) on torch-CPU float32 ops (F.conv3d /
    F.conv_transpose3d: the oneDNN kernel family TF-CPU dispatches to) -- the secondary CPU bracket of SURVEY §8(d).
    Returns (logits [B,D,D,D,1], bce [B], tp, fp, fn) as numpy."""
    enc, dec = config['encoder'], config['decoder']
    L = config['z_category_dim']
    f32 = torch.float32
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))

    def bn(h, P, pre):
        sc = T(P[pre + '/gamma']) / torch.sqrt(T(P[pre + '/moving_variance']) + BN_EPS)
        return h * sc + (T(P[pre + '/beta']) - T(P[pre + '/moving_mean']) * sc)

    with torch.no_grad():
        h = T(x)
        fl, st = enc['filter_num_list'], enc['strides_list']
        for i in range(len(fl) - 1):
            h = _act(bn(_conv_same(h, T(enc_p['conv%d/kernel' % i]), st[i]), enc_p, 'bn%d' % i), enc['activation'])
        i = len(fl) - 1
        e = _conv_same(h, T(enc_p['conv%d/kernel' % i]), st[i]).mean(dim=(1, 2, 3))
        mu, lv = e[:, :L], torch.clamp(e[:, L:2 * L], -10.0, 10.0)
        z = mu + torch.sqrt(torch.exp(lv)) * T(eps)
        side = dec['output_shape'][0] // int(np.prod(dec['strides_list']))
        ch = max(dec['filter_num_list'][0] // 64, 8)
        t = _act(bn(z @ T(dec_p['dense/kernel']) + T(dec_p['dense/bias']), dec_p, 'bn_dense'), dec['activation'])
        t = t.reshape(-1, side, side, side, ch)
        fl, st = dec['filter_num_list'], dec['strides_list']
        for i in range(len(fl) - 1):
            t = _act(bn(_convT_same(t, T(dec_p['convT%d/kernel' % i]), st[i]), dec_p, 'bnT%d' % i), dec['activation'])
        i = len(fl) - 1
        logits = _convT_same(t, T(dec_p['convT%d/kernel' % i]), st[i])
        p = torch.sigmoid(logits)
        y = T(x)
        q = torch.clamp(p, 1e-7, 1.0 - 1e-7)
        B = y.shape[0]
        bce = -(0.6 * y * torch.log(q) + 0.4 * (1.0 - y) * torch.log(1.0 - q)).reshape(B, -1).sum(-1)
        yh = (p >= 0.5).to(f32)
        tp = (y * yh).reshape(B, -1).sum(-1); fp = ((1 - y) * yh).reshape(B, -1).sum(-1); fn = (y * (1 - yh)).reshape(B, -1).sum(-1)
    return logits.numpy(), bce.numpy(), tp.numpy(), fp.numpy(), fn.numpy()
