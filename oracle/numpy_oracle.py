"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the voxel VAE hot path (numpy, definition level).

PARITY UNPINNED: the reference (bogus2000/anytime-3D-reconstruction) is Python on
TensorFlow 2.x; TensorFlow is not installable here (no network, no wheel), the reference
ships no tests, golden vectors, weights or saved outputs.  This file
restates the reference's algorithm from its source text plus TensorFlow/Keras' documented
op semantics.  It is pinned only by agreement with two further independent statements
(oracle/voxvae_oracle.c and the torch-CPU functional cross-check in tests/).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product (anytime-3d-reconstruction_amd/) never does.

Every function cites the reference file:line it follows (paths relative to /root/reference).
dtype is a parameter: float64 gives the ground truth the golden fixtures are made from,
float32 mimics the reference's arithmetic type.
"""
import numpy as np

BN_EPS = 1e-3          # tf.keras.layers.BatchNormalization default epsilon
BN_MOMENTUM = 0.99     # ... default momentum
LRELU_ALPHA = 0.3      # tf.keras.layers.LeakyReLU default alpha


# ----------------------------------------------------------------------------- primitives
def same_pad(n, k, s):
    """TensorFlow 'SAME' rule: out = ceil(n/s); pad_total = max((out-1)*s + k - n, 0);
    pad_before = pad_total // 2 (the extra cell goes after)."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def conv3d_same(x, w, stride):
    """tf.keras.layers.Conv3D(padding='same', use_bias=False) -- autoencoder3D.py:27-30, 86-88.
    x [B,D,H,W,Ci], w [kd,kh,kw,Ci,Co] (Keras kernel layout) -> [B,D',H',W',Co].
    Cross-correlation: y[o] = sum_t xpad[s*o + t] . w[t]."""
    B, D, H, W, Ci = x.shape
    k = w.shape[0]
    od, pbd, pad_ = same_pad(D, k, stride)
    oh, pbh, pah = same_pad(H, k, stride)
    ow, pbw, paw = same_pad(W, k, stride)
    xp = np.zeros((B, D + pbd + pad_, H + pbh + pah, W + pbw + paw, Ci), x.dtype)
    xp[:, pbd:pbd + D, pbh:pbh + H, pbw:pbw + W, :] = x
    y = np.zeros((B, od, oh, ow, w.shape[4]), x.dtype)
    s = stride
    for td in range(k):
        for th in range(k):
            for tw in range(k):
                win = xp[:, td:td + s * od:s, th:th + s * oh:s, tw:tw + s * ow:s, :]
                y += win @ w[td, th, tw]
    return y


def conv3d_transpose_same(x, w, stride):
    """tf.keras.layers.Conv3DTranspose(padding='same', use_bias=False) -- autoencoder3D.py:42-45,
    129-132.  x [B,D,H,W,Ci], w [kd,kh,kw,Co,Ci] (Keras transposed-kernel layout) -> [B,sD,sH,sW,Co].
    Defined, as TensorFlow defines it, as the gradient of conv3d_same (output size n*s, kernel
    [k,k,k,Co,Ci] read as a forward kernel Co->Ci) with respect to its input:
    ypad[s*i + t] += x[i] . w[t]^T, then crop pad_before."""
    B, D, H, W, Ci = x.shape
    k = w.shape[0]
    s = stride
    n_out = [D * s, H * s, W * s]
    pads = [same_pad(n, k, s) for n in n_out]
    for n_in, (o, _, _) in zip((D, H, W), pads):
        assert o == n_in
    yp = np.zeros((B,) + tuple(n + p[1] + p[2] for n, p in zip(n_out, pads)) + (w.shape[3],), x.dtype)
    for td in range(k):
        for th in range(k):
            for tw in range(k):
                yp[:, td:td + s * D:s, th:th + s * H:s, tw:tw + s * W:s, :] += x @ w[td, th, tw].T
    (_, pbd, _), (_, pbh, _), (_, pbw, _) = pads
    return yp[:, pbd:pbd + n_out[0], pbh:pbh + n_out[1], pbw:pbw + n_out[2], :].copy()


def batchnorm_inference(x, gamma, beta, mean, var, eps=BN_EPS):
    """BatchNormalization(training=False): gamma*(x-mean)/sqrt(var+eps)+beta on the last axis
    -- autoencoder3D.py:31, 46, 62."""
    dt = x.dtype
    return gamma.astype(dt) * (x - mean.astype(dt)) / np.sqrt(var.astype(dt) + dt.type(eps)) + beta.astype(dt)


def batchnorm_training(x, gamma, beta, eps=BN_EPS):
    """BatchNormalization(training=True): batch mean / biased batch variance over every axis but
    the last.  Returns (y, batch_mean, batch_var); the caller updates the moving statistics with
    moving = moving*momentum + batch*(1-momentum)  (Keras uses the biased variance there too)."""
    dt = x.dtype
    axes = tuple(range(x.ndim - 1))
    m = x.mean(axis=axes)
    v = ((x - m) ** 2).mean(axis=axes)
    return gamma.astype(dt) * (x - m) / np.sqrt(v + dt.type(eps)) + beta.astype(dt), m, v


def activation(x, kind):
    """autoencoder3D.py:33-38 -- ELU(alpha=1) / ReLU / LeakyReLU(alpha=0.3)."""
    if kind == 'elu':
        return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))
    if kind == 'relu':
        return np.maximum(x, 0)
    if kind == 'lrelu':
        return np.where(x > 0, x, x * x.dtype.type(LRELU_ALPHA))
    return x


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# ----------------------------------------------------------------------------- net_core
def decoder_seed_shape(structure):
    """autoencoder3D.py:115-120."""
    side = int(structure['output_shape'][0] // int(np.prod(structure['strides_list'])))
    ch = int(structure['filter_num_list'][0] // 64)
    return side, max(ch, 8)


def _bn(p, prefix, x, training):
    g, b = p[prefix + '/gamma'], p[prefix + '/beta']
    if training:
        y, m, v = batchnorm_training(x, g, b)
        return y, (m, v)
    return batchnorm_inference(x, g, b, p[prefix + '/moving_mean'], p[prefix + '/moving_variance']), None


def encoder3D_forward(structure, p, x, training=False, dtype=np.float64, return_stats=False):
    """encoder3D -- autoencoder3D.py:72-102.  x [B,D,D,D,1] -> [B,E]."""
    h = x.astype(dtype)
    fl, st = structure['filter_num_list'], structure['strides_list']
    stats = {}
    for i in range(len(fl) - 1):                                   # :84-85 conv3DEnc
        h = conv3d_same(h, p['conv%d/kernel' % i].astype(dtype), st[i])
        h, bs = _bn(p, 'bn%d' % i, h, training)
        stats['bn%d' % i] = bs
        h = activation(h, structure['activation'])
    i = len(fl) - 1
    h = conv3d_same(h, p['conv%d/kernel' % i].astype(dtype), st[i])  # :86-88, no BN, no act
    if structure['final_pool'] == 'average':                       # :90-91
        h = h.mean(axis=(1, 2, 3))
    elif structure['final_pool'] == 'max':                         # :92-93
        h = h.max(axis=(1, 2, 3))
    if structure['final_activation'] == 'sigmoid':                 # :97-98
        h = sigmoid(h)
    return (h, stats) if return_stats else h


def decoder3D_forward(structure, p, z, training=False, dtype=np.float64, return_stats=False):
    """decoder3D -- autoencoder3D.py:104-139.  z [B,L] -> (logits, probabilities) [B,D,D,D,1]."""
    side, ch = decoder_seed_shape(structure)
    stats = {}
    h = z.astype(dtype) @ p['dense/kernel'].astype(dtype) + p['dense/bias'].astype(dtype)   # :59-61
    h, bs = _bn(p, 'bn_dense', h, training)                                                  # :62
    stats['bn_dense'] = bs
    h = activation(h, structure['activation'])                                               # :63-68
    h = h.reshape(-1, side, side, side, ch)                                                  # :125
    fl, st = structure['filter_num_list'], structure['strides_list']
    for i in range(len(fl) - 1):                                                             # :127-128
        h = conv3d_transpose_same(h, p['convT%d/kernel' % i].astype(dtype), st[i])
        h, bs = _bn(p, 'bnT%d' % i, h, training)
        stats['bnT%d' % i] = bs
        h = activation(h, structure['activation'])
    i = len(fl) - 1
    logits = conv3d_transpose_same(h, p['convT%d/kernel' % i].astype(dtype), st[i])          # :129-132
    if structure['final_activation'] == 'sigmoid':                                           # :134-136
        out = sigmoid(logits)
    else:
        out = logits
    return (logits, out, stats) if return_stats else (logits, out)


# ----------------------------------------------------------------------------- module/function.py
def sampling(mu, logvar, eps):
    """function.py:35-38 with the tf.random.normal draw injected: mu + sqrt(exp(logVar))*eps."""
    return mu + np.sqrt(np.exp(logvar)) * eps.astype(mu.dtype)


def kl_loss(mean, logvar, mean_t, logvar_t):
    """function.py:84-98."""
    return np.sum(0.5 * (logvar_t - logvar) + (np.exp(logvar) + np.square(mean - mean_t)) / (2.0 * np.exp(logvar_t)) - 0.5,
                  axis=-1)


def binary_loss(x_pred, x_target, epsilon=1e-7, gamma=0.5, b_range=False):
    """function.py:73-82.  x_pred are PROBABILITIES.  The clip constants are taken in the dtype
    of x_pred, as TensorFlow does: in float32 1-1e-7 is 0.99999988."""
    dt = x_pred.dtype
    b = float(b_range)
    n = int(np.prod(x_pred.shape[1:]))
    t = x_target.reshape(-1, n).astype(dt)
    q = x_pred.reshape(-1, n)
    yt = dt.type(-b) + dt.type(2.0 * b + 1.0) * t
    yp = np.clip(q, dt.type(epsilon), dt.type(1.0) - dt.type(epsilon))
    g = dt.type(gamma)
    return -np.sum(g * yt * np.log(yp) + (dt.type(1.0) - g) * (dt.type(1.0) - yt) * np.log(dt.type(1.0) - yp), axis=-1)


def voxel_precision_recall(x_target, x_pred, prob=0.5):
    """function.py:100-115 -> (TP, FP, FN) per sample; threshold is >= on the probability."""
    n = int(np.prod(x_target.shape[1:]))
    yt = x_target.reshape(-1, n).astype(np.float64)
    yp = (x_pred.reshape(-1, n) >= prob).astype(np.float64)
    return (yt * yp).sum(-1), ((1 - yt) * yp).sum(-1), (yt * (1 - yp)).sum(-1)


def pr_rc(tp, fp, fn):
    """nolbo.py:1443-1445 / 1500-1501."""
    return np.mean(tp / (tp + fp + 1e-10)), np.mean(tp / (tp + fn + 1e-10))


def iou(tp, fp, fn):
    """Not in the reference: IoU_b = TP/(TP+FP+FN), derived from its counts."""
    return tp / np.maximum(tp + fp + fn, 1.0)


# ----------------------------------------------------------------------------- module/nolbo.py
def split_mean_logvar(enc_out, L):
    """nolbo.py:1417-1420 / 1464-1468: mean = out[:, :L]; logvar = clip(out[:, L:2L], -10, 10)."""
    return enc_out[..., :L], np.clip(enc_out[..., L:2 * L], -10.0, 10.0)


def _nearest_category_acc(z, cats, onehot, mask=None):
    """nolbo.py:1489-1494 (mask=None) and :1505-1506 (masked distance)."""
    d = np.square(z[:, None, :] - cats[None, :, :])
    if mask is not None:
        d = mask[:, None, :] * d
    dist = d.sum(-1)
    idx = np.argmin(dist, axis=-1)
    return idx, np.mean((idx == np.argmax(onehot, axis=-1)).astype(np.float64))


def _shape_metrics(probs, target, gamma=0.6):
    """nolbo.py:1497-1501."""
    bce = binary_loss(probs, target, gamma=gamma, b_range=False)
    tp, fp, fn = voxel_precision_recall(target, probs)
    pr, rc = pr_rc(tp, fp, fn)
    return bce, tp, fp, fn, np.mean(bce), pr, rc


def vae_get_eval(config, enc_p, dec_p, inputs, category_vectors, eps, missing_prob=0.0,
                 mask=None, eps2=None, training=False, dtype=np.float64, variational=True, details=False):
    """nolboSingleObject_modelnet_category_VAE.getEval -- nolbo.py:1449-1528 (variational=True) and
    ..._AE.getEval -- nolbo.py:1260-1332 (variational=False), with the three random draws the
    reference makes internally (sampling eps :1470, the np.random mask :1475, the prior eps :1508)
    injected as arguments so the result is a function of its inputs."""
    x, y, onehot = inputs
    L = config['z_category_dim']
    cats = category_vectors.astype(dtype)
    enc_out = encoder3D_forward(config['encoder'], enc_p, x, training, dtype)
    if variational:
        mu, lv = split_mean_logvar(enc_out, L)
        z = sampling(mu, lv, eps)
    else:
        mu, lv, z = enc_out, None, enc_out
    if missing_prob > 0:
        m = mask.astype(dtype)
        z = z * m                                                        # :1477
        z = np.where(z == 0, cats.mean(axis=0)[None, :] * np.ones_like(z), z)   # :1481-1482
    else:
        m = np.ones_like(z)
    _, acc = _nearest_category_acc(z, cats, onehot)
    logits, probs = decoder3D_forward(config['decoder'], dec_p, z, training, dtype)
    bce, tp, fp, fn, loss_shape, pr, rc = _shape_metrics(probs, y.astype(dtype))
    det = {'enc_out': enc_out, 'mu': mu, 'logvar': lv, 'z': z, 'logits': logits, 'bce': bce,
           'tp': tp, 'fp': fp, 'fn': fn}
    if variational:
        det['kl'] = kl_loss(mu, lv, np.zeros_like(mu), np.zeros_like(lv))
    if missing_prob == 0.0:
        out = (probs, loss_shape, pr, rc, acc, 0, 0, 0, 0, 0)            # :1503
        return (out, det) if details else out
    idx, _ = _nearest_category_acc(z, cats, onehot, mask=m)             # :1505-1506
    z_prior = sampling(cats[idx], np.zeros_like(z), eps2)                # :1507-1509
    z_corr = np.where(m == 0, z_prior, z)                                # :1510
    _, acc_c = _nearest_category_acc(z_corr, cats, onehot)              # :1512-1518
    logits_c, probs_c = decoder3D_forward(config['decoder'], dec_p, z_corr, training, dtype)
    bce_c, tp_c, fp_c, fn_c, loss_c, pr_c, rc_c = _shape_metrics(probs_c, y.astype(dtype))
    det.update({'z_corr': z_corr, 'logits_c': logits_c, 'bce_c': bce_c, 'tp_c': tp_c, 'fp_c': fp_c,
                'fn_c': fn_c, 'argmin_masked': idx})
    out = (probs, loss_shape, pr, rc, acc, probs_c, loss_c, pr_c, rc_c, acc_c)
    return (out, det) if details else out


def vae_get_latent(config, enc_p, x, eps, dtype=np.float64, variational=True):
    """getLatent -- nolbo.py:1557-1566 (VAE: sampled z) / :1355-1358 (AE: raw encoder output)."""
    enc_out = encoder3D_forward(config['encoder'], enc_p, x, False, dtype)
    if not variational:
        return enc_out
    mu, lv = split_mean_logvar(enc_out, config['z_category_dim'])
    return sampling(mu, lv, eps)


def regulizer_loss(z_mean, z_logvar, dist_in_z_space, class_input=None):
    """reference src/module/function.py:40-71 -> [B]."""
    m, lv = np.asarray(z_mean, np.float64), np.asarray(z_logvar, np.float64)
    diff = (np.abs(m[:, None, :] - m[None, :, :]) / np.exp(0.5 * lv)[:, None, :]).sum(-1)      # scaled by row i's log-variance
    d = diff - dist_in_z_space
    d = np.where(d > 0, 0.0, d ** 2)
    if class_input is not None:
        c = np.asarray(class_input, np.float64)
        same = (np.abs(c[:, None, :] - c[None, :, :]).sum(-1) == 0).astype(np.float64)
        d = d * same
    return d.sum(-1)
