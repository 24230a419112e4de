"""Device-side engines for the two sub-models of the hot path.

`EncoderEngine` / `DecoderEngine` own the float32 master weights (Keras layouts, torch CUDA
tensors), the packed MFMA panels derived from them and the layer chain, and drive the C ABI
(voxvae.lib).  torch is used for device memory and the current HIP stream only; every
arithmetic step is a libvoxvae kernel.  Reference: src/net_core/autoencoder3D.py:72-139.
"""
import ctypes
import os

import numpy as np
import torch

from . import lib as L

BN_EPS = 1e-3  # Keras BatchNormalization default (autoencoder3D.py:31)

# Diagnostic only (profiles/microbench/fp8_schemes.py): callable(layer name, activation tensor) -> tensor applied to the input of
# every stride-2 layer, so that a quantisation scheme can be evaluated on the real kernels before a kernel is written for it.
LAYER_INPUT_HOOK = None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _tdtype(dt):
    return torch.bfloat16 if dt == L.VV_BF16 else torch.float32


def _require_gpu():
    if not torch.cuda.is_available():
        raise L.VoxVaeError('no HIP device visible: the voxel VAE path runs on MI355X only (no CPU fallback)')


class LayerTimer:
    """Optional per-layer HIP-event timing on the launch stream (bench.py's roofline leg).  `only` restricts the
    events to one layer name so the timed region carries two events per step, not two per layer."""

    def __init__(self, only=None):
        self.only = only
        self.events = {}

    def begin(self, name):
        if self.only is not None and name != self.only:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        return name, e0, e1

    def end(self, tok):
        if tok is not None:
            tok[2].record()
            self.events.setdefault(tok[0], []).append((tok[1], tok[2]))

    def summary_ms(self):
        """name -> (launches, mean ms); call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self.events.items()}


class _Workspace:
    """Grow-only scratch buffers shared by the layers of one engine (split-K slabs, loss partials), ONE PER HIP STREAM: the
    same engine can then run independent batches on several streams at once (the chunked host-array path of getEval,
    voxvae/hostio.py) without a second in-flight step overwriting the first one's slabs."""

    def __init__(self, device):
        self.device = device
        self.bufs = {}

    def get(self, nbytes):
        nbytes = max(int(nbytes), 16)
        key = torch.cuda.current_stream(self.device).cuda_stream
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = self.bufs[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return buf


def round_e4m3(x):
    """float tensor -> nearest OCP e4m3fn value (round half to even, saturating at 448), in float32 arithmetic; bit-for-bit what a
    cast to torch.float8_e4m3fn gives for |x| <= 448 (tests/test_host_logic.py), without depending on that dtype's kernels."""
    x = x.float()
    ax = x.abs().clamp(max=448.0)
    _, e = torch.frexp(ax)                    # ax = m 2^e with m in [0.5, 1)
    e = (e - 1).clamp(min=-6)                 # the binade's exponent; the subnormals share 2^-6
    step = torch.exp2((e - 3).float())
    q = (torch.round(ax / step) * step).clamp(max=448.0)
    return torch.where(x < 0, -q, q)


# The taps ONE output element sums: all 64 of a stride-2 / stride-1 convolution kernel [kd,kh,kw,Cin,Cout]; for the stride-2 transposed
# convolution [kd,kh,kw,Cout,Cin] the 8 taps of an output-parity class (tap parity = 1 - output parity per axis).
CONV_TAP_GROUPS = (tuple(range(64)),)
CONVT_TAP_GROUPS = tuple(tuple((kd * 4 + kh) * 4 + kw for kd in range(4) for kh in range(4) for kw in range(4)
                               if (kd % 2, kh % 2, kw % 2) == (pd, ph, pw)) for pd in range(2) for ph in range(2) for pw in range(2))


def quant_fp8(w, cout_axis, tap_groups=None):
    """Per-output-channel scaling ahead of the fp8 pack: returns (q, s) with q = the e4m3fn image of w / s as float32 values and
    s = max|w| / 256 per channel (e4m3fn holds +-448; its relative precision does not depend on the scale, its range and subnormal
    floor do).  The caller folds s into the per-channel scale vector the epilogue applies; the pack kernels convert q exactly.

    tap_groups (round 4): ERROR DIFFUSION over the taps of one (cin, cout) pair.  Rounded independently, the 64 (or 8) tap weights
    a pair contributes to one output carry ~sqrt(n)/sqrt(12) ulps of summed error, and that sum is multiplied by whatever part of
    the activation is common to the taps -- which, for the locally constant feature maps of occupancy grids, is most of it.  With
    the running rounding error of a group carried into the next tap, the errors of a group sum to <= half an ulp whatever n is
    (partial sums over a raster range of the group likewise, which is what a SAME-padding border sees); the price is ~sqrt(2) more
    error per single weight.  Measured at the trained operating points (profiles/microbench/fp8_schemes.py, 256 samples): the IoU
    cost of the WEIGHT rounding of policy 'wide' goes from 3.2e-4 (32^3) / 2.9e-4 (64^3) to < 5e-5 -- it disappears in the noise."""
    red = [d for d in range(w.dim()) if d != cout_axis]
    s = w.abs().amax(dim=red).clamp_min(1e-20) / 256.0
    shape = [1] * w.dim()
    shape[cout_axis] = -1
    ws = w / s.view(shape)
    if tap_groups is None or os.environ.get('VV_FP8_SHAPED', '1') == '0':
        return round_e4m3(ws).contiguous(), s.contiguous()
    if w.dim() != 5 or w.shape[0] * w.shape[1] * w.shape[2] != 64:
        raise ValueError('tap groups are defined for [4,4,4,a,b] kernels, got %s' % (tuple(w.shape),))
    flat = ws.reshape(64, w.shape[3], w.shape[4])
    q = torch.empty_like(flat)
    for g in tap_groups:
        carry = torch.zeros_like(flat[0])
        for t in g:
            v = flat[t] + carry
            q[t] = round_e4m3(v)
            carry = v - q[t]
    return q.reshape(w.shape).contiguous(), s.contiguous()


def fp8_layers_off():
    """Layers kept on bf16 operands in 'fp8' mode: VV_FP8_OFF=E5,D2 (names as in bench.py's layer table)."""
    return set(n for n in os.environ.get('VV_FP8_OFF', '').replace(' ', '').split(',') if n)


class _EngineBase:
    def __init__(self, structure, dtype, device):
        _require_gpu()
        L.load()
        self.structure = structure
        self.dt = L.DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
        # 'fp8': inference mode in which the MFMA layers whose Cin is a multiple of 128 run on e4m3fn operands (weights
        # quantised per output channel, the scale folded into the BatchNorm scale; activations stored as fp8 between
        # consecutive fp8 layers); every other layer, the latent algebra and the losses are the bf16 path.
        self.fp8 = self.dt == L.VV_FP8
        if self.fp8:
            self.dt = L.VV_BF16
            from . import fp8_policy
            self.fp8_policy = fp8_policy()        # 'wide': only the layers with a direct fp8 kernel; 'mid' / 'most' / 'all': voxvae.set_fp8_policy
        self.tdt = _tdtype(self.dt)
        self.device = torch.device(device)
        self.params = {}          # name -> float32 CUDA tensor, Keras layout (the trainable/master copy)
        self.packed = {}
        self.ws = _Workspace(self.device)
        self._dirty = True
        self._folded, self._want_fold = False, True
        self.act = L.ACT[structure['activation']]
        self.timer = None         # LayerTimer or None
        self.tag = ''

    def _fp8_off(self):
        """Layers kept on bf16 operands in 'fp8' mode: the VV_FP8_OFF override plus what the policy excludes.  'mid' and 'most' are policy
        'all' minus a set: 'mid' keeps fp8 on the two widest stride-2 layers of each side (E2, E3 / D3, D4 of the five-layer models),
        'most' only takes the encoder tail back (the layer whose error moves the whole latent)."""
        off = set(fp8_layers_off())
        pol, n = getattr(self, 'fp8_policy', 'wide'), len(self.filters)
        enc = isinstance(self, EncoderEngine)
        if pol in ('mid', 'most') and enc:
            off.add('E%d' % n)
        if pol == 'mid':
            if enc:
                off |= set('E%d' % (i + 1) for i in range(3, n - 1))
            else:
                off |= set('D%d' % (i + 1) for i in range(1, n - 3))
        return off

    def _call(self, layer, fn, *args):
        t = self.timer
        tok = t.begin(self.tag + layer) if t is not None else None
        L.call(fn, *args)
        if tok is not None:
            t.end(tok)

    # ---- weights
    def set_params(self, params):
        for k, v in params.items():
            t = torch.as_tensor(np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v, dtype=torch.float32)
            if k in self.params and tuple(self.params[k].shape) != tuple(t.shape):
                raise ValueError('%s: shape %s != %s' % (k, tuple(t.shape), tuple(self.params[k].shape)))
            self.params[k] = t.to(self.device).contiguous()
        self._dirty = True

    def get_params(self):
        return {k: v.detach().cpu().numpy() for k, v in self.params.items()}

    def _fold(self, prefix, channels, repeat=1, bias=None):
        if not self._want_fold:
            return None, None
        p = self.params
        scale = torch.empty(channels * repeat, dtype=torch.float32, device=self.device)
        shift = torch.empty_like(scale)
        L.call('vv_fold_bn', L.ptr(p[prefix + '/gamma']), L.ptr(p[prefix + '/beta']), L.ptr(p[prefix + '/moving_mean']),
               L.ptr(p[prefix + '/moving_variance']), L.ptr(bias), BN_EPS, L.ptr(scale), L.ptr(shift), channels, repeat,
               _stream())
        return scale, shift

    def _empty(self, *shape, dtype=None):
        return torch.empty(shape, dtype=dtype or self.tdt, device=self.device)

    def _quant_fp8(self, w, cout_axis):
        return quant_fp8(w, cout_axis, CONV_TAP_GROUPS if cout_axis == 4 else CONVT_TAP_GROUPS)

    def _as_fp8(self, h, label):
        """bf16 activation -> fp8 copy (the hand-over from a bf16-only layer into an fp8 stretch)."""
        o = torch.empty(h.shape, dtype=torch.uint8, device=self.device)
        self._call(label, 'vv_convert', L.ptr(h), L.ptr(o), h.numel(), L.VV_BF16, L.VV_FP8, _stream())
        return o

    def ensure_packed(self, fold=True):
        """Refresh the packed weight images after a weight change.  fold=False (the training step, which uses batch
        statistics) skips the folded moving-statistics scale/shift vectors; the next inference call packs them."""
        if self._dirty or (fold and not self._folded):
            self._want_fold = fold
            self._pack()
            self._dirty = False
            self._folded = fold


def _check_cubic_pow2(shape):
    d = int(shape[0])
    if len(shape) != 4 or shape[1] != d or shape[2] != d or d & (d - 1):
        raise ValueError('voxel grid must be cubic with a power-of-two side, got %s' % (shape,))
    return d


class EncoderEngine(_EngineBase):
    """encoder3D (autoencoder3D.py:72-102): [B,D,D,D,1] -> [B,E] float32."""

    def __init__(self, structure, dtype='bf16', device='cuda:0'):
        super().__init__(structure, dtype, device)
        s = structure
        self.D = _check_cubic_pow2(s['input_shape'])
        self.filters = [int(c) for c in s['filter_num_list']]
        n = len(self.filters)
        if s['input_shape'][-1] != 1:
            raise NotImplementedError('encoder input must have 1 channel (occupancy grid)')
        if any(int(k) != 4 for k in s['filter_size_list']) or [int(v) for v in s['strides_list']] != [2] * (n - 1) + [1]:
            raise NotImplementedError('encoder3D kernels cover filter size 4 with strides [2]*(n-1)+[1] '
                                      '(every config of the reference)')
        if s['final_pool'] not in ('average', 'max', 'None', None):
            raise NotImplementedError("final_pool=%r: 'average' (every reference config), 'max' or 'None' (autoencoder3D.py:90-95)" % s['final_pool'])
        self.pool_max = s['final_pool'] == 'max'
        self.pool_none = s['final_pool'] in ('None', None)      # no pooling: the model returns the last conv's [B,S,S,S,E] map
        self.final_sigmoid = s['final_activation'] == 'sigmoid'  # tf.nn.sigmoid on the output (autoencoder3D.py:97-99)
        if not self.final_sigmoid and s['final_activation'] not in (None, 'None', 'linear'):
            raise NotImplementedError('encoder final_activation %r' % s['final_activation'])
        if self.D >> (n - 1) < 1:
            raise ValueError('grid too small for %d stride-2 layers' % (n - 1))
        self.S = self.D >> (n - 1)      # side of the last feature map
        self.E = self.filters[-1]
        if (self.pool_max or self.pool_none) and (self.S ** 3 * self.E) * (self.S ** 3 * self.filters[-2]) > (1 << 28):
            # the max pool is not linear, so the last conv runs position by position as one dense panel [S^3 E][S^3 Cin]; that is 4 M
            # elements at the 32^3 geometry (S = 2) and 268 M at 64^3 (S = 4): no reference config asks for it there
            raise NotImplementedError("final_pool=%r with a %d^3 last feature map" % (s['final_pool'], self.S))

    def param_shapes(self):
        shp, cin = {}, 1
        for i, c in enumerate(self.filters):
            shp['conv%d/kernel' % i] = (4, 4, 4, cin, c)
            if i < len(self.filters) - 1:
                for nme in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
                    shp['bn%d/%s' % (i, nme)] = (c,)
            cin = c
        return shp

    def _pack(self):
        p, f, st = self.params, self.filters, _stream()
        self.packed = {'scale0': None}
        self.packed['scale0'], self.packed['shift0'] = self._fold('bn0', f[0])
        self.packed['w0'] = self._empty(f[0], 64)
        L.call('vv_pack_conv_k4', L.ptr(p['conv0/kernel']), L.ptr(self.packed['w0']), 1, f[0], self.dt, st)
        for i in range(1, len(f) - 1):
            # Cin 64 (the second layer) has an fp8 form too (tap-pair rows); VV_FP8_E2=0 keeps it on the bf16 direct kernel
            q = self.fp8 and (f[i - 1] % 128 == 0 or (f[i - 1] == 64 and os.environ.get('VV_FP8_E2', '1') != '0')) \
                and ('E%d' % (i + 1)) not in self._fp8_off()
            if q and self.fp8_policy == 'wide':
                q = bool(L.load().vv_conv3d_k4s2_direct_fp8_supported(self.D >> i, f[i - 1], f[i])) and os.environ.get('VV_FP8_E2', '1') != 'igemm'
            wk = p['conv%d/kernel' % i]
            if q:
                wk, qs = self._quant_fp8(wk, 4)
            small = (not q and not os.environ.get('VV_NO_SKIP')
                     and bool(L.load().vv_conv3d_k4s2_skip_supported(self.D >> i, f[i - 1], f[i], self.dt)
                              or L.load().vv_conv3d_k4s2_pos_supported(self.D >> i, f[i - 1], f[i], self.dt)))
            if small and not self._want_fold:
                w = None        # the training step packs the skip / position image of this layer per use (voxvae/train.py: _conv)
            else:
                w = self._empty(f[i], 64 * f[i - 1], dtype=torch.uint8 if q else None)
                L.call('vv_pack_conv_k4', L.ptr(wk), L.ptr(w), f[i - 1], f[i], L.VV_FP8 if q else self.dt, st)
            self.packed['w%d' % i] = w
            self.packed['scale%d' % i], self.packed['shift%d' % i] = self._fold('bn%d' % i, f[i])
            if q:
                self.packed['q%d' % i] = True
                if self.packed['scale%d' % i] is not None:
                    self.packed['scale%d' % i].mul_(qs)
            elif (self._want_fold and not os.environ.get('VV_NO_SKIP')
                  and (L.load().vv_conv3d_k4s2_skip_supported(self.D >> i, f[i - 1], f[i], self.dt)
                       or L.load().vv_conv3d_k4s2_pos_supported(self.D >> i, f[i - 1], f[i], self.dt))):
                # the 8^3 -> 4^3 layer: whole samples resident in LDS, padded taps skipped; the 4^3 -> 2^3 layer: position-major
                # split-K GEMM -- same [tap][Cin/64][Cout][64] panel (inference path; the training step keeps the implicit-GEMM
                # panel above, which also serves its data-gradient passes)
                ws = self._empty(64 * f[i - 1] * f[i])
                L.call('vv_pack_conv_k4_skip', L.ptr(p['conv%d/kernel' % i]), L.ptr(ws), f[i - 1], f[i], st)
                self.packed['ws%d' % i] = ws
        i = len(f) - 1
        if self.pool_max or self.pool_none:
            # tf.reduce_max over the positions (autoencoder3D.py:92-93) / no pooling: the conv output itself is needed -> full panel [S^3 E][S^3 Cin]
            w = self._empty(self.S ** 3 * f[i], self.S ** 3 * f[i - 1])
            L.call('vv_pack_conv_k4s1_full', L.ptr(p['conv%d/kernel' % i]), L.ptr(w), self.S, f[i - 1], f[i], self.dt, st)
            self.packed['w%d' % i] = w
            return
        q = self.fp8 and self.fp8_policy != 'wide' and (self.S ** 3 * f[i - 1]) % 128 == 0 and ('E%d' % (i + 1)) not in self._fp8_off()
        wk = p['conv%d/kernel' % i]
        if q:
            wk, qs = self._quant_fp8(wk, 4)
            self.packed['q%d' % i], self.packed['scale%d' % i] = True, qs
        w = self._empty(f[i], self.S ** 3 * f[i - 1], dtype=torch.uint8 if q else None)
        L.call('vv_pack_conv_k4s1_meanpool', L.ptr(wk), L.ptr(w), self.S, f[i - 1], f[i], L.VV_FP8 if q else self.dt, st)
        self.packed['w%d' % i] = w

    def forward(self, x, stop_before_tail=False, stop_before_pos=False):
        """x: float32 CUDA tensor [B,D,D,D,1] (contiguous) -> enc_out float32 [B,E].
        stop_before_tail: return the last stride-2 activation [B,S,S,S,C] instead (the fused latent tail consumes it).
        stop_before_pos (with stop_before_tail): when the last stride-2 layer is the position-major 4^3 -> 2^3 form, stop in FRONT of it and
        return its input [B,4,4,4,C]: latent_tail(..., pos_layer=True) runs that layer and the tail as one fused call (its split-K
        partial sums are summed by the tail's first kernel)."""
        self.ensure_packed()
        B, D, f, pk, st = x.shape[0], self.D, self.filters, self.packed, _stream()
        if tuple(x.shape[1:]) != (D, D, D, 1) or x.dtype != torch.float32 or not x.is_contiguous():
            raise ValueError('encoder input must be contiguous float32 [B,%d,%d,%d,1], got %s %s' % (D, D, D, tuple(x.shape), x.dtype))
        side = D // 2
        # element type of h: the engine's dtype, or fp8 inside an fp8 stretch (the first layer stores e4m3fn itself when the
        # second layer is fp8 and its plane-form kernel applies)
        hdt = L.VV_FP8 if (pk.get('q1', False) and D >= 32 and f[0] == 64 and self.dt == L.VV_BF16) else self.dt
        h = self._empty(B, side, side, side, f[0], dtype=torch.uint8 if hdt == L.VV_FP8 else None)
        self._call('E1', 'vv_conv3d_first_fwd_io', L.ptr(x), L.ptr(pk['w0']), L.ptr(pk['scale0']), L.ptr(pk['shift0']),
               L.ptr(h), B, D, f[0], self.act, self.dt, hdt, st)
        for i in range(1, len(f) - 1):
            q, nq = pk.get('q%d' % i, False), pk.get('q%d' % (i + 1), False)
            odt = L.VV_FP8 if nq else self.dt
            name = 'E%d' % (i + 1)
            if LAYER_INPUT_HOOK is not None:
                h = LAYER_INPUT_HOOK(name, h)
            if q:
                if hdt != L.VV_FP8:
                    h = self._as_fp8(h, name + 'c')
                o = self._empty(B, side // 2, side // 2, side // 2, f[i], dtype=torch.uint8 if nq else None)
                if (not os.environ.get('VV_NO_DIRECT') and os.environ.get('VV_FP8_E2', '1') != 'igemm' and odt != L.VV_F32
                        and L.load().vv_conv3d_k4s2_direct_fp8_supported(side, f[i - 1], f[i])):
                    # the widest encoder layer: fp8 twin of its direct kernel, same packed weights as the implicit GEMM
                    self._call(name, 'vv_conv3d_k4s2_direct_fp8_fwd', L.ptr(h), L.ptr(pk['w%d' % i]), L.ptr(pk['scale%d' % i]),
                               L.ptr(pk['shift%d' % i]), L.ptr(o), B, side, f[i - 1], f[i], self.act, odt, st)
                else:
                    ws = self.ws.get(L.load().vv_conv3d_k4s2_workspace_bytes(B, side, f[i - 1], f[i], L.VV_FP8))
                    self._call(name, 'vv_conv3d_k4s2_fwd_io', L.ptr(h), L.ptr(pk['w%d' % i]), L.ptr(pk['scale%d' % i]), L.ptr(pk['shift%d' % i]),
                               L.ptr(o), B, side, f[i - 1], f[i], self.act, L.VV_FP8, odt, L.ptr(ws), ws.numel(), st)
                hdt = odt
            elif ('ws%d' % i) in pk and not nq:
                if stop_before_pos and stop_before_tail and side == 4 and i == len(f) - 2 and hdt == self.dt and not pk.get('q%d' % (i + 1), False):
                    return h
                o = self._empty(B, side // 2, side // 2, side // 2, f[i])
                if side == 4:
                    ws = self.ws.get(L.load().vv_conv3d_k4s2_pos_workspace_bytes(B, f[i - 1], f[i]))
                    self._call(name, 'vv_conv3d_k4s2_pos_fwd', L.ptr(h), L.ptr(pk['ws%d' % i]), L.ptr(pk['scale%d' % i]),
                               L.ptr(pk['shift%d' % i]), L.ptr(o), B, side, f[i - 1], f[i], self.act, self.dt, L.ptr(ws), ws.numel(), st)
                else:
                    self._call(name, 'vv_conv3d_k4s2_skip_fwd', L.ptr(h), L.ptr(pk['ws%d' % i]), L.ptr(pk['scale%d' % i]),
                               L.ptr(pk['shift%d' % i]), L.ptr(o), B, side, f[i - 1], f[i], self.act, self.dt, st)
                hdt = odt
            elif not os.environ.get('VV_NO_DIRECT') and L.load().vv_conv3d_k4s2_direct_supported(side, f[i - 1], f[i], self.dt):
                o = self._empty(B, side // 2, side // 2, side // 2, f[i], dtype=torch.uint8 if nq else None)
                self._call(name, 'vv_conv3d_k4s2_direct_fwd_io', L.ptr(h), L.ptr(pk['w%d' % i]), L.ptr(pk['scale%d' % i]),
                           L.ptr(pk['shift%d' % i]), L.ptr(o), B, side, f[i - 1], f[i], self.act, self.dt, odt, st)
                hdt = odt
            else:
                o = self._empty(B, side // 2, side // 2, side // 2, f[i], dtype=torch.uint8 if nq else None)
                ws = self.ws.get(L.load().vv_conv3d_k4s2_workspace_bytes(B, side, f[i - 1], f[i], self.dt))
                self._call(name, 'vv_conv3d_k4s2_fwd_io', L.ptr(h), L.ptr(pk['w%d' % i]), L.ptr(pk['scale%d' % i]), L.ptr(pk['shift%d' % i]),
                           L.ptr(o), B, side, f[i - 1], f[i], self.act, self.dt, odt, L.ptr(ws), ws.numel(), st)
                hdt = odt
            h, side = o, side // 2
        i = len(f) - 1
        K = side ** 3 * f[i - 1]
        q = pk.get('q%d' % i, False)
        if self.pool_max or self.pool_none:
            if stop_before_tail:
                raise L.VoxVaeError("stop_before_tail: the fused latent tail folds the MEAN pool into its weights")
            if hdt != self.dt:
                raise L.VoxVaeError("final_pool='max' takes a %s activation" % self.tdt)
            P = side ** 3
            ws = self.ws.get(L.load().vv_dense_workspace_bytes(B, P * f[i], K, self.dt))
            full = self._empty(B, P, f[i], dtype=torch.float32)
            self._call('E%d' % (i + 1), 'vv_dense_fwd', L.ptr(h), L.ptr(pk['w%d' % i]), None, None, L.ptr(full), B, P * f[i], K, 0,
                       self.dt, L.VV_F32, L.ptr(ws), ws.numel(), st)
            if self.pool_none:
                out = full.view(B, side, side, side, f[i])
            else:
                out = self._empty(B, f[i], dtype=torch.float32)
                self._call('E%dp' % (i + 1), 'vv_max_over_positions', L.ptr(full), L.ptr(out), B, P, f[i], st)
            if self.final_sigmoid:
                self._call('E%ds' % (i + 1), 'vv_sigmoid_f32', L.ptr(out), L.ptr(out), out.numel(), st)
            return out
        if stop_before_tail:
            if q or hdt != self.dt:
                raise L.VoxVaeError('stop_before_tail: the fused latent tail takes a %s activation' % self.tdt)
            if self.final_sigmoid:
                raise L.VoxVaeError("stop_before_tail: the fused latent tail has no sigmoid on the encoder output")
            return h
        if q and hdt != L.VV_FP8:
            h = self._as_fp8(h, 'E%dc' % (i + 1))
        ddt = L.VV_FP8 if q else self.dt
        ws = self.ws.get(L.load().vv_dense_workspace_bytes(B, f[i], K, ddt))
        out = self._empty(B, f[i], dtype=torch.float32)
        self._call('E%d' % (i + 1), 'vv_dense_fwd', L.ptr(h), L.ptr(pk['w%d' % i]), L.ptr(pk.get('scale%d' % i)), None, L.ptr(out), B, f[i], K, 0,
                   ddt, L.VV_F32, L.ptr(ws), ws.numel(), st)
        if self.final_sigmoid:
            self._call('E%ds' % (i + 1), 'vv_sigmoid_f32', L.ptr(out), L.ptr(out), out.numel(), st)
        return out


class DecoderEngine(_EngineBase):
    """decoder3D (autoencoder3D.py:104-139): z [B,L] -> logits/probabilities [B,D,D,D,1] float32,
    with the BCE / TP / FP / FN reductions of function.py:73-115 fused into the last layer."""

    def __init__(self, structure, dtype='bf16', device='cuda:0'):
        super().__init__(structure, dtype, device)
        s = structure
        self.D = _check_cubic_pow2(s['output_shape'])
        self.filters = [int(c) for c in s['filter_num_list']]
        n = len(self.filters)
        if any(int(k) != 4 for k in s['filter_size_list']) or [int(v) for v in s['strides_list']] != [1] + [2] * (n - 1):
            raise NotImplementedError('decoder3D kernels cover filter size 4 with strides [1]+[2]*(n-1) '
                                      '(every config of the reference)')
        if self.filters[-1] != 1 or s['output_shape'][-1] != 1:
            raise NotImplementedError('decoder must end in 1 channel')
        self.L = int(s['input_dim'])
        self.S = self.D >> (n - 1)                      # autoencoder3D.py:115
        self.ch = max(self.filters[0] // 64, 8)         # autoencoder3D.py:116-118
        self.final_sigmoid = s['final_activation'] == 'sigmoid'
        if not self.final_sigmoid and s['final_activation'] not in (None, 'None', 'linear'):
            raise NotImplementedError('decoder final_activation %r' % s['final_activation'])

    def param_shapes(self):
        lin = self.S ** 3 * self.ch
        shp = {'dense/kernel': (self.L, lin), 'dense/bias': (lin,)}
        for nme in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
            shp['bn_dense/' + nme] = (lin,)
        cin = self.ch
        for i, c in enumerate(self.filters):
            shp['convT%d/kernel' % i] = (4, 4, 4, c, cin)
            if i < len(self.filters) - 1:
                for nme in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
                    shp['bnT%d/%s' % (i, nme)] = (c,)
            cin = c
        return shp

    def _pack(self):
        p, f, st, S3 = self.params, self.filters, _stream(), self.S ** 3
        lin = S3 * self.ch
        pk = self.packed = {}
        pk['wd'] = self._empty(lin, self.L)
        L.call('vv_pack_dense', L.ptr(p['dense/kernel']), L.ptr(pk['wd']), self.L, lin, self.dt, st)
        pk['scaled'], pk['shiftd'] = self._fold('bn_dense', lin, 1, p['dense/bias'])
        pk['w0'] = self._empty(S3 * f[0], lin)
        L.call('vv_pack_convT_k4s1_dense', L.ptr(p['convT0/kernel']), L.ptr(pk['w0']), self.S, self.ch, f[0], self.dt, st)
        pk['scale0'], pk['shift0'] = self._fold('bnT0', f[0], S3)
        # fp8 mode: the stride-2 layers with Cin % 128 == 0 run on fp8 operands -- the 128 -> 64 layer on the fp8 twin of its
        # direct kernel (VV_FP8_LAST=igemm: fp8 implicit GEMM, VV_FP8_LAST=0: bf16 direct kernel), the others on the implicit GEMM
        for i in range(1, len(f) - 1):
            side_i = self.S << (i - 1)
            direct = not os.environ.get('VV_NO_DIRECT') and bool(L.load().vv_convT3d_k4s2_direct_supported(side_i, f[i - 1], f[i], self.dt))
            q = self.fp8 and f[i - 1] % 128 == 0 and ('D%d' % (i + 1)) not in self._fp8_off()
            mode = os.environ.get('VV_FP8_LAST', 'direct')
            direct8 = q and direct and mode not in ('0', 'igemm') and bool(L.load().vv_convT3d_k4s2_direct_fp8_supported(side_i, f[i - 1], f[i]))
            if q and self.fp8_policy == 'wide' and not direct8:
                q = False
            if q and direct and mode == '0':
                q = False
            wk = p['convT%d/kernel' % i]
            if q:
                wk, qs = self._quant_fp8(wk, 3)
            if direct8:
                pk['wq8f%d' % i] = self._empty(64 * f[i - 1] * f[i], dtype=torch.uint8)
                L.call('vv_pack_convT_k4s2_frag_fp8', L.ptr(wk), L.ptr(pk['wq8f%d' % i]), f[i - 1], f[i], st)
            elif (not q and not self._want_fold and not os.environ.get('VV_NO_SKIP')
                  and (L.load().vv_convT3d_k4s2_skip_supported(side_i, f[i - 1], f[i], self.dt)
                       or L.load().vv_convT3d_k4s2_pos_supported(side_i, f[i - 1], f[i], self.dt))):
                pk['w%d' % i] = None    # training step: the skip / position image is packed per use (voxvae/train.py: _convT)
            elif (not q and not self._want_fold and direct and not os.environ.get('VV_NO_WHOLE') and not os.environ.get('VV_NO_DIRECT')
                  and L.load().vv_convT3d_k4s2_whole_supported(side_i, f[i - 1], f[i], self.dt)):
                pk['w%d' % i] = None    # training step: this layer runs on the whole-sample kernel (image 'ww' below)
            else:
                pk['w%d' % i] = self._empty(8, f[i], 8 * f[i - 1], dtype=torch.uint8 if q else None)
                L.call('vv_pack_convT_k4s2', L.ptr(wk), L.ptr(pk['w%d' % i]), f[i - 1], f[i], L.VV_FP8 if q else self.dt, st)
            pk['scale%d' % i], pk['shift%d' % i] = self._fold('bnT%d' % i, f[i])
            if q:
                pk['q%d' % i] = True
                if pk['scale%d' % i] is not None:
                    pk['scale%d' % i].mul_(qs)
            elif direct:
                whole = not os.environ.get('VV_NO_WHOLE') and bool(L.load().vv_convT3d_k4s2_whole_supported(side_i, f[i - 1], f[i], self.dt))
                if not (whole and not self._want_fold):       # (the training step runs the whole-sample kernel: no fragment image)
                    pk['wf%d' % i] = self._empty(64 * f[i - 1] * f[i])
                    L.call('vv_pack_convT_k4s2_frag', L.ptr(p['convT%d/kernel' % i]), L.ptr(pk['wf%d' % i]), f[i - 1], f[i], st)
                if whole:
                    # the 8^3 x 128 -> 16^3 x 64 layer of the 32^3 model: one whole sample resident in LDS per workgroup
                    pk['ww%d' % i] = self._empty(64 * f[i - 1] * f[i])
                    L.call('vv_pack_convT_k4s2_skip', L.ptr(p['convT%d/kernel' % i]), L.ptr(pk['ww%d' % i]), f[i - 1], f[i], st)
            elif (self._want_fold and not os.environ.get('VV_NO_SKIP')
                  and (L.load().vv_convT3d_k4s2_skip_supported(side_i, f[i - 1], f[i], self.dt)
                       or L.load().vv_convT3d_k4s2_pos_supported(side_i, f[i - 1], f[i], self.dt))):
                # the 4^3 -> 8^3 and 2^3 -> 4^3 layers (see the encoder's twins)
                pk['ws%d' % i] = self._empty(64 * f[i - 1] * f[i])
                L.call('vv_pack_convT_k4s2_skip', L.ptr(p['convT%d/kernel' % i]), L.ptr(pk['ws%d' % i]), f[i - 1], f[i], st)

    def forward(self, z_act, target=None, want_logits=False, gamma=0.6, epsilon=1e-7, want_metrics=False, h1=None):
        """z_act: [B,L] in the activation dtype.  target: float32 [B,D,D,D,1] or None.
        Returns (out, logits, stats): out = probabilities (final_activation 'sigmoid') or logits;
        stats float32 [B,4] = per-sample (bce, TP, FP, FN) against target (zeros if None).
        want_metrics: also return float32 [4] = (mean bce, precision, recall, IoU) -- nolbo.py:1498-1501 -- from the same
        reduction launch as `stats` (a fourth return value).
        h1: the output of the first (stride-1) decoder layer [B,S,S,S,C0] when the fused latent tail has produced it already
        (z_act is then not read)."""
        self.ensure_packed()
        f, pk, st, S, D = self.filters, self.packed, _stream(), self.S, self.D
        lin = S ** 3 * self.ch
        n0 = S ** 3 * f[0]
        if h1 is not None:
            B = h1.shape[0]
            if h1.dtype != self.tdt or h1.numel() != B * n0 or not h1.is_contiguous():
                raise ValueError('h1 must be contiguous %s [B,%d,%d,%d,%d]' % (self.tdt, S, S, S, f[0]))
            h, hdt = h1, self.dt
        else:
            B = z_act.shape[0]
            if z_act.dtype != self.tdt or tuple(z_act.shape) != (B, self.L) or not z_act.is_contiguous():
                raise ValueError('decoder input must be contiguous %s [B,%d]' % (self.tdt, self.L))
            ws = self.ws.get(L.load().vv_dense_workspace_bytes(B, lin, self.L, self.dt))
            t = self._empty(B, lin)
            self._call('D0', 'vv_dense_fwd', L.ptr(z_act), L.ptr(pk['wd']), L.ptr(pk['scaled']), L.ptr(pk['shiftd']), L.ptr(t), B, lin,
                   self.L, self.act, self.dt, self.dt, L.ptr(ws), ws.numel(), st)
            ws = self.ws.get(L.load().vv_dense_workspace_bytes(B, n0, lin, self.dt))
            hdt = L.VV_FP8 if pk.get('q1', False) else self.dt          # D1 hands fp8 to an fp8 D2
            h = self._empty(B, S, S, S, f[0], dtype=torch.uint8 if hdt == L.VV_FP8 else None)
            self._call('D1', 'vv_dense_fwd', L.ptr(t), L.ptr(pk['w0']), L.ptr(pk['scale0']), L.ptr(pk['shift0']), L.ptr(h), B, n0, lin,
                   self.act, self.dt, hdt, L.ptr(ws), ws.numel(), st)
        side = S
        for i in range(1, len(f) - 1):
            name = 'D%d' % (i + 1)
            q, nq = pk.get('q%d' % i, False), pk.get('q%d' % (i + 1), False)
            odt = L.VV_FP8 if nq else self.dt
            if LAYER_INPUT_HOOK is not None:
                h = LAYER_INPUT_HOOK(name, h)
            if ('ww%d' % i) in pk and hdt == self.dt:
                o = self._empty(B, 2 * side, 2 * side, 2 * side, f[i])
                self._call(name, 'vv_convT3d_k4s2_whole_fwd', L.ptr(h), L.ptr(pk['ww%d' % i]), L.ptr(pk['scale%d' % i]),
                           L.ptr(pk['shift%d' % i]), L.ptr(o), B, side, f[i - 1], f[i], self.act, self.dt, st)
                h, side, hdt = o, 2 * side, self.dt
                continue
            if ('wf%d' % i) in pk:
                o = self._empty(B, 2 * side, 2 * side, 2 * side, f[i])
                self._call(name, 'vv_convT3d_k4s2_direct_fwd', L.ptr(h), L.ptr(pk['wf%d' % i]), L.ptr(pk['scale%d' % i]),
                           L.ptr(pk['shift%d' % i]), L.ptr(o), B, side, f[i - 1], f[i], self.act, self.dt, st)
                h, side, hdt = o, 2 * side, self.dt
                continue
            if ('ws%d' % i) in pk and hdt == self.dt:      # (an fp8 next layer converts this bf16 output itself: cheaper than the implicit GEMM)
                o = self._empty(B, 2 * side, 2 * side, 2 * side, f[i])
                if side == 2:
                    ws = self.ws.get(L.load().vv_convT3d_k4s2_pos_workspace_bytes(B, f[i - 1], f[i]))
                    self._call(name, 'vv_convT3d_k4s2_pos_fwd', L.ptr(h), L.ptr(pk['ws%d' % i]), L.ptr(pk['scale%d' % i]),
                               L.ptr(pk['shift%d' % i]), L.ptr(o), B, side, f[i - 1], f[i], self.act, self.dt, L.ptr(ws), ws.numel(), st)
                else:
                    self._call(name, 'vv_convT3d_k4s2_skip_fwd', L.ptr(h), L.ptr(pk['ws%d' % i]), L.ptr(pk['scale%d' % i]),
                               L.ptr(pk['shift%d' % i]), L.ptr(o), B, side, f[i - 1], f[i], self.act, self.dt, st)
                h, side, hdt = o, 2 * side, self.dt
                continue
            if q and hdt != L.VV_FP8:
                h = self._as_fp8(h, name + 'c')
            if ('wq8f%d' % i) in pk:                          # fp8 direct kernel: e4m3fn in, bf16 out
                # VV_FP8_D5=1: hand e4m3fn to the fp8 form of the final layer (sweep form, large batches).  Off by default: it is
                # 5 % faster at 32^3 and not at all at 64^3, and takes the IoU delta at 64^3 from 6e-5 to 4e-4 (gate 1e-3)
                o8 = (i == len(f) - 2 and 2 * side >= 8 and B * ((2 * side) // 8) ** 2 >= 128 and os.environ.get('VV_FP8_D5', '0') == '1')
                o = self._empty(B, 2 * side, 2 * side, 2 * side, f[i], dtype=torch.uint8 if o8 else None)
                self._call(name, 'vv_convT3d_k4s2_direct_fp8_fwd', L.ptr(h), L.ptr(pk['wq8f%d' % i]), L.ptr(pk['scale%d' % i]),
                           L.ptr(pk['shift%d' % i]), L.ptr(o), B, side, f[i - 1], f[i], self.act, L.VV_FP8 if o8 else self.dt, st)
                h, side, hdt = o, 2 * side, (L.VV_FP8 if o8 else self.dt)
                continue
            idt = L.VV_FP8 if q else self.dt
            ws = self.ws.get(L.load().vv_convT3d_k4s2_workspace_bytes(B, side, f[i - 1], f[i], idt))
            o = self._empty(B, 2 * side, 2 * side, 2 * side, f[i], dtype=torch.uint8 if nq else None)
            self._call(name, 'vv_convT3d_k4s2_fwd_io', L.ptr(h), L.ptr(pk['w%d' % i]), L.ptr(pk['scale%d' % i]), L.ptr(pk['shift%d' % i]),
                   L.ptr(o), B, side, f[i - 1], f[i], self.act, idt, odt, L.ptr(ws), ws.numel(), st)
            h, side, hdt = o, 2 * side, odt
        if target is None:
            target = torch.zeros(B, D, D, D, 1, dtype=torch.float32, device=self.device)
        elif target.dtype != torch.float32 or target.numel() != B * D ** 3 or not target.is_contiguous():
            raise ValueError('target must be contiguous float32 [B,%d,%d,%d,1]' % (D, D, D))
        probs = self._empty(B, D, D, D, 1, dtype=torch.float32) if self.final_sigmoid else None
        logits = self._empty(B, D, D, D, 1, dtype=torch.float32) if (want_logits or not self.final_sigmoid) else None
        stats = self._empty(B, 4, dtype=torch.float32)
        ws = self.ws.get(L.load().vv_convT3d_final_bce_workspace_bytes(B, side))
        if want_metrics:
            metrics = self._empty(4, dtype=torch.float32)
            self._call('D%d' % len(f), 'vv_convT3d_final_bce_metrics_fwd', L.ptr(h), L.ptr(self.params['convT%d/kernel' % (len(f) - 1)]),
                       L.ptr(target), L.ptr(probs), L.ptr(logits), L.ptr(stats), L.ptr(metrics), B, side, f[-2], gamma, epsilon, hdt,
                       L.ptr(ws), ws.numel(), st)
            return (probs if self.final_sigmoid else logits), logits, stats, metrics
        self._call('D%d' % len(f), 'vv_convT3d_final_bce_fwd', L.ptr(h), L.ptr(self.params['convT%d/kernel' % (len(f) - 1)]), L.ptr(target),
               L.ptr(probs), L.ptr(logits), L.ptr(stats), B, side, f[-2], gamma, epsilon, hdt, L.ptr(ws), ws.numel(), st)
        return (probs if self.final_sigmoid else logits), logits, stats


def reparam_kl(enc_out, eps, latent, act_dtype, drop_mask=None, drop_scale=1.0, want_stats=False):
    """Fused slice|clip|sampling|dropout|KL (nolbo.py:1417-1431; function.py:35-38, 84-98)."""
    B = enc_out.shape[0]
    dev = enc_out.device
    z = torch.empty(B, latent, dtype=torch.float32, device=dev)
    z_act = z if act_dtype == L.VV_F32 else torch.empty(B, latent, dtype=torch.bfloat16, device=dev)
    kl = torch.empty(B, dtype=torch.float32, device=dev)
    mean = torch.empty_like(z) if want_stats else None
    logvar = torch.empty_like(z) if want_stats else None
    L.call('vv_reparam_kl_fwd', L.ptr(enc_out), L.ptr(eps), L.ptr(drop_mask), float(drop_scale), L.ptr(z),
           L.ptr(z_act) if act_dtype != L.VV_F32 else None, act_dtype, L.ptr(kl), L.ptr(mean), L.ptr(logvar), B, latent,
           _stream())
    return z, z_act, kl, mean, logvar


def latent_tail_supported(enc, dec, variational):
    """True when encoder tail -> reparam/KL -> Dense -> first decoder layer can run as the two fused launches of latent_tail.hip."""
    if getattr(enc, 'pool_max', False) or getattr(enc, 'pool_none', False) or getattr(enc, 'final_sigmoid', False):
        return False
    if enc.dt != L.VV_BF16 or enc.fp8 or dec.fp8 or dec.dt != L.VV_BF16 or os.environ.get('VV_NO_LATENT_TAIL'):
        return False
    Lz = dec.L
    K5 = enc.S ** 3 * enc.filters[-2]
    # Measured: at the 32^3 model (K5 = n1 = 4096) the two fused launches take 0.028 ms against 0.045 ms for the five calls;
    # at the 64^3 model (K5 = n1 = 32768: 128 K slices of float32 slabs to sum, 8x the seed columns) 0.197 ms against 0.06 ms.
    if K5 > 8192 or dec.S ** 3 * dec.filters[0] > 8192:
        return False
    return bool(L.load().vv_latent_tail_supported(K5, enc.E, Lz, dec.S ** 3 * dec.ch, dec.S ** 3 * dec.filters[0], int(variational), L.VV_BF16)) \
        and enc.act == dec.act


def pos_latent_tail_supported(enc, dec, variational, batch=1):
    """True when the last stride-2 encoder layer (4^3 -> 2^3, position-major form) and the latent tail can run as ONE fused call
    (vv_conv_pos_latent_tail_fwd: the layer's split-K partial sums are summed inside the tail; one launch, so the batch's layer
    input has to fit 32-bit buffer offsets)."""
    if not latent_tail_supported(enc, dec, variational) or os.environ.get('VV_NO_POS_TAIL') or enc.S != 2 or len(enc.filters) < 3:
        return False
    if batch * 64 * enc.filters[-3] * 2 > 0x7FFFFFFF:
        return False
    ne = len(enc.filters) - 1
    enc.ensure_packed()
    if ('ws%d' % (ne - 1)) not in enc.packed or enc.packed.get('q%d' % (ne - 1), False):
        return False
    return bool(L.load().vv_conv_pos_latent_tail_supported(enc.filters[-3], enc.filters[-2], enc.E, dec.L, dec.S ** 3 * dec.ch,
                                                           dec.S ** 3 * dec.filters[0], int(variational), L.VV_BF16))


def latent_tail(enc, dec, h, eps, variational, want_enc_out=False, pos_layer=False):
    """Fused latent tail (vv_latent_tail_fwd): h = EncoderEngine.forward(x, stop_before_tail=True).
    pos_layer: h = EncoderEngine.forward(x, stop_before_tail=True, stop_before_pos=True), the INPUT of the last stride-2 layer
    (pos_latent_tail_supported): that layer runs inside the call (vv_conv_pos_latent_tail_fwd).
    Returns (z float32 [B,L], z_act bf16, kl [B] or None, enc_out or None, h1 = the decoder's first-layer output)."""
    enc.ensure_packed()
    dec.ensure_packed()
    B, dev = h.shape[0], h.device
    Lz, E = dec.L, enc.E
    K5 = enc.S ** 3 * enc.filters[-2]
    lin, n1 = dec.S ** 3 * dec.ch, dec.S ** 3 * dec.filters[0]
    ne = len(enc.filters) - 1
    if pos_layer:
        if tuple(h.shape[1:]) != (4, 4, 4, enc.filters[-3]):
            raise L.VoxVaeError('latent_tail(pos_layer=True) takes the [B,4,4,4,%d] input of the last stride-2 layer, got %s' % (enc.filters[-3], tuple(h.shape)))
        z = torch.empty(B, Lz, dtype=torch.float32, device=dev)
        z_act = torch.empty(B, Lz, dtype=torch.bfloat16, device=dev)
        kl = torch.empty(B, dtype=torch.float32, device=dev) if variational else None
        enc_out = torch.empty(B, E, dtype=torch.float32, device=dev) if want_enc_out else None
        h1 = torch.empty(B, dec.S, dec.S, dec.S, dec.filters[0], dtype=torch.bfloat16, device=dev)
        c3, c4 = enc.filters[-3], enc.filters[-2]
        ws = enc.ws.get(L.load().vv_conv_pos_latent_tail_workspace_bytes(B, c3, c4, E))
        pe, pd = enc.packed, dec.packed
        i4 = ne - 1
        enc._call('E%dLT' % ne, 'vv_conv_pos_latent_tail_fwd', L.ptr(h), L.ptr(pe['ws%d' % i4]), L.ptr(pe['scale%d' % i4]), L.ptr(pe['shift%d' % i4]),
                  c3, c4, L.ptr(pe['w%d' % ne]), L.ptr(pe.get('scale%d' % ne)), L.ptr(eps), L.ptr(pd['wd']), L.ptr(pd['scaled']), L.ptr(pd['shiftd']),
                  L.ptr(pd['w0']), L.ptr(pd['scale0']), L.ptr(pd['shift0']), L.ptr(enc_out), L.ptr(z), L.ptr(z_act), L.ptr(kl), L.ptr(h1), B, E, Lz,
                  lin, n1, int(variational), dec.act, L.VV_BF16, L.ptr(ws), ws.numel(), _stream())
        return z, z_act, kl, enc_out, h1
    z = torch.empty(B, Lz, dtype=torch.float32, device=dev)
    z_act = torch.empty(B, Lz, dtype=torch.bfloat16, device=dev)
    kl = torch.empty(B, dtype=torch.float32, device=dev) if variational else None
    enc_out = torch.empty(B, E, dtype=torch.float32, device=dev) if want_enc_out else None
    h1 = torch.empty(B, dec.S, dec.S, dec.S, dec.filters[0], dtype=torch.bfloat16, device=dev)
    ws = enc.ws.get(L.load().vv_latent_tail_workspace_bytes(B, K5, E, n1))
    pe, pd = enc.packed, dec.packed
    enc._call('LT', 'vv_latent_tail_fwd', L.ptr(h), L.ptr(pe['w%d' % ne]), L.ptr(pe.get('scale%d' % ne)), L.ptr(eps), L.ptr(pd['wd']),
              L.ptr(pd['scaled']), L.ptr(pd['shiftd']), L.ptr(pd['w0']), L.ptr(pd['scale0']), L.ptr(pd['shift0']), L.ptr(enc_out), L.ptr(z),
              L.ptr(z_act), L.ptr(kl), L.ptr(h1), B, K5, E, Lz, lin, n1, int(variational), dec.act, L.VV_BF16, L.ptr(ws), ws.numel(), _stream())
    return z, z_act, kl, enc_out, h1


def shape_metrics(stats):
    """[B,4] (bce,TP,FP,FN) -> float32 [4] = (mean bce, precision, recall, IoU) -- nolbo.py:1498-1501."""
    out = torch.empty(4, dtype=torch.float32, device=stats.device)
    L.call('vv_shape_metrics', L.ptr(stats), L.ptr(out), stats.shape[0], _stream())
    return out
