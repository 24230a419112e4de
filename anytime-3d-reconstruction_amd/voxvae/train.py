"""Training step of the modelnet VAE / AE (reference src/module/nolbo.py:1411-1447 / 1230-1258) on the C ABI.

Forward with batch-statistics BatchNorm, backward (data gradients through the forward implicit-GEMM kernels,
weight gradients through the reduction-over-rows MFMA kernel), Keras Adam.  float32 only this round.

Data parallel (modelled on the reference's only multi-GPU path, src/module/AE3D.py:46-48, 86-104):
one process per GPU, each rank runs its own batch shard with its OWN BatchNorm statistics, the loss is scaled
by the GLOBAL batch, and the gradients are summed across ranks by one bucketed all-reduce (RCCL over xGMI with
backend 'nccl'; gloo in the CPU tests of the bucketing logic) before Adam runs on every replica.
"""
import ctypes
import os

import numpy as np

import torch

from . import engine as E
from . import lib as L

BN_EPS, BN_MOMENTUM = 1e-3, 0.99          # Keras BatchNormalization defaults
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-7   # tf.keras.optimizers.Adam defaults
ADAM_CHUNK = 16384                            # elements per record of the vv_adam_step_multi table


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class GradBuckets:
    """Flat gradient storage: every trainable tensor's gradient is a view into one of a few large float32
    buffers, so the cross-rank sum is a handful of large all-reduces (per-link-bound xGMI rings want few, big
    messages) issued in backward order.  Pure tensor bookkeeping: usable (and tested) on CPU with gloo.

    wire='f32' (default): one SUM all-reduce per bucket on the float32 buffer (106 MB per step at the 32^3 model).
    wire='bf16' (round 4): half the bytes, and float32 ACCUMULATION -- a bucket is cut into one shard per rank and reduced as
        all-to-all (every rank sends shard j of its bf16 copy to rank j) -> float32 sum of the `world` received shards in rank order
        (the same order on every rank: deterministic) -> all-gather of the reduced shards as bf16 -> widened into the float32 bucket.
        The values on the wire are bf16, the sum is not: one rounding of each contribution and one of the result, whatever the
        number of ranks (a bf16 ring all-reduce rounds after every hop).  It is also the direct reduce-scatter + all-gather that
        the fully connected xGMI topology wants: every rank talks to its 7 peers at once instead of pushing 2 x 7/8 of the bucket
        through one ring link (DESIGN.md section 5 has the bytes per link)."""

    def __init__(self, named_shapes, device, bucket_bytes=32 << 20, wire='f32', world_size=1):
        if wire not in ('f32', 'bf16'):
            raise ValueError(wire)
        self.wire, self.world = wire, max(int(world_size), 1)
        self.views, self.buckets = {}, []
        cur, cur_n = [], 0
        plan = []
        for name, shape in named_shapes:
            n = 1
            for s in shape:
                n *= int(s)
            n_pad = (n + 3) // 4 * 4              # keep every view 16-byte aligned
            if cur and (cur_n + n_pad) * 4 > bucket_bytes:
                plan.append((cur, cur_n))
                cur, cur_n = [], 0
            cur.append((name, shape, n, cur_n))
            cur_n += n_pad
        if cur:
            plan.append((cur, cur_n))
        self.bucket_of, self.members = {}, []
        gran = 8 * self.world                    # a bucket splits into `world` shards of whole 16-byte bf16 pieces
        for items, total in plan:
            total = (total + gran - 1) // gran * gran
            buf = torch.zeros(total, dtype=torch.float32, device=device)
            self.buckets.append(buf)
            self.members.append([name for name, _, _, _ in items])
            for name, shape, n, off in items:
                self.views[name] = buf[off:off + n].view(*shape)
                self.bucket_of[name] = len(self.buckets) - 1
        self._wire_bufs = None       # wire='bf16': per bucket (bf16 send copy, received shards [world, shard], reduced shard, gathered)
        self._comm_stream = None
        self.always_reduce = False   # True: issue the collectives in a one-rank group too (exercises the RCCL path on one GPU)
        self.producer_streams = ()   # CUDA streams that write gradients (Trainer: launch stream + weight-gradient stream): a bucket's
                                     # collective is ordered behind ALL of them, whichever stream its last member was enqueued on
        self.profile = None          # a list: finish() appends (event before, event after) -- the EXPOSED collective time
        self.begin_step()

    def _distributed(self, group):
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or self.always_reduce)

    def blocking_all_reduce_ms(self, group=None, repeats=5):
        """Calibration for the overlap report: every bucket all-reduced back to back with nothing else on the device, in ms
        per step's worth of buckets (the buckets are left multiplied by world^repeats: call it outside a training step)."""
        import torch.distributed as dist
        if not self._distributed(group):
            return 0.0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(repeats):
            for i, b in enumerate(self.buckets):
                if self.wire == 'bf16':
                    w = self._launch_bf16(i, group, dist)
                    if w:
                        w()
                else:
                    dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / repeats

    def all_reduce(self, group=None):
        """Sum every bucket across the ranks (blocking form: all launched together, then waited)."""
        self.begin_step()
        self.finish(group)

    # -- overlapped form: the backward pass reports gradients as their kernels are enqueued; a bucket's all-reduce is
    # launched the moment its last member is ready (buckets are laid out in backward order), so the ring transfers of
    # the decoder's gradients run under the encoder's backward kernels.  With backend 'nccl' (RCCL) the collective runs
    # on the process group's own stream after an event on the current stream; wait() re-joins it.
    def begin_step(self):
        self._pending = [set(m) for m in self.members]
        self._works = [None] * len(self.buckets)
        self.launch_order = []

    def ready(self, names, group=None):
        for name in names:
            b = self.bucket_of[name]
            self._pending[b].discard(name)
            if not self._pending[b] and self._works[b] is None:
                self._launch(b, group)

    def _launch(self, b, group):
        self.launch_order.append(b)
        if self._distributed(group):
            import torch.distributed as dist
            if self.buckets[b].is_cuda and self.producer_streams:
                cur = torch.cuda.current_stream(self.buckets[b].device)
                for ps in self.producer_streams:
                    if ps != cur:
                        cur.wait_stream(ps)
            if self.wire == 'bf16':
                self._works[b] = self._launch_bf16(b, group, dist)
            else:
                self._works[b] = dist.all_reduce(self.buckets[b], op=dist.ReduceOp.SUM, group=group, async_op=True)
        else:
            self._works[b] = False

    def wire_bytes_per_step(self):
        """Bytes one rank puts on the wire per step (both phases of a reduction counted), for the bench line."""
        n = sum(int(b.numel()) for b in self.buckets)
        w = self.world
        per_elem = 4 if self.wire == 'f32' else 2
        return int(2 * (w - 1) / max(w, 1) * n * per_elem)

    def _launch_bf16(self, b, group, dist):
        """Direct reduce-scatter + all-gather with bf16 on the wire and a float32 sum (class docstring).  On a GPU the three steps
        run on a side stream behind an event on the launch stream, so that the backward kernels enqueued after this call do not wait
        for them; finish() joins the side stream.  Returns a callable that finish() invokes (the Work-like object of this form)."""
        buf = self.buckets[b]
        world = dist.get_world_size(group)
        if buf.numel() % (8 * world):
            raise ValueError('bucket of %d elements does not split into %d shards (GradBuckets(world_size=...) must be the group size)' % (buf.numel(), world))
        shard = buf.numel() // world
        if self._wire_bufs is None:
            self._wire_bufs = {}
        if b not in self._wire_bufs:
            bf = torch.bfloat16
            self._wire_bufs[b] = (torch.empty(buf.numel(), dtype=bf, device=buf.device), torch.empty(world, shard, dtype=bf, device=buf.device),
                                  torch.empty(shard, dtype=bf, device=buf.device), torch.empty(buf.numel(), dtype=bf, device=buf.device))
        send, recv, red, full = self._wire_bufs[b]

        def chain():
            send.copy_(buf)                                              # one rounding of this rank's contribution
            dist.all_to_all_single(recv.view(-1), send, group=group)     # recv[r] = rank r's copy of MY shard
            red.copy_(recv.float().sum(dim=0))                           # float32 sum in rank order, one rounding of the result
            dist.all_gather_into_tensor(full, red, group=group)
            buf.copy_(full)

        if not buf.is_cuda:
            chain()
            return False
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=buf.device)
        cur = torch.cuda.current_stream(buf.device)
        self._comm_stream.wait_stream(cur)
        with torch.cuda.stream(self._comm_stream):
            chain()
        return lambda: torch.cuda.current_stream(buf.device).wait_stream(self._comm_stream)

    def finish(self, group=None):
        prof = self.profile is not None and self.buckets[0].is_cuda
        if prof:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()                   # the backward kernels are all enqueued: whatever the waits below add is exposed
        for b in range(len(self.buckets)):
            if self._works[b] is None:
                self._launch(b, group)
        for w in self._works:
            if w:
                w() if callable(w) else w.wait()
        if prof:
            e1.record()
            self.profile.append((e0, e1))


class _BN:
    """Per-layer BatchNorm training state (batch statistics + folded scale/shift) on device."""

    def __init__(self, c, device):
        self.c = c
        for n in ('mean', 'var', 'rstd', 'scale', 'shift'):
            setattr(self, n, torch.empty(c, dtype=torch.float32, device=device))


class Trainer:
    def __init__(self, enc, dec, variational=True, learning_rate=1e-4, world_size=1, group=None, grad_wire=None):
        """grad_wire: 'f32' (default) or 'bf16' -- what the gradient buckets put on the wire (GradBuckets); VOXVAE_GRAD_WIRE sets the default."""
        if enc is not None and enc.dt != dec.dt:
            raise ValueError('encoder and decoder engines must share one activation dtype')
        if dec.fp8 or (enc is not None and enc.fp8):
            raise ValueError("fit() runs in 'f32' or 'bf16'; 'fp8' is an inference mode (quantised weight images)")
        if enc is not None and (getattr(enc, 'pool_none', False) or getattr(enc, 'final_sigmoid', False)):
            raise NotImplementedError("fit() with final_pool='None' / an encoder final_activation: inference only (no reference config trains with them)")
        if enc is not None and getattr(enc, 'pool_max', False):
            raise NotImplementedError("fit() with final_pool='max': the training step folds the mean pool into the last conv's panel; "
                                      "no reference config trains with the max pool")
        # 'f32': everything on the exact-f32 MFMA path (parity mode).  'bf16': mixed precision -- activations, their
        # gradients and the MFMA operands in bf16, float32 master weights / Adam moments / BatchNorm statistics / losses,
        # weight gradients accumulated in float32 (f32 MFMA over widened operands).
        self.dt = dec.dt
        self.tdt = torch.bfloat16 if self.dt == L.VV_BF16 else torch.float32
        self.enc, self.dec, self.var, self.lr = enc, dec, variational, float(learning_rate)   # enc None: decoder-only (image -> 3D model)
        self.dev = dec.device
        self.world, self.group = int(world_size), group
        self.t = 0
        self.debug = None          # set to a dict to capture intermediate tensors of the next step (tests)
        self.overlap = True        # launch each gradient bucket's all-reduce as soon as its last member is enqueued
        # Weight gradients on a SECOND stream (round 4, opt-in: VOXVAE_WGRAD_STREAM=1): wgrad(layer i) needs only dL/d(conv out of i) and
        # the layer's input, and nothing downstream needs it before Adam, so it can fork off the chain while that continues with the next
        # layer's BatchNorm backward sweeps (HBM streams without LDS beside MFMA / LDS work).  Measured on MI355X it is SLOWER: 2.74 against
        # 2.65 ms per step, the weight-gradient kernels take 0.28 instead of 0.13 ms each beside the sweeps and nothing is gained back
        # (profiles/r04_train_wgrad_stream_ab.json) -- the same sign as round 3's side-stream weight packing.  Bit-identical results.
        self.wgrad_stream = None
        if str(self.dev).startswith('cuda') and os.environ.get('VOXVAE_WGRAD_STREAM', '0') == '1':
            self.wgrad_stream = torch.cuda.Stream(device=self.dev)
        self.ws = E._Workspace(self.dev)
        names = [] if enc is None else [('enc/' + k, v.shape) for k, v in enc.params.items() if not k.endswith(('moving_mean', 'moving_variance'))]
        names += [('dec/' + k, v.shape) for k, v in dec.params.items() if not k.endswith(('moving_mean', 'moving_variance'))]
        # backward order: decoder tail first ... encoder head last
        self.order = list(reversed(names))
        self.grads = GradBuckets(self.order, self.dev, wire=grad_wire or os.environ.get('VOXVAE_GRAD_WIRE', 'f32'), world_size=self.world)
        self._side_used = False
        self.m = {n: torch.zeros(s, dtype=torch.float32, device=self.dev) for n, s in names}
        self.v = {n: torch.zeros(s, dtype=torch.float32, device=self.dev) for n, s in names}

    @classmethod
    def forward_only(cls, enc=None, dec=None):
        """A Trainer without gradient / optimiser state: the training-mode forward of ONE sub-model (the builder-level
        `model(x, training=True)`, reference nolbo.py:1426 / AE3D.py:72-73)."""
        eng = dec if dec is not None else enc
        if eng.fp8:
            raise ValueError("training=True runs in 'f32' or 'bf16'; 'fp8' is an inference mode")
        t = cls.__new__(cls)
        t.enc, t.dec, t.dt = enc, dec, eng.dt
        t.tdt = torch.bfloat16 if t.dt == L.VV_BF16 else torch.float32
        t.dev, t.ws, t.debug, t.var = eng.device, E._Workspace(eng.device), None, False
        return t

    def _moved_statistics(self, *engines):
        """The moving statistics were updated in place by a training-mode forward: the folded inference scale / shift
        vectors of these engines are stale (the weight images are not)."""
        for e in engines:
            if e is not None:
                e._folded = False

    def encoder_training_mode(self, x):
        """encoder(x, training=True): batch-statistics BatchNorm, moving statistics move; -> enc_out float32 [B, E]."""
        self.enc.ensure_packed(fold=False)
        enc_out, _ = self._encoder_forward(x, x.shape[0])
        self._moved_statistics(self.enc)
        return enc_out

    # ------------------------------------------------------------------ helpers
    def _p(self, name):
        eng, key = (self.enc, name[4:]) if name.startswith('enc/') else (self.dec, name[4:])
        return eng.params[key]

    def _g(self, name):
        return self.grads.views[name]

    def _ready(self, *names):
        if self.overlap:
            self.grads.ready(names, self.group)

    def _empty(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    def _aempty(self, *shape):
        return torch.empty(shape, dtype=self.tdt, device=self.dev)        # an activation / activation gradient

    @staticmethod
    def _dt(t):
        return L.VV_BF16 if t.dtype == torch.bfloat16 else L.VV_F32

    def _cast(self, t):
        """float32 tensor -> the activation dtype (a copy only in bf16 mode)."""
        if self.dt == L.VV_F32:
            return t
        o = torch.empty(t.shape, dtype=self.tdt, device=self.dev)
        L.call('vv_convert', L.ptr(t), L.ptr(o), t.numel(), L.VV_F32, self.dt, _st())
        return o

    def _bn_fwd(self, c, rows, ch, eng, prefix, act):
        bn = _BN(ch, self.dev)
        p = eng.params
        ws = self.ws.get(L.load().vv_bn_workspace_bytes(rows, ch))
        L.call('vv_bn_train_stats', L.ptr(c), rows, ch, L.ptr(p[prefix + '/gamma']), L.ptr(p[prefix + '/beta']), BN_EPS, BN_MOMENTUM,
               L.ptr(bn.mean), L.ptr(bn.var), L.ptr(bn.rstd), L.ptr(bn.scale), L.ptr(bn.shift), L.ptr(p[prefix + '/moving_mean']),
               L.ptr(p[prefix + '/moving_variance']), self._dt(c), L.ptr(ws), ws.numel(), _st())
        h = torch.empty_like(c)
        L.call('vv_bn_act_fwd', L.ptr(c), L.ptr(bn.scale), L.ptr(bn.shift), L.ptr(h), rows, ch, act, self._dt(c), _st())
        return h, bn

    def _bn_fwd_from_partials(self, c, partial, nblocks, rows, ch, eng, prefix, act):
        """_bn_fwd for a layer whose producer left the per-block column sums (vv_convT3d_k4s2_whole_stats_fwd): finalise + apply."""
        bn = _BN(ch, self.dev)
        p = eng.params
        L.call('vv_bn_finalize_stats', L.ptr(partial), nblocks, rows, ch, L.ptr(p[prefix + '/gamma']), L.ptr(p[prefix + '/beta']), BN_EPS, BN_MOMENTUM,
               L.ptr(bn.mean), L.ptr(bn.var), L.ptr(bn.rstd), L.ptr(bn.scale), L.ptr(bn.shift), L.ptr(p[prefix + '/moving_mean']),
               L.ptr(p[prefix + '/moving_variance']), _st())
        h = torch.empty_like(c)
        L.call('vv_bn_act_fwd', L.ptr(c), L.ptr(bn.scale), L.ptr(bn.shift), L.ptr(h), rows, ch, act, self._dt(c), _st())
        return h, bn

    def _bn_bwd(self, c, dh, bn, rows, gname, bname, act):
        dc = torch.empty_like(c)
        ws = self.ws.get(L.load().vv_bn_workspace_bytes(rows, bn.c))
        L.call('vv_bn_act_bwd', L.ptr(c), L.ptr(dh), L.ptr(bn.scale), L.ptr(bn.shift), L.ptr(bn.mean), L.ptr(bn.rstd),
               L.ptr(self._g(gname)), L.ptr(self._g(bname)), L.ptr(dc), rows, bn.c, act, self._dt(c), L.ptr(ws), ws.numel(), _st())
        self._ready(gname, bname)
        return dc

    def _dense(self, x, panel, m, n, k, shift=None, f32_out=False):
        """y[m,n] = x[m,k] @ panel[n,k]^T (+ shift); x and panel in the activation dtype, y in it too unless f32_out."""
        y = self._empty(m, n) if f32_out else self._aempty(m, n)
        ws = self.ws.get(L.load().vv_dense_workspace_bytes(m, n, k, self.dt))
        L.call('vv_dense_fwd', L.ptr(x), L.ptr(panel), None, L.ptr(shift), L.ptr(y), m, n, k, 0, self.dt, self._dt(y), L.ptr(ws),
               ws.numel(), _st())
        return y

    def _transposed_panel(self, panel_f32, rows, cols):
        """[rows][cols] float32 panel -> its transpose [cols][rows] in the activation dtype (vv_pack_dense reads a Keras
        [In][Out] array and writes [Out][In])."""
        o = self._aempty(cols, rows)
        L.call('vv_pack_dense', L.ptr(panel_f32), L.ptr(o), rows, cols, self.dt, _st())
        return o

    def _wgrad_dense(self, a, g, out, rows, m, n):
        ws = self.ws.get(L.load().vv_wgrad_workspace_bytes(rows, m, n))
        L.call('vv_wgrad_dense', L.ptr(a), L.ptr(g), L.ptr(out), rows, m, n, m, self._dt(a), self._dt(g), L.ptr(ws), ws.numel(), _st())

    def _wgrad_conv(self, src, g, out, batch, side, cin, cout, ready=None):
        """Weight gradient of a stride-2 layer into `out` (a view of a gradient bucket).  ready: the gradient's name -- reported to
        the buckets from the stream the kernel was enqueued on (a bucket's all-reduce is ordered behind that stream)."""
        o = side // 2
        ws_stream = getattr(self, 'wgrad_stream', None)
        if ws_stream is None:
            self._wgrad_conv_launch(src, g, out, batch, side, cin, cout, o)
            if ready:
                self._ready(ready)
            return
        main = torch.cuda.current_stream(self.dev)
        ws_stream.wait_stream(main)
        for t in (src, g):
            t.record_stream(ws_stream)               # allocated on the launch stream, read on this one
        with torch.cuda.stream(ws_stream):
            self._wgrad_conv_launch(src, g, out, batch, side, cin, cout, o)
            if ready:
                self._ready(ready)
        self._side_used = True

    def _wgrad_conv_launch(self, src, g, out, batch, side, cin, cout, o):
        ws = self.ws.get(L.load().vv_wgrad_workspace_bytes(batch * o ** 3, 64 * cin, cout))
        tm = getattr(self, 'timer', None)            # bench.py's roofline leg: HIP events around the launch, on its stream
        tok = tm.begin('wgrad:%d:%d:%d' % (side, cin, cout)) if tm is not None else None
        L.call('vv_wgrad_conv_k4s2', L.ptr(src), L.ptr(g), L.ptr(out), batch, side, cin, cout, self._dt(src), self._dt(g), L.ptr(ws),
               ws.numel(), _st())
        if tm is not None:
            tm.end(tok)

    def _conv(self, x, w_keras, B, side, cin, cout, packed=None):
        """Conv3D k4 s2 of x [B,side^3,cin] with a Keras kernel array read as [4,4,4,cin,cout]: the forward layers (packed =
        the engine's image of the same weights) and the data gradients of the transposed layers (packed here)."""
        dt, st = self.dt, _st()
        wp = packed
        y = self._aempty(B, side // 2, side // 2, side // 2, cout)
        lib = L.load()
        if not os.environ.get('VV_NO_SKIP') and (lib.vv_conv3d_k4s2_skip_supported(side, cin, cout, dt) or lib.vv_conv3d_k4s2_pos_supported(side, cin, cout, dt)):
            # 8^3 -> 4^3 / 4^3 -> 2^3 (bf16): the whole-samples-in-LDS and position-major kernels of the evaluation path, raw output
            # (no scale / shift / activation: BatchNorm follows with batch statistics); their weight image is packed per use
            wsk = getattr(self, '_prepacked', {}).get((0, w_keras.data_ptr(), cin, cout))
            if wsk is None:
                wsk = self._aempty(64 * cin * cout)
                L.call('vv_pack_conv_k4_skip', L.ptr(w_keras), L.ptr(wsk), cin, cout, st)
            if side == 4:
                ws = self.ws.get(lib.vv_conv3d_k4s2_pos_workspace_bytes(B, cin, cout))
                L.call('vv_conv3d_k4s2_pos_fwd', L.ptr(x), L.ptr(wsk), None, None, L.ptr(y), B, side, cin, cout, 0, dt, L.ptr(ws), ws.numel(), st)
            else:
                L.call('vv_conv3d_k4s2_skip_fwd', L.ptr(x), L.ptr(wsk), None, None, L.ptr(y), B, side, cin, cout, 0, dt, st)
            return y
        if wp is None:
            wp = self._aempty(cout, 64 * cin)
            L.call('vv_pack_conv_k4', L.ptr(w_keras), L.ptr(wp), cin, cout, dt, st)
        if L.load().vv_conv3d_k4s2_direct_supported(side, cin, cout, dt):
            L.call('vv_conv3d_k4s2_direct_fwd', L.ptr(x), L.ptr(wp), None, None, L.ptr(y), B, side, cin, cout, 0, dt, st)
        else:
            ws = self.ws.get(L.load().vv_conv3d_k4s2_workspace_bytes(B, side, cin, cout, dt))
            L.call('vv_conv3d_k4s2_fwd', L.ptr(x), L.ptr(wp), None, None, L.ptr(y), B, side, cin, cout, 0, dt, L.ptr(ws), ws.numel(), st)
        return y

    def _convT(self, x, w_keras, B, side, cin, cout, packed=None, packed_frag=None, packed_whole=None):
        """Conv3DTranspose k4 s2 of x [B,side^3,cin] with a Keras kernel array read as [4,4,4,cout,cin]: the forward
        transposed layers (packed / packed_frag = the engine's images) and the data gradients of the strided convolutions."""
        dt, st = self.dt, _st()
        y = self._aempty(B, 2 * side, 2 * side, 2 * side, cout)
        lib = L.load()
        if not os.environ.get('VV_NO_SKIP') and (lib.vv_convT3d_k4s2_skip_supported(side, cin, cout, dt) or lib.vv_convT3d_k4s2_pos_supported(side, cin, cout, dt)):
            wsk = getattr(self, '_prepacked', {}).get((1, w_keras.data_ptr(), cin, cout))     # 4^3 -> 8^3 / 2^3 -> 4^3 (bf16): see _conv
            if wsk is None:
                wsk = self._aempty(64 * cin * cout)
                L.call('vv_pack_convT_k4s2_skip', L.ptr(w_keras), L.ptr(wsk), cin, cout, st)
            if side == 2:
                ws = self.ws.get(lib.vv_convT3d_k4s2_pos_workspace_bytes(B, cin, cout))
                L.call('vv_convT3d_k4s2_pos_fwd', L.ptr(x), L.ptr(wsk), None, None, L.ptr(y), B, side, cin, cout, 0, dt, L.ptr(ws), ws.numel(), st)
            else:
                L.call('vv_convT3d_k4s2_skip_fwd', L.ptr(x), L.ptr(wsk), None, None, L.ptr(y), B, side, cin, cout, 0, dt, st)
            return y
        if (L.load().vv_convT3d_k4s2_whole_supported(side, cin, cout, dt) and not os.environ.get('VV_NO_DIRECT')
                and not os.environ.get('VV_NO_WHOLE')):
            # 8^3 x 128 -> 16^3 x 64 (the widest decoder layer forward, and the data gradient of the widest encoder layer):
            # whole-sample kernel; its weight image is packed here (the weights change every step)
            wk = packed_whole                        # the forward layer: the engine's image; a data gradient: the step's prepacked one
            if wk is None:
                wk = getattr(self, '_prepacked', {}).get((1, w_keras.data_ptr(), cin, cout))
            if wk is None:
                wk = self._aempty(64 * cin * cout)
                L.call('vv_pack_convT_k4s2_skip', L.ptr(w_keras), L.ptr(wk), cin, cout, st)
            L.call('vv_convT3d_k4s2_whole_fwd', L.ptr(x), L.ptr(wk), None, None, L.ptr(y), B, side, cin, cout, 0, dt, st)
        elif L.load().vv_convT3d_k4s2_direct_supported(side, cin, cout, dt) and not os.environ.get('VV_NO_DIRECT'):
            wf = packed_frag
            if wf is None:
                wf = self._aempty(64 * cin * cout)
                L.call('vv_pack_convT_k4s2_frag', L.ptr(w_keras), L.ptr(wf), cin, cout, st)
            L.call('vv_convT3d_k4s2_direct_fwd', L.ptr(x), L.ptr(wf), None, None, L.ptr(y), B, side, cin, cout, 0, dt, st)
        else:
            wp = packed
            if wp is None:
                wp = self._aempty(8, cout, 8 * cin)
                L.call('vv_pack_convT_k4s2', L.ptr(w_keras), L.ptr(wp), cin, cout, dt, st)
            ws = self.ws.get(L.load().vv_convT3d_k4s2_workspace_bytes(B, side, cin, cout, dt))
            L.call('vv_convT3d_k4s2_fwd', L.ptr(x), L.ptr(wp), None, None, L.ptr(y), B, side, cin, cout, 0, dt, L.ptr(ws), ws.numel(), st)
        return y

    # ------------------------------------------------------------------ one step
    def step(self, x, y, eps=None, drop_mask=None, drop_scale=1.0):
        """x, y: float32 CUDA [B,D,D,D,1].  Returns device tensors (loss_kl or None, stats [B,4], metrics [4])."""
        self.enc.ensure_packed(fold=False)
        self.dec.ensure_packed(fold=False)      # (packing the decoder on a side stream under the encoder's forward was measured: +0.11 ms per step)
        self.grads.begin_step()
        self._set_producer_streams()
        self._prepack()
        B = x.shape[0]
        inv_gb = 1.0 / float(B * self.world)      # loss scaled by the GLOBAL batch (AE3D.py:46-48)
        enc_out, est = self._encoder_forward(x, B)
        kl, stats, metrics, de = self._latent_decoder(enc_out, y, eps, drop_mask, drop_scale, B, inv_gb)
        self._encoder_backward(x, de, est, B)
        self._apply()
        return kl, stats, metrics

    def _prepack(self):
        """Every skip / position / whole-sample weight image the step will ask for (forward layers and data gradients: nine at the 32^3
        model), packed by ONE vv_pack_skip_images call = two launches, instead of nine 5-8 us launches spread over the step.  _conv /
        _convT look the images up by (kind, weight pointer, cin, cout) and pack per use when they find none (other entry points).  The
        images are dropped again when Adam has moved the weights (_apply)."""
        self._prepacked = {}
        if self.dt != L.VV_BF16 or os.environ.get('VV_NO_SKIP') or os.environ.get('VV_NO_PREPACK'):
            return
        lib, dt = L.load(), self.dt
        jobs = []                                                     # (kind, weight tensor, cin, cout)

        def conv_ok(side, cin, cout):
            return side >= 2 and (lib.vv_conv3d_k4s2_skip_supported(side, cin, cout, dt) or lib.vv_conv3d_k4s2_pos_supported(side, cin, cout, dt))

        def convT_ok(side, cin, cout, whole):
            if side >= 1 and (lib.vv_convT3d_k4s2_skip_supported(side, cin, cout, dt) or lib.vv_convT3d_k4s2_pos_supported(side, cin, cout, dt)):
                return True
            return bool(whole and lib.vv_convT3d_k4s2_whole_supported(side, cin, cout, dt) and not os.environ.get('VV_NO_DIRECT') and not os.environ.get('VV_NO_WHOLE'))

        if self.enc is not None:
            fe, side = self.enc.filters, self.enc.D // 2
            for i in range(1, len(fe) - 1):
                w = self.enc.params['conv%d/kernel' % i]
                if conv_ok(side, fe[i - 1], fe[i]):
                    jobs.append((0, w, fe[i - 1], fe[i]))             # the forward layer
                if convT_ok(side // 2, fe[i], fe[i - 1], True):
                    jobs.append((1, w, fe[i], fe[i - 1]))             # its data gradient (a transposed convolution with the same array)
                side //= 2
        fd, side = self.dec.filters, self.dec.S
        for i in range(1, len(fd) - 1):
            w = self.dec.params['convT%d/kernel' % i]
            if convT_ok(side, fd[i - 1], fd[i], False):
                jobs.append((1, w, fd[i - 1], fd[i]))                 # the forward layer (the whole-sample layer uses the engine's image)
            if conv_ok(2 * side, fd[i], fd[i - 1]):
                jobs.append((0, w, fd[i], fd[i - 1]))                 # its data gradient
            side *= 2
        if not jobs:
            return
        n = len(jobs)
        outs = [self._aempty(64 * cin * cout) for _, _, cin, cout in jobs]
        kinds = (ctypes.c_int * n)(*[k for k, _, _, _ in jobs])
        wp = (ctypes.c_void_p * n)(*[w.data_ptr() for _, w, _, _ in jobs])
        op = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
        ci = (ctypes.c_int * n)(*[c for _, _, c, _ in jobs])
        co = (ctypes.c_int * n)(*[c for _, _, _, c in jobs])
        L.call('vv_pack_skip_images', kinds, wp, op, ci, co, n, _st())
        for (k, w, cin, cout), o in zip(jobs, outs):
            self._prepacked[(k, w.data_ptr(), cin, cout)] = o

    def _join_wgrad(self):
        """The launch stream waits for the weight-gradient stream: every weight gradient is in its bucket."""
        if getattr(self, 'wgrad_stream', None) is not None and self._side_used:
            torch.cuda.current_stream(self.dev).wait_stream(self.wgrad_stream)
            self._side_used = False

    def _set_producer_streams(self):
        if getattr(self, 'wgrad_stream', None) is not None:
            self.grads.producer_streams = (torch.cuda.current_stream(self.dev), self.wgrad_stream)

    def step_from_latent(self, enc_out, y, eps=None, drop_mask=None, drop_scale=1.0, l2=0.0):
        """Decoder-only step for the image -> 3D model (nolbo.py:786-833): enc_out [B, 2L] (mean | logVar) comes from a 2D
        encoder owned by the caller.  Trains the decoder and returns (loss_kl, stats, metrics, d total / d enc_out) so the
        caller can continue the backward pass through its own encoder.  l2: coefficient of the decoder's kernel / bias
        regularisers when the caller's loss includes them."""
        self.dec.ensure_packed(fold=False)
        self.grads.begin_step()
        self._set_producer_streams()
        self._prepack()
        overlap, self.overlap = self.overlap, self.overlap and l2 == 0      # the l2 terms are added before the cross-rank sum
        B = enc_out.shape[0]
        inv_gb = 1.0 / float(B * self.world)
        kl, stats, metrics, de = self._latent_decoder(enc_out, y, eps, drop_mask, drop_scale, B, inv_gb)
        self.overlap = overlap
        if l2 > 0:      # + sum(decoder.losses) in the total loss (nolbo.py:819-823): d/dw of l2 * sum(w^2)
            self._join_wgrad()
            for name, _ in self.order:
                if name.endswith('/kernel') or name == 'dec/dense/bias':
                    self._g(name).add_(self._p(name), alpha=2.0 * l2 / self.world)
        self._apply()
        return kl, stats, metrics, de

    def step_custom_latent(self, x, y, latent_fn):
        """Full step with the latent algebra supplied by the caller as torch autograd code (the class-conditional prior
        model, nolbo.py:1620-1676: KL to a learned prior, prior / posterior mixing, pairwise regulariser).
        latent_fn(enc_out) -> (z_input [B,L], extra_loss scalar, aux): enc_out is a leaf that requires grad; the decoder
        runs on z_input.detach(), and d shape-loss / d z_input plus extra_loss are back-propagated through latent_fn's
        graph to enc_out (continuing into the HIP encoder backward) and to whatever parameters latent_fn touched."""
        if self.var:
            raise ValueError('step_custom_latent expects a Trainer built with variational=False (the decoder input is z_input)')
        self.enc.ensure_packed(fold=False)
        self.dec.ensure_packed(fold=False)
        self.grads.begin_step()
        self._set_producer_streams()
        self._prepack()
        B = x.shape[0]
        inv_gb = 1.0 / float(B * self.world)
        enc_out, est = self._encoder_forward(x, B)
        leaf = enc_out.detach().requires_grad_(True)
        with torch.enable_grad():
            z_input, extra, aux = latent_fn(leaf)
        _, stats, metrics, dz = self._latent_decoder(z_input.detach().contiguous(), y, None, None, 1.0, B, inv_gb)
        torch.autograd.backward([z_input, extra], [dz, torch.ones_like(extra)])
        self._encoder_backward(x, leaf.grad.contiguous(), est, B)
        self._apply()
        return stats, metrics, aux

    def _encoder_forward(self, x, B):
        if getattr(self.enc, 'pool_max', False) or getattr(self.enc, 'pool_none', False) or getattr(self.enc, 'final_sigmoid', False):
            raise NotImplementedError("training-mode forward with final_pool='max' / 'None' or an encoder final_activation")
        enc, st, dt = self.enc, _st(), self.dt
        D, fe, act = enc.D, enc.filters, enc.act
        # ---------------- encoder forward (raw conv -> batch stats -> BN + act)
        ec, eh, ebn = [], [], []
        side = D // 2
        c = self._aempty(B, side, side, side, fe[0])
        L.call('vv_conv3d_first_fwd', L.ptr(x), L.ptr(enc.packed['w0']), None, None, L.ptr(c), B, D, fe[0], 0, dt, st)
        h, bn = self._bn_fwd(c, B * side ** 3, fe[0], enc, 'bn0', act)
        ec.append(c); eh.append(h); ebn.append(bn)
        for i in range(1, len(fe) - 1):
            c = self._conv(eh[-1], enc.params['conv%d/kernel' % i], B, side, fe[i - 1], fe[i], packed=enc.packed['w%d' % i])
            side //= 2
            h, bn = self._bn_fwd(c, B * side ** 3, fe[i], enc, 'bn%d' % i, act)
            ec.append(c); eh.append(h); ebn.append(bn)
        ne = len(fe) - 1
        K5 = side ** 3 * fe[ne - 1]
        enc_out = self._dense(eh[-1], enc.packed['w%d' % ne], B, fe[ne], K5, f32_out=True)
        return enc_out, (ec, eh, ebn, side, K5)

    def _latent(self, enc_out, eps, drop_mask, drop_scale, B):
        """slice | clip | sampling | dropout | KL (float32 in both modes; z_act is the decoder's operand)."""
        Lz = self.dec.L
        if self.var:
            if eps is None:
                eps = torch.randn(B, Lz, dtype=torch.float32, device=self.dev)
            z, z_act, kl, _, _ = E.reparam_kl(enc_out, eps, Lz, self.dt, drop_mask, drop_scale)
        else:
            z, kl = enc_out, None
            z_act = self._cast(z)
            if drop_mask is not None:
                raise NotImplementedError('latent dropout for the AE class')
        return z, z_act, kl, eps

    def forward_training_mode(self, x, y, eps=None, z_fn=None):
        """The model called with training=True and NO optimisation step (getEval(training=True), reference nolbo.py:1449,
        1463, 1496): BatchNorm normalises with the batch statistics and updates its moving statistics, nothing else changes.
        z_fn(z float32 [B,L]) -> z may edit the latent before the decoder (missing-latent masking).
        Returns (z, kl or None, probs, stats [B,4], metrics [4])."""
        self.enc.ensure_packed(fold=False)
        self.dec.ensure_packed(fold=False)
        B = x.shape[0]
        enc_out, _ = self._encoder_forward(x, B)
        z, z_act, kl, _ = self._latent(enc_out, eps, None, 1.0, B)
        if z_fn is not None:
            z = z_fn(z)
            z_act = self._cast(z)
        fw = self._decoder_forward(z_act, y, B)
        self._moved_statistics(self.enc, self.dec)
        return z, kl, fw['probs'], fw['stats'], fw['metrics']

    def decoder_training_mode(self, z, y):
        """Decoder half of forward_training_mode for an edited latent (the corrected pass of getEval)."""
        self.dec.ensure_packed(fold=False)
        fw = self._decoder_forward(self._cast(z), y, z.shape[0])
        self._moved_statistics(self.dec)
        return fw['probs'], fw['stats'], fw['metrics']

    def _latent_decoder(self, enc_out, y, eps, drop_mask, drop_scale, B, inv_gb):
        dec, dev, st, dt = self.dec, self.dev, _st(), self.dt
        D, fd, act = dec.D, dec.filters, dec.act
        Lz = dec.L
        z, z_act, kl, eps = self._latent(enc_out, eps, drop_mask, drop_scale, B)
        fw = self._decoder_forward(z_act, y, B)
        S, ch = dec.S, dec.ch
        lin, n1 = S ** 3 * ch, S ** 3 * fd[0]
        c_d0, t0, bn_d0, c_d1, bn_d1 = fw['c_d0'], fw['t0'], fw['bn_d0'], fw['c_d1'], fw['bn_d1']
        dc_, dh_, dbn, probs, stats, metrics, side = fw['dc_'], fw['dh_'], fw['dbn'], fw['probs'], fw['stats'], fw['metrics'], fw['side']
        nd = len(fd) - 1
        w5 = dec.params['convT%d/kernel' % nd]
        return self._decoder_backward_rest(locals())

    def _decoder_forward(self, z_act, y, B):
        dec, dev, st, dt = self.dec, self.dev, _st(), self.dt
        D, fd, act = dec.D, dec.filters, dec.act
        Lz = dec.L
        # ---------------- decoder forward
        S, ch = dec.S, dec.ch
        lin = S ** 3 * ch
        c_d0 = self._dense(z_act, dec.packed['wd'], B, lin, Lz, shift=dec.params['dense/bias'])
        t0, bn_d0 = self._bn_fwd(c_d0, B, lin, dec, 'bn_dense', act)
        n1 = S ** 3 * fd[0]
        c_d1 = self._dense(t0, dec.packed['w0'], B, n1, lin)
        h_d1, bn_d1 = self._bn_fwd(c_d1, B * S ** 3, fd[0], dec, 'bnT0', act)
        dc_, dh_, dbn = [c_d1], [h_d1], [bn_d1]
        side = S
        lib = L.load()
        for i in range(1, len(fd) - 1):
            ww = dec.packed.get('ww%d' % i)
            if (ww is not None and not os.environ.get('VV_NO_STATS_FUSION') and not os.environ.get('VV_NO_WHOLE') and not os.environ.get('VV_NO_DIRECT')
                    and lib.vv_convT3d_k4s2_whole_supported(side, fd[i - 1], fd[i], dt)):
                # the widest decoder layer: its kernel leaves the column sums of its own output (no statistics sweep over 134 MB)
                c = self._aempty(B, 2 * side, 2 * side, 2 * side, fd[i])
                nblk = lib.vv_convT3d_k4s2_whole_stats_blocks(B)
                part = self._empty(nblk * 2 * fd[i])
                L.call('vv_convT3d_k4s2_whole_stats_fwd', L.ptr(dh_[-1]), L.ptr(ww), L.ptr(c), L.ptr(part), part.numel() * 4, B, side, fd[i - 1], fd[i],
                       dt, st)
                side *= 2
                h, bn = self._bn_fwd_from_partials(c, part, nblk, B * side ** 3, fd[i], dec, 'bnT%d' % i, act)
            else:
                c = self._convT(dh_[-1], dec.params['convT%d/kernel' % i], B, side, fd[i - 1], fd[i], packed=dec.packed['w%d' % i],
                                packed_frag=dec.packed.get('wf%d' % i), packed_whole=ww)
                side *= 2
                h, bn = self._bn_fwd(c, B * side ** 3, fd[i], dec, 'bnT%d' % i, act)
            dc_.append(c); dh_.append(h); dbn.append(bn)
        nd = len(fd) - 1
        w5 = dec.params['convT%d/kernel' % nd]
        probs = self._empty(B, D, D, D, 1)
        stats = self._empty(B, 4)
        ws = self.ws.get(L.load().vv_convT3d_final_bce_workspace_bytes(B, side))
        L.call('vv_convT3d_final_bce_fwd', L.ptr(dh_[-1]), L.ptr(w5), L.ptr(y), L.ptr(probs), None, L.ptr(stats), B, side, fd[nd - 1],
               0.6, 1e-7, dt, L.ptr(ws), ws.numel(), st)
        metrics = E.shape_metrics(stats)
        return {'c_d0': c_d0, 't0': t0, 'bn_d0': bn_d0, 'c_d1': c_d1, 'bn_d1': bn_d1, 'dc_': dc_, 'dh_': dh_, 'dbn': dbn, 'probs': probs,
                'stats': stats, 'metrics': metrics, 'side': side}

    def _decoder_backward_rest(self, fwd):
        """Backward half of _latent_decoder; `fwd` = its locals (forward intermediates)."""
        dec, dev, st, dt = self.dec, self.dev, _st(), self.dt
        D, fd, act = dec.D, dec.filters, dec.act
        B, y, inv_gb, enc_out, eps, z, z_act, kl = (fwd[k] for k in ('B', 'y', 'inv_gb', 'enc_out', 'eps', 'z', 'z_act', 'kl'))
        drop_mask, drop_scale, Lz, S, ch, lin, n1, nd, w5 = (fwd[k] for k in ('drop_mask', 'drop_scale', 'Lz', 'S', 'ch', 'lin', 'n1', 'nd', 'w5'))
        c_d0, t0, bn_d0, c_d1, bn_d1 = (fwd[k] for k in ('c_d0', 't0', 'bn_d0', 'c_d1', 'bn_d1'))
        dc_, dh_, dbn, probs, stats, metrics, side = (fwd[k] for k in ('dc_', 'dh_', 'dbn', 'probs', 'stats', 'metrics', 'side'))
        # ---------------- backward: decoder tail
        dlogit = self._empty(B, D, D, D, 1)
        L.call('vv_bce_bwd', L.ptr(probs), L.ptr(y), L.ptr(dlogit), B, D ** 3, 0.6, 1e-7, inv_gb, st)
        cl = fd[nd - 1]
        self._wgrad_conv(dlogit, dh_[-1], self._g('dec/convT%d/kernel' % nd), B, D, 1, cl, ready='dec/convT%d/kernel' % nd)   # [64 taps][cl] = Keras [4,4,4,1,cl]
        w5p = self._aempty(cl, 64)
        L.call('vv_pack_conv_k4', L.ptr(w5), L.ptr(w5p), 1, cl, dt, st)
        dh = self._aempty(B, side, side, side, cl)
        L.call('vv_conv3d_first_fwd', L.ptr(dlogit), L.ptr(w5p), None, None, L.ptr(dh), B, D, cl, 0, dt, st)
        for i in range(nd - 1, 0, -1):                       # stride-2 transposed convs
            cin, cout = fd[i - 1], fd[i]
            dcv = self._bn_bwd(dc_[i], dh, dbn[i], B * side ** 3, 'dec/bnT%d/gamma' % i, 'dec/bnT%d/beta' % i, act)
            wk = dec.params['convT%d/kernel' % i]            # Keras [4,4,4,cout,cin]
            self._wgrad_conv(dcv, dh_[i - 1], self._g('dec/convT%d/kernel' % i), B, side, cout, cin, ready='dec/convT%d/kernel' % i)
            dh = self._conv(dcv, wk, B, side, cout, cin)     # read as a forward conv kernel [4,4,4,Cin_c=cout,Cout_c=cin]
            side //= 2
        # D1 (dense panel over the S^3 seed)
        dcv = self._bn_bwd(c_d1, dh, bn_d1, B * S ** 3, 'dec/bnT0/gamma', 'dec/bnT0/beta', act)
        dpanel = self._empty(n1, lin)
        self._wgrad_dense(dcv, t0, dpanel, B, n1, lin)
        L.call('vv_unpack_convT_dense_grad', L.ptr(dpanel), L.ptr(self._g('dec/convT0/kernel')), S, ch, fd[0], st)
        self._ready('dec/convT0/kernel')
        w0f = self._empty(n1, lin)
        L.call('vv_pack_convT_k4s1_dense', L.ptr(dec.params['convT0/kernel']), L.ptr(w0f), S, ch, fd[0], L.VV_F32, st)
        dt0 = self._dense(dcv, self._transposed_panel(w0f, n1, lin), B, lin, n1)
        # D0 (Dense + bias)
        dcv0 = self._bn_bwd(c_d0, dt0, bn_d0, B, 'dec/bn_dense/gamma', 'dec/bn_dense/beta', act)
        L.call('vv_colsum', L.ptr(dcv0), L.ptr(self._g('dec/dense/bias')), B, lin, self._dt(dcv0), st)
        self._wgrad_dense(z_act, dcv0, self._g('dec/dense/kernel'), B, Lz, lin)
        self._ready('dec/dense/bias', 'dec/dense/kernel')
        dz = self._dense(dcv0, self._cast(dec.params['dense/kernel']), B, Lz, lin, f32_out=True)   # Keras [L][lin] is the [N][K] panel of the data gradient

        # ---------------- backward: latent
        if self.var:
            de = self._empty(B, 2 * Lz)
            L.call('vv_reparam_kl_bwd', L.ptr(enc_out), L.ptr(eps), L.ptr(dz), L.ptr(drop_mask), float(drop_scale), L.ptr(de), B, Lz,
                   inv_gb, st)
        else:
            de = dz
        if self.debug is not None:
            self.debug.update({'dlogit': dlogit, 'probs': probs, 'enc_out': enc_out, 'z': z, 'dz': dz, 'de': de, 'c_d0': c_d0, 'dcv0': dcv0,
                               't0': t0, 'dt0': dt0, 'h_dec': dh_, 'c_dec': dc_})
        return kl, stats, metrics, de

    def _encoder_backward(self, x, de, est, B):
        enc, st = self.enc, _st()
        D, fe, act = enc.D, enc.filters, enc.act
        ec, eh, ebn, Sside_e, K5 = est
        ne = len(fe) - 1
        # ---------------- backward: encoder
        E_out = fe[ne]
        dpanel = self._empty(E_out, K5)
        self._wgrad_dense(de, eh[-1], dpanel, B, E_out, K5)
        L.call('vv_unpack_meanpool_grad', L.ptr(dpanel), L.ptr(self._g('enc/conv%d/kernel' % ne)), Sside_e, fe[ne - 1], E_out, st)
        self._ready('enc/conv%d/kernel' % ne)
        wef = self._empty(E_out, K5)
        L.call('vv_pack_conv_k4s1_meanpool', L.ptr(enc.params['conv%d/kernel' % ne]), L.ptr(wef), Sside_e, fe[ne - 1], E_out, L.VV_F32, st)
        dh = self._dense(self._cast(de), self._transposed_panel(wef, E_out, K5), B, K5, E_out)
        side = Sside_e
        for i in range(ne - 1, 0, -1):                       # stride-2 convs
            cin, cout = fe[i - 1], fe[i]
            dcv = self._bn_bwd(ec[i], dh, ebn[i], B * side ** 3, 'enc/bn%d/gamma' % i, 'enc/bn%d/beta' % i, act)
            wk = enc.params['conv%d/kernel' % i]             # Keras [4,4,4,cin,cout]
            self._wgrad_conv(eh[i - 1], dcv, self._g('enc/conv%d/kernel' % i), B, 2 * side, cin, cout, ready='enc/conv%d/kernel' % i)
            dh = self._convT(dcv, wk, B, side, cout, cin)    # read as a transposed kernel [4,4,4,Cout_T=cin,Cin_T=cout]
            side *= 2
        dcv = self._bn_bwd(ec[0], dh, ebn[0], B * side ** 3, 'enc/bn0/gamma', 'enc/bn0/beta', act)
        self._wgrad_conv(x, dcv, self._g('enc/conv0/kernel'), B, D, 1, fe[0], ready='enc/conv0/kernel')
        if self.debug is not None:
            self.debug.update({'h_enc': eh, 'c_enc': ec})

    def _apply(self):
        # ---------------- cross-rank gradient sum, then Adam on every replica
        st = _st()
        self._prepacked = {}                       # Adam moves the weights below: the step's images die with it
        self._join_wgrad()
        self.grads.finish(self.group)
        self.t += 1
        lr_t = self.lr * (1.0 - ADAM_B2 ** self.t) ** 0.5 / (1.0 - ADAM_B1 ** self.t)
        # one launch for all variables: a device table of <= 16384-element chunks, rebuilt only when a tensor moved
        ptrs = tuple(self._p(name).data_ptr() for name, _ in self.order)
        if getattr(self, '_adam_ptrs', None) != ptrs:
            recs = []
            for name, _ in self.order:
                p, g, m, v = self._p(name), self._g(name), self.m[name], self.v[name]
                for off in range(0, p.numel(), ADAM_CHUNK):
                    recs.append((p.data_ptr() + 4 * off, g.data_ptr() + 4 * off, m.data_ptr() + 4 * off, v.data_ptr() + 4 * off,
                                 min(ADAM_CHUNK, p.numel() - off)))
            self._adam_table = torch.from_numpy(np.asarray(recs, dtype=np.int64)).to(self.dev)
            self._adam_ptrs = ptrs
        L.call('vv_adam_step_multi', L.ptr(self._adam_table), self._adam_table.shape[0], lr_t, ADAM_B1, ADAM_B2, ADAM_EPS, st)
        if self.enc is not None:
            self.enc._dirty = True
        self.dec._dirty = True
