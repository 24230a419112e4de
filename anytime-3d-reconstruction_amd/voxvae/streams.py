"""Throughput mode for evaluation: independent batches issued round-robin on several HIP streams.

One evaluation step is a chain of 13 dependent launches; several of them (the 4^3 <-> 2^3 layers, the latent tail, the
reduction passes) are latency-bound and leave most of the chip idle, and every launch has a ramp-up and a tail.  Batches are
independent of each other (reference test loop: test_modelnet_VAE.py:114-130 calls getEval batch after batch), so the next
batch can fill those holes if it runs on its own stream.  The engines keep one split-K / slab workspace PER STREAM (round 3),
so ONE model serves every stream (rounds 1-2 replicated the model per stream, 53 MB of bf16 weights each; `replicas=True` still does).

    ev = StreamedEvaluator(lambda: build_model(), streams=2)
    for x, y, eps in batches:
        outs.append(ev.submit(x, y, eps))      # returns the step's device tensors; asynchronous
    ev.synchronize()

Measured on MI355X (32^3, batch 256, bf16; profiles/microbench/mb_streams3.py, four interleaved repetitions in one process):
0.535 ms/step on one stream, 0.478-0.480 on two, 0.463-0.469 on three, 0.484-0.486 on four.
"""
import torch


class StreamedEvaluator:
    def __init__(self, model_factory, streams=3, device=None, replicas=False, priorities=None):
        """priorities: optional list of HIP stream priorities (0 = normal, -1 = high), one per stream (an experiment knob: a stream that
        wins every arbitration keeps its producer -> consumer hand-offs adjacent on the device)."""
        if streams < 1:
            raise ValueError('streams must be >= 1')
        self.models = [model_factory() for _ in range(streams)] if replicas else [model_factory()] * streams
        self.device = torch.device(device) if device is not None else self.models[0]._device
        pr = list(priorities) if priorities else [0] * streams
        if len(pr) != streams:
            raise ValueError('one priority per stream')
        self.streams = [torch.cuda.Stream(device=self.device, priority=int(pr[i])) for i in range(streams)] if streams > 1 else [None]
        self._next = 0
        self._pack_event, self._pack_pending = None, set()

    def _ensure_packed(self, model):
        """Weight packing is lazy (engine.forward -> ensure_packed) and the streams share one model: whichever stream ran first
        after a weight change would pack, and the next stream -- which only waits for the CALLER's stream -- would read images that
        are still being written (and the repack frees the old images while other streams may still read them).  So the pack runs
        here, on the caller's stream, after every stream has drained what it had in flight, and every stream's next step waits for
        it (an event: the caller may issue its next submit from another stream)."""
        enc, dec = model._enc_eng, model._dec_eng
        if not (enc._dirty or dec._dirty or not enc._folded or not dec._folded):
            return
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            if s is not None:
                cur.wait_stream(s)
        enc.ensure_packed()
        dec.ensure_packed()
        self._pack_event = torch.cuda.Event()
        self._pack_event.record(cur)
        self._pack_pending = set(id(s) for s in self.streams if s is not None)

    def submit(self, x, y, eps=None):
        """Enqueue model.eval_forward_device(x, y, eps) on the next stream; x, y, eps must be ready on the caller's current
        stream (the replica's stream waits for it).  Returns (pred, stats, metrics, kl) device tensors that are valid for
        consumers on the caller's stream after `synchronize()` (or after waiting on `last_stream`)."""
        i = self._next
        self._next = (i + 1) % len(self.models)
        s = self.streams[i]
        self._ensure_packed(self.models[i])
        if s is None:
            return self.models[i].eval_forward_device(x, y, eps)
        s.wait_stream(torch.cuda.current_stream(self.device))
        if self._pack_event is not None and id(s) in self._pack_pending:
            s.wait_event(self._pack_event)
            self._pack_pending.discard(id(s))
        for t in (x, y, eps):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(s)                 # the caller may free its inputs before this stream has read them
        with torch.cuda.stream(s):
            out = self.models[i].eval_forward_device(x, y, eps)
        self.last_stream = s
        return out

    def synchronize(self):
        for s in self.streams:
            if s is not None:
                s.synchronize()
        torch.cuda.current_stream(self.device).synchronize()


class PendingEval:
    """What HostPipeline.submit returns: the getEval tuple of one batch, complete once `get()` has returned."""
    __slots__ = ('_out', '_event')

    def __init__(self, out, event):
        self._out, self._event = out, event

    def done(self):
        return self._event.query()

    def get(self):
        """Waits for this batch only (its kernels and the download of its prediction) and returns getEval's tuple."""
        self._event.synchronize()
        return self._out


class HostPipeline:
    """The reference's test loop (test_modelnet_VAE.py:114-130: getEval on host arrays, np.array(pred), next batch) with the batches
    overlapped: `submit` enqueues one getEval -- upload, kernels, download of the prediction into a pinned block -- on the next of a
    few HIP streams and returns at once; the caller converts batch k while batches k + 1 .. k + depth - 1 are in flight.

        pipe, pending = HostPipeline(model, depth=3), collections.deque()
        for batch in loader:
            pending.append(pipe.submit(inputs=(x, x, onehot), category_vectors=cats))
            if len(pending) == pipe.depth:
                out = pending.popleft().get(); pred = np.array(out[0]); ...
        (drain the deque the same way)

    Same kernels, same per-sample arithmetic as the synchronous call: results are bit-identical (tests/test_gpu_api.py).  The
    synchronous call is bound by its own download (33.6 MB of float32 probabilities per 256-batch = 0.60 ms beside 0.49 ms of kernels);
    here the download of batch k runs under the kernels of batch k + 1."""

    def __init__(self, model, depth=3):
        if depth < 1:
            raise ValueError('depth must be >= 1')
        self.model, self.depth = model, int(depth)
        self.device = model._device
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.depth)]
        self._next = 0

    def submit(self, inputs, category_vectors, missing_prob=0.0, **kw):
        m = self.model
        cur = torch.cuda.current_stream(self.device)
        engines = [e for e in (getattr(m, '_enc_eng', None), getattr(m, '_dec_eng', None)) if e is not None]   # (the image -> 3D model has no voxel encoder)
        if any(e._dirty or not e._folded for e in engines):
            for s in self.streams:                  # weight images are shared by the streams: repack with nothing in flight
                cur.wait_stream(s)
            for e in engines:
                e.ensure_packed()
            for s in self.streams:
                s.wait_stream(cur)
        s = self.streams[self._next]
        self._next = (self._next + 1) % self.depth
        s.wait_stream(cur)
        m._lazy_host = True
        try:
            with torch.cuda.stream(s):
                out = m.getEval(inputs=inputs, category_vectors=category_vectors, missing_prob=missing_prob, **kw)
                ev = torch.cuda.Event()
                ev.record(s)
        finally:
            m._lazy_host = False
        return PendingEval(out, ev)
