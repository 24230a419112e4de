"""Host-array calling convention: numpy in, numpy out, every call -- what the reference's loops do
(test_modelnet_VAE.py:114-130: `getEval(inputs=numpy ...)` then `np.array(output_images_pred)`).

Measured on the GPU box (profiles/microbench/mb_h2d.py; 33.6 MB = one 256-batch of 32^3 float32 grids):

    host -> device, pageable numpy source      0.60 ms   (PCIe rate; staging through pinned memory gains nothing)
    device -> host into a pinned buffer        0.60 ms
    device -> host, `.cpu().numpy()`           4.3  ms   (pageable destination)
    np.empty(33.6 MB) + first touch            20   ms   (8192 page faults: a FRESH array per call is the expensive part)
    host -> device and device -> host at once  1.2  ms   (they do not overlap on this platform)

Two things follow.  (1) Predictions are copied into PINNED buffers that are recycled: `to_host()` hands the caller a numpy
array that lives in a pinned block and takes the block back when the array is garbage-collected (weakref finaliser) -- a caller
that drops or overwrites its previous result never pays a page fault or a second copy; a caller that keeps every result
(the reference's test loop appends them to a list) gets fresh pinned blocks up to a cap and pageable arrays after that.
(2) getEval splits a host batch into two halves: the upload of the second half runs under the kernels of the first, and the
download of the first under the kernels of the second (two streams, ONE engine: the engines keep a workspace per stream).
Every kernel of the path computes a sample independently of the others in its batch with a summation order that does not
depend on the batch size, so the halves give the same bits as the whole batch (tests/test_gpu_api.py pins that).

`set_prediction_host_dtype('float16' | 'uint8')` (opt-in) converts the probabilities on the device before they cross PCIe:
half / a quarter of the bytes ('uint8' = the occupancy p >= 0.5, what the precision / recall tooling thresholds anyway).

Round 4, the input side: occupancy grids are {0,1}, so a batch is 1 bit per voxel -- 1 MB instead of 33.6 MB.  Packing a float32
batch per call on the host costs more than it saves (one numpy pass over 33.6 MB is ~5 ms; the upload it would replace is 0.6 ms), so
the bits are made ONCE, where the data enters: `PackedVoxels` is a host batch that keeps the bits and looks like the float32 array
(`np.array()`, `.shape`, indexing); `dataLoader(..., packed=True)` keeps its split as bits and hands out PackedVoxels batches, and every
model method uploads the bits and unpacks them on the device (`vv_unpack_bits_gather`) -- the same float32 tensor, bit for bit.
"""
import weakref

import numpy as np
import torch

_STATE = {'pred_dtype': 'float32', 'max_outstanding': 8, 'max_pooled_bytes': 1 << 30}
_POOL = {}            # bucket bytes -> [free pinned uint8 tensors]
_OUT = {'n': 0}       # pinned blocks currently owned by caller-visible arrays
_LRU = []             # bucket sizes in order of last use (oldest first): what is evicted when the free blocks pass max_pooled_bytes


def _bucket(nbytes):
    """Block sizes are rounded up to 1/8-octave buckets so that many distinct tensor sizes share few pinned blocks."""
    nbytes = max(int(nbytes), 4096)
    top = 1 << (nbytes - 1).bit_length()
    step = max(top >> 3, 4096)
    return (nbytes + step - 1) // step * step


def pooled_bytes():
    return sum(k * len(v) for k, v in _POOL.items())


def set_prediction_host_dtype(name):
    """'float32' (default: the reference's contract), 'float16' or 'uint8' (occupancy p >= 0.5): what np.array(prediction)
    returns for the large prediction tensors.  Opt-in: it changes the dtype the caller sees."""
    if name not in ('float32', 'float16', 'uint8'):
        raise ValueError(name)
    _STATE['pred_dtype'] = name


def prediction_host_dtype():
    return _STATE['pred_dtype']


def _release(block, nbytes):
    _OUT['n'] -= 1
    _POOL.setdefault(nbytes, []).append(block)
    if nbytes in _LRU:
        _LRU.remove(nbytes)
    _LRU.append(nbytes)
    # page-locked memory is a machine-wide resource: free blocks beyond the cap go back to the OS, least recently used size first
    while pooled_bytes() > _STATE['max_pooled_bytes'] and _LRU:
        old = _LRU[0]
        if _POOL.get(old):
            _POOL[old].pop()
        if not _POOL.get(old):
            _POOL.pop(old, None)
            _LRU.pop(0)


def _pinned_block(nbytes):
    free = _POOL.get(nbytes)
    if free:
        return free.pop()
    if _OUT['n'] >= _STATE['max_outstanding']:
        return None                                   # the caller keeps everything: do not pin the machine down
    return torch.empty(nbytes, dtype=torch.uint8).pin_memory()


def pinned_array(shape, dtype):
    """A numpy array in a recycled pinned block (None when the cap of outstanding blocks is reached), plus the torch view
    of the same memory for `copy_(..., non_blocking=True)`."""
    dtype = np.dtype(dtype)
    used = int(np.prod(shape)) * dtype.itemsize
    nbytes = _bucket(used)
    block = _pinned_block(nbytes)
    if block is None:
        return None, None
    root = block.numpy()                              # a fresh ndarray over the block: numpy makes it the .base of every view below
    arr = root[:used].view(dtype).reshape(shape)
    _OUT['n'] += 1
    weakref.finalize(root, _release, block, nbytes)   # ... so the block returns to the pool only when the last view is gone
    tview = torch.from_numpy(arr)
    return arr, tview


_TORCH = {'float32': torch.float32, 'float16': torch.float16, 'uint8': torch.uint8}


def device_prediction_as(t):
    """The device-side conversion of a probability tensor to the opted-in host dtype."""
    pd = _STATE['pred_dtype']
    if pd == 'float16':
        return t.to(torch.float16)
    if pd == 'uint8':
        return (t >= 0.5).to(torch.uint8)
    return t


def to_host(t, big=1 << 20):
    """Device tensor -> numpy.  Large float32 tensors go through a recycled pinned block (one async copy + a stream
    synchronise, no pageable staging); everything else takes the plain path."""
    if t.dtype == torch.bfloat16:
        t = t.float()
    if not t.is_cuda or t.numel() * t.element_size() < big or not t.is_contiguous():
        return t.detach().cpu().numpy()
    arr, tv = pinned_array(tuple(t.shape), {torch.float32: 'float32', torch.float16: 'float16', torch.uint8: 'uint8',
                                            torch.int32: 'int32'}.get(t.dtype, None) or torch.empty(0, dtype=t.dtype).numpy().dtype)
    if arr is None:
        return t.detach().cpu().numpy()
    tv.copy_(t.detach(), non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return arr


class PackedVoxels(object):
    """A host batch of occupancy grids as packed bits (voxel v of a sample = bit v & 7 of byte v >> 3, the layout of vv_pack_bits /
    vv_unpack_bits_gather and of np.packbits(..., bitorder='little')) that stands in for the float32 array `[B,D,D,D,1]` the
    reference's loader returns (modelnet_dataset.py:83): `np.array(p)` / `p[i]` give the float32 values, `.shape` / `.dtype` / `len()`
    are the float array's.  The model classes recognise it and move 1 bit per voxel across PCIe."""
    __slots__ = ('bits', 'shape')

    def __init__(self, bits, shape):
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        shape = tuple(int(v) for v in shape)
        vox = int(np.prod(shape[1:]))
        if vox % 8 or bits.shape != (shape[0], vox // 8):
            raise ValueError('bits %s do not match the float shape %s' % (bits.shape, shape))
        self.bits, self.shape = bits, shape

    dtype = np.dtype('float32')
    ndim = property(lambda self: len(self.shape))

    def __len__(self):
        return self.shape[0]

    def _unpack(self, bits):
        return np.unpackbits(bits, axis=1, bitorder='little').astype(np.float32).reshape((bits.shape[0],) + self.shape[1:])

    def __array__(self, dtype=None, copy=None):
        a = self._unpack(self.bits)
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            return self._unpack(self.bits[idx:idx + 1] if idx != -1 else self.bits[-1:])[0]
        if isinstance(idx, tuple):
            return np.asarray(self)[idx]
        sub = self.bits[idx]
        return PackedVoxels(sub, (sub.shape[0],) + self.shape[1:])

    def to_device(self, device, lo=0, hi=None):
        """float32 CUDA tensor of samples [lo, hi): 1 bit per voxel over PCIe, unpacked by vv_unpack_bits_gather on the current stream."""
        import ctypes
        from . import lib as L
        hi = self.shape[0] if hi is None else hi
        n, vox = hi - lo, int(np.prod(self.shape[1:]))
        dbits = torch.from_numpy(self.bits[lo:hi]).to(device)
        out = torch.empty((n,) + self.shape[1:], dtype=torch.float32, device=device)
        L.call('vv_unpack_bits_gather', L.ptr(dbits), None, L.ptr(out), n, vox, ctypes.c_void_p(torch.cuda.current_stream(out.device).cuda_stream))
        return out

    def __repr__(self):
        return 'PackedVoxels(shape=%s, %d bytes of bits)' % (self.shape, self.bits.nbytes)


def pack_voxels(x, threshold=0.5):
    """float array [B,...] -> PackedVoxels (bit = x > threshold: the reference binarises its grids the same way when it loads them,
    modelnet_dataset.py:25).  For a {0,1} array `np.array(pack_voxels(x))` is x."""
    x = np.asarray(x)
    flat = x.reshape(x.shape[0], -1)
    return PackedVoxels(np.packbits(flat > threshold, axis=1, bitorder='little'), x.shape)


class HostPrediction(object):
    """What the chunked getEval returns for `pred`: the host array is already there (its download ran under the kernels of the
    next chunk); the device chunks are kept for callers that want the tensor."""
    __slots__ = ('host', 'chunks', '_t', '_handed_out', '_ready')

    def __init__(self, host, chunks, ready=None):
        """ready: CUDA events the downloads complete behind (the pipelined form, voxvae.streams.HostPipeline): the first access to the
        host array waits for them; None = the array is complete (the synchronous getEval)."""
        self.host, self.chunks, self._t, self._handed_out, self._ready = host, chunks, None, False, ready

    def wait(self):
        if self._ready:
            for e in self._ready:
                e.synchronize()
            self._ready = None
        return self

    @property
    def shape(self):
        return tuple(self.host.shape)

    @property
    def dtype(self):
        return self.host.dtype

    @property
    def t(self):
        if self._t is None:
            self._t = self.chunks[0] if len(self.chunks) == 1 else torch.cat(self.chunks, dim=0)
        return self._t

    def torch(self):
        return self.t

    def numpy(self):
        self.wait()
        return self._view()

    def __array__(self, dtype=None, copy=None):
        """`np.array(pred)` -- what the reference's loop does with the prediction (test_modelnet_VAE.py:128) -- asks numpy for a
        copy.  A fresh 33.6 MB array costs 20 ms of page faults on first touch (header of this file), 12x the whole call, and the
        block was downloaded for this call only, so the FIRST such caller is handed the block's array itself as its copy (writable,
        its own from then on; the block returns to the pool when that array is dropped).  Everything after that -- `pred.numpy()`,
        `pred[idx]`, `np.asarray(pred)`, a second `np.array(pred)` -- sees the same memory through READ-ONLY views (later np.array
        calls get real copies), so nothing can be changed through them behind the first caller's back; an in-place edit by the first
        caller of its own array does show in those views.  A different dtype always copies."""
        self.wait()
        if dtype is not None and np.dtype(dtype) != self.host.dtype:
            return self.host.astype(dtype)
        if copy is False:
            return self._view()
        if not self._handed_out:
            self._handed_out = True
            return self.host
        return self.host.copy()

    def _view(self):
        if not self._handed_out:
            return self.host
        v = self.host.view()
        v.flags.writeable = False
        return v

    def __len__(self):
        return self.host.shape[0]

    def __getitem__(self, idx):
        self.wait()
        return self._view()[idx]

    def __repr__(self):
        return 'HostPrediction(shape=%s, dtype=%s)' % (self.shape, self.dtype)
