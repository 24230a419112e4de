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
"""
import weakref

import numpy as np
import torch

_STATE = {'pred_dtype': 'float32', 'max_outstanding': 8}
_POOL = {}            # nbytes -> [free pinned uint8 tensors]
_OUT = {'n': 0}       # pinned blocks currently owned by caller-visible arrays


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
    nbytes = int(np.prod(shape)) * dtype.itemsize
    block = _pinned_block(nbytes)
    if block is None:
        return None, None
    root = block.numpy()                              # a fresh ndarray over the block: numpy makes it the .base of every view below
    arr = root.view(dtype).reshape(shape)
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
                                            torch.int32: 'int32'}.get(t.dtype, None) or t.cpu().numpy().dtype)
    if arr is None:
        return t.detach().cpu().numpy()
    tv.copy_(t.detach(), non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return arr


class HostPrediction(object):
    """What the chunked getEval returns for `pred`: the host array is already there (its download ran under the kernels of the
    next chunk); the device chunks are kept for callers that want the tensor."""
    __slots__ = ('host', 'chunks', '_handed_out')

    def __init__(self, host, chunks):
        self.host, self.chunks, self._handed_out = host, chunks, False

    @property
    def shape(self):
        return tuple(self.host.shape)

    @property
    def dtype(self):
        return self.host.dtype

    @property
    def t(self):
        return torch.cat(self.chunks, dim=0)

    def torch(self):
        return self.t

    def numpy(self):
        return self.host

    def __array__(self, dtype=None, copy=None):
        # np.array(pred) asks for a copy: the FIRST caller gets the pinned-block array itself (nobody else writes to it; that
        # is the whole point of downloading into it), later callers and explicit dtype changes get real copies
        if dtype is not None and np.dtype(dtype) != self.host.dtype:
            return self.host.astype(dtype)
        if copy is False or not self._handed_out:
            self._handed_out = self._handed_out or copy is not False
            return self.host
        return self.host.copy()

    def __len__(self):
        return self.host.shape[0]

    def __getitem__(self, idx):
        return self.host[idx]

    def __repr__(self):
        return 'HostPrediction(shape=%s, dtype=%s)' % (self.shape, self.dtype)
