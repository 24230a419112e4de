"""DeviceArray: what the model API hands back in place of a tf.Tensor.

The reference's callers only ever do `np.array(t)`, `float(t)`, `len(t)` and tuple-unpacking on
what getEval/fit return (test_modelnet_VAE.py:128-138), so this wrapper offers exactly that over a
torch CUDA tensor; conversion to host happens (and synchronises) only when asked for."""
import numpy as np
import torch


class DeviceArray(object):
    __slots__ = ('t',)

    def __init__(self, t):
        self.t = t

    @property
    def shape(self):
        return tuple(self.t.shape)

    @property
    def dtype(self):
        return np.dtype('float32') if self.t.dtype in (torch.float32, torch.bfloat16) else np.dtype('int32')

    def torch(self):
        return self.t

    def numpy(self):
        from . import hostio
        t = self.t
        if t.is_cuda and t.dim() == 5 and t.dtype == torch.float32 and hostio.prediction_host_dtype() != 'float32':
            t = hostio.device_prediction_as(t)          # opt-in: probabilities cross PCIe as float16 / uint8 occupancy
        return hostio.to_host(t)                        # large tensors: recycled pinned block, no pageable staging

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __float__(self):
        return float(self.numpy().reshape(-1)[0])

    def __len__(self):
        return self.t.shape[0]

    def __getitem__(self, idx):
        """A DeviceArray of the indexed part (still on the device); `np.array(a[idx])` / `float(a[idx])` bring it to the host.
        (HostPrediction, the host-array path's `pred`, indexes into its downloaded block and returns numpy: both support
        `np.array(pred[i])`, which is what the reference's callers do.)"""
        return DeviceArray(self.t[idx])

    def __repr__(self):
        return 'DeviceArray(shape=%s, dtype=%s, device=%s)' % (self.shape, self.t.dtype, self.t.device)


def as_device_f32(x, device):
    """numpy / DeviceArray / torch -> contiguous float32 tensor on `device` (host->device copy if needed)."""
    if isinstance(x, DeviceArray):
        x = x.t
    from .hostio import HostPrediction, PackedVoxels
    if isinstance(x, PackedVoxels):                    # 1 bit per voxel over PCIe, unpacked on the device (voxvae/hostio.py)
        return x.to_device(device)
    if isinstance(x, HostPrediction):
        x = x.t
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device)
