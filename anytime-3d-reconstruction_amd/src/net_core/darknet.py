"""2D image encoders of the image -> 3D models, mirror of the reference's src/net_core/darknet.py.

Outside the voxel hot path: stock PyTorch ops, no hand-written kernels.  Same builder
names and arguments as the reference -- `Darknet19(name, activation)` (darknet.py:96-135) and
`head2D(name, input_shape, output_dim, filter_num_list, filter_size_list, last_pooling, activation)`
(darknet.py:152-173) -- returning callables `model(x, training=False)` over channels-last images [B,H,W,3] with the
Keras attributes the model classes use: `.output_shape`, `.trainable_variables`, `.losses`, `.save_weights`,
`.load_weights`.

Keras semantics restated: Conv2D 'same' stride 1 without bias, BatchNormalization(momentum 0.99, epsilon 1e-3),
ELU / LeakyReLU(0.1) / ReLU, MaxPool2D(2, 2, 'same') (= ceil-mode pooling), Glorot-uniform kernels, and the
kernel_regularizer l2(0.0005) that only the head's convolutions carry (darknet.py:140-141,160-161).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _act(name):
    if name == 'lrelu':
        return nn.LeakyReLU(0.1)
    if name == 'elu':
        return nn.ELU()
    if name == 'relu':
        return nn.ReLU()
    return nn.Identity()


class _ConvBNAct(nn.Module):
    """Darknet19Conv / convHead (darknet.py:83-94, 137-150): Conv2D(same, no bias) -> BatchNormalization -> activation."""

    def __init__(self, cin, cout, k, activation, l2=0.0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=1, padding=k // 2, bias=False)
        nn.init.xavier_uniform_(self.conv.weight)
        self.bn = nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01)   # Keras momentum 0.99 = torch momentum 0.01
        self.act = _act(activation)
        self.l2 = l2

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class _Keras2D(nn.Module):
    """Channels-last in / out, `model(x, training=...)` call form, and the handful of Keras attributes the callers use."""

    def __init__(self, name, device=None):
        super().__init__()
        self.name = name
        self._device = torch.device(device if device is not None else ('cuda:0' if torch.cuda.is_available() else 'cpu'))

    def _finish(self):
        self.to(self._device)
        self.eval()
        return self

    def body(self, x):
        raise NotImplementedError

    def __call__(self, x, training=False):
        if not torch.is_tensor(x):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        x = x.to(self._device, torch.float32)
        self.train(bool(training))
        with torch.set_grad_enabled(bool(training)):
            return nn.Module.__call__(self, x)

    def forward(self, x):
        return self.body(x.permute(0, 3, 1, 2))

    @property
    def trainable_variables(self):
        return [p for p in self.parameters() if p.requires_grad]

    @property
    def losses(self):
        """Keras `model.losses`: one l2 term per regularised kernel."""
        return [m.l2 * (m.conv.weight ** 2).sum() for m in self.modules() if isinstance(m, _ConvBNAct) and m.l2 > 0] + \
               [l2 * (w ** 2).sum() for w, l2 in getattr(self, '_extra_l2', [])]

    def save_weights(self, path):
        torch.save(self.state_dict(), path + '.pt')

    def load_weights(self, path):
        self.load_state_dict(torch.load(path + '.pt', map_location=self._device))


class _Darknet19(_Keras2D):
    # (filters, kernel) runs separated by 'M' = MaxPool2D(2, 2, 'same'); darknet.py:99-133
    _PLAN = [(32, 3), 'M', (64, 3), 'M', (128, 3), (64, 1), (128, 3), 'M', (256, 3), (128, 1), (256, 3), 'M',
             (512, 3), (256, 1), (512, 3), (256, 1), (512, 3), 'M', (1024, 3), (512, 1), (1024, 3), (512, 1), (1024, 3)]

    def __init__(self, name=None, activation='elu', device=None):
        super().__init__(name, device)
        layers, cin = [], 3
        for item in self._PLAN:
            if item == 'M':
                layers.append(nn.MaxPool2d(2, 2, ceil_mode=True))
            else:
                layers.append(_ConvBNAct(cin, item[0], item[1], activation))
                cin = item[0]
        self.layers = nn.Sequential(*layers)
        self.output_shape = (None, None, None, cin)
        self._finish()

    def body(self, x):
        return self.layers(x).permute(0, 2, 3, 1)


class _Head2D(_Keras2D):
    def __init__(self, name, input_shape, output_dim, filter_num_list, filter_size_list, last_pooling=None, activation='elu',
                 device=None):
        super().__init__(name, device)
        cin, layers = int(input_shape[-1]), []
        for c, k in zip(filter_num_list, filter_size_list):
            layers.append(_ConvBNAct(cin, c, k, activation, l2=0.0005))
            cin = c
        self.layers = nn.Sequential(*layers)
        self.last = nn.Conv2d(cin, output_dim, 1, bias=False)
        nn.init.xavier_uniform_(self.last.weight)
        self._extra_l2 = [(self.last.weight, 0.0005)]
        self.last_pooling = last_pooling
        self.output_shape = (None, output_dim) if last_pooling in ('max', 'average') else (None, None, None, output_dim)
        self._finish()

    def forward(self, x):                                  # input is the backbone's channels-last feature map
        x = self.last(self.layers(x.permute(0, 3, 1, 2)))
        if self.last_pooling == 'max':
            return x.amax(dim=(2, 3))
        if self.last_pooling == 'average':
            return x.mean(dim=(2, 3))
        return x.permute(0, 2, 3, 1)


def Darknet19(name=None, activation='elu', device=None):
    print('Darknet19', name)
    m = _Darknet19(name=name, activation=activation, device=device)
    print('end Darknet19')
    return m


def head2D(name, input_shape, output_dim, filter_num_list, filter_size_list, last_pooling=None, activation='elu', device=None):
    print('head start')
    m = _Head2D(name, input_shape, output_dim, filter_num_list, filter_size_list, last_pooling, activation, device)
    print('end head2D')
    return m
