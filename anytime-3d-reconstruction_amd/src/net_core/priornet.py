"""Class-conditional prior network, mirror of the reference's src/net_core/priornet.py:12-59.

A tiny MLP over one-hot class vectors (40 -> ... -> latent), outside the voxel hot path:
stock PyTorch ops.  `priornet(structure)` returns a callable `model(onehot, training=False) -> (mean, log_var)` with
the Keras attributes the model class uses (`.trainable_variables`, `.losses`, `.save_weights`, `.load_weights`).

Restated semantics: input 2x-1; hidden layers Dense(bias, l2 0.0005) -> BatchNormalization(eps 1e-3, momentum 0.99)
-> Dropout(0.2) -> activation; last layer Dense(bias, l2 0.0005) without normalisation; `const_log_var`: None -> a
second, independent branch of the same shape produces log_var; a number -> that constant; NaN -> zeros."""
import numpy as np
import torch
import torch.nn as nn

priornet_structure = {
    'name': 'priornet',
    'input_dim': 40,  # class num (one-hot vector)
    'unit_num_list': [64, 32, 16],
    'core_activation': 'elu',
    'const_log_var': None,
}


def _act(name):
    return {'lrelu': nn.LeakyReLU(0.3), 'relu': nn.ReLU(), 'elu': nn.ELU()}.get(name, nn.Identity())   # Keras LeakyReLU() default 0.3


def _dense(cin, cout):
    d = nn.Linear(cin, cout, bias=True)
    nn.init.xavier_uniform_(d.weight)
    nn.init.zeros_(d.bias)
    return d


def _branch(input_dim, units, act):
    layers, cin = [], input_dim
    for u in units[:-1]:                                   # priorDense, priornet.py:12-25
        layers += [_dense(cin, u), nn.BatchNorm1d(u, eps=1e-3, momentum=0.01), nn.Dropout(0.2), _act(act)]
        cin = u
    layers.append(_dense(cin, units[-1]))
    return nn.Sequential(*layers)


class _PriorNet(nn.Module):
    def __init__(self, structure, device=None):
        super().__init__()
        self.name = structure['name']
        self._device = torch.device(device if device is not None else ('cuda:0' if torch.cuda.is_available() else 'cpu'))
        units, act = list(structure['unit_num_list']), structure['core_activation']
        self.mean = _branch(structure['input_dim'], units, act)
        c = structure.get('const_log_var')
        self.const_log_var = c
        self.log_var = _branch(structure['input_dim'], units, act) if c is None else None
        self.to(self._device)
        self.eval()

    def __call__(self, x, training=False):
        if not torch.is_tensor(x):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        x = x.to(self._device, torch.float32)
        self.train(bool(training))
        with torch.set_grad_enabled(bool(training)):
            return nn.Module.__call__(self, x)

    def forward(self, x):
        x = 2.0 * x - 1.0
        m = self.mean(x)
        if self.log_var is not None:
            return m, self.log_var(x)
        c = self.const_log_var
        return m, (float(c) * torch.ones_like(m) if c == c else torch.zeros_like(m))

    @property
    def trainable_variables(self):
        return [p for p in self.parameters() if p.requires_grad]

    @property
    def losses(self):
        return [0.0005 * (m.weight ** 2).sum() for m in self.modules() if isinstance(m, nn.Linear)]

    def save_weights(self, path):
        torch.save(self.state_dict(), path + '.pt')

    def load_weights(self, path):
        self.load_state_dict(torch.load(path + '.pt', map_location=self._device))


def priornet(structure, device=None):
    print('priornet', structure['name'])
    return _PriorNet(structure, device)
