"""Algorithmic work of the hot path per reconstruction: "valid" MACs count
only the filter taps that touch real data (TF 'SAME' zero padding skipped).  Used by bench.py's roofline leg."""


def _axis_pairs_s2(n_in):
    """(output, tap) pairs per axis of a k4 s2 SAME conv on n_in cells (= pairs of its transposed conv)."""
    return sum(1 for o in range(n_in // 2) for t in range(4) if 0 <= 2 * o - 1 + t < n_in)


def _axis_pairs_s1(n):
    """k4 s1 SAME (pad 1 before, 2 after)."""
    return sum(1 for o in range(n) for t in range(4) if 0 <= o - 1 + t < n)


def layer_macs(config):
    """-> list of (layer name, valid MACs per sample, dense-im2col MACs per sample)."""
    enc, dec = config['encoder'], config['decoder']
    out = []
    side, cin = enc['input_shape'][0], 1
    f = enc['filter_num_list']
    for i, c in enumerate(f[:-1]):
        out.append(('E%d' % (i + 1), _axis_pairs_s2(side) ** 3 * cin * c, (side // 2) ** 3 * 64 * cin * c))
        side, cin = side // 2, c
    out.append(('E%d' % len(f), _axis_pairs_s1(side) ** 3 * cin * f[-1], side ** 3 * 64 * cin * f[-1]))
    f = dec['filter_num_list']
    n = len(f)
    side = dec['output_shape'][0] >> (n - 1)
    ch = max(f[0] // 64, 8)
    out.append(('D0', dec['input_dim'] * side ** 3 * ch, dec['input_dim'] * side ** 3 * ch))
    out.append(('D1', _axis_pairs_s1(side) ** 3 * ch * f[0], side ** 3 * 64 * ch * f[0]))
    cin = f[0]
    for i in range(1, n):
        out.append(('D%d' % (i + 1), _axis_pairs_s2(2 * side) ** 3 * cin * f[i], side ** 3 * 64 * cin * f[i]))
        side, cin = 2 * side, f[i]
    return out


def flops_per_reconstruction(config):
    lm = layer_macs(config)
    return 2 * sum(v for _, v, _ in lm), 2 * sum(d for _, _, d in lm)
