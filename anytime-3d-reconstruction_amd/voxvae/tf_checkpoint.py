"""Reader / writer for TensorFlow checkpoint files ("tensor bundles"), in pure Python + numpy.

The reference saves its models with `tf.keras.Model.save_weights(<dir>/<structure name>)` (reference nolbo.py:1568-1574):
without an `.h5` suffix that is the TF checkpoint format -- `<prefix>.index` + `<prefix>.data-00000-of-00001`.  TensorFlow is not
installed here and cannot be, so this module reads (and, for round trips and for handing weights back, writes) that format itself:

    <prefix>.index   an SSTable in the LevelDB table format (tensorflow/core/lib/io/table*): data blocks of prefix-compressed
                     (key, value) entries + restart array, each followed by a 5-byte trailer (compression type, masked CRC32C);
                     an index block of (last key -> BlockHandle); an (empty) metaindex block; a 48-byte footer holding the two
                     BlockHandles and the magic number 0xdb4775248b80fb57.  Key "" holds a BundleHeaderProto, every other key a
                     BundleEntryProto {dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5, crc32c = 6}.
    <prefix>.data-NNNNN-of-MMMMM   the tensors' little-endian bytes at [offset, offset + size) of their shard.

Keras writes object-graph keys: `layer_with_weights-<i>/<attribute>/.ATTRIBUTES/VARIABLE_VALUE`, i = the layer's index among the
model's layers THAT HAVE WEIGHTS, in creation order (Conv3D, BatchNormalization, ... as autoencoder3D.py:26-139 creates them),
attribute in {kernel, bias, gamma, beta, moving_mean, moving_variance}; `load_keras_checkpoint` maps them onto this repo's
parameter names through the same per-layer order `Model.keras_variable_names` uses.

HONESTY NOTE: no file written by TensorFlow exists in this environment (the reference ships no weights), so the
reader is tested against this module's own writer and against hand-built byte strings of the format's documented pieces
(tests/test_host_logic.py).  It verifies the footer magic and every index / data block CRC of the table it parses, so a file it
misunderstands fails loudly rather than yielding wrong weights; snappy-compressed blocks (not produced by TensorFlow's bundle
writer, which sets kNoCompression) are refused.  Tensor CRCs in the data shards are not checked (CRC32C over 100 MB in Python).
"""
import os
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'), 5: np.dtype('<i2'), 6: np.dtype('i1'),
           9: np.dtype('<i8'), 10: np.dtype('?'), 17: np.dtype('<u2'), 19: np.dtype('<f2'), 22: np.dtype('<u4'), 23: np.dtype('<u8')}
_DT_STRING, _DT_BFLOAT16 = 7, 14
_DT_OF = {np.dtype('<f4'): 1, np.dtype('<f8'): 2, np.dtype('<i4'): 3, np.dtype('<i8'): 9}

# ---------------------------------------------------------------------------------------------- CRC32C (Castagnoli), masked as LevelDB does
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data, crc=0):
    t = _crc_table()
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


_CRC_LANE = 4096          # bytes per lane of crc32c_bulk
_CRC_ADV = None           # [4][256] uint32: the operator "append _CRC_LANE zero bytes" on a raw CRC register, byte-sliced


def _crc_advance_tables():
    """CRC registers are linear over GF(2): feeding n zero bytes maps the register through a fixed 32 x 32 bit matrix.  The images of
    the 32 unit vectors are computed once by running the byte-wise recurrence, then folded into four 256-entry tables."""
    global _CRC_ADV
    if _CRC_ADV is None:
        t = np.array(_crc_table(), dtype=np.uint32)
        basis = np.uint32(1) << np.arange(32, dtype=np.uint32)
        for _ in range(_CRC_LANE):
            basis = t[basis & 0xFF] ^ (basis >> 8)
        adv = np.zeros((4, 256), dtype=np.uint32)
        for byte in range(4):
            for v in range(256):
                acc = np.uint32(0)
                for bit in range(8):
                    if v >> bit & 1:
                        acc ^= basis[8 * byte + bit]
                adv[byte, v] = acc
        _CRC_ADV = adv
    return _CRC_ADV


def crc32c_bulk(data):
    """crc32c of a large buffer (tensor bytes: tens of MB) in numpy: the buffer is cut into lanes of 4096 bytes whose raw registers
    are advanced TOGETHER, one byte position per numpy operation, and the lane results are chained with the precomputed
    "append 4096 zero bytes" operator (CRC linearity).  Equal to crc32c() on every input (tests/test_host_logic.py)."""
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    n = a.size
    if n < 4 * _CRC_LANE:
        return crc32c(a.tobytes())
    lanes = n // _CRC_LANE
    body = a[:lanes * _CRC_LANE].reshape(lanes, _CRC_LANE)
    t = np.array(_crc_table(), dtype=np.uint32)
    reg = np.zeros(lanes, dtype=np.uint32)
    reg[0] = 0xFFFFFFFF                                  # the initial register enters through the first lane only
    for j in range(_CRC_LANE):
        reg = t[(reg ^ body[:, j]) & 0xFF] ^ (reg >> 8)
    adv = _crc_advance_tables()
    c = 0
    for r in reg.tolist():                               # c = advance(c) ^ lane register
        c = int(adv[0, c & 0xFF] ^ adv[1, (c >> 8) & 0xFF] ^ adv[2, (c >> 16) & 0xFF] ^ adv[3, (c >> 24) & 0xFF]) ^ r
    tt = _crc_table()
    for b in a[lanes * _CRC_LANE:].tolist():             # the tail, byte-wise
        c = tt[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------- varints / protobuf pieces
def _get_varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf):
    """Flat protobuf parse -> {field number: [values]} (varints as int, fixed32/64 as int, length-delimited as bytes)."""
    out, pos = {}, 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('protobuf wire type %d' % wt)
        out.setdefault(field, []).append(v)
    return out


def _shape_of(shape_bytes):
    dims = []
    for d in _parse_proto(shape_bytes).get(2, []):          # TensorShapeProto.dim
        size = _parse_proto(d).get(1, [0])[0]
        dims.append(size - (1 << 64) if size >= 1 << 63 else size)
    return tuple(dims)


def _field(num, wt, payload):
    return _put_varint((num << 3) | wt) + payload


def _entry_proto(dtype, shape, shard, offset, size, crc):
    dims = b''.join(_field(2, 2, _put_varint(len(p)) + p) for p in (_field(1, 0, _put_varint(int(s))) for s in shape))
    msg = _field(1, 0, _put_varint(dtype)) + _field(2, 2, _put_varint(len(dims)) + dims)
    if shard:
        msg += _field(3, 0, _put_varint(shard))
    if offset:
        msg += _field(4, 0, _put_varint(offset))
    msg += _field(5, 0, _put_varint(size)) + _field(6, 5, struct.pack('<I', crc))
    return msg


# ---------------------------------------------------------------------------------------------- table (SSTable) reading
def _read_block(buf, offset, size):
    """Block contents at a BlockHandle, after checking its trailer (1 byte compression type + masked crc32c of contents + type)."""
    body, ctype = buf[offset:offset + size], buf[offset + size]
    stored = struct.unpack_from('<I', buf, offset + size + 1)[0]
    if _mask(crc32c(bytes(body) + bytes([ctype]))) != stored:
        raise ValueError('checkpoint index: block CRC mismatch at offset %d' % offset)
    if ctype != 0:
        raise NotImplementedError('checkpoint index: compressed block (type %d); TensorFlow bundle writers use no compression' % ctype)
    return body


def _block_entries(block):
    """(key, value) pairs of a table block (keys prefix-compressed against the previous one)."""
    nrestarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestarts
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def read_index(path):
    """<prefix>.index -> [(key bytes, value bytes)] in key order."""
    buf = memoryview(open(path, 'rb').read())
    if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError('%s is not a TensorFlow checkpoint index (bad table magic)' % path)
    foot = buf[len(buf) - 48:]
    _, p = _get_varint(foot, 0)
    _, p = _get_varint(foot, p)                     # metaindex handle (unused)
    ioff, p = _get_varint(foot, p)
    isize, p = _get_varint(foot, p)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, isize)):
        off, q = _get_varint(handle, 0)
        size, q = _get_varint(handle, q)
        out.extend(_block_entries(_read_block(buf, off, size)))
    return out


def read_checkpoint(prefix):
    """{key: numpy array} of every numeric tensor of the checkpoint `<prefix>.index` / `<prefix>.data-*`."""
    entries = read_index(prefix + '.index')
    if not entries or entries[0][0] != b'':
        raise ValueError('checkpoint index lacks the bundle header')
    header = _parse_proto(entries[0][1])
    num_shards = header.get(1, [1])[0]
    if header.get(2, [0])[0] != 0:
        raise NotImplementedError('big-endian checkpoint')
    shards, out = {}, {}
    for key, val in entries[1:]:
        e = _parse_proto(val)
        dt, shard = e.get(1, [0])[0], e.get(3, [0])[0]
        if dt == _DT_STRING or 7 in e:                # strings (the object graph) and sliced tensors carry no weights of ours
            continue
        offset, size = e.get(4, [0])[0], e.get(5, [0])[0]
        if shard not in shards:
            shards[shard] = np.memmap('%s.data-%05d-of-%05d' % (prefix, shard, num_shards), dtype=np.uint8, mode='r')
        raw = np.asarray(shards[shard][offset:offset + size])
        stored = e.get(6, [0])[0]
        if stored and _mask(crc32c_bulk(raw)) != stored:      # BundleEntryProto.crc32c = masked crc32c of the tensor's bytes (0: not recorded)
            raise ValueError('checkpoint tensor %r: crc32c mismatch (data shard corrupt or misread)' % key)
        shape = _shape_of(e.get(2, [b''])[0])
        if dt == _DT_BFLOAT16:
            arr = (raw.view('<u2').astype(np.uint32) << 16).view(np.float32)
        elif dt in _DTYPES:
            arr = raw.view(_DTYPES[dt]).copy()
        else:
            raise NotImplementedError('checkpoint tensor %r: dtype enum %d' % (key, dt))
        out[key.decode()] = arr.reshape(shape)
    return out


# ---------------------------------------------------------------------------------------------- writing (round trips; handing weights back)
def _block(entries, restart_interval=16):
    out, restarts, prev = bytearray(), [], b''
    for n, (key, val) in enumerate(entries):
        shared = 0
        if n % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(val)) + key[shared:] + val
        prev = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_checkpoint(prefix, tensors, block_bytes=4096):
    """{key: array} -> `<prefix>.index` + `<prefix>.data-00000-of-00001` (float32 / float64 / int32 / int64 tensors)."""
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    keys = sorted(tensors)
    entries = [(b'', _field(1, 0, _put_varint(1)) + _field(3, 2, b'\x02\x08\x01'))]     # num_shards = 1, version {producer = 1}
    offset = 0
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        for k in keys:
            a = np.asarray(tensors[k], order='C')
            a = a.astype(a.dtype.newbyteorder('<'), copy=False)
            if a.dtype not in _DT_OF:
                raise NotImplementedError('write_checkpoint: dtype %s' % a.dtype)
            raw = a.tobytes()
            f.write(raw)
            # masked crc32c of the tensor bytes: TensorFlow's BundleReader::GetValue compares it and returns DataLoss on a mismatch
            entries.append((k.encode(), _entry_proto(_DT_OF[a.dtype], a.shape, 0, offset, len(raw), _mask(crc32c_bulk(raw)))))
            offset += len(raw)
    with open(prefix + '.index', 'wb') as f:
        pos, index, cur, cur_bytes = 0, [], [], 0

        def flush():
            nonlocal pos, cur, cur_bytes
            if not cur:
                return
            body = _block(cur)
            f.write(body + b'\x00' + struct.pack('<I', _mask(crc32c(body + b'\x00'))))
            index.append((cur[-1][0], _put_varint(pos) + _put_varint(len(body))))
            pos += len(body) + 5
            cur, cur_bytes = [], 0
        for kv in entries:
            cur.append(kv)
            cur_bytes += len(kv[0]) + len(kv[1])
            if cur_bytes >= block_bytes:
                flush()
        flush()
        meta = _block([])
        f.write(meta + b'\x00' + struct.pack('<I', _mask(crc32c(meta + b'\x00'))))
        meta_handle = _put_varint(pos) + _put_varint(len(meta))
        pos += len(meta) + 5
        ib = _block(index, restart_interval=1)
        f.write(ib + b'\x00' + struct.pack('<I', _mask(crc32c(ib + b'\x00'))))
        foot = meta_handle + _put_varint(pos) + _put_varint(len(ib))
        f.write(foot + b'\x00' * (40 - len(foot)) + struct.pack('<Q', MAGIC))


# ---------------------------------------------------------------------------------------------- Keras object-graph keys <-> this repo's names
_SUFFIX = '/.ATTRIBUTES/VARIABLE_VALUE'


def keras_object_graph_keys(own_names):
    """[(checkpoint key, own parameter name)] for a model whose parameters are `own_names` in creation order
    ('conv0/kernel', 'bn0/gamma', ...): layer_with_weights-<i> counts the layers in that order."""
    out, layer_idx = [], {}
    for name in own_names:
        layer, leaf = name.rsplit('/', 1)
        if layer not in layer_idx:
            layer_idx[layer] = len(layer_idx)
        out.append(('layer_with_weights-%d/%s%s' % (layer_idx[layer], leaf, _SUFFIX), name))
    return out


def load_keras_checkpoint(prefix, own_shapes):
    """TF checkpoint written by `keras_model.save_weights(prefix)` -> {own name: float32 array}, checked against `own_shapes`
    ({own name: shape} in creation order).  Optimizer slots and the object graph are ignored."""
    tensors = read_checkpoint(prefix)
    params = {}
    for key, name in keras_object_graph_keys(list(own_shapes)):
        if key not in tensors:
            raise ValueError('%s.index lacks %s (for %s)' % (prefix, key, name))
        a = np.asarray(tensors[key], dtype=np.float32)
        if tuple(a.shape) != tuple(own_shapes[name]):
            raise ValueError('%s: checkpoint shape %s, model shape %s' % (name, tuple(a.shape), tuple(own_shapes[name])))
        params[name] = a
    return params


def save_keras_checkpoint(prefix, params_in_creation_order):
    """The inverse: {own name: array} (creation order) -> a TF checkpoint with Keras' object-graph keys, loadable by
    `keras_model.load_weights(prefix)` on a TensorFlow box as far as the tensor entries go (every entry carries the masked crc32c of
    its bytes, which BundleReader checks; UNTESTED against TensorFlow itself, which is not installable here; no object-graph proto is written;
    Keras falls back to name-based matching of the `layer_with_weights-*` keys only when the graph is present, so prefer the
    `.npz` route of INTEGRATION.md section C for that direction)."""
    keys = dict((name, key) for key, name in keras_object_graph_keys(list(params_in_creation_order)))
    write_checkpoint(prefix, {keys[n]: np.asarray(v, dtype=np.float32) for n, v in params_in_creation_order.items()})
