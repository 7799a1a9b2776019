"""TensorFlow checkpoint "tensor bundle" (V2) files without TensorFlow: reader and writer.

What ``tf.train.Saver().save(sess, prefix, write_meta_graph=False)`` leaves on disk
(reference src/train.py:121) and ``saver.restore`` reads (src/train.py:93-94,
src/eval_embed_reason.py:24-27) is

    <prefix>.index                   an SSTable (LevelDB table format) mapping variable name ->
                                     serialized BundleEntryProto, plus the key "" -> BundleHeaderProto
    <prefix>.data-00000-of-00001     the raw little-endian tensor bytes, concatenated
    checkpoint                       a text CheckpointState naming the latest prefix

This module restates those formats from their published definitions (tensorflow/core/util/
tensor_bundle/tensor_bundle.cc, tensorflow/core/protobuf/tensor_bundle.proto, tensorflow/core/lib/io/
format.cc + block_builder.cc, i.e. the LevelDB table format: prefix-compressed blocks with restart
points, 5-byte block trailers with a masked CRC-32C, 48-byte footer with the magic 0xdb4775248b80fb57)
so that checkpoints of the TF1 reference load into, and are written by, this package with no TensorFlow
in the loop.

PARITY UNPINNED: no TensorFlow exists in the build container or on the GPU box and the reference ships
no checkpoint (``/trial`` is git-ignored there), so these files have not been exchanged with a real TF
process.  What the tests pin: the CRC-32C and masking known answers of the LevelDB/TF sources, the
footer magic, byte-exact round trips through this reader (blocks with prefix compression, several
blocks, several dtypes) and detection of corrupted blocks and tensors.
"""
import os
import struct

import numpy as np

# ---------------------------------------------------------------------------------------- CRC-32C
_POLY = 0x82F63B78                      # Castagnoli, reflected


def _make_table():
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (_POLY if c & 1 else 0)
        t[i] = c
    return t


_T = _make_table()
_TL = [int(x) for x in _T]


def _crc_bytes(crc, data):
    for b in data:
        crc = _TL[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc


def _zero_op_tables(nbytes):
    """four 256-entry tables applying 'advance the raw CRC register over nbytes zero bytes' to a 32-bit state"""
    cols = []
    for bit in range(32):                 # the operator is linear over GF(2): image of every basis vector
        c = 1 << bit
        c = _crc_bytes(c, bytes(nbytes)) if nbytes <= 4096 else None
        cols.append(c)
    tabs = []
    for byte in range(4):
        t = []
        for v in range(256):
            x = 0
            for k in range(8):
                if v >> k & 1:
                    x ^= cols[8 * byte + k]
            t.append(x)
        tabs.append(t)
    return tabs


_CHUNK = 4096
_ZOP = None


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli) of ``data`` (bytes-like), continuing from ``crc``.  Long inputs are cut into
    4 KiB chunks whose registers advance together in numpy (one table lookup per byte position for all
    chunks), then folded left to right with the 'append 4096 zero bytes' operator."""
    global _ZOP
    buf = np.frombuffer(memoryview(data).cast('B'), np.uint8) if not isinstance(data, np.ndarray) else data.view(np.uint8).reshape(-1)
    n = buf.size
    reg = (~crc) & 0xFFFFFFFF
    nfull = n // _CHUNK
    if nfull >= 8:
        body = np.ascontiguousarray(buf[:nfull * _CHUNK].reshape(nfull, _CHUNK).T)     # [byte position][chunk]
        r = np.zeros(nfull, np.uint32)
        for i in range(_CHUNK):           # raw registers of every chunk from a ZERO start
            r = _T[(r ^ body[i]) & 0xFF] ^ (r >> np.uint32(8))
        if _ZOP is None:
            _ZOP = _zero_op_tables(_CHUNK)
        z0, z1, z2, z3 = _ZOP
        for c in r.tolist():              # reg <- advance(reg over one chunk of zeros) xor raw(chunk)
            reg = z0[reg & 0xFF] ^ z1[(reg >> 8) & 0xFF] ^ z2[(reg >> 16) & 0xFF] ^ z3[reg >> 24] ^ c
        tail = buf[nfull * _CHUNK:]
    else:
        tail = buf
    reg = _crc_bytes(reg, tail.tobytes())
    return (~reg) & 0xFFFFFFFF


def mask_crc(c):
    """leveldb/tensorflow crc32c::Mask"""
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def unmask_crc(m):
    r = (m - 0xA282EAD8) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------- varints / protobuf
def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _read_varint(buf, pos):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _pb_fields(buf):
    """yields (field number, wire type, value) of one serialized message"""
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]; pos += 8
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            v = buf[pos:pos + n]; pos += n
        elif wt == 5:
            v = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
          17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DT_OF = {np.dtype(v): k for k, v in DTYPES.items()}


def _encode_entry(dtype, shape, shard, offset, size, crc_masked):
    dims = b''.join(b'\x12' + _varint(len(d)) + d for d in (b'\x08' + _varint(s) for s in shape))   # TensorShapeProto.dim = 2, Dim.size = 1
    out = b'\x08' + _varint(_DT_OF[np.dtype(dtype)])              # dtype = 1
    out += b'\x12' + _varint(len(dims)) + dims                     # shape = 2
    if shard:
        out += b'\x18' + _varint(shard)                            # shard_id = 3
    if offset:
        out += b'\x20' + _varint(offset)                           # offset = 4
    out += b'\x28' + _varint(size)                                 # size = 5
    out += b'\x35' + struct.pack('<I', crc_masked)                 # crc32c = 6 (fixed32)
    return out


def _decode_entry(buf):
    e = dict(dtype=0, shape=[], shard=0, offset=0, size=0, crc=None, slices=False)
    for f, wt, v in _pb_fields(buf):
        if f == 1: e['dtype'] = v
        elif f == 2:
            for f2, _, v2 in _pb_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _pb_fields(v2):
                        if f3 == 1: size = v3
                    e['shape'].append(size)
        elif f == 3: e['shard'] = v
        elif f == 4: e['offset'] = v
        elif f == 5: e['size'] = v
        elif f == 6: e['crc'] = struct.unpack('<I', v)[0]
        elif f == 7: e['slices'] = True
    return e


def _encode_header(num_shards):
    # BundleHeaderProto: num_shards = 1, endianness = 2 (LITTLE = 0, omitted), version = 3 {producer = 1}
    return b'\x08' + _varint(num_shards) + b'\x1a\x02\x08\x01'


# ---------------------------------------------------------------------------------------- LevelDB table
_MAGIC = 0xDB4775248B80FB57
_RESTART_INTERVAL = 16


class _BlockBuilder:
    def __init__(self):
        self.buf = bytearray(); self.restarts = [0]; self.count = 0; self.last = b''

    def add(self, key, value):
        shared = 0
        if self.count % _RESTART_INTERVAL == 0:
            if self.count:
                self.restarts.append(len(self.buf))
        else:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last = key; self.count += 1

    def size(self):
        return len(self.buf) + 4 * (len(self.restarts) + 1)

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def _write_block(f, contents):
    off = f.tell()
    f.write(contents)
    f.write(b'\x00' + struct.pack('<I', mask_crc(crc32c(contents + b'\x00'))))      # type 0 = no compression
    return off, len(contents)


def write_table(path, items, block_size=262144):
    """items: iterable of (key bytes, value bytes) in strictly increasing key order"""
    with open(path, 'wb') as f:
        index = _BlockBuilder(); blk = _BlockBuilder(); prev = None
        for k, v in items:
            assert prev is None or k > prev, "keys must be strictly increasing"
            prev = k
            blk.add(k, v)
            if blk.size() >= block_size:
                off, n = _write_block(f, blk.finish())
                index.add(blk.last, _varint(off) + _varint(n))
                blk = _BlockBuilder()
        if blk.count or not index.count:
            off, n = _write_block(f, blk.finish())
            index.add(blk.last, _varint(off) + _varint(n))
        moff, mn = _write_block(f, _BlockBuilder().finish())                 # empty metaindex block
        ioff, inn = _write_block(f, index.finish())
        footer = _varint(moff) + _varint(mn) + _varint(ioff) + _varint(inn)
        f.write(footer + bytes(40 - len(footer)) + struct.pack('<Q', _MAGIC))


def _read_block(buf, off, n, verify=True):
    contents, typ = buf[off:off + n], buf[off + n]
    if verify:
        want = unmask_crc(struct.unpack('<I', buf[off + n + 1:off + n + 5])[0])
        if crc32c(buf[off:off + n + 1]) != want:
            raise ValueError("table block at %d: checksum mismatch" % off)
    if typ != 0:
        raise NotImplementedError("compressed table block (type %d); tensor bundles are written uncompressed" % typ)
    return contents


def _block_entries(blk):
    nrest = struct.unpack('<I', blk[-4:])[0]
    end = len(blk) - 4 * (nrest + 1)
    pos, key = 0, b''
    while pos < end:
        shared, pos = _read_varint(blk, pos)
        non, pos = _read_varint(blk, pos)
        vl, pos = _read_varint(blk, pos)
        key = key[:shared] + bytes(blk[pos:pos + non]); pos += non
        yield key, bytes(blk[pos:pos + vl]); pos += vl


def read_table(path, verify=True):
    buf = open(path, 'rb').read()
    if len(buf) < 48 or struct.unpack('<Q', buf[-8:])[0] != _MAGIC:
        raise ValueError("%s is not a LevelDB table (bad magic)" % path)
    foot = buf[-48:]
    _, p = _read_varint(foot, 0); _, p = _read_varint(foot, p)
    ioff, p = _read_varint(foot, p); inn, p = _read_varint(foot, p)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, inn, verify)):
        off, q = _read_varint(handle, 0); n, q = _read_varint(handle, q)
        out.extend(_block_entries(_read_block(buf, off, n, verify)))
    return out


# ---------------------------------------------------------------------------------------- bundles
def _data_name(prefix, shard, nshards):
    return "%s.data-%05d-of-%05d" % (prefix, shard, nshards)


def write_bundle(prefix, tensors, checkpoint_state=True):
    """tensors: {variable name: array}.  Writes <prefix>.index and <prefix>.data-00000-of-00001 the way
    tf.train.Saver (BundleWriter) does: tensors in sorted name order, little-endian, one shard."""
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    items = [(b'', _encode_header(1))]
    off = 0
    with open(_data_name(prefix, 0, 1), 'wb') as f:
        for name in sorted(tensors, key=lambda s: s.encode()):
            shape = np.asarray(tensors[name]).shape          # (ascontiguousarray turns a scalar into shape (1,))
            a = np.ascontiguousarray(tensors[name])
            if a.dtype.byteorder == '>':
                a = a.astype(a.dtype.newbyteorder('<'))
            raw = a.view(np.uint8).reshape(-1) if a.size else np.zeros(0, np.uint8)
            f.write(raw.tobytes())
            items.append((name.encode(), _encode_entry(a.dtype, shape, 0, off, raw.size, mask_crc(crc32c(raw)))))
            off += raw.size
    write_table(prefix + '.index', items)
    if checkpoint_state:
        base = os.path.basename(prefix)
        with open(os.path.join(d or '.', 'checkpoint'), 'w') as f:
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))


def read_bundle(prefix, names=None, verify=True):
    """{variable name: array} of a TF V2 checkpoint; ``names``: only these (None = all)."""
    items = read_table(prefix + '.index', verify)
    if not items or items[0][0] != b'':
        raise ValueError("bundle index lacks the header entry")
    nshards, endian = 1, 0
    for f, _, v in _pb_fields(items[0][1]):
        if f == 1: nshards = v
        elif f == 2: endian = v
    if endian != 0:
        raise NotImplementedError("big-endian bundle")
    shards, out = {}, {}
    for k, v in items[1:]:
        name = k.decode()
        if names is not None and name not in names:
            continue
        e = _decode_entry(v)
        if e['slices']:
            raise NotImplementedError("partitioned variable %s" % name)
        if e['dtype'] not in DTYPES:
            raise NotImplementedError("dtype enum %d of %s" % (e['dtype'], name))
        if e['shard'] not in shards:
            shards[e['shard']] = np.memmap(_data_name(prefix, e['shard'], nshards), np.uint8, 'r')
        raw = np.asarray(shards[e['shard']][e['offset']:e['offset'] + e['size']])
        if verify and e['crc'] is not None and mask_crc(crc32c(raw)) != e['crc']:
            raise ValueError("tensor %s: checksum mismatch" % name)
        out[name] = raw.view(DTYPES[e['dtype']]).reshape(e['shape']).copy()
    return out


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the prefix named by <directory>/checkpoint, or None"""
    p = os.path.join(directory, 'checkpoint')
    if not os.path.exists(p):
        return None
    for line in open(p):
        if line.startswith('model_checkpoint_path:'):
            name = line.split(':', 1)[1].strip().strip('"')
            return name if os.path.isabs(name) else os.path.join(directory, name)
    return None
