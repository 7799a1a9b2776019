"""TensorFlow V2 checkpoint files without TensorFlow (argsim_amd/tf_bundle.py): known answers of the CRC-32C
and LevelDB-table layer, round trips and corruption detection.  No TF process exists here to exchange files
with and the reference ships no checkpoint: cross-implementation parity is unpinned (stated in the module)."""
import os
import struct

import numpy as np
import pytest

from argsim_amd import tf_bundle as tb


def test_crc32c_known_answers():
    # RFC 3720 appendix B.4 / the LevelDB crc32c_test vectors
    assert tb.crc32c(b'123456789') == 0xE3069283
    assert tb.crc32c(bytes(32)) == 0x8A9136AA
    assert tb.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tb.crc32c(bytes(range(32))) == 0x46DD794E
    assert tb.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    # extension and the masked form of leveldb: Mask(crc("foo")) round trips and differs from the crc
    c = tb.crc32c(b'foo')
    assert tb.crc32c(b'hello world') == tb.crc32c(b' world', tb.crc32c(b'hello'))
    assert tb.unmask_crc(tb.mask_crc(c)) == c and tb.mask_crc(c) != c


def test_crc32c_chunked_path_equals_bytewise():
    rng = np.random.default_rng(0)
    for n in (0, 1, 4095, 4096, 8 * 4096, 8 * 4096 + 17, 300_001):
        data = rng.integers(0, 256, n, dtype=np.uint8)
        ref = (~tb._crc_bytes(0xFFFFFFFF, data.tobytes())) & 0xFFFFFFFF
        assert tb.crc32c(data) == ref, n
        assert tb.crc32c(data.tobytes()) == ref, n


def test_table_roundtrip_prefix_compression_and_many_blocks(tmp_path):
    keys = sorted({('scope%d/layer%d/%s' % (i % 7, i % 13, n)).encode() for i in range(200) for n in ('kernel', 'bias', 'kernel/Adam')})
    items = [(b'', b'header')] + [(k, k[::-1] * 3) for k in keys]
    p = str(tmp_path / 't.index')
    tb.write_table(p, items, block_size=512)            # forces dozens of data blocks
    assert tb.read_table(p) == items
    raw = open(p, 'rb').read()
    assert struct.unpack('<Q', raw[-8:])[0] == 0xDB4775248B80FB57 and len(raw) > 48
    tb.write_table(p, items)                            # one block
    assert tb.read_table(p) == items
    # a flipped byte inside a block is caught by the block checksum
    bad = bytearray(open(p, 'rb').read()); bad[10] ^= 0x40
    open(p, 'wb').write(bytes(bad))
    with pytest.raises(ValueError):
        tb.read_table(p)


def test_bundle_roundtrip_dtypes_shapes_and_corruption(tmp_path):
    rng = np.random.default_rng(1)
    tensors = {
        'global_step': np.asarray(12345, np.int64),
        'embed/embedding': rng.standard_normal((300, 64)).astype(np.float32),
        'latent/mu/kernel': rng.standard_normal((128, 32)).astype(np.float32),
        'latent/mu/bias': rng.standard_normal(32).astype(np.float32),
        'train/beta1_power': np.asarray(0.9 ** 7, np.float32),
        'misc/ids': rng.integers(0, 8192, (5, 7)).astype(np.int32),
        'misc/empty': np.zeros((0, 4), np.float32),
        'misc/big': rng.standard_normal(200_000).astype(np.float32),       # exercises the chunked CRC path
    }
    prefix = str(tmp_path / 'ckpt' / 'trial0')
    tb.write_bundle(prefix, tensors)
    assert os.path.exists(prefix + '.index') and os.path.exists(prefix + '.data-00000-of-00001')
    assert tb.latest_checkpoint(str(tmp_path / 'ckpt')) == prefix
    got = tb.read_bundle(prefix)
    assert sorted(got) == sorted(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    only = tb.read_bundle(prefix, names={'latent/mu/bias'})
    assert list(only) == ['latent/mu/bias']
    # the data file is the tensors in sorted-name order, nothing else
    assert os.path.getsize(prefix + '.data-00000-of-00001') == sum(v.nbytes for v in tensors.values())
    # a flipped tensor byte is caught by the per-tensor checksum
    data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read()); data[100] ^= 1
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
    with pytest.raises(ValueError):
        tb.read_bundle(prefix)


def test_entry_proto_wire_format():
    # BundleEntryProto{dtype: DT_FLOAT, shape{dim{size:3} dim{size:5}}, offset: 60, size: 60, crc32c: 0x01020304}
    e = tb._encode_entry(np.float32, (3, 5), 0, 60, 60, 0x01020304)
    assert e == bytes([0x08, 0x01, 0x12, 0x08, 0x12, 0x02, 0x08, 0x03, 0x12, 0x02, 0x08, 0x05, 0x20, 60, 0x28, 60, 0x35, 4, 3, 2, 1])
    d = tb._decode_entry(e)
    assert (d['dtype'], d['shape'], d['offset'], d['size'], d['crc']) == (1, [3, 5], 60, 60, 0x01020304)
    assert tb._encode_header(1) == bytes([0x08, 0x01, 0x1A, 0x02, 0x08, 0x01])


def test_cudnn_canonical_and_opaque_conversions_roundtrip():
    from argsim_amd import ckpt
    rng = np.random.default_rng(2)
    D = 16
    names, sd, shapes = [], {}, {}
    for base, In in (('encode/rnn1/fwd/', D), ('encode/rnn1/bwd/', D), ('encode/rnn2/fwd/', 2 * D), ('decode/rnn/l1/', D), ('decode/rnn/l2/', D)):
        for n, shp in (('W', (3 * D, In)), ('R', (3 * D, D)), ('bW', (3 * D,)), ('bR', (3 * D,))):
            names.append(base + n); shapes[base + n] = shp
            sd[base + n] = rng.standard_normal(shp).astype(np.float32)
    sd['embed/embedding'] = rng.standard_normal((32, D)).astype(np.float32); names.append('embed/embedding')
    tf = ckpt.to_tf_names(sd)
    assert tf['decode/rnn/cudnn_gru/rnn/multi_rnn_cell/cell_1/cudnn_compatible_gru_cell/gates/kernel'].shape == (2 * D, 2 * D)
    back = ckpt.from_tf_names(tf, names)
    for k in names:
        if k.endswith('/bW') or k.endswith('/bR'):
            continue
        assert np.array_equal(back[k], sd[k]), k
    for base in ('encode/rnn1/fwd/', 'decode/rnn/l2/'):            # r/u biases come back summed into bW: same function
        assert np.allclose(back[base + 'bW'][:2 * D] + back[base + 'bR'][:2 * D], sd[base + 'bW'][:2 * D] + sd[base + 'bR'][:2 * D])
        assert np.array_equal(back[base + 'bW'][2 * D:], sd[base + 'bW'][2 * D:]) and np.array_equal(back[base + 'bR'][2 * D:], sd[base + 'bR'][2 * D:])
    groups = ckpt._opaque_groups(names)
    assert groups['decode/rnn'] == ['decode/rnn/l1/', 'decode/rnn/l2/'] and groups['encode/rnn2/fwd'] == ['encode/rnn2/fwd/']
    for scope, bases in groups.items():
        flat = ckpt._to_opaque(lambda n: sd[n], bases)
        assert flat.size == sum(sd[b + n].size for b in bases for n in ('W', 'R', 'bW', 'bR'))
        rec = ckpt._from_opaque(flat, bases, shapes)
        for b in bases:
            for n in ('W', 'R', 'bW', 'bR'):
                assert np.array_equal(rec[b + n], sd[b + n]), b + n
