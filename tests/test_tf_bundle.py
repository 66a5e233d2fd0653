"""TensorFlow V2 checkpoint (tensor bundle) reader / writer: known-answer vectors of the formats it is made
of, round trips, corruption detection, and the model-level save / restore with TF variable names.
(TensorFlow itself is not available: see the module docstring -- parity unpinned against a TF-written file.)"""
import os
import struct

import numpy as np
import pytest

from ml_super_resolution_amd import tf_bundle as tb


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors + the classic check value
    assert tb.crc32c(b'\x00' * 32) == 0x8A9136AA
    assert tb.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert tb.crc32c(bytes(range(32))) == 0x46DD794E
    assert tb.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert tb.crc32c(b'123456789') == 0xE3069283
    # the array variant (slicing-by-8) agrees with the byte loop on ragged lengths
    rng = np.random.default_rng(0)
    for n in (0, 1, 7, 8, 9, 63, 64, 1000, 4099):
        a = rng.integers(0, 256, n, dtype=np.uint8)
        assert tb.crc32c_array(a) == tb.crc32c(a.tobytes())
    # masking is a bijection and matches TensorFlow's definition on a worked example
    crc = tb.crc32c(b'foo')
    assert tb.unmask_crc(tb.mask_crc(crc)) == crc
    assert tb.mask_crc(crc) != crc
    assert tb.mask_crc(0) == 0xa282ead8


def test_varint_and_protobuf_helpers():
    for n in (0, 1, 127, 128, 300, 2 ** 32 - 1, 2 ** 63 + 5):
        enc = tb._put_varint(n)
        assert tb._get_varint(enc, 0) == (n, len(enc))
    assert tb._put_varint(300) == b'\xac\x02'                      # protobuf docs example
    msg = tb._pb_varint_field(1, 150) + tb._pb_bytes_field(2, b'testing') + tb._pb_fixed32_field(6, 0xdeadbeef)
    assert msg[:3] == b'\x08\x96\x01'                              # protobuf docs: field 1 = 150
    assert list(tb._pb_fields(msg)) == [(1, 0, 150), (2, 2, b'testing'), (6, 5, 0xdeadbeef)]
    assert tb._decode_shape(tb._encode_shape((3, 3, 64, 64))) == (3, 3, 64, 64)
    assert tb._decode_shape(tb._encode_shape(())) == ()


def test_snappy_decompress_hand_assembled_stream():
    # "abcdabcdabcdabcdXYZ": literal "abcd", copy(offset 4, len 12) as 1-byte-offset copy (len 4..11) + 2-byte copy, literal "XYZ"
    want = b'abcd' * 4 + b'XYZ'
    stream = bytes([len(want)])                                    # preamble: uncompressed length (varint)
    stream += bytes([(4 - 1) << 2]) + b'abcd'                      # literal, length 4
    stream += bytes([((8 - 4) << 2) | 1, 4])                       # copy 1-byte offset: length 8, offset 4 (overlapping)
    stream += bytes([((4 - 1) << 2) | 2, 4, 0])                    # copy 2-byte offset: length 4, offset 4
    stream += bytes([(3 - 1) << 2]) + b'XYZ'
    assert tb.snappy_decompress(stream) == want
    long_lit = bytes(range(256)) * 2                               # literal with a 2-byte length (tag 61)
    stream = tb._put_varint(len(long_lit)) + bytes([61 << 2]) + struct.pack('<H', len(long_lit) - 1) + long_lit
    assert tb.snappy_decompress(stream) == long_lit
    with pytest.raises(ValueError):
        tb.snappy_decompress(bytes([4, ((4 - 1) << 2) | 2, 9, 0]))   # copy before any output


def test_block_parser_on_hand_assembled_block():
    # two entries with prefix compression, one restart point at 0 (table_format.txt)
    body = b''
    body += bytes([0, 5, 2]) + b'apple' + b'v1'
    body += bytes([3, 3, 2]) + b'ly!' + b'v2'                      # key = "app" + "ly!"
    block = body + struct.pack('<I', 0) + struct.pack('<I', 1)
    assert list(tb._block_entries(block)) == [(b'apple', b'v1'), (b'apply!', b'v2')]


def test_table_round_trip_many_blocks(tmp_path):
    rng = np.random.default_rng(1)
    items = [(b'', b'header')]
    for i in range(700):
        items.append((('conv2d_%03d/kernel/part-%d' % (i // 3, i)).encode(), rng.bytes(int(rng.integers(0, 90)))))
    items.sort()
    path = str(tmp_path / 't.index')
    tb.write_table(path, items, block_size=512)
    got = tb.read_table(path)
    assert got == dict(items)
    assert os.path.getsize(path) > 48
    # snappy-compressed blocks are accepted: rewrite the file's first data block as a stored-literal snappy stream
    with open(path, 'rb') as f:
        raw = bytearray(f.read())
    with pytest.raises(ValueError):
        tb.write_table(path, [(b'b', b''), (b'a', b'')])           # keys must increase
    # corruption is detected
    raw[10] ^= 0x40
    bad = str(tmp_path / 'bad.index')
    open(bad, 'wb').write(raw)
    with pytest.raises(ValueError):
        tb.read_table(bad)
    assert tb.read_table(bad, verify=False) is not None


def test_bundle_round_trip_and_checks(tmp_path):
    rng = np.random.default_rng(2)
    tensors = {
        'conv2d/kernel': rng.normal(size=(3, 3, 3, 64)).astype(np.float32),
        'conv2d/bias': np.zeros(64, np.float32),
        'conv2d_19/kernel': rng.normal(size=(3, 3, 64, 3)).astype(np.float32),
        'conv2d/kernel/Adam': rng.normal(size=(3, 3, 3, 64)).astype(np.float32),
        'global_step': np.asarray(25600, np.int64),
        'beta1_power': np.asarray(0.9 ** 7, np.float32),
        'flags': np.array([[1, 0], [0, 1]], np.uint8),
    }
    prefix = str(tmp_path / 'ckpt' / 'model.ckpt-25600')
    tb.save_checkpoint(prefix, tensors)
    assert os.path.exists(prefix + '.index') and os.path.exists(prefix + '.data-00000-of-00001')
    assert tb.is_checkpoint_prefix(prefix) and not tb.is_checkpoint_prefix(prefix + '.pt')
    got = tb.load_checkpoint(prefix)
    assert set(got) == set(tensors)
    for k in tensors:
        assert got[k].dtype == np.asarray(tensors[k]).dtype and got[k].shape == np.asarray(tensors[k]).shape
        assert np.array_equal(got[k], tensors[k])
    listed = {n: (s, d) for n, s, d in tb.list_variables(prefix)}
    assert listed['conv2d_19/kernel'] == ((3, 3, 64, 3), np.dtype(np.float32))
    assert listed['global_step'] == ((), np.dtype(np.int64))
    assert set(tb.load_checkpoint(prefix, names={'conv2d/bias'})) == {'conv2d/bias'}
    # the header says one little-endian shard; data offsets tile the data file exactly
    table = tb.read_table(prefix + '.index')
    assert dict((f, v) for f, _, v in tb._pb_fields(table[b'']))[1] == 1
    total = sum(tb._decode_entry(v)['size'] for k, v in table.items() if k)
    assert total == os.path.getsize(prefix + '.data-00000-of-00001')
    # a flipped data byte fails the per-tensor CRC
    with open(prefix + '.data-00000-of-00001', 'r+b') as f:
        f.seek(5)
        b = f.read(1)
        f.seek(5)
        f.write(bytes([b[0] ^ 1]))
    with pytest.raises(ValueError):
        tb.load_checkpoint(prefix)
    assert tb.load_checkpoint(prefix, verify=False) is not None


def test_model_level_save_restore_with_tf_names(tmp_path):
    torch = pytest.importorskip('torch')
    from ml_super_resolution_amd import engine
    specs = [engine.LayerSpec(3, 3, 8, act='relu', scope='conv2d'), engine.LayerSpec(3, 8, 3, scope='conv2d_1')]
    a = engine.ConvStack(specs, device='cpu', residual=True, weight_decay=1e-4)
    g = torch.Generator().manual_seed(0)
    a.params.copy_(torch.randn(a.params.shape, generator=g))
    a.opt_m = torch.randn(a.params.shape, generator=g)
    a.opt_v = torch.rand(a.params.shape, generator=g)
    a.global_step = 1234
    prefix = str(tmp_path / 'model.ckpt-1234')
    a.save_tf_checkpoint(prefix)
    names = {n for n, _, _ in tb.list_variables(prefix)}
    assert {'conv2d/kernel', 'conv2d/bias', 'conv2d_1/kernel', 'conv2d_1/bias', 'global_step', 'beta1_power', 'beta2_power',
            'conv2d/kernel/Adam', 'conv2d/kernel/Adam_1', 'conv2d_1/bias/Adam_1'} <= names
    b = engine.ConvStack(specs, device='cpu', residual=True, weight_decay=1e-4)
    b.load_checkpoint(prefix)
    assert b.global_step == 1234
    for i in range(2):
        assert torch.equal(a.kernel(i), b.kernel(i)) and torch.equal(a.bias(i), b.bias(i))
        assert torch.equal(a.kernel(i, a.opt_m), b.kernel(i, b.opt_m)) and torch.equal(a.bias(i, a.opt_v), b.bias(i, b.opt_v))
    # ESPCN's extract_weights reads the same format (espcn/espcn/model_espcn.py:150-166)
    from ml_super_resolution_amd.espcn import model_espcn
    tb.save_checkpoint(str(tmp_path / 'espcn'), {'f1/kernel': np.ones((5, 5, 3, 64), np.float32), 'f1/bias': np.zeros(64, np.float32),
                                                'f1/kernel/Adam': np.zeros((5, 5, 3, 64), np.float32), 'global_step': np.asarray(3, np.int64)})
    w = model_espcn.extract_weights(None, str(tmp_path / 'espcn'))
    assert set(w) == {'f1/kernel:0', 'f1/bias:0'}
    with pytest.raises(KeyError):
        c = engine.ConvStack([engine.LayerSpec(3, 3, 8, scope='other')], device='cpu')
        c.load_tf_checkpoint(prefix)


def test_snappy_compressor_round_trips_and_uses_every_element_kind():
    """snappy_compress -> snappy_decompress on inputs that force literals of every length class, 1- and 2-byte-offset
    copies and overlapping copies (a run of one byte is a copy of offset 1 onto its own output)."""
    rng = np.random.default_rng(0)
    cases = [b'', b'a', b'abcd' * 3, b'x' * 1000, bytes(rng.integers(0, 256, 5000, dtype=np.uint8)),
             bytes(rng.integers(0, 256, 200, dtype=np.uint8)),
             b''.join(b'conv2d_%d/kernel/Adam_1' % i for i in range(40)),
             bytes(rng.integers(0, 4, 70000, dtype=np.uint8)),                       # offsets beyond 2048, long matches
             bytes(rng.integers(0, 256, 300, dtype=np.uint8)) * 20]
    kinds = set()
    for data in cases:
        packed = tb.snappy_compress(data)
        assert tb.snappy_decompress(packed) == data
        pos = len(tb._put_varint(len(data)))
        while pos < len(packed):                       # walk the element stream: which tags occur?
            tag = packed[pos]
            kind = tag & 3
            if kind == 0:
                ln = tag >> 2
                extra = ln - 59 if ln >= 60 else 0
                ln = (int.from_bytes(packed[pos + 1:pos + 1 + extra], 'little') if extra else ln) + 1
                kinds.add('literal%d' % extra)
                pos += 1 + extra + ln
            else:
                kinds.add('copy%d' % kind)
                pos += 2 if kind == 1 else 3
    assert {'literal0', 'literal1', 'literal2', 'copy1', 'copy2'} <= kinds
    assert len(tb.snappy_compress(b'x' * 1000)) < 60


def test_multi_shard_bundle_with_compressed_index(tmp_path):
    """What a sharded tf.train.Saver writes: tensors spread over several data files (BundleEntryProto.shard_id), and a
    Snappy-compressed .index (LevelDB's table builder compresses blocks that shrink by 1/8 or more)."""
    rng = np.random.default_rng(3)
    tensors = {}
    for i in range(20):
        scope = 'conv2d' if i == 0 else 'conv2d_%d' % i
        tensors[scope + '/kernel'] = rng.normal(size=(3, 3, 4, 8)).astype(np.float32)
        tensors[scope + '/bias'] = rng.normal(size=(8,)).astype(np.float32)
        tensors[scope + '/kernel/Adam'] = rng.normal(size=(3, 3, 4, 8)).astype(np.float32)
        tensors[scope + '/kernel/Adam_1'] = rng.normal(size=(3, 3, 4, 8)).astype(np.float32)
    tensors['global_step'] = np.asarray(7, np.int64)
    prefix = str(tmp_path / 'model.ckpt-7')
    tb.save_checkpoint(prefix, tensors, num_shards=3, compress_index=True)
    assert sorted(os.listdir(str(tmp_path))) == ['model.ckpt-7.data-00000-of-00003', 'model.ckpt-7.data-00001-of-00003',
                                                 'model.ckpt-7.data-00002-of-00003', 'model.ckpt-7.index']
    # the index really holds compressed blocks (type byte 1 after a block body) and is smaller than the raw one
    tb.save_checkpoint(str(tmp_path / 'raw' / 'model.ckpt-7'), tensors, num_shards=3)
    assert os.path.getsize(prefix + '.index') < 0.8 * os.path.getsize(str(tmp_path / 'raw' / 'model.ckpt-7.index'))
    got = tb.load_checkpoint(prefix)
    assert set(got) == set(tensors)
    for k in tensors:
        assert np.array_equal(got[k], tensors[k]) and got[k].dtype == np.asarray(tensors[k]).dtype
    table = tb.read_table(prefix + '.index')
    assert dict((f, v) for f, _, v in tb._pb_fields(table[b'']))[1] == 3
    assert {tb._decode_entry(v)['shard_id'] for k, v in table.items() if k} == {0, 1, 2}
    # a missing shard is an error, not silent zeros
    os.remove(prefix + '.data-00001-of-00003')
    with pytest.raises(OSError):
        tb.load_checkpoint(prefix)


def test_unsupported_bundle_features_are_refused_loudly(tmp_path):
    """Big-endian bundles and partitioned (sliced) variables -- neither can come from the reference's single-device
    graphs -- raise instead of returning wrong data."""
    prefix = str(tmp_path / 'be')
    entry = (tb._pb_varint_field(1, 1) + tb._pb_bytes_field(2, tb._encode_shape((2,))) + tb._pb_varint_field(5, 8) +
             tb._pb_fixed32_field(6, 0))
    header_be = tb._pb_varint_field(1, 1) + tb._pb_varint_field(2, 1)
    tb.write_table(prefix + '.index', [(b'', header_be), (b'v', entry)])
    open(prefix + '.data-00000-of-00001', 'wb').write(b'\\0' * 8)
    with pytest.raises(ValueError, match='big-endian'):
        tb.load_checkpoint(prefix)
    prefix = str(tmp_path / 'sliced')
    sliced = entry + tb._pb_bytes_field(7, b'')
    tb.write_table(prefix + '.index', [(b'', tb._pb_varint_field(1, 1)), (b'v', sliced)])
    open(prefix + '.data-00000-of-00001', 'wb').write(b'\\0' * 8)
    with pytest.raises(ValueError, match='partitioned'):
        tb.load_checkpoint(prefix)


def test_saver_housekeeping_atomic_writes_max_to_keep_and_scan_fallback(tmp_path):
    """tf.train.Saver() keeps 5 checkpoints and tf.train.latest_checkpoint reads the state file
    (enet/enet/experiment_train.py:96-117).  A bundle is complete once its `.index` exists (temporary names until
    then); without a usable state file the newest complete `model.ckpt-N` is found instead of restarting at step 0."""
    from ml_super_resolution_amd import tf_bundle as B
    d = tmp_path / 'ckpt'
    d.mkdir()
    for n in range(999, 8999, 1000):
        prefix = str(d / ('model.ckpt-%d' % n))
        B.save_checkpoint(prefix, {'global_step': np.asarray(n, np.int64), 'w': np.full((3,), n, np.float32)})
        B.update_checkpoint_state(prefix)
    names = sorted(os.listdir(str(d)))
    assert not [n for n in names if n.endswith('.tmp')]
    kept = [3999, 4999, 5999, 6999, 7999]
    assert sorted(n for n in names if n.endswith('.index')) == ['model.ckpt-%d.index' % n for n in kept]
    assert len([n for n in names if '.data-' in n]) == 5
    text = open(str(d / 'checkpoint')).read().splitlines()
    assert text == ['model_checkpoint_path: "model.ckpt-7999"'] + ['all_model_checkpoint_paths: "model.ckpt-%d"' % n for n in kept]
    assert B.latest_checkpoint(str(d)) == str(d / 'model.ckpt-7999')
    # the state file is gone (a copied directory): plain lookup finds nothing, the scan finds the newest bundle
    os.remove(str(d / 'checkpoint'))
    assert B.latest_checkpoint(str(d)) is None
    assert B.latest_checkpoint(str(d), scan=True) == str(d / 'model.ckpt-7999')
    # a save that was killed half way: data written, index still under its temporary name -> not a checkpoint
    open(str(d / 'model.ckpt-8999.data-00000-of-00001'), 'wb').write(b'x' * 20)
    open(str(d / 'model.ckpt-8999.index.tmp'), 'wb').write(b'y')
    assert B.latest_checkpoint(str(d), scan=True) == str(d / 'model.ckpt-7999')
    # a state file that names a prefix which is not there
    open(str(d / 'checkpoint'), 'w').write('model_checkpoint_path: "model.ckpt-8999"\n')
    assert B.latest_checkpoint(str(d)) is None
    assert B.latest_checkpoint(str(d), scan=True) == str(d / 'model.ckpt-7999')
    assert int(B.load_checkpoint(B.latest_checkpoint(str(d), scan=True))['global_step']) == 7999
    assert B.newest_bundle(str(tmp_path / 'nothing')) is None


def test_saver_housekeeping_never_deletes_checkpoints_it_did_not_write(tmp_path):
    """tf.train.Saver only deletes checkpoints the same Saver instance wrote.  A directory that already holds checkpoints
    (an earlier run's, TensorFlow's own -- listed in the state file, possibly by absolute path) is resumed in and saved
    to more than max_to_keep times: the older files stay on disk and stay listed (round-3 advisor finding)."""
    from ml_super_resolution_amd import tf_bundle as B
    d = tmp_path / 'ckpt'
    d.mkdir()
    elsewhere = tmp_path / 'elsewhere'
    elsewhere.mkdir()
    t = {'w': np.zeros((2,), np.float32)}
    for n in (100, 200, 300):
        B.save_checkpoint(str(d / ('model.ckpt-%d' % n)), t)
    B.save_checkpoint(str(elsewhere / 'model.ckpt-7'), t)
    open(str(d / 'checkpoint'), 'w').write(
        'model_checkpoint_path: "model.ckpt-300"\n' + 'all_model_checkpoint_paths: "%s"\n' % str(elsewhere / 'model.ckpt-7') +
        ''.join('all_model_checkpoint_paths: "model.ckpt-%d"\n' % n for n in (100, 200, 300)) +
        'all_model_checkpoint_paths: "model.ckpt-gone"\n')
    B._saved_by_this_process.pop(str(d), None)
    for n in range(1000, 9000, 1000):
        prefix = str(d / ('model.ckpt-%d' % n))
        B.save_checkpoint(prefix, t)
        B.update_checkpoint_state(prefix)
    for n in (100, 200, 300):
        assert B.is_checkpoint_prefix(str(d / ('model.ckpt-%d' % n)))
    assert B.is_checkpoint_prefix(str(elsewhere / 'model.ckpt-7'))
    mine = [4000, 5000, 6000, 7000, 8000]
    assert sorted(int(n[len('model.ckpt-'):-len('.index')]) for n in os.listdir(str(d)) if n.endswith('.index')) == [100, 200, 300] + mine
    text = open(str(d / 'checkpoint')).read().splitlines()
    assert text[0] == 'model_checkpoint_path: "model.ckpt-8000"'
    assert text[1:] == (['all_model_checkpoint_paths: "%s"' % str(elsewhere / 'model.ckpt-7')] +
                        ['all_model_checkpoint_paths: "model.ckpt-%d"' % n for n in [100, 200, 300] + mine])   # (the dangling entry is dropped)
    # overwriting an existing prefix: the result is a complete, readable bundle
    B.save_checkpoint(str(d / 'model.ckpt-8000'), {'w': np.ones((2,), np.float32)})
    assert float(B.load_checkpoint(str(d / 'model.ckpt-8000'))['w'][0]) == 1.0


def test_tf_beta_power_uses_float32_base():
    from ml_super_resolution_amd import tf_bundle as B
    b, p = np.float32(0.999), np.float32(0.999)
    for _ in range(50000):
        p = np.float32(p * b)
    got = B.tf_beta_power(0.999, 50000)
    assert got.dtype == np.float32 and abs(float(got) / float(p) - 1.0) < 1e-4          # the float32 product's random walk
    assert abs(0.999 ** 50001 / float(p) - 1.0) > 3e-4                                  # the Python-float base is a step's factor off by now
    assert float(B.tf_beta_power(0.9, 0)) == float(np.float32(0.9))
