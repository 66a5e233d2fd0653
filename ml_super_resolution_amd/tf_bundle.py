"""
tf_bundle.py -- reader / writer for TensorFlow "V2" checkpoints (tensor bundles), NumPy only.

What the reference does with checkpoints: `saver.restore(session, ckpt_path)` on a
`tf.train.Saver` checkpoint prefix -- vdsr/vdsr/experiment_resolve.py:53-62,
espcn/espcn/model_espcn.py:150-166 (`extract_weights`), enet/enet/experiment_resolve.py:37-58 --
and `tf.train.Saver().save(...)` while training (vdsr/vdsr/experiment_train.py:100-121).  A V2
checkpoint `<prefix>` is two files:

    <prefix>.index                  an SSTable (TensorFlow's copy of the LevelDB table format) that maps
                                    ""           -> BundleHeaderProto   {num_shards, endianness, version}
                                    tensor name  -> BundleEntryProto    {dtype, shape, shard_id, offset, size, crc32c}
    <prefix>.data-00000-of-00001    the raw little-endian tensor bytes at those offsets

TensorFlow is not installable here and the reference ships no checkpoint, so this module is a
restatement of the published format (tensorflow/core/util/tensor_bundle/, core/lib/io/table_format.txt,
core/protobuf/tensor_bundle.proto) -- **parity unpinned**: it is tested by round trip, by a hand-assembled
index block, and by the CRC-32C / Snappy known-answer vectors of their specifications, not against a
file written by TensorFlow.

Table format recap:  file = data blocks, metaindex block, index block, 48-byte footer.
  block   = entries, restart array (fixed32 each), fixed32 restart count
  entry   = varint32 shared_key_bytes, varint32 unshared_key_bytes, varint32 value_bytes, key delta, value
  every block is followed by a 5-byte trailer: compression type (0 none, 1 snappy), masked CRC-32C
  index block entry: key >= last key of a data block, value = BlockHandle (varint64 offset, varint64 size)
  footer  = metaindex BlockHandle, index BlockHandle, zero padding to 40 bytes, magic 0xdb4775248b80fb57 (LE)
"""
import glob
import os
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57

# DataType enum (tensorflow/core/framework/types.proto) <-> numpy
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
           10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_OF = {np.dtype(v): k for k, v in _DTYPES.items()}


# ---- CRC-32C (Castagnoli), TensorFlow's masking -------------------------------------------------
def _make_crc_table():
    tbl = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tbl.append(c)
    return np.array(tbl, dtype=np.uint32)


_CRC_TABLE = _make_crc_table()


def crc32c(data, crc=0):
    """CRC-32C of bytes-like `data` (RFC 3720 polynomial, reflected)."""
    c = (crc ^ 0xFFFFFFFF) & 0xFFFFFFFF
    tbl = _CRC_TABLE
    for b in bytes(data):
        c = int(tbl[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def crc32c_array(a):
    """CRC-32C of a (possibly large) array's bytes.  Short inputs: a table-driven loop over bytes grouped 8 at a time
    (slicing-by-8).  Long inputs (a discriminator's 8192x1024 dense kernel is 32 MB; the byte loop manages ~5 MB/s): the
    data is cut into K equal chunks whose CRCs advance in lock step -- one NumPy table lookup per byte POSITION for all
    chunks at once -- and the chunk CRCs are then joined with the shift operator of zlib's crc32_combine
    (CRC(A || B) = shift_len(B)(CRC(A)) xor CRC(B) for conditioned CRCs)."""
    data = np.frombuffer(np.asarray(a).tobytes(), dtype=np.uint8)
    if len(data) < (1 << 18):
        return _crc32c_np(data)
    K = 4096
    clen = len(data) // K
    body = data[:K * clen].reshape(K, clen)
    tbl = _CRC_TABLE
    c = np.full(K, 0xFFFFFFFF, dtype=np.uint32)
    for j in range(clen):
        c = tbl[(c ^ body[:, j]) & 0xFF] ^ (c >> np.uint32(8))
    c ^= np.uint32(0xFFFFFFFF)
    op = _crc32c_shift_operator(clen)
    total = int(c[0])
    for i in range(1, K):
        total = _gf2_apply(op, total) ^ int(c[i])
    tail = data[K * clen:]
    if len(tail):
        total = _gf2_apply(_crc32c_shift_operator(len(tail)), total) ^ _crc32c_np(tail)
    return total


def _gf2_apply(mat, vec):
    """mat: list of 32 column words; returns mat * vec over GF(2)."""
    out, i = 0, 0
    while vec:
        if vec & 1:
            out ^= mat[i]
        vec >>= 1
        i += 1
    return out


def _gf2_square(mat):
    return [_gf2_apply(mat, mat[i]) for i in range(32)]


_SHIFT_OPS = {}


def _crc32c_shift_operator(nbytes):
    """The 32x32 GF(2) matrix that advances a CRC-32C register over `nbytes` zero bytes (zlib crc32_combine's
    construction with the reflected Castagnoli polynomial 0x82F63B78)."""
    if nbytes in _SHIFT_OPS:
        return _SHIFT_OPS[nbytes]
    odd = [0x82F63B78] + [1 << i for i in range(31)]          # one zero BIT
    even = _gf2_square(odd)                                    # two bits
    odd = _gf2_square(even)                                    # four bits
    # identity, then multiply in the operators of the set bits of nbytes (each squaring doubles the span; the first
    # squaring below yields the operator for 8 bits = one byte)
    result = [1 << i for i in range(32)]
    n = nbytes
    cur = odd
    while n:
        cur = _gf2_square(cur)                                 # 8, 16, 32, ... bits = 1, 2, 4, ... bytes
        if n & 1:
            result = [_gf2_apply(cur, result[i]) for i in range(32)]
        n >>= 1
    _SHIFT_OPS[nbytes] = result
    return result


_SLICE8 = None


def _slice8_tables():
    global _SLICE8
    if _SLICE8 is None:
        t = np.zeros((8, 256), dtype=np.uint32)
        t[0] = _CRC_TABLE
        for k in range(1, 8):
            t[k] = t[0][t[k - 1] & 0xFF] ^ (t[k - 1] >> 8)
        _SLICE8 = t
    return _SLICE8


def _crc32c_np(data):
    t = _slice8_tables()
    c = 0xFFFFFFFF
    n8 = len(data) // 8
    if n8:
        words = data[:n8 * 8].reshape(n8, 8)
        lo = (words[:, 0].astype(np.uint32) | (words[:, 1].astype(np.uint32) << 8) |
              (words[:, 2].astype(np.uint32) << 16) | (words[:, 3].astype(np.uint32) << 24))
        # the recurrence over words is inherently serial; Python loop over words (8 bytes per iteration)
        t0, t1, t2, t3, t4, t5, t6, t7 = (t[k] for k in range(8))
        w4, w5, w6, w7 = words[:, 4], words[:, 5], words[:, 6], words[:, 7]
        for i in range(n8):
            x = c ^ int(lo[i])
            c = (int(t7[x & 0xFF]) ^ int(t6[(x >> 8) & 0xFF]) ^ int(t5[(x >> 16) & 0xFF]) ^ int(t4[(x >> 24) & 0xFF]) ^
                 int(t3[w4[i]]) ^ int(t2[w5[i]]) ^ int(t1[w6[i]]) ^ int(t0[w7[i]]))
    for b in data[n8 * 8:]:
        c = int(_CRC_TABLE[(c ^ int(b)) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(crc):
    """tensorflow/core/lib/hash/crc32c.h: rotate right by 15 and add a constant."""
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def unmask_crc(masked):
    rot = (masked - 0xa282ead8) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ---- varints / tiny protobuf ----------------------------------------------------------------------
def _put_varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def _get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError('malformed varint')


def _pb_fields(buf):
    """Yields (field_number, wire_type, value) of one protobuf message; value is an int (varint / fixed)
    or bytes (length-delimited)."""
    pos = 0
    buf = bytes(buf)
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield fn, wt, v


def _pb_varint_field(fn, v):
    return _put_varint((fn << 3) | 0) + _put_varint(v)


def _pb_bytes_field(fn, b):
    return _put_varint((fn << 3) | 2) + _put_varint(len(b)) + bytes(b)


def _pb_fixed32_field(fn, v):
    return _put_varint((fn << 3) | 5) + struct.pack('<I', v)


def _encode_shape(shape):
    # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1; } }
    return b''.join(_pb_bytes_field(2, _pb_varint_field(1, int(d))) for d in shape)


def _decode_shape(buf):
    dims = []
    for fn, _, v in _pb_fields(buf):
        if fn == 2:
            size = 0
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    size = v2 - (1 << 64) if v2 >= (1 << 63) else v2
            dims.append(size)
    return tuple(dims)


# ---- Snappy (raw block format) decompression; TensorFlow may compress table blocks -----------------
def snappy_decompress(buf):
    buf = bytes(buf)
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                  # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:                                  # copy, 1-byte offset
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:                                # copy, 2-byte offset
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], 'little')
            pos += 2
        else:                                          # copy, 4-byte offset
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError('malformed snappy stream')
        for _ in range(ln):                            # (may overlap its own output)
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('snappy length mismatch')
    return bytes(out)


def snappy_compress(data):
    """A small Snappy (raw block format) compressor: greedy matching of 4-byte sequences through a hash table, copies
    with 1- and 2-byte offsets, literals of any length.  TensorFlow's tables are written by LevelDB's builder, which
    Snappy-compresses a block when that saves at least 1/8; write_table(compress=True) does the same with this, so
    that the reader is exercised on compressed blocks (checkpoint keys such as conv2d_12/kernel/Adam_1 repeat a lot)."""
    data = bytes(data)
    n = len(data)
    out = bytearray(_put_varint(n))

    def literal(lo, hi):
        while lo < hi:
            ln = min(hi - lo, 1 << 16)
            if ln <= 60:
                out.append((ln - 1) << 2)
            else:
                nb = 1 if ln - 1 < 256 else 2
                out.append((59 + nb) << 2)
                out.extend((ln - 1).to_bytes(nb, 'little'))
            out.extend(data[lo:lo + ln])
            lo += ln

    table, i, lit = {}, 0, 0
    while i + 4 <= n:
        key = data[i:i + 4]
        cand = table.get(key)
        table[key] = i
        if cand is not None and i - cand < 65536:
            m = 4
            while i + m < n and m < 64 and data[cand + m] == data[i + m]:
                m += 1
            literal(lit, i)
            off = i - cand
            if m <= 11 and off < 2048:
                out.append(((m - 4) << 2) | 1 | ((off >> 8) << 5))
                out.append(off & 0xFF)
            else:
                out.append(((m - 1) << 2) | 2)
                out.extend(off.to_bytes(2, 'little'))
            i += m
            lit = i
        else:
            i += 1
    literal(lit, n)
    return bytes(out)


# ---- table blocks ------------------------------------------------------------------------------------
def _read_block(f, offset, size, verify=True):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) != size + 5:
        raise ValueError('truncated table block')
    body, ctype = raw[:size], raw[size]
    if verify:
        want = unmask_crc(struct.unpack_from('<I', raw, size + 1)[0])
        if crc32c(raw[:size + 1]) != want:
            raise ValueError('table block checksum mismatch')
    if ctype == 1:
        body = snappy_decompress(body)
    elif ctype != 0:
        raise ValueError('unknown block compression type %d' % ctype)
    return body


def _block_entries(block):
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos = 0
    key = b''
    while pos < end:
        shared, pos = _get_varint(block, pos)
        unshared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + unshared]
        pos += unshared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _read_handle(buf, pos):
    off, pos = _get_varint(buf, pos)
    size, pos = _get_varint(buf, pos)
    return off, size, pos


def read_table(path, verify=True):
    """{key bytes: value bytes} of an SSTable file."""
    with open(path, 'rb') as f:
        f.seek(0, os.SEEK_END)
        total = f.tell()
        if total < 48:
            raise ValueError('%s: too short for a table' % path)
        f.seek(total - 48)
        footer = f.read(48)
        if struct.unpack_from('<Q', footer, 40)[0] != _MAGIC:
            raise ValueError('%s: not a TensorFlow / LevelDB table (bad magic)' % path)
        _, _, p = _read_handle(footer, 0)                     # metaindex: unused
        ioff, isize, _ = _read_handle(footer, p)
        out = {}
        for _, handle in _block_entries(_read_block(f, ioff, isize, verify)):
            boff, bsize, _ = _read_handle(handle, 0)
            for k, v in _block_entries(_read_block(f, boff, bsize, verify)):
                out[bytes(k)] = bytes(v)
        return out


class _BlockBuilder(object):
    def __init__(self, restart_interval=16):
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b''
        self.interval = restart_interval

    def add(self, key, value):
        shared = 0
        if self.count % self.interval == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4


def write_table(path, items, block_size=4096, compress=False):
    """items: iterable of (key bytes, value bytes) in strictly increasing key order.  compress=False: blocks are
    stored raw (type 0); compress=True: a block is stored Snappy-compressed (type 1) when that saves at least 1/8 of
    it, as LevelDB's table builder (which writes TensorFlow's .index files) does."""
    with open(path, 'wb') as f:
        def emit(body):
            off = f.tell()
            ctype = 0
            if compress:
                packed = snappy_compress(body)
                if len(packed) < len(body) - len(body) // 8:
                    body, ctype = packed, 1
            trailer = bytes([ctype])
            f.write(body + trailer + struct.pack('<I', mask_crc(crc32c(body + trailer))))
            return off, len(body)

        index = _BlockBuilder(restart_interval=1)
        blk = _BlockBuilder()
        prev = None
        for key, value in items:
            if prev is not None and not key > prev:
                raise ValueError('table keys must be strictly increasing')
            if blk.count and blk.size() >= block_size:
                off, size = emit(blk.finish())
                index.add(prev, _put_varint(off) + _put_varint(size))
                blk = _BlockBuilder()
            blk.add(key, value)
            prev = key
        if blk.count:
            off, size = emit(blk.finish())
            index.add(prev, _put_varint(off) + _put_varint(size))
        moff, msize = emit(_BlockBuilder().finish())
        ioff, isize = emit(index.finish())
        footer = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
        f.write(footer + b'\0' * (40 - len(footer)) + struct.pack('<Q', _MAGIC))


# ---- bundle level ------------------------------------------------------------------------------------
def _data_path(prefix, shard, num_shards):
    return '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)


def list_variables(prefix, verify=True):
    """[(name, shape, numpy dtype)] of a checkpoint, like tf.train.list_variables."""
    out = []
    for k, v in sorted(read_table(prefix + '.index', verify).items()):
        if k == b'':
            continue
        e = _decode_entry(v)
        out.append((k.decode(), e['shape'], e['dtype']))
    return out


def _decode_entry(buf):
    e = {'dtype': None, 'shape': (), 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'sliced': False}
    for fn, _, v in _pb_fields(buf):
        if fn == 1:
            if v not in _DTYPES:
                raise ValueError('unsupported tensor dtype enum %d' % v)
            e['dtype'] = np.dtype(_DTYPES[v])
        elif fn == 2:
            e['shape'] = _decode_shape(v)
        elif fn == 3:
            e['shard_id'] = v
        elif fn == 4:
            e['offset'] = v
        elif fn == 5:
            e['size'] = v
        elif fn == 6:
            e['crc32c'] = v
        elif fn == 7:
            e['sliced'] = True
    return e


def load_checkpoint(prefix, verify=True, names=None):
    """{variable name: ndarray} of a V2 checkpoint `<prefix>` (what `saver.restore` + `session.run(v)` yields
    in espcn/espcn/model_espcn.py:150-166).  `names`: optional subset.  verify: check the table block and
    per-tensor CRC-32Cs."""
    table = read_table(prefix + '.index', verify)
    if b'' not in table:
        raise ValueError('%s.index: no bundle header' % prefix)
    num_shards, endian = 1, 0
    for fn, _, v in _pb_fields(table[b'']):
        if fn == 1:
            num_shards = v
        elif fn == 2:
            endian = v
    if endian != 0:
        raise ValueError('big-endian bundles are not supported')
    out = {}
    files = {}
    try:
        for k, v in table.items():
            if k == b'':
                continue
            name = k.decode()
            if names is not None and name not in names:
                continue
            e = _decode_entry(v)
            if e['sliced']:
                raise ValueError('%s: partitioned (sliced) variables are not supported' % name)
            f = files.get(e['shard_id'])
            if f is None:
                f = files[e['shard_id']] = open(_data_path(prefix, e['shard_id'], num_shards), 'rb')
            f.seek(e['offset'])
            raw = f.read(e['size'])
            if len(raw) != e['size']:
                raise ValueError('%s: truncated tensor data' % name)
            n = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
            if n * e['dtype'].itemsize != e['size']:
                raise ValueError('%s: size %d does not match shape %s' % (name, e['size'], e['shape']))
            arr = np.frombuffer(raw, dtype=e['dtype']).reshape(e['shape']).copy()
            if verify and e['crc32c'] is not None and crc32c_array(arr) != unmask_crc(e['crc32c']):
                raise ValueError('%s: tensor checksum mismatch' % name)
            out[name] = arr
    finally:
        for f in files.values():
            f.close()
    return out


def save_checkpoint(prefix, tensors, num_shards=1, compress_index=False):
    """Writes {name: array-like} as a V2 checkpoint that `tf.train.Saver.restore` / `tf.train.load_checkpoint` read:
    `<prefix>.index` and `<prefix>.data-0000k-of-0000n`.  The Saver of a single-device graph writes one shard
    (num_shards=1, the default); a sharded Saver spreads the tensors over several data files -- num_shards > 1 deals
    them out round-robin (each BundleEntryProto names its shard_id).  compress_index: see write_table."""
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    items = []
    # written under temporary names and renamed when complete, the index last: a prefix whose `.index` exists is whole
    # (a run killed during a save leaves only `*.tmp` files behind)
    files = [open(_data_path(prefix, s, num_shards) + '.tmp', 'wb') for s in range(num_shards)]
    offsets = [0] * num_shards
    try:
        for j, name in enumerate(sorted(tensors, key=lambda s: s.encode())):
            a = np.asarray(tensors[name])           # (np.ascontiguousarray would turn a 0-d array into 1-d)
            if a.dtype not in _DTYPE_OF:
                raise ValueError('%s: dtype %s has no TensorFlow counterpart here' % (name, a.dtype))
            raw = a.tobytes()
            shard = j % num_shards
            files[shard].write(raw)
            entry = (_pb_varint_field(1, _DTYPE_OF[a.dtype]) + _pb_bytes_field(2, _encode_shape(a.shape)) +
                     (_pb_varint_field(3, shard) if shard else b'') +
                     (_pb_varint_field(4, offsets[shard]) if offsets[shard] else b'') + _pb_varint_field(5, len(raw)) +
                     _pb_fixed32_field(6, mask_crc(crc32c_array(a))))
            items.append((name.encode(), entry))
            offsets[shard] += len(raw)
    finally:
        for f in files:
            f.close()
    # BundleHeaderProto { num_shards; endianness = LITTLE (0, default: omitted); version { producer = 1 } }
    header = _pb_varint_field(1, num_shards) + _pb_bytes_field(3, _pb_varint_field(1, 1))
    write_table(prefix + '.index.tmp', [(b'', header)] + items, compress=compress_index)
    # (overwriting an existing prefix: its index goes first, so that a kill between the renames below never leaves the
    # OLD index over NEW data -- a prefix without an index is simply not a checkpoint, see is_checkpoint_prefix)
    if os.path.exists(prefix + '.index'):
        os.remove(prefix + '.index')
    for sh in range(num_shards):
        os.replace(_data_path(prefix, sh, num_shards) + '.tmp', _data_path(prefix, sh, num_shards))
    os.replace(prefix + '.index.tmp', prefix + '.index')


def tf_beta_power(beta, steps):
    """The value of AdamOptimizer's `beta1_power` / `beta2_power` accumulator after `steps` applies, as a float32 scalar:
    TF-1.x creates it with the value beta and multiplies it by beta (a float32 tensor) once per apply, so the checkpoint
    holds float32(beta) ** (steps + 1) up to the random walk of the float32 roundings (~6e-8 * sqrt(steps) relative).
    The base is float32(beta), not the Python float: float32(0.999) = 0.99900001287..., and over 40,000 steps the two
    differ by a whole step's factor."""
    return np.asarray(float(np.float32(beta)) ** (int(steps) + 1), dtype=np.float32)


# Prefixes THIS process has saved, per directory: the only ones update_checkpoint_state may delete.  tf.train.Saver keeps
# the same kind of list per Saver instance (`_last_checkpoints`) -- checkpoints it finds listed in the state file of a
# directory it resumes in (an earlier run's, TensorFlow's own) stay listed and stay on disk.
_saved_by_this_process = {}


def update_checkpoint_state(prefix, max_to_keep=5):
    """Writes `<dir>/checkpoint`, the CheckpointState text proto that tf.train.Saver.save maintains and
    tf.train.latest_checkpoint(dir) reads (vdsr/vdsr/experiment_train.py:108): the newest prefix as
    `model_checkpoint_path`, the known prefixes under `all_model_checkpoint_paths` (paths relative to the
    directory, as the Saver writes them).  Like tf.train.Saver(max_to_keep=5) -- the default every script of the
    reference uses -- at most `max_to_keep` of the prefixes THIS PROCESS saved stay on disk: the older ones of them are
    deleted and dropped from the list.  Entries the state file already held (an earlier run's checkpoints, TensorFlow's
    own, absolute paths elsewhere) are never deleted; they stay listed in front, as long as they exist.  The state file
    itself is replaced atomically."""
    d = os.path.dirname(os.path.abspath(prefix))
    name = os.path.basename(prefix)
    state = os.path.join(d, 'checkpoint')
    known = []
    if os.path.exists(state):
        for line in open(state):
            line = line.strip()
            if line.startswith('all_model_checkpoint_paths:'):
                known.append(line.split(':', 1)[1].strip().strip('"'))
    mine = _saved_by_this_process.setdefault(d, [])
    if name in mine:
        mine.remove(name)
    mine.append(name)
    dropped = mine[:-max_to_keep] if max_to_keep and len(mine) > max_to_keep else []
    del mine[:len(dropped)]
    foreign = [k for k in known if k not in mine and k not in dropped and k != name and
               is_checkpoint_prefix(k if os.path.isabs(k) else os.path.join(d, k))]
    known = foreign + mine
    with open(state + '.tmp', 'w') as f:
        f.write('model_checkpoint_path: "%s"\n' % name)
        for k in known:
            f.write('all_model_checkpoint_paths: "%s"\n' % k)
    os.replace(state + '.tmp', state)
    for k in dropped:
        old = os.path.join(d, k)         # (names this process wrote into this directory: never absolute, never foreign)
        for path in [old + '.index'] + glob.glob(glob.escape(old) + '.data-?????-of-?????'):
            if os.path.exists(path):
                os.remove(path)


def latest_checkpoint(ckpt_dir, scan=False):
    """tf.train.latest_checkpoint(dir): the `model_checkpoint_path` of the `checkpoint` state file, if that
    prefix exists; None otherwise.  scan=True: when the state file is missing, or names a prefix that is not there
    (a copy of the directory without it; a run killed between writing a bundle and the state file), fall back to the
    highest-numbered complete `model.ckpt-<N>` bundle of the directory instead of silently starting from step 0."""
    state = os.path.join(ckpt_dir, 'checkpoint')
    if os.path.exists(state):
        for line in open(state):
            line = line.strip()
            if line.startswith('model_checkpoint_path:'):
                name = line.split(':', 1)[1].strip().strip('"')
                prefix = name if os.path.isabs(name) else os.path.join(ckpt_dir, name)
                if is_checkpoint_prefix(prefix):
                    return prefix
                break
    return newest_bundle(ckpt_dir) if scan else None


def newest_bundle(ckpt_dir, stem='model.ckpt-'):
    """The complete bundle `<stem><N>` with the largest N in `ckpt_dir` (index and every data shard present)."""
    best = None
    if not (ckpt_dir and os.path.isdir(ckpt_dir)):
        return None
    for name in os.listdir(ckpt_dir):
        if not (name.startswith(stem) and name.endswith('.index')):
            continue
        try:
            n = int(name[len(stem):-len('.index')])
        except ValueError:
            continue
        prefix = os.path.join(ckpt_dir, name[:-len('.index')])
        shards = glob.glob(glob.escape(prefix) + '.data-?????-of-?????')
        if shards and len(shards) == int(shards[0].rsplit('-', 1)[1]) and (best is None or n > best[0]):
            best = (n, prefix)
    return best[1] if best else None


def is_checkpoint_prefix(path):
    return isinstance(path, str) and os.path.exists(path + '.index')
