"""TensorFlow checkpoint-v2 ("tensor bundle") reader / writer in pure Python + NumPy.

The reference restores encoder weights and the codebook with ``tf.train.Saver.restore``
(/root/reference/auto_pose/ae/ae_factory.py:149-172, ae_embed.py:60,91); trained AAEs are
distributed as such checkpoints (README.md:182-186).  TensorFlow is not available on the
MI355X image, so this module reads the files directly and hands ``Saver.restore`` the same
{variable name: array} dict the native ``.npz`` container holds.

Format (restated from TensorFlow's published sources; no TensorFlow code is used or needed):

* ``<prefix>.index`` is an SSTable in LevelDB's table format (leveldb doc/table_format.md;
  tensorflow/core/lib/io/table*.cc, format.cc): data blocks of prefix-compressed
  (shared, non_shared, value_len, key_delta, value) entries + a restart array, each block
  followed by a 1-byte compression type (0 none, 1 snappy) and a masked CRC-32C; an index
  block of (last_key -> BlockHandle); a 48-byte footer (metaindex handle, index handle,
  padding, magic 0xdb4775248b80fb57).
* key ""  -> BundleHeaderProto {num_shards=1, endianness=2, version=3}
  key name -> BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (fixed32,
  masked CRC-32C of the tensor bytes), slices=7}   (tensorflow/core/protobuf/tensor_bundle.proto)
* ``<prefix>.data-SSSSS-of-NNNNN`` hold the raw little-endian tensor bytes at [offset, offset+size).
* ``checkpoint`` (text CheckpointState proto): model_checkpoint_path / all_model_checkpoint_paths.

PARITY NOTE: no TensorFlow-written checkpoint exists in this environment (none ships with the
reference, no network), so the reader is verified against (a) this module's own writer, (b) CRC-32C /
varint / snappy known-answer vectors, (c) checkpoints assembled in the tests WITHOUT this module's
writers -- entries serialised by the protobuf runtime from the published tensor_bundle.proto, tables
laid out by an independent builder (several blocks, snappy-compressed blocks, two data shards) -- and
(d) a corruption suite (truncation at every structural boundary, flipped bits, unknown enums, sliced
variables, big-endian header, missing shards: each must raise, none may return arrays)
(tests/test_tf_checkpoint.py).  "parity unpinned" against real TF output until a sample checkpoint is
available.
"""
from __future__ import annotations

import os
import re
import struct
import sys

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
DT_TO_NUMPY = {
    1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'), 5: np.dtype('<i2'),
    6: np.dtype('i1'), 9: np.dtype('<i8'), 10: np.dtype('?'), 17: np.dtype('<u2'), 19: np.dtype('<f2'),
    22: np.dtype('<u4'), 23: np.dtype('<u8'),
    14: np.dtype('<u2'),          # DT_BFLOAT16: returned as raw bit patterns
}
NUMPY_TO_DT = {np.dtype('float32'): 1, np.dtype('float64'): 2, np.dtype('int32'): 3, np.dtype('uint8'): 4,
               np.dtype('int16'): 5, np.dtype('int8'): 6, np.dtype('int64'): 9, np.dtype('bool'): 10,
               np.dtype('uint16'): 17, np.dtype('float16'): 19, np.dtype('uint32'): 22, np.dtype('uint64'): 23}


# ---------------------------------------------------------------- CRC-32C (Castagnoli) ------
def _make_crc_tables():
    poly = 0x82F63B78
    t0 = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        t0[i] = c
    tabs = [t0]
    for _ in range(7):
        prev = tabs[-1]
        tabs.append((prev >> np.uint32(8)) ^ t0[prev & np.uint32(0xFF)])
    return tabs


_CRC_TABLES = _make_crc_tables()
_CRC_T0 = [int(v) for v in _CRC_TABLES[0]]


def _crc_state_scalar(buf, c):
    """Advance the (inverted) CRC register c over the uint8 array buf, 8 bytes per step."""
    T = _CRC_TABLES
    n = len(buf)
    words = n // 8
    if words:
        b = buf[:words * 8].reshape(words, 8)
        lo = (b[:, 0].astype(np.uint32) | (b[:, 1].astype(np.uint32) << 8) | (b[:, 2].astype(np.uint32) << 16) |
              (b[:, 3].astype(np.uint32) << 24)).tolist()
        hi = b[:, 4:8].tolist()
        t7, t6, t5, t4 = T[7].tolist(), T[6].tolist(), T[5].tolist(), T[4].tolist()
        t3, t2, t1, t0 = T[3].tolist(), T[2].tolist(), T[1].tolist(), T[0].tolist()
        for w in range(words):
            x = c ^ lo[w]
            h = hi[w]
            c = (t7[x & 0xFF] ^ t6[(x >> 8) & 0xFF] ^ t5[(x >> 16) & 0xFF] ^ t4[x >> 24] ^
                 t3[h[0]] ^ t2[h[1]] ^ t1[h[2]] ^ t0[h[3]])
    for v in buf[words * 8:].tolist():
        c = _CRC_T0[(c ^ v) & 0xFF] ^ (c >> 8)
    return c


_CHUNK = 4096
_SHIFT_TABLES = []          # advance-the-register-over-_CHUNK-zero-bytes, as 4 byte-indexed tables


def _shift_tables():
    if not _SHIFT_TABLES:
        t0 = _CRC_TABLES[0]
        v = (np.arange(256, dtype=np.uint32)[None, :] << (np.arange(4, dtype=np.uint32) * 8)[:, None]).reshape(-1)
        for _ in range(_CHUNK):
            v = t0[v & np.uint32(0xFF)] ^ (v >> np.uint32(8))
        _SHIFT_TABLES.extend(row.tolist() for row in v.reshape(4, 256))
    return _SHIFT_TABLES


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli) of a bytes-like object or array.  Large inputs are cut into 4 KiB
    chunks whose registers advance together as NumPy vectors (the CRC is linear over GF(2));
    the per-chunk results are then chained with a precomputed shift-by-4-KiB map."""
    if isinstance(data, np.ndarray):
        buf = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    else:
        buf = np.frombuffer(data, dtype=np.uint8)
    c = (~crc) & 0xFFFFFFFF
    nchunks = len(buf) // _CHUNK
    if nchunks >= 4:
        T = _CRC_TABLES
        body = buf[:nchunks * _CHUNK].reshape(nchunks, _CHUNK // 8, 8)
        lo = body[:, :, :4].copy().view('<u4')[:, :, 0]
        state = np.zeros(nchunks, dtype=np.uint32)
        m = np.uint32(0xFF)
        for w in range(_CHUNK // 8):
            x = state ^ lo[:, w]
            h = body[:, w, 4:]
            state = (T[7][x & m] ^ T[6][(x >> np.uint32(8)) & m] ^ T[5][(x >> np.uint32(16)) & m] ^ T[4][x >> np.uint32(24)] ^
                     T[3][h[:, 0]] ^ T[2][h[:, 1]] ^ T[1][h[:, 2]] ^ T[0][h[:, 3]])
        s0, s1, s2, s3 = _shift_tables()
        for r in state.tolist():
            c = s0[c & 0xFF] ^ s1[(c >> 8) & 0xFF] ^ s2[(c >> 16) & 0xFF] ^ s3[c >> 24] ^ r
        buf = buf[nchunks * _CHUNK:]
    c = _crc_state_scalar(buf, c)
    return (~c) & 0xFFFFFFFF


def mask_crc(crc):
    """tensorflow/core/lib/hash/crc32c.h Mask(): rotate right by 15, add a constant."""
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked):
    rot = (masked - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ---------------------------------------------------------------- varints / protobuf wire ---
def put_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def get_varint(buf, pos):
    result = shift = 0
    while True:
        if pos >= len(buf):
            raise ValueError('truncated varint')
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError('varint too long')


def _proto_fields(buf):
    """Yield (field_number, wire_type, value) of one protobuf message; value is int or bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            if len(v) != ln:
                raise ValueError('truncated length-delimited field')
            pos += ln
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield field, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


class BundleEntry(object):
    __slots__ = ('dtype', 'shape', 'shard_id', 'offset', 'size', 'crc32c', 'has_slices')

    def __init__(self, dtype=0, shape=(), shard_id=0, offset=0, size=0, crc=0, has_slices=False):
        self.dtype, self.shape, self.shard_id, self.offset, self.size, self.crc32c, self.has_slices = \
            dtype, tuple(shape), shard_id, offset, size, crc, has_slices

    @classmethod
    def parse(cls, buf):
        e = cls()
        for field, wt, v in _proto_fields(buf):
            if field == 1:
                e.dtype = v
            elif field == 2:                      # TensorShapeProto
                dims = []
                for f2, _, v2 in _proto_fields(v):
                    if f2 == 2:                   # Dim {size=1, name=2}
                        size = 0
                        for f3, _, v3 in _proto_fields(v2):
                            if f3 == 1:
                                size = _signed64(v3)
                        dims.append(size)
                    elif f2 == 3 and v2:
                        raise ValueError('tensor of unknown rank in checkpoint')
                e.shape = tuple(dims)
            elif field == 3:
                e.shard_id = v
            elif field == 4:
                e.offset = v
            elif field == 5:
                e.size = v
            elif field == 6:
                e.crc32c = v
            elif field == 7:
                e.has_slices = True
        return e

    def serialize(self):
        out = bytearray()
        if self.dtype:
            out += b'\x08' + put_varint(self.dtype)
        shape = bytearray()
        for d in self.shape:
            dim = (b'\x08' + put_varint(d)) if d else b''
            shape += b'\x12' + put_varint(len(dim)) + dim
        out += b'\x12' + put_varint(len(shape)) + shape
        if self.shard_id:
            out += b'\x18' + put_varint(self.shard_id)
        if self.offset:
            out += b'\x20' + put_varint(self.offset)
        if self.size:
            out += b'\x28' + put_varint(self.size)
        out += b'\x35' + struct.pack('<I', self.crc32c)
        return bytes(out)


# ---------------------------------------------------------------- snappy (decompress only) --
def snappy_decompress(buf):
    """Raw snappy block format (google/snappy format_description.txt)."""
    n, pos = get_varint(buf, 0)
    out = bytearray()
    ln = len(buf)
    while pos < ln:
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                              # literal
            size = tag >> 2
            if size >= 60:
                nb = size - 59
                size = int.from_bytes(buf[pos:pos + nb], 'little')
                pos += nb
            size += 1
            out += buf[pos:pos + size]
            pos += size
            continue
        if kind == 1:
            length = ((tag >> 2) & 7) + 4
            offset = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            length = (tag >> 2) + 1
            offset = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            length = (tag >> 2) + 1
            offset = int.from_bytes(buf[pos:pos + 4], 'little')
            pos += 4
        if offset == 0 or offset > len(out):
            raise ValueError('corrupt snappy stream (bad copy offset)')
        for _ in range(length):                    # copies may overlap their own output
            out.append(out[-offset])
    if len(out) != n:
        raise ValueError('corrupt snappy stream (length %d, header says %d)' % (len(out), n))
    return bytes(out)


# ---------------------------------------------------------------- SSTable ---------------------
def _read_block(f, offset, size, verify):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) != size + 5:
        raise ValueError('truncated table block at %d' % offset)
    contents, ctype, stored = raw[:size], raw[size], struct.unpack('<I', raw[size + 1:])[0]
    if verify and unmask_crc(stored) != crc32c(raw[:size + 1]):
        raise ValueError('table block checksum mismatch at offset %d' % offset)
    if ctype == 0:
        return contents
    if ctype == 1:
        return snappy_decompress(contents)
    raise ValueError('unknown block compression type %d' % ctype)


def _block_entries(block):
    if len(block) < 4:
        raise ValueError('table block too small')
    num_restarts = struct.unpack('<I', block[-4:])[0]
    limit = len(block) - 4 - 4 * num_restarts
    if limit < 0:
        raise ValueError('corrupt restart array')
    pos, key = 0, b''
    while pos < limit:
        shared, pos = get_varint(block, pos)
        non_shared, pos = get_varint(block, pos)
        vlen, pos = get_varint(block, pos)
        if shared > len(key):
            raise ValueError('corrupt key prefix')
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        value = bytes(block[pos:pos + vlen])
        pos += vlen
        yield key, value


def _get_handle(buf, pos):
    off, pos = get_varint(buf, pos)
    size, pos = get_varint(buf, pos)
    return (off, size), pos


def read_table(path, verify_checksums=True):
    """All (key, value) pairs of an SSTable file, in key order."""
    out = []
    with open(path, 'rb') as f:
        f.seek(0, os.SEEK_END)
        total = f.tell()
        if total < 48:
            raise ValueError('%s: too short to be a table (%d bytes)' % (path, total))
        f.seek(total - 48)
        footer = f.read(48)
        if struct.unpack('<Q', footer[40:])[0] != TABLE_MAGIC:
            raise ValueError('%s: bad table magic (not a TensorFlow checkpoint index)' % path)
        _, pos = _get_handle(footer, 0)                    # metaindex (unused)
        (ioff, isize), _ = _get_handle(footer, pos)
        for _, hv in _block_entries(_read_block(f, ioff, isize, verify_checksums)):
            (boff, bsize), _ = _get_handle(hv, 0)
            out.extend(_block_entries(_read_block(f, boff, bsize, verify_checksums)))
    return out


def _build_block(pairs, restart_interval=16):
    out, restarts, last = bytearray(), [], b''
    for i, (k, v) in enumerate(pairs):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            m = min(len(k), len(last))
            while shared < m and k[shared] == last[shared]:
                shared += 1
        out += put_varint(shared) + put_varint(len(k) - shared) + put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_table(path, pairs, block_size=4096):
    """pairs: iterable of (key bytes, value bytes); written sorted, uncompressed."""
    pairs = sorted(pairs)
    with open(path, 'wb') as f:
        def emit(block):
            off = f.tell()
            f.write(block + b'\x00' + struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
            return off, len(block)
        index, cur, cur_bytes = [], [], 0
        for k, v in pairs:
            cur.append((k, v))
            cur_bytes += len(k) + len(v) + 8
            if cur_bytes >= block_size:
                off, size = emit(_build_block(cur))
                index.append((cur[-1][0], put_varint(off) + put_varint(size)))
                cur, cur_bytes = [], 0
        if cur or not index:
            off, size = emit(_build_block(cur))
            index.append((cur[-1][0] if cur else b'', put_varint(off) + put_varint(size)))
        moff, msize = emit(_build_block([]))
        ioff, isize = emit(_build_block(index, restart_interval=1))
        footer = put_varint(moff) + put_varint(msize) + put_varint(ioff) + put_varint(isize)
        f.write(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC))


# ---------------------------------------------------------------- bundle ----------------------
class BundleReader(object):
    """reader = BundleReader('/path/chkpt-30000'); reader.names(); reader.tensor(name)"""

    def __init__(self, prefix, verify_checksums=True):
        self.prefix = prefix
        self.verify = verify_checksums
        index = prefix + '.index'
        if not os.path.exists(index):
            raise FileNotFoundError('%s not found (expected a TensorFlow checkpoint-v2 prefix)' % index)
        self.num_shards, self.entries = 1, {}
        for key, value in read_table(index, verify_checksums):
            if key == b'':
                for field, _, v in _proto_fields(value):
                    if field == 1:
                        self.num_shards = v
                    elif field == 2 and v != 0:
                        raise ValueError('big-endian checkpoints are not supported')
                if not 1 <= self.num_shards <= 100000:
                    raise ValueError('%s: implausible shard count %r in the bundle header' % (index, self.num_shards))
            else:
                self.entries[key.decode('utf-8')] = BundleEntry.parse(value)

    def names(self):
        return sorted(self.entries)

    def shape(self, name):
        return self.entries[name].shape

    def _data_path(self, shard):
        return '%s.data-%05d-of-%05d' % (self.prefix, shard, self.num_shards)

    def tensor(self, name):
        e = self.entries[name]
        if e.has_slices:
            raise ValueError('%r is a partitioned (sliced) variable; not supported' % name)
        if e.dtype not in DT_TO_NUMPY:
            raise ValueError('%r has unsupported dtype enum %d' % (name, e.dtype))
        dt = DT_TO_NUMPY[e.dtype]
        count = int(np.prod(e.shape, dtype=np.int64)) if e.shape else 1
        if count * dt.itemsize != e.size:
            raise ValueError('%r: %d bytes on disk, shape %s needs %d' % (name, e.size, e.shape, count * dt.itemsize))
        if not 0 <= e.shard_id < self.num_shards:
            raise ValueError('%r: shard %d outside the %d data shard(s) the header announces' % (name, e.shard_id, self.num_shards))
        if e.offset < 0 or e.size < 0:
            raise ValueError('%r: negative offset/size in its index entry' % name)
        path = self._data_path(e.shard_id)
        if not os.path.exists(path):
            raise FileNotFoundError('%s not found (data shard of %r)' % (path, name))
        with open(path, 'rb') as f:
            f.seek(e.offset)
            raw = f.read(e.size)
        if len(raw) != e.size:
            raise ValueError('%r: data shard truncated' % name)
        if self.verify and unmask_crc(e.crc32c) != crc32c(raw):
            raise ValueError('%r: tensor checksum mismatch (corrupt data shard)' % name)
        return np.frombuffer(raw, dtype=dt).reshape(e.shape).copy()


def write_bundle(prefix, tensors):
    """Single-shard checkpoint-v2 writer: {name: array} -> <prefix>.index + .data-00000-of-00001.
    Lets ae_embed output travel back to a TensorFlow installation (tf.train.load_checkpoint) and
    gives the reader's tests their fixtures."""
    pairs, offset = [], 0
    header = b'\x08\x01' + b'\x1a\x02\x08\x01'          # num_shards=1, version{producer=1}
    pairs.append((b'', header))
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        for name in sorted(tensors):
            a = np.asarray(tensors[name], order='C')          # (ascontiguousarray would turn rank 0 into rank 1)
            if a.dtype not in NUMPY_TO_DT:
                raise ValueError('%r: dtype %s has no TensorFlow counterpart here' % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder('<'), copy=False).tobytes()
            f.write(raw)
            e = BundleEntry(NUMPY_TO_DT[a.dtype], a.shape, 0, offset, len(raw), mask_crc(crc32c(raw)))
            pairs.append((name.encode('utf-8'), e.serialize()))
            offset += len(raw)
    write_table(prefix + '.index', pairs)


# ---------------------------------------------------------------- AAE variable mapping --------
_SLOT = re.compile(r'/(Adam(_\d+)?|Momentum|RMSProp(_\d+)?|ExponentialMovingAverage)$')
_CODEBOOK_VARS = ('embedding_normalized', 'embed_obj_bbs_var')


def checkpoint_scopes(names):
    """Experiment scopes present in a checkpoint ('<exp>/conv2d/kernel' -> '<exp>')."""
    return sorted({n.rsplit('/conv2d/kernel', 1)[0] for n in names if n.endswith('/conv2d/kernel')})


def load_aae_variables(prefix, scope=None, verify_checksums=True):
    """Read a trained AAE checkpoint into the native container layout.

    Returns (weights, embedding_normalized or None, embed_obj_bbs or None) where weights maps
    the scope-stripped TF names ('conv2d/kernel', 'dense/bias', 'batch_normalization/gamma',
    decoder variables 'dense_1/...', 'conv2d_4/...' too) to arrays.  Optimizer slots, counters
    and everything outside `scope` are dropped.  scope=None: the checkpoint must hold exactly
    one experiment scope (or none at all)."""
    reader = BundleReader(prefix, verify_checksums)
    names = reader.names()
    if scope is None:
        scopes = checkpoint_scopes(names)
        if len(scopes) > 1:
            raise ValueError('checkpoint holds several experiments %s: pass scope=' % scopes)
        scope = scopes[0] if scopes else ''
    lead = scope + '/' if scope else ''
    weights, emb, bbs = {}, None, None
    for n in names:
        if not n.startswith(lead) or _SLOT.search(n):
            continue
        short = n[len(lead):]
        head = short.split('/')[0]
        if short in _CODEBOOK_VARS:
            if short == 'embedding_normalized':
                emb = reader.tensor(n).astype(np.float32)
            else:
                bbs = reader.tensor(n).astype(np.int32)
        elif re.match(r'^(conv2d|dense|batch_normalization)(_\d+)?$', head) and '/' in short:
            weights[short] = reader.tensor(n)
    if 'conv2d/kernel' not in weights:
        raise ValueError('%s: no encoder variables under scope %r (found scopes %s)' % (prefix, scope, checkpoint_scopes(names)))
    return weights, emb, bbs


_STATE_LINE = re.compile(r'^\s*(model_checkpoint_path|all_model_checkpoint_paths)\s*:\s*"(.*)"\s*$')


def parse_checkpoint_state(text):
    """Text CheckpointState proto -> (model_checkpoint_path, [all paths]); (None, []) if the text
    is not in that format."""
    latest, every = None, []
    for line in text.splitlines():
        m = _STATE_LINE.match(line)
        if not m:
            continue
        val = m.group(2).encode('utf-8').decode('unicode_escape')
        if m.group(1) == 'model_checkpoint_path':
            latest = val
        else:
            every.append(val)
    return latest, every


def main(argv=None):
    """python -m augmentedautoencoder_amd.tf_checkpoint list <prefix>
       python -m augmentedautoencoder_amd.tf_checkpoint convert <prefix> <out.npz> [scope]
       python -m augmentedautoencoder_amd.tf_checkpoint export <in.npz> <prefix> [scope]"""
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) >= 2 and argv[0] == 'list':
        r = BundleReader(argv[1])
        for n in r.names():
            e = r.entries[n]
            print('%-60s dtype=%d shape=%s bytes=%d' % (n, e.dtype, list(e.shape), e.size))
        return 0
    if len(argv) >= 3 and argv[0] == 'convert':
        from . import weights as W
        weights, emb, bbs = load_aae_variables(argv[1], argv[3] if len(argv) > 3 else None)
        W.save_npz(argv[2], weights, emb, bbs)
        print('wrote %s: %d weight arrays%s%s' % (argv[2], len(weights), ', codebook %s' % (emb.shape,) if emb is not None else '',
                                                  ', obj bbs' if bbs is not None else ''))
        return 0
    if len(argv) >= 3 and argv[0] == 'export':
        from . import weights as W
        weights, emb, bbs = W.load_npz(argv[1])
        lead = argv[3] + '/' if len(argv) > 3 else ''
        blob = {lead + k: v for k, v in weights.items()}
        if emb is not None:
            blob[lead + 'embedding_normalized'] = emb
        if bbs is not None:
            blob[lead + 'embed_obj_bbs_var'] = bbs
        write_bundle(argv[2], blob)
        print('wrote %s.index and %s.data-00000-of-00001 (%d variables)' % (argv[2], argv[2], len(blob)))
        return 0
    print(main.__doc__)
    return 2


if __name__ == '__main__':
    sys.exit(main())
