"""TensorFlow checkpoint-v2 reader/writer (augmentedautoencoder_amd/tf_checkpoint.py): CRC-32C
and varint known answers, snappy decoding, protobuf wire compatibility checked against the
real google.protobuf runtime on a restatement of tensor_bundle.proto, SSTable round trips
(multi-block, prefix compression, corruption detection, a hand-assembled snappy block), and
the AAE variable mapping through factory.restore_checkpoint.
No TensorFlow-written file exists in this environment: see the module's PARITY NOTE."""
import configparser
import os
import struct

import numpy as np
import pytest

from augmentedautoencoder_amd import ae_factory as factory, session as S, tf_checkpoint as T, weights as W
from oracle import synth


def test_crc32c_known_answers_and_chunked_path():
    # RFC 3720 appendix B.4 vectors + the classic check value
    assert T.crc32c(b'123456789') == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert T.crc32c(b'') == 0
    rng = np.random.default_rng(0)
    big = rng.integers(0, 256, 5 * 4096 + 123, dtype=np.uint8)
    scalar = (~T._crc_state_scalar(big, 0xFFFFFFFF)) & 0xFFFFFFFF
    assert T.crc32c(big) == scalar                                   # vectorised chunks == bytewise definition
    assert T.crc32c(big.tobytes()[777:], T.crc32c(big.tobytes()[:777])) == scalar   # incremental
    for v in (0, 1, 0xDEADBEEF, 0xFFFFFFFF, scalar):
        assert T.unmask_crc(T.mask_crc(v)) == v
    # LevelDB's documented example: Mask(crc32c("foo")) != crc, and masking twice differs again
    c = T.crc32c(b'foo')
    assert T.mask_crc(c) != c and T.mask_crc(T.mask_crc(c)) != c


def test_varints():
    assert T.put_varint(0) == b'\x00' and T.put_varint(1) == b'\x01' and T.put_varint(127) == b'\x7f'
    assert T.put_varint(128) == b'\x80\x01' and T.put_varint(300) == b'\xac\x02'
    assert T.put_varint(-1) == b'\xff' * 9 + b'\x01'                 # int64 -1 as protobuf writes it
    for v in (0, 1, 127, 128, 16383, 16384, 2 ** 32 - 1, 2 ** 63, 2 ** 64 - 1):
        got, pos = T.get_varint(T.put_varint(v) + b'tail', 0)
        assert got == v and pos == len(T.put_varint(v))
    with pytest.raises(ValueError):
        T.get_varint(b'\x80\x80', 0)


def test_snappy_decoder():
    # literal only
    assert T.snappy_decompress(b'\x05' + bytes([4 << 2]) + b'hello') == b'hello'
    # literal 'ab' + copy(offset 2, length 6) with a 1-byte-offset tag -> 'abababab' (overlapping copy)
    tag1 = ((6 - 4) << 2) | 1
    assert T.snappy_decompress(b'\x08' + bytes([1 << 2]) + b'ab' + bytes([tag1, 2])) == b'abababab'
    # 2-byte-offset copy and a long literal (length byte follows the tag)
    lit = bytes(range(70))
    stream = T.put_varint(70 + 10) + bytes([60 << 2, 69]) + lit + bytes([((10 - 1) << 2) | 2]) + struct.pack('<H', 70)
    assert T.snappy_decompress(stream) == lit + lit[:10]
    with pytest.raises(ValueError):
        T.snappy_decompress(b'\x04' + bytes([1 << 2]) + b'ab' + bytes([tag1, 9]))     # offset beyond output


def _bundle_proto_classes():
    """tensor_bundle.proto / tensor_shape.proto restated as descriptors for the real protobuf runtime."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name='aae_test_bundle.proto', package='aaetest', syntax='proto3')
    shape = fd.message_type.add(name='TensorShapeProto')
    dim = shape.nested_type.add(name='Dim')
    dim.field.add(name='size', number=1, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    dim.field.add(name='name', number=2, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    shape.field.add(name='dim', number=2, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name='.aaetest.TensorShapeProto.Dim')
    shape.field.add(name='unknown_rank', number=3, type=F.TYPE_BOOL, label=F.LABEL_OPTIONAL)
    entry = fd.message_type.add(name='BundleEntryProto')
    entry.field.add(name='dtype', number=1, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    entry.field.add(name='shape', number=2, type=F.TYPE_MESSAGE, label=F.LABEL_OPTIONAL, type_name='.aaetest.TensorShapeProto')
    entry.field.add(name='shard_id', number=3, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    entry.field.add(name='offset', number=4, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    entry.field.add(name='size', number=5, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    entry.field.add(name='crc32c', number=6, type=F.TYPE_FIXED32, label=F.LABEL_OPTIONAL)
    sl = fd.message_type.add(name='TensorSliceProto')
    ext = sl.nested_type.add(name='Extent')
    ext.field.add(name='start', number=1, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    ext.field.add(name='length', number=2, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    sl.field.add(name='extent', number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name='.aaetest.TensorSliceProto.Extent')
    entry.field.add(name='slices', number=7, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name='.aaetest.TensorSliceProto')
    header = fd.message_type.add(name='BundleHeaderProto')
    header.field.add(name='num_shards', number=1, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    header.field.add(name='endianness', number=2, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = message_factory.GetMessageClass
    return get(pool.FindMessageTypeByName('aaetest.BundleEntryProto')), get(pool.FindMessageTypeByName('aaetest.BundleHeaderProto'))


def test_bundle_entry_wire_format_against_protobuf_runtime():
    pytest.importorskip('google.protobuf')
    Entry, Header = _bundle_proto_classes()
    # protobuf -> our parser
    m = Entry(dtype=1, shard_id=0, offset=123456789012, size=4 * 5 * 5 * 3 * 128, crc32c=0xDEADBEEF)
    for d in (5, 5, 3, 128):
        m.shape.dim.add(size=d)
    e = T.BundleEntry.parse(m.SerializeToString())
    assert (e.dtype, e.shape, e.shard_id, e.offset, e.size, e.crc32c) == (1, (5, 5, 3, 128), 0, 123456789012, 38400, 0xDEADBEEF)
    scalar = Entry(dtype=9, size=8, crc32c=7)
    scalar.shape.SetInParent()                                       # rank-0 tensor: empty shape message
    es = T.BundleEntry.parse(scalar.SerializeToString())
    assert es.shape == () and es.dtype == 9 and es.size == 8
    # our serializer -> protobuf
    back = Entry.FromString(T.BundleEntry(3, (92232, 4), 0, 77, 92232 * 16, 0x01020304).serialize())
    assert back.dtype == 3 and [d.size for d in back.shape.dim] == [92232, 4]
    assert (back.offset, back.size, back.crc32c, back.shard_id) == (77, 92232 * 16, 0x01020304, 0)
    z = Entry.FromString(T.BundleEntry(1, (0, 7), 0, 0, 0, 5).serialize())     # zero-sized dim keeps its slot
    assert [d.size for d in z.shape.dim] == [0, 7]
    h = Header.FromString(b'\x08\x01\x1a\x02\x08\x01')                 # the header write_bundle emits
    assert h.num_shards == 1 and h.endianness == 0


def test_table_round_trip_multiblock_and_corruption(tmp_path):
    rng = np.random.default_rng(1)
    pairs = [(('scope/layer_%03d/kernel' % i).encode(), rng.integers(0, 256, int(rng.integers(0, 90)), dtype=np.uint8).tobytes())
             for i in range(400)]
    pairs.append((b'', b'header'))
    path = str(tmp_path / 't.index')
    T.write_table(path, pairs, block_size=512)                       # many data blocks, shared key prefixes
    assert T.read_table(path) == sorted(pairs)
    raw = bytearray(open(path, 'rb').read())
    raw[100] ^= 0x40
    open(path, 'wb').write(bytes(raw))
    with pytest.raises(ValueError, match='checksum'):
        T.read_table(path)
    assert len(T.read_table(path, verify_checksums=False)) == len(pairs)      # still parses; flipped bit lives in a value
    open(path, 'wb').write(b'not a table' * 10)
    with pytest.raises(ValueError, match='magic'):
        T.read_table(path)


def test_hand_assembled_table_with_snappy_block(tmp_path):
    """Bytes laid out by hand from the LevelDB table format description (not by write_table):
    one snappy-compressed data block, restart interval 2, prefix-compressed second key."""
    def entry(shared, key_delta, value):
        return T.put_varint(shared) + T.put_varint(len(key_delta)) + T.put_varint(len(value)) + key_delta + value
    restart2 =len(entry(0, b'apple', b'1') + entry(3, b'ly', b'22'))
    block = entry(0, b'apple', b'1') + entry(3, b'ly', b'22') + entry(0, b'banana', b'') + struct.pack('<III', 0, restart2, 2)
    snap = T.put_varint(len(block)) + bytes([(len(block) - 1) << 2]) + block            # one literal (len < 60)
    assert len(block) < 60

    def with_trailer(contents, ctype):
        return contents + bytes([ctype]) + struct.pack('<I', T.mask_crc(T.crc32c(contents + bytes([ctype]))))
    out = bytearray()
    data_off = len(out); out += with_trailer(snap, 1)
    meta = struct.pack('<II', 0, 1)
    meta_off = len(out); out += with_trailer(meta, 0)
    index = entry(0, b'c', T.put_varint(data_off) + T.put_varint(len(snap))) + struct.pack('<II', 0, 1)
    index_off = len(out); out += with_trailer(index, 0)
    footer = T.put_varint(meta_off) + T.put_varint(len(meta)) + T.put_varint(index_off) + T.put_varint(len(index))
    out += footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', T.TABLE_MAGIC)
    path = str(tmp_path / 'hand.index')
    open(path, 'wb').write(bytes(out))
    assert T.read_table(path) == [(b'apple', b'1'), (b'apply', b'22'), (b'banana', b'')]


CFG = """
[Paths]
MODEL_PATH: /nonexistent/model.ply
BACKGROUND_IMAGES_GLOB: /nonexistent/*.jpg
[Dataset]
MODEL: reconst
H: 16
W: 16
C: 3
RADIUS: 700
RENDER_DIMS: (720, 540)
K: [1075.65, 0, 720/2, 0, 1073.90, 540/2, 0, 0, 1]
[Embedding]
EMBED_BB: True
MIN_N_VIEWS: 12
NUM_CYCLO: 6
[Network]
BATCH_NORMALIZATION: True
LATENT_SPACE_SIZE: 128
NUM_FILTER: [32, 64]
STRIDES: [2, 2]
KERNEL_SIZE_ENCODER: 5
[Training]
BATCH_SIZE: 16
"""


def _tf_style_checkpoint(ckpt_dir, scope='my_exp', step=30000):
    """What tf.train.Saver leaves behind for a trained + embedded AAE: scoped encoder, BN and
    decoder variables, Adam slots, counters, the two codebook variables, a text 'checkpoint'."""
    w = synth.make_weights(seed=9, shape=(16, 16, 3), num_filter=[32, 64], strides=[2, 2], latent=128, batch_norm=True)
    rng = np.random.default_rng(3)
    emb = rng.standard_normal((72, 128)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    bbs = rng.integers(0, 500, (72, 4)).astype(np.int32)
    blob = {scope + '/' + k: v for k, v in w.items()}
    blob[scope + '/embedding_normalized'] = emb
    blob[scope + '/embed_obj_bbs_var'] = bbs
    blob[scope + '/dense_1/kernel'] = rng.standard_normal((128, 256)).astype(np.float32)      # decoder
    blob[scope + '/conv2d_2/kernel'] = rng.standard_normal((5, 5, 64, 3)).astype(np.float32)
    blob[scope + '/conv2d/kernel/Adam'] = np.zeros_like(w['conv2d/kernel'])
    blob[scope + '/conv2d/kernel/Adam_1'] = np.zeros_like(w['conv2d/kernel'])
    blob['beta1_power'] = np.float32(0.9)
    blob['global_step'] = np.int64(step)
    prefix = os.path.join(ckpt_dir, 'chkpt-%d' % step)
    T.write_bundle(prefix, blob)
    with open(os.path.join(ckpt_dir, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "chkpt-%d"\nall_model_checkpoint_paths: "chkpt-20000"\nall_model_checkpoint_paths: "chkpt-%d"\n' % (step, step))
    return prefix, w, emb, bbs


def test_bundle_round_trip_and_aae_variable_mapping(tmp_path):
    prefix, w, emb, bbs = _tf_style_checkpoint(str(tmp_path))
    r = T.BundleReader(prefix)
    assert 'my_exp/conv2d/kernel' in r.names() and r.shape('my_exp/dense/kernel') == (4 * 4 * 64, 128)
    assert r.tensor('global_step') == 30000 and r.tensor('global_step').shape == ()
    assert np.array_equal(r.tensor('my_exp/conv2d_1/kernel'), w['conv2d_1/kernel'])
    assert T.checkpoint_scopes(r.names()) == ['my_exp']
    weights, e, b = T.load_aae_variables(prefix)
    assert set(w) <= set(weights) and all(np.array_equal(weights[k], w[k]) for k in w)
    assert 'dense_1/kernel' in weights and not any('Adam' in k for k in weights) and 'global_step' not in weights
    assert np.array_equal(e, emb) and np.array_equal(b, bbs) and b.dtype == np.int32
    with pytest.raises(ValueError, match='no encoder variables'):
        T.load_aae_variables(prefix, scope='other_exp')
    # a flipped byte in the data shard is caught by the per-tensor CRC
    data = prefix + '.data-00000-of-00001'
    raw = bytearray(open(data, 'rb').read())
    raw[len(raw) // 2] ^= 1
    open(data, 'wb').write(bytes(raw))
    with pytest.raises(ValueError, match='checksum'):
        T.load_aae_variables(prefix)


def test_factory_restore_checkpoint_reads_a_tf_checkpoint(tmp_path):
    """ae_factory.restore_checkpoint(session, saver, ckpt_dir) on a directory holding a TF-style
    checkpoint (reference call site: ae_embed.py:60, aae_image.py:52-54)."""
    ckpt_dir = str(tmp_path)
    prefix, w, emb, bbs = _tf_style_checkpoint(ckpt_dir)
    st = S.get_checkpoint_state(ckpt_dir)
    assert st.model_checkpoint_path == prefix and [os.path.basename(p) for p in st.all_model_checkpoint_paths] == ['chkpt-20000', 'chkpt-30000']
    S.reset_default_graph()
    args = configparser.ConfigParser()
    args.read_string(CFG)
    with S.variable_scope('my_exp'):
        ds = factory.build_dataset('', args)
        enc = factory.build_encoder(S.Placeholder(ds.shape), args)
        cb = factory.build_codebook(enc, ds, args)
    factory.restore_checkpoint(None, None, ckpt_dir)
    assert all(np.array_equal(enc.weights[k], w[k]) for k in w)
    assert np.array_equal(cb.embedding_value(), emb) and np.array_equal(cb.embed_obj_bbs_value(), bbs)
    factory.restore_checkpoint(None, S.Saver(scope='my_exp'), ckpt_dir, at_step=30000)
    assert np.array_equal(enc.weights['dense/bias'], w['dense/bias'])
    S.reset_default_graph()


def test_saving_into_a_tensorflow_checkpoint_dir_makes_the_new_file_the_one_restored(tmp_path):
    """README flow: downloaded TF checkpoint -> ae_embed (restore, rebuild the codebook, saver.save with the same
    step, ae_embed.py:60-91) -> inference restores the directory.  The rebuilt codebook must be what comes back --
    from the newest-checkpoint restore and from an at_step restore -- not the embedding stored in the TF bundle."""
    ckpt_dir = str(tmp_path)
    prefix, w, emb, bbs = _tf_style_checkpoint(ckpt_dir)
    args = configparser.ConfigParser()
    args.read_string(CFG)

    def build():
        S.reset_default_graph()
        with S.variable_scope('my_exp'):
            ds = factory.build_dataset('', args)
            enc = factory.build_encoder(S.Placeholder(ds.shape), args)
            return enc, factory.build_codebook(enc, ds, args)

    enc, cb = build()
    saver = S.Saver()
    factory.restore_checkpoint(None, saver, ckpt_dir)
    rng = np.random.default_rng(11)
    new_emb = rng.standard_normal(emb.shape).astype(np.float32)
    new_emb /= np.linalg.norm(new_emb, axis=1, keepdims=True)
    new_bbs = rng.integers(0, 500, bbs.shape).astype(np.int32)
    cb.assign_embedding(new_emb)
    cb.assign_obj_bbs(new_bbs)
    saver.save(None, os.path.join(ckpt_dir, 'chkpt'), global_step=30000)
    st = S.get_checkpoint_state(ckpt_dir)
    assert os.path.basename(st.model_checkpoint_path) == 'chkpt-30000.npz'
    assert [os.path.basename(p) for p in st.all_model_checkpoint_paths] == ['chkpt-20000', 'chkpt-30000', 'chkpt-30000.npz']
    for at_step in (None, 30000):
        enc2, cb2 = build()
        factory.restore_checkpoint(None, S.Saver(), ckpt_dir, at_step=at_step)
        assert np.array_equal(cb2.embedding_value(), new_emb) and np.array_equal(cb2.embed_obj_bbs_value(), new_bbs), at_step
        assert all(np.array_equal(enc2.weights[k], w[k]) for k in w)
    # saving again (same and another step) keeps one index format and one entry per file
    saver_b = S.Saver()
    saver_b.save(None, os.path.join(ckpt_dir, 'chkpt'), global_step=30000)
    saver_b.save(None, os.path.join(ckpt_dir, 'chkpt'), global_step=30001)
    st = S.get_checkpoint_state(ckpt_dir)
    assert [os.path.basename(p) for p in st.all_model_checkpoint_paths] == ['chkpt-20000', 'chkpt-30000', 'chkpt-30000.npz', 'chkpt-30001.npz']
    assert os.path.basename(st.model_checkpoint_path) == 'chkpt-30001.npz'
    # an index in the old one-path-per-line form is still read and is upgraded by the next save
    with open(os.path.join(ckpt_dir, 'checkpoint'), 'w') as f:
        f.write('chkpt-30000.npz\n')
    assert os.path.basename(S.get_checkpoint_state(ckpt_dir).model_checkpoint_path) == 'chkpt-30000.npz'
    saver_b.save(None, os.path.join(ckpt_dir, 'chkpt'), global_step=30002)
    st = S.get_checkpoint_state(ckpt_dir)
    assert [os.path.basename(p) for p in st.all_model_checkpoint_paths] == ['chkpt-30000.npz', 'chkpt-30002.npz']
    S.reset_default_graph()


# ---- checkpoints laid out WITHOUT this module's writers, and the corruption suite ---------------------------------
def _snappy_literals(raw):
    """A valid snappy stream made of literal elements only (tag 60: one extra length byte), chunks of <= 200 bytes."""
    out = bytearray(T.put_varint(len(raw)))
    for a in range(0, len(raw), 200):
        chunk = raw[a:a + 200]
        out += (bytes([(len(chunk) - 1) << 2]) if len(chunk) <= 60 else bytes([60 << 2, len(chunk) - 1])) + chunk
    return bytes(out)


def _independent_table(path, pairs, per_block=3, restart_interval=2, snappy_blocks=()):
    """LevelDB table format written out from its description: prefix-compressed entries, a restart point every
    `restart_interval` entries, `per_block` entries per data block, block trailer = type byte + masked CRC-32C,
    an (empty) metaindex block, an index block keyed by each block's last key, the 48-byte footer."""
    def varint(v):
        out = bytearray()
        while v >= 0x80:
            out.append((v & 0x7F) | 0x80)
            v >>= 7
        out.append(v)
        return bytes(out)

    def block(entries, interval):
        out, restarts, last = bytearray(), [], b''
        for i, (k, v) in enumerate(entries):
            shared = 0
            if i % interval == 0:
                restarts.append(len(out))
            else:
                while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                    shared += 1
            out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
            last = k
        for r in restarts or [0]:
            out += struct.pack('<I', r)
        return bytes(out + struct.pack('<I', max(len(restarts), 1)))

    blob, index = bytearray(), []

    def emit(contents, compress):
        body, ctype = (_snappy_literals(contents), 1) if compress else (contents, 0)
        off = len(blob)
        blob.extend(body + bytes([ctype]) + struct.pack('<I', T.mask_crc(T.crc32c(body + bytes([ctype])))))
        return off, len(body)

    for bi, a in enumerate(range(0, len(pairs), per_block)):
        chunk = pairs[a:a + per_block]
        off, size = emit(block(chunk, restart_interval), bi in snappy_blocks)
        index.append((chunk[-1][0], varint(off) + varint(size)))
    moff, msize = emit(block([], 1), False)
    ioff, isize = emit(block(index, 1), False)
    footer = varint(moff) + varint(msize) + varint(ioff) + varint(isize)
    blob.extend(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57))
    open(path, 'wb').write(bytes(blob))


_NP_TO_DT = {'float32': 1, 'float64': 2, 'int32': 3, 'int64': 9, 'uint8': 4}


def _independent_checkpoint(ckpt_dir, tensors, shards=2, snappy_blocks=(1,), mutate=None, name='ind-7'):
    """{name: array} -> checkpoint-v2 files built from the protobuf runtime + _independent_table (neither
    tf_checkpoint.write_bundle, write_table nor BundleEntry.serialize is involved).  Tensors alternate between the
    data shards and are separated by gaps (tensors are offset-addressed).  mutate(name, entry message) may alter an
    entry before it is serialised; mutate('', header message) the header."""
    Entry, Header = _bundle_proto_classes()
    prefix = os.path.join(ckpt_dir, name)
    files = [open('%s.data-%05d-of-%05d' % (prefix, i, shards), 'wb') for i in range(shards)]
    offs, kv = [0] * shards, {}
    for n, (tname, a) in enumerate(sorted(tensors.items())):
        a = np.asarray(a)
        sh = n % shards
        gap = b'\xEE' * (n % 4)
        raw = a.tobytes()
        files[sh].write(gap + raw)
        m = Entry(dtype=_NP_TO_DT[a.dtype.name], shard_id=sh, offset=offs[sh] + len(gap), size=len(raw), crc32c=T.mask_crc(T.crc32c(raw)))
        m.shape.SetInParent()
        for d in a.shape:
            m.shape.dim.add(size=d)
        if mutate:
            mutate(tname, m)
        kv[tname.encode()] = m.SerializeToString()
        offs[sh] += len(gap) + len(raw)
    for f in files:
        f.close()
    h = Header(num_shards=shards)
    if mutate:
        mutate('', h)
    kv[b''] = h.SerializeToString()
    _independent_table(prefix + '.index', sorted(kv.items()), snappy_blocks=snappy_blocks)
    return prefix


def _aae_blob(scope='exp'):
    w = synth.make_weights(seed=19, shape=(16, 16, 3), num_filter=[32, 64], strides=[2, 2], latent=128, batch_norm=False)
    rng = np.random.default_rng(5)
    blob = {scope + '/' + k: v for k, v in w.items()}
    blob[scope + '/embedding_normalized'] = rng.standard_normal((72, 128)).astype(np.float32)
    blob[scope + '/embed_obj_bbs_var'] = rng.integers(0, 500, (72, 4)).astype(np.int32)
    blob['global_step'] = np.int64(7)
    blob[scope + '/conv2d/kernel/Adam'] = np.zeros_like(w['conv2d/kernel'])
    return w, blob


def test_reader_on_a_checkpoint_built_without_this_modules_writers(tmp_path):
    """Index entries from the protobuf runtime, table bytes from an independent builder (3 entries per block, restart
    interval 2, one snappy-compressed block), two data shards with gaps between the tensors."""
    pytest.importorskip('google.protobuf')
    w, blob = _aae_blob()
    prefix = _independent_checkpoint(str(tmp_path), blob)
    r = T.BundleReader(prefix)
    assert r.num_shards == 2 and sorted(r.names()) == sorted(blob)
    for k, v in blob.items():
        got = r.tensor(k)
        assert got.dtype == np.asarray(v).dtype and got.shape == np.asarray(v).shape and np.array_equal(got, v), k
    weights, emb, bbs = T.load_aae_variables(prefix)
    assert all(np.array_equal(weights[k], w[k]) for k in w) and emb.shape == (72, 128) and bbs.dtype == np.int32
    assert not any('Adam' in k for k in weights)
    # the reader's own writer must produce an index the independent expectations agree with, too: same tensors back
    T.write_bundle(str(tmp_path / 'own-7'), blob)
    own = T.BundleReader(str(tmp_path / 'own-7'))
    assert all(np.array_equal(own.tensor(k), r.tensor(k)) for k in blob)


def test_corrupt_checkpoints_raise_and_never_return_arrays(tmp_path):
    """Everything a wrong guess about the on-disk layout, a damaged download or an unsupported feature can present:
    each case must end in an exception that names the problem -- silently loading garbage weights is the failure
    mode that matters here (ae_factory.py:149-172 restores whatever the file says)."""
    pytest.importorskip('google.protobuf')
    w, blob = _aae_blob()
    d = str(tmp_path)
    good = _independent_checkpoint(d, blob)
    reference = {k: T.BundleReader(good).tensor(k) for k in blob}

    def load_all(prefix):
        r = T.BundleReader(prefix)
        return {k: r.tensor(k) for k in r.names()}

    def clone(tag):
        import shutil
        for suffix in ('.index', '.data-00000-of-00002', '.data-00001-of-00002'):
            shutil.copy(good + suffix, os.path.join(d, tag) + suffix)
        return os.path.join(d, tag)

    # 1. index truncated at every structural neighbourhood (and every 11th byte): never a successful load
    raw = open(good + '.index', 'rb').read()
    for cut in sorted(set(list(range(0, len(raw), 11)) + [len(raw) - k for k in (1, 8, 47, 48, 49, 60)])):
        p = clone('trunc')
        open(p + '.index', 'wb').write(raw[:cut])
        with pytest.raises((ValueError, IndexError, struct.error)):
            load_all(p)
    # 2. one flipped bit anywhere in the index: an exception, or (footer padding is not covered by a checksum) the same tensors
    rng = np.random.default_rng(0)
    for pos in rng.choice(len(raw), 120, replace=False):
        p = clone('flip')
        b = bytearray(raw)
        b[pos] ^= 1 << int(rng.integers(0, 8))
        open(p + '.index', 'wb').write(bytes(b))
        try:
            got = load_all(p)
        except (ValueError, IndexError, struct.error, KeyError, FileNotFoundError, UnicodeDecodeError):
            continue
        assert set(got) == set(reference) and all(np.array_equal(got[k], reference[k]) for k in got), 'bit flip at %d went unnoticed' % pos
    # 3. data shard damage: flipped byte, truncation, missing file
    p = clone('data')
    b = bytearray(open(p + '.data-00001-of-00002', 'rb').read())
    b[len(b) // 3] ^= 0x10
    open(p + '.data-00001-of-00002', 'wb').write(bytes(b))
    with pytest.raises(ValueError, match='checksum'):
        load_all(p)
    open(p + '.data-00001-of-00002', 'wb').write(bytes(b[:len(b) // 2]))
    with pytest.raises(ValueError, match='truncated|checksum'):
        load_all(p)
    os.remove(p + '.data-00001-of-00002')
    with pytest.raises(FileNotFoundError, match='data shard'):
        load_all(p)
    # 4. entries the reader does not support or that contradict themselves
    victim = 'exp/conv2d_1/kernel'

    def case(mut, match, exc=ValueError, shards=2):
        prefix = _independent_checkpoint(d, blob, shards=shards, mutate=mut, name='bad')
        with pytest.raises(exc, match=match):
            load_all(prefix)

    case(lambda n, m: setattr(m, 'dtype', 21) if n == victim else None, 'unsupported dtype enum 21')
    case(lambda n, m: m.slices.add().extent.add(start=0, length=16) if n == victim else None, 'partitioned')
    case(lambda n, m: setattr(m, 'endianness', 1) if n == '' else None, 'big-endian')
    case(lambda n, m: setattr(m, 'shard_id', 5) if n == victim else None, 'shard 5 outside')
    case(lambda n, m: setattr(m, 'size', m.size - 4) if n == victim else None, 'bytes on disk')
    case(lambda n, m: setattr(m, 'offset', m.offset + (1 << 40)) if n == victim else None, 'truncated')
    case(lambda n, m: setattr(m.shape, 'unknown_rank', True) if n == victim else None, 'unknown rank')
    case(lambda n, m: setattr(m, 'num_shards', -1) if n == '' else None, 'implausible shard count')
    case(lambda n, m: setattr(m, 'num_shards', 0) if n == '' else None, 'not found', exc=FileNotFoundError)   # proto3 drops the 0: reads as 1 shard
    case(lambda n, m: setattr(m, 'crc32c', m.crc32c ^ 1) if n == victim else None, 'checksum')
    # 5. a header that announces more shards than exist
    case(lambda n, m: setattr(m, 'num_shards', 3) if n == '' else None, 'not found', exc=FileNotFoundError)


def test_decoder_dense_name_follows_the_kernel_shape_when_the_encoder_is_variational():
    """VARIATIONAL > 0 inserts Encoder.q_sigma as 'dense_1' (encoder.py:70-80): the decoder's dense layer is then
    'dense_2'.  The loader picks the name whose kernel is [latent, h0*w0*F0]."""
    from augmentedautoencoder_amd.weights import DecoderConfig, ordered_decoder_weight_arrays
    cfg = DecoderConfig(shape=(16, 16, 3), num_filter=(32, 64), strides=(2, 2), kernel_size=5, latent_space_size=128)
    from augmentedautoencoder_amd.synth import make_decoder_weights_for
    plain = make_decoder_weights_for(cfg, seed=3)
    assert cfg.variable_names(plain)[0] == 'dense_1'
    arrays_plain = ordered_decoder_weight_arrays(plain, cfg)
    var = dict(plain)
    var['dense_2/kernel'], var['dense_2/bias'] = var.pop('dense_1/kernel'), var.pop('dense_1/bias')
    var['dense_1/kernel'] = np.zeros((4 * 4 * 64, 128), np.float32)            # q_sigma: [flatten, latent]
    var['dense_1/bias'] = np.zeros((128,), np.float32)
    assert cfg.variable_names(var)[0] == 'dense_2'
    arrays_var = ordered_decoder_weight_arrays(var, cfg)
    assert len(arrays_var) == len(arrays_plain) and all(np.array_equal(a, b) for a, b in zip(arrays_var, arrays_plain))
    broken = {k: v for k, v in var.items() if not k.startswith('dense_2/')}
    with pytest.raises(ValueError, match='shape|missing'):
        ordered_decoder_weight_arrays(broken, cfg)


def test_cli_convert_and_export(tmp_path, capsys):
    prefix, w, emb, bbs = _tf_style_checkpoint(str(tmp_path))
    npz = str(tmp_path / 'native.npz')
    assert T.main(['convert', prefix, npz]) == 0
    weights, e, b = W.load_npz(npz)
    assert all(np.array_equal(weights[k], w[k]) for k in w) and np.array_equal(e, emb) and np.array_equal(b, bbs)
    back = str(tmp_path / 'back' )
    assert T.main(['export', npz, back, 'renamed']) == 0
    w2, e2, b2 = T.load_aae_variables(back)                          # scope auto-detected
    assert all(np.array_equal(w2[k], w[k]) for k in w) and np.array_equal(e2, emb)
    assert T.main(['list', back]) == 0 and 'renamed/conv2d/kernel' in capsys.readouterr().out
    assert T.main([]) == 2
