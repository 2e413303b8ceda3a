"""CPU: TF V2 checkpoint reader.  TensorFlow is not installable here, so the
format is pinned by (a) constants and encodings from its public spec (table magic, masked crc32c test
vector, varints, footer size) and (b) a round trip through an independent writer that lays the file out like
BundleWriter (prefix-compressed keys, restart arrays, multi-block index, EMA names)."""
import struct

import os

import numpy as np
import pytest

from oracle import iaf_oracle as O
from pwv_amd import tf_checkpoint as T


def test_crc32c_known_answers():
    assert T.crc32c(b'123456789') == 0xe3069283                   # the standard CRC-32C check value
    assert T.crc32c(b'\x00' * 32) == 0x8a9136aa                   # RFC 3720 B.4 test vector
    assert T.crc32c(b'\xff' * 32) == 0x62a8ab43
    assert T.masked_crc(b'') == ((0 >> 15 | 0 << 17) + 0xa282ead8) & 0xffffffff


def test_varint_and_snappy():
    for v in (0, 1, 127, 128, 300, 2 ** 32 + 5, 2 ** 63 - 1):
        enc = T._put_varint(v)
        assert T._varint(enc, 0) == (v, len(enc))
    # snappy: literal "abcd" + copy(offset 4, len 8) -> "abcdabcdabcd"
    comp = bytes([12, (4 - 1) << 2]) + b'abcd' + bytes([((8 - 4) << 2) | 1, 4])
    assert T._snappy_decompress(comp) == b'abcdabcdabcd'


def test_round_trip_model_weights(tmp_path):
    cfg = O.ModelConfig(dilations=[[1, 2, 4]], n_iaf=1)
    w = O.init_weights(cfg, seed=3)
    ck = dict(w)
    for k, v in w.items():                                          # EMA shadows + optimizer slots like a real run
        ck[k + '/ExponentialMovingAverage'] = v * 0.5
        ck[k + '/Adam'] = np.zeros_like(v)
    ck['global_step'] = np.array(1234, dtype=np.int64)
    ck['learning_rate'] = np.array(2e-4, dtype=np.float32)
    prefix = str(tmp_path / 'model-1234')
    T.write_tf_checkpoint(prefix, ck, block_entries=7)              # several data blocks + a multi-entry index
    raw = open(prefix + '.index', 'rb').read()
    assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57 and len(raw) > 48
    listed = T.list_variables(prefix)
    assert set(listed) == set(ck) and listed['global_step'] == ((), 9)
    assert listed['iaf_vocoder/iaf0/scalar/dilated_stack/layer2/skip'] == ((1, 64, 128), 1)
    got = T.read_tf_checkpoint(prefix)
    assert set(got) == set(ck)
    for k in ck:
        assert got[k].dtype == ck[k].dtype and np.array_equal(got[k], ck[k]), k
    only = T.read_tf_checkpoint(prefix, name_filter=lambda n: n.endswith('/filter'))
    assert only and all(k.endswith('/filter') for k in only)
    # corruption is detected through the block crc
    bad = bytearray(raw)
    bad[10] ^= 0xff
    open(prefix + '.index', 'wb').write(bytes(bad))
    with pytest.raises(ValueError):
        T.read_tf_checkpoint(prefix)
    open(prefix + '.index', 'wb').write(b'not a table' * 10)
    with pytest.raises(ValueError):
        T.list_variables(prefix)


def test_latest_checkpoint_and_store_loading(tmp_path):
    from pwv_amd.variables import VariableStore
    w = {'a/w': np.ones((2, 3), np.float32), 'a/w/ExponentialMovingAverage': np.full((2, 3), 7, np.float32)}
    T.write_tf_checkpoint(str(tmp_path / 'model-5'), w)
    T.write_tf_checkpoint(str(tmp_path / 'model-9'), w)
    (tmp_path / 'checkpoint').write_text('model_checkpoint_path: "model-5"\nall_model_checkpoint_paths: "model-5"\n')
    assert T.latest_checkpoint(str(tmp_path)) == str(tmp_path / 'model-5')
    (tmp_path / 'checkpoint').unlink()
    assert T.latest_checkpoint(str(tmp_path)).endswith('model-9') or T.latest_checkpoint(str(tmp_path)).endswith('model-5')
    store = VariableStore(device='cpu')
    assert store.load_checkpoint(str(tmp_path / 'model-5'), use_ema=True) == 1
    assert float(store.vars['a/w'][0, 0]) == 7.0                  # EMA shadow preferred (generate.py:59-63)


def _tensors(rng):
    return {'iaf_vocoder/iaf0/scalar/dilated_stack/layer0/filter': rng.randn(2, 64, 64).astype(np.float32),
            'iaf_vocoder/iaf0/scalar/dilated_stack/layer0/filter/ExponentialMovingAverage': rng.randn(2, 64, 64).astype(np.float32),
            'iaf_vocoder/cond/dense': rng.randn(1, 80, 80).astype(np.float32),
            'big/table': rng.randn(1200, 1100).astype(np.float32),          # 5.3 MB: > 4 MB, crosses many crc chunks
            'global_step': np.array(123456, dtype=np.int64), 'empty': np.zeros((0, 3), np.float32),
            'beta1_power': np.array(0.9, dtype=np.float32)}


@pytest.mark.parametrize('num_shards,compress,block_entries', [(1, False, 64), (3, False, 2), (2, True, 3), (1, True, 1)])
def test_multi_shard_and_snappy_framed_indexes(tmp_path, num_shards, compress, block_entries):
    """What a real tensorpack ModelSaver checkpoint can look like (generate.py:55-66 restores it): several
    .data-0000k-of-0000N shards, snappy-framed index blocks, many small index blocks, a > 4 MB variable, scalars and
    empty tensors -- every tensor's crc32c is verified on read."""
    rng = np.random.RandomState(1)
    t = _tensors(rng)
    prefix = str(tmp_path / 'model-7')
    T.write_tf_checkpoint(prefix, t, block_entries=block_entries, num_shards=num_shards, compress=compress)
    assert sorted(os.path.basename(p) for p in __import__('glob').glob(prefix + '.data-*')) == \
        ['model-7.data-%05d-of-%05d' % (k, num_shards) for k in range(num_shards)]
    got = T.read_tf_checkpoint(prefix)
    assert set(got) == set(t)
    for k in t:
        assert got[k].dtype == t[k].dtype and got[k].shape == t[k].shape and np.array_equal(got[k], t[k])
    assert T.list_variables(prefix)['big/table'][0] == (1200, 1100)


def test_snappy_decoder_copy_ops():
    """The decoder paths real snappy writers use and the literal-only test writer does not: 1-byte-offset copies (incl. an
    overlapping run), 2-byte-offset copies, long literals; and malformed streams are rejected."""
    # literal "abcd"; kind-1 copy (1-byte offset) len 4 off 4; kind-1 copy len 9 off 1 (overlapping run of 'd');
    # kind-2 copy (2-byte offset) len 5 off 8
    body = bytes([(4 - 1) << 2]) + b'abcd' + bytes([((4 - 4) << 2) | 1, 4]) + bytes([((9 - 4) << 2) | 1, 1]) + \
        bytes([((5 - 1) << 2) | 2, 8, 0])
    sofar = b'abcd' + b'abcd' + b'd' * 9
    want = sofar + sofar[-8:][:5]
    assert T._snappy_decompress(bytes([len(want)]) + body) == want
    long_lit = bytes(range(256)) * 3
    assert T._snappy_decompress(T._snappy_compress_literals(long_lit)) == long_lit
    with pytest.raises(ValueError):
        T._snappy_decompress(bytes([8, (4 - 1) << 2]) + b'abcd' + bytes([((4 - 4) << 2) | 1, 9]))      # offset beyond the output
    with pytest.raises(ValueError):
        T._snappy_decompress(bytes([9, (4 - 1) << 2]) + b'abcd')                                       # declared length mismatch


def test_corruption_and_truncation_fail_cleanly(tmp_path):
    rng = np.random.RandomState(2)
    t = _tensors(rng)
    prefix = str(tmp_path / 'm')
    T.write_tf_checkpoint(prefix, t, num_shards=2)
    shard1 = prefix + '.data-00001-of-00002'
    raw = open(shard1, 'rb').read()
    # a flipped bit inside a tensor: crc32c mismatch names the variable
    bad = bytearray(raw)
    bad[len(bad) // 2] ^= 0x10
    open(shard1, 'wb').write(bytes(bad))
    with pytest.raises(ValueError, match='crc32c mismatch'):
        T.read_tf_checkpoint(prefix)
    assert 'big/table' in T.read_tf_checkpoint(prefix, verify=False)            # explicit opt-out still reads
    # a truncated shard
    open(shard1, 'wb').write(raw[:len(raw) // 3])
    with pytest.raises(ValueError, match='truncated data file'):
        T.read_tf_checkpoint(prefix)
    # a missing shard
    os.remove(shard1)
    with pytest.raises(FileNotFoundError, match='data-00001-of-00002'):
        T.read_tf_checkpoint(prefix)
    open(shard1, 'wb').write(raw)
    assert set(T.read_tf_checkpoint(prefix)) == set(t)
    # a truncated / garbled index: ValueError, never IndexError / struct.error
    idx = open(prefix + '.index', 'rb').read()
    for cut in (len(idx) - 1, len(idx) // 2, 60, 10):
        open(prefix + '.index', 'wb').write(idx[:cut])
        with pytest.raises(ValueError):
            T.read_tf_checkpoint(prefix)
    garbled = bytearray(idx)
    garbled[5] ^= 0xff
    open(prefix + '.index', 'wb').write(bytes(garbled))
    with pytest.raises(ValueError):
        T.read_tf_checkpoint(prefix)


def test_ema_preference_and_skipped_entries(tmp_path):
    """EMA shadows win when asked for (generate.py:59-63); Adam slots are filtered out by the loader."""
    from pwv_amd.variables import VariableStore
    rng = np.random.RandomState(3)
    t = _tensors(rng)
    t['iaf_vocoder/cond/dense/Adam'] = np.zeros((1, 80, 80), np.float32)
    prefix = str(tmp_path / 'ck')
    T.write_tf_checkpoint(prefix, t, num_shards=2, compress=True)
    store = VariableStore(device='cpu')
    n = store.load_checkpoint(prefix, use_ema=True)
    name = 'iaf_vocoder/iaf0/scalar/dilated_stack/layer0/filter'
    assert np.array_equal(store.vars[name].numpy(), t[name + '/ExponentialMovingAverage'])
    assert 'iaf_vocoder/cond/dense/Adam' not in store.vars and n == len(t) - 2
    store2 = VariableStore(device='cpu')
    store2.load_checkpoint(prefix, use_ema=False)
    assert np.array_equal(store2.vars[name].numpy(), t[name])


def test_reads_a_bundle_its_writer_never_touched():
    """tests/golden/tf_bundle_fixture.* was assembled byte by byte by tests/golden/make_tf_bundle_fixture.py from the
    published LevelDB table / TensorBundle formats -- that script does not import pwv_amd.tf_checkpoint -- so a shared
    misunderstanding of BundleHeaderProto / BundleEntryProto tags, the restart array, the block trailer or the footer
    between OUR writer and OUR reader cannot hide here.  (Prefix-compressed keys, an int64 scalar, an EMA shadow.)"""
    import os
    from pwv_amd.tf_checkpoint import list_variables, read_tf_checkpoint
    from pwv_amd.variables import VariableStore
    prefix = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tf_bundle_fixture')
    src = open(os.path.join(os.path.dirname(prefix), 'make_tf_bundle_fixture.py')).read()
    assert 'import pwv_amd' not in src and 'from pwv_amd' not in src and 'tf_checkpoint import' not in src
    assert list_variables(prefix) == {'global_step': ((), 9), 'iaf_vocoder/cond/dense': ((1, 2, 3), 1),
                                      'iaf_vocoder/cond/dense/ExponentialMovingAverage': ((1, 2, 3), 1)}
    got = read_tf_checkpoint(prefix, verify=True)
    assert got['global_step'].dtype == np.int64 and int(got['global_step']) == 1234
    np.testing.assert_array_equal(got['iaf_vocoder/cond/dense'], np.arange(6, dtype=np.float32).reshape(1, 2, 3) / 8)
    np.testing.assert_array_equal(got['iaf_vocoder/cond/dense/ExponentialMovingAverage'], -np.arange(6, dtype=np.float32).reshape(1, 2, 3) / 4)
    # a flipped payload byte is caught by the tensor's crc, a flipped index byte by the block trailer
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for ext in ('.index', '.data-00000-of-00001'):
            shutil.copy(prefix + ext, os.path.join(d, 'm' + ext))
        raw = bytearray(open(os.path.join(d, 'm.data-00000-of-00001'), 'rb').read())
        raw[9] ^= 1
        open(os.path.join(d, 'm.data-00000-of-00001'), 'wb').write(bytes(raw))
        with pytest.raises(ValueError, match='crc32c mismatch'):
            read_tf_checkpoint(os.path.join(d, 'm'))
    # EMA preference through the variable store (generate.py:59-63)
    store = VariableStore(device='cpu')
    store.load_checkpoint(prefix, use_ema=True)
    assert float(store.vars['iaf_vocoder/cond/dense'][0, 1, 2]) == -1.25 and store.ema_missing() == []
    store2 = VariableStore(device='cpu')
    store2.load_checkpoint(prefix, use_ema=False)
    assert float(store2.vars['iaf_vocoder/cond/dense'][0, 1, 2]) == 0.625


def test_use_ema_restore_reports_variables_without_a_shadow():
    """generate.py:59-63 maps every trainable variable of 'iaf_vocoder' to its ExponentialMovingAverage name: a checkpoint
    lacking one fails Saver.restore.  The store records the fallback so that generate() can fail the same way."""
    from pwv_amd.variables import VariableStore
    store = VariableStore(device='cpu')
    w = {'iaf_vocoder/a': np.ones((2,), np.float32), 'iaf_vocoder/a/ExponentialMovingAverage': np.zeros((2,), np.float32),
         'iaf_vocoder/b': np.ones((3,), np.float32)}
    store.load_dict(w, use_ema=True)
    assert store.ema_missing() == ['iaf_vocoder/b'] and float(store.vars['iaf_vocoder/a'][0]) == 0.0
    store.load_dict({'iaf_vocoder/b/ExponentialMovingAverage': np.full((3,), 2, np.float32)}, use_ema=True)
    assert store.ema_missing() == []


def test_use_ema_restore_covers_trainable_variables_only():
    """generate.py:57-63: with hp.train.use_ema the Saver's var_list is {ema.average_name(v): v for v in
    tf.trainable_variables('iaf_vocoder')} -- the moving statistics of a batch norm (modules.py:266) are not trainable, have no
    shadow (models.py:72-75) and are not restored at all: they keep global_variables_initializer's zeros / ones.  A valid
    reference checkpoint of a normalize='bn' model must therefore load without a "no shadow" error, and leave them alone."""
    from pwv_amd.variables import VariableStore
    bn = 'iaf_vocoder/iaf0/scalar/causal_layer/normalize/batch_normalization/'
    w = {bn + 'gamma': np.full((4,), 2, np.float32), bn + 'gamma/ExponentialMovingAverage': np.full((4,), 3, np.float32),
         bn + 'beta': np.zeros((4,), np.float32), bn + 'beta/ExponentialMovingAverage': np.ones((4,), np.float32),
         bn + 'moving_mean': np.full((4,), 5, np.float32), bn + 'moving_variance': np.full((4,), 7, np.float32)}
    store = VariableStore(device='cpu')
    for leaf, init in (('gamma', 'ones'), ('beta', 'zeros'), ('moving_mean', 'zeros'), ('moving_variance', 'ones')):
        store.get_variable(bn + leaf, [4], init)
    with pytest.warns(UserWarning, match='trainable variables only'):      # (said once: the bn statistics stay at their initial values)
        store.load_dict(w, use_ema=True)
    assert store.ema_missing() == [] and store.not_restored() == []
    assert float(store.vars[bn + 'gamma'][0]) == 3.0 and float(store.vars[bn + 'beta'][0]) == 1.0
    assert float(store.vars[bn + 'moving_mean'][0]) == 0.0 and float(store.vars[bn + 'moving_variance'][0]) == 1.0
    assert sorted(store.left_at_init) == [bn + 'moving_mean', bn + 'moving_variance']
    assert bn + 'moving_mean' not in store.trainable_variables('iaf_vocoder') and bn + 'gamma' in store.trainable_variables('iaf_vocoder')
    # without use_ema the Saver restores every variable of the graph, the statistics included (generate.py:56: var_list = None)
    store2 = VariableStore(device='cpu')
    store2.load_dict(w, use_ema=False)
    assert float(store2.vars[bn + 'moving_mean'][0]) == 5.0 and float(store2.vars[bn + 'gamma'][0]) == 2.0
    store2.get_variable(bn.replace('iaf0', 'iaf1') + 'moving_mean', [4], 'zeros')
    assert store2.not_restored() == [bn.replace('iaf0', 'iaf1') + 'moving_mean']      # ... and misses one it lacks
    # the answer is about the MOST RECENT restore: an earlier use_ema restore must not hide a non-trainable variable
    # that a later plain restore (Saver with var_list = None) would have failed on
    store3 = VariableStore(device='cpu')
    for leaf, init in (('gamma', 'ones'), ('beta', 'zeros'), ('moving_mean', 'zeros'), ('moving_variance', 'ones')):
        store3.get_variable(bn + leaf, [4], init)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        store3.load_dict(w, use_ema=True)
    assert store3.not_restored() == []
    store3.load_dict({k: v for k, v in w.items() if 'moving' not in k}, use_ema=False)
    assert store3.not_restored() == [bn + 'moving_mean', bn + 'moving_variance'] and not store3.left_at_init
