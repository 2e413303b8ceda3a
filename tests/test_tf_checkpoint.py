"""CPU: TF V2 checkpoint reader.  TensorFlow is not installable here, so the
format is pinned by (a) constants and encodings from its public spec (table magic, masked crc32c test
vector, varints, footer size) and (b) a round trip through an independent writer that lays the file out like
BundleWriter (prefix-compressed keys, restart arrays, multi-block index, EMA names)."""
import struct

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
