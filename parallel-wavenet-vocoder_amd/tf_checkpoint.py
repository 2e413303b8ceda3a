"""Reader (and minimal writer) for TensorFlow "V2" checkpoints without TensorFlow
(SURVEY.md section 8 f-1).  The reference saves with tensorpack's ModelSaver / tf.train.Saver
(train.py:51) and restores by variable name, preferring EMA shadows (generate.py:55-66).

Format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table*):
  <prefix>.index             an SSTable in LevelDB's table format:
        data blocks | metaindex block | index block | 48-byte footer
        block   = entries, restart offsets (fixed32 each), n_restarts (fixed32); then a 5-byte trailer
                  (1 byte compression: 0 none / 1 snappy, 4 bytes masked crc32c)
        entry   = varint32 shared, varint32 non_shared, varint32 value_len, key suffix, value
        footer  = BlockHandle(metaindex) BlockHandle(index) padded to 40 bytes + magic 0xdb4775248b80fb57 (LE)
        key ""  -> BundleHeaderProto {1: num_shards, 2: endianness, 3: version}
        key name-> BundleEntryProto  {1: dtype, 2: TensorShapeProto{2: dim{1: size}}, 3: shard_id, 4: offset,
                                      5: size, 6: crc32c (fixed32), 7: slices}
  <prefix>.data-SSSSS-of-NNNNN  raw little-endian tensor bytes at [offset, offset + size)
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Iterator, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
           10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


# ---------------------------------------------------------------------------------- primitives
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tbl = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
            tbl.append(c)
        _CRC_TABLE = tbl
    return _CRC_TABLE


def _crc_update(c: int, data: bytes) -> int:
    tbl = _crc_table()
    for b in data:
        c = tbl[(c ^ b) & 0xff] ^ (c >> 8)
    return c


def _gf2_times(mat, vec: int) -> int:
    out, i = 0, 0
    while vec:
        if vec & 1:
            out ^= mat[i]
        vec >>= 1
        i += 1
    return out


def _zeros_operator(nbytes: int):
    """The 32x32 GF(2) matrix that advances a raw CRC-32C register over `nbytes` zero bytes (zlib's crc32_combine
    construction with the Castagnoli polynomial)."""
    odd = [0x82f63b78] + [1 << i for i in range(31)]            # operator for one zero BIT
    sq = lambda m: [_gf2_times(m, m[i]) for i in range(32)]
    op = sq(sq(sq(odd)))                                        # one zero BYTE
    result = None
    n = nbytes
    while n:
        if n & 1:
            result = op if result is None else [_gf2_times(op, result[i]) for i in range(32)]
        n >>= 1
        if n:
            op = sq(op)
    return result if result is not None else [1 << i for i in range(32)]


def crc32c(data) -> int:
    """CRC-32C (Castagnoli) of bytes / a uint8 buffer.  Large buffers are cut into equal chunks whose registers advance in
    lockstep as numpy vectors (one table lookup per byte POSITION instead of per byte) and are then combined with the
    zero-advance operator -- pure Python would take ~1 s per MB."""
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
    n = buf.size
    if n < 1 << 14:
        return _crc_update(0xffffffff, buf.tobytes()) ^ 0xffffffff
    lanes = int(min(4096, n // 2048))
    chunk = n // lanes
    tbl = np.array(_crc_table(), dtype=np.uint32)
    body = buf[:lanes * chunk].reshape(lanes, chunk)
    regs = np.zeros(lanes, dtype=np.uint32)
    regs[0] = 0xffffffff                                        # only the first chunk carries the initial register
    for i in range(chunk):
        regs = tbl[(regs ^ body[:, i]) & 0xff] ^ (regs >> 8)
    op = _zeros_operator(chunk)
    c = int(regs[0])
    for k in range(1, lanes):
        c = _gf2_times(op, c) ^ int(regs[k])                    # crc(A || B) register = advance(reg A, |B|) xor reg0(B)
    return _crc_update(c, buf[lanes * chunk:].tobytes()) ^ 0xffffffff


def masked_crc(data) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xffffffff


def _snappy_decompress(src: bytes) -> bytes:
    n, pos = _varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += src[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = src[pos] | (src[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError('snappy: copy offset out of range')
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('snappy: length mismatch')
    return bytes(out)


def _snappy_compress_literals(data: bytes) -> bytes:
    """A valid snappy stream made of literals only (what the writer emits for compress=True; real writers also emit
    copies -- the decoder's copy paths are exercised by hand-built streams in tests/test_tf_checkpoint.py)."""
    out = bytearray(_put_varint(len(data)))
    pos = 0
    while pos < len(data):
        ln = min(len(data) - pos, 65536)
        if ln <= 60:
            out.append((ln - 1) << 2)
        else:
            nb = (max(ln - 1, 1).bit_length() + 7) // 8
            out.append((59 + nb) << 2)
            out += (ln - 1).to_bytes(nb, 'little')
        out += data[pos:pos + ln]
        pos += ln
    return bytes(out)


# ---------------------------------------------------------------------------------- table reader
def _read_block(f: bytes, offset: int, size: int, verify: bool) -> bytes:
    if offset < 0 or size < 0 or offset + size + 5 > len(f):
        raise ValueError('checkpoint index: block [%d, +%d) runs past the end of the file (truncated?)' % (offset, size))
    raw = f[offset:offset + size]
    ctype = f[offset + size]
    if verify:
        want = struct.unpack('<I', f[offset + size + 1:offset + size + 5])[0]
        if masked_crc(f[offset:offset + size + 1]) != want:
            raise ValueError('checkpoint index: block crc mismatch at %d' % offset)
    if ctype == 0:
        return raw
    if ctype == 1:
        return _snappy_decompress(raw)
    raise ValueError('checkpoint index: unknown block compression %d' % ctype)


def _block_entries(block: bytes) -> Iterator[Tuple[bytes, bytes]]:
    n_restarts = struct.unpack('<I', block[-4:])[0]
    limit = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b''
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _table_items(data: bytes, verify: bool) -> Iterator[Tuple[bytes, bytes]]:
    if len(data) < 48 or struct.unpack('<Q', data[-8:])[0] != TABLE_MAGIC:
        raise ValueError('not a TensorFlow checkpoint index (bad table magic)')
    footer = data[-48:]
    _, p = _varint(footer, 0)            # metaindex handle (unused)
    _, p = _varint(footer, p)
    ioff, p = _varint(footer, p)
    isize, p = _varint(footer, p)
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, q = _varint(handle, 0)
        bsize, q = _varint(handle, q)
        for kv in _block_entries(_read_block(data, boff, bsize, verify)):
            yield kv


# ---------------------------------------------------------------------------------- protos
def _proto_fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield field, wt, v


def _parse_entry(buf: bytes):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'slices': 0, 'crc': None}
    for field, wt, v in _proto_fields(buf):
        if field == 1:
            e['dtype'] = v
        elif field == 2:
            for f2, _, v2 in _proto_fields(v):
                if f2 == 2:                                  # dim
                    size = 0
                    for f3, _, v3 in _proto_fields(v2):
                        if f3 == 1:
                            size = v3 - (1 << 64) if v3 >= (1 << 63) else v3
                    e['shape'].append(size)
        elif field == 3:
            e['shard_id'] = v
        elif field == 4:
            e['offset'] = v
        elif field == 5:
            e['size'] = v
        elif field == 6:
            e['crc'] = struct.unpack('<I', v)[0]
        elif field == 7:
            e['slices'] += 1
    return e


def _index_items(prefix: str, verify: bool):
    with open(prefix + '.index', 'rb') as f:
        data = f.read()
    try:
        return list(_table_items(data, verify))
    except (IndexError, struct.error) as e:       # a varint / fixed32 read ran off a truncated or garbled block
        raise ValueError('corrupt TensorFlow checkpoint index %s.index (%s)' % (prefix, e))


def list_variables(prefix: str, verify: bool = True) -> Dict[str, Tuple[Tuple[int, ...], int]]:
    out = {}
    for key, val in _index_items(prefix, verify):
        if key:
            e = _parse_entry(val)
            out[key.decode()] = (tuple(e['shape']), e['dtype'])
    return out


def read_tf_checkpoint(prefix: str, verify: bool = True, name_filter=None, skipped: Optional[list] = None) -> Dict[str, np.ndarray]:
    """All (or the filtered) tensors of checkpoint `<prefix>` as {variable name: ndarray}.  `verify`: check the block
    checksums of the index and every tensor's crc32c.  Entries this reader does not materialise (string tensors,
    partitioned variables) are appended to `skipped` -- callers that need every variable must look at it (generate()
    checks coverage against the model's variables)."""
    num_shards, entries = 1, {}
    for key, val in _index_items(prefix, verify):
        if not key:
            for field, _, v in _proto_fields(val):
                if field == 1:
                    num_shards = v
                if field == 2 and v != 0:
                    raise ValueError('big-endian checkpoints are not supported')
        else:
            try:
                entries[key.decode()] = _parse_entry(val)
            except (IndexError, ValueError, struct.error) as e:
                raise ValueError('corrupt TensorFlow checkpoint index %s.index: entry %r (%s)' % (prefix, key, e))
    shards = {}
    out = {}
    for name, e in entries.items():
        if name_filter is not None and not name_filter(name):
            continue
        if e['dtype'] not in _DTYPES or e['slices']:
            if skipped is not None:
                skipped.append(name)
            continue                                          # strings / partitioned variables: not on this path
        sid = e['shard_id']
        if not 0 <= sid < num_shards:
            raise ValueError('checkpoint %s: %s lives in shard %d of %d' % (prefix, name, sid, num_shards))
        if sid not in shards:
            path = '%s.data-%05d-of-%05d' % (prefix, sid, num_shards)
            if not os.path.exists(path):
                raise FileNotFoundError('checkpoint %s: data shard %s is missing' % (prefix, os.path.basename(path)))
            shards[sid] = np.memmap(path, dtype=np.uint8, mode='r') if os.path.getsize(path) else np.zeros(0, np.uint8)
        dt = np.dtype(_DTYPES[e['dtype']])
        want = int(np.prod(e['shape'], dtype=np.int64)) * dt.itemsize if e['shape'] else dt.itemsize
        if e['size'] != want:
            raise ValueError('checkpoint %s: %s has %d bytes for shape %s %s' % (prefix, name, e['size'], e['shape'], dt))
        if e['offset'] + e['size'] > shards[sid].size:
            raise ValueError('checkpoint %s: %s [%d, +%d) runs past the end of shard %d (%d bytes): truncated data file'
                             % (prefix, name, e['offset'], e['size'], sid, shards[sid].size))
        raw = np.asarray(shards[sid][e['offset']:e['offset'] + e['size']])
        if verify and e['crc'] is not None and masked_crc(raw) != e['crc']:
            raise ValueError('checkpoint %s: crc32c mismatch in the bytes of %s (corrupt data shard %d)' % (prefix, name, sid))
        arr = np.frombuffer(raw.tobytes(), dtype=dt.newbyteorder('<'))
        out[name] = arr.reshape(e['shape']).astype(dt)
    return out


def latest_checkpoint(logdir: str) -> Optional[str]:
    """tf.train.latest_checkpoint: the prefix named by `<logdir>/checkpoint`, else the newest *.index."""
    state = os.path.join(logdir, 'checkpoint')
    if os.path.exists(state):
        with open(state) as f:
            for line in f:
                if line.startswith('model_checkpoint_path:'):
                    p = line.split(':', 1)[1].strip().strip('"')
                    p = p if os.path.isabs(p) else os.path.join(logdir, p)
                    if os.path.exists(p + '.index'):
                        return p
    import glob
    cands = sorted(glob.glob(os.path.join(logdir, '*.index')), key=os.path.getmtime)
    return cands[-1][:-len('.index')] if cands else None


# ---------------------------------------------------------------------------------- writer (tests / export)
def _build_block(items, restart_interval: int = 16) -> bytes:
    out, restarts, last, count = bytearray(), [], b'', 0
    for key, val in items:
        if count % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            while shared < min(len(last), len(key)) and last[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(val)) + key[shared:] + val
        last = key
        count += 1
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_tf_checkpoint(prefix: str, tensors: Dict[str, np.ndarray], block_entries: int = 64, num_shards: int = 1,
                        compress: bool = False) -> None:
    """Write {name: array} as a V2 checkpoint laid out like tensorflow::BundleWriter's output: tensors dealt round-robin
    over `num_shards` data files, every entry with its masked crc32c, index blocks optionally snappy-framed."""
    def field(n, wt, payload):
        return _put_varint((n << 3) | wt) + payload

    items = [(b'', field(1, 0, _put_varint(num_shards)) + field(2, 0, _put_varint(0)) +
              field(3, 2, _put_varint(2) + field(1, 0, _put_varint(1))))]
    files = [open('%s.data-%05d-of-%05d' % (prefix, k, num_shards), 'wb') for k in range(num_shards)]
    offsets = [0] * num_shards
    try:
        for i, name in enumerate(sorted(tensors)):
            arr = np.asarray(tensors[name], order='C')          # (ascontiguousarray would turn scalars into shape (1,))
            raw = arr.astype(arr.dtype.newbyteorder('<')).tobytes()
            sid = i % num_shards
            shape = b''.join(field(2, 2, (lambda d: _put_varint(len(d)) + d)(field(1, 0, _put_varint(int(s))))) for s in arr.shape)
            entry = (field(1, 0, _put_varint(_DTYPE_CODES[arr.dtype])) + field(2, 2, _put_varint(len(shape)) + shape) +
                     (field(3, 0, _put_varint(sid)) if sid else b'') +
                     field(4, 0, _put_varint(offsets[sid])) + field(5, 0, _put_varint(len(raw))) +
                     field(6, 5, struct.pack('<I', masked_crc(raw))))
            items.append((name.encode(), entry))
            files[sid].write(raw)
            offsets[sid] += len(raw)
    finally:
        for f in files:
            f.close()
    out = bytearray()

    def emit(block: bytes) -> bytes:
        body, ctype = (_snappy_compress_literals(block), b'\x01') if compress else (block, b'\x00')
        handle = _put_varint(len(out)) + _put_varint(len(body))
        out.extend(body + ctype + struct.pack('<I', masked_crc(body + ctype)))
        return handle

    index_items = []
    for i in range(0, len(items), block_entries):
        chunk = items[i:i + block_entries]
        index_items.append((chunk[-1][0] + b'\x00', emit(_build_block(chunk))))
    meta = emit(_build_block([]))
    index = emit(_build_block(index_items, restart_interval=1))
    footer = meta + index
    out.extend(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC))
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out))
