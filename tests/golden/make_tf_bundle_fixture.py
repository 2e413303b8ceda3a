#!/usr/bin/env python
"""Hand-assembles a TensorFlow V2 checkpoint (TensorBundle) byte by byte from the PUBLISHED formats -- LevelDB's table
format (doc/table_format.md: prefix-compressed entries, restart array, 5-byte block trailer with a masked CRC-32C,
48-byte footer, magic 0xdb4775248b80fb57) and tensorflow/core/protobuf/tensor_bundle.proto (BundleHeaderProto,
BundleEntryProto) -- WITHOUT importing pwv_amd.tf_checkpoint, so that the reader is pinned by bytes its own writer
never touched (tests/test_tf_checkpoint.py::test_reads_a_bundle_its_writer_never_touched).  Re-run to regenerate
tests/golden/tf_bundle_fixture.{index,data-00000-of-00001}; the values are fixed below."""
import os
import struct
import numpy as np

def varint(v):
    out = b''
    while True:
        out += bytes([(v & 0x7f) | (0x80 if v >> 7 else 0)]); v >>= 7
        if not v: return out

def crc32c(data, crc=0):                       # Castagnoli, reflected polynomial 0x82F63B78, bit by bit
    crc ^= 0xffffffff
    for b in data:
        crc ^= b
        for _ in range(8): crc = (crc >> 1) ^ (0x82f63b78 if crc & 1 else 0)
    return crc ^ 0xffffffff

mask = lambda c: (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xffffffff           # leveldb crc32c::Mask
field = lambda num, wt, payload: varint(num << 3 | wt) + payload                # protobuf key + payload
ld = lambda num, b: field(num, 2, varint(len(b)) + b)                           # length-delimited

def block(items, interval):                    # entries + restart array + count, then the trailer (type 0 = uncompressed)
    out, restarts, last = b'', [], b''
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % interval == 0: restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]: shared += 1
        out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v; last = k
    out += b''.join(struct.pack('<I', r) for r in (restarts or [0])) + struct.pack('<I', len(restarts or [0]))
    return out + b'\x00' + struct.pack('<I', mask(crc32c(out + b'\x00')))

tensors = {b'global_step': np.array(1234, np.int64),
           b'iaf_vocoder/cond/dense': np.arange(6, dtype=np.float32).reshape(1, 2, 3) / 8,
           b'iaf_vocoder/cond/dense/ExponentialMovingAverage': -np.arange(6, dtype=np.float32).reshape(1, 2, 3) / 4}
data, items = b'', [(b'', field(1, 0, varint(1)) + ld(3, field(1, 0, varint(1))))]     # header: num_shards 1, version.producer 1
for name in sorted(tensors):
    t = tensors[name]; raw = t.astype(t.dtype.newbyteorder('<')).tobytes()
    shape = b''.join(ld(2, field(1, 0, varint(d))) for d in t.shape)
    items.append((name, field(1, 0, varint({'float32': 1, 'int64': 9}[t.dtype.name])) + ld(2, shape) + field(4, 0, varint(len(data)))
                  + field(5, 0, varint(len(raw))) + field(6, 5, struct.pack('<I', mask(crc32c(raw))))))
    data += raw
blk = block(items, 16)
meta = block([], 16)
handle = lambda off, blk_: varint(off) + varint(len(blk_) - 5)
index = block([(items[-1][0], handle(0, blk))], 1)
footer = handle(len(blk), meta) + handle(len(blk) + len(meta), index)
out = blk + meta + index + footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
here = os.path.dirname(os.path.abspath(__file__))
open(os.path.join(here, 'tf_bundle_fixture.index'), 'wb').write(out)
open(os.path.join(here, 'tf_bundle_fixture.data-00000-of-00001'), 'wb').write(data)
