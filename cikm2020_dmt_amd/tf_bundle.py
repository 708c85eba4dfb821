"""TensorFlow checkpoint V2 container ("tensor bundle": `<prefix>.index` + `<prefix>.data-00000-of-00001`), written and read WITHOUT
TensorFlow -- what `tf.train.Saver.save / restore` exchange at run_dnn.py:258-261, 301-304, 379-388 of the reference.

TensorFlow is not installed here (nor is any TF-written checkpoint shipped with the reference), so this module follows the published
format and is pinned only by its own invariants (block CRCs, footer magic, round trips): **container parity unpinned**.  Sources of the
format (tensorflow r1.12, un-vendored third party; the reference pins `tensorflow==1.12`):
  * tensorflow/core/util/tensor_bundle/tensor_bundle.cc (BundleWriter / BundleReader): one metadata table + data shards; key "" holds the
    BundleHeaderProto, every other key is a tensor name whose value is a BundleEntryProto; tensor bytes are concatenated in key order;
    `crc32c` of an entry is the MASKED CRC-32C of the tensor's bytes;
  * tensorflow/core/protobuf/tensor_bundle.proto, framework/tensor_shape.proto, framework/types.proto (DT_FLOAT = 1, DT_INT32 = 3,
    DT_INT64 = 9), framework/versions.proto;
  * tensorflow/core/lib/io/{table_builder,block_builder,format}.cc -- the LevelDB table format: prefix-compressed entries with restart
    points every 16 keys, a 5-byte block trailer (compression type 0 + masked CRC-32C of block + type), an index block of
    (separator key -> BlockHandle), an (empty) metaindex block, and a 48-byte footer ending in the magic 0xdb4775248b80fb57.
CRC-32C comes from libdmt_input.so (hardware instruction; the same code that checks TFRecord frames).
"""
from __future__ import annotations

import ctypes as C
import os
import struct
from typing import Dict, List, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
BLOCK_SIZE = 262144            # table::Options default (a reader does not depend on it)
RESTART_INTERVAL = 16
DT_FLOAT, DT_INT32, DT_INT64 = 1, 3, 9
_NP_OF = {DT_FLOAT: np.float32, DT_INT32: np.int32, DT_INT64: np.int64}
_DT_OF = {np.dtype(np.float32): DT_FLOAT, np.dtype(np.int32): DT_INT32, np.dtype(np.int64): DT_INT64}


def _crc_lib():
    from .data_feed import native
    lib = native.load()
    lib.dmt_masked_crc32c.restype = C.c_uint32
    lib.dmt_masked_crc32c.argtypes = [C.c_void_p, C.c_uint64]
    return lib


def masked_crc32c(data) -> int:
    """crc32c::Mask(crc32c::Value(data)): ((crc >> 15) | (crc << 17)) + 0xa282ead8."""
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data.view(np.uint8).reshape(-1)
    buf = np.ascontiguousarray(buf)
    return int(_crc_lib().dmt_masked_crc32c(buf.ctypes.data_as(C.c_void_p), buf.nbytes))


# ---------------------------------------------------------------------------------------------------------------- varints / protobuf
def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    n, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        n |= (b & 0x7F) << shift
        if not b & 0x80:
            return n, pos
        shift += 7


def _field(num: int, wire: int) -> bytes:
    return _varint((num << 3) | wire)


def _pb_varint(num: int, v: int) -> bytes:
    return _field(num, 0) + _varint(v) if v else b""        # proto3: default values are not written


def _pb_bytes(num: int, payload: bytes) -> bytes:
    return _field(num, 2) + _varint(len(payload)) + payload


def _shape_proto(shape) -> bytes:
    return b"".join(_pb_bytes(2, _pb_varint(1, int(s))) for s in shape)      # repeated Dim dim = 2 { int64 size = 1 }


def header_proto() -> bytes:
    """BundleHeaderProto { num_shards = 1; endianness = LITTLE (0: default, omitted); version { producer = 1 } }."""
    return _pb_varint(1, 1) + _pb_bytes(3, _pb_varint(1, 1))


def entry_proto(dtype: int, shape, offset: int, size: int, crc: int) -> bytes:
    """BundleEntryProto { dtype = 1; shape = 2; shard_id = 3 (0); offset = 4; size = 5; fixed32 crc32c = 6 }."""
    return (_pb_varint(1, dtype) + _pb_bytes(2, _shape_proto(shape)) + _pb_varint(4, offset) + _pb_varint(5, size) +
            _field(6, 5) + struct.pack("<I", crc))


def _parse_pb(buf: bytes) -> Dict[int, list]:
    out: Dict[int, list] = {}
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        num, wire = key >> 3, key & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wire == 5:
            v = struct.unpack("<I", buf[pos:pos + 4])[0]
            pos += 4
        elif wire == 1:
            v = struct.unpack("<Q", buf[pos:pos + 8])[0]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wire)
        out.setdefault(num, []).append(v)
    return out


# ---------------------------------------------------------------------------------------------------------------- table (LevelDB format)
class _BlockBuilder:
    def __init__(self):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b""

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.count % RESTART_INTERVAL == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def size(self) -> int:
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self) -> bytes:
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _write_block(f, contents: bytes) -> Tuple[int, int]:
    off = f.tell()
    trailer_type = b"\x00"                                                   # kNoCompression
    f.write(contents + trailer_type + struct.pack("<I", masked_crc32c(contents + trailer_type)))
    return off, len(contents)


def write_table(path: str, items: List[Tuple[bytes, bytes]]):
    """items: (key, value) pairs in strictly increasing bytewise key order."""
    with open(path, "wb") as f:
        index, blk = _BlockBuilder(), _BlockBuilder()

        def flush():
            nonlocal blk
            if blk.count:
                off, size = _write_block(f, blk.finish())
                index.add(blk.last, _varint(off) + _varint(size))           # separator = the block's last key (any key >= it and < the next block's first is valid)
                blk = _BlockBuilder()

        prev = None
        for key, value in items:
            if prev is not None and not key > prev:
                raise ValueError("table keys must be strictly increasing")
            prev = key
            blk.add(key, value)
            if blk.size() >= BLOCK_SIZE:
                flush()
        flush()
        meta_off, meta_size = _write_block(f, _BlockBuilder().finish())     # empty metaindex block
        idx_off, idx_size = _write_block(f, index.finish())
        footer = _varint(meta_off) + _varint(meta_size) + _varint(idx_off) + _varint(idx_size)
        footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
        f.write(footer)


def _read_block(buf: bytes, off: int, size: int) -> List[Tuple[bytes, bytes]]:
    contents, trailer = buf[off:off + size], buf[off + size:off + size + 5]
    if trailer[0] != 0:
        raise ValueError("compressed table blocks are not supported (tensor bundles are written uncompressed)")
    if struct.unpack("<I", trailer[1:5])[0] != masked_crc32c(contents + trailer[:1]):
        raise ValueError("table block checksum mismatch at offset %d" % off)
    n_restarts = struct.unpack("<I", contents[-4:])[0]
    end = len(contents) - 4 - 4 * n_restarts
    out, pos, last = [], 0, b""
    while pos < end:
        shared, pos = _read_varint(contents, pos)
        non_shared, pos = _read_varint(contents, pos)
        vlen, pos = _read_varint(contents, pos)
        key = last[:shared] + contents[pos:pos + non_shared]
        pos += non_shared
        out.append((key, contents[pos:pos + vlen]))
        pos += vlen
        last = key
    return out


def read_table(path: str) -> List[Tuple[bytes, bytes]]:
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != TABLE_MAGIC:
        raise ValueError("%s is not a table file (bad magic)" % path)
    footer = buf[-48:-8]
    pos = 0
    _m_off, pos = _read_varint(footer, pos)
    _m_size, pos = _read_varint(footer, pos)
    i_off, pos = _read_varint(footer, pos)
    i_size, pos = _read_varint(footer, pos)
    items = []
    for _sep, handle in _read_block(buf, i_off, i_size):
        off, p = _read_varint(handle, 0)
        size, _ = _read_varint(handle, p)
        items += _read_block(buf, off, size)
    return items


# ---------------------------------------------------------------------------------------------------------------- bundle
def data_file(prefix: str) -> str:
    return prefix + ".data-00000-of-00001"


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray]):
    """`tensors`: variable name (graph name without ':0', e.g. 'DnnModel/click/click-output/weights') -> array.
    Writes `<prefix>.data-00000-of-00001` and `<prefix>.index` (the data file first, both through temporaries)."""
    names = sorted(tensors, key=lambda s: s.encode())
    if any(n == "" for n in names):
        raise ValueError("the empty name is the bundle header's key")
    items = [(b"", header_proto())]
    tmp_d, tmp_i = data_file(prefix) + ".tmp", prefix + ".index.tmp"
    with open(tmp_d, "wb") as f:
        off = 0
        for n in names:
            a = np.asarray(tensors[n], order="C")            # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in _DT_OF:
                raise TypeError("tensor %s: dtype %s is not supported (float32 / int32 / int64)" % (n, a.dtype))
            raw = a.reshape(-1).view(np.uint8)
            f.write(raw.tobytes() if raw.nbytes < (1 << 20) else memoryview(raw))
            items.append((n.encode(), entry_proto(_DT_OF[a.dtype], a.shape, off, a.nbytes, masked_crc32c(raw))))
            off += a.nbytes
    write_table(tmp_i, items)
    os.replace(tmp_d, data_file(prefix))
    os.replace(tmp_i, prefix + ".index")


def read_bundle(prefix: str, verify: bool = True) -> Dict[str, np.ndarray]:
    items = read_table(prefix + ".index")
    if not items or items[0][0] != b"":
        raise ValueError("%s.index has no bundle header" % prefix)
    hdr = _parse_pb(items[0][1])
    if hdr.get(1, [1])[0] != 1:
        raise ValueError("multi-shard bundles are not supported (num_shards = %d)" % hdr[1][0])
    if hdr.get(2, [0])[0] != 0:
        raise ValueError("big-endian bundles are not supported")
    out = {}
    with open(data_file(prefix), "rb") as f:
        for key, val in items[1:]:
            e = _parse_pb(val)
            dt = e.get(1, [0])[0]
            if dt not in _NP_OF:
                raise TypeError("tensor %s: DataType %d is not supported" % (key.decode(), dt))
            shape = tuple(_parse_pb(d).get(1, [0])[0] for d in _parse_pb(e.get(2, [b""])[0]).get(2, []))
            off, size = e.get(4, [0])[0], e.get(5, [0])[0]
            f.seek(off)
            a = np.frombuffer(f.read(size), dtype=_NP_OF[dt])
            if verify and masked_crc32c(a) != e.get(6, [0])[0]:
                raise ValueError("tensor %s: checksum mismatch" % key.decode())
            out[key.decode()] = a.reshape(shape).copy()
    return out


def write_checkpoint_state(model_path: str, ckpt_name: str, all_names: List[str]):
    """The `checkpoint` text file tf.train.Saver keeps beside the bundles (CheckpointState: what tf.train.latest_checkpoint reads)."""
    with open(os.path.join(model_path, "checkpoint.tmp"), "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % ckpt_name)
        for n in all_names:
            f.write('all_model_checkpoint_paths: "%s"\n' % n)
    os.replace(os.path.join(model_path, "checkpoint.tmp"), os.path.join(model_path, "checkpoint"))
