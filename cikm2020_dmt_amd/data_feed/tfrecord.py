"""TFRecord framing + tf.Example wire codec (host side, pure Python + numpy).

Keeps the reference's on-disk schema (SURVEY.md §8b "TFRecord schema (keep)"):
the reader side of DMT_code/data_feed/tfrecord_mask.py:23-84 (parse_single_line)
consumes `label` float[], `mask` float[5], `features` float[615], `header` bytes[],
every id feature as a bytes list and `<name>Wts` as a float list.

Framing (TFRecord): u64 length | u32 masked_crc32c(length) | payload | u32 masked_crc32c(payload).
tf.Example wire format:
    Example  { Features features = 1; }
    Features { map<string, Feature> feature = 1; }      (map entry: key = 1, value = 2)
    Feature  { oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3; } }
    BytesList { repeated bytes value = 1; }  FloatList { repeated float value = 1 [packed]; }
    Int64List { repeated int64 value = 1 [packed]; }
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Union

import numpy as np

# ----------------------------------------------------------------------------- CRC32C
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        poly = 0x82F63B78
        tab = np.zeros(256, dtype=np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ poly if (c & 1) else (c >> 1)
            tab[i] = c
        _CRC_TABLE = [int(x) for x in tab]
    return _CRC_TABLE


def crc32c(data: bytes) -> int:
    tab = _crc_table()
    c = 0xFFFFFFFF
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ----------------------------------------------------------------------------- framing
def read_records(path: str, verify_crc: bool = False) -> Iterator[bytes]:
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if len(head) == 0:
                return
            if len(head) < 12:
                raise IOError("truncated TFRecord header in %s" % path)
            (length,) = struct.unpack("<Q", head[:8])
            if verify_crc:
                (lcrc,) = struct.unpack("<I", head[8:12])
                if masked_crc32c(head[:8]) != lcrc:
                    raise IOError("TFRecord length crc mismatch in %s" % path)
            payload = f.read(length)
            tail = f.read(4)
            if len(payload) < length or len(tail) < 4:
                raise IOError("truncated TFRecord payload in %s" % path)
            if verify_crc:
                (pcrc,) = struct.unpack("<I", tail)
                if masked_crc32c(payload) != pcrc:
                    raise IOError("TFRecord payload crc mismatch in %s" % path)
            yield payload


def write_records(path: str, payloads) -> int:
    n = 0
    with open(path, "wb") as f:
        for p in payloads:
            head = struct.pack("<Q", len(p))
            f.write(head)
            f.write(struct.pack("<I", masked_crc32c(head)))
            f.write(p)
            f.write(struct.pack("<I", masked_crc32c(p)))
            n += 1
    return n


# ----------------------------------------------------------------------------- protobuf wire
def _read_varint(buf: bytes, pos: int):
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not (b & 0x80):
            return result, pos
        shift += 7


def _write_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _fields(buf: bytes):
    """Yield (field_number, wire_type, value) over one message."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield fno, wt, v


FeatureValue = Union[List[bytes], np.ndarray]


def _decode_feature(buf: bytes) -> FeatureValue:
    for fno, wt, v in _fields(buf):
        if fno == 1:  # BytesList
            return [bv for f2, _, bv in _fields(v) if f2 == 1]
        if fno == 2:  # FloatList
            chunks = []
            for f2, wt2, fv in _fields(v):
                if f2 != 1:
                    continue
                chunks.append(np.frombuffer(fv, dtype="<f4"))  # packed (wt 2) or single fixed32 (wt 5)
            return np.concatenate(chunks) if chunks else np.zeros(0, np.float32)
        if fno == 3:  # Int64List
            vals = []
            for f2, wt2, iv in _fields(v):
                if f2 != 1:
                    continue
                if wt2 == 0:
                    vals.append(iv)
                else:
                    p = 0
                    while p < len(iv):
                        x, p = _read_varint(iv, p)
                        vals.append(x)
            arr = np.array(vals, dtype=np.uint64).astype(np.int64)
            return arr
    return []  # empty oneof


def decode_example(payload: bytes) -> Dict[str, FeatureValue]:
    out: Dict[str, FeatureValue] = {}
    for fno, _, feats in _fields(payload):
        if fno != 1:
            continue
        for f2, _, entry in _fields(feats):
            if f2 != 1:
                continue
            key = None
            val: FeatureValue = []
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    key = v.decode("utf-8")
                elif f3 == 2:
                    val = _decode_feature(v)
            if key is not None:
                out[key] = val
    return out


def _ld(fno: int, body: bytes) -> bytes:
    return _write_varint((fno << 3) | 2) + _write_varint(len(body)) + body


def encode_example(features: Dict[str, FeatureValue]) -> bytes:
    entries = bytearray()
    for key in sorted(features.keys()):
        val = features[key]
        if isinstance(val, (list, tuple)) and (len(val) == 0 or isinstance(val[0], (bytes, bytearray))):
            inner = b"".join(_ld(1, bytes(b)) for b in val)
            feat = _ld(1, inner)
        else:
            arr = np.asarray(val)
            if arr.dtype.kind == "f":
                feat = _ld(2, _ld(1, arr.astype("<f4").tobytes()) if arr.size else b"")
            else:
                packed = b"".join(_write_varint(int(x) & 0xFFFFFFFFFFFFFFFF) for x in arr.reshape(-1))
                feat = _ld(3, _ld(1, packed) if arr.size else b"")
        entry = _ld(1, key.encode("utf-8")) + _ld(2, feat)
        entries += _ld(1, entry)
    return _ld(1, bytes(entries))
