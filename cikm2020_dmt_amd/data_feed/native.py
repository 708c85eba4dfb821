"""Native input stage: ctypes binding of libdmt_input.so (include/dmt_input.h) and the batch reader built on it.

Replaces the pure-Python path  tfrecord.read_records -> tfrecord.decode_example -> LookupTables.transform_id2index ->
SparseTensorValue.to_padded  with one C++ call per batch (TFRecord framing + CRC32C, tf.Example wire decode, vocabulary
lookup with FarmHash OOV buckets, zero-padded int32 columns), i.e. what TF's C++ runtime does for the reference's
data_feed/tfrecord_mask.py:23-84,120-158 and data_feed/index_tables.py:8-45.  The outputs are exactly the arrays
DeviceBatch.from_inputs(..., pad_to=max lengths) builds (tests/test_input_native.py compares them bit for bit).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(os.path.dirname(_HERE), "csrc_input", "libdmt_input.so")
_lib = None

EXPORTED_SYMBOLS = ["dmt_input_last_error", "dmt_input_version", "dmt_crc32c", "dmt_masked_crc32c", "dmt_fingerprint64",
                    "dmt_tfrecord_open", "dmt_tfrecord_next", "dmt_tfrecord_close", "dmt_vocab_create", "dmt_vocab_lookup",
                    "dmt_vocab_destroy", "dmt_parse_batch", "dmt_tfrecord_parse_batch"]


class InputError(RuntimeError):
    pass


class FeatureSpec(C.Structure):
    _fields_ = [("name", C.c_char_p), ("vocab", C.c_void_p), ("max_len", C.c_int32), ("idx", C.c_void_p), ("wts", C.c_void_p),
                ("lens", C.c_void_p), ("dense", C.c_void_p), ("n_wts_not_one", C.c_void_p)]


def load():
    """The library, or InputError -- there is no Python fallback behind this module (data_feed.tfrecord is the oracle side)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise InputError("%s is missing: run `make -C cikm2020_dmt_amd/csrc_input`" % _SO)
    lib = C.CDLL(_SO)
    lib.dmt_input_last_error.restype = C.c_char_p
    lib.dmt_input_version.restype = C.c_int32
    for name in ("dmt_crc32c", "dmt_masked_crc32c"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_uint64]
        getattr(lib, name).restype = C.c_uint32
    lib.dmt_fingerprint64.argtypes = [C.c_void_p, C.c_uint64]
    lib.dmt_fingerprint64.restype = C.c_uint64
    lib.dmt_tfrecord_open.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]
    lib.dmt_tfrecord_next.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.dmt_tfrecord_close.argtypes = [C.c_void_p]
    lib.dmt_tfrecord_close.restype = None
    lib.dmt_vocab_create.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_int64, C.c_int64, C.POINTER(C.c_void_p)]
    lib.dmt_vocab_lookup.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
    lib.dmt_vocab_lookup.restype = C.c_int64
    lib.dmt_vocab_destroy.argtypes = [C.c_void_p]
    lib.dmt_vocab_destroy.restype = None
    lib.dmt_parse_batch.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_int32, C.POINTER(FeatureSpec), C.c_int32, C.c_int32]
    lib.dmt_tfrecord_parse_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(FeatureSpec), C.c_int32, C.c_int32]
    _lib = lib
    return lib


def _check(rc):
    if rc < 0:
        raise InputError(load().dmt_input_last_error().decode())
    return rc


def crc32c(data: bytes) -> int:
    return int(load().dmt_crc32c(data, len(data)))


def masked_crc32c(data: bytes) -> int:
    return int(load().dmt_masked_crc32c(data, len(data)))


def fingerprint64(data: bytes) -> int:
    return int(load().dmt_fingerprint64(data, len(data)))


class Vocab:
    """index_table_from_tensor(mapping=keys, num_oov_buckets=id_size - len(keys), default_value=0)."""

    def __init__(self, keys: Sequence, id_size: int):
        lib = load()
        bs = [k if isinstance(k, (bytes, bytearray)) else str(k).encode("utf-8") for k in keys]
        arr = (C.c_char_p * len(bs))(*bs)
        lens = (C.c_uint32 * len(bs))(*[len(b) for b in bs])
        h = C.c_void_p()
        _check(lib.dmt_vocab_create(arr, lens, len(bs), int(id_size), C.byref(h)))
        self._h, self._lib = h, lib

    def lookup(self, s) -> int:
        b = s if isinstance(s, (bytes, bytearray)) else str(s).encode("utf-8")
        return int(self._lib.dmt_vocab_lookup(self._h, bytes(b), len(b)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.dmt_vocab_destroy(self._h)
                self._h = None
        except Exception:
            pass


def read_records(path: str, verify_crc: bool = True) -> Iterator[bytes]:
    lib = load()
    h = C.c_void_p()
    _check(lib.dmt_tfrecord_open(path.encode(), 1 if verify_crc else 0, C.byref(h)))
    try:
        p, n = C.c_void_p(), C.c_uint64()
        while True:
            rc = _check(lib.dmt_tfrecord_next(h, C.byref(p), C.byref(n)))
            if rc == 0:
                return
            yield C.string_at(p, n.value)
    finally:
        lib.dmt_tfrecord_close(h)


class BatchParser:
    """Parses lists of serialized tf.Example payloads into the padded columns of a DeviceBatch.

    id_features:    [(feature key, Vocab, max_len)]   -> idx int32 [B, max_len], lens int32 [B], wts float32 [B, max_len]
    float_features: [(feature key, length)]           -> float32 [B, length]   (`features`, `mask`, `label`)
    """

    def __init__(self, id_features: Sequence[Tuple[str, Vocab, int]], float_features: Sequence[Tuple[str, int]], n_threads: int = 0):
        self.id_features = list(id_features)
        self.float_features = list(float_features)
        self.n_threads = n_threads if n_threads > 0 else min(os.cpu_count() or 1, 16)
        self.pinned = False      # set True to get the batch columns in one page-locked buffer (DeviceBatch.from_columns uploads it once)
        # ring > 0: batches() hands out the SAME `ring` output buffers in turn instead of a fresh one per batch (a fresh 32 MB buffer is
        # 8 000 first-touch page faults, taken by all parser threads at once, and an munmap when it dies): a batch is then valid until
        # `ring` more have been produced -- the consumer must have uploaded / copied it by then (DeviceBatch.from_columns does)
        self.ring = 0
        self._ring_bufs, self._ring_pos = [], 0
        self._lib = load()

    def _alloc(self, B: int):
        if self.ring > 0:
            if self._ring_bufs and (self._ring_bufs[0][3] != (B, self.pinned) or len(self._ring_bufs) > self.ring):
                self._ring_bufs, self._ring_pos = [], 0
            if len(self._ring_bufs) < self.ring:
                self._ring_bufs.append(self._alloc_fresh(B) + ((B, self.pinned),))
                out, specs, keep, _key = self._ring_bufs[-1]
            else:
                out, specs, keep, _key = self._ring_bufs[self._ring_pos % self.ring]
                ev = out["__ring_slot__"].pop("event", None)       # the consumer's asynchronous upload of the batch this buffer held
                if ev is not None:
                    ev.synchronize()
                out["__wts_not_one__"][:] = 0
            self._ring_pos += 1
            return out, specs, keep
        return self._alloc_fresh(B)

    def _alloc_fresh(self, B: int):
        """All output columns of a batch in ONE host buffer (pinned when torch + a GPU are present, so the batch goes up in a single
        asynchronous copy): layout[name] = (byte offset, dtype, shape).  `out` holds numpy views into it."""
        layout, off = {}, 0

        def add(name, dtype, shape):
            nonlocal off
            off = (off + 63) // 64 * 64
            layout[name] = (off, dtype, shape)
            off += int(np.prod(shape)) * np.dtype(dtype).itemsize

        for (name, _v, T) in self.id_features:
            add(name, np.int32, (B, T)); add(name + "Wts", np.float32, (B, T)); add(name + "/lens", np.int32, (B,))
        add("__wts_not_one__", np.int32, (len(self.id_features),))
        for (name, n) in self.float_features:
            add(name, np.float32, (B, n))
        total = (off + 63) // 64 * 64
        tbuf = None
        if self.pinned:
            import torch
            tbuf = torch.empty(total, dtype=torch.uint8, pin_memory=True)
            raw = tbuf.numpy()
        else:
            raw = np.empty(total, np.uint8)
        out: Dict[str, np.ndarray] = {}
        for name, (o, dt, shape) in layout.items():
            out[name] = raw[o:o + int(np.prod(shape)) * np.dtype(dt).itemsize].view(dt).reshape(shape)
        out["__wts_not_one__"][:] = 0
        specs = (FeatureSpec * (len(self.id_features) + len(self.float_features)))()
        keep = []
        cnt = out["__wts_not_one__"]
        for i, (name, vocab, T) in enumerate(self.id_features):
            nm = name.encode()
            keep.append(nm)
            specs[i] = FeatureSpec(nm, vocab._h, T, out[name].ctypes.data, out[name + "Wts"].ctypes.data, out[name + "/lens"].ctypes.data, None,
                                   cnt.ctypes.data + 4 * i)
        for j, (name, n) in enumerate(self.float_features):
            nm = name.encode()
            keep.append(nm)
            specs[len(self.id_features) + j] = FeatureSpec(nm, None, n, None, None, None, out[name].ctypes.data, None)
        out["__buffer__"] = (tbuf if tbuf is not None else raw, layout, [f for (f, _v, _t) in self.id_features])
        out["__ring_slot__"] = {}        # ring mode: DeviceBatch.from_columns leaves the event of its upload here (waited for before reuse)
        return out, specs, keep

    def parse(self, payloads: List[bytes]) -> Dict[str, np.ndarray]:
        B = len(payloads)
        out, specs, _keep = self._alloc(B)
        ptrs = (C.c_void_p * B)(*[C.cast(C.c_char_p(p), C.c_void_p) for p in payloads])
        lens_a = (C.c_uint64 * B)(*[len(p) for p in payloads])
        _check(self._lib.dmt_parse_batch(ptrs, lens_a, B, specs, len(specs), self.n_threads))
        return out

    def batches(self, paths: Sequence[str], batch_size: int, verify_crc: bool = True, drop_remainder: bool = False):
        """Batches straight off the files, one native call each (read + crc + decode + lookup; the records never become Python
        objects).  A batch does not span files: the short batch at the end of a file is yielded unless drop_remainder."""
        lib = self._lib
        for path in paths:
            h = C.c_void_p()
            _check(lib.dmt_tfrecord_open(path.encode(), 1 if verify_crc else 0, C.byref(h)))
            try:
                while True:
                    out, specs, _keep = self._alloc(batch_size)
                    n = _check(lib.dmt_tfrecord_parse_batch(h, batch_size, specs, len(specs), self.n_threads))
                    if n == 0:
                        break
                    if n < batch_size:
                        if not drop_remainder:
                            yield {k: (v[:n] if isinstance(v, np.ndarray) and not k.startswith("__") else v) for k, v in out.items() if k != "__buffer__"}
                        break
                    yield out
            finally:
                lib.dmt_tfrecord_close(h)


def parser_for_spec(spec: dict, vocabs: Dict[str, Vocab], max_lens: Dict[str, int], n_threads: int = 0) -> BatchParser:
    """The parser of the model's input schema: every id feature of embedding_list / embedding_list_bias with the vocabulary of
    its embedding name, `features` [feature_dimension], `mask` [5], `label` [1] (tfrecord_mask.py:24-40)."""
    ids, seen = [], set()
    for (name, _rows, _dim, feat, _side) in list(spec["embedding_list"]) + list(spec["embedding_list_bias"]):
        if feat in seen:
            continue
        seen.add(feat)
        ids.append((feat, vocabs[name], int(max_lens[feat])))
    floats = [("features", int(spec["feature_dimension"])), ("mask", 5), ("label", 1)]
    return BatchParser(ids, floats, n_threads)
