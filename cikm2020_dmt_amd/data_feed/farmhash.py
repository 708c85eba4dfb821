"""FarmHash Fingerprint64 (farmhashna::Hash64), the hash behind TF's string_to_hash_bucket_fast, which
tf.contrib.lookup.index_table_from_tensor(num_oov_buckets=...) uses for out-of-vocabulary ids
(/root/reference/DMT_code/data_feed/index_tables.py:27-28).

Third-party algorithm restated from Google FarmHash (farmhash.cc, namespace farmhashna), the version vendored by
tensorflow==1.12; TensorFlow is absent here, so this is pinned only by FarmHash's structural constants and the
documented example tf.strings.to_hash_bucket_fast(["Hello", "TensorFlow", "2.x"], 3) == [0, 2, 2]
(tests/test_host.py).
"""
from __future__ import annotations

M64 = (1 << 64) - 1
K0 = 0xC3A5C85C97CB3127
K1 = 0xB492B66FBE98F273
K2 = 0x9AE16A3B2F90404F


def _f64(s: bytes, i: int) -> int:
    return int.from_bytes(s[i:i + 8], "little")


def _f32(s: bytes, i: int) -> int:
    return int.from_bytes(s[i:i + 4], "little")


def _rot(v: int, sh: int) -> int:
    return v if sh == 0 else ((v >> sh) | (v << (64 - sh))) & M64


def _smix(v: int) -> int:
    return v ^ (v >> 47)


def _hl16(u: int, v: int, mul: int) -> int:
    a = ((u ^ v) * mul) & M64
    a ^= a >> 47
    b = ((v ^ a) * mul) & M64
    b ^= b >> 47
    return (b * mul) & M64


def _h0to16(s: bytes) -> int:
    n = len(s)
    if n >= 8:
        mul = (K2 + n * 2) & M64
        a = (_f64(s, 0) + K2) & M64
        b = _f64(s, n - 8)
        c = (_rot(b, 37) * mul + a) & M64
        d = ((_rot(a, 25) + b) * mul) & M64
        return _hl16(c, d, mul)
    if n >= 4:
        mul = (K2 + n * 2) & M64
        a = _f32(s, 0)
        return _hl16((n + (a << 3)) & M64, _f32(s, n - 4), mul)
    if n > 0:
        a, b, c = s[0], s[n >> 1], s[n - 1]
        y = (a + (b << 8)) & 0xFFFFFFFF
        z = (n + (c << 2)) & 0xFFFFFFFF
        return (_smix(((y * K2) & M64) ^ ((z * K0) & M64)) * K2) & M64
    return K2


def _h17to32(s: bytes) -> int:
    n = len(s)
    mul = (K2 + n * 2) & M64
    a = (_f64(s, 0) * K1) & M64
    b = _f64(s, 8)
    c = (_f64(s, n - 8) * mul) & M64
    d = (_f64(s, n - 16) * K2) & M64
    return _hl16((_rot((a + b) & M64, 43) + _rot(c, 30) + d) & M64, (a + _rot((b + K2) & M64, 18) + c) & M64, mul)


def _h33to64(s: bytes) -> int:
    n = len(s)
    mul = (K2 + n * 2) & M64
    a = (_f64(s, 0) * K2) & M64
    b = _f64(s, 8)
    c = (_f64(s, n - 8) * mul) & M64
    d = (_f64(s, n - 16) * K2) & M64
    y = (_rot((a + b) & M64, 43) + _rot(c, 30) + d) & M64
    z = _hl16(y, (a + _rot((b + K2) & M64, 18) + c) & M64, mul)
    e = (_f64(s, 16) * mul) & M64
    f = _f64(s, 24)
    g = ((y + _f64(s, n - 32)) * mul) & M64
    h = ((z + _f64(s, n - 24)) * mul) & M64
    return _hl16((_rot((e + f) & M64, 43) + _rot(g, 30) + h) & M64, (e + _rot((f + a) & M64, 18) + g) & M64, mul)


def _weak32(s: bytes, i: int, a: int, b: int):
    w, x, y, z = _f64(s, i), _f64(s, i + 8), _f64(s, i + 16), _f64(s, i + 24)
    a = (a + w) & M64
    b = _rot((b + a + z) & M64, 21)
    c = a
    a = (a + x) & M64
    a = (a + y) & M64
    b = (b + _rot(a, 44)) & M64
    return (a + z) & M64, (b + c) & M64


def fingerprint64(s: bytes) -> int:
    n = len(s)
    if n <= 32:
        return _h0to16(s) if n <= 16 else _h17to32(s)
    if n <= 64:
        return _h33to64(s)
    seed = 81
    x = seed
    y = (seed * K1 + 113) & M64
    z = (_smix((y * K2 + 113) & M64) * K2) & M64
    v = (0, 0)
    w = (0, 0)
    x = (x * K2 + _f64(s, 0)) & M64
    end = ((n - 1) // 64) * 64
    last64 = end + ((n - 1) & 63) - 63
    i = 0
    while True:
        x = (_rot((x + y + v[0] + _f64(s, i + 8)) & M64, 37) * K1) & M64
        y = (_rot((y + v[1] + _f64(s, i + 48)) & M64, 42) * K1) & M64
        x ^= w[1]
        y = (y + v[0] + _f64(s, i + 40)) & M64
        z = (_rot((z + w[0]) & M64, 33) * K1) & M64
        v = _weak32(s, i, (v[1] * K1) & M64, (x + w[0]) & M64)
        w = _weak32(s, i + 32, (z + w[1]) & M64, (y + _f64(s, i + 16)) & M64)
        z, x = x, z
        i += 64
        if i == end:
            break
    mul = (K1 + ((z & 0xFF) << 1)) & M64
    i = last64
    w = ((w[0] + ((n - 1) & 63)) & M64, w[1])
    v = ((v[0] + w[0]) & M64, v[1])
    w = ((w[0] + v[0]) & M64, w[1])
    x = (_rot((x + y + v[0] + _f64(s, i + 8)) & M64, 37) * mul) & M64
    y = (_rot((y + v[1] + _f64(s, i + 48)) & M64, 42) * mul) & M64
    x ^= (w[1] * 9) & M64
    y = (y + v[0] * 9 + _f64(s, i + 40)) & M64
    z = (_rot((z + w[0]) & M64, 33) * mul) & M64
    v = _weak32(s, i, (v[1] * mul) & M64, (x + w[0]) & M64)
    w = _weak32(s, i + 32, (z + w[1]) & M64, (y + _f64(s, i + 16)) & M64)
    z, x = x, z
    return _hl16((_hl16(v[0], w[0], mul) + (_smix(y) * K0) + z) & M64, (_hl16(v[1], w[1], mul) + x) & M64, mul)


def to_hash_bucket_fast(s: bytes, num_buckets: int) -> int:
    return fingerprint64(s) % num_buckets
