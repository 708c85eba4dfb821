// libdmt_input.so: host-side input stage (see include/dmt_input.h).  Plain C++17, no dependencies.
#include "../../include/dmt_input.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <condition_variable>
#include <algorithm>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ------------------------------------------------------------------------------------------------------------ CRC-32C
struct CrcTables {
  uint32_t t[8][256];
  CrcTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
  }
};
const CrcTables& crc_tables() {
  static const CrcTables tabs;
  return tabs;
}

#if defined(__x86_64__)
__attribute__((target("sse4.2"))) uint32_t crc32c_hw(const uint8_t* p, uint64_t n) {
  uint64_t c = 0xFFFFFFFFu;
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    c = __builtin_ia32_crc32di(c, w);
    p += 8;
    n -= 8;
  }
  uint32_t c32 = (uint32_t)c;
  while (n--) c32 = __builtin_ia32_crc32qi(c32, *p++);
  return c32 ^ 0xFFFFFFFFu;
}
const bool g_have_sse42 = __builtin_cpu_supports("sse4.2");
#endif

uint32_t crc32c_impl(const uint8_t* p, uint64_t n) {
#if defined(__x86_64__)
  if (g_have_sse42) return crc32c_hw(p, n);     // the CRC32 instruction implements exactly this polynomial
#endif
  const CrcTables& T = crc_tables();
  uint32_t c = 0xFFFFFFFFu;
  while (n >= 8) {   // slicing by 8
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;
    c = T.t[7][w & 0xFF] ^ T.t[6][(w >> 8) & 0xFF] ^ T.t[5][(w >> 16) & 0xFF] ^ T.t[4][(w >> 24) & 0xFF] ^ T.t[3][(w >> 32) & 0xFF] ^
        T.t[2][(w >> 40) & 0xFF] ^ T.t[1][(w >> 48) & 0xFF] ^ T.t[0][(w >> 56) & 0xFF];
    p += 8;
    n -= 8;
  }
  while (n--) c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

inline uint32_t mask_crc(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xA282EAD8u; }

// ------------------------------------------------------------------------------------------------ FarmHash na::Hash64
// Restated from the published algorithm (Google FarmHash, namespace farmhashna, as vendored by tensorflow 1.12).
constexpr uint64_t K0 = 0xC3A5C85C97CB3127ull, K1 = 0xB492B66FBE98F273ull, K2 = 0x9AE16A3B2F90404Full;

inline uint64_t f64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint64_t f32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t rot(uint64_t v, int s) { return s == 0 ? v : (v >> s) | (v << (64 - s)); }
inline uint64_t smix(uint64_t v) { return v ^ (v >> 47); }
inline uint64_t hl16(uint64_t u, uint64_t v, uint64_t mul) {
  uint64_t a = (u ^ v) * mul;
  a ^= a >> 47;
  uint64_t b = (v ^ a) * mul;
  b ^= b >> 47;
  return b * mul;
}
uint64_t h0to16(const uint8_t* s, uint64_t n) {
  if (n >= 8) {
    const uint64_t mul = K2 + n * 2, a = f64(s) + K2, b = f64(s + n - 8);
    const uint64_t c = rot(b, 37) * mul + a, d = (rot(a, 25) + b) * mul;
    return hl16(c, d, mul);
  }
  if (n >= 4) {
    const uint64_t mul = K2 + n * 2, a = f32(s);
    return hl16(n + (a << 3), f32(s + n - 4), mul);
  }
  if (n > 0) {
    const uint8_t a = s[0], b = s[n >> 1], c = s[n - 1];
    const uint32_t y = (uint32_t)a + ((uint32_t)b << 8), z = (uint32_t)n + ((uint32_t)c << 2);
    return smix(y * K2 ^ z * K0) * K2;
  }
  return K2;
}
uint64_t h17to32(const uint8_t* s, uint64_t n) {
  const uint64_t mul = K2 + n * 2, a = f64(s) * K1, b = f64(s + 8), c = f64(s + n - 8) * mul, d = f64(s + n - 16) * K2;
  return hl16(rot(a + b, 43) + rot(c, 30) + d, a + rot(b + K2, 18) + c, mul);
}
uint64_t h33to64(const uint8_t* s, uint64_t n) {
  const uint64_t mul = K2 + n * 2, a = f64(s) * K2, b = f64(s + 8), c = f64(s + n - 8) * mul, d = f64(s + n - 16) * K2;
  const uint64_t y = rot(a + b, 43) + rot(c, 30) + d, z = hl16(y, a + rot(b + K2, 18) + c, mul);
  const uint64_t e = f64(s + 16) * mul, f = f64(s + 24), g = (y + f64(s + n - 32)) * mul, h = (z + f64(s + n - 24)) * mul;
  return hl16(rot(e + f, 43) + rot(g, 30) + h, e + rot(f + a, 18) + g, mul);
}
inline void weak32(const uint8_t* s, uint64_t a, uint64_t b, uint64_t& o0, uint64_t& o1) {
  const uint64_t w = f64(s), x = f64(s + 8), y = f64(s + 16), z = f64(s + 24);
  a += w;
  b = rot(b + a + z, 21);
  const uint64_t c = a;
  a += x;
  a += y;
  b += rot(a, 44);
  o0 = a + z;
  o1 = b + c;
}
uint64_t fingerprint64_impl(const uint8_t* s, uint64_t n) {
  if (n <= 32) return n <= 16 ? h0to16(s, n) : h17to32(s, n);
  if (n <= 64) return h33to64(s, n);
  const uint64_t seed = 81;
  uint64_t x = seed, y = seed * K1 + 113, z = smix(y * K2 + 113) * K2;
  uint64_t v0 = 0, v1 = 0, w0 = 0, w1 = 0;
  x = x * K2 + f64(s);
  const uint8_t* end = s + ((n - 1) / 64) * 64;
  const uint8_t* last64 = end + ((n - 1) & 63) - 63;
  do {
    x = rot(x + y + v0 + f64(s + 8), 37) * K1;
    y = rot(y + v1 + f64(s + 48), 42) * K1;
    x ^= w1;
    y += v0 + f64(s + 40);
    z = rot(z + w0, 33) * K1;
    weak32(s, v1 * K1, x + w0, v0, v1);
    weak32(s + 32, z + w1, y + f64(s + 16), w0, w1);
    const uint64_t t = z; z = x; x = t;
    s += 64;
  } while (s != end);
  const uint64_t mul = K1 + ((z & 0xFF) << 1);
  s = last64;
  w0 += (n - 1) & 63;
  v0 += w0;
  w0 += v0;
  x = rot(x + y + v0 + f64(s + 8), 37) * mul;
  y = rot(y + v1 + f64(s + 48), 42) * mul;
  x ^= w1 * 9;
  y += v0 * 9 + f64(s + 40);
  z = rot(z + w0, 33) * mul;
  weak32(s, v1 * mul, x + w0, v0, v1);
  weak32(s + 32, z + w1, y + f64(s + 16), w0, w1);
  const uint64_t t = z; z = x; x = t;
  return hl16(hl16(v0, w0, mul) + smix(y) * K0 + z, hl16(v1, w1, mul) + x, mul);
}

// ------------------------------------------------------------------------------------------------------ protobuf wire
struct Span {
  const uint8_t* p;
  const uint8_t* e;
};
inline bool rd_varint(Span& s, uint64_t& out) {
  uint64_t r = 0;
  int sh = 0;
  while (s.p < s.e && sh < 64) {
    const uint8_t b = *s.p++;
    r |= (uint64_t)(b & 0x7F) << sh;
    if (!(b & 0x80)) { out = r; return true; }
    sh += 7;
  }
  return false;
}
// next field of a message: field number, wire type, and for length-delimited fields the sub-span
inline bool rd_field(Span& s, uint32_t& fno, uint32_t& wt, Span& sub, uint64_t& scalar) {
  uint64_t key;
  if (!rd_varint(s, key)) return false;
  fno = (uint32_t)(key >> 3);
  wt = (uint32_t)(key & 7);
  switch (wt) {
    case 0: return rd_varint(s, scalar);
    case 1: if (s.e - s.p < 8) return false; sub.p = s.p; sub.e = s.p + 8; s.p += 8; return true;
    case 2: {
      uint64_t ln;
      if (!rd_varint(s, ln) || (uint64_t)(s.e - s.p) < ln) return false;
      sub.p = s.p; sub.e = s.p + ln; s.p += ln;
      return true;
    }
    case 5: if (s.e - s.p < 4) return false; sub.p = s.p; sub.e = s.p + 4; s.p += 4; return true;
    default: return false;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------ vocabulary
// Open addressing over (hash, key bytes in one arena): a lookup allocates nothing (the ids of a record are looked up ~700 times).
struct FlatMap {
  struct Slot { uint64_t h; uint32_t off, len; int64_t val; };
  std::vector<Slot> slots;
  std::string arena;
  uint64_t mask = 0;
  void init(size_t n) {
    size_t cap = 16;
    while (cap < n * 2 + 2) cap <<= 1;
    slots.assign(cap, Slot{0, 0, 0xFFFFFFFFu, 0});
    mask = cap - 1;
  }
  static uint64_t hash_of(const uint8_t* p, uint64_t n) { return fingerprint64_impl(p, n) | 1ull; }
  // inserts unless present (the first occurrence wins)
  void put(const uint8_t* p, uint32_t n, int64_t val) {
    const uint64_t h = hash_of(p, n);
    for (uint64_t i = h & mask;; i = (i + 1) & mask) {
      Slot& s = slots[i];
      if (s.len == 0xFFFFFFFFu) { s.h = h; s.off = (uint32_t)arena.size(); s.len = n; s.val = val; arena.append((const char*)p, n); return; }
      if (s.h == h && s.len == n && memcmp(arena.data() + s.off, p, n) == 0) return;
    }
  }
  bool get(const uint8_t* p, uint64_t n, uint64_t h, int64_t& val) const {
    for (uint64_t i = h & mask;; i = (i + 1) & mask) {
      const Slot& s = slots[i];
      if (s.len == 0xFFFFFFFFu) return false;
      if (s.h == h && s.len == n && memcmp(arena.data() + s.off, p, n) == 0) { val = s.val; return true; }
    }
  }
};

struct dmt_vocab {
  FlatMap index;
  int64_t n_keys, buckets;
};

// The file is mapped read-only: records are verified and decoded in place (no copy); the serial part of a batch is the walk over
// the 12-byte record headers.
struct dmt_tfrecord_reader {
  int fd;
  const uint8_t* base;   // mapping (nullptr for an empty file)
  uint64_t size, pos;
  int verify;
  std::string path;
  std::vector<uint64_t> offs, lens;   // batch mode: payload offsets / lengths of the records of the current batch
  uint64_t populated = 0;             // batch mode: bytes [0, populated) of the mapping have been pre-faulted (populate_ahead)
};

namespace {

struct Target {
  int feat;      // index into feats
  bool is_wts;
};

struct ParseCtx {
  const dmt_feature_spec* feats;
  int n_feats;
  FlatMap by_key;              // feature key -> index into targets
  std::vector<Target> targets;
};

int decode_float_list(Span v, float* dst, int cap, int& count, const char* key) {
  // FloatList { repeated float value = 1 [packed or not] }
  count = 0;
  Span s = v;
  while (s.p < s.e) {
    uint32_t fno, wt;
    Span sub{nullptr, nullptr};
    uint64_t sc;
    if (!rd_field(s, fno, wt, sub, sc)) return fail(DMT_IN_ERR_FORMAT, "malformed FloatList in feature '%s'", key);
    if (fno != 1 || (wt != 2 && wt != 5)) continue;
    const int n = (int)((sub.e - sub.p) / 4);
    if (count + n > cap) return fail(DMT_IN_ERR_RANGE, "feature '%s' has more than %d values", key, cap);
    memcpy(dst + count, sub.p, (size_t)n * 4);
    count += n;
  }
  return DMT_IN_OK;
}

int parse_one(const ParseCtx& cx, const uint8_t* payload, uint64_t len, int b) {
  Span ex{payload, payload + len};
  while (ex.p < ex.e) {
    uint32_t fno, wt;
    Span feats{nullptr, nullptr};
    uint64_t sc;
    if (!rd_field(ex, fno, wt, feats, sc)) return fail(DMT_IN_ERR_FORMAT, "malformed Example (record %d)", b);
    if (fno != 1 || wt != 2) continue;                  // Example.features
    while (feats.p < feats.e) {
      Span entry{nullptr, nullptr};
      if (!rd_field(feats, fno, wt, entry, sc)) return fail(DMT_IN_ERR_FORMAT, "malformed Features (record %d)", b);
      if (fno != 1 || wt != 2) continue;                // map entry
      Span key{nullptr, nullptr}, val{nullptr, nullptr};
      while (entry.p < entry.e) {
        Span sub{nullptr, nullptr};
        if (!rd_field(entry, fno, wt, sub, sc)) return fail(DMT_IN_ERR_FORMAT, "malformed map entry (record %d)", b);
        if (fno == 1 && wt == 2) key = sub;
        else if (fno == 2 && wt == 2) val = sub;
      }
      if (key.p == nullptr) continue;
      int64_t ti;
      if (!cx.by_key.get(key.p, (uint64_t)(key.e - key.p), FlatMap::hash_of(key.p, (uint64_t)(key.e - key.p)), ti)) continue;   // not a model input
      const Target* it = &cx.targets[(size_t)ti];
      const dmt_feature_spec& F = cx.feats[it->feat];
      const int T = F.max_len;
      // Feature { oneof: bytes_list = 1, float_list = 2, int64_list = 3 }
      Span fv = val;
      while (fv.p != nullptr && fv.p < fv.e) {
        Span lst{nullptr, nullptr};
        if (!rd_field(fv, fno, wt, lst, sc)) return fail(DMT_IN_ERR_FORMAT, "malformed Feature '%s' (record %d)", F.name, b);
        if (wt != 2) continue;
        if (fno == 1 && F.vocab != nullptr && !it->is_wts) {          // BytesList of ids
          int n = 0;
          Span bl = lst;
          while (bl.p < bl.e) {
            Span id{nullptr, nullptr};
            if (!rd_field(bl, fno, wt, id, sc)) return fail(DMT_IN_ERR_FORMAT, "malformed BytesList '%s' (record %d)", F.name, b);
            if (fno != 1 || wt != 2) continue;
            if (n >= T) return fail(DMT_IN_ERR_RANGE, "feature '%s' has more than %d ids (record %d)", F.name, T, b);
            F.idx[(size_t)b * T + n] = (int32_t)dmt_vocab_lookup(F.vocab, id.p, (uint64_t)(id.e - id.p));
            ++n;
          }
          F.lens[b] = n;
        } else if (fno == 2) {                                                // FloatList
          int n = 0;
          float* dst = it->is_wts ? (F.wts ? F.wts + (size_t)b * T : nullptr) : (F.vocab == nullptr ? F.dense + (size_t)b * T : nullptr);
          if (dst == nullptr) continue;
          const int rc = decode_float_list(lst, dst, T, n, F.name);
          if (rc != DMT_IN_OK) return rc;
          if (it->is_wts && F.n_wts_not_one) {
            int bad = 0;
            for (int k = 0; k < n; ++k) bad += (dst[k] != 1.0f);
            if (bad) __atomic_fetch_add(F.n_wts_not_one, bad, __ATOMIC_RELAXED);
          }
          if (F.vocab == nullptr && n != T && n != 0)
            return fail(DMT_IN_ERR_RANGE, "float feature '%s' has %d values, expected %d (record %d)", F.name, n, T, b);
        }
      }
    }
  }
  return DMT_IN_OK;
}

}  // namespace

extern "C" {

const char* dmt_input_last_error(void) { return g_err; }
int32_t dmt_input_version(void) { return 1; }

uint32_t dmt_crc32c(const void* data, uint64_t n) { return crc32c_impl((const uint8_t*)data, n); }
uint32_t dmt_masked_crc32c(const void* data, uint64_t n) { return mask_crc(crc32c_impl((const uint8_t*)data, n)); }
uint64_t dmt_fingerprint64(const void* data, uint64_t n) { return fingerprint64_impl((const uint8_t*)data, n); }

// header at r->pos: 1 = ok (payload offset / length returned, pos advanced), 0 = end of file, < 0 = error
static int next_header(dmt_tfrecord_reader* r, uint64_t& off, uint64_t& n) {
  if (r->pos == r->size) return 0;
  if (r->size - r->pos < 12) return fail(DMT_IN_ERR_FORMAT, "truncated TFRecord header in %s", r->path.c_str());
  const uint8_t* head = r->base + r->pos;
  memcpy(&n, head, 8);
  uint32_t lcrc;
  memcpy(&lcrc, head + 8, 4);
  if (r->verify && mask_crc(crc32c_impl(head, 8)) != lcrc) return fail(DMT_IN_ERR_FORMAT, "TFRecord length crc mismatch in %s", r->path.c_str());
  if (n > r->size || r->size - r->pos - 12 < n + 4) return fail(DMT_IN_ERR_FORMAT, "truncated TFRecord payload in %s", r->path.c_str());
  off = r->pos + 12;
  r->pos += 12 + n + 4;
  return 1;
}

static int check_payload(const dmt_tfrecord_reader* r, uint64_t off, uint64_t n, int b) {
  uint32_t pcrc;
  memcpy(&pcrc, r->base + off + n, 4);
  if (mask_crc(crc32c_impl(r->base + off, n)) != pcrc) return fail(DMT_IN_ERR_FORMAT, "TFRecord payload crc mismatch in %s (record %d)", r->path.c_str(), b);
  return DMT_IN_OK;
}

int dmt_tfrecord_open(const char* path, int32_t verify_crc, dmt_tfrecord_reader** out) {
  if (!path || !out) return fail(DMT_IN_ERR_ARG, "dmt_tfrecord_open: null argument");
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return fail(DMT_IN_ERR_IO, "dmt_tfrecord_open: cannot open %s", path);
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); return fail(DMT_IN_ERR_IO, "dmt_tfrecord_open: cannot stat %s", path); }
  const uint8_t* base = nullptr;
  if (st.st_size > 0) {
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { close(fd); return fail(DMT_IN_ERR_IO, "dmt_tfrecord_open: cannot map %s", path); }
    madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    base = (const uint8_t*)m;
  }
  dmt_tfrecord_reader* r = new dmt_tfrecord_reader();
  r->fd = fd; r->base = base; r->size = (uint64_t)st.st_size; r->pos = 0; r->verify = verify_crc; r->path = path;
  *out = r;
  return DMT_IN_OK;
}

int dmt_tfrecord_next(dmt_tfrecord_reader* r, const uint8_t** payload, uint64_t* len) {
  if (!r || !payload || !len) return fail(DMT_IN_ERR_ARG, "dmt_tfrecord_next: null argument");
  uint64_t off = 0, n = 0;
  const int rc = next_header(r, off, n);
  if (rc <= 0) return rc;
  if (r->verify) {
    const int c = check_payload(r, off, n, 0);
    if (c != DMT_IN_OK) return c;
  }
  *payload = r->base + off;
  *len = n;
  return 1;
}

void dmt_tfrecord_close(dmt_tfrecord_reader* r) {
  if (!r) return;
  if (r->base) munmap((void*)r->base, (size_t)r->size);
  if (r->fd >= 0) close(r->fd);
  delete r;
}

int dmt_vocab_create(const char* const* keys, const uint32_t* key_lens, int64_t n_keys, int64_t id_size, dmt_vocab** out) {
  if (!out || n_keys < 0 || (n_keys > 0 && (!keys || !key_lens))) return fail(DMT_IN_ERR_ARG, "dmt_vocab_create: bad argument");
  if (id_size < n_keys) return fail(DMT_IN_ERR_ARG, "dmt_vocab_create: id_size %lld < %lld keys", (long long)id_size, (long long)n_keys);
  dmt_vocab* v = new dmt_vocab();
  v->n_keys = n_keys;
  v->buckets = id_size - n_keys;
  v->index.init((size_t)n_keys);
  for (int64_t i = 0; i < n_keys; ++i) v->index.put((const uint8_t*)keys[i], key_lens[i], i);   // keeps the first occurrence
  *out = v;
  return DMT_IN_OK;
}

int64_t dmt_vocab_lookup(const dmt_vocab* v, const void* id, uint64_t n) {
  const uint64_t fp = fingerprint64_impl((const uint8_t*)id, n);     // one hash serves the table probe and the OOV bucket
  int64_t val;
  if (v->index.get((const uint8_t*)id, n, fp | 1ull, val)) return val;
  return v->buckets > 0 ? v->n_keys + (int64_t)(fp % (uint64_t)v->buckets) : 0;
}

void dmt_vocab_destroy(dmt_vocab* v) { delete v; }

}  // extern "C"

static int build_ctx(ParseCtx& cx, int32_t B, const dmt_feature_spec* feats, int32_t n_feats) {
  cx.feats = feats;
  cx.n_feats = n_feats;
  cx.by_key.init((size_t)n_feats * 2);
  for (int i = 0; i < n_feats; ++i) {
    const dmt_feature_spec& F = feats[i];
    if (!F.name || F.max_len <= 0) return fail(DMT_IN_ERR_ARG, "feature %d has no name / max_len", i);
    const std::string nm(F.name);
    if (F.vocab) {
      if (!F.idx || !F.lens) return fail(DMT_IN_ERR_ARG, "id feature '%s' needs idx and lens", F.name);
      const std::string wn = nm + "Wts";
      cx.by_key.put((const uint8_t*)nm.data(), (uint32_t)nm.size(), (int64_t)cx.targets.size());
      cx.targets.push_back(Target{i, false});
      cx.by_key.put((const uint8_t*)wn.data(), (uint32_t)wn.size(), (int64_t)cx.targets.size());
      cx.targets.push_back(Target{i, true});
    } else {
      if (!F.dense) return fail(DMT_IN_ERR_ARG, "float feature '%s' needs dense", F.name);
      cx.by_key.put((const uint8_t*)nm.data(), (uint32_t)nm.size(), (int64_t)cx.targets.size());
      cx.targets.push_back(Target{i, false});
    }
  }
  (void)B;
  return DMT_IN_OK;
}

// row b of every output := 0 (done by the worker that owns the row)
static void zero_row(const ParseCtx& cx, int b) {
  for (int i = 0; i < cx.n_feats; ++i) {
    const dmt_feature_spec& F = cx.feats[i];
    const size_t T = (size_t)F.max_len;
    if (F.vocab) {
      memset(F.idx + (size_t)b * T, 0, T * sizeof(int32_t));
      F.lens[b] = 0;
      if (F.wts) memset(F.wts + (size_t)b * T, 0, T * sizeof(float));
    } else {
      memset(F.dense + (size_t)b * T, 0, T * sizeof(float));
    }
  }
}

// A process-wide pool of parked worker threads.  std::thread per call cost ~60 us per thread in a process of this size (clone + stack
// mapping + join): 2 ms of a 4.7 ms batch at 32 threads, 8 ms at 128 -- the floor that made more parser threads SLOWER.  One job at a
// time (callers serialise on job_mu; the parser's calls are per-batch and short); workers sleep on a condition variable between jobs.
class WorkerPool {
 public:
  static WorkerPool& get() { static WorkerPool* p = new WorkerPool(); return *p; }     // (never destroyed: no join at process exit)
  // run task(t) for t in [0, nt) -- task 0 on the caller, the others on pool threads -- and return when all are done
  void run(int nt, const std::function<void(int)>& task) {
    std::lock_guard<std::mutex> job(job_mu_);
    {
      std::unique_lock<std::mutex> lk(mu_);
      if (pid_ != getpid()) {            // a fork()ed child has none of the parent's threads: start over
        pid_ = getpid();
        threads_.clear();                // (detached handles: nothing to join)
        epoch_ = 0;
      }
      while ((int)threads_.size() < nt - 1) {
        const int id = (int)threads_.size();
        threads_.emplace_back([this, id]() { loop(id); });
        threads_.back().detach();
      }
      task_ = &task;
      n_tasks_ = nt;
      pending_ = nt - 1;
      ++epoch_;
    }
    cv_.notify_all();
    task(0);
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this]() { return pending_ == 0; });
    task_ = nullptr;
  }

 private:
  void loop(int id) {
    unsigned long long seen = 0;
    for (;;) {
      const std::function<void(int)>* task = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&]() { return epoch_ != seen; });
        seen = epoch_;
        if (id + 1 < n_tasks_) task = task_;
      }
      if (task) {
        (*task)(id + 1);
        std::unique_lock<std::mutex> lk(mu_);
        if (--pending_ == 0) done_cv_.notify_one();
      }
    }
  }
  std::mutex job_mu_, mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> threads_;
  const std::function<void(int)>* task_ = nullptr;
  int n_tasks_ = 0, pending_ = 0;
  unsigned long long epoch_ = 0;
  pid_t pid_ = getpid();
};

// run fn(b) for b in [0, n) on nt threads; the first failure wins
template <typename F>
static int parallel_rows(int n, int n_threads, F&& fn) {
  int nt = n_threads < 1 ? 1 : n_threads;
  if (nt > n) nt = n > 0 ? n : 1;
  if (nt == 1) {
    for (int b = 0; b < n; ++b) {
      const int rc = fn(b);
      if (rc != DMT_IN_OK) return rc;
    }
    return DMT_IN_OK;
  }
  std::vector<int> rcs((size_t)nt, DMT_IN_OK);
  std::vector<std::string> errs((size_t)nt);
  const std::function<void(int)> task = [&](int t) {
    const int b0 = (int)((long long)n * t / nt), b1 = (int)((long long)n * (t + 1) / nt);
    for (int b = b0; b < b1; ++b) {
      const int rc = fn(b);
      if (rc != DMT_IN_OK) { rcs[t] = rc; errs[t] = g_err; return; }
    }
  };
  WorkerPool::get().run(nt, task);
  for (int t = 0; t < nt; ++t)
    if (rcs[t] != DMT_IN_OK) return fail(rcs[t], "%s", errs[t].c_str());
  return DMT_IN_OK;
}

extern "C" {

int dmt_parse_batch(const uint8_t* const* payloads, const uint64_t* payload_lens, int32_t B, const dmt_feature_spec* feats,
                    int32_t n_feats, int32_t n_threads) {
  if (B < 0 || n_feats < 0 || (B > 0 && (!payloads || !payload_lens)) || (n_feats > 0 && !feats)) return fail(DMT_IN_ERR_ARG, "dmt_parse_batch: bad argument");
  ParseCtx cx;
  const int rc = build_ctx(cx, B, feats, n_feats);
  if (rc != DMT_IN_OK) return rc;
  return parallel_rows(B, n_threads, [&](int b) { zero_row(cx, b); return parse_one(cx, payloads[b], payload_lens[b], b); });
}

// Pre-fault the mapping ahead of the header walk, IN PARALLEL: a page of a fresh mapping costs a fault at its first touch, the header
// walk touches one page per record serially (4 096 faults: ~2 ms of a batch), and 32 parser threads faulting the rest one page at a time
// queue on the address space's locks -- a serial floor of ~6 ms per 4 096-record batch on the 256-core host (scripts/input_scaling.py:
// 16 % parallel efficiency at 32 threads).  madvise(MADV_POPULATE_READ) maps a whole range from the page cache in one call; every
// worker takes a slice of the window.  (Advisory: a kernel without it -- EINVAL -- leaves the lazy faults in place.)
static void populate_ahead(dmt_tfrecord_reader* r, int n_threads) {
#ifdef MADV_POPULATE_READ
  const uint64_t WINDOW = 96ull << 20;                 // >= one batch of the benchmark's records (42 MB) with room to spare
  if (!r->base || r->populated >= r->size || r->populated >= r->pos + WINDOW / 2) return;
  const uint64_t lo = r->populated, hi = std::min<uint64_t>(r->size, std::max<uint64_t>(r->pos, lo) + WINDOW);
  const uint64_t PAGE = 4096;
  int nt = n_threads < 1 ? 1 : n_threads;
  const uint64_t pages = (hi - lo + PAGE - 1) / PAGE;
  if ((uint64_t)nt > pages / 64 + 1) nt = (int)(pages / 64 + 1);
  const std::function<void(int)> task = [&](int t) {
    const uint64_t p0 = lo / PAGE + pages * (uint64_t)t / (uint64_t)nt, p1 = lo / PAGE + pages * (uint64_t)(t + 1) / (uint64_t)nt;
    if (p1 <= p0) return;
    const uint64_t a = p0 * PAGE, b = std::min<uint64_t>(p1 * PAGE, (r->size + PAGE - 1) / PAGE * PAGE);
    if (b > a) (void)madvise((void*)(r->base + a), (size_t)(b - a), MADV_POPULATE_READ);
  };
  if (nt == 1) task(0); else WorkerPool::get().run(nt, task);
  r->populated = hi;
#else
  (void)r; (void)n_threads;
#endif
}

int dmt_tfrecord_parse_batch(dmt_tfrecord_reader* r, int32_t B, const dmt_feature_spec* feats, int32_t n_feats, int32_t n_threads) {
  if (!r || B <= 0 || n_feats < 0 || (n_feats > 0 && !feats)) return fail(DMT_IN_ERR_ARG, "dmt_tfrecord_parse_batch: bad argument");
  populate_ahead(r, n_threads);
  // serial part: walk up to B record headers; the payload crc and the decode run in the workers, in place on the mapping
  r->offs.clear(); r->lens.clear();
  for (int b = 0; b < B; ++b) {
    uint64_t off = 0, n = 0;
    const int rc = next_header(r, off, n);
    if (rc < 0) return rc;
    if (rc == 0) break;
    r->offs.push_back(off);
    r->lens.push_back(n);
  }
  const int nrec = (int)r->offs.size();
  ParseCtx cx;
  const int rc = build_ctx(cx, B, feats, n_feats);
  if (rc != DMT_IN_OK) return rc;
  const int prc = parallel_rows(B, n_threads, [&](int b) {
    zero_row(cx, b);                                   // rows past the last record of a short batch stay zero
    if (b >= nrec) return (int)DMT_IN_OK;
    if (r->verify) {
      const int c = check_payload(r, r->offs[b], r->lens[b], b);
      if (c != DMT_IN_OK) return c;
    }
    return parse_one(cx, r->base + r->offs[b], r->lens[b], b);
  });
  return prc != DMT_IN_OK ? prc : nrec;
}

}  // extern "C"
