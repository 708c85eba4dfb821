// The index plane's sort, written for this path (no vendor library): a STABLE least-significant-digit radix sort of (key, value) pairs
// with 8-bit digits over the low `end_bit` bits of the key, and the segmentation of the sorted keys (segment id per entry, the distinct
// keys, their count).  Keys here are global embedding-row ids (table base + row, < 2^23 for the reference vocabularies, < 2^27 with the
// 100 M-row SKU table of BASELINE configs[3]), values the entry numbers, so three (four) passes do; equal rows keep their entry order,
// which is what makes the per-row sums of the backward reduction reproducible (reference semantics: base.py:87-89,115-116 -- every id
// of a feature reads / updates the row of its vocabulary index; run_dnn.py:203-207 densifies the IndexedSlices per variable).
//
// One pass = three launches, all sized by the tile count (a tile = 4096 consecutive entries = one workgroup of 256 threads):
//   rs_hist_kernel     counts[tile][256]: digit histogram of the tile (per-wavefront LDS counters fed by ballot matches: one LDS add
//                      per distinct digit of a 64-entry round, no same-address atomics -- Zipf ids make every round a pile-up);
//                      the tile's totals are also added (integer atomics) to those of its CHUNK: the tiles are cut into <= 128 chunks;
//   rs_scan_kernel     ONE workgroup over the <= 128 x 256 chunk totals: per chunk and digit the entries of this digit in earlier chunks,
//                      per digit the entries with a smaller digit (a 256-wide scan); it leaves the chunk totals zeroed for the next pass;
//   rs_scatter_kernel  a tile adds the counts of its chunk's earlier tiles (<= chunk - 1 coalesced 1 KB reads: <= 9 at 5 M entries),
//                      ranks its entries -- wavefront w owns entries [1024 w, 1024 w + 1024) of the tile, 16 rounds of 64; the rank
//                      inside the tile is (entries of the digit in earlier wavefronts) + (in earlier rounds of this wavefront) +
//                      (in lower lanes of this round): stable by construction --, reorders keys and values by digit through LDS and
//                      writes every digit's run with consecutive lanes on consecutive addresses.
// The segmentation is three more launches (count the run heads per tile, scan the tile counts in one workgroup, write).
// Nothing here allocates or synchronises; the caller owns the workspace (query with ws == nullptr, as before).
#include "dmt_common.h"

namespace {

constexpr int RS_NT = 256, RS_KPT = 16, RS_TILE = RS_NT * RS_KPT, RS_RADIX = 256, RS_MAXCHUNK = 128;
constexpr int RS_WAVES = RS_NT / 64, RS_SUB = RS_TILE / RS_WAVES;       // entries per wavefront

// lanes of the wavefront (among `valid` ones) whose 8-bit digit equals this lane's
__device__ __forceinline__ unsigned long long rs_match(unsigned d, bool valid) {
  unsigned long long m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long bb = __ballot(bit);
    m &= bit ? bb : ~bb;
  }
  return m;
}

__global__ __launch_bounds__(RS_NT) void rs_hist_kernel(const uint32_t* __restrict__ keys, long long n, int shift, uint32_t mask,
                                                        int* __restrict__ counts, int* __restrict__ csum, int chunk) {
  __shared__ int wcnt[RS_WAVES][RS_RADIX];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < RS_WAVES * RS_RADIX; i += RS_NT) (&wcnt[0][0])[i] = 0;
  const long long base = (long long)blockIdx.x * RS_TILE + (long long)w * RS_SUB;
  uint32_t key[RS_KPT];
#pragma unroll
  for (int r = 0; r < RS_KPT; ++r) {                  // every request of the tile up front: ONE exposed round trip per workgroup
    const long long e = base + r * 64 + lane;
    key[r] = e < n ? keys[e] : 0u;
  }
  __syncthreads();
  volatile int* wc = wcnt[w];
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < RS_KPT; ++r) {
    const bool valid = base + r * 64 + lane < n;
    const unsigned d = valid ? ((key[r] >> shift) & mask) : 0u;
    const unsigned long long m = rs_match(d, valid);
    if (valid && (m & lt) == 0ull) wc[d] = wc[d] + (int)__popcll(m);      // the lowest lane of every distinct digit: distinct addresses
  }
  __syncthreads();
  int s = 0;
#pragma unroll
  for (int k = 0; k < RS_WAVES; ++k) s += wcnt[k][tid];
  counts[(long long)blockIdx.x * RS_RADIX + tid] = s;
  if (s != 0) atomicAdd(csum + ((int)blockIdx.x / chunk) * RS_RADIX + tid, s);      // the chunk's digit totals (integer sums: any order, one result)
}

// One workgroup of 1024 threads over the chunk totals csum[chunk][256] (<= 128 chunks): thread (d = tid & 255, q = tid >> 8) owns the chunks of
// quarter q of digit d.  Out: cexcl[chunk][d] = entries of digit d in earlier chunks, dbase[d] = entries with a smaller digit; csum is left
// zeroed for the next pass.
__global__ __launch_bounds__(1024) void rs_scan_kernel(int* __restrict__ csum, int nchunks, int* __restrict__ cexcl, int* __restrict__ dbase) {
  __shared__ int qsum[4][RS_RADIX];
  __shared__ int tot[RS_RADIX];
  const int tid = threadIdx.x, d = tid & 255, q = tid >> 8;
  const int per = (nchunks + 3) / 4;
  const int c0 = q * per, c1 = (c0 + per < nchunks) ? c0 + per : nchunks;
  int v[RS_MAXCHUNK / 4];
#pragma unroll
  for (int i = 0; i < RS_MAXCHUNK / 4; ++i) v[i] = (c0 + i < c1) ? csum[(c0 + i) * RS_RADIX + d] : 0;      // (independent requests: one round trip)
  int s = 0;
#pragma unroll
  for (int i = 0; i < RS_MAXCHUNK / 4; ++i) s += v[i];
  qsum[q][d] = s;
  __syncthreads();
  int run = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) run += (k < q) ? qsum[k][d] : 0;
  if (q == 0) tot[d] = qsum[0][d] + qsum[1][d] + qsum[2][d] + qsum[3][d];
#pragma unroll
  for (int i = 0; i < RS_MAXCHUNK / 4; ++i)
    if (c0 + i < c1) { cexcl[(c0 + i) * RS_RADIX + d] = run; run += v[i]; csum[(c0 + i) * RS_RADIX + d] = 0; }
  __syncthreads();
  // exclusive scan of the 256 digit totals (Hillis-Steele over LDS: 8 steps)
  int t = tid < RS_RADIX ? tot[tid] : 0;
  const int mine = t;
  for (int off = 1; off < RS_RADIX; off <<= 1) {
    int add = 0;
    if (tid < RS_RADIX && tid >= off) add = tot[tid - off];
    __syncthreads();
    if (tid < RS_RADIX) { t += add; tot[tid] = t; }
    __syncthreads();
  }
  if (tid < RS_RADIX) dbase[tid] = t - mine;
}

template <bool IOTA>
__global__ __launch_bounds__(RS_NT) void rs_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                           uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, long long n,
                                                           int shift, uint32_t mask, const int* __restrict__ counts,
                                                           const int* __restrict__ cexcl, const int* __restrict__ dbase, int chunk) {
  __shared__ uint32_t sk[RS_TILE];
  __shared__ uint32_t sv[RS_TILE];
  __shared__ int wcnt[RS_WAVES][RS_RADIX];
  __shared__ int goff[RS_RADIX];       // first output position of the tile's entries with digit d
  __shared__ int lstart[RS_RADIX];     // first position of digit d inside the reordered tile
  __shared__ int wsum[RS_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tile = blockIdx.x;
  for (int i = tid; i < RS_WAVES * RS_RADIX; i += RS_NT) (&wcnt[0][0])[i] = 0;
  {
    // global offset of digit `tid`: the chunk's base + this digit's entries in the chunk's earlier tiles
    const int c = tile / chunk;
    int off = dbase[tid] + cexcl[c * RS_RADIX + tid];
#pragma unroll 8
    for (int t = c * chunk; t < tile; ++t) off += counts[(long long)t * RS_RADIX + tid];
    goff[tid] = off;
  }
  __syncthreads();
  const long long base = (long long)tile * RS_TILE + (long long)w * RS_SUB;
  uint32_t key[RS_KPT], val[RS_KPT];
  int rank[RS_KPT];
#pragma unroll
  for (int r = 0; r < RS_KPT; ++r) {
    const long long e = base + r * 64 + lane;
    key[r] = e < n ? keys_in[e] : 0xFFFFFFFFu;
    if constexpr (IOTA) val[r] = (uint32_t)e;
    else val[r] = e < n ? vals_in[e] : 0u;
  }
  volatile int* wc = wcnt[w];
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < RS_KPT; ++r) {
    const bool valid = base + r * 64 + lane < n;
    const unsigned d = valid ? ((key[r] >> shift) & mask) : 0u;
    const unsigned long long m = rs_match(d, valid);
    const int old = valid ? wc[d] : 0;
    const int below = (int)__popcll(m & lt);
    if (valid && below == 0) wc[d] = old + (int)__popcll(m);
    rank[r] = old + below;
  }
  __syncthreads();
  {
    // per digit: exclusive prefix over the wavefronts (in place), the tile's count, then an exclusive scan of the counts over the digits
    int run = 0;
#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k) { const int v = wcnt[k][tid]; wcnt[k][tid] = run; run += v; }
    // block-wide exclusive scan of `run` (256 values: wave scan by shuffles + 4 wave totals)
    int incl = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int wbase = 0;
#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k) wbase += (k < w) ? wsum[k] : 0;
    lstart[tid] = wbase + incl - run;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_KPT; ++r) {
    if (base + r * 64 + lane < n) {
      const unsigned d = (key[r] >> shift) & mask;
      const int lp = lstart[d] + wcnt[w][d] + rank[r];
      sk[lp] = key[r];
      sv[lp] = val[r];
    }
  }
  __syncthreads();
  const long long left = n - (long long)tile * RS_TILE;
  const int tile_n = left < RS_TILE ? (int)left : RS_TILE;
#pragma unroll 4
  for (int j = 0; j < RS_KPT; ++j) {
    const int i = j * RS_NT + tid;
    if (i < tile_n) {
      const uint32_t k = sk[i];
      const unsigned d = (k >> shift) & mask;
      const long long dst = (long long)goff[d] + (i - lstart[d]);
      keys_out[dst] = k;
      vals_out[dst] = sv[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------ segmentation of sorted keys
// thread t of a tile owns the 16 consecutive entries [16 t, 16 t + 16): the scan stays in entry order
__device__ __forceinline__ int sh_flags(const uint32_t* __restrict__ k, long long n, long long e0, uint32_t (&key)[RS_KPT], unsigned& flags) {
  uint32_t prev = e0 > 0 && e0 - 1 < n ? k[e0 - 1] : 0u;
  if (e0 + RS_KPT <= n) {
#pragma unroll
    for (int j = 0; j < RS_KPT; j += 4) {
      const uint4 v = *reinterpret_cast<const uint4*>(k + e0 + j);
      key[j] = v.x; key[j + 1] = v.y; key[j + 2] = v.z; key[j + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < RS_KPT; ++j) key[j] = e0 + j < n ? k[e0 + j] : 0u;
  }
  int cnt = 0;
  flags = 0u;
#pragma unroll
  for (int j = 0; j < RS_KPT; ++j) {
    const bool head = (e0 + j < n) && (e0 + j == 0 || key[j] != prev);
    flags |= head ? (1u << j) : 0u;
    cnt += head ? 1 : 0;
    prev = key[j];
  }
  return cnt;
}

__global__ __launch_bounds__(RS_NT) void sh_count_kernel(const uint32_t* __restrict__ k, long long n, int* __restrict__ tile_heads) {
  __shared__ int wsum[RS_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  uint32_t key[RS_KPT];
  unsigned flags;
  int c = sh_flags(k, n, (long long)blockIdx.x * RS_TILE + (long long)tid * RS_KPT, key, flags);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if (lane == 0) wsum[w] = c;
  __syncthreads();
  if (tid == 0) {
    int s = 0;
#pragma unroll
    for (int i = 0; i < RS_WAVES; ++i) s += wsum[i];
    tile_heads[blockIdx.x] = s;
  }
}

// exclusive scan of the tile counts, in place, by one workgroup
__global__ __launch_bounds__(1024) void sh_scan_kernel(int* __restrict__ tile_heads, int ntiles) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int t0 = 0; t0 < ntiles; t0 += 1024) {
    const int t = t0 + tid;
    const int v = t < ntiles ? tile_heads[t] : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int wbase = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) wbase += (i < w) ? wsum[i] : 0;
    const int carry = carry_s;
    if (t < ntiles) tile_heads[t] = carry + wbase + incl - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + wbase + incl;
    __syncthreads();
  }
}

__global__ __launch_bounds__(RS_NT) void sh_write_kernel(const uint32_t* __restrict__ k, long long n, uint32_t invalid, const int* __restrict__ tile_off,
                                                         int* __restrict__ seg, uint32_t* __restrict__ uniq, int* __restrict__ n_uniq) {
  __shared__ int wsum[RS_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long long e0 = (long long)blockIdx.x * RS_TILE + (long long)tid * RS_KPT;
  uint32_t key[RS_KPT];
  unsigned flags;
  const int c = sh_flags(k, n, e0, key, flags);
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int s = tile_off[blockIdx.x] + incl - c;            // heads before this thread's first entry
#pragma unroll
  for (int i = 0; i < RS_WAVES; ++i) s += (i < w) ? wsum[i] : 0;
  int out[RS_KPT];
#pragma unroll
  for (int j = 0; j < RS_KPT; ++j) {
    if (flags & (1u << j)) {
      if (e0 + j < n) uniq[s] = key[j];
      ++s;
    }
    out[j] = s - 1;                                    // 0-based id of the run the entry belongs to
    if (e0 + j == n - 1) n_uniq[0] = (key[j] >= invalid) ? s - 1 : s;      // (the invalid key sorts last: its run is not a row)
  }
  if (e0 + RS_KPT <= n) {
#pragma unroll
    for (int j = 0; j < RS_KPT; j += 4) *reinterpret_cast<int4*>(seg + e0 + j) = make_int4(out[j], out[j + 1], out[j + 2], out[j + 3]);
  } else {
#pragma unroll
    for (int j = 0; j < RS_KPT; ++j) if (e0 + j < n) seg[e0 + j] = out[j];
  }
}

struct SortPlan { int ntiles, chunk, nchunks, npass; size_t tmp_keys, tmp_vals, counts, csum, cexcl, dbase, total; };
static SortPlan sort_plan(int64_t n, int end_bit) {
  SortPlan p;
  p.ntiles = (int)cdiv64(n > 0 ? n : 1, RS_TILE);
  p.chunk = (p.ntiles + RS_MAXCHUNK - 1) / RS_MAXCHUNK;
  p.nchunks = (p.ntiles + p.chunk - 1) / p.chunk;
  p.npass = (end_bit + 7) / 8;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  p.tmp_keys = 0;
  p.tmp_vals = p.tmp_keys + (p.npass > 1 ? al((size_t)n * 4) : 0);
  p.counts = p.tmp_vals + (p.npass > 1 ? al((size_t)n * 4) : 0);
  p.csum = p.counts + al((size_t)p.ntiles * RS_RADIX * 4);
  p.cexcl = p.csum + al((size_t)RS_MAXCHUNK * RS_RADIX * 4);
  p.dbase = p.cexcl + al((size_t)RS_MAXCHUNK * RS_RADIX * 4);
  p.total = p.dbase + al((size_t)RS_RADIX * 4);
  return p;
}

__global__ __launch_bounds__(256) void rs_zero_kernel(int* __restrict__ x, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) x[i] = 0;
}

}  // namespace

// keys_out / vals_out = (keys_in, vals_in) stably sorted by the low end_bit bits of the key.  vals_in == nullptr: the values are the
// entry numbers 0 .. n - 1 (what dmt_embgrad_keys would write).  Inputs are not modified; outputs must not alias them.
extern "C" int dmt_sort_pairs(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                              int64_t n, int32_t end_bit, void* ws, uint64_t* ws_bytes, void* stream) {
  DMT_CHECK_ARG(ws_bytes != nullptr, "dmt_sort_pairs: ws_bytes is null");
  DMT_CHECK_ARG(end_bit > 0 && end_bit <= 32, "dmt_sort_pairs: bad end_bit");
  DMT_CHECK_ARG(n >= 0 && n < (1ll << 31) - RS_TILE, "dmt_sort_pairs: n out of range (32-bit positions)");
  const SortPlan p = sort_plan(n, end_bit);
  if (ws == nullptr) { *ws_bytes = p.total; return DMT_OK; }
  DMT_CHECK_ARG(*ws_bytes >= p.total, "dmt_sort_pairs: workspace too small (%llu < %llu)", (unsigned long long)*ws_bytes, (unsigned long long)p.total);
  if (n == 0) return DMT_OK;
  DMT_CHECK_ARG(keys_in && keys_out && vals_out, "dmt_sort_pairs: null argument");
  DMT_CHECK_ARG((((uintptr_t)ws) & 15) == 0, "dmt_sort_pairs: workspace must be 16-byte aligned");
  DMT_CHECK_ARG(keys_in != keys_out && vals_in != vals_out, "dmt_sort_pairs: outputs must not alias the inputs");
  hipStream_t st = (hipStream_t)stream;
  unsigned char* w8 = (unsigned char*)ws;
  uint32_t* tk = (uint32_t*)(w8 + p.tmp_keys);
  uint32_t* tv = (uint32_t*)(w8 + p.tmp_vals);
  int* counts = (int*)(w8 + p.counts);
  int* csum = (int*)(w8 + p.csum);
  int* cexcl = (int*)(w8 + p.cexcl);
  int* dbase = (int*)(w8 + p.dbase);
  const uint32_t* ki = keys_in;
  const uint32_t* vi = vals_in;
  hipLaunchKernelGGL(rs_zero_kernel, dim3(32), dim3(256), 0, st, csum, p.nchunks * RS_RADIX);     // (every pass's scan leaves it zeroed for the next)
  for (int ps = 0; ps < p.npass; ++ps) {
    // the last pass lands in the caller's buffers; the ones before it alternate so that it does
    const bool to_out = ((p.npass - 1 - ps) & 1) == 0;
    uint32_t* ko = to_out ? keys_out : tk;
    uint32_t* vo = to_out ? vals_out : tv;
    const int shift = 8 * ps;
    const int bits = end_bit - shift < 8 ? end_bit - shift : 8;
    const uint32_t mask = (1u << bits) - 1u;
    hipLaunchKernelGGL(rs_hist_kernel, dim3(p.ntiles), dim3(RS_NT), 0, st, ki, (long long)n, shift, mask, counts, csum, p.chunk);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, st, csum, p.nchunks, cexcl, dbase);
    if (vi == nullptr)
      hipLaunchKernelGGL((rs_scatter_kernel<true>), dim3(p.ntiles), dim3(RS_NT), 0, st, ki, vi, ko, vo, (long long)n, shift, mask, (const int*)counts, (const int*)cexcl, (const int*)dbase, p.chunk);
    else
      hipLaunchKernelGGL((rs_scatter_kernel<false>), dim3(p.ntiles), dim3(RS_NT), 0, st, ki, vi, ko, vo, (long long)n, shift, mask, (const int*)counts, (const int*)cexcl, (const int*)dbase, p.chunk);
    ki = ko;
    vi = vo;
  }
  DMT_CHECK_LAUNCH("dmt_sort_pairs");
  return DMT_OK;
}

extern "C" int dmt_segment_heads(const uint32_t* sorted_keys, int64_t n, uint32_t invalid_key, int32_t* seg_id,
                                 uint32_t* uniq_keys, int32_t* n_uniq, void* ws, uint64_t* ws_bytes, void* stream) {
  DMT_CHECK_ARG(ws_bytes != nullptr, "dmt_segment_heads: ws_bytes is null");
  DMT_CHECK_ARG(n >= 0 && n < (1ll << 31) - RS_TILE, "dmt_segment_heads: n out of range");
  const int ntiles = (int)cdiv64(n > 0 ? n : 1, RS_TILE);
  const size_t need = (((size_t)ntiles * 4) + 255) & ~(size_t)255;
  if (ws == nullptr) { *ws_bytes = need; return DMT_OK; }
  DMT_CHECK_ARG(*ws_bytes >= need, "dmt_segment_heads: workspace too small");
  DMT_CHECK_ARG(sorted_keys && seg_id && uniq_keys && n_uniq && n > 0, "dmt_segment_heads: null argument");
  DMT_CHECK_ARG(((((uintptr_t)sorted_keys) | ((uintptr_t)seg_id) | ((uintptr_t)ws)) & 15) == 0, "dmt_segment_heads: keys, seg_id and workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int* tile_heads = (int*)ws;
  hipLaunchKernelGGL(sh_count_kernel, dim3(ntiles), dim3(RS_NT), 0, st, sorted_keys, (long long)n, tile_heads);
  hipLaunchKernelGGL(sh_scan_kernel, dim3(1), dim3(1024), 0, st, tile_heads, ntiles);
  hipLaunchKernelGGL(sh_write_kernel, dim3(ntiles), dim3(RS_NT), 0, st, sorted_keys, (long long)n, invalid_key, (const int*)tile_heads, seg_id, uniq_keys, n_uniq);
  DMT_CHECK_LAUNCH("dmt_segment_heads");
  return DMT_OK;
}
