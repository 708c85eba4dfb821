// MFMA GEMM with fused epilogue (bias / relu / relu-gradient gate / residual), batched, split-K.
//   bf16 operands: v_mfma_f32_32x32x16_bf16      fp32 operands: v_mfma_f32_32x32x2_f32 (exact fma chain)
// Workgroup = 4 wavefronts (2x2), tile 128x128, each wave 64x64 = 2x2 MFMA tiles of 32x32 (64 fp32 acc VGPRs).
// Operands are staged global -> registers -> LDS; the next K tile's global loads are in flight while the current one
// is multiplied.  Either operand may be
//   k-contiguous   (mode 0): LDS image [row][k], row stride BK + one 16-byte pad, fragments by ds_read_b128;
//   row-contiguous (mode 1): bf16: LDS image [k][row] copied as it lies in memory (coalesced 256-byte rows, 16-byte LDS
//                  writes, row stride 320 B so four k rows fall in four different bank quarters) and transposed for free
//                  by ds_read_b64_tr_b16 when the fragment is read; fp32: transposed on the LDS write;
//   strided        (mode 2): scalar loads,
// so the same kernel serves Y = X W^T-shadow, dX = dY W and dW = X^T dY (reduction over the strided dim of both).
#include "dmt_common.h"
#include <atomic>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int BM = 128, BN = 128, NT = 256;
constexpr int PERSIST_GRID = 256 * 3;   // 3 resident workgroups per CU (a multiple of the 8 XCDs)
constexpr int LDT = 160;   // bf16 row-contiguous LDS image: [BK][LDT] (128 rows + 32 pad: 320 B = 256 + 64)

template <typename T> struct Cfg;
template <> struct Cfg<bf16_t> { static constexpr int EPV = 8, BK = 64, LDK = 72, NV = 4; };   // 128 x 64 x 2 B / 16 B / 256 thr
template <> struct Cfg<float>  { static constexpr int EPV = 4, BK = 16, LDK = 20, NV = 2; };

struct GemmArgs {
  int M, N, K;
  const void* A; long long a_rs, a_cs;
  const void* B; long long b_rs, b_cs;
  void* C; long long ldc;
  const float* bias;
  int act_ncols;
  const void* gate; long long ldg;
  const void* resid; long long ldr;
  int a_ones_row;
  float* c_last;
  int split_k, batch, accumulate;
  long long a_bs, b_bs, c_bs, bias_bs, gate_bs, resid_bs, clast_bs;
  int a_mode, b_mode;   // 0: k-contiguous vectors, 1: row-contiguous vectors, 2: scalar
  int k_per_split;
  int out_f32;
  int vec_epi;
  int fast_ok;   // operand extents fit the 32-bit buffer offsets
  int gx, gy, gz, inner, panels;   // XCD-aware block remap (see decode_tile)
  int total_blocks;                // work ids (the grid is persistent: min(total_blocks, 3 workgroups per CU))
};

union Vec16 {
  uint4 u;
  float f[4];
  bf16_t h[8];
};

template <typename T> __device__ __forceinline__ T one_val();
template <> __device__ __forceinline__ float one_val<float>() { return 1.0f; }
template <> __device__ __forceinline__ bf16_t one_val<bf16_t>() { return (bf16_t)0x3F80; }

template <typename T> __device__ __forceinline__ void vset(Vec16& v, int i, T x);
template <> __device__ __forceinline__ void vset<float>(Vec16& v, int i, float x) { v.f[i] = x; }
template <> __device__ __forceinline__ void vset<bf16_t>(Vec16& v, int i, bf16_t x) { v.h[i] = x; }
template <typename T> __device__ __forceinline__ T vget(const Vec16& v, int i);
template <> __device__ __forceinline__ float vget<float>(const Vec16& v, int i) { return v.f[i]; }
template <> __device__ __forceinline__ bf16_t vget<bf16_t>(const Vec16& v, int i) { return v.h[i]; }

// Load this thread's two 16-byte pieces of a [128 rows] x [BK] operand tile.
//   elem(r, k) = P[r*rs + k*cs];  rows >= R_real read as 0 (or 1 for the ones row), k >= k_end read as 0.
// MODE 0: k contiguous (cs == 1), 16-byte vectors along k;  MODE 1: rows contiguous (rs == 1), vectors along rows,
// lanes along k;  MODE 2: arbitrary strides, scalar loads.  All register indexing is compile-time (no scratch).
template <typename T, int MODE>
__device__ __forceinline__ void load_tile(Vec16 (&reg)[Cfg<T>::NV], const T* __restrict__ P, long long rs, long long cs,
                                          int row0, int k0, int R_real, int k_end, int ones_row, int tid) {
  constexpr int EPV = Cfg<T>::EPV, BK = Cfg<T>::BK;
#pragma unroll
  for (int p = 0; p < Cfg<T>::NV; ++p) {
    const int v = tid + p * NT;
    Vec16 x;
    x.u = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (MODE == 0) {
      const int r = row0 + v / (BK / EPV);
      const int k = k0 + (v % (BK / EPV)) * EPV;
      if (r < R_real) {
        const T* src = P + (long long)r * rs + k;
        if (k + EPV <= k_end) {
          x.u = *reinterpret_cast<const uint4*>(src);
        } else {
#pragma unroll
          for (int i = 0; i < EPV; ++i)
            if (k + i < k_end) vset<T>(x, i, src[i]);
        }
      } else if (r == ones_row) {
#pragma unroll
        for (int i = 0; i < EPV; ++i)
          if (k + i < k_end) vset<T>(x, i, one_val<T>());
      }
    } else if constexpr (MODE == 1) {
      // bf16: 16 consecutive threads cover one k row of the tile (128 rows = 256 B)
      const int k = k0 + (sizeof(T) == 2 ? (v / 16) : (v % BK));
      const int r = row0 + (sizeof(T) == 2 ? (v % 16) * EPV : (v / BK) * EPV);
      if (k < k_end) {
        const T* src = P + (long long)k * cs + r;
        if (r + EPV <= R_real) {
          x.u = *reinterpret_cast<const uint4*>(src);
        } else {
#pragma unroll
          for (int i = 0; i < EPV; ++i)
            if (r + i < R_real) vset<T>(x, i, src[i]);
        }
#pragma unroll
        for (int i = 0; i < EPV; ++i)
          if (r + i == ones_row) vset<T>(x, i, one_val<T>());
      }
    } else {
      const int r = row0 + v / (BK / EPV);
      const int k = k0 + (v % (BK / EPV)) * EPV;
#pragma unroll
      for (int i = 0; i < EPV; ++i) {
        if (k + i < k_end) {
          if (r < R_real) vset<T>(x, i, P[(long long)r * rs + (long long)(k + i) * cs]);
          else if (r == ones_row) vset<T>(x, i, one_val<T>());
        }
      }
    }
    reg[p] = x;
  }
}

template <typename T, int MODE>
__device__ __forceinline__ void store_tile(T* __restrict__ S, const Vec16 (&reg)[Cfg<T>::NV], int tid) {
  constexpr int EPV = Cfg<T>::EPV, BK = Cfg<T>::BK, LDK = Cfg<T>::LDK;
  if constexpr (MODE == 1 && sizeof(T) == 2) {
#pragma unroll
    for (int p = 0; p < Cfg<T>::NV; ++p) {
      const int v = tid + p * NT;
      *reinterpret_cast<uint4*>(S + (v / 16) * LDT + (v % 16) * EPV) = reg[p].u;   // plain copy: [k][row]
    }
    return;
  }
#pragma unroll
  for (int p = 0; p < Cfg<T>::NV; ++p) {
    const int v = tid + p * NT;
    if constexpr (MODE == 1) {
      const int k = v % BK;
      const int r = (v / BK) * EPV;
#pragma unroll
      for (int i = 0; i < EPV; ++i) S[(r + i) * LDK + k] = vget<T>(reg[p], i);
    } else {
      const int r = v / (BK / EPV);
      const int k = (v % (BK / EPV)) * EPV;
      *reinterpret_cast<uint4*>(S + r * LDK + k) = reg[p].u;
    }
  }
}

// ---- hot-loop loads: one buffer descriptor per operand and block (base = the block's first row, so per-thread byte
// offsets stay small), per-thread voffsets computed ONCE, the k position travels in a scalar register.  Rows past the end
// of a k-contiguous operand fall outside the descriptor's range and read as 0 (hardware bounds check), so M / N tails
// need no predicate.  Zero VALU address math per load.
template <typename T, int MODE>
struct FastLoad {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff[Cfg<T>::NV];
  long long kstride_bytes;   // bytes per unit of k

  __device__ __forceinline__ void init(const T* P, long long rs, long long cs, int row0, int R_real, int K, int tid) {
    constexpr int EPV = Cfg<T>::EPV, BK = Cfg<T>::BK;
    long long valid_elems;
    const T* base;
    if constexpr (MODE == 0) {
      base = P + (long long)row0 * rs;
      valid_elems = (R_real > row0) ? ((long long)(R_real - row0 - 1) * rs + K) : 0;
      kstride_bytes = (long long)sizeof(T);
#pragma unroll
      for (int p = 0; p < Cfg<T>::NV; ++p) {
        const int v = tid + p * NT;
        voff[p] = (int)(((long long)(v / (BK / EPV)) * rs + (v % (BK / EPV)) * EPV) * (long long)sizeof(T));
      }
    } else {
      base = P + row0;
      valid_elems = (long long)(K - 1) * cs + (R_real - row0);
      kstride_bytes = cs * (long long)sizeof(T);
#pragma unroll
      for (int p = 0; p < Cfg<T>::NV; ++p) {
        const int v = tid + p * NT;
        if constexpr (sizeof(T) == 2)
          voff[p] = (int)(((long long)(v / 16) * cs + (v % 16) * EPV) * (long long)sizeof(T));
        else
          voff[p] = (int)(((long long)(v % BK) * cs + (v / BK) * EPV) * (long long)sizeof(T));
      }
    }
    // rounded up to whole dwords: the range check is per dword, and an odd bf16 element count would blank the last valid
    // element together with its (in-allocation: the leading dimension is a multiple of 8) pad neighbour
    long long bytes = (valid_elems * (long long)sizeof(T) + 3) & ~3ll;
    bytes = bytes < 0 ? 0 : (bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : bytes);
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, (int)(unsigned)bytes, 0x00020000);
  }

  __device__ __forceinline__ void load(Vec16 (&reg)[Cfg<T>::NV], int k0) const {
    const int soff = (int)((long long)k0 * kstride_bytes);
#pragma unroll
    for (int p = 0; p < Cfg<T>::NV; ++p) {
      auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[p], soff, 0);
      reg[p].u = make_uint4((unsigned)v[0], (unsigned)v[1], (unsigned)v[2], (unsigned)v[3]);
    }
  }
};

// Row-contiguous fast path: overwrite the elements of logical row `ones_row` (the all-ones row that yields the bias
// gradient) in this thread's staged vectors.
template <typename T>
__device__ __forceinline__ void patch_ones_row(Vec16 (&reg)[Cfg<T>::NV], int row0, int ones_row, int tid) {
  constexpr int EPV = Cfg<T>::EPV, BK = Cfg<T>::BK;
#pragma unroll
  for (int p = 0; p < Cfg<T>::NV; ++p) {
    const int v = tid + p * NT;
    const int r = row0 + (sizeof(T) == 2 ? (v % 16) * EPV : (v / BK) * EPV);
    if (ones_row >= r && ones_row < r + EPV) {
#pragma unroll
      for (int i = 0; i < EPV; ++i)
        if (r + i == ones_row) vset<T>(reg[p], i, one_val<T>());
    }
  }
}

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) { return dmt_pack_bf16(lo, hi); }

// packed pair of bf16: relu (as int16 a bf16 is negative exactly when its sign bit is set) ...
typedef __attribute__((ext_vector_type(2))) short s16x2_t;
__device__ __forceinline__ unsigned pk_relu_bf16(unsigned x) {
  s16x2_t v = __builtin_bit_cast(s16x2_t, x);
  const s16x2_t z = {0, 0};
  v = __builtin_elementwise_max(v, z);
  return __builtin_bit_cast(unsigned, v);
}
// ... and the 0xFFFF / 0 mask of "element > 0"
__device__ __forceinline__ unsigned pk_pos_mask_bf16(unsigned g) {
  s16x2_t v = __builtin_bit_cast(s16x2_t, g);
  const s16x2_t z = {0, 0}, one = {1, 1};
  v = __builtin_elementwise_max(__builtin_elementwise_min(v, one), z);   // 1 where > 0 else 0
  return __builtin_bit_cast(unsigned, v) * 0xFFFFu;                        // 0x0001 -> 0xFFFF per half (no carry between halves)
}

// 16-byte non-temporal load through a buffer descriptor (data streamed once must not displace the operand panels in L2)
__device__ __forceinline__ uint4 load_nt16(__amdgpu_buffer_rsrc_t r, int voff) {
  auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 2);
  return make_uint4((unsigned)v[0], (unsigned)v[1], (unsigned)v[2], (unsigned)v[3]);
}

constexpr int EPI_LD = 68;                       // fp32 words per staged epilogue row (64 + 4 pad)
constexpr int EPI_WAVE_WORDS = 32 * EPI_LD;      // one wave stages 32 rows x 64 cols at a time
constexpr int OPER_BYTES = 64 * LDT * 2;          // one operand tile: bf16 [64][160] (mode 1) >= [128][72] (mode 0) >= fp32 [128][20]
constexpr int SMEM_BYTES_OPER = 2 * OPER_BYTES;
constexpr int SMEM_BYTES_EPI = 4 * EPI_WAVE_WORDS * 4;
constexpr int SMEM_BYTES = SMEM_BYTES_OPER > SMEM_BYTES_EPI ? SMEM_BYTES_OPER : SMEM_BYTES_EPI;

struct TileCoord {
  int m0, n0, bt, ks, k_begin, k_end;
  bool valid;
};

// XCD-aware mapping of the linear work id L to an output tile.  The dispatcher places workgroup b on XCD (b % 8) and the
// persistent loop keeps L % 8 == b % 8; each XCD has a private L2.  All tiles that share an operand panel get ids with the
// same (L % 8) and consecutive (L / 8), so they run concurrently on ONE XCD and the panel is fetched into that L2 once:
//   no split-K : panel = (M tile, batch), swept over its N tiles          (A panel shared)
//   split-K    : panel = K chunk, swept over its (N tile, M tile) pairs   (both operand chunks shared)
// A wrong placement guess costs speed only, never correctness.
__device__ __forceinline__ TileCoord decode_tile(const GemmArgs& g, int L) {
  TileCoord t;
  t.valid = false;
  const int xcd = L & 7, jq = L >> 3;
  const int pi = jq / g.inner, in = jq - pi * g.inner;
  const int P = pi * 8 + xcd;
  t.m0 = t.n0 = t.bt = t.ks = t.k_begin = t.k_end = 0;
  if (P >= g.panels) return t;
  int bx, by, bz;
  if (g.split_k > 1) { bz = P; bx = in % g.gx; by = in / g.gx; }
  else { by = P % g.gy; bz = P / g.gy; bx = in; }
  t.bt = bz / g.split_k;
  t.ks = bz - t.bt * g.split_k;
  t.m0 = by * BM;
  t.n0 = bx * BN;
  t.k_begin = t.ks * g.k_per_split;
  t.k_end = (t.k_begin + g.k_per_split < g.K) ? (t.k_begin + g.k_per_split) : g.K;
  t.valid = (t.k_begin < t.k_end);
  return t;
}

// Persistent kernel: a workgroup walks the work ids b, b + gridDim.x, ...; the first operand tiles of the NEXT output tile
// are requested before the current tile's epilogue, so their latency hides behind the C stores.
template <typename T, int AMODE, int BMODE>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_kernel(const GemmArgs g) {
  constexpr int BK = Cfg<T>::BK, LDK = Cfg<T>::LDK;
  // one buffer per operand; the epilogue staging reuses the same memory once the K loop is done
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  T* As0 = reinterpret_cast<T*>(smem);
  T* Bs0 = reinterpret_cast<T*>(smem + OPER_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M_real = g.a_ones_row ? g.M - 1 : g.M;
  const int ones_row = g.a_ones_row ? g.M - 1 : -1;

  f32x16_t acc[2][2];
  Vec16 ra[Cfg<T>::NV], rb[Cfg<T>::NV];
  FastLoad<T, (AMODE == 1 ? 1 : 0)> fa;
  FastLoad<T, (BMODE == 1 ? 1 : 0)> fb;

  // fragment of a 32-row sub-tile starting at row rb, k offset kk: lane l holds row rb + (l & 31), k = kk + 8*(l>>5) .. +8
  auto frag = [&](const T* S, auto mode, int rb0, int kk) -> bf16x8_t {
    if constexpr (decltype(mode)::value == 1) {
      // each 16-lane group reads a [4 k][16 rows] block; lane i points at k row (i >> 2), rows 4*(i & 3)..+4 and receives
      // the block's column i, i.e. 4 consecutive k of row i
      const int i16 = lane & 15, grp = lane >> 4;
      const T* p = S + (kk + 8 * (grp >> 1) + (i16 >> 2)) * LDT + rb0 + 16 * (grp & 1) + 4 * (i16 & 3);
      typedef __attribute__((address_space(3))) bf16x4_t* lds_p;
      const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p));
      const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + 4 * LDT));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    } else {
      return *reinterpret_cast<const bf16x8_t*>(&S[(rb0 + (lane & 31)) * LDK + kk + 8 * (lane >> 5)]);
    }
  };
  auto compute = [&](const T* As, const T* Bs) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int kk = 0; kk < BK; kk += 16) {
        bf16x8_t af[2], bfv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = frag(As, std::integral_constant<int, AMODE>{}, wm * 64 + i * 32, kk);
#pragma unroll
        for (int j = 0; j < 2; ++j) bfv[j] = frag(Bs, std::integral_constant<int, BMODE>{}, wn * 64 + j * 32, kk);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfv[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        float af[2], bfv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = As[(wm * 64 + i * 32 + (lane & 31)) * LDK + kk + (lane >> 5)];
#pragma unroll
        for (int j = 0; j < 2; ++j) bfv[j] = Bs[(wn * 64 + j * 32 + (lane & 31)) * LDK + kk + (lane >> 5)];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bfv[j], acc[i][j], 0, 0, 0);
      }
    }
  };
  // fast path per operand: full k-step and (k-contiguous: no ones row in this tile; row-contiguous: always).  Row tails
  // never need a predicate on the fast path: k-contiguous operands read rows past the end as 0 (buffer bounds check);
  // row-contiguous operands read whatever follows in memory (or 0 past the allocation), but those tile rows only feed
  // output rows / columns >= M / N, which the epilogue never stores.  The ones row is patched in after the load.
  auto all_fast = [&](const TileCoord& t) -> bool {
    const bool ones_in = (ones_row >= t.m0 && ones_row < t.m0 + BM);
    const bool a_f = g.fast_ok && (AMODE == 0 ? !ones_in : (AMODE == 1));
    const bool b_f = g.fast_ok && (BMODE == 0 || BMODE == 1);
    return a_f && b_f && (t.k_begin + BK <= t.k_end);
  };
  auto init_fast = [&](const TileCoord& t) {
    fa.init(reinterpret_cast<const T*>(g.A) + (long long)t.bt * g.a_bs, g.a_rs, g.a_cs, t.m0, M_real, g.K, tid);
    fb.init(reinterpret_cast<const T*>(g.B) + (long long)t.bt * g.b_bs, g.b_cs, g.b_rs, t.n0, g.N, g.K, tid);
  };

  bool pre = false;   // ra / rb already hold (or have in flight) the first k tile of the output tile about to start
  const int G = (int)gridDim.x;
  for (int L = (int)blockIdx.x; L < g.total_blocks; L += G) {
    const TileCoord t = decode_tile(g, L);
    if (!t.valid) continue;
    const int m0 = t.m0, n0 = t.n0, bt = t.bt, ks = t.ks, k_begin = t.k_begin, k_end = t.k_end;
    const T* A = reinterpret_cast<const T*>(g.A) + (long long)bt * g.a_bs;
    const T* Bp = reinterpret_cast<const T*>(g.B) + (long long)bt * g.b_bs;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // B(k, n) = B[k*b_rs + n*b_cs]: the LDS "row" is n  =>  row stride = b_cs, k stride = b_rs
    const bool ones_here = (ones_row >= m0 && ones_row < m0 + BM);
    const bool a_fast = g.fast_ok && (AMODE == 0 ? !ones_here : (AMODE == 1));
    const bool b_fast = g.fast_ok && (BMODE == 0 || BMODE == 1);
    if (!pre) init_fast(t);

    // Software pipeline: the loads of tile k+1 are issued before tile k is multiplied and are only waited for (and, for
    // the ones row, patched) when they are stored to LDS after the barrier that ends the multiply.  The all-fast case has
    // its own loop with its last tile peeled: any merge of the load destinations with another path (the guarded loader, a
    // conditional load) makes the compiler copy them right after issue, which waits for the loads before the MFMA block.
    int k_done = k_begin;
    if (all_fast(t)) {
      const int k_fast_end = k_begin + ((k_end - k_begin) / BK) * BK;
      const bool patch = (AMODE == 1) && ones_here;
      if (!pre) {
        fa.load(ra, k_begin);
        fb.load(rb, k_begin);
      }
      if (patch) patch_ones_row<T>(ra, m0, ones_row, tid);
      store_tile<T, AMODE>(As0, ra, tid);
      store_tile<T, BMODE>(Bs0, rb, tid);
      __syncthreads();
      for (int k0 = k_begin; k0 + BK < k_fast_end; k0 += BK) {
        fa.load(ra, k0 + BK);
        fb.load(rb, k0 + BK);
        __builtin_amdgcn_sched_barrier(0);   // keep the loads ahead of the MFMA block (the scheduler would sink them)
        compute(As0, Bs0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (patch) patch_ones_row<T>(ra, m0, ones_row, tid);
        store_tile<T, AMODE>(As0, ra, tid);
        store_tile<T, BMODE>(Bs0, rb, tid);
        __syncthreads();
      }
      compute(As0, Bs0);
      __syncthreads();
      k_done = k_fast_end;
    }
    // guarded loop: K tail of the fast case, and every tile of the other cases
    for (int k0 = k_done; k0 < k_end; k0 += BK) {
      if (a_fast && k0 + BK <= k_end) {
        fa.load(ra, k0);
        if (AMODE == 1 && ones_here) patch_ones_row<T>(ra, m0, ones_row, tid);
      } else {
        load_tile<T, AMODE>(ra, A, g.a_rs, g.a_cs, m0, k0, M_real, k_end, ones_row, tid);
      }
      if (b_fast && k0 + BK <= k_end) fb.load(rb, k0);
      else load_tile<T, BMODE>(rb, Bp, g.b_cs, g.b_rs, n0, k0, g.N, k_end, -1, tid);
      store_tile<T, AMODE>(As0, ra, tid);
      store_tile<T, BMODE>(Bs0, rb, tid);
      __syncthreads();
      compute(As0, Bs0);
      __syncthreads();
    }

    // request the first operand tiles of this workgroup's next output tile; they land while C is written
    pre = false;
    if (L + G < g.total_blocks) {
      const TileCoord tn = decode_tile(g, L + G);
      if (tn.valid && all_fast(tn)) {
        init_fast(tn);
        fa.load(ra, tn.k_begin);
        fb.load(rb, tn.k_begin);
        pre = true;
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const float* bias = g.bias ? g.bias + (long long)bt * g.bias_bs : nullptr;
    const T* gate = g.gate ? reinterpret_cast<const T*>(g.gate) + (long long)bt * g.gate_bs : nullptr;
    const bool add_bias = (bias != nullptr) && (ks == 0);

    bool done = false;
    if constexpr (sizeof(T) == 2) {
      if (g.vec_epi) {
        // bf16 output, 16-byte aligned rows: stage 32 x 64 fp32 per wave in LDS (the operand tiles are dead), then
        // every lane finishes 8 consecutive columns: vector gate / residual loads and ONE 16-byte store per row piece.
        float* Cw = reinterpret_cast<float*>(smem) + wave * EPI_WAVE_WORDS;
        const bf16_t* resid = g.resid ? reinterpret_cast<const bf16_t*>(g.resid) + (long long)bt * g.resid_bs : nullptr;
        bf16_t* Cg = reinterpret_cast<bf16_t*>(g.C) + (long long)bt * g.c_bs;
        const int cbase = n0 + wn * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int cl = j * 32 + (lane & 31);
            const int col = cbase + cl;
            const float bv = (add_bias && col < g.N) ? bias[col] : 0.f;
            const bool relu = col < g.act_ncols;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float x = acc[i][j][r] + bv;
              if (relu) x = fmaxf(x, 0.f);
              Cw[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPI_LD + cl] = x;
            }
          }
          __builtin_amdgcn_wave_barrier();
          const int rbase = m0 + wm * 64 + i * 32;
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int rl = it * 8 + (lane >> 3);
            const int cl = (lane & 7) * 8;
            const int row = rbase + rl, col = cbase + cl;
            if (row < g.M && col < g.N) {
              float x[8];
              const float4 lo = *reinterpret_cast<const float4*>(&Cw[rl * EPI_LD + cl]);
              const float4 hi = *reinterpret_cast<const float4*>(&Cw[rl * EPI_LD + cl + 4]);
              x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w; x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
              if (col + 8 <= g.N) {
                if (gate) {
                  Vec16 gv; gv.u = *reinterpret_cast<const uint4*>(gate + (long long)row * g.ldg + col);
#pragma unroll
                  for (int e = 0; e < 8; ++e) if (!(bf2f(gv.h[e]) > 0.f)) x[e] = 0.f;
                }
                if (resid) {
                  Vec16 rv; rv.u = *reinterpret_cast<const uint4*>(resid + (long long)row * g.ldr + col);
#pragma unroll
                  for (int e = 0; e < 8; ++e) x[e] += bf2f(rv.h[e]);
                }
                uint4 o;
                o.x = cvt_pk_bf16(x[0], x[1]); o.y = cvt_pk_bf16(x[2], x[3]);
                o.z = cvt_pk_bf16(x[4], x[5]); o.w = cvt_pk_bf16(x[6], x[7]);
                *reinterpret_cast<uint4*>(Cg + (long long)row * g.ldc + col) = o;
              } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  if (col + e < g.N) {
                    float y = x[e];
                    if (gate && !(bf2f(gate[(long long)row * g.ldg + col + e]) > 0.f)) y = 0.f;
                    if (resid) y += bf2f(resid[(long long)row * g.ldr + col + e]);
                    Cg[(long long)row * g.ldc + col + e] = f2bf(y);
                  }
                }
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
        done = true;
      }
    }

    if (!done)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        if (col >= g.N) continue;
        const float bv = add_bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row >= g.M) continue;
          float x = acc[i][j][r] + bv;
          if (col < g.act_ncols) x = fmaxf(x, 0.f);
          if (gate) {
            if (!(ldf<T>(gate + (long long)row * g.ldg + col) > 0.f)) x = 0.f;
          }
          if (row == ones_row) {
            float* cl = g.c_last + (long long)bt * g.clast_bs;
            if (g.split_k > 1 || g.accumulate) atomicAdd(cl + col, x); else cl[col] = x;
            continue;
          }
          if (g.out_f32) {
            float* Cp = reinterpret_cast<float*>(g.C) + (long long)bt * g.c_bs + (long long)row * g.ldc + col;
            if (g.resid && ks == 0)
              x += reinterpret_cast<const float*>(g.resid)[(long long)bt * g.resid_bs + (long long)row * g.ldr + col];
            if (g.split_k > 1 || g.accumulate) atomicAdd(Cp, x); else *Cp = x;
          } else {
            bf16_t* Cp = reinterpret_cast<bf16_t*>(g.C) + (long long)bt * g.c_bs + (long long)row * g.ldc + col;
            if (g.resid)
              x += bf2f(reinterpret_cast<const bf16_t*>(g.resid)[(long long)bt * g.resid_bs + (long long)row * g.ldr + col]);
            *Cp = f2bf(x);
          }
        }
      }
    }
    __syncthreads();   // the staging area is the next tile's operand buffer
  }
}

// =====================================================================================================================
// Direct-to-LDS variant for the forward / input-gradient GEMMs: bf16, BOTH operands k-contiguous, K % 8 == 0, bf16 C,
// no split-K (a last k-step shorter than 64: the 16-byte chunks at k >= K are requested out of the descriptor's range and arrive as zeros).  ds_write_b128 moves only ~79 B/clk/CU, so staging a 32 KB k-step through VGPRs costs more LDS time than
// its 16 MFMAs per wave take; here `buffer_load_dwordx4 ... lds` writes the operand tiles into LDS without a VGPR pass.
//   * LDS image per operand: [128 rows][64 k] bf16, 128-byte rows, NO padding (the DMA destination is wave-uniform base
//     + lane * 16); the 16-byte slot s of row r holds logical k chunk s ^ ((r >> 1) & 7).  The swizzle is applied on the
//     per-lane GLOBAL address (still whole 128-byte lines per row) and makes the ds_read_b128 fragment reads (32 rows x
//     16 B per half-wave) conflict-free for the hardware's 16-lane groups.
//   * two LDS stages; the DMA for k-step i+1 (or for the first k-step of the workgroup's NEXT output tile) is issued
//     before k-step i is multiplied and is only waited for with a counted vmcnt(8) (8 DMA instructions per wave per
//     stage) -- raw s_barrier, never __syncthreads, which would drain the queue.
//   * MFMA operands are swapped (D^T = B A^T), so a lane ends up with 4 CONSECUTIVE columns of one row per register
//     quad: the epilogue adds bias / relu in registers and stages packed bf16 (ds_write_b64) -- or fp32 quads when a
//     residual must be added before the single rounding -- then finishes rows with 16-byte gate loads and stores.
//   * the whole bias vector (N <= 2304; wider outputs only without a bias) is put in LDS once per (persistent) workgroup.
// Tried and rejected (measured on the 204800-row shapes): a 256 x 128 tile with 8 waves and THREE DMA stages (two k-steps in
// flight, 25 % less operand traffic per FLOP) is 5-20 % slower -- the k-step is not DMA-latency bound; hand-ordered fragment
// double buffering in compute() is re-scheduled by hipcc and changes nothing; 8 waves per 128 x 128 tile (32 x 64 per wave, 4 waves
// per SIMD at 127 VGPRs) is equal or slower too (0.296 / 0.265 / 0.197 ms): the extra fragment reads per MFMA cancel the extra
// latency hiding.  Counters for this kernel (profiles/r01d_*): MFMA busy 31 %, LDS active 28 %, waves 40 % in waitcnt/barrier and
// 32 % in issue stalls.
constexpr int GL_STAGE_BYTES = 34816;   // A tile 16 KB | B tile 16 KB | 2 KB slack: = 4 waves x [32][68] fp32 of epilogue staging
constexpr int GL_MAX_N = 2304;           // widest output WITH a bias (the MMoE layer-0 block: 4 x 512 expert columns + 8 gate logits)
constexpr int GL_SMEM = 2 * GL_STAGE_BYTES + GL_MAX_N * 4;
constexpr int GL_GRID = 256 * 2;        // 2 resident workgroups per CU (77 KB of LDS each)
constexpr int GL_OOB = (int)0x80000000u; // a voffset past every descriptor's range: the lane's 16 bytes arrive as zeros

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_glds_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[GL_SMEM];   // the ONLY LDS object (a second one de-pipelines the DMA)
  typedef __attribute__((address_space(3))) void* lds_vp;
  typedef bf16_t T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  float* bias_lds = reinterpret_cast<float*>(smem + 2 * GL_STAGE_BYTES);

  for (int c = tid; c < GL_MAX_N; c += NT) bias_lds[c] = (g.bias != nullptr && c < g.N) ? g.bias[c] : 0.f;

  // loader: wave w, round p fills LDS bytes [(4p + w) * 1024, +1024) of an operand tile = rows 8(4p + w) .. +8
  const int lrow = wave * 8 + (lane >> 3);
  const int lchunk = (lane & 7) ^ ((lrow >> 1) & 7);   // (32p + lrow) >> 1 & 7 does not depend on p
  const int a_voff = (int)(((long long)lrow * g.a_rs + lchunk * 8) * 2);
  const int b_voff = (int)(((long long)lrow * g.b_cs + lchunk * 8) * 2);
  const int a_pstep = (int)(32 * g.a_rs * 2), b_pstep = (int)(32 * g.b_cs * 2);
  // the last k-step of a reduction that is not a multiple of 64: this lane's chunk lies at k >= K (in the NEXT row's bytes) -> zeros
  const int ktail = g.K & 63;
  const bool chunk_in = (ktail == 0) || (lchunk * 8 < ktail);
  const int a_voff_t = chunk_in ? a_voff : GL_OOB, b_voff_t = chunk_in ? b_voff : GL_OOB;
  // fragment reads: lane l, row (l & 31) of a 32-row block, k chunk 2*(kk/16) + (l >> 5) -> slot = chunk ^ ((row >> 1) & 7)
  const int fr_base = (lane & 31) * 128;
  const int fr_y = ((((lane >> 1) & 7) ^ (lane >> 5))) * 16;

  auto make_rsrc = [&](const T* base, long long rows_left, long long rs) {
    long long bytes = rows_left > 0 ? ((rows_left - 1) * rs + g.K) * 2 : 0;
    bytes = bytes > 0x7FFFFFFFll ? 0x7FFFFFFFll : bytes;     // (a tile addresses < 2^31 bytes from its base: fast_ok; GL_OOB stays outside)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, (int)(unsigned)bytes, 0x00020000);
  };
  auto issue = [&](int stage, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, int k0, bool last) {
    unsigned char* sb = smem + stage * GL_STAGE_BYTES + wave * 1024;
    const int av = last ? a_voff_t : a_voff, bv = last ? b_voff_t : b_voff;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vp)(sb + p * 4096), 16, av, k0 * 2 + p * a_pstep, 0, 0);
#pragma unroll
    for (int p = 0; p < 4; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vp)(sb + 16384 + p * 4096), 16, bv, k0 * 2 + p * b_pstep, 0, 0);
  };

  f32x16_t acc[2][2];
  auto compute = [&](int stage) {
    const unsigned char* As = smem + stage * GL_STAGE_BYTES + wm * 64 * 128 + fr_base;
    const unsigned char* Bs = smem + stage * GL_STAGE_BYTES + 16384 + wn * 64 * 128 + fr_base;
#pragma unroll
    for (int kk = 0; kk < 64; kk += 16) {
      const int ko = (kk * 2) ^ fr_y;
      bf16x8_t af[2], bfv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(As + i * 4096 + ko);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfv[j] = *reinterpret_cast<const bf16x8_t*>(Bs + j * 4096 + ko);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfv[j], af[i], acc[i][j], 0, 0, 0);   // swapped: D^T
    }
  };

  const T* Ag = reinterpret_cast<const T*>(g.A);
  const T* Bg = reinterpret_cast<const T*>(g.B);
  const bf16_t* gate = reinterpret_cast<const bf16_t*>(g.gate);
  const bf16_t* resid = reinterpret_cast<const bf16_t*>(g.resid);
  bf16_t* Cg = reinterpret_cast<bf16_t*>(g.C);
  const int nk = (g.K + 63) / 64;
  const int G = (int)gridDim.x;
  int cur = 0;
  bool pre = false;
  bool relaxed = false;   // the only VMEM ops younger than the prefetched stage are >= 8 C stores of the previous epilogue
  __syncthreads();   // bias_lds
  // work id L = 8 * (pi * inner + in) + xcd  ->  M panel P = 8 * pi + xcd, N tile `in` (no split-K, no batch here).
  // L advances by G = 8 * gq per tile, so (pi, in) are updated incrementally: no integer division in the tile loop.
  const int gq = G >> 3;
  const int q_step = gq / g.inner, r_step = gq - q_step * g.inner;
  const int xcd = (int)blockIdx.x & 7;
  int pi = ((int)blockIdx.x >> 3) / g.inner, in = ((int)blockIdx.x >> 3) - pi * g.inner;
  for (int L = (int)blockIdx.x; L < g.total_blocks; L += G) {
    const int P = pi * 8 + xcd;
    const int m0 = P * BM, n0 = in * BN;
    // the next tile of this workgroup
    int pi_n = pi + q_step, in_n = in + r_step;
    if (in_n >= g.inner) { in_n -= g.inner; ++pi_n; }
    const int Pn = pi_n * 8 + xcd;
    pi = pi_n; in = in_n;
    if (P >= g.panels) break;   // P grows with L: nothing valid follows
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(Ag + (long long)m0 * g.a_rs, g.M - m0, g.a_rs);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(Bg + (long long)n0 * g.b_cs, g.N - n0, g.b_cs);
    long long c_bytes = (long long)(g.M - m0) * g.ldc * 2;
    c_bytes = c_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : c_bytes;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(Cg + (long long)m0 * g.ldc, 0, (int)(unsigned)c_bytes, 0x00020000);
    // gate / residual rows of this tile (streamed once: non-temporal).  Only the interior fast path uses them.
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(gate ? gate + (long long)m0 * g.ldg : Cg), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(resid ? resid + (long long)m0 * g.ldr : Cg), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (!pre) issue(cur, ra, rb, 0, nk == 1);
    // ONE barrier per k-step: after it, stage `cur` has landed for every wave and every wave has left stage cur^1 (its
    // previous multiply, or the previous tile's epilogue staging), so the next DMA can be aimed at cur^1 right away.
    for (int kt = 0; kt + 1 < nk; ++kt) {
      // (in-order completion: "at most 8 outstanding" already implies the stage has landed; the stores drain behind the multiply)
      if (relaxed) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      relaxed = false;
      __builtin_amdgcn_s_barrier();
      issue(cur ^ 1, ra, rb, (kt + 1) * 64, kt + 2 == nk);
      compute(cur);
      cur ^= 1;
    }
    // last k-step: the DMA queue is kept busy with the first k-step of this workgroup's next output tile
    if (relaxed) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    relaxed = false;
    __builtin_amdgcn_s_barrier();
    pre = false;
    if (L + G < g.total_blocks && Pn < g.panels) {
      const int m0n = Pn * BM, n0n = in_n * BN;
      issue(cur ^ 1, make_rsrc(Ag + (long long)m0n * g.a_rs, g.M - m0n, g.a_rs), make_rsrc(Bg + (long long)n0n * g.b_cs, g.N - n0n, g.b_cs), 0, nk == 1);
      pre = true;
    }
    compute(cur);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave is done reading stage `cur`: it becomes the epilogue staging area

    // ---- epilogue (stage `cur` is free: it is the staging area).  Swapped C layout of the 32x32 MFMA:
    //      output row = lane & 31, output col = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    unsigned char* stg = smem + cur * GL_STAGE_BYTES + wave * (32 * EPI_LD * 4);
    const int ml = lane & 31, h4 = 4 * (lane >> 5);
    const int cbase = n0 + wn * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rbase = m0 + wm * 64 + i * 32;
      if (resid == nullptr) {
        bf16_t* Sw = reinterpret_cast<bf16_t*>(stg);   // [32][72] bf16
        // relu on the rounded value: as int16, a bf16 is negative exactly when its sign bit is set
        const int relu_mode = (cbase + 64 <= g.act_ncols) ? 1 : (cbase >= g.act_ncols ? 0 : 2);   // wave-uniform
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nl = j * 32 + 8 * q + h4;
            const float4 bv = *reinterpret_cast<const float4*>(&bias_lds[min(cbase + nl, GL_MAX_N - 4)]);   // (N > GL_MAX_N: no bias, all zeros)
            float x0 = acc[i][j][4 * q + 0] + bv.x, x1 = acc[i][j][4 * q + 1] + bv.y;
            float x2 = acc[i][j][4 * q + 2] + bv.z, x3 = acc[i][j][4 * q + 3] + bv.w;
            if (relu_mode == 2) {
              const int cg0 = cbase + nl;
              if (cg0 + 0 < g.act_ncols) x0 = fmaxf(x0, 0.f);
              if (cg0 + 1 < g.act_ncols) x1 = fmaxf(x1, 0.f);
              if (cg0 + 2 < g.act_ncols) x2 = fmaxf(x2, 0.f);
              if (cg0 + 3 < g.act_ncols) x3 = fmaxf(x3, 0.f);
            }
            uint2 o;
            o.x = cvt_pk_bf16(x0, x1);
            o.y = cvt_pk_bf16(x2, x3);
            if (relu_mode == 1) {
              o.x = pk_relu_bf16(o.x);
              o.y = pk_relu_bf16(o.y);
            }
            *reinterpret_cast<uint2*>(&Sw[ml * 72 + nl]) = o;
          }
        __builtin_amdgcn_wave_barrier();
        const bool interior = (rbase + 32 <= g.M) && (cbase + 64 <= g.N);   // wave-uniform
        if (interior) {
          const int cl = (lane & 7) * 8;
          const long long rrow = rbase + (lane >> 3);
          uint4 v[4], gv[4];
          if (gate) {
#pragma unroll
            for (int it = 0; it < 4; ++it) gv[it] = load_nt16(rg, (int)((((long long)(wm * 64 + i * 32 + it * 8 + (lane >> 3))) * g.ldg + cbase + cl) * 2));
          }
#pragma unroll
          for (int it = 0; it < 4; ++it) v[it] = *reinterpret_cast<const uint4*>(&Sw[(it * 8 + (lane >> 3)) * 72 + cl]);
          if (gate) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              v[it].x &= pk_pos_mask_bf16(gv[it].x); v[it].y &= pk_pos_mask_bf16(gv[it].y);
              v[it].z &= pk_pos_mask_bf16(gv[it].z); v[it].w &= pk_pos_mask_bf16(gv[it].w);
            }
          }
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
            u32x4_t o = {v[it].x, v[it].y, v[it].z, v[it].w};
            const int voff = (int)((((long long)(wm * 64 + i * 32 + it * 8 + (lane >> 3))) * g.ldc + cbase + cl) * 2);
            // C is written once and not read by this kernel: non-temporal, so it does not push the operand panels out of L2
            __builtin_amdgcn_raw_buffer_store_b128(o, rc, voff, 0, 2);
          }
        } else {
#pragma unroll 1
          for (int it = 0; it < 4; ++it) {
            const int rl = it * 8 + (lane >> 3), cl = (lane & 7) * 8;
            const int row = rbase + rl, col = cbase + cl;
            if (row < g.M && col < g.N) {
              Vec16 v;
              v.u = *reinterpret_cast<const uint4*>(&Sw[rl * 72 + cl]);
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (col + e < g.N) {
                  bf16_t y = v.h[e];
                  if (gate && !(bf2f(gate[(long long)row * g.ldg + col + e]) > 0.f)) y = 0;
                  Cg[(long long)row * g.ldc + col + e] = y;
                }
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      } else {
        float* Cw = reinterpret_cast<float*>(stg);     // [32][68] fp32: bias, relu, gate, residual, THEN one rounding
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nl = j * 32 + 8 * q + h4;
            const float4 bv = *reinterpret_cast<const float4*>(&bias_lds[min(cbase + nl, GL_MAX_N - 4)]);   // (N > GL_MAX_N: no bias, all zeros)
            float4 x;
            x.x = acc[i][j][4 * q + 0] + bv.x; x.y = acc[i][j][4 * q + 1] + bv.y;
            x.z = acc[i][j][4 * q + 2] + bv.z; x.w = acc[i][j][4 * q + 3] + bv.w;
            const int cg0 = cbase + nl;
            if (cg0 + 0 < g.act_ncols) x.x = fmaxf(x.x, 0.f);
            if (cg0 + 1 < g.act_ncols) x.y = fmaxf(x.y, 0.f);
            if (cg0 + 2 < g.act_ncols) x.z = fmaxf(x.z, 0.f);
            if (cg0 + 3 < g.act_ncols) x.w = fmaxf(x.w, 0.f);
            *reinterpret_cast<float4*>(&Cw[ml * EPI_LD + nl]) = x;
          }
        __builtin_amdgcn_wave_barrier();
        const bool interior = (rbase + 32 <= g.M) && (cbase + 64 <= g.N);   // wave-uniform
        if (interior) {
          const int cl = (lane & 7) * 8;
          uint4 gv[4], rv[4];
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const long long rloc = wm * 64 + i * 32 + it * 8 + (lane >> 3);
            rv[it] = load_nt16(rr, (int)((rloc * g.ldr + cbase + cl) * 2));
            if (gate) gv[it] = load_nt16(rg, (int)((rloc * g.ldg + cbase + cl) * 2));
          }
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int rl = it * 8 + (lane >> 3);
            float4 lo = *reinterpret_cast<const float4*>(&Cw[rl * EPI_LD + cl]);
            float4 hi = *reinterpret_cast<const float4*>(&Cw[rl * EPI_LD + cl + 4]);
            if (gate) {
              const unsigned m0_ = pk_pos_mask_bf16(gv[it].x), m1_ = pk_pos_mask_bf16(gv[it].y), m2_ = pk_pos_mask_bf16(gv[it].z), m3_ = pk_pos_mask_bf16(gv[it].w);
              lo.x = (m0_ & 0xFFFFu) ? lo.x : 0.f; lo.y = (m0_ >> 16) ? lo.y : 0.f; lo.z = (m1_ & 0xFFFFu) ? lo.z : 0.f; lo.w = (m1_ >> 16) ? lo.w : 0.f;
              hi.x = (m2_ & 0xFFFFu) ? hi.x : 0.f; hi.y = (m2_ >> 16) ? hi.y : 0.f; hi.z = (m3_ & 0xFFFFu) ? hi.z : 0.f; hi.w = (m3_ >> 16) ? hi.w : 0.f;
            }
            lo.x += __uint_as_float(rv[it].x << 16); lo.y += __uint_as_float(rv[it].x & 0xFFFF0000u);
            lo.z += __uint_as_float(rv[it].y << 16); lo.w += __uint_as_float(rv[it].y & 0xFFFF0000u);
            hi.x += __uint_as_float(rv[it].z << 16); hi.y += __uint_as_float(rv[it].z & 0xFFFF0000u);
            hi.z += __uint_as_float(rv[it].w << 16); hi.w += __uint_as_float(rv[it].w & 0xFFFF0000u);
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
            u32x4_t o = {cvt_pk_bf16(lo.x, lo.y), cvt_pk_bf16(lo.z, lo.w), cvt_pk_bf16(hi.x, hi.y), cvt_pk_bf16(hi.z, hi.w)};
            const int voff = (int)((((long long)(wm * 64 + i * 32 + rl)) * g.ldc + cbase + cl) * 2);
            __builtin_amdgcn_raw_buffer_store_b128(o, rc, voff, 0, 2);
          }
        } else {
#pragma unroll 1
          for (int it = 0; it < 4; ++it) {
            const int rl = it * 8 + (lane >> 3), cl = (lane & 7) * 8;
            const int row = rbase + rl, col = cbase + cl;
            if (row < g.M && col < g.N) {
              float x[8];
              const float4 lo = *reinterpret_cast<const float4*>(&Cw[rl * EPI_LD + cl]);
              const float4 hi = *reinterpret_cast<const float4*>(&Cw[rl * EPI_LD + cl + 4]);
              x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w; x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
              if (col + 8 <= g.N) {
                if (gate) {
                  Vec16 gv;
                  gv.u = *reinterpret_cast<const uint4*>(gate + (long long)row * g.ldg + col);
#pragma unroll
                  for (int e = 0; e < 8; ++e)
                    if (!(bf2f(gv.h[e]) > 0.f)) x[e] = 0.f;
                }
                Vec16 rv;
                rv.u = *reinterpret_cast<const uint4*>(resid + (long long)row * g.ldr + col);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] += bf2f(rv.h[e]);
                uint4 o;
                o.x = cvt_pk_bf16(x[0], x[1]); o.y = cvt_pk_bf16(x[2], x[3]);
                o.z = cvt_pk_bf16(x[4], x[5]); o.w = cvt_pk_bf16(x[6], x[7]);
                *reinterpret_cast<uint4*>(Cg + (long long)row * g.ldc + col) = o;
              } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                  if (col + e < g.N) {
                    float y = x[e];
                    if (gate && !(bf2f(gate[(long long)row * g.ldg + col + e]) > 0.f)) y = 0.f;
                    y += bf2f(resid[(long long)row * g.ldr + col + e]);
                    Cg[(long long)row * g.ldc + col + e] = f2bf(y);
                  }
              }
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    relaxed = pre && (m0 + wm * 64 + 64 <= g.M) && (cbase + 64 <= g.N);
    cur ^= 1;   // the next tile's first k-step is (arriving) in the other stage; its first barrier also ends this staging
  }
}

// =====================================================================================================================
// Direct-to-LDS variant for the weight-gradient GEMMs dW[M,N] (+)= A B with BOTH operands row-contiguous
// (A(m,k) = X[k][m], B(k,n) = dY[k][n]: the reduction runs over the strided dim), bf16 operands, fp32 C, split-K.
//   * LDS image per operand and stage: [64 k][128 rows] bf16 exactly as it lies in memory (256-byte k rows, DMA
//     destination lane-linear), except that the 64-byte chunk c of k row r is stored at chunk c ^ (r & 3): the four k rows
//     a ds_read_b64_tr_b16 group touches (same chunk) then sit in four different bank quarters.  The transpose to
//     k-contiguous MFMA fragments is done by the transpose-read, as in gemm_kernel's mode 1.
//   * the all-ones row that yields the bias gradient is a COLUMN of this image: in the one M tile that holds it, 64 threads
//     rewrite that column after the DMA has landed (one extra barrier per k-step, that tile only).
//   * same 2-stage DMA pipeline / persistent tile loop as gemm_glds_kernel; epilogue = fp32 atomics (split-K / accumulate).
//   * BKS k rows per DMA stage, NS stages (two workgroups per CU either way: NS * BKS * 512 B = 64 KB).  The kernel is bound by
//     the latency of its operand stream, i.e. by the bytes in flight per CU: <64, 2> keeps one 32 KB stage in flight per
//     workgroup, <32, 4> three 16 KB stages.
template <int BKS, int NS>
__device__ __forceinline__ void gemm_dw_glds_body(const GemmArgs& g, const int bid, const int G) {
  constexpr int OPB = BKS * 256;     // bytes of one operand image: [BKS k][128 rows] bf16
  constexpr int STAGE = 2 * OPB;     // A image | B image
  constexpr int RND = BKS / 16;      // DMA rounds per operand and stage (a round = 16 k rows = 4 per wave)
  constexpr int LPS = 2 * RND;       // DMA instructions per wave and stage
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE];
  typedef __attribute__((address_space(3))) void* lds_vp;
  typedef __attribute__((address_space(3))) bf16x4_t* lds_p4;
  typedef bf16_t T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int M_real = g.a_ones_row ? g.M - 1 : g.M;
  const int ones_row = g.a_ones_row ? g.M - 1 : -1;

  // loader: wave w, round p fills bytes [(4p + w) * 1024, +1024) of an operand image = k rows 4(4p + w) .. +4
  const int lr = wave * 4 + (lane >> 4);                       // k row (round 0)
  const int lchunk = ((lane & 15) >> 2) ^ ((lane >> 4) & 3);   // logical 64-byte chunk landing in this lane's slot
  const int a_voff = (int)(((long long)lr * g.a_cs + lchunk * 32 + (lane & 3) * 8) * 2);
  const int b_voff = (int)(((long long)lr * g.b_rs + lchunk * 32 + (lane & 3) * 8) * 2);
  const long long a_kb = g.a_cs * 2, b_kb = g.b_rs * 2;        // bytes per unit of k
  // transpose-read fragments: 16-lane group grp reads a [4 k][16 rows] block; lane i16 points at k row (i16 >> 2)
  const int i16 = lane & 15, grp = lane >> 4;
  const int fr_lane = (8 * (grp >> 1) + (i16 >> 2)) * 256 + 32 * (grp & 1) + 8 * (i16 & 3);
  auto fr_chunk = [&](int c) { return ((c ^ (i16 >> 2)) * 64) + fr_lane; };   // (no array: a dynamically indexed one is promoted to LDS)

  auto issue = [&](int stage, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, int k0) {
    unsigned char* sb = smem + stage * STAGE + wave * 1024;
#pragma unroll
    for (int p = 0; p < RND; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vp)(sb + p * 4096), 16, a_voff, (int)((k0 + p * 16) * a_kb), 0, 0);
#pragma unroll
    for (int p = 0; p < RND; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vp)(sb + OPB + p * 4096), 16, b_voff, (int)((k0 + p * 16) * b_kb), 0, 0);
  };
  f32x16_t acc[2][2];
  // The fragment reads are inline asm on purpose: hipcc makes every LDS access it can see wait vmcnt(0) for an LDS-DMA in
  // flight (it cannot tell the two stages apart), which would serialise the DMA of k-step i+1 with the multiply of k-step i.
  // Reads of 16-k slice s+1 are issued before slice s is multiplied; LDS returns in order, so "at most 8 outstanding"
  // means slice s has arrived.  The waits name the registers they release ("+v") so no MFMA is scheduled above them.
  const unsigned lds0 = (unsigned)(unsigned long long)((lds_vp)smem);
  const unsigned aA0 = lds0 + fr_chunk(wm * 2 + 0), aA1 = lds0 + fr_chunk(wm * 2 + 1);
  const unsigned aB0 = lds0 + OPB + fr_chunk(wn * 2 + 0), aB1 = lds0 + OPB + fr_chunk(wn * 2 + 1);
#define DMT_TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define DMT_TR_SLICE(R, so, kk)                                                                             \
  DMT_TR_READ(R[0], aA0 + so, (kk) * 256); DMT_TR_READ(R[1], aA0 + so, (kk) * 256 + 1024);                    \
  DMT_TR_READ(R[2], aA1 + so, (kk) * 256); DMT_TR_READ(R[3], aA1 + so, (kk) * 256 + 1024);                    \
  DMT_TR_READ(R[4], aB0 + so, (kk) * 256); DMT_TR_READ(R[5], aB0 + so, (kk) * 256 + 1024);                    \
  DMT_TR_READ(R[6], aB1 + so, (kk) * 256); DMT_TR_READ(R[7], aB1 + so, (kk) * 256 + 1024);
#define DMT_TR_WAIT(R, n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7]))
#define DMT_TR_MFMA(R)                                                                                      \
  {                                                                                                         \
    const bf16x8_t a0 = __builtin_shufflevector(R[0], R[1], 0, 1, 2, 3, 4, 5, 6, 7), a1 = __builtin_shufflevector(R[2], R[3], 0, 1, 2, 3, 4, 5, 6, 7); \
    const bf16x8_t b0 = __builtin_shufflevector(R[4], R[5], 0, 1, 2, 3, 4, 5, 6, 7), b1 = __builtin_shufflevector(R[6], R[7], 0, 1, 2, 3, 4, 5, 6, 7); \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);                           \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);                           \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);                           \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);                           \
  }
  auto compute = [&](int stage) {
    const unsigned so = (unsigned)stage * STAGE;
    bf16x4_t r0[8], r1[8];
    DMT_TR_SLICE(r0, so, 0)
    DMT_TR_SLICE(r1, so, 16)
    DMT_TR_WAIT(r0, 8);
    DMT_TR_MFMA(r0)
    if constexpr (BKS == 64) {
      DMT_TR_SLICE(r0, so, 32)
      DMT_TR_WAIT(r1, 8);
      DMT_TR_MFMA(r1)
      DMT_TR_SLICE(r1, so, 48)
      DMT_TR_WAIT(r0, 8);
      DMT_TR_MFMA(r0)
    }
    DMT_TR_WAIT(r1, 0);
    DMT_TR_MFMA(r1)
  };
#undef DMT_TR_READ
  auto rsrc_of = [&](const T* base, long long k_stride, int row0, int rows_real) {
    // (the range check works on whole dwords: with an odd number of valid elements in the last k row the dword that holds the
    //  last one would read as zero, so the bound is rounded up to the next dword -- that element is a pad column of the same
    //  row, inside the allocation because the leading dimension is a multiple of 8, and only feeds discarded / patched rows)
    long long bytes = (((long long)(g.K - 1) * k_stride + (rows_real - row0)) * 2 + 3) & ~3ll;
    bytes = bytes < 0 ? 0 : (bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base + row0), 0, (int)(unsigned)bytes, 0x00020000);
  };

  const T* Ag = reinterpret_cast<const T*>(g.A);
  const T* Bg = reinterpret_cast<const T*>(g.B);
  // issue cursor: runs NS-1 k-steps ahead of the compute cursor through the same (tile, k-step) sequence, across tile ends
  int iL = bid, ik = 0, ik_end = 0, istage = 0, inflight = 0;
  bool ivalid = false;
  __amdgpu_buffer_rsrc_t ira = rsrc_of(Ag, g.a_cs, 0, M_real), irb = rsrc_of(Bg, g.b_rs, 0, g.N);
  auto itile = [&]() {
    ivalid = false;
    for (; iL < g.total_blocks; iL += G) {
      const TileCoord ti = decode_tile(g, iL);
      if (!ti.valid) continue;
      ira = rsrc_of(Ag + (long long)ti.bt * g.a_bs, g.a_cs, ti.m0, M_real); irb = rsrc_of(Bg + (long long)ti.bt * g.b_bs, g.b_rs, ti.n0, g.N);
      ik = ti.k_begin; ik_end = ti.k_end; ivalid = true;
      return;
    }
  };
  auto issue_next = [&]() {
    if (!ivalid) return;
    issue(istage, ira, irb, ik);
    istage = (istage + 1 == NS) ? 0 : istage + 1;
    ++inflight;
    ik += BKS;
    if (ik >= ik_end) { iL += G; itile(); }
  };
  itile();
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0) issue_next();
  int cur = 0;
  for (int L = bid; L < g.total_blocks; L += G) {
    const TileCoord t = decode_tile(g, L);
    if (!t.valid) continue;
    const int m0 = t.m0, n0 = t.n0, ks = t.ks;
    const bool ones_here = (ones_row >= m0 && ones_row < m0 + BM);
    const int ones_ml = ones_row - m0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k0 = t.k_begin; k0 < t.k_end; k0 += BKS) {
      // stage `cur` must have landed: at most (inflight - 1) younger stages (LPS DMA instructions each) may still be out.  The
      // first k-step of a tile follows an epilogue whose atomics also count in vmcnt and may retire out of order with the
      // loads, so it drains everything.
      if (k0 == t.k_begin || inflight <= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (inflight == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
      __builtin_amdgcn_s_barrier();          // stage `cur` landed for every wave; every wave has left the stage before it
      if (ones_here) {                       // the bias-gradient row: column `ones_ml` of the A image := 1.0
        if (tid < BKS) {
          const unsigned q = lds0 + (unsigned)(cur * STAGE + tid * 256 + (((ones_ml >> 5) ^ (tid & 3)) * 64) + (ones_ml & 31) * 2);
          const unsigned one = 0x3F80u;
          asm volatile("ds_write_b16 %0, %1" ::"v"(q), "v"(one) : "memory");   // (asm: a visible LDS store would wait for all DMA)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      issue_next();                          // refills the stage consumed in the previous k-step
      compute(cur);
      cur = (cur + 1 == NS) ? 0 : cur + 1;
      --inflight;
    }
    // (the next k-step's barrier orders these reads of the stage against its next DMA: that DMA is only issued after it)

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const bool atomic = (g.split_k > 1) || g.accumulate;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        if (col >= g.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row >= g.M) continue;
          const float x = acc[i][j][r];
          float* dst = (row == ones_row) ? (g.c_last + (long long)t.bt * g.clast_bs + col)
                                         : (reinterpret_cast<float*>(g.C) + (long long)t.bt * g.c_bs + (long long)row * g.ldc + col);
          if (atomic) atomicAdd(dst, x); else *dst = x;
        }
      }
    (void)ks;
  }
}

template <int BKS, int NS>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_dw_glds_kernel(const GemmArgs g) {
  gemm_dw_glds_body<BKS, NS>(g, (int)blockIdx.x, (int)gridDim.x);
}

// Several reduction GEMMs of this class in ONE launch (dmt_gemm_dw_batched): the B-row weight gradients of a decoder -- six launches
// of 13-17 us on one lane, each alone on a chip it cannot fill (8-50 workgroups) -- run side by side.  The jobs travel in the kernel
// argument (no table in memory, no copy); workgroup b belongs to the job whose range [start[j], start[j + 1]) holds it and walks that
// job's work ids from its local position with the job's own stride.
constexpr int DW_MAX_JOBS = 12;
struct DwJobs {
  int n;
  int start[DW_MAX_JOBS + 1];
  GemmArgs job[DW_MAX_JOBS];
};
static_assert(sizeof(DwJobs) <= 3800, "the job pack travels in the kernel argument segment (4 KB)");

template <int BKS, int NS>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_dw_glds_jobs_kernel(const DwJobs p) {
  const int b = (int)blockIdx.x;
  int j = 0;
#pragma unroll 1
  while (j + 1 < p.n && b >= p.start[j + 1]) ++j;
  gemm_dw_glds_body<BKS, NS>(p.job[j], b - p.start[j], p.start[j + 1] - p.start[j]);
}

template <typename T>
void launch_gemm(const GemmArgs& g, dim3 grid, hipStream_t st) {
#define DMT_GEMM_CASE(AM, BMo) \
  if (g.a_mode == AM && g.b_mode == BMo) { hipLaunchKernelGGL((gemm_kernel<T, AM, BMo>), grid, dim3(NT), 0, st, g); return; }
  DMT_GEMM_CASE(0, 0) DMT_GEMM_CASE(0, 1) DMT_GEMM_CASE(0, 2)
  DMT_GEMM_CASE(1, 0) DMT_GEMM_CASE(1, 1) DMT_GEMM_CASE(1, 2)
  DMT_GEMM_CASE(2, 0) DMT_GEMM_CASE(2, 1) DMT_GEMM_CASE(2, 2)
#undef DMT_GEMM_CASE
}

// =====================================================================================================================
// B-row GEMMs with a short reduction (the decoders' per-head products around the raw-memory attention, dmt_q1mem.hip: M = batch rows,
// K <= 384, N <= 320, batch = heads; bf16, both operands k-contiguous, bf16 C; bias / relu / residual epilogue).  The tiled kernels
// above are built for long reductions (a k-step pipeline, a persistent tile loop); on these shapes -- one or two k-steps, K not a
// multiple of 64, batch strides -- they ran the generic path at ~25 us per launch, 14 launches per step.  Here a workgroup owns
// [64 rows x all N columns] of one batch member: the A tile and the WHOLE B operand go to LDS in one round of 16-byte requests, all in
// flight together (branch-free: rows past the end re-read the last row and are zeroed on the way in; the K tail is zero-filled to a
// multiple of 16), then KP / 16 MFMA steps per 32 x 32 output tile, operands swapped (D^T = B A^T) so that a lane owns ONE output row
// and four consecutive columns per register quad: the epilogue is 8-byte pieces per lane.
constexpr int SG_BM = 64;
constexpr int SG_IT = 28;                 // 16-byte pieces per thread (256 threads): (64 + NP) * KP / 8 <= 7168
__host__ __device__ inline int sg_kp(int K) { return (K + 15) & ~15; }
__host__ __device__ inline int sg_np(int N) { return (N + 31) & ~31; }
__host__ __device__ inline size_t sg_lds(int N, int K) { return (size_t)(SG_BM + sg_np(N)) * (sg_kp(K) + 8) * 2; }

__global__ __launch_bounds__(NT) void gemm_small_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sg_smem[];
  const int KP = sg_kp(g.K), NP = sg_np(g.N), RS = KP + 8, CH = g.K / 8, CHP = KP / 8;
  bf16_t* As = reinterpret_cast<bf16_t*>(sg_smem);      // [64][RS]
  bf16_t* Bs = As + SG_BM * RS;                         // [NP][RS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bt = blockIdx.y, m0 = blockIdx.x * SG_BM;
  const bf16_t* Ag = reinterpret_cast<const bf16_t*>(g.A) + (long long)bt * g.a_bs + (long long)m0 * g.a_rs;
  const bf16_t* Bg = reinterpret_cast<const bf16_t*>(g.B) + (long long)bt * g.b_bs;
  const int a_rows = (g.M - m0) < SG_BM ? (g.M - m0) : SG_BM;
  const int total = (SG_BM + NP) * CHP;
  uint4 v[SG_IT];
#pragma unroll
  for (int i = 0; i < SG_IT; ++i) {
    int c = tid + i * NT;
    c = c < total ? c : total - 1;
    const int r = c / CHP, ch = c - r * CHP;
    const int chc = ch < CH ? ch : CH - 1;
    const bf16_t* src = r < SG_BM ? Ag + (long long)(r < a_rows ? r : a_rows - 1) * g.a_rs
                                  : Bg + (long long)((r - SG_BM) < g.N ? (r - SG_BM) : g.N - 1) * g.b_cs;
    v[i] = *reinterpret_cast<const uint4*>(src + chc * 8);
  }
#pragma unroll
  for (int i = 0; i < SG_IT; ++i) {
    const int c = tid + i * NT;
    if (c < total) {
      const int r = c / CHP, ch = c - r * CHP;
      const bool ok = ch < CH && (r < SG_BM ? r < a_rows : (r - SG_BM) < g.N);
      *reinterpret_cast<uint4*>(As + (long long)r * RS + ch * 8) = ok ? v[i] : make_uint4(0u, 0u, 0u, 0u);   // (Bs follows As: row r >= 64)
    }
  }
  __syncthreads();

  const int wm = wave & 1, wn = wave >> 1;              // row block (32 rows), parity of the column blocks
  const int l31 = lane & 31, hi = lane >> 5;
  const int ncb = NP / 32;                              // column blocks: this wave takes wn, wn + 2, ...
  constexpr int MAXJ = 5;                               // N <= 320
  f32x16_t acc[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const bf16_t* arow = As + (32 * wm + l31) * RS + 8 * hi;
  const bf16_t* brow = Bs + (32 * wn + l31) * RS + 8 * hi;
  for (int s = 0; s < KP / 16; ++s) {
    const bf16x8_t ax = *reinterpret_cast<const bf16x8_t*>(arow + 16 * s);
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      if (wn + 2 * j < ncb) {                            // (scalar)
        const bf16x8_t bw = *reinterpret_cast<const bf16x8_t*>(brow + (long long)(64 * j) * RS + 16 * s);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw, ax, acc[j], 0, 0, 0);   // D^T[n, m]: lane = row m, registers = columns
      }
    }
  }
  // ---- epilogue: x = acc + bias; relu on the first act_ncols columns; + residual; one rounding (the order of gemm_kernel)
  const long long m = (long long)m0 + 32 * wm + l31;
  if (m < g.M) {
    const float* bias = g.bias ? g.bias + (long long)bt * g.bias_bs : nullptr;
    const bf16_t* rrow = g.resid ? reinterpret_cast<const bf16_t*>(g.resid) + (long long)bt * g.resid_bs + m * g.ldr : nullptr;
    bf16_t* crow = reinterpret_cast<bf16_t*>(g.C) + (long long)bt * g.c_bs + m * g.ldc;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      if (wn + 2 * j >= ncb) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = 32 * (wn + 2 * j) + 8 * q + 4 * hi;
        if (n0 >= g.N) continue;
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x[i] = acc[j][4 * q + i] + (bias ? bias[n0 + i] : 0.f);
          if (n0 + i < g.act_ncols) x[i] = fmaxf(x[i], 0.f);
        }
        if (rrow) {
          const uint2 rv = *reinterpret_cast<const uint2*>(rrow + n0);
          x[0] += __uint_as_float(rv.x << 16); x[1] += __uint_as_float(rv.x & 0xFFFF0000u);
          x[2] += __uint_as_float(rv.y << 16); x[3] += __uint_as_float(rv.y & 0xFFFF0000u);
        }
        uint2 ov;
        ov.x = dmt_pack_bf16(x[0], x[1]);
        ov.y = dmt_pack_bf16(x[2], x[3]);
        *reinterpret_cast<uint2*>(crow + n0) = ov;
      }
    }
  }
}

int pick_mode(const void* P, long long rs, long long cs, int esz, int epv) {
  const bool aligned = (((uintptr_t)P) % 16 == 0);
  if (cs == 1 && aligned && (rs % epv == 0)) return 0;
  if (rs == 1 && aligned && (cs % epv == 0)) return 1;
  (void)esz;
  return 2;
}

}  // namespace

enum GemmRoute { GR_GENERIC = 0, GR_SMALL = 1, GR_DW_GLDS = 2, GR_GLDS = 3 };

// descriptor -> kernel arguments + the kernel that takes them (shared by dmt_gemm and dmt_gemm_dw_batched)
static int gemm_prepare(const dmt_gemm_desc* d, GemmArgs& g, int& route, long long& nblk_out) {
  DMT_CHECK_ARG(d != nullptr, "dmt_gemm: null descriptor");
  DMT_CHECK_ARG(d->in_dtype == DMT_F32 || d->in_dtype == DMT_BF16, "dmt_gemm: bad in_dtype");
  DMT_CHECK_ARG(d->out_dtype == DMT_F32 || d->out_dtype == DMT_BF16, "dmt_gemm: bad out_dtype");
  DMT_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "dmt_gemm: M, N, K must be positive (%d %d %d)", d->M, d->N, d->K);
  DMT_CHECK_ARG(d->A && d->B && d->C, "dmt_gemm: null operand");
  const int split = d->split_k > 1 ? d->split_k : 1;
  const int batch = d->batch > 1 ? d->batch : 1;
  DMT_CHECK_ARG((split == 1 && !d->accumulate) || d->out_dtype == DMT_F32, "dmt_gemm: split_k / accumulate need an fp32 C");
  DMT_CHECK_ARG(split == 1 || (d->act_ncols == 0 && d->gate == nullptr), "dmt_gemm: split_k cannot fuse relu / gate");
  DMT_CHECK_ARG(!d->a_ones_row || d->c_last != nullptr, "dmt_gemm: a_ones_row needs c_last");
  DMT_CHECK_ARG(!d->a_ones_row || d->M >= 2, "dmt_gemm: a_ones_row needs M >= 2");
  DMT_CHECK_ARG(d->in_dtype == d->out_dtype || d->out_dtype == DMT_F32, "dmt_gemm: bf16 output needs bf16 operands");
  g.M = d->M; g.N = d->N; g.K = d->K;
  g.A = d->A; g.a_rs = d->a_rs; g.a_cs = d->a_cs;
  g.B = d->B; g.b_rs = d->b_rs; g.b_cs = d->b_cs;
  g.C = d->C; g.ldc = d->ldc;
  g.bias = d->bias; g.act_ncols = d->act_ncols;
  g.gate = d->gate; g.ldg = d->ldg; g.resid = d->resid; g.ldr = d->ldr;
  g.a_ones_row = d->a_ones_row ? 1 : 0; g.c_last = d->c_last;
  g.split_k = split; g.batch = batch; g.accumulate = d->accumulate ? 1 : 0;
  g.a_bs = d->a_bs; g.b_bs = d->b_bs; g.c_bs = d->c_bs; g.bias_bs = d->bias_bs; g.gate_bs = d->gate_bs;
  g.resid_bs = d->resid_bs; g.clast_bs = d->clast_bs;
  g.out_f32 = (d->out_dtype == DMT_F32);
  const int esz = d->in_dtype == DMT_F32 ? 4 : 2;
  const int epv = 16 / esz;
  const int bk = d->in_dtype == DMT_F32 ? 16 : 64;
  g.a_mode = pick_mode(d->A, d->a_rs, d->a_cs, esz, epv);
  if (g.a_mode != 2 && batch > 1 && (d->a_bs % epv != 0)) g.a_mode = 2;
  // B rows in LDS are n: row stride b_cs, k stride b_rs
  g.b_mode = pick_mode(d->B, d->b_cs, d->b_rs, esz, epv);
  if (g.b_mode != 2 && batch > 1 && (d->b_bs % epv != 0)) g.b_mode = 2;
  long long kps = (d->K + split - 1) / split;
  kps = ((kps + bk - 1) / bk) * bk;
  g.k_per_split = (int)kps;
  g.gx = (d->N + BN - 1) / BN; g.gy = (d->M + BM - 1) / BM; g.gz = batch * split;
  if (split > 1) { g.inner = g.gx * g.gy; g.panels = g.gz; } else { g.inner = g.gx; g.panels = g.gy * g.gz; }
  const long long nblk = 8ll * g.inner * ((g.panels + 7) / 8);
  DMT_CHECK_ARG(nblk < 0x7FFFFFFFll, "dmt_gemm: grid too large");
  g.total_blocks = (int)nblk;
  // vectorised epilogue: bf16 in/out, no split / ones row, every row of C / gate / resid 16-byte aligned
  auto al16 = [](const void* q, long long ld, long long bs) { return q == nullptr || (((uintptr_t)q) % 16 == 0 && ld % 8 == 0 && bs % 8 == 0); };
  g.vec_epi = (d->in_dtype == DMT_BF16 && d->out_dtype == DMT_BF16 && split == 1 && !d->accumulate && !d->a_ones_row && al16(d->C, d->ldc, d->c_bs) &&
               al16(d->gate, d->ldg, d->gate_bs) && al16(d->resid, d->ldr, d->resid_bs)) ? 1 : 0;
  {
    // 32-bit buffer offsets: per-thread voffset spans <= 128 rows (mode 0) or BK k-rows (mode 1); the scalar k offset
    // spans the whole reduction extent of one operand
    const long long a_k_bytes = (long long)d->K * (g.a_mode == 1 ? d->a_cs : 1) * esz, b_k_bytes = (long long)d->K * (g.b_mode == 1 ? d->b_rs : 1) * esz;
    const long long a_v_bytes = (g.a_mode == 1 ? 64ll * d->a_cs : 128ll * d->a_rs) * esz, b_v_bytes = (g.b_mode == 1 ? 64ll * d->b_rs : 128ll * d->b_cs) * esz;
    g.fast_ok = (a_k_bytes + a_v_bytes < 0x7FFFFFFFll && b_k_bytes + b_v_bytes < 0x7FFFFFFFll) ? 1 : 0;
  }
  // the B-row class (short reduction, any batch): one round of loads, whole B operand in LDS
  const bool small_shape = d->in_dtype == DMT_BF16 && g.a_mode == 0 && g.b_mode == 0 && g.vec_epi && d->gate == nullptr &&
                           d->K % 8 == 0 && d->K <= 384 && d->N % 4 == 0 && d->N <= 320 && d->M <= 65536 &&
                           (long long)(SG_BM + sg_np(d->N)) * (sg_kp(d->K) / 8) <= (long long)SG_IT * NT && sg_lds(d->N, d->K) <= 120 * 1024 &&
                           (d->resid == nullptr || d->ldr % 4 == 0);
  // direct-to-LDS: whole 64-wide k-steps as in round 1; a shorter last k-step (K % 8 == 0) where the B-row kernel does not apply
  const bool glds = d->in_dtype == DMT_BF16 && g.a_mode == 0 && g.b_mode == 0 && g.vec_epi && g.fast_ok && batch == 1 &&
                    (d->K % 64 == 0 || (d->K % 8 == 0 && !small_shape)) && (d->N <= GL_MAX_N || d->bias == nullptr);
  const bool dw_glds = d->in_dtype == DMT_BF16 && g.out_f32 && g.a_mode == 1 && g.b_mode == 1 && g.fast_ok &&
                       (d->K % 64 == 0) && d->bias == nullptr && d->gate == nullptr && d->resid == nullptr && d->act_ncols == 0;
  const bool small = small_shape && !glds && !dw_glds;
  route = small ? GR_SMALL : (dw_glds ? GR_DW_GLDS : (glds ? GR_GLDS : GR_GENERIC));
  nblk_out = nblk;
  return DMT_OK;
}

extern "C" int dmt_gemm(const dmt_gemm_desc* d, void* stream) {
  GemmArgs g;
  int route = GR_GENERIC;
  long long nblk = 0;
  const int prc = gemm_prepare(d, g, route, nblk);
  if (prc != DMT_OK) return prc;
  const int split = g.split_k, batch = g.batch;
  const bool small = route == GR_SMALL, dw_glds = route == GR_DW_GLDS, glds = route == GR_GLDS;
  dim3 grid((unsigned)(nblk < PERSIST_GRID ? nblk : PERSIST_GRID));
  hipStream_t st = (hipStream_t)stream;
  if (small) {
    // the attribute belongs to the (function, device) pair: set it on every device this process launches from (a host-side table
    // lookup per call after the first; no process-wide flag that a second GPU or a second host thread could find already set)
    const size_t dyn = (size_t)sg_lds(d->N, d->K);
    if (dyn > 64 * 1024) {
      int dev = 0;
      (void)hipGetDevice(&dev);
      static std::atomic<unsigned long long> attr_done{0ull};        // bit per device ordinal < 64
      const unsigned long long bit = dev < 64 ? (1ull << dev) : 0ull;
      if (bit == 0ull || (attr_done.load(std::memory_order_acquire) & bit) == 0ull) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
        if (e != hipSuccess) {
          dmt_set_error("dmt_gemm(small): hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed on device %d: %s", dev, hipGetErrorString(e));
          return DMT_ERR_LAUNCH;
        }
        attr_done.fetch_or(bit, std::memory_order_release);
      }
    }
    hipLaunchKernelGGL(gemm_small_kernel, dim3((unsigned)((d->M + SG_BM - 1) / SG_BM), (unsigned)batch), dim3(NT), sg_lds(d->N, d->K), st, g);
    DMT_CHECK_LAUNCH("dmt_gemm(small)");
    return DMT_OK;
  }
  if (dw_glds) {
    const dim3 gd((unsigned)(nblk < GL_GRID ? nblk : GL_GRID));
    // long reductions (the 204800-row weight gradients) run three 16 KB stages ahead, short ones one 32 KB stage:
    // measured +3..7 % resp. -8 % for the other choice
    if ((long long)d->K / split > 4096) hipLaunchKernelGGL((gemm_dw_glds_kernel<32, 4>), gd, dim3(NT), 0, st, g);
    else hipLaunchKernelGGL((gemm_dw_glds_kernel<64, 2>), gd, dim3(NT), 0, st, g);
  } else if (glds) {
    hipLaunchKernelGGL(gemm_glds_kernel, dim3((unsigned)(nblk < GL_GRID ? nblk : GL_GRID)), dim3(NT), 0, st, g);
    DMT_CHECK_LAUNCH("dmt_gemm(glds)");
    return DMT_OK;
  } else if (d->in_dtype == DMT_F32) {
    launch_gemm<float>(g, grid, st);
  } else {
    launch_gemm<bf16_t>(g, grid, st);
  }
  DMT_CHECK_LAUNCH("dmt_gemm");
  return DMT_OK;
}

// n reduction GEMMs of the weight-gradient class (bf16 operands with the reduction dimension as their ROW index, fp32 C, K % 64 == 0,
// no bias / relu / gate / residual: what dmt_gemm routes to gemm_dw_glds_kernel<64, 2>) in one launch.  All or nothing: a descriptor of
// another class makes the call fail with DMT_ERR_UNSUPPORTED before anything is launched (the caller then launches them one by one).
// Two jobs must not write the same elements of C unless both accumulate (then they meet in fp32 atomics, like the splits of one job).
extern "C" int dmt_gemm_dw_batched(const dmt_gemm_desc* descs, int32_t n, void* stream) {
  DMT_CHECK_ARG(descs != nullptr && n > 0, "dmt_gemm_dw_batched: no jobs");
  hipStream_t st = (hipStream_t)stream;
  // pass 1: every descriptor is checked before the first launch (all or nothing also when n spans several launches)
  for (int j = 0; j < n; ++j) {
    GemmArgs probe;
    int route = GR_GENERIC;
    long long nblk = 0;
    const int prc = gemm_prepare(&descs[j], probe, route, nblk);
    if (prc != DMT_OK) return prc;
    const long long kps = (long long)descs[j].K / (probe.split_k > 1 ? probe.split_k : 1);
    if (route != GR_DW_GLDS || kps > 4096) {
      dmt_set_error("dmt_gemm_dw_batched: job %d (M %d, N %d, K %d) is not of the B-row weight-gradient class", j, descs[j].M, descs[j].N, descs[j].K);
      return DMT_ERR_UNSUPPORTED;
    }
  }
  for (int j0 = 0; j0 < n; j0 += DW_MAX_JOBS) {
    DwJobs p;
    p.n = (n - j0) < DW_MAX_JOBS ? (n - j0) : DW_MAX_JOBS;
    int at = 0;
    for (int j = 0; j < p.n; ++j) {
      int route = GR_GENERIC;
      long long nblk = 0;
      const int prc = gemm_prepare(&descs[j0 + j], p.job[j], route, nblk);
      if (prc != DMT_OK) return prc;
      const int nb = (int)(nblk < GL_GRID ? nblk : GL_GRID);
      p.start[j] = at;
      at += nb;
    }
    p.start[p.n] = at;
    for (int j = p.n + 1; j <= DW_MAX_JOBS; ++j) p.start[j] = at;
    hipLaunchKernelGGL((gemm_dw_glds_jobs_kernel<64, 2>), dim3((unsigned)at), dim3(NT), 0, st, p);
    DMT_CHECK_LAUNCH("dmt_gemm_dw_batched");
  }
  return DMT_OK;
}
