// Weight gradients of the d_model-wide layers as ONE wide-block reduction GEMM (bf16 operands, fp32 atomics out):
//
//     C[i, j] += sum_m A[m, i] * B[m, j]          A: [M, 320]   B: [M, N]   C: [320, N] (or stored transposed: C^T[j, i])
//
// i.e. dW = X^T dY for the layers whose input OR output is d_model = 320 wide (FFN dense / dense_1, the packed QKV projection):
// the weight gradients tf.gradients emits for tf.layers.dense at TransformerModel_util.py:188-190, 224-228.  The reduction runs
// over the M = batch x sequence rows (204 800), so the kernel is a stream over both operands; what the 128 x 128-tile kernel
// (gemm_dw_glds_kernel) loses is operand bytes per FLOP: 512 B of DMA per MFMA.  Here a workgroup owns a [320 x 256] block of C
//   * 8 wavefronts (two per SIMD, 256 registers), wavefront (wi, wj) holds [160 x 64] = 5 x 2 accumulator tiles (160 registers);
//   * one stage = 32 rows of M: the A rows (32 x 640 B) and the block's B columns (32 x 512 B) arrive by DMA exactly as they lie
//     in memory (row stride padded to 704 / 576 B in LDS: the four k rows of a transpose read fall into four bank quarters);
//   * both operands are row-contiguous and the reduction runs over rows, so the MFMA fragments are transposed on the way out of LDS
//     by ds_read_b64_tr_b16 (7 fragments feed 10 MFMAs);
//   * 230 B of DMA per MFMA, three 40 KB stages in the ring (80 KB in flight per CU);
//   * split over M across workgroups (all column blocks of one split on the same XCD, so the A rows are fetched once per XCD),
//     fp32 atomics into C; the bias gradient (column sums of B, or of A for the transposed form) is accumulated beside the MFMAs
//     with v_dot2 on the fragments already in registers.
#include "dmt_common.h"
#include <utility>

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(2))) short s16x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) void* lds_vp;

constexpr int DW_NT = 512;
constexpr int DW_AW = 320;                 // A width (i)
constexpr int DW_BW = 256;                 // columns of B per workgroup (j)
constexpr int DW_KS = 32;                  // rows of M per stage
constexpr int DW_SA = DW_AW * 2 + 64;      // LDS row stride of the A image (704 B)
constexpr int DW_SB = DW_BW * 2 + 64;      // ... of the B image (576 B)
constexpr int DW_A_BYTES = DW_KS * DW_SA;  // 22528 = 22 pieces
constexpr int DW_B_BYTES = DW_KS * DW_SB;  // 18432 = 18 pieces
constexpr int DW_STAGE = DW_A_BYTES + DW_B_BYTES;   // 40960
constexpr int DW_PIECES = DW_STAGE / 1024;          // 40
constexpr int DW_PPW = DW_PIECES / 8;               // 5 per wavefront
constexpr int DW_NS = 3;
static_assert(DW_A_BYTES % 1024 == 0 && DW_B_BYTES % 1024 == 0 && DW_PIECES % 8 == 0, "stage layout");

struct DwArgs {
  const bf16_t* A; long long ld_a;
  const bf16_t* B; long long ld_b;
  long long M;
  int N;
  float* C; long long ldc;
  int transposed;        // 0: C[i * ldc + j]   1: C[j * ldc + i]
  float* bias;           // or null
  int bias_of;           // 1: column sums of B -> bias[j]   2: column sums of A -> bias[i]
  int jblocks, splits;
  long long rows_per_split;   // multiple of DW_KS
  float* ws;                  // ordered form: partial blocks ws[split][320][ws_ld] (+ bias partials behind them), else null
  int ws_ld;                  // = jblocks * DW_BW
  float* ws_bias;             // [splits][ws_nb]
  int ws_nb;
};

template <int OFF> __device__ __forceinline__ void dw_tr(bf16x4_t& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
// 14 transpose reads (5 A tiles x 2 halves, 2 B tiles x 2 halves) of one 16-row k-step
struct KFrag { bf16x4_t a[5][2]; bf16x4_t b[2][2]; };
template <int N> __device__ __forceinline__ void dw_wait(KFrag& f) {
  asm volatile("s_waitcnt lgkmcnt(%14)"
               : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.a[2][0]), "+v"(f.a[2][1]), "+v"(f.a[3][0]),
                 "+v"(f.a[3][1]), "+v"(f.a[4][0]), "+v"(f.a[4][1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1])
               : "i"(N));
}

template <int... I, typename F>
__device__ __forceinline__ void dfor_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void dfor(F&& f) { dfor_impl(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ float dot_ones(bf16x4_t v, float acc) {
  // acc += v[0] + v[1] + v[2] + v[3]   (bf16 -> fp32 exactly: the value is the high half of the fp32 word)
  const unsigned lo = (unsigned)(unsigned short)v[0] | ((unsigned)(unsigned short)v[1] << 16);
  const unsigned hi = (unsigned)(unsigned short)v[2] | ((unsigned)(unsigned short)v[3] << 16);
  acc += __uint_as_float(lo << 16) + __uint_as_float(lo & 0xFFFF0000u);
  acc += __uint_as_float(hi << 16) + __uint_as_float(hi & 0xFFFF0000u);
  return acc;
}

template <int BIAS_OF, bool TRANSPOSED>
__global__ __launch_bounds__(DW_NT, 2) void wgrad320_kernel(const DwArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DW_NS * DW_STAGE];   // the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave & 1, wj = wave >> 1;
  // block -> (split, column block): the column blocks of one split share an XCD (block b runs on XCD b % 8), so the split's A rows
  // come out of that XCD's L2 for all of them
  const int b = (int)blockIdx.x;
  const int xcd = b & 7, q = b >> 3;
  const int jb = q % g.jblocks;
  const int split = (q / g.jblocks) * 8 + xcd;
  if (split >= g.splits) return;
  const long long row_begin = (long long)split * g.rows_per_split;
  long long rows = g.M - row_begin;
  if (rows > g.rows_per_split) rows = g.rows_per_split;
  if (rows <= 0) return;
  const int nstages = (int)((rows + DW_KS - 1) / DW_KS);

  // ---- DMA: piece qq = wave * 5 + p of a stage; lane -> (row, byte column) of the padded LDS image
  const bf16_t* Ab = g.A + row_begin * g.ld_a;
  const bf16_t* Bb = g.B + row_begin * g.ld_b + (long long)jb * DW_BW;
  long long a_bytes = rows * g.ld_a * 2, b_bytes = (rows - 1) * g.ld_b * 2 + ((long long)g.N - (long long)jb * DW_BW) * 2;
  if (a_bytes > 0x7FFFFFFFll) a_bytes = 0x7FFFFFFFll;
  if (b_bytes > 0x7FFFFFFFll) b_bytes = 0x7FFFFFFFll;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ab), 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bb), 0, (int)b_bytes, 0x00020000);
  int voff[DW_PPW];
#pragma unroll
  for (int p = 0; p < DW_PPW; ++p) {
    const int qq = wave * DW_PPW + p;
    if (qq < DW_A_BYTES / 1024) {
      const int L = qq * 1024 + lane * 16;
      const int row = L / DW_SA, col = L % DW_SA;
      voff[p] = (col < DW_AW * 2) ? (int)(row * g.ld_a * 2 + col) : 0;
    } else {
      const int L = (qq - DW_A_BYTES / 1024) * 1024 + lane * 16;
      const int row = L / DW_SB, col = L % DW_SB;
      voff[p] = (col < DW_BW * 2) ? (int)(row * g.ld_b * 2 + col) : 0;
    }
  }
  const int a_step = (int)(DW_KS * g.ld_a * 2), b_step = (int)(DW_KS * g.ld_b * 2);   // bytes per stage
  auto issue = [&](int buf, int s) {
    unsigned char* sb = smem + buf * DW_STAGE + wave * DW_PPW * 1024;
#pragma unroll
    for (int p = 0; p < DW_PPW; ++p) {
      const int qq = wave * DW_PPW + p;   // wave-uniform
      if (qq < DW_A_BYTES / 1024) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vp)(sb + p * 1024), 16, voff[p], s * a_step, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vp)(sb + p * 1024), 16, voff[p], s * b_step, 0, 0);
    }
  };

  // ---- transpose-read fragments: 16-lane group grp reads a [4 k][16 col] block; lane i16 points at k row (i16 >> 2), 8-byte chunk i16 & 3
  const int i16 = lane & 15, grp = lane >> 4, hi = lane >> 5;
  const unsigned lds0 = (unsigned)(unsigned long long)((lds_vp)smem);
  const unsigned a_lane = lds0 + (8 * hi + (i16 >> 2)) * DW_SA + (160 * wi + 16 * (grp & 1) + 4 * (i16 & 3)) * 2;
  const unsigned b_lane = lds0 + DW_A_BYTES + (8 * hi + (i16 >> 2)) * DW_SB + (64 * wj + 16 * (grp & 1) + 4 * (i16 & 3)) * 2;

  f32x16_t acc[5][2];
#pragma unroll
  for (int ti = 0; ti < 5; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;
  float bsum[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) bsum[t] = 0.f;

  auto rd = [&](KFrag& f, unsigned so, auto ksc) {
    constexpr int ks = decltype(ksc)::value;
    dfor<5>([&](auto tc) {
      constexpr int ti = decltype(tc)::value;
      dw_tr<ks * 16 * DW_SA + ti * 64>(f.a[ti][0], a_lane + so);
      dw_tr<ks * 16 * DW_SA + 4 * DW_SA + ti * 64>(f.a[ti][1], a_lane + so);
    });
    dfor<2>([&](auto tc) {
      constexpr int tj = decltype(tc)::value;
      dw_tr<ks * 16 * DW_SB + tj * 64>(f.b[tj][0], b_lane + so);
      dw_tr<ks * 16 * DW_SB + 4 * DW_SB + tj * 64>(f.b[tj][1], b_lane + so);
    });
  };
  auto mm = [&](KFrag& f) {
    bf16x8_t af[5], bfr[2];
#pragma unroll
    for (int ti = 0; ti < 5; ++ti) af[ti] = __builtin_shufflevector(f.a[ti][0], f.a[ti][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) bfr[tj] = __builtin_shufflevector(f.b[tj][0], f.b[tj][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int ti = 0; ti < 5; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ti], bfr[tj], acc[ti][tj], 0, 0, 0);
    if constexpr (BIAS_OF == 1) {
      if (wi == 0) {   // wave-uniform: the two i halves hold the same B fragments
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) bsum[tj] = dot_ones(f.b[tj][1], dot_ones(f.b[tj][0], bsum[tj]));
      }
    } else if constexpr (BIAS_OF == 2) {
      if (wj == 0) {
#pragma unroll
        for (int ti = 0; ti < 5; ++ti) bsum[ti] = dot_ones(f.a[ti][1], dot_ones(f.a[ti][0], bsum[ti]));
      }
    }
  };

  issue(0, 0);
  issue(1, 1);
  int buf = 0;
#pragma unroll 1
  for (int s = 0; s < nstages; ++s) {
    // stage s has landed for this wavefront (only the 5 pieces of stage s + 1 may still be open) ...
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(DW_PPW) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // ... and for all of them; everybody has left stage s - 1, whose buffer stage s + 2 refills
    // The fragment reads go out FIRST, the ring's refill behind them (round 5): an LDS-DMA instruction holds its wavefront 100-300 cycles
    // at issue -- with the five of them in front, the reads' round trip started only after ~1000 cycles in which this wavefront offered
    // the matrix pipe nothing; now the reads are in flight under the DMA issue.
    // (past the last stage the pieces fall outside the descriptors' range and read as zero into a buffer nobody multiplies)
    const unsigned so = (unsigned)buf * DW_STAGE;
    KFrag f0, f1;
    rd(f0, so, std::integral_constant<int, 0>{});
    rd(f1, so, std::integral_constant<int, 1>{});
    issue(buf == 0 ? 2 : buf - 1, s + 2);
    dw_wait<14>(f0);
    mm(f0);
    dw_wait<0>(f1);
    mm(f1);
    buf = (buf + 1 == DW_NS) ? 0 : buf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // the ring is quiet: LDS becomes the transpose scratch of the epilogue

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col (j) = lane & 31, row (i) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int jbase = jb * DW_BW + 64 * wj;
  if (g.ws != nullptr) {
    // ordered form: this split's partial block, untransposed, into its own slab of the workspace (no atomics)
    float* W = g.ws + (long long)split * DW_AW * g.ws_ld;
#pragma unroll
    for (int ti = 0; ti < 5; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        const int j = jbase + 32 * tj + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = 160 * wi + 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * hi;
          W[(long long)i * g.ws_ld + j] = acc[ti][tj][r];
        }
      }
    if constexpr (BIAS_OF == 1) {
      if (wi == 0) {
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          const float v = bsum[tj] + __shfl_xor(bsum[tj], 32, 64);
          if (hi == 0) g.ws_bias[(long long)split * g.ws_nb + jbase + 32 * tj + (lane & 31)] = v;
        }
      }
    } else if constexpr (BIAS_OF == 2) {
      if (wj == 0 && jb == 0) {
#pragma unroll
        for (int ti = 0; ti < 5; ++ti) {
          const float v = bsum[ti] + __shfl_xor(bsum[ti], 32, 64);
          if (hi == 0) g.ws_bias[(long long)split * g.ws_nb + 160 * wi + 32 * ti + (lane & 31)] = v;
        }
      }
    }
    return;
  }
  if constexpr (!TRANSPOSED) {
#pragma unroll
    for (int ti = 0; ti < 5; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        const int j = jbase + 32 * tj + (lane & 31);
        if (j < g.N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = 160 * wi + 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * hi;
            atomicAdd(g.C + (long long)i * g.ldc + j, acc[ti][tj][r]);
          }
        }
      }
  } else {
    float* S = reinterpret_cast<float*>(smem) + wave * (32 * 33);   // per-wavefront [32 i][33] scratch
#pragma unroll
    for (int ti = 0; ti < 5; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
#pragma unroll
        for (int r = 0; r < 16; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + (lane & 31)] = acc[ti][tj][r];
        __builtin_amdgcn_wave_barrier();
        const int i = 160 * wi + 32 * ti + (lane & 31);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
          const int jl = 2 * jj + hi;
          const int j = jbase + 32 * tj + jl;
          const float v = S[(lane & 31) * 33 + jl];
          if (j < g.N) atomicAdd(g.C + (long long)j * g.ldc + i, v);
        }
        __builtin_amdgcn_wave_barrier();
      }
  }
  if constexpr (BIAS_OF == 1) {
    if (wi == 0) {
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        const float v = bsum[tj] + __shfl_xor(bsum[tj], 32, 64);
        const int j = jbase + 32 * tj + (lane & 31);
        if (hi == 0 && j < g.N) atomicAdd(g.bias + j, v);
      }
    }
  } else if constexpr (BIAS_OF == 2) {
    if (wj == 0 && jb == 0) {
#pragma unroll
      for (int ti = 0; ti < 5; ++ti) {
        const float v = bsum[ti] + __shfl_xor(bsum[ti], 32, 64);
        if (hi == 0) atomicAdd(g.bias + 160 * wi + 32 * ti + (lane & 31), v);
      }
    }
  }
}

// ordered form, second launch: C (+)= sum over the splits, in split order, of their partial blocks; likewise the bias partials
__global__ __launch_bounds__(256) void wgrad320_reduce_kernel(const DwArgs g) {
  const long long n_c = (long long)DW_AW * g.N;
  const long long n_all = n_c + (g.bias_of ? g.ws_nb : 0);
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n_all; e += (long long)gridDim.x * 256) {
    if (e < n_c) {
      const int i = (int)(e / g.N), j = (int)(e % g.N);
      float s = 0.f;
      for (int sp = 0; sp < g.splits; ++sp) s += g.ws[((long long)sp * DW_AW + i) * g.ws_ld + j];
      float* dst = g.transposed ? g.C + (long long)j * g.ldc + i : g.C + (long long)i * g.ldc + j;
      *dst += s;
    } else {
      const int k = (int)(e - n_c);
      const int lim = g.bias_of == 1 ? g.N : DW_AW;
      if (k < lim) {
        float s = 0.f;
        for (int sp = 0; sp < g.splits; ++sp) s += g.ws_bias[(long long)sp * g.ws_nb + k];
        g.bias[k] += s;
      }
    }
  }
}

static void dw_plan(long long M, int N, long long ld_max, int& jblocks, int& splits, long long& rps) {
  jblocks = (N + DW_BW - 1) / DW_BW;
  // one workgroup per CU: splits = the multiple of 8 that fills (at most) the 256 CUs, bounded by the work
  splits = (256 / jblocks) / 8 * 8;
  if (splits < 8) splits = 8;
  const long long max_splits = (M + DW_KS - 1) / DW_KS;
  if (splits > max_splits) splits = (int)max_splits;
  // (the descriptor offsets of one split are 32-bit)
  rps = (M + splits - 1) / splits;
  rps = (rps + DW_KS - 1) / DW_KS * DW_KS;
  while (rps * ld_max * 2 > 0x7FFFFFFFll) { splits *= 2; rps = ((M + splits - 1) / splits + DW_KS - 1) / DW_KS * DW_KS; }
  splits = (int)((M + rps - 1) / rps);      // (rounding rows_per_split up may leave the last splits empty: every split below has rows)
}

}  // namespace

extern "C" uint64_t dmt_wgrad320_det_ws_bytes(int64_t M, int32_t N) {
  if (M <= 0 || N <= 0) return 0;
  int jblocks, splits; long long rps;
  dw_plan(M, N, 4096, jblocks, splits, rps);            // (an upper bound on the splits for any row stride up to 4096 elements)
  const int nb = jblocks * DW_BW > DW_AW ? jblocks * DW_BW : DW_AW;
  const long long per = (long long)DW_AW * jblocks * DW_BW + nb;
  return (uint64_t)(((splits + 7) / 8 * 8) * per * 4 + 256);
}

extern "C" int dmt_wgrad320(const dmt_wgrad_desc* d, void* stream) {
  DMT_CHECK_ARG(d != nullptr, "dmt_wgrad320: null descriptor");
  DMT_CHECK_ARG(d->A && d->B && d->C && d->M > 0 && d->N > 0, "dmt_wgrad320: bad argument");
  DMT_CHECK_ARG(d->a_cols == DW_AW, "dmt_wgrad320: A must be %d columns wide (got %d)", DW_AW, d->a_cols);
  DMT_CHECK_ARG(d->ld_a % 8 == 0 && d->ld_b % 8 == 0 && (((uintptr_t)d->A | (uintptr_t)d->B) & 15) == 0, "dmt_wgrad320: operand rows must be 16-byte aligned");
  DMT_CHECK_ARG(d->N % 8 == 0, "dmt_wgrad320: N must be a multiple of 8");
  DMT_CHECK_ARG(d->bias_of == 0 || d->bias != nullptr, "dmt_wgrad320: bias_of needs a bias pointer");
  DMT_CHECK_ARG(d->bias_of >= 0 && d->bias_of <= 2, "dmt_wgrad320: bad bias_of");
  DMT_CHECK_ARG((long long)DW_KS * d->ld_b * 2 * 4 < 0x7FFFFFFFll, "dmt_wgrad320: row stride too large");
  DwArgs g;
  g.A = (const bf16_t*)d->A; g.ld_a = d->ld_a;
  g.B = (const bf16_t*)d->B; g.ld_b = d->ld_b;
  g.M = d->M; g.N = d->N;
  g.C = d->C; g.ldc = d->ldc;
  g.transposed = d->transposed ? 1 : 0;
  g.bias = d->bias; g.bias_of = d->bias_of;
  int splits; long long rps;
  dw_plan(d->M, d->N, d->ld_b > d->ld_a ? d->ld_b : d->ld_a, g.jblocks, splits, rps);
  g.splits = splits;
  g.rows_per_split = rps;
  g.ws = nullptr; g.ws_bias = nullptr; g.ws_ld = g.jblocks * DW_BW; g.ws_nb = g.ws_ld > DW_AW ? g.ws_ld : DW_AW;
  if (d->det_ws != nullptr) {
    const long long need = ((long long)splits * ((long long)DW_AW * g.ws_ld + g.ws_nb)) * 4;
    DMT_CHECK_ARG((((uintptr_t)d->det_ws) & 15) == 0 && d->det_ws_bytes >= (uint64_t)need,
                  "dmt_wgrad320: the ordered form needs a 16-byte aligned workspace of dmt_wgrad320_det_ws_bytes(M, N) bytes");
    g.ws = (float*)d->det_ws;
    g.ws_bias = g.ws + (long long)splits * DW_AW * g.ws_ld;
  }
  const int grid = ((splits + 7) / 8) * 8 * g.jblocks;
  hipStream_t st = (hipStream_t)stream;
#define DW_LAUNCH(BO, TR) hipLaunchKernelGGL((wgrad320_kernel<BO, TR>), dim3(grid), dim3(DW_NT), 0, st, g)
  if (d->bias_of == 0) { if (g.transposed) DW_LAUNCH(0, true); else DW_LAUNCH(0, false); }
  else if (d->bias_of == 1) { if (g.transposed) DW_LAUNCH(1, true); else DW_LAUNCH(1, false); }
  else { if (g.transposed) DW_LAUNCH(2, true); else DW_LAUNCH(2, false); }
#undef DW_LAUNCH
  DMT_CHECK_LAUNCH("dmt_wgrad320");
  if (g.ws != nullptr) {
    const long long n_all = (long long)DW_AW * g.N + g.ws_nb;
    hipLaunchKernelGGL(wgrad320_reduce_kernel, dim3((unsigned)cdiv64(n_all, 256)), dim3(256), 0, st, g);
    DMT_CHECK_LAUNCH("dmt_wgrad320(ordered reduce)");
  }
  return DMT_OK;
}
