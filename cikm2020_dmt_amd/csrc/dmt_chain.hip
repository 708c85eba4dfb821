// Two chained MFMA GEMMs whose intermediate never leaves the compute unit (bf16 operands, fp32 accumulation):
//
//     out[m, :] = epi( mid(  in[m, :] . A1^T  ) . A2^T )            in: [M, KIN]   mid: [M, NMID]   out: [M, NOUT]
//
//   mode DMT_CHAIN_FFN_LN    y = LN(relu(x W1 + b1) W2 + b2 + x)          ff() + ln() of TransformerModel_util.py:212-235, 58-78
//   mode DMT_CHAIN_FFN_BWD   dx = ((ds W2^T) * [h > 0]) W1^T + ds         its input gradient (the relu gate is a bit mask the
//                                                                          forward wrote: one bit per element of h)
//
// Shape of the computation (one workgroup = 8 wavefronts = 4 producer / consumer pairs = 128 rows, two wavefronts per SIMD):
//   * a pair owns 32 rows.  The producer keeps the input rows in REGISTERS as MFMA B fragments for the whole tile (KIN / 16 fragments)
//     and multiplies them with one 32-column slice of A1 per step; the consumer keeps the output tile out^T [NOUT x 32] in its
//     accumulators (NOUT / 32 tiles of 16 registers) and multiplies the handed-over mid tile with the matching slice of A2;
//   * the weights stream HBM/L2 -> LDS by DMA (buffer_load ... lds) as a ring of three stages; stage jt holds the 32 mid columns
//     jt*32 .. +32: the A1 rows that produce them (32 x KIN) and the A2 columns that consume them (NOUT x 32).  Both lie in a
//     prebuilt bf16 IMAGE (dmt_chain_image_build) in exactly the byte order of the LDS stage, row strides padded to an odd number of
//     16-byte slots (conflict-free ds_read_b128), so a stage is one linear 1 KB-per-instruction copy;
//   * everything is computed TRANSPOSED: mid^T tile [32 j x 32 m] = A1[j, :] . in^T  (A operand = weights from LDS, B operand = the
//     resident input fragments), so a lane holds ONE input row m = lane & 31 in every accumulator.  The mid tile feeds the second GEMM
//     straight from the accumulator registers as its B operand: the MFMA k-slot (half h, element e) of a lane holds
//     j = 4h + (e & 3) + 8 (e >> 2) of a 16-chunk, and the image stores A2 (and A1, for the residual) with the same permutation of k
//     inside every 16-chunk -- a reduction does not care about the order of its terms as long as both operands agree;
//   * with the same permutation applied to the input fragments, the residual x[m, n] that the epilogue adds to accumulator register
//     (tile t, r) is element 4 ((r >> 2) & 1) + (r & 3) of input fragment 2 t + (r >> 3): no data movement;
//   * LayerNorm row statistics: a row is spread over the two lanes m and m + 32 only, so mean / variance are in-lane sums plus one
//     exchange with the partner lane.
// Nothing but the input rows (once), the weight image (once per 128 rows, from L2) and the outputs touch memory; h is written
// only when the caller needs it for the weight gradients (training).
#include "dmt_common.h"
#include <utility>
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(3))) void* lds_vp;

constexpr int CH_NT = 512;   // 8 wavefronts = 4 producer / consumer pairs
constexpr int CH_NS = 3;     // DMA ring depth

template <int KIN_, int NMID_, int NOUT_>
struct Geo {
  static constexpr int KIN = KIN_, NMID = NMID_, NOUT = NOUT_;
  static constexpr int KC = KIN / 16;              // input fragments (k chunks of 16)
  static constexpr int NJT = NMID / 32;            // mid tiles = DMA stages per row tile
  static constexpr int NOT = (NOUT + 31) / 32;     // output tiles
  static constexpr int A1_STRIDE = KIN * 2 + 16;   // bytes; (KIN / 8 + 1) 16-byte slots: odd
  static constexpr int A1_BYTES = 32 * A1_STRIDE;
  static constexpr int A2_STRIDE = 80;             // 4 slots of data + 1 pad slot
  static constexpr int A2_OFF = A1_BYTES;
  static constexpr int A2_BYTES = NOT * 32 * A2_STRIDE;
  static constexpr int BIAS_OFF = A2_OFF + A2_BYTES;
  static constexpr int RAW = BIAS_OFF + 128;
  static constexpr int STAGE = (RAW + 8191) / 8192 * 8192;
  static constexpr int PER_WAVE = STAGE / 8192;    // DMA instructions (1 KB each) per wavefront and stage
  static constexpr int HAND_BYTES = 4 * 4096;      // producer -> consumer hand-off: [pair][slot 2][fragment 2][64 lanes][16 B]
  static constexpr int NST = NJT + 1;              // image stages: stage u = A1 / bias1 of mid tile u  |  A2 of mid tile u - 1
  static constexpr long long IMAGE_BYTES = (long long)NST * STAGE;
  static_assert(KIN % 16 == 0 && NMID % 32 == 0, "chain geometry");
  static_assert(CH_NS * STAGE + HAND_BYTES <= 160 * 1024, "LDS ring exceeds 160 KB");
};

template <int... I, typename F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(std::make_integer_sequence<int, N>{}, f); }

// k position p (0..15) of an image chunk holds original index perm16(p): the order in which the 32x32 MFMA accumulator hands its rows
// to a B operand (see the header).
__host__ __device__ constexpr int perm16(int p) { return 4 * (p >> 3) + (p & 3) + 8 * ((p >> 2) & 1); }

// ---------------------------------------------------------------------------------------------------------------- image
struct ImgArgs {
  const float* a1; long long a1_rs, a1_cs;   // A1[j, k] = a1[j * a1_rs + k * a1_cs]      j < NMID, k < KIN
  const float* a2; long long a2_rs, a2_cs;   // A2[n, j] = a2[n * a2_rs + j * a2_cs]      n < NOUT, j < NMID
  const float* bias1;                        // [NMID] or null
  unsigned char* img;
  int kin, nmid, nout;
  int a1_stride, a1_bytes, a2_bytes, bias_off, stage;
  long long slots;
};

__device__ __forceinline__ void chain_image_body(const ImgArgs& g) {
  for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < g.slots; s += (long long)gridDim.x * 256) {
    const long long byte = s * 16;
    const int u = (int)(byte / g.stage), off = (int)(byte % g.stage);   // stage u: A1 / bias1 of mid tile u, A2 of mid tile u - 1
    const int njt = g.nmid / 32;
    u32x4_t o = {0u, 0u, 0u, 0u};
    if (off < g.a1_bytes) {
      const int row = off / g.a1_stride, cb = off % g.a1_stride;
      if (cb < g.kin * 2 && u < njt) {
        const int jt = u;
        const int c = cb / 32, half = (cb % 32) / 16;
        unsigned short h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 16 * c + perm16(8 * half + e);
          h[e] = f2bf(g.a1[(long long)(jt * 32 + row) * g.a1_rs + (long long)k * g.a1_cs]);
        }
        o = u32x4_t{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16), (unsigned)h[4] | ((unsigned)h[5] << 16),
                    (unsigned)h[6] | ((unsigned)h[7] << 16)};
      }
    } else if (off < g.a1_bytes + g.a2_bytes) {
      const int o2 = off - g.a1_bytes;
      const int n = o2 / 80, sl = (o2 % 80) / 16;
      if (sl < 4 && n < g.nout && u >= 1) {
        const int jt = u - 1;
        const int f = sl >> 1, half = sl & 1;
        unsigned short h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int j = jt * 32 + 16 * f + perm16(8 * half + e);
          h[e] = f2bf(g.a2[(long long)n * g.a2_rs + (long long)j * g.a2_cs]);
        }
        o = u32x4_t{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16), (unsigned)h[4] | ((unsigned)h[5] << 16),
                    (unsigned)h[6] | ((unsigned)h[7] << 16)};
      }
    } else if (off < g.bias_off + 128) {
      const int i0 = (off - g.bias_off) / 4;
      if (g.bias1 != nullptr && u < njt) {
        const int jt = u;
        o = u32x4_t{__float_as_uint(g.bias1[jt * 32 + i0 + 0]), __float_as_uint(g.bias1[jt * 32 + i0 + 1]),
                    __float_as_uint(g.bias1[jt * 32 + i0 + 2]), __float_as_uint(g.bias1[jt * 32 + i0 + 3])};
      }
    }
    *reinterpret_cast<u32x4_t*>(g.img + byte) = o;
  }
}

__global__ __launch_bounds__(256) void chain_image_kernel(const ImgArgs g) { chain_image_body(g); }

// every weight image of a model in ONE launch: blockIdx.y = job (a table of ImgArgs in device memory, built once)
__global__ __launch_bounds__(256) void chain_image_batched_kernel(const ImgArgs* __restrict__ jobs) {
  const ImgArgs g = jobs[blockIdx.y];
  chain_image_body(g);
}

// ---------------------------------------------------------------------------------------------------------------- chain
struct ChainArgs {
  long long M;
  const bf16_t* in; long long ld_in;
  const unsigned char* image;
  const float* bias2; const float* gamma; const float* beta; float eps;
  bf16_t* s_out; bf16_t* y_out; long long ld_out;
  float* stats;
  bf16_t* mid_out; long long ld_mid;
  unsigned short* mask;
  int tiles;
  unsigned long long* trace;   // timing experiments (DBG & 512): per-wavefront cycle sums of the loop's phases, workgroup 0; else null
};

// shader-cycle stamp (s_memtime; the wait also drains the LDS queue: used only where that queue is empty anyway)
__device__ __forceinline__ unsigned long long ch_now() {
#if defined(__HIP_DEVICE_COMPILE__)      // (the lambdas that call this are compiled for the host as well: no device asm there)
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
#else
  return 0ull;
#endif
}

// LDS fragment reads are inline asm on purpose: hipcc makes every LDS access it can see wait vmcnt(0) while an LDS-DMA is in
// flight (it cannot tell the ring stages apart), which would serialise the weight stream with the multiplies.  LDS returns in
// order, so "at most n outstanding" releases the oldest reads; the waits name the registers they release ("+v") so that no MFMA is
// scheduled above them.
template <int OFF> __device__ __forceinline__ void ch_read128(bf16x8_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF> __device__ __forceinline__ void ch_read128f(f32x4_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N> __device__ __forceinline__ void ch_wait(bf16x8_t& a, bf16x8_t& b, bf16x8_t& c, bf16x8_t& d, bf16x8_t& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "i"(N));
}
__device__ __forceinline__ void ch_write128(unsigned addr, u32x4_t v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <int N> __device__ __forceinline__ void ch_wait2(bf16x8_t& a, bf16x8_t& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N)); }
__device__ __forceinline__ void ch_fake(bf16x8_t& r) { asm volatile("" : "=v"(r)); }
__device__ __forceinline__ void ch_keep(f32x16_t& h) { asm volatile("" : "+v"(h)); }
__device__ __forceinline__ void ch_keepb(bf16x8_t& h) { asm volatile("" : "+v"(h)); }
__device__ __forceinline__ void ch_keepu(unsigned& h) { asm volatile("" : "+v"(h)); }
template <int N> __device__ __forceinline__ void ch_wait8(bf16x8_t (&R)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7]) : "i"(N));
}
template <int N> __device__ __forceinline__ void ch_wait4f(f32x4_t& a, f32x4_t& b, f32x4_t& c, f32x4_t& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N));
}

__device__ __forceinline__ void swap_lo(unsigned& a, unsigned& b) {
  // v_permlane32_swap: lanes 32-63 of a  <->  lanes 0-31 of b
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
}

// DBG (timing experiments only, results are garbage): 1 no DMA, 2 no side-output stores (mid, mask), 4 no LDS fragment reads,
// 8 no MFMA, 16 no per-stage workgroup barrier, 32 no hand-off / relu work
template <typename G, int MODE, int DBG = 0>
__global__ __launch_bounds__(CH_NT, 2) void chain2_kernel(const ChainArgs g) {
  constexpr int KC = G::KC, NJT = G::NJT, NOT = G::NOT, STAGE = G::STAGE, PER_WAVE = G::PER_WAVE, NST = G::NST;
  constexpr int NF = 2 * NOT;   // GEMM-2 fragments (output tile, k half) per mid tile
  static_assert(PER_WAVE * 8192 == STAGE, "a stage is a whole number of 8 KB rounds");
  // 8 wavefronts = 4 producer / consumer pairs on the 4 SIMDs (wavefronts w and w + 4 share a SIMD), a pair owns 32 rows:
  //   producer (w < 4):  holds the input rows as B fragments, GEMM 1 of every mid tile, mid op, hands the tile over through LDS
  //   consumer (w >= 4): holds the output accumulators, GEMM 2 of every mid tile, epilogue (LayerNorm, stores)
  // Both sides issue 20 MFMAs per mid tile into the same matrix pipe; whatever one of them waits for (LDS reads, the accumulator
  // conversion, DMA issue, the dependent GEMM-1 chain) the other one's MFMAs fill.
  __shared__ __attribute__((aligned(16))) unsigned char smem[CH_NS * STAGE + G::HAND_BYTES];   // the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave & 3;
  const bool producer = wave < 4;
  const int ml = lane & 31, hi = lane >> 5;
  const unsigned lds0 = (unsigned)(unsigned long long)((lds_vp)smem);
  const unsigned a1_lane = lds0 + ml * G::A1_STRIDE + 16 * hi;
  const unsigned a2_lane = lds0 + G::A2_OFF + ml * G::A2_STRIDE + 16 * hi;
  const unsigned bias_lane = lds0 + G::BIAS_OFF + 16 * hi;
  const unsigned hand_lane = lds0 + CH_NS * STAGE + pair * 4096 + lane * 16;   // [slot 2][frag 2][64 lanes][16 B] per pair

  const __amdgpu_buffer_rsrc_t rimg =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(g.image), 0, (int)G::IMAGE_BYTES, 0x00020000);
  // The DMA ring is fed by the CONSUMER wavefronts only (PER_C pieces of 1 KB per wavefront and stage).  vmcnt retires in issue
  // order, loads and stores alike: a producer that issued ring pieces had to see its own side-output stores of the previous mid tile
  // (h, gate bits: partial-line writes, microseconds under load) retire before it could tell that a stage had landed.  Now a producer
  // never waits on vmcnt in the loop -- the barrier tells it a stage is there -- and the consumers' queue holds ring pieces only.
  constexpr int PER_C = 2 * PER_WAVE;
  constexpr int PER_C_EFF = (DBG & 2048) ? 10 : PER_C;      // (timing experiment 2048: a 40 KB stage -- what an unpadded image would stream)
  auto issue1 = [&](int buf, int u, int p) {
    if constexpr ((DBG & 1) != 0) return;
    if constexpr ((DBG & 2048) != 0) { if (p >= PER_C_EFF) return; }
    unsigned char* sb = smem + buf * STAGE + (wave - 4) * 1024 + p * 4096;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rimg, (lds_vp)sb, 16, lane * 16, u * STAGE + (wave - 4) * 1024 + p * 4096, 0, 0);
  };

  const int G_ = (int)gridDim.x;
  if ((int)blockIdx.x >= g.tiles) return;
  // static priority for the producers: their 20 MFMAs of a mid tile go first, so the accumulator conversion that follows them runs
  // beside the consumer's MFMAs instead of after them (the condition is wave-uniform: s_setprio ignores EXEC)
  if (wave < 4) __builtin_amdgcn_s_setprio(2);
  if (!producer) {
#pragma unroll
    for (int p = 0; p < PER_C; ++p) issue1(0, 0, p);
#pragma unroll
    for (int p = 0; p < PER_C; ++p) issue1(1, 1, p);
  }
  int gs = 0;   // global stage counter of this workgroup

  // producers: the raw input rows of the NEXT row tile are requested in the middle of the current tile's loop (one workgroup per CU:
  // nothing else would hide the ~2 us of these row-strided loads at a tile's start)
  u32x4_t Xn[KC];
  auto request_rows = [&](int tile_) {
    const long long m_ = (long long)tile_ * 128 + pair * 32 + ml;
    const bf16_t* xr = g.in + (m_ < g.M ? m_ : (g.M - 1)) * g.ld_in + 8 * hi;
#pragma unroll
    for (int c = 0; c < KC; ++c) Xn[c] = *reinterpret_cast<const u32x4_t*>(xr + 16 * c);
  };
  if (producer) request_rows((int)blockIdx.x);

  if (producer) {
    for (int tile = (int)blockIdx.x; tile < g.tiles; tile += G_) {
      const long long row0 = (long long)tile * 128 + pair * 32;
      const long long m = row0 + ml;
      const bool mvalid = m < g.M;
      const long long mc = mvalid ? m : (g.M - 1);
      const long long blk = row0 >> 5;

      // top of an iteration, both roles: stage gs has landed (the consumers, which issued its pieces, have seen them retire) and
      // everybody has left stage gs - 1
      constexpr bool TRACE = (DBG & 1024) != 0;
      unsigned long long tr_t = 0, tr_acc[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
      if constexpr (TRACE) tr_t = ch_now();
      auto tr_mark = [&](int ph) {          // phase ph ends here (summed over the stages of a tile; workgroup 0's first tile is reported)
        if constexpr (TRACE) { const unsigned long long n = ch_now(); tr_acc[ph] += n - tr_t; tr_t = n; }
      };
      auto top = [&](int u) {
        if (!producer) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PER_C_EFF) : "memory");      // at most the pieces of stage gs + 1 are open
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (TRACE) tr_mark(0);                                                  // 0: own LDS queue drained (hand-off writes)
        if constexpr ((DBG & 16) == 0) __builtin_amdgcn_s_barrier();
        if constexpr (TRACE) tr_mark(1);                                                  // 1: barrier
        (void)u;
      };

      // ---- input rows -> B fragments (k permuted inside every 16-chunk: the lower lane takes elements 0-3 | 8-11, the upper 4-7 | 12-15)
      bf16x8_t X[KC];
      {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          const u32x4_t vl = Xn[c];
          uint4 v = make_uint4(vl[0], vl[1], vl[2], vl[3]);
          if (!mvalid) v = make_uint4(0u, 0u, 0u, 0u);
          swap_lo(v.x, v.z);
          swap_lo(v.y, v.w);
          X[c] = __builtin_bit_cast(bf16x8_t, v);
        }
      }
      unsigned bits_next = 0;
      if constexpr (MODE == DMT_CHAIN_FFN_BWD) bits_next = g.mask[(blk * NJT + 0) * 64 + lane];
      if constexpr (TRACE) { ch_keepb(X[0]); ch_keepb(X[KC - 1]); tr_mark(5); }     // 5: prologue (rows -> fragments)
#pragma unroll 1
      for (int u = 0; u < ((DBG & 64) ? 1 : NJT); ++u, ++gs) {
        const int buf = gs % CH_NS;
        top(u);
        if (u == NJT / 4 && tile + G_ < g.tiles) request_rows(tile + G_);
        unsigned bits = bits_next;
        if constexpr (MODE == DMT_CHAIN_FFN_BWD) {
          bits_next = g.mask[(blk * NJT + (u + 1 < NJT ? u + 1 : u)) * 64 + lane];   // gate bits of the next tile, older than the DMA below
          asm volatile("" ::: "memory");
        }
        const unsigned so = (unsigned)buf * STAGE;
        const unsigned a1a = a1_lane + so;
        constexpr int PB = (KC % 5 == 0) ? 5 : 4, NBAT = KC / PB;
        static_assert(KC % PB == 0, "KIN / 16 must be a multiple of 4 or 5");
        bf16x8_t R0[5], R1[5];
        f32x4_t b4[4];
        f32x16_t Ha;   // (one accumulator chain: the consumer's MFMAs on the same SIMD fill its dependency gaps)
        auto rd = [&](bf16x8_t (&R)[5], auto bic) {
          constexpr int b = decltype(bic)::value;
          sfor<5>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int c = b * PB + (i < PB ? i : 0);
            if constexpr ((DBG & 4) != 0) ch_fake(R[i]); else ch_read128<c * 32>(R[i], a1a);
          });
        };
        auto mm = [&](bf16x8_t (&R)[5], auto bic) {
          constexpr int b = decltype(bic)::value;
          sfor<PB>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int c = b * PB + i;
            if constexpr ((DBG & 8) != 0) Ha[i] += __builtin_bit_cast(f32x4_t, R[i])[0] + __builtin_bit_cast(f32x4_t, X[c])[1];
            else Ha = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R[i], X[c], Ha, 0, 0, 0);
          });
        };
        if constexpr (MODE == DMT_CHAIN_FFN_LN) {
          sfor<4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            ch_read128f<q * 32>(b4[q], bias_lane + so);
          });
        }
        rd(R0, std::integral_constant<int, 0>{});
        sfor<NBAT>([&](auto bic) {
          constexpr int b = decltype(bic)::value;
          if constexpr (b + 1 < NBAT) {
            if constexpr ((b & 1) == 0) rd(R1, std::integral_constant<int, b + 1>{});
            else rd(R0, std::integral_constant<int, b + 1>{});
          }
          if constexpr (b == 0) {
            if constexpr (MODE == DMT_CHAIN_FFN_LN) {
              ch_wait4f<(NBAT > 1 ? 10 : 5)>(b4[0], b4[1], b4[2], b4[3]);
#pragma unroll
              for (int r = 0; r < 16; ++r) Ha[r] = b4[r >> 2][r & 3];   // bias1 is the accumulator's initial value
            } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) Ha[r] = 0.f;
            }
          }
          if constexpr (b + 1 < NBAT) {
            if constexpr ((b & 1) == 0) { ch_wait<5>(R0[0], R0[1], R0[2], R0[3], R0[4]); mm(R0, bic); }
            else { ch_wait<5>(R1[0], R1[1], R1[2], R1[3], R1[4]); mm(R1, bic); }
          } else {
            if constexpr ((b & 1) == 0) { ch_wait<0>(R0[0], R0[1], R0[2], R0[3], R0[4]); mm(R0, bic); }
            else { ch_wait<0>(R1[0], R1[1], R1[2], R1[3], R1[4]); mm(R1, bic); }
          }
        });
        if constexpr (TRACE) tr_mark(2);                                                  // 2: fragment reads + MFMA issue
        // ---- mid op on the accumulator: lane (m, hi) holds j = 32 u + 8 q + 4 hi + i in register 4 q + i
        f32x16_t H = Ha;
        unsigned hb[8], ho[8];
        if constexpr (MODE == DMT_CHAIN_FFN_LN) {
          // relu and its mask on the PACKED pairs: bf16 keeps the sign bit, so max(int16, 0) is the relu of either half, and a half that is
          // not zero afterwards was positive (a positive fp32 below bf16's smallest denormal rounds to +0: its relu output is 0 either way).
          // 8 conversions + 8 packed max + 2 per pair for the bits, where the fp32 form took a compare, two selects and an or per element.
          typedef __attribute__((ext_vector_type(2))) short s16x2_t;
          typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;
          unsigned acc = 0;
#pragma unroll
          for (int p = 0; p < 8; ++p) {
            const s16x2_t v = __builtin_bit_cast(s16x2_t, dmt_pack_bf16(H[2 * p], H[2 * p + 1]));
            const s16x2_t zero = {0, 0};
            const s16x2_t rl = __builtin_elementwise_max(v, zero);
            hb[p] = __builtin_bit_cast(unsigned, rl);
            const u16x2_t one = {1, 1};
            const unsigned nz = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, rl), one));   // bit 0 / bit 16
            acc |= nz << (2 * p);
          }
          bits = (acc & 0xFFFFu) | (acc >> 15);        // element 2p -> bit 2p, element 2p + 1 -> bit 2p + 1
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) H[r] = ((bits >> r) & 1u) ? H[r] : 0.f;
#pragma unroll
          for (int p = 0; p < 8; ++p) hb[p] = dmt_pack_bf16(H[2 * p], H[2 * p + 1]);
        }
        if constexpr (TRACE) { ch_keepu(hb[0]); ch_keepu(hb[7]); tr_mark(3); }   // 3: MFMA drain + mid op + pack
        // hand the tile to the consumer: two B fragments, slot u & 1
        ch_write128(hand_lane + (u & 1) * 2048, u32x4_t{hb[0], hb[1], hb[2], hb[3]});
        ch_write128(hand_lane + (u & 1) * 2048 + 1024, u32x4_t{hb[4], hb[5], hb[6], hb[7]});
        if constexpr ((DBG & 2) == 0) {
          if constexpr (MODE == DMT_CHAIN_FFN_LN) {
            if (g.mask != nullptr) g.mask[(blk * NJT + u) * 64 + lane] = (unsigned short)bits;
          }
          if (g.mid_out != nullptr) {
#pragma unroll
            for (int p = 0; p < 8; ++p) ho[p] = hb[p];
            // row m, columns 16 p + 8 hi .. +8 of the tile after pairing q = 2p (kept by the lower lane) with q = 2p + 1 (upper lane)
            swap_lo(ho[0], ho[2]); swap_lo(ho[1], ho[3]);
            swap_lo(ho[4], ho[6]); swap_lo(ho[5], ho[7]);
            if constexpr ((DBG & 256) != 0) {
              // (timing experiment: the same bytes as fully coalesced 1 KB stores, garbage layout)
              bf16_t* dst = g.mid_out + ((blk * NJT + u) * 2) * 512 + lane * 8;
              *reinterpret_cast<u32x4_t*>(dst) = u32x4_t{ho[0], ho[1], ho[2], ho[3]};
              *reinterpret_cast<u32x4_t*>(dst + 512) = u32x4_t{ho[4], ho[5], ho[6], ho[7]};
            } else if (mvalid) {
              bf16_t* dst = g.mid_out + m * g.ld_mid + u * 32 + 8 * hi;
              // (plain stores: a row receives 64 bytes per mid tile, the L2 merges the pieces of a line; non-temporal partial-line
              //  stores were 4x slower)
              *reinterpret_cast<u32x4_t*>(dst) = u32x4_t{ho[0], ho[1], ho[2], ho[3]};
              *reinterpret_cast<u32x4_t*>(dst + 16) = u32x4_t{ho[4], ho[5], ho[6], ho[7]};
            }
          }
        }
        if constexpr (TRACE) tr_mark(4);                                                  // 4: hand-off writes + side-output stores issued
      }
      // (iteration NJT: the consumer multiplies the last mid tile)
      top(NJT);
      ++gs;
      if constexpr (TRACE) {
        if (blockIdx.x == 0 && tile == 0 && lane == 0 && g.trace != nullptr)
          for (int i = 0; i < 6; ++i) g.trace[wave * 8 + i] = tr_acc[i];
      }
    }
  } else {
    for (int tile = (int)blockIdx.x; tile < g.tiles; tile += G_) {
      const long long row0 = (long long)tile * 128 + pair * 32;
      const long long m = row0 + ml;
      const bool mvalid = m < g.M;
      const long long mc = mvalid ? m : (g.M - 1);
      const long long blk = row0 >> 5;

      // top of an iteration, both roles: stage gs has landed (the consumers, which issued its pieces, have seen them retire) and
      // everybody has left stage gs - 1
      constexpr bool TRACE = (DBG & 1024) != 0;
      unsigned long long tr_t = 0, tr_acc[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
      if constexpr (TRACE) tr_t = ch_now();
      auto tr_mark = [&](int ph) {
        if constexpr (TRACE) { const unsigned long long n = ch_now(); tr_acc[ph] += n - tr_t; tr_t = n; }
      };
      auto top = [&](int u) {
        if (!producer) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PER_C_EFF) : "memory");      // at most the pieces of stage gs + 1 are open
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (TRACE) tr_mark(0);                                                  // 0: my DMA pieces of this stage have landed (vmcnt)
        if constexpr ((DBG & 16) == 0) __builtin_amdgcn_s_barrier();
        if constexpr (TRACE) tr_mark(1);                                                  // 1: barrier
        (void)u;
      };

      // ---- consumer: out^T accumulators start as bias2 + residual (register (t, 4 q + i) of lane (m, hi) is column 32 t + 8 q + 4 hi + i)
      f32x16_t Y[NOT];
      {
        // Every residual piece is requested back to back from a VALID address (row clamped to M - 1, column to 0) and masked afterwards:
        // a load under a per-lane predicate is waited for (vmcnt(0)) before hipcc issues the next one -- 20 serial round trips per
        // tile, ~90 us of a 600 us launch (seen in the ISA; DESIGN.md section 3 "a recurring compiler effect").
        const bf16_t* xrow = g.in + mc * g.ld_in;
        u32x4_t xraw[2 * NOT];
#pragma unroll
        for (int t = 0; t < NOT; ++t)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int col = 32 * t + 16 * p + 8 * hi;
            xraw[2 * t + p] = *reinterpret_cast<const u32x4_t*>(xrow + (col < G::KIN ? col : 0));
          }
        sfor<NOT>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          unsigned xv[8];
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int col = 32 * t + 16 * p + 8 * hi;
            const bool ok = col < G::KIN && mvalid;
            const u32x4_t v = xraw[2 * t + p];
            xv[4 * p + 0] = ok ? v[0] : 0u; xv[4 * p + 1] = ok ? v[1] : 0u; xv[4 * p + 2] = ok ? v[2] : 0u; xv[4 * p + 3] = ok ? v[3] : 0u;
          }
          // undo the store pairing: the lower lane holds columns 16 p + 0..7, the upper 16 p + 8..15; register group q wants 8 q + 4 hi + 0..3
          swap_lo(xv[0], xv[2]); swap_lo(xv[1], xv[3]);
          swap_lo(xv[4], xv[6]); swap_lo(xv[5], xv[7]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n0 = 32 * t + 8 * q + 4 * hi;
            f32x4_t b = {0.f, 0.f, 0.f, 0.f};
            if (MODE == DMT_CHAIN_FFN_LN && n0 < G::NOUT) b = *reinterpret_cast<const f32x4_t*>(g.bias2 + n0);
            Y[t][4 * q + 0] = b[0] + __uint_as_float(xv[2 * q] << 16);
            Y[t][4 * q + 1] = b[1] + __uint_as_float(xv[2 * q] & 0xFFFF0000u);
            Y[t][4 * q + 2] = b[2] + __uint_as_float(xv[2 * q + 1] << 16);
            Y[t][4 * q + 3] = b[3] + __uint_as_float(xv[2 * q + 1] & 0xFFFF0000u);
          }
        });
      }
      if constexpr (TRACE) { ch_keep(Y[0]); ch_keep(Y[NOT - 1]); tr_mark(5); }   // 5: prologue (residual rows -> accumulators)
      // (iteration 0: nothing produced yet; the consumer only keeps the ring going)
      top(0);
      {
        const int dbuf = (gs + 2) % CH_NS, du = 2 % NST;
#pragma unroll
        for (int p = 0; p < PER_C; ++p) issue1(dbuf, du, p);
      }
      ++gs;
#pragma unroll 1
      for (int u = 1; u <= ((DBG & 64) ? 1 : NJT); ++u, ++gs) {
        const int buf = gs % CH_NS;
        top(u);
        const int dbuf = (gs + 2) % CH_NS, du = (u + 2) % NST;
        const unsigned so = (unsigned)buf * STAGE;
        const unsigned a2a = a2_lane + so;
        constexpr int PB = (NF % 5 == 0) ? 5 : ((NF % 4 == 0) ? 4 : 3), NBAT = NF / PB;
        static_assert(NF % PB == 0, "2 * output tiles must be a multiple of 3, 4 or 5");
        bf16x8_t R0[5], R1[5];
        bf16x8_t Hb0, Hb1;
        ch_read128<0>(Hb0, hand_lane + ((u - 1) & 1) * 2048);
        ch_read128<1024>(Hb1, hand_lane + ((u - 1) & 1) * 2048);
        auto rd = [&](bf16x8_t (&R)[5], auto bic) {
          constexpr int b = decltype(bic)::value;
          sfor<5>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int fi = b * PB + (i < PB ? i : 0);
            constexpr int t = fi >> 1, f = fi & 1;
            if constexpr ((DBG & 4) != 0) ch_fake(R[i]); else ch_read128<t * 32 * G::A2_STRIDE + f * 32>(R[i], a2a);
          });
        };
        auto mm = [&](bf16x8_t (&R)[5], auto bic) {
          constexpr int b = decltype(bic)::value;
          sfor<PB>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int fi = b * PB + i;
            constexpr int t = fi >> 1, f = fi & 1;
            if constexpr ((DBG & 8) != 0) Y[t][i] += __builtin_bit_cast(f32x4_t, R[i])[0] + __builtin_bit_cast(f32x4_t, f == 0 ? Hb0 : Hb1)[1];
            else Y[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R[i], f == 0 ? Hb0 : Hb1, Y[t], 0, 0, 0);
          });
        };
        rd(R0, std::integral_constant<int, 0>{});
        sfor<NBAT>([&](auto bic) {
          constexpr int b = decltype(bic)::value;
          if constexpr (b + 1 < NBAT) {
            if constexpr ((b & 1) == 0) rd(R1, std::integral_constant<int, b + 1>{});
            else rd(R0, std::integral_constant<int, b + 1>{});
          }
#pragma unroll
          for (int p = (PER_C * b) / NBAT; p < (PER_C * (b + 1)) / NBAT; ++p) issue1(dbuf, du, p);
          if constexpr (b == 0) ch_wait2<(NBAT > 1 ? 10 : 5)>(Hb0, Hb1);
          if constexpr (b + 1 < NBAT) {
            if constexpr ((b & 1) == 0) { ch_wait<5>(R0[0], R0[1], R0[2], R0[3], R0[4]); mm(R0, bic); }
            else { ch_wait<5>(R1[0], R1[1], R1[2], R1[3], R1[4]); mm(R1, bic); }
          } else {
            if constexpr ((b & 1) == 0) { ch_wait<0>(R0[0], R0[1], R0[2], R0[3], R0[4]); mm(R0, bic); }
            else { ch_wait<0>(R1[0], R1[1], R1[2], R1[3], R1[4]); mm(R1, bic); }
          }
        });
        if constexpr (TRACE) tr_mark(2);                                                  // 2: fragment reads + DMA issue + MFMA issue
      }

      // ---- epilogue (consumer): LayerNorm over the row (spread over the lanes m and m + 32), stores
      float mean = 0.f, rstd = 1.f;
      if constexpr (MODE == DMT_CHAIN_FFN_LN) {
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NOT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = 32 * t + 8 * (r >> 2) + 4 * hi + (r & 3);
            if constexpr (G::NOUT % 32 != 0) { if (n >= G::NOUT) Y[t][r] = 0.f; }
            sum += Y[t][r];
          }
        sum += __shfl_xor(sum, 32, 64);
        mean = sum / (float)G::NOUT;
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < NOT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = 32 * t + 8 * (r >> 2) + 4 * hi + (r & 3);
            const float d = (n < G::NOUT) ? (Y[t][r] - mean) : 0.f;
            sq += d * d;
          }
        sq += __shfl_xor(sq, 32, 64);
        const float den = sqrtf(sq / (float)G::NOUT + g.eps);
        rstd = 1.f / den;
        if (g.stats != nullptr && mvalid && hi == 0) { g.stats[2 * m] = mean; g.stats[2 * m + 1] = rstd; }
      }
      // stores: 16 bytes per lane after pairing q = 2p (lower lane) with q = 2p + 1 (upper lane): row m, columns 32 t + 16 p + 8 hi .. +8
      sfor<NOT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        unsigned su[8], yu[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          su[2 * q] = dmt_pack_bf16(Y[t][4 * q + 0], Y[t][4 * q + 1]);
          su[2 * q + 1] = dmt_pack_bf16(Y[t][4 * q + 2], Y[t][4 * q + 3]);
          if constexpr (MODE == DMT_CHAIN_FFN_LN) {
            const int n0 = 32 * t + 8 * q + 4 * hi;
            f32x4_t gm = {0.f, 0.f, 0.f, 0.f}, bt = {0.f, 0.f, 0.f, 0.f};
            if (n0 < G::NOUT) { gm = *reinterpret_cast<const f32x4_t*>(g.gamma + n0); bt = *reinterpret_cast<const f32x4_t*>(g.beta + n0); }
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = gm[i] * ((Y[t][4 * q + i] - mean) * rstd) + bt[i];
            yu[2 * q] = dmt_pack_bf16(o[0], o[1]);
            yu[2 * q + 1] = dmt_pack_bf16(o[2], o[3]);
          }
        }
        swap_lo(su[0], su[2]); swap_lo(su[1], su[3]);
        swap_lo(su[4], su[6]); swap_lo(su[5], su[7]);
        if constexpr (MODE == DMT_CHAIN_FFN_LN) {
          swap_lo(yu[0], yu[2]); swap_lo(yu[1], yu[3]);
          swap_lo(yu[4], yu[6]); swap_lo(yu[5], yu[7]);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int col = 32 * t + 16 * p + 8 * hi;
          if (mvalid && col < G::NOUT && ((DBG & 128) == 0 || su[0] == 0x12345u)) {
            if (g.s_out != nullptr)
              *reinterpret_cast<u32x4_t*>(g.s_out + m * g.ld_out + col) = u32x4_t{su[4 * p], su[4 * p + 1], su[4 * p + 2], su[4 * p + 3]};
            if constexpr (MODE == DMT_CHAIN_FFN_LN)
              *reinterpret_cast<u32x4_t*>(g.y_out + m * g.ld_out + col) = u32x4_t{yu[4 * p], yu[4 * p + 1], yu[4 * p + 2], yu[4 * p + 3]};
          }
        }
      });
      if constexpr (TRACE) {
        tr_mark(3);                                                                       // 3: MFMA drain + epilogue (LayerNorm, stores issued)
        if (blockIdx.x == 0 && tile == 0 && lane == 0 && g.trace != nullptr)
          for (int i = 0; i < 6; ++i) g.trace[wave * 8 + i] = tr_acc[i];
      }
    }
  }
  // the ring runs two stages ahead of the last multiply: let those pieces land before the workgroup (and its LDS) goes away
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename G>
int launch_chain(const dmt_chain_desc* d, hipStream_t st) {
  ChainArgs a;
  a.M = d->M;
  a.in = (const bf16_t*)d->in; a.ld_in = d->ld_in;
  a.image = (const unsigned char*)d->image;
  a.bias2 = d->bias2; a.gamma = d->gamma; a.beta = d->beta; a.eps = d->eps;
  a.s_out = (bf16_t*)d->s_out; a.y_out = (bf16_t*)d->y_out; a.ld_out = d->ld_out;
  a.stats = d->stats;
  a.mid_out = (bf16_t*)d->mid_out; a.ld_mid = d->ld_mid;
  a.mask = (unsigned short*)d->mask;
  a.tiles = (int)cdiv64(d->M, 128);
  a.trace = nullptr;
  const int grid = a.tiles < 256 ? a.tiles : 256;
#ifdef DMT_TIMING_EXPERIMENTS   // (scripts/ ablations only: `make EXPERIMENTS=1`; the shipped library reads no environment)
  if constexpr (G::KIN == 320) {
    const char* dbg = getenv("DMT_CHAIN_DEBUG");   // timing experiments (see DBG above)
    const int v = dbg ? atoi(dbg) : 0;
    if (v != 0 && d->mode == DMT_CHAIN_FFN_LN) {
#define DMT_CHAIN_DBG(V) case V: hipLaunchKernelGGL((chain2_kernel<G, DMT_CHAIN_FFN_LN, V>), dim3(grid), dim3(CH_NT), 0, st, a); break;
      if (v == 2048) { hipLaunchKernelGGL((chain2_kernel<G, DMT_CHAIN_FFN_LN, 2048>), dim3(grid), dim3(CH_NT), 0, st, a); DMT_CHECK_LAUNCH("dmt_chain2(debug)"); return DMT_OK; }
      if (v == 1024) {
        // cycle sums of the loop's phases, per wavefront, of workgroup 0's first tile (printed after the launch: this variant synchronises)
        static unsigned long long* tr = nullptr;
        if (tr == nullptr) (void)hipMalloc(&tr, 64 * sizeof(unsigned long long));
        (void)hipMemsetAsync(tr, 0, 64 * sizeof(unsigned long long), st);
        a.trace = tr;
        hipLaunchKernelGGL((chain2_kernel<G, DMT_CHAIN_FFN_LN, 1024>), dim3(grid), dim3(CH_NT), 0, st, a);
        unsigned long long h[64];
        (void)hipMemcpyAsync(h, tr, sizeof(h), hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        static int printed = 0;
        if (printed++ % 13 == 12) {
          fprintf(stderr, "chain2 trace (cycles per tile; producer: lgkm | barrier | reads+mfma issue | drain+midop | writes+stores | prologue; consumer: vmcnt | barrier | body | epilogue | - | prologue)\n");
          for (int w = 0; w < 8; ++w)
            fprintf(stderr, "  wave %d (%s): %8llu %8llu %8llu %8llu %8llu %8llu\n", w, w < 4 ? "producer" : "consumer", h[w * 8 + 0], h[w * 8 + 1], h[w * 8 + 2],
                    h[w * 8 + 3], h[w * 8 + 4], h[w * 8 + 5]);
        }
        return DMT_OK;
      }
      switch (v) {
        DMT_CHAIN_DBG(1) DMT_CHAIN_DBG(2) DMT_CHAIN_DBG(3) DMT_CHAIN_DBG(4) DMT_CHAIN_DBG(8) DMT_CHAIN_DBG(16) DMT_CHAIN_DBG(7) DMT_CHAIN_DBG(11)
        DMT_CHAIN_DBG(15) DMT_CHAIN_DBG(31) DMT_CHAIN_DBG(95) DMT_CHAIN_DBG(223) DMT_CHAIN_DBG(159) DMT_CHAIN_DBG(128) DMT_CHAIN_DBG(130) DMT_CHAIN_DBG(256) DMT_CHAIN_DBG(384)
        default: dmt_set_error("dmt_chain2: DMT_CHAIN_DEBUG=%d is not compiled", v); return DMT_ERR_ARG;
      }
#undef DMT_CHAIN_DBG
      DMT_CHECK_LAUNCH("dmt_chain2(debug)");
      return DMT_OK;
    }
  }
#endif
  if (d->mode == DMT_CHAIN_FFN_LN)
    hipLaunchKernelGGL((chain2_kernel<G, DMT_CHAIN_FFN_LN>), dim3(grid), dim3(CH_NT), 0, st, a);
  else
    hipLaunchKernelGGL((chain2_kernel<G, DMT_CHAIN_FFN_BWD>), dim3(grid), dim3(CH_NT), 0, st, a);
  DMT_CHECK_LAUNCH("dmt_chain2");
  return DMT_OK;
}

// ---------------------------------------------------------------------------------------------------------------- projection
// GEMM 1 of the chain alone:  out[m, :] = in[m, :] . W + b  for long-row activations (the packed Q | K | V projection).  The plain
// tiled GEMM re-reads a 128-row activation tile once per 128 output columns and synchronises every 64 k; here the wavefront's 32 input
// rows are loaded ONCE into registers (KIN / 16 MFMA B fragments) and the weights stream through the LDS ring as 32-column slices of a
// prebuilt image -- one barrier per 32 x KIN slice, output tiles stored straight from the accumulators.  4 compute wavefronts =
// 128 rows per workgroup + a loader wavefront (320 threads), one workgroup per CU (72 KB of LDS: kernels of the other lanes fit beside it); the rows
// of the next 128-row unit are requested before the slice loop of the current one.
template <int KIN_, int N_>
struct PGeo {
  static constexpr int KIN = KIN_, N = N_;
  static constexpr int KC = KIN / 16, NJT = N / 32;
  static constexpr int A1_STRIDE = KIN * 2 + 16, A1_BYTES = 32 * A1_STRIDE, BIAS_OFF = A1_BYTES;
  static constexpr int RAW = BIAS_OFF + 128;
  static constexpr int STAGE = (RAW + 4095) / 4096 * 4096;
  static constexpr int PER_WAVE = STAGE / 4096;          // 4 wavefronts x 1 KB per DMA instruction
  static constexpr long long IMAGE_BYTES = (long long)NJT * STAGE;
  static_assert(KIN % 16 == 0 && N % 32 == 0, "projection geometry");
  static_assert(CH_NS * STAGE <= 160 * 1024, "LDS ring");
};

constexpr int PJ_NT = 320;   // 4 compute wavefronts (32 rows each) + the loader

struct ProjArgs {
  long long M;
  const bf16_t* in; long long ld_in;
  const unsigned char* image;
  bf16_t* out; long long ld_out;
  int tiles;
};

// DBG (timing experiments only, results are garbage): 1 no DMA, 2 no output stores, 4 no LDS fragment reads, 8 no MFMA
template <typename G, int DBG = 0>
__global__ __launch_bounds__(PJ_NT, 1) void proj_kernel(const ProjArgs g) {
  constexpr int KC = G::KC, NJT = G::NJT, STAGE = G::STAGE, PER_WAVE = G::PER_WAVE;
  __shared__ __attribute__((aligned(16))) unsigned char smem[CH_NS * STAGE];   // the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ml = lane & 31, hi = lane >> 5;
  const unsigned lds0 = (unsigned)(unsigned long long)((lds_vp)smem);
  const unsigned a1_lane = lds0 + ml * G::A1_STRIDE + 16 * hi;
  const unsigned bias_lane = lds0 + G::BIAS_OFF + 16 * hi;
  const __amdgpu_buffer_rsrc_t rimg =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(g.image), 0, (int)G::IMAGE_BYTES, 0x00020000);
  // The weight stream belongs to ONE extra wavefront (wave 4, the loader): vmcnt counts loads AND stores, and a store leaves it only
  // when the L2 has taken the data -- a compute wavefront that waited for "my DMA pieces have landed" also sat out its own output
  // stores of the previous slice (measured: 78 of 220 us).  The loader has only DMA in flight, so "at most one stage open" is exact;
  // the compute wavefronts never wait on vmcnt inside the slice loop.
  auto issue_stage = [&](int buf, int u) {
    if constexpr ((DBG & 1) != 0) return;
#pragma unroll
    for (int p = 0; p < STAGE / 1024; ++p) {
      unsigned char* sb = smem + buf * STAGE + p * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rimg, (lds_vp)sb, 16, lane * 16, u * STAGE + p * 1024, 0, 0);
    }
  };
  const bool loader = wave == 4;
  const int G_ = (int)gridDim.x;
  if ((int)blockIdx.x >= g.tiles) return;
  if (loader) {
    issue_stage(0, 0);
    issue_stage(1, 1 % NJT);
    int gs = 0;
    for (int tile = (int)blockIdx.x; tile < g.tiles; tile += G_) {
#pragma unroll 1
      for (int u = 0; u < ((DBG & 32) != 0 ? 1 : NJT); ++u, ++gs) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(STAGE / 1024) : "memory");   // stage gs has landed (stage gs + 1 may be open)
        if constexpr ((DBG & 64) == 0) __builtin_amdgcn_s_barrier();           // everybody has left stage gs - 1 = ring slot gs + 2
        issue_stage((gs + 2) % CH_NS, (u + 2) % NJT);
      }
    }
    return;
  }
  int gs = 0;
  // The rows of the NEXT tile are requested before this tile's slice loop and turned into fragments after it: their round trip (a
  // lane reads its own row in 16-byte pieces: 32 rows x 32 bytes per instruction, 63 us per launch when it was waited for at the
  // top of every tile) runs under the 30 slices.  160 registers of rows -> one workgroup per CU (two wavefronts per SIMD).
  u32x4_t XR[KC];
  auto load_rows = [&](int tile, u32x4_t (&R)[KC]) {
    const long long mm = (long long)tile * 128 + wave * 32 + ml;
    const bf16_t* xr = g.in + (mm < g.M ? mm : g.M - 1) * g.ld_in + 8 * hi;
#pragma unroll
    for (int c = 0; c < KC; ++c) R[c] = *reinterpret_cast<const u32x4_t*>(xr + 16 * c);
  };
  load_rows((int)blockIdx.x, XR);
  for (int tile = (int)blockIdx.x; tile < g.tiles; tile += G_) {
    const long long m = (long long)tile * 128 + wave * 32 + ml;
    const bool mvalid = m < g.M;
    bf16x8_t X[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      uint4 v = make_uint4(XR[c][0], XR[c][1], XR[c][2], XR[c][3]);
      swap_lo(v.x, v.z);
      swap_lo(v.y, v.w);
      X[c] = __builtin_bit_cast(bf16x8_t, v);
    }
    if (tile + G_ < g.tiles) load_rows(tile + G_, XR);
#pragma unroll 1
    for (int u = 0; u < ((DBG & 32) != 0 ? 1 : NJT); ++u, ++gs) {
      const int buf = gs % CH_NS;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr ((DBG & 64) == 0) __builtin_amdgcn_s_barrier();       // the loader has seen stage gs land; everybody has left stage gs - 1
      const unsigned so = (unsigned)buf * STAGE;
      const unsigned a1a = a1_lane + so;
      constexpr int PB = (KC % 5 == 0) ? 5 : 4, NBAT = KC / PB;
      static_assert(KC % PB == 0, "KIN / 16 must be a multiple of 4 or 5");
      bf16x8_t R0[5], R1[5];
      f32x4_t b4[4];
      f32x16_t Ha;
      auto rd = [&](bf16x8_t (&R)[5], auto bic) {
        constexpr int b = decltype(bic)::value;
        sfor<5>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          constexpr int c = b * PB + (i < PB ? i : 0);
          if constexpr ((DBG & 4) != 0) ch_fake(R[i]); else ch_read128<c * 32>(R[i], a1a);
        });
      };
      auto mm = [&](bf16x8_t (&R)[5], auto bic) {
        constexpr int b = decltype(bic)::value;
        sfor<PB>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          constexpr int c = b * PB + i;
          if constexpr ((DBG & 8) != 0) { Ha[i] += __builtin_bit_cast(f32x4_t, R[i])[0] + __builtin_bit_cast(f32x4_t, X[c])[1]; }
          else Ha = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R[i], X[c], Ha, 0, 0, 0);
        });
      };
      sfor<4>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        ch_read128f<q * 32>(b4[q], bias_lane + so);
      });
      rd(R0, std::integral_constant<int, 0>{});
      sfor<NBAT>([&](auto bic) {
        constexpr int b = decltype(bic)::value;
        if constexpr (b + 1 < NBAT) {
          if constexpr ((b & 1) == 0) rd(R1, std::integral_constant<int, b + 1>{});
          else rd(R0, std::integral_constant<int, b + 1>{});
        }
        if constexpr (b == 0) {
          ch_wait4f<(NBAT > 1 ? 10 : 5)>(b4[0], b4[1], b4[2], b4[3]);
#pragma unroll
          for (int r = 0; r < 16; ++r) Ha[r] = b4[r >> 2][r & 3];   // the bias is the accumulator's initial value
        }
        if constexpr (b + 1 < NBAT) {
          if constexpr ((b & 1) == 0) { ch_wait<5>(R0[0], R0[1], R0[2], R0[3], R0[4]); mm(R0, bic); }
          else { ch_wait<5>(R1[0], R1[1], R1[2], R1[3], R1[4]); mm(R1, bic); }
        } else {
          if constexpr ((b & 1) == 0) { ch_wait<0>(R0[0], R0[1], R0[2], R0[3], R0[4]); mm(R0, bic); }
          else { ch_wait<0>(R1[0], R1[1], R1[2], R1[3], R1[4]); mm(R1, bic); }
        }
      });
      // lane (m, hi) holds column 32 u + 8 q + 4 hi + i in register 4 q + i: pair q = 2p (lower lane keeps) with q = 2p + 1 (upper lane)
      unsigned ho[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) ho[p] = dmt_pack_bf16(Ha[2 * p], Ha[2 * p + 1]);
      swap_lo(ho[0], ho[2]); swap_lo(ho[1], ho[3]);
      swap_lo(ho[4], ho[6]); swap_lo(ho[5], ho[7]);
      if (mvalid && ((DBG & 2) == 0 || ho[0] == 0x12345678u)) {
        bf16_t* dst = g.out + m * g.ld_out + u * 32 + 8 * hi;
        *reinterpret_cast<u32x4_t*>(dst) = u32x4_t{ho[0], ho[1], ho[2], ho[3]};
        *reinterpret_cast<u32x4_t*>(dst + 16) = u32x4_t{ho[4], ho[5], ho[6], ho[7]};
      }
    }
  }
}

template <typename F>
int proj_dispatch(int kin, int n, F&& f) {
  if (kin == 320 && n == 960) return f(PGeo<320, 960>{});
  dmt_set_error("dmt_proj: unsupported geometry kin=%d n=%d (built: 320/960)", kin, n);
  return DMT_ERR_UNSUPPORTED;
}

template <typename F>
int chain_dispatch(int kin, int nmid, int nout, F&& f) {
  if (kin == 320 && nmid == 1280 && nout == 320) return f(Geo<320, 1280, 320>{});
  if (kin == 80 && nmid == 320 && nout == 80) return f(Geo<80, 320, 80>{});
  dmt_set_error("dmt_chain: unsupported geometry kin=%d nmid=%d nout=%d (built: 320/1280/320, 80/320/80)", kin, nmid, nout);
  return DMT_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int dmt_chain_supported(int32_t kin, int32_t nmid, int32_t nout) {
  return (kin == 320 && nmid == 1280 && nout == 320) || (kin == 80 && nmid == 320 && nout == 80);
}

extern "C" int dmt_chain_image_bytes(int32_t kin, int32_t nmid, int32_t nout, int64_t* bytes) {
  DMT_CHECK_ARG(bytes != nullptr, "dmt_chain_image_bytes: null output");
  return chain_dispatch(kin, nmid, nout, [&](auto geo) {
    *bytes = decltype(geo)::IMAGE_BYTES;
    return DMT_OK;
  });
}

extern "C" int dmt_chain_image_build(int32_t kin, int32_t nmid, int32_t nout, const float* a1, int64_t a1_rs, int64_t a1_cs, const float* a2,
                                     int64_t a2_rs, int64_t a2_cs, const float* bias1, void* image, void* stream) {
  DMT_CHECK_ARG(a1 && a2 && image, "dmt_chain_image_build: null pointer");
  return chain_dispatch(kin, nmid, nout, [&](auto geo) {
    typedef decltype(geo) G;
    ImgArgs g;
    g.a1 = a1; g.a1_rs = a1_rs; g.a1_cs = a1_cs;
    g.a2 = a2; g.a2_rs = a2_rs; g.a2_cs = a2_cs;
    g.bias1 = bias1;
    g.img = (unsigned char*)image;
    g.kin = kin; g.nmid = nmid; g.nout = nout;
    g.a1_stride = G::A1_STRIDE; g.a1_bytes = G::A1_BYTES; g.a2_bytes = G::A2_BYTES; g.bias_off = G::BIAS_OFF; g.stage = G::STAGE;
    g.slots = G::IMAGE_BYTES / 16;
    long long nb = cdiv64(g.slots, 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(chain_image_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, g);
    DMT_CHECK_LAUNCH("dmt_chain_image_build");
    return DMT_OK;
  });
}

extern "C" int dmt_chain2(const dmt_chain_desc* d, void* stream) {
  DMT_CHECK_ARG(d != nullptr, "dmt_chain2: null descriptor");
  DMT_CHECK_ARG(d->mode == DMT_CHAIN_FFN_LN || d->mode == DMT_CHAIN_FFN_BWD, "dmt_chain2: bad mode %d", d->mode);
  DMT_CHECK_ARG(d->M > 0 && d->in && d->image, "dmt_chain2: bad argument");
  DMT_CHECK_ARG(d->ld_in % 8 == 0 && ((uintptr_t)d->in & 15) == 0, "dmt_chain2: input rows must be 16-byte aligned");
  DMT_CHECK_ARG(d->M < (1ll << 31) - 256, "dmt_chain2: too many rows");
  if (d->mode == DMT_CHAIN_FFN_LN) {
    DMT_CHECK_ARG(d->bias2 && d->gamma && d->beta && d->y_out, "dmt_chain2(ffn_ln): bias2 / gamma / beta / y_out are required");
    DMT_CHECK_ARG(d->kin == d->nout, "dmt_chain2(ffn_ln): the residual needs kin == nout");
  } else {
    DMT_CHECK_ARG(d->mask && d->s_out, "dmt_chain2(ffn_bwd): mask and s_out (dx) are required");
    DMT_CHECK_ARG(d->kin == d->nout, "dmt_chain2(ffn_bwd): the residual needs kin == nout");
  }
  DMT_CHECK_ARG(d->ld_out % 8 == 0 && (((uintptr_t)d->s_out | (uintptr_t)d->y_out) & 15) == 0, "dmt_chain2: output rows must be 16-byte aligned");
  DMT_CHECK_ARG(d->mid_out == nullptr || (d->ld_mid % 8 == 0 && ((uintptr_t)d->mid_out & 15) == 0), "dmt_chain2: mid rows must be 16-byte aligned");
  return chain_dispatch(d->kin, d->nmid, d->nout, [&](auto geo) { return launch_chain<decltype(geo)>(d, (hipStream_t)stream); });
}

extern "C" int dmt_proj_supported(int32_t kin, int32_t n) { return kin == 320 && n == 960; }

extern "C" int dmt_proj_image_bytes(int32_t kin, int32_t n, int64_t* bytes) {
  DMT_CHECK_ARG(bytes != nullptr, "dmt_proj_image_bytes: null output");
  return proj_dispatch(kin, n, [&](auto geo) {
    *bytes = decltype(geo)::IMAGE_BYTES;
    return DMT_OK;
  });
}

extern "C" int dmt_proj_image_build(int32_t kin, int32_t n, const float* w, int64_t w_rs, int64_t w_cs, const float* bias, void* image,
                                    void* stream) {
  DMT_CHECK_ARG(w && image, "dmt_proj_image_build: null pointer");
  return proj_dispatch(kin, n, [&](auto geo) {
    typedef decltype(geo) G;
    ImgArgs g;
    g.a1 = w; g.a1_rs = w_cs; g.a1_cs = w_rs;      // A1[j, k] = W[k, j]
    g.a2 = nullptr; g.a2_rs = 0; g.a2_cs = 0;
    g.bias1 = bias;
    g.img = (unsigned char*)image;
    g.kin = kin; g.nmid = n; g.nout = 0;
    g.a1_stride = G::A1_STRIDE; g.a1_bytes = G::A1_BYTES; g.a2_bytes = 0; g.bias_off = G::BIAS_OFF; g.stage = G::STAGE;
    g.slots = G::IMAGE_BYTES / 16;
    long long nb = cdiv64(g.slots, 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(chain_image_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, g);
    DMT_CHECK_LAUNCH("dmt_proj_image_build");
    return DMT_OK;
  });
}

extern "C" int dmt_proj(int32_t kin, int32_t n, int64_t M, const void* in, int64_t ld_in, const void* image, void* out, int64_t ld_out,
                        void* stream) {
  DMT_CHECK_ARG(M > 0 && in && image && out, "dmt_proj: bad argument");
  DMT_CHECK_ARG(ld_in % 8 == 0 && ((uintptr_t)in & 15) == 0, "dmt_proj: input rows must be 16-byte aligned");
  DMT_CHECK_ARG(ld_out % 8 == 0 && ((uintptr_t)out & 15) == 0, "dmt_proj: output rows must be 16-byte aligned");
  DMT_CHECK_ARG(M < (1ll << 31) - 256, "dmt_proj: too many rows");
  return proj_dispatch(kin, n, [&](auto geo) {
    typedef decltype(geo) G;
    ProjArgs a;
    a.M = M;
    a.in = (const bf16_t*)in; a.ld_in = ld_in;
    a.image = (const unsigned char*)image;
    a.out = (bf16_t*)out; a.ld_out = ld_out;
    a.tiles = (int)cdiv64(M, 128);
    int grid = a.tiles < 256 ? a.tiles : 256;
#ifdef DMT_TIMING_EXPERIMENTS   // (scripts/ ablations only: `make EXPERIMENTS=1`)
    if (const char* gq = getenv("DMT_PROJ_GRID")) { const int v = atoi(gq); if (v > 0 && v < grid) grid = v; }
    const char* dbg = getenv("DMT_PROJ_DEBUG");   // timing experiments (see DBG above)
    switch (dbg ? atoi(dbg) : 0) {
      case 1: hipLaunchKernelGGL((proj_kernel<G, 1>), dim3(grid), dim3(PJ_NT), 0, (hipStream_t)stream, a); break;
      case 2: hipLaunchKernelGGL((proj_kernel<G, 2>), dim3(grid), dim3(PJ_NT), 0, (hipStream_t)stream, a); break;
      case 4: hipLaunchKernelGGL((proj_kernel<G, 4>), dim3(grid), dim3(PJ_NT), 0, (hipStream_t)stream, a); break;
      case 5: hipLaunchKernelGGL((proj_kernel<G, 5>), dim3(grid), dim3(PJ_NT), 0, (hipStream_t)stream, a); break;
      case 7: hipLaunchKernelGGL((proj_kernel<G, 7>), dim3(grid), dim3(PJ_NT), 0, (hipStream_t)stream, a); break;
      case 8: hipLaunchKernelGGL((proj_kernel<G, 8>), dim3(grid), dim3(PJ_NT), 0, (hipStream_t)stream, a); break;
      case 79: hipLaunchKernelGGL((proj_kernel<G, 79>), dim3(grid), dim3(PJ_NT), 0, (hipStream_t)stream, a); break;
      case 64: hipLaunchKernelGGL((proj_kernel<G, 64>), dim3(grid), dim3(PJ_NT), 0, (hipStream_t)stream, a); break;
      case 47: hipLaunchKernelGGL((proj_kernel<G, 47>), dim3(grid), dim3(PJ_NT), 0, (hipStream_t)stream, a); break;
      case 15: hipLaunchKernelGGL((proj_kernel<G, 15>), dim3(grid), dim3(PJ_NT), 0, (hipStream_t)stream, a); break;
      default: hipLaunchKernelGGL((proj_kernel<G>), dim3(grid), dim3(PJ_NT), 0, (hipStream_t)stream, a);
    }
#else
    hipLaunchKernelGGL((proj_kernel<G>), dim3(grid), dim3(PJ_NT), 0, (hipStream_t)stream, a);
#endif
    DMT_CHECK_LAUNCH("dmt_proj");
    return DMT_OK;
  });
}

// ---- batched image rebuild: the caller keeps a device table of jobs (dmt_image_job_bytes() each), filled on the host by
//      dmt_chain_image_job / dmt_proj_image_job with the same arguments as the one-image builders
extern "C" int32_t dmt_image_job_bytes(void) { return (int32_t)sizeof(ImgArgs); }

extern "C" int dmt_chain_image_job(int32_t kin, int32_t nmid, int32_t nout, const float* a1, int64_t a1_rs, int64_t a1_cs, const float* a2,
                                   int64_t a2_rs, int64_t a2_cs, const float* bias1, void* image, void* job_out) {
  DMT_CHECK_ARG(a1 && a2 && image && job_out, "dmt_chain_image_job: null pointer");
  return chain_dispatch(kin, nmid, nout, [&](auto geo) {
    typedef decltype(geo) G;
    ImgArgs g;
    g.a1 = a1; g.a1_rs = a1_rs; g.a1_cs = a1_cs;
    g.a2 = a2; g.a2_rs = a2_rs; g.a2_cs = a2_cs;
    g.bias1 = bias1;
    g.img = (unsigned char*)image;
    g.kin = kin; g.nmid = nmid; g.nout = nout;
    g.a1_stride = G::A1_STRIDE; g.a1_bytes = G::A1_BYTES; g.a2_bytes = G::A2_BYTES; g.bias_off = G::BIAS_OFF; g.stage = G::STAGE;
    g.slots = G::IMAGE_BYTES / 16;
    *reinterpret_cast<ImgArgs*>(job_out) = g;
    return DMT_OK;
  });
}

extern "C" int dmt_proj_image_job(int32_t kin, int32_t n, const float* w, int64_t w_rs, int64_t w_cs, const float* bias, void* image,
                                  void* job_out) {
  DMT_CHECK_ARG(w && image && job_out, "dmt_proj_image_job: null pointer");
  return proj_dispatch(kin, n, [&](auto geo) {
    typedef decltype(geo) G;
    ImgArgs g;
    g.a1 = w; g.a1_rs = w_cs; g.a1_cs = w_rs;
    g.a2 = nullptr; g.a2_rs = 0; g.a2_cs = 0;
    g.bias1 = bias;
    g.img = (unsigned char*)image;
    g.kin = kin; g.nmid = n; g.nout = 0;
    g.a1_stride = G::A1_STRIDE; g.a1_bytes = G::A1_BYTES; g.a2_bytes = 0; g.bias_off = G::BIAS_OFF; g.stage = G::STAGE;
    g.slots = G::IMAGE_BYTES / 16;
    *reinterpret_cast<ImgArgs*>(job_out) = g;
    return DMT_OK;
  });
}

extern "C" int dmt_image_build_batched(int32_t n_jobs, const void* jobs_dev, void* stream) {
  DMT_CHECK_ARG(n_jobs > 0 && jobs_dev, "dmt_image_build_batched: bad argument");
  DMT_CHECK_ARG(n_jobs <= 65535, "dmt_image_build_batched: too many jobs");
  hipLaunchKernelGGL(chain_image_batched_kernel, dim3(256, (unsigned)n_jobs), dim3(256), 0, (hipStream_t)stream, (const ImgArgs*)jobs_dev);
  DMT_CHECK_LAUNCH("dmt_image_build_batched");
  return DMT_OK;
}

