// Long-sequence multi-head attention core (64 < T <= 256: BASELINE "long-seq variant", clk / ord histories of 200), forward and
// backward, flash style: no [B,H,T,T] tensor reaches HBM.  Same arithmetic and masking as the short kernels of dmt_attn.hip
// (TransformerModel_util.py:11-56, 80-108): keys past k_len <- -2^32+1 before the softmax, rows of padded queries <- -2^32+1 after
// it, dropout on the weights after the query mask, no output projection, residual added in the epilogue.
//
// One workgroup of 4 wavefronts per (example, head); its LDS (< 80 KB at T <= 208) lets two workgroups share a CU, so one stages its
// operands while the other computes (two wavefronts per SIMD, 256 VGPRs each).
//   forward   K [Tk][dh] and V [Tk][dh] are staged once in LDS.  Wave w owns query tile w (32 queries): S^T = K Q^T for ALL keys stays
//             in registers (NTK x 16 fp32 per lane: lane = query column, registers = keys), so the softmax is the exact two-pass
//             form of the reference; the packed weights are directly the B operand of O^T = V^T P^T (V^T comes out of LDS through
//             ds_read_b64_tr_b16).  Per (query tile, key tile): 5 + 6 MFMA 32x32x16 at dh = 80.
//   backward  phase 1 (wave = query tile; K, V in LDS): S^T strip -> row max / sum -> P (fp32, registers); dP^T = V dO^T tile by
//             tile, twice: once for D = sum_k P dP, once for dS = P (dP - D) / sqrt(dh) -> dQ^T = K^T dS^T.  (m, 1/sum, D) of every
//             query go to LDS.
//             phase 2 (wave = key tile; Q, dO re-staged over K, V): S = Q K^T and dP = dO V^T with lane = key column, registers =
//             queries; P and dS follow from the stored row statistics, and, packed, are directly the B operands of
//             dV^T += dO^T P and dK^T += Q^T dS: no transposition through LDS, no cross-wave reduction, no atomics.
//             Per (query tile, key tile): 21 + 22 MFMA.
// The score / weight MFMAs run in bf16, or -- mma_dtype = DMT_FP8 -- in OCP e4m3 (v_mfma_f32_32x32x16_fp8_fp8) with power-of-two
// scales per operand tile (see the F8 helpers below).
#include "dmt_common.h"

namespace {

constexpr float PADDING_NUM = -4294967295.0f;   // -2**32 + 1 (TransformerModel_util.py:81)
constexpr float LOG2E = 1.44269504088896340736f;
constexpr int LNW = 4;                           // wavefronts per workgroup

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_p4;

struct LongArgs {
  int B, H, Tq, Tk;
  const bf16_t* Q; long long q_bs, q_rs;
  const bf16_t* K; long long k_bs, k_rs;
  const bf16_t* V; long long v_bs, v_rs;
  const int* q_lens;
  const int* k_lens;
  const bf16_t* resid; long long r_bs, r_rs;
  bf16_t* out; long long o_bs, o_rs;
  const bf16_t* dout; long long do_bs, do_rs;
  bf16_t* dQ; long long dq_bs, dq_rs;
  bf16_t* dK; long long dk_bs, dk_rs;
  bf16_t* dV; long long dv_bs, dv_rs;
  unsigned drop_seed, drop_thr;
  float drop_inv;
  int drop_on;
};

template <int DH> struct LongCfg {
  static constexpr int CH = DH / 8;                 // 16-byte chunks per row
  static constexpr int RS = DH + 8;                 // row stride of a natural-order tile (fragment reads: rows 4 banks apart mod 32)
  static constexpr int rsv() { int r = (DH + 7) / 8 * 8; while (r % 128 != 32 && r % 128 != 96) r += 8; return r; }
  static constexpr int RSV = rsv();                 // row stride of the V tile (transposing reads of 4 consecutive rows: disjoint banks)
  static constexpr int NK = (DH + 15) / 16;
  static constexpr int NDT = (DH + 31) / 32;
  static constexpr int RO = 36;                     // fp32 staging row stride of one [32 q][32 dims] forward output sub-tile
  static constexpr int SLD = 40;                    // bf16 staging row stride of one [32 rows][32 dims] gradient sub-tile
};

__device__ __forceinline__ bf16x8_t zero8() {
  union { bf16x8_t v; uint4 q; } u;
  u.q = make_uint4(0u, 0u, 0u, 0u);
  return u.v;
}

// 16-byte MFMA fragment of one row: elements j0 .. j0+7 (zero past dh)
template <int DH>
__device__ __forceinline__ bf16x8_t row_frag(const bf16_t* __restrict__ row, int j0) {
  return (j0 + 8 <= DH) ? *reinterpret_cast<const bf16x8_t*>(row + j0) : zero8();
}

// Workgroup-cooperative copy of rows [0, n_rows) of two [.., dh] operands into LDS tiles of `pad_a` / `pad_b` rows (zero filled).
// All the 16-byte requests of a thread (MAXA, MAXB = compile-time bounds of the tile heights) are in flight together: rows past
// n_rows re-read the last row (branch-free loads; replaced by zeros on the way into LDS).  (The rolled load -> wait -> store loop
// this replaces paid one memory latency per 16 bytes and thread: 18 in a row at T = 200, most of the workgroup's lifetime.)
template <int DH, int MAXA, int MAXB>
__device__ __forceinline__ void stage_rows2(bf16_t* __restrict__ dst_a, int ld_a, const bf16_t* __restrict__ src_a, long long rs_a, int n_a,
                                            int pad_a, bf16_t* __restrict__ dst_b, int ld_b, const bf16_t* __restrict__ src_b, long long rs_b,
                                            int n_b, int pad_b, int tid) {
  constexpr int CH = DH / 8, NT = LNW * 64;
  constexpr int ITA = (MAXA * CH + NT - 1) / NT, ITB = (MAXB * CH + NT - 1) / NT;
  uint4 va[ITA], vb[ITB];
#pragma unroll
  for (int i = 0; i < ITA; ++i) {
    const int c = tid + i * NT, row = c / CH, ch = c - row * CH;
    const int rr = row < n_a ? row : n_a - 1;
    va[i] = *reinterpret_cast<const uint4*>(src_a + (long long)rr * rs_a + ch * 8);
  }
#pragma unroll
  for (int i = 0; i < ITB; ++i) {
    const int c = tid + i * NT, row = c / CH, ch = c - row * CH;
    const int rr = row < n_b ? row : n_b - 1;
    vb[i] = *reinterpret_cast<const uint4*>(src_b + (long long)rr * rs_b + ch * 8);
  }
#pragma unroll
  for (int i = 0; i < ITA; ++i) {
    const int c = tid + i * NT, row = c / CH, ch = c - row * CH;
    if (row < pad_a) *reinterpret_cast<uint4*>(dst_a + row * ld_a + ch * 8) = row < n_a ? va[i] : make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int i = 0; i < ITB; ++i) {
    const int c = tid + i * NT, row = c / CH, ch = c - row * CH;
    if (row < pad_b) *reinterpret_cast<uint4*>(dst_b + row * ld_b + ch * 8) = row < n_b ? vb[i] : make_uint4(0u, 0u, 0u, 0u);
  }
}

// A operand X^T (rows = dims dt*32 .., k = 16 rows of X starting at `row0`) through transposing LDS reads.
//   slots = true : k-slot i of lane-half h <-> row row0 + (i&3) + 8(i>>2) + 4h   (the order of a packed accumulator)
template <int LD>
__device__ __forceinline__ bf16x8_t frag_T_slots(const bf16_t* __restrict__ X, int row0, int dt, int lane) {
  const int half = lane >> 5, i16 = lane & 15, jgrp = (lane >> 4) & 1;
  const bf16_t* vp = X + (row0 + 4 * half + (i16 >> 2)) * LD + dt * 32 + 16 * jgrp + 4 * (i16 & 3);
  const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p4)(vp));
  const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p4)(vp + 8 * LD));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// accumulator registers 8u .. 8u+7 (u = 0, 1) -> one packed bf16 B operand
__device__ __forceinline__ bf16x8_t pack_half(const f32x16_t& x, int u) {
  union { bf16x8_t v; unsigned w[4]; } f;
#pragma unroll
  for (int i = 0; i < 4; ++i) f.w[i] = dmt_pack_bf16(x[8 * u + 2 * i], x[8 * u + 2 * i + 1]);
  return f.v;
}

// element i (compile-time constant after unrolling) of a packed operand, as fp32
__device__ __forceinline__ float unpack_at(const bf16x8_t& v, int i) {
  union { bf16x8_t v; unsigned w[4]; } f;
  f.v = v;
  const unsigned w = f.w[i >> 1];
  return __uint_as_float((i & 1) ? (w & 0xffff0000u) : (w << 16));
}

__device__ __forceinline__ f32x16_t zero16() {
  f32x16_t z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

__device__ __forceinline__ f32x16_t mma(const bf16x8_t& a, const bf16x8_t& b, const f32x16_t& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// one [32 rows][32 dims] accumulator half (lane = output row, registers = dims 8g + 4 half + {0..3}) -> bf16, 8-byte stores: the two
// lane halves of a row write 16 contiguous bytes per instruction, four instructions complete a 64-byte segment.  (Staging the tile
// through LDS for 16-byte stores costs the LDS that lets a second workgroup onto the CU.)
template <int DH>
__device__ __forceinline__ void store_direct(const f32x16_t& o, bf16_t* __restrict__ rowp, bool row_ok, int dt, int half) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int j = dt * 32 + 8 * g + 4 * half;
    if (row_ok && j + 4 <= DH) {
      uint2 ov;
      ov.x = dmt_pack_bf16(o[4 * g + 0], o[4 * g + 1]);
      ov.y = dmt_pack_bf16(o[4 * g + 2], o[4 * g + 3]);
      *reinterpret_cast<uint2*>(rowp + j) = ov;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ forward
template <int DH, int NTK, bool DROP>
__global__ __launch_bounds__(LNW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_long_fwd_kernel(const LongArgs a) {
  typedef LongCfg<DH> CF;
  constexpr int NK = CF::NK, NDT = CF::NDT, RS = CF::RS, RSV = CF::RSV;
  // K rows past Tk are never staged: their scores are replaced (select) before use, so whatever finite data follows the K rows
  // serves.  NTK = 7 is the T <= 208 build: 208 K rows + 224 zero-filled V rows = 79.6 KB, two workgroups per CU.
  constexpr int KROWS = (NTK == 7) ? 208 : NTK * 32;
  __shared__ __attribute__((aligned(16))) bf16_t lds[KROWS * RS + NTK * 32 * RSV + 64];
  bf16_t* Kl = lds;
  bf16_t* Vl = lds + KROWS * RS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int Tq = a.Tq, Tk = a.Tk;
  const int half = lane >> 5, l31 = lane & 31;
  const bf16_t* Qg = a.Q + (long long)b * a.q_bs + h * DH;
  const bf16_t* Rg = a.resid ? a.resid + (long long)b * a.r_bs + h * DH : nullptr;
  bf16_t* Og = a.out + (long long)b * a.o_bs + h * DH;
  stage_rows2<DH, KROWS, NTK * 32>(Kl, RS, a.K + (long long)b * a.k_bs + h * DH, a.k_rs, Tk, Tk < KROWS ? (Tk + 7) & ~7 : KROWS,
                                   Vl, RSV, a.V + (long long)b * a.v_bs + h * DH, a.v_rs, Tk, NTK * 32, tid);
  int klen = a.k_lens ? a.k_lens[b] : Tk;
  klen = klen < 0 ? 0 : (klen > Tk ? Tk : klen);
  const int qlen = a.q_lens ? a.q_lens[b] : Tq;
  const float kscale = LOG2E / sqrtf((float)DH);
  const int kl0 = klen - 4 * half, tk0 = Tk - 4 * half;   // slot constant c: key = c + 4 half
  const int nqt = (Tq + 31) >> 5;
  const int kfull = klen >> 5;                            // key tiles [0, kfull) hold valid keys only
  bool synced = false;
  for (int qt = wave; qt < nqt; qt += LNW) {
    const int q = qt * 32 + l31;
    // (opaque copies: the ~250 lane masks "c >= kl" / "c >= tk" are otherwise hoisted out of this loop and spilled as SGPR pairs)
    int kl = kl0, tk = tk0;
    asm volatile("" : "+v"(kl), "+v"(tk));
    bf16x8_t bQ[NK];
#pragma unroll
    for (int s2 = 0; s2 < NK; ++s2) bQ[s2] = q < Tq ? row_frag<DH>(Qg + (long long)q * a.q_rs, s2 * 16 + 8 * half) : zero8();
    // the residual pieces of the epilogue (this lane's row q, dims dt*32 + 8g + 4 half + {0..3}), requested now
    uint2 rres[NDT][4];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = dt * 32 + 8 * g + 4 * half;
        rres[dt][g] = (Rg && q < Tq && j + 4 <= DH) ? *reinterpret_cast<const uint2*>(Rg + (long long)q * a.r_rs + j) : make_uint2(0u, 0u);
      }
    if (!synced) { __syncthreads(); synced = true; }     // (the first tile's Q / residual requests are in flight across the barrier)
    // ---- S^T = K Q^T, all key tiles
    f32x16_t acc[NTK];
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt) acc[kt] = zero16();
#pragma unroll
    for (int s2 = 0; s2 < NK; ++s2)
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt)
        acc[kt] = mma(row_frag<DH>(Kl + (kt * 32 + l31) * RS, s2 * 16 + 8 * half), bQ[s2], acc[kt]);
    // ---- masked softmax over the keys of query column q.  The two key masks apply only from the first key tile that holds a key
    // >= k_len (k_len is one number per workgroup: a scalar branch per tile), the query mask only in a query tile that holds a
    // padded query; dropout is a compile-time variant.  Same arithmetic, in the same order, for every element either way.
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt) {
      if (kt < kfull) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float x = acc[kt][r] * kscale;
          acc[kt][r] = x;
          m = fmaxf(m, x);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
          float x = acc[kt][r] * kscale;
          x = (c >= kl) ? PADDING_NUM * LOG2E : x;
          x = (c >= tk) ? -3.0e38f : x;
          acc[kt][r] = x;
          m = fmaxf(m, x);
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(acc[kt][r] - m);
        acc[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv_sum = __builtin_amdgcn_rcpf(sum);
    const unsigned dbase = (unsigned)((b * a.H + h) * Tq + q) * (unsigned)Tk + 4u * half;
    bf16x8_t pB[2 * NTK];
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kt][r] *= inv_sum;
    if (qt * 32 + 32 > qlen) {             // (scalar) rows of padded queries: the constant on every key that exists
      const bool qpad = (q >= qlen);
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
          if (qpad) acc[kt][r] = (c < tk) ? PADDING_NUM : 0.f;
        }
    }
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt) {
      if constexpr (DROP) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
          acc[kt][r] = dmt_drop_keep(a.drop_seed, dbase + (unsigned)c, a.drop_thr) ? acc[kt][r] * a.drop_inv : 0.f;
        }
      }
      pB[2 * kt] = pack_half(acc[kt], 0);
      pB[2 * kt + 1] = pack_half(acc[kt], 1);
    }
    // ---- O^T = V^T P^T per 32 output dims, + residual, one rounding
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      f32x16_t o = zero16();
#pragma unroll
      for (int u = 0; u < 2 * NTK; ++u) o = mma(frag_T_slots<RSV>(Vl, 16 * u, dt, lane), pB[u], o);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned w0 = rres[dt][g].x, w1 = rres[dt][g].y;
        o[4 * g + 0] += __uint_as_float(w0 << 16); o[4 * g + 1] += __uint_as_float(w0 & 0xffff0000u);
        o[4 * g + 2] += __uint_as_float(w1 << 16); o[4 * g + 3] += __uint_as_float(w1 & 0xffff0000u);
      }
      store_direct<DH>(o, Og + (long long)q * a.o_rs, q < Tq, dt, half);
    }
  }
  if (!synced) __syncthreads();
}

// ----------------------------------------------------------------------------------------------- forward, fp8 (OCP e4m3) MFMA
// BASELINE configs[4] "fp8 MFMA attention path": S^T = K Q^T and O^T = V^T P^T run on v_mfma_f32_32x32x16_fp8_fp8 (fp32 accumulate).
//   * Q, K, V are converted bf16 -> e4m3 as they are staged (saturating at +-448: they are projections of LayerNorm outputs, O(1));
//     K lives in LDS as [key][dh] bytes, V transposed as [dim][key] bytes, so both A operands are plain 4/8-byte LDS reads.
//   * the weights P in [0, 1/keep] are scaled by 2^8 before the conversion (a uniform 1/200 would otherwise be an e4m3 subnormal with two
//     mantissa bits); the exact power of two is divided out of O.  Rows of padded queries (weights = -2^32+1, far outside e4m3) carry
//     the constant in that output scale instead: P8 = 1.
//   * softmax, masks, dropout and the residual add are the fp32 code of the bf16 kernel.  The backward pass stays bf16
//     (attn_long_bwd_kernel): it differentiates the bf16 function, the usual pairing for fp8 forward passes.
// Same MFMA shape and rate as bf16 (the 2x-rate MX-scaled 32x32x64 instruction pays only once this kernel is MFMA-bound: at
// ~12 % MFMA utilisation its cost is softmax VALU, LDS and latency); what fp8 buys here is half the LDS bytes per workgroup.
typedef long f8x8_t;      // 8 e4m3 values (2 VGPRs)

__device__ __forceinline__ unsigned cvt4_f8(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}
__device__ __forceinline__ float sat8(float x) { return fminf(fmaxf(x, -448.f), 448.f); }
// 8 bf16 (one 16-byte piece of a row) -> 8 e4m3
__device__ __forceinline__ uint2 bf16x8_to_f8(const uint4& u) {
  float f[8];
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xFFFF0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xFFFF0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xFFFF0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xFFFF0000u);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = sat8(f[i]);
  return make_uint2(cvt4_f8(f[0], f[1], f[2], f[3]), cvt4_f8(f[4], f[5], f[6], f[7]));
}
__device__ __forceinline__ f8x8_t as_f8x8(const uint2& u) {
  union { uint2 u; f8x8_t l; } x;
  x.u = u;
  return x.l;
}
__device__ __forceinline__ f32x16_t mma8(f8x8_t a, f8x8_t b, const f32x16_t& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
}

template <int DH, int NTK, bool DROP>
__global__ __launch_bounds__(LNW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_long_fwd_f8_kernel(const LongArgs a) {
  typedef LongCfg<DH> CF;
  constexpr int NK = CF::NK, NDT = CF::NDT, CH = DH / 8;
  constexpr int RS8 = DH + 8;                  // bytes per K row
  constexpr int VS8 = NTK * 32 + 8;            // bytes per V^T row (one head dim, all keys)
  constexpr int VDIMS = NDT * 32;
  __shared__ __attribute__((aligned(16))) unsigned char Kl[NTK * 32 * RS8];
  __shared__ __attribute__((aligned(16))) unsigned char Vt[VDIMS * VS8];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int Tq = a.Tq, Tk = a.Tk;
  const int half = lane >> 5, l31 = lane & 31;
  const bf16_t* Qg = a.Q + (long long)b * a.q_bs + h * DH;
  const bf16_t* Kg = a.K + (long long)b * a.k_bs + h * DH;
  const bf16_t* Vg = a.V + (long long)b * a.v_bs + h * DH;
  const bf16_t* Rg = a.resid ? a.resid + (long long)b * a.r_bs + h * DH : nullptr;
  bf16_t* Og = a.out + (long long)b * a.o_bs + h * DH;
  // ---- stage K as e4m3 rows, V as e4m3 columns (keys past Tk: zeros -- 0 x anything finite)
  for (int i = tid; i < VDIMS * VS8 / 4; i += LNW * 64) reinterpret_cast<unsigned*>(Vt)[i] = 0u;
  __syncthreads();
  {
    // (all of a thread's 16-byte requests in flight together, see stage_rows2)
    constexpr int IT = (NTK * 32 * CH + LNW * 64 - 1) / (LNW * 64);
    uint4 kraw[IT], vraw[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = tid + i * LNW * 64, row = c / CH, ch = c - row * CH;
      const int rr = row < Tk ? row : Tk - 1;
      kraw[i] = *reinterpret_cast<const uint4*>(Kg + (long long)rr * a.k_rs + ch * 8);
      vraw[i] = *reinterpret_cast<const uint4*>(Vg + (long long)rr * a.v_rs + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = tid + i * LNW * 64, row = c / CH, ch = c - row * CH;
      if (row < NTK * 32) {
        uint2 k8 = make_uint2(0u, 0u);
        if (row < Tk) {
          k8 = bf16x8_to_f8(kraw[i]);
          const uint2 v8 = bf16x8_to_f8(vraw[i]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            Vt[(ch * 8 + e) * VS8 + row] = (unsigned char)(v8.x >> (8 * e));
            Vt[(ch * 8 + 4 + e) * VS8 + row] = (unsigned char)(v8.y >> (8 * e));
          }
        }
        *reinterpret_cast<uint2*>(Kl + row * RS8 + ch * 8) = k8;
      }
    }
  }
  int klen = a.k_lens ? a.k_lens[b] : Tk;
  klen = klen < 0 ? 0 : (klen > Tk ? Tk : klen);
  const int qlen = a.q_lens ? a.q_lens[b] : Tq;
  const float kscale = LOG2E / sqrtf((float)DH);
  const int kl0 = klen - 4 * half, tk0 = Tk - 4 * half;
  const int nqt = (Tq + 31) >> 5;
  const int kfull = klen >> 5;
  bool synced = false;
  for (int qt = wave; qt < nqt; qt += LNW) {
    const int q = qt * 32 + l31;
    int kl = kl0, tk = tk0;
    asm volatile("" : "+v"(kl), "+v"(tk));
    f8x8_t bQ[NK];
#pragma unroll
    for (int s2 = 0; s2 < NK; ++s2) {
      const int j0 = s2 * 16 + 8 * half;
      bQ[s2] = (q < Tq && j0 + 8 <= DH) ? as_f8x8(bf16x8_to_f8(*reinterpret_cast<const uint4*>(Qg + (long long)q * a.q_rs + j0))) : 0L;
    }
    if (!synced) { __syncthreads(); synced = true; }
    f32x16_t acc[NTK];
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt) acc[kt] = zero16();
#pragma unroll
    for (int s2 = 0; s2 < NK; ++s2) {
      const int j0 = s2 * 16 + 8 * half;
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt) {
        const f8x8_t kf = (j0 + 8 <= DH) ? as_f8x8(*reinterpret_cast<const uint2*>(Kl + (kt * 32 + l31) * RS8 + j0)) : 0L;
        acc[kt] = mma8(kf, bQ[s2], acc[kt]);
      }
    }
    // (masks per tile / query tile, dropout at compile time: see attn_long_fwd_kernel)
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt) {
      if (kt < kfull) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float x = acc[kt][r] * kscale;
          acc[kt][r] = x;
          m = fmaxf(m, x);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
          float x = acc[kt][r] * kscale;
          x = (c >= kl) ? PADDING_NUM * LOG2E : x;
          x = (c >= tk) ? -3.0e38f : x;
          acc[kt][r] = x;
          m = fmaxf(m, x);
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(acc[kt][r] - m);
        acc[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    // the residual pieces of the epilogue (this lane's row q, dims dt*32 + 8g + 4 half + {0..3}), requested now: the normalisation,
    // dropout and packing below cover their latency, and the Q fragments are dead (168 registers: three workgroups per CU)
    uint2 rres[NDT][4];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = dt * 32 + 8 * g + 4 * half;
        rres[dt][g] = (Rg && q < Tq && j + 4 <= DH) ? *reinterpret_cast<const uint2*>(Rg + (long long)q * a.r_rs + j) : make_uint2(0u, 0u);
      }
    const bool qpad = (q >= qlen);
    const float pscale = __builtin_amdgcn_rcpf(sum) * 256.f;          // P * 2^8 -> e4m3
    const float oscale = qpad ? PADDING_NUM : (1.f / 256.f);
    const unsigned dbase = (unsigned)((b * a.H + h) * Tq + q) * (unsigned)Tk + 4u * half;
    f8x8_t pB[2 * NTK];
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kt][r] *= pscale;
    if (qt * 32 + 32 > qlen) {
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
          if (qpad) acc[kt][r] = (c < tk) ? 1.f : 0.f;
        }
    }
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt) {
      if constexpr (DROP) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
          acc[kt][r] = dmt_drop_keep(a.drop_seed, dbase + (unsigned)c, a.drop_thr) ? acc[kt][r] * a.drop_inv : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
        pB[2 * kt + u] = as_f8x8(make_uint2(cvt4_f8(acc[kt][8 * u + 0], acc[kt][8 * u + 1], acc[kt][8 * u + 2], acc[kt][8 * u + 3]),
                                            cvt4_f8(acc[kt][8 * u + 4], acc[kt][8 * u + 5], acc[kt][8 * u + 6], acc[kt][8 * u + 7])));
    }
    // ---- O^T = V^T P^T: A = V^T rows (dims) in accumulator-slot key order: keys 16u + 4 half + {0..3} and 8 further
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      f32x16_t o = zero16();
      const unsigned char* vrow = Vt + (dt * 32 + l31) * VS8 + 4 * half;
#pragma unroll
      for (int u = 0; u < 2 * NTK; ++u) {
        const uint2 av = make_uint2(*reinterpret_cast<const unsigned*>(vrow + 16 * u), *reinterpret_cast<const unsigned*>(vrow + 16 * u + 8));
        o = mma8(as_f8x8(av), pB[u], o);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned w0 = rres[dt][g].x, w1 = rres[dt][g].y;
        o[4 * g + 0] = o[4 * g + 0] * oscale + __uint_as_float(w0 << 16); o[4 * g + 1] = o[4 * g + 1] * oscale + __uint_as_float(w0 & 0xffff0000u);
        o[4 * g + 2] = o[4 * g + 2] * oscale + __uint_as_float(w1 << 16); o[4 * g + 3] = o[4 * g + 3] * oscale + __uint_as_float(w1 & 0xffff0000u);
      }
      store_direct<DH>(o, Og + (long long)q * a.o_rs, q < Tq, dt, half);
    }
  }
  if (!synced) __syncthreads();
}

// ----------------------------------------------------------------------------------------------------------- backward
template <int DH, int NTK, bool DROP>
__global__ __launch_bounds__(LNW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_long_bwd_kernel(const LongArgs a) {
  typedef LongCfg<DH> CF;
  constexpr int NK = CF::NK, NDT = CF::NDT, RS = CF::RS;
  // Two operand tiles of TR rows (phase 1: K, V; phase 2: Q, dO).  The T <= 208 build keeps 208 rows each (78.7 KB in all: two
  // workgroups per CU); fragment reads of rows 208..223 then fall into the next tile (finite data) resp. the zeroed tail -- their
  // products are discarded by selects or multiplied by exact zeros.
  constexpr int TR = (NTK == 7) ? 208 : NTK * 32;
  constexpr int TAIL = (NTK * 32 - TR) * RS + 64;
  __shared__ __attribute__((aligned(16))) float s_m[NTK * 32], s_inv[NTK * 32], s_D[NTK * 32];
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * TR * RS + TAIL];
  bf16_t* XA = lds;                  // phase 1: K     phase 2: Q
  bf16_t* XB = lds + TR * RS;        // phase 1: V     phase 2: dO
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int Tq = a.Tq, Tk = a.Tk;
  const int half = lane >> 5, l31 = lane & 31;
  const bf16_t* Qg = a.Q + (long long)b * a.q_bs + h * DH;
  const bf16_t* Kg = a.K + (long long)b * a.k_bs + h * DH;
  const bf16_t* Vg = a.V + (long long)b * a.v_bs + h * DH;
  const bf16_t* dOg = a.dout + (long long)b * a.do_bs + h * DH;
  const int nqt = (Tq + 31) >> 5, nkt = (Tk + 31) >> 5;
  stage_rows2<DH, TR, TR>(XA, RS, Kg, a.k_rs, Tk, TR, XB, RS, Vg, a.v_rs, Tk, TR, tid);
  for (int i = tid; i < TAIL / 2; i += LNW * 64) reinterpret_cast<unsigned*>(lds + 2 * TR * RS)[i] = 0u;
  int klen = a.k_lens ? a.k_lens[b] : Tk;
  klen = klen < 0 ? 0 : (klen > Tk ? Tk : klen);
  const int qlen = a.q_lens ? a.q_lens[b] : Tq;
  const float kscale = LOG2E / sqrtf((float)DH), inv_sc = 1.0f / sqrtf((float)DH);
  const int kfull = klen >> 5;                    // key tiles [0, kfull) hold valid keys only

  // ================= phase 1: wave = query tile.  dQ and the row statistics.
  {
    const int kl0 = klen - 4 * half, tk0 = Tk - 4 * half;
    bf16_t* dQg = a.dQ + (long long)b * a.dq_bs + h * DH;
    bool synced = false;
    for (int qt = wave; qt < nqt; qt += LNW) {
      const int q = qt * 32 + l31;
      int kl = kl0, tk = tk0;
      asm volatile("" : "+v"(kl), "+v"(tk));
      bf16x8_t bQ[NK], bD[NK];
#pragma unroll
      for (int s2 = 0; s2 < NK; ++s2) {
        bQ[s2] = q < Tq ? row_frag<DH>(Qg + (long long)q * a.q_rs, s2 * 16 + 8 * half) : zero8();
        bD[s2] = q < Tq ? row_frag<DH>(dOg + (long long)q * a.do_rs, s2 * 16 + 8 * half) : zero8();
      }
      if (!synced) { __syncthreads(); synced = true; }
      f32x16_t acc[NTK];
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt) acc[kt] = zero16();
#pragma unroll
      for (int s2 = 0; s2 < NK; ++s2)
#pragma unroll
        for (int kt = 0; kt < NTK; ++kt)
          acc[kt] = mma(row_frag<DH>(XA + (kt * 32 + l31) * RS, s2 * 16 + 8 * half), bQ[s2], acc[kt]);
      // (key masks from the first tile that holds a key >= k_len on, query mask per query tile, dropout at compile time: see
      // attn_long_fwd_kernel)
      float m = -3.0e38f;
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt) {
        if (kt < kfull) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float x = acc[kt][r] * kscale;
            acc[kt][r] = x;
            m = fmaxf(m, x);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
            float x = acc[kt][r] * kscale;
            x = (c >= kl) ? PADDING_NUM * LOG2E : x;
            x = (c >= tk) ? -3.0e38f : x;
            acc[kt][r] = x;
            m = fmaxf(m, x);
          }
        }
      }
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(acc[kt][r] - m);
          acc[kt][r] = e;
          sum += e;
        }
      sum += __shfl_xor(sum, 32, 64);
      const float inv_sum = __builtin_amdgcn_rcpf(sum);
      // the weights wait for dP as packed bf16 (half the registers of the fp32 strip; they are rounded to bf16 for the MFMAs of the
      // forward pass anyway)
      bf16x8_t Pp[2 * NTK];
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kt][r] *= inv_sum;
        Pp[2 * kt] = pack_half(acc[kt], 0);
        Pp[2 * kt + 1] = pack_half(acc[kt], 1);
      }
      // ---- D = sum_k P dP  (dP^T = V dO^T, gradient w.r.t. the pre-dropout weights); the keep bits are remembered for pass 3
      const unsigned dbase = (unsigned)((b * a.H + h) * Tq + q) * (unsigned)Tk + 4u * half;
      unsigned keepb[DROP ? (NTK + 1) / 2 : 1];
      float dot = 0.f;
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt) {
        f32x16_t dp = zero16();
#pragma unroll
        for (int s2 = 0; s2 < NK; ++s2) dp = mma(row_frag<DH>(XB + (kt * 32 + l31) * RS, s2 * 16 + 8 * half), bD[s2], dp);
        unsigned bits = 0xFFFFu;
        if constexpr (DROP) {
          bits = 0u;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
            bits |= (dmt_drop_keep(a.drop_seed, dbase + (unsigned)c, a.drop_thr) ? 1u : 0u) << r;
          }
          if (kt & 1) keepb[kt >> 1] |= bits << 16;
          else keepb[kt >> 1] = bits;
        }
        const bool edge = !(kt < kfull);                                     // (scalar) this tile holds keys past k_len
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
          float g = dp[r];
          if constexpr (DROP) g = ((bits >> r) & 1u) ? g * a.drop_inv : 0.f;
          if (edge) g = (c < tk) ? g : 0.f;                                  // (rows past Tk of the V tile may be another tile's data)
          dot += unpack_at(Pp[2 * kt + (r >> 3)], r & 7) * g;
        }
      }
      dot += __shfl_xor(dot, 32, 64);
      if (half == 0) { s_m[q] = m; s_inv[q] = inv_sum; s_D[q] = dot; }
      // ---- dS = P (dP - D) / sqrt(dh)  -> packed B operands of dQ^T = K^T dS^T
      const bool qpad_tile = qt * 32 + 32 > qlen;                            // (scalar) the tile holds a padded query
      const bool qpad = (q >= qlen);
      bf16x8_t dsB[2 * NTK];
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt) {
        f32x16_t dp = zero16();
#pragma unroll
        for (int s2 = 0; s2 < NK; ++s2) dp = mma(row_frag<DH>(XB + (kt * 32 + l31) * RS, s2 * 16 + 8 * half), bD[s2], dp);
        unsigned bits = 0xFFFFu;
        if constexpr (DROP) bits = (keepb[kt >> 1] >> (16 * (kt & 1))) & 0xFFFFu;
        const bool edge = !(kt < kfull);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
          float g = dp[r];
          if constexpr (DROP) g = ((bits >> r) & 1u) ? g * a.drop_inv : 0.f;
          float ds = unpack_at(Pp[2 * kt + (r >> 3)], r & 7) * (g - dot) * inv_sc;
          if (edge) ds = (c < kl) ? ds : 0.f;                                // no gradient into masked keys
          dp[r] = ds;
        }
        if (qpad_tile) {
#pragma unroll
          for (int r = 0; r < 16; ++r) dp[r] = qpad ? 0.f : dp[r];           // constant rows
        }
        dsB[2 * kt] = pack_half(dp, 0);
        dsB[2 * kt + 1] = pack_half(dp, 1);
      }
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        f32x16_t o = zero16();
#pragma unroll
        for (int u = 0; u < 2 * NTK; ++u) o = mma(frag_T_slots<RS>(XA, 16 * u, dt, lane), dsB[u], o);
        store_direct<DH>(o, dQg + (long long)q * a.dq_rs, q < Tq, dt, half);
      }
    }
    if (!synced) __syncthreads();
  }
  __syncthreads();
  stage_rows2<DH, TR, TR>(XA, RS, Qg, a.q_rs, Tq, TR, XB, RS, dOg, a.do_rs, Tq, TR, tid);

  // ================= phase 2: wave = key tile.  dK and dV.
  {
    bf16_t* dKg = a.dK + (long long)b * a.dk_bs + h * DH;
    bf16_t* dVg = a.dV + (long long)b * a.dv_bs + h * DH;
    bool synced = false;
    for (int kt = wave; kt < nkt; kt += LNW) {
      const int key = kt * 32 + l31;
      const bool kvalid = key < klen, kin = key < Tk;
      bf16x8_t bK[NK], bV[NK];
#pragma unroll
      for (int s2 = 0; s2 < NK; ++s2) {
        bK[s2] = kin ? row_frag<DH>(Kg + (long long)key * a.k_rs, s2 * 16 + 8 * half) : zero8();
        bV[s2] = kin ? row_frag<DH>(Vg + (long long)key * a.v_rs, s2 * 16 + 8 * half) : zero8();
      }
      if (!synced) { __syncthreads(); synced = true; }
      f32x16_t dVt[NDT], dKt[NDT];
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) { dVt[dt] = zero16(); dKt[dt] = zero16(); }
      for (int qt = 0; qt < nqt; ++qt) {
        f32x16_t s = zero16(), dp = zero16();
#pragma unroll
        for (int s2 = 0; s2 < NK; ++s2) {
          s = mma(row_frag<DH>(XA + (qt * 32 + l31) * RS, s2 * 16 + 8 * half), bK[s2], s);      // S = Q K^T: lane = key, registers = queries
          dp = mma(row_frag<DH>(XB + (qt * 32 + l31) * RS, s2 * 16 + 8 * half), bV[s2], dp);    // dP = dO V^T
        }
        // (scalar) a tile pair of valid keys and live queries needs no select at all
        const bool inner = (kt * 32 + 32 <= klen) && (qt * 32 + 32 <= qlen) && (qt * 32 + 32 <= Tq);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int q0 = qt * 32 + 8 * g + 4 * half;
          const float4 m4 = *reinterpret_cast<const float4*>(s_m + q0);
          const float4 i4 = *reinterpret_cast<const float4*>(s_inv + q0);
          const float4 d4 = *reinterpret_cast<const float4*>(s_D + q0);
          const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ii[4] = {i4.x, i4.y, i4.z, i4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
          if (inner) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * g + e, q = q0 + e;
              const float x = s[r] * kscale;
              float pv = __builtin_amdgcn_exp2f(x - mm[e]) * ii[e];
              float gq = dp[r];
              if constexpr (DROP) {
                const bool keep = dmt_drop_keep(a.drop_seed, (unsigned)((b * a.H + h) * Tq + q) * (unsigned)Tk + (unsigned)key, a.drop_thr);
                gq = keep ? gq * a.drop_inv : 0.f;
                dp[r] = pv * (gq - dd[e]) * inv_sc;
                pv = keep ? pv * a.drop_inv : 0.f;
              } else {
                dp[r] = pv * (gq - dd[e]) * inv_sc;
              }
              s[r] = pv;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * g + e, q = q0 + e;
              float x = s[r] * kscale;
              x = kvalid ? x : PADDING_NUM * LOG2E;
              float pv = kin ? __builtin_amdgcn_exp2f(x - mm[e]) * ii[e] : 0.f;
              float gq = dp[r];
              bool keep = true;
              if constexpr (DROP) {
                keep = dmt_drop_keep(a.drop_seed, (unsigned)((b * a.H + h) * Tq + q) * (unsigned)Tk + (unsigned)key, a.drop_thr);
                gq = keep ? gq * a.drop_inv : 0.f;
              }
              float ds = kvalid ? pv * (gq - dd[e]) * inv_sc : 0.f;
              if (q >= qlen) { ds = 0.f; pv = kin ? PADDING_NUM : 0.f; }
              if constexpr (DROP) pv = keep ? pv * a.drop_inv : 0.f;
              if (q >= Tq) { ds = 0.f; pv = 0.f; }       // rows past Tq of the Q / dO tiles are not data
              s[r] = pv;            // the weights as they multiply V (query mask and dropout applied)
              dp[r] = ds;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bf16x8_t pb = pack_half(s, u), db = pack_half(dp, u);
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            dVt[dt] = mma(frag_T_slots<RS>(XB, qt * 32 + 16 * u, dt, lane), pb, dVt[dt]);      // dV^T += dO^T P
            dKt[dt] = mma(frag_T_slots<RS>(XA, qt * 32 + 16 * u, dt, lane), db, dKt[dt]);      // dK^T += Q^T dS
          }
        }
      }
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        store_direct<DH>(dVt[dt], dVg + (long long)key * a.dv_rs, kin, dt, half);
        store_direct<DH>(dKt[dt], dKg + (long long)key * a.dk_rs, kin, dt, half);
      }
    }
    if (!synced) __syncthreads();
  }
}

// ----------------------------------------------------------------------------------------- single query (decoder), long keys
// Tq == 1, 64 < Tk <= 256 (the target item attends over a 200-step encoded history: TransformerModel.py decoder).  Memory bound:
// one wavefront per (example, head); 16 lanes x 16 bytes cover one key row, four rows per wave instruction, every K / V row moves
// in coalesced 16-byte pieces.  Scores, weights and dS live in LDS (one float per key).  Forward reads K and V once; backward reads
// K twice (scores; dQ / dK) and V once.  Arithmetic as attn_q1_kernel of dmt_attn.hip (IEEE division, expf).
__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xFFFF0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xFFFF0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xFFFF0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xFFFF0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(dmt_pack_bf16(f[0], f[1]), dmt_pack_bf16(f[2], f[3]), dmt_pack_bf16(f[4], f[5]), dmt_pack_bf16(f[6], f[7]));
}
__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}

template <int DH, bool BWD>
__global__ __launch_bounds__(256) void attn_q1_long_kernel(const LongArgs a) {
  constexpr int CPR = DH / 8;        // 16-byte chunks per row
  constexpr int UN = 4;              // row passes in flight
  __shared__ float s_p[4][256], s_g[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long wid = (long long)blockIdx.x * 4 + wave;
  if (wid >= (long long)a.B * a.H) return;
  const int b = (int)(wid / a.H), h = (int)(wid % a.H);
  const int g = lane >> 4, c = lane & 15;
  const bool cact = c < CPR;
  const int cc = cact ? c : 0;       // (inactive lanes re-read chunk 0: branch-free loads; their values are kept out of every sum)
  const int Tk = a.Tk;
  int klen = a.k_lens ? a.k_lens[b] : Tk;
  klen = klen < 0 ? 0 : (klen > Tk ? Tk : klen);
  const int qlen = a.q_lens ? a.q_lens[b] : 1;
  const bool qpad = (0 >= qlen);
  const float sc = sqrtf((float)DH);
  const bf16_t* Kg = a.K + (long long)b * a.k_bs + h * DH + cc * 8;
  const bf16_t* Vg = a.V + (long long)b * a.v_bs + h * DH + cc * 8;
  float qf[8], df[8];
  unpack8(*reinterpret_cast<const uint4*>(a.Q + (long long)b * a.q_bs + h * DH + cc * 8), qf);
  if constexpr (BWD) unpack8(*reinterpret_cast<const uint4*>(a.dout + (long long)b * a.do_bs + h * DH + cc * 8), df);
  float* P = s_p[wave];
  float* G = s_g[wave];
  const int npass = (Tk + 3) >> 2;
  // ---- scores: s[k] = q . K[k] / sqrt(dh)
  for (int p0 = 0; p0 < npass; p0 += UN) {
    uint4 kr[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      int row = 4 * (p0 + u) + g;
      row = row < Tk ? row : Tk - 1;
      kr[u] = *reinterpret_cast<const uint4*>(Kg + (long long)row * a.k_rs);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      float kf[8];
      unpack8(kr[u], kf);
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) part = fmaf(qf[i], kf[i], part);
      part = group16_sum(cact ? part : 0.f);
      const int row = 4 * (p0 + u) + g;
      if (c == 0 && row < Tk) P[row] = part / sc;
    }
  }
  __builtin_amdgcn_wave_barrier();
  // ---- softmax over the keys (lane owns keys lane, lane + 64, ...)
  float sv[4], m = -3.0e38f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int key = lane + 64 * i;
    float x = key < Tk ? P[key] : -3.0e38f;
    if (key < Tk && key >= klen) x = PADDING_NUM;
    sv[i] = x;
    m = fmaxf(m, x);
  }
  m = wave_max(m);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int key = lane + 64 * i;
    sv[i] = key < Tk ? expf(sv[i] - m) : 0.f;
    sum += sv[i];
  }
  sum = wave_sum(sum);
  float keepf[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int key = lane + 64 * i;
    sv[i] = sv[i] / sum;
    keepf[i] = 1.f;
    if (a.drop_on) keepf[i] = dmt_drop_keep(a.drop_seed, (unsigned)((b * a.H + h) * a.Tq) * (unsigned)Tk + (unsigned)key, a.drop_thr) ? a.drop_inv : 0.f;
  }
  __builtin_amdgcn_wave_barrier();
  if constexpr (!BWD) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = lane + 64 * i;
      float pv = qpad ? PADDING_NUM : sv[i];
      if (key < Tk) P[key] = pv * keepf[i];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- out = sum_k P[k] V[k] + resid
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int p0 = 0; p0 < npass; p0 += UN) {
      uint4 vr[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        int row = 4 * (p0 + u) + g;
        row = row < Tk ? row : Tk - 1;
        vr[u] = *reinterpret_cast<const uint4*>(Vg + (long long)row * a.v_rs);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int row = 4 * (p0 + u) + g;
        const float pk = row < Tk ? P[row] : 0.f;
        float vf[8];
        unpack8(vr[u], vf);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(pk, vf[i], acc[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] += __shfl_xor(acc[i], 16, 64); acc[i] += __shfl_xor(acc[i], 32, 64); }
    if (g == 0 && cact) {
      if (a.resid) {
        float rf[8];
        unpack8(*reinterpret_cast<const uint4*>(a.resid + (long long)b * a.r_bs + h * DH + c * 8), rf);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += rf[i];
      }
      *reinterpret_cast<uint4*>(a.out + (long long)b * a.o_bs + h * DH + c * 8) = pack8(acc);
    }
  } else {
    // ---- dP[k] = dO . V[k]; dV[k] = P_dropped[k] dO (same pass over V)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = lane + 64 * i;
      if (key < Tk) P[key] = (qpad ? PADDING_NUM : sv[i]) * keepf[i];       // the weights as they multiplied V
    }
    __builtin_amdgcn_wave_barrier();
    bf16_t* dVg = a.dV + (long long)b * a.dv_bs + h * DH + cc * 8;
    for (int p0 = 0; p0 < npass; p0 += UN) {
      uint4 vr[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        int row = 4 * (p0 + u) + g;
        row = row < Tk ? row : Tk - 1;
        vr[u] = *reinterpret_cast<const uint4*>(Vg + (long long)row * a.v_rs);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int row = 4 * (p0 + u) + g;
        float vf[8];
        unpack8(vr[u], vf);
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) part = fmaf(df[i], vf[i], part);
        part = group16_sum(cact ? part : 0.f);
        if (row < Tk) {
          if (c == 0) G[row] = part;
          if (cact) {
            const float pk = P[row];
            float ov[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ov[i] = pk * df[i];
            *reinterpret_cast<uint4*>(dVg + (long long)row * a.dv_rs) = pack8(ov);
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- dS = P (dP - sum P dP) / sqrt(dh)
    float dpv[4], dot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = lane + 64 * i;
      dpv[i] = key < Tk ? G[key] * keepf[i] : 0.f;
      dot += key < Tk ? sv[i] * dpv[i] : 0.f;
    }
    dot = wave_sum(dot);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = lane + 64 * i;
      float ds = (!qpad && key < klen) ? sv[i] * (dpv[i] - dot) / sc : 0.f;
      if (key < Tk) G[key] = ds;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- dQ = sum_k dS[k] K[k]; dK[k] = dS[k] q
    bf16_t* dKg = a.dK + (long long)b * a.dk_bs + h * DH + cc * 8;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int p0 = 0; p0 < npass; p0 += UN) {
      uint4 kr[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        int row = 4 * (p0 + u) + g;
        row = row < Tk ? row : Tk - 1;
        kr[u] = *reinterpret_cast<const uint4*>(Kg + (long long)row * a.k_rs);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int row = 4 * (p0 + u) + g;
        const float ds = row < Tk ? G[row] : 0.f;
        float kf[8];
        unpack8(kr[u], kf);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(ds, kf[i], acc[i]);
        if (row < Tk && cact) {
          float ok[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) ok[i] = ds * qf[i];
          *reinterpret_cast<uint4*>(dKg + (long long)row * a.dk_rs) = pack8(ok);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] += __shfl_xor(acc[i], 16, 64); acc[i] += __shfl_xor(acc[i], 32, 64); }
    if (g == 0 && cact) *reinterpret_cast<uint4*>(a.dQ + (long long)b * a.dq_bs + h * DH + c * 8) = pack8(acc);
  }
}

template <bool BWD>
void launch_q1_long(const LongArgs& a, int dh, hipStream_t st) {
  const dim3 grid((unsigned)cdiv64((long long)a.B * a.H, 4)), block(256);
  switch (dh) {
    case 16: hipLaunchKernelGGL((attn_q1_long_kernel<16, BWD>), grid, block, 0, st, a); break;
    case 32: hipLaunchKernelGGL((attn_q1_long_kernel<32, BWD>), grid, block, 0, st, a); break;
    case 64: hipLaunchKernelGGL((attn_q1_long_kernel<64, BWD>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((attn_q1_long_kernel<80, BWD>), grid, block, 0, st, a); break;
  }
}

bool al16(const void* p, long long s0, long long s1) { return p == nullptr || (((uintptr_t)p) % 16 == 0 && s0 % 8 == 0 && s1 % 8 == 0); }

int fill(LongArgs& a, const dmt_attn_desc* d) {
  a.B = d->B; a.H = d->H; a.Tq = d->Tq; a.Tk = d->Tk;
  a.Q = (const bf16_t*)d->Q; a.q_bs = d->q_bs; a.q_rs = d->q_rs;
  a.K = (const bf16_t*)d->K; a.k_bs = d->k_bs; a.k_rs = d->k_rs;
  a.V = (const bf16_t*)d->V; a.v_bs = d->v_bs; a.v_rs = d->v_rs;
  a.q_lens = d->q_lens; a.k_lens = d->k_lens;
  a.resid = (const bf16_t*)d->resid; a.r_bs = d->r_bs; a.r_rs = d->r_rs;
  a.out = (bf16_t*)d->out; a.o_bs = d->o_bs; a.o_rs = d->o_rs;
  a.drop_on = (d->drop_keep > 0.f && d->drop_keep < 1.f) ? 1 : 0;
  a.drop_seed = d->drop_seed;
  a.drop_thr = a.drop_on ? (unsigned)(d->drop_keep * 16777216.0f) : 0u;
  a.drop_inv = a.drop_on ? 1.f / d->drop_keep : 1.f;
  a.dout = nullptr; a.dQ = a.dK = a.dV = nullptr;
  a.do_bs = a.do_rs = a.dq_bs = a.dq_rs = a.dk_bs = a.dk_rs = a.dv_bs = a.dv_rs = 0;
  return 0;
}

int ntk_of(int T) { return T <= 128 ? 4 : (T <= 208 ? 7 : 8); }

}  // namespace

extern "C" int dmt_attn_long_supported(int32_t dtype, int32_t dh, int32_t Tq, int32_t Tk) {
  return (dtype == DMT_BF16 && (dh == 16 || dh == 32 || dh == 64 || dh == 80) && Tq >= 1 && Tk >= 1 && Tq <= 256 && Tk <= 256) ? 1 : 0;
}

#define DMT_LONG_DISPATCH_D(KERNEL, ARGS, DROP)                                                                           \
  do {                                                                                                                    \
    const int ntk = ntk_of(d_->Tq > d_->Tk ? d_->Tq : d_->Tk);                                                            \
    const dim3 grid((unsigned)((long long)d_->B * d_->H)), block(LNW * 64);                                               \
    switch (d_->dh) {                                                                                                     \
      case 16: hipLaunchKernelGGL((KERNEL<16, 8, DROP>), grid, block, 0, st, ARGS); break;                                \
      case 32: hipLaunchKernelGGL((KERNEL<32, 8, DROP>), grid, block, 0, st, ARGS); break;                                \
      case 64: hipLaunchKernelGGL((KERNEL<64, 8, DROP>), grid, block, 0, st, ARGS); break;                                \
      default:                                                                                                            \
        if (ntk == 4) hipLaunchKernelGGL((KERNEL<80, 4, DROP>), grid, block, 0, st, ARGS);                                \
        else if (ntk == 7) hipLaunchKernelGGL((KERNEL<80, 7, DROP>), grid, block, 0, st, ARGS);                           \
        else hipLaunchKernelGGL((KERNEL<80, 8, DROP>), grid, block, 0, st, ARGS);                                         \
    }                                                                                                                     \
  } while (0)
#define DMT_LONG_DISPATCH(KERNEL, ARGS)                                                                                   \
  do {                                                                                                                    \
    if ((ARGS).drop_on) DMT_LONG_DISPATCH_D(KERNEL, ARGS, true);                                                          \
    else DMT_LONG_DISPATCH_D(KERNEL, ARGS, false);                                                                        \
  } while (0)

extern "C" int dmt_attn_long_fwd(const dmt_attn_desc* d_, void* stream) {
  DMT_CHECK_ARG(d_ != nullptr && d_->Q && d_->K && d_->V && d_->out, "dmt_attn_long_fwd: null argument");
  if (!dmt_attn_long_supported(d_->dtype, d_->dh, d_->Tq, d_->Tk)) {
    dmt_set_error("dmt_attn_long_fwd: unsupported (bf16, dh in 16/32/64/80, T <= 256): dtype %d dh %d Tq %d Tk %d", d_->dtype, d_->dh, d_->Tq, d_->Tk);
    return DMT_ERR_UNSUPPORTED;
  }
  DMT_CHECK_ARG(al16(d_->Q, d_->q_bs, d_->q_rs) && al16(d_->K, d_->k_bs, d_->k_rs) && al16(d_->V, d_->v_bs, d_->v_rs) &&
                al16(d_->resid, d_->r_bs, d_->r_rs) && al16(d_->out, d_->o_bs, d_->o_rs), "dmt_attn_long_fwd: rows must be 16-byte aligned");
  LongArgs a;
  fill(a, d_);
  hipStream_t st = (hipStream_t)stream;
  if (d_->Tq == 1) {
    launch_q1_long<false>(a, d_->dh, st);
    DMT_CHECK_LAUNCH("dmt_attn_long_fwd(q1)");
    return DMT_OK;
  }
  if (d_->mma_dtype == DMT_FP8_E4M3) {
    DMT_LONG_DISPATCH(attn_long_fwd_f8_kernel, a);
    DMT_CHECK_LAUNCH("dmt_attn_long_fwd(fp8)");
    return DMT_OK;
  }
  DMT_CHECK_ARG(d_->mma_dtype == 0 || d_->mma_dtype == d_->dtype, "dmt_attn_long_fwd: mma_dtype must be 0, the operand dtype or DMT_FP8_E4M3");
  DMT_LONG_DISPATCH(attn_long_fwd_kernel, a);
  DMT_CHECK_LAUNCH("dmt_attn_long_fwd");
  return DMT_OK;
}

extern "C" int dmt_attn_long_bwd(const dmt_attn_bwd_desc* d, void* stream) {
  DMT_CHECK_ARG(d != nullptr, "dmt_attn_long_bwd: null descriptor");
  const dmt_attn_desc* d_ = &d->f;
  DMT_CHECK_ARG(d_->Q && d_->K && d_->V && d->dout && d->dQ && d->dK && d->dV, "dmt_attn_long_bwd: null argument");
  if (!dmt_attn_long_supported(d_->dtype, d_->dh, d_->Tq, d_->Tk)) {
    dmt_set_error("dmt_attn_long_bwd: unsupported (bf16, dh in 16/32/64/80, T <= 256): dtype %d dh %d Tq %d Tk %d", d_->dtype, d_->dh, d_->Tq, d_->Tk);
    return DMT_ERR_UNSUPPORTED;
  }
  DMT_CHECK_ARG(al16(d_->Q, d_->q_bs, d_->q_rs) && al16(d_->K, d_->k_bs, d_->k_rs) && al16(d_->V, d_->v_bs, d_->v_rs) &&
                al16(d->dout, d->do_bs, d->do_rs) && al16(d->dQ, d->dq_bs, d->dq_rs) && al16(d->dK, d->dk_bs, d->dk_rs) &&
                al16(d->dV, d->dv_bs, d->dv_rs), "dmt_attn_long_bwd: rows must be 16-byte aligned");
  LongArgs a;
  fill(a, d_);
  a.dout = (const bf16_t*)d->dout; a.do_bs = d->do_bs; a.do_rs = d->do_rs;
  a.dQ = (bf16_t*)d->dQ; a.dq_bs = d->dq_bs; a.dq_rs = d->dq_rs;
  a.dK = (bf16_t*)d->dK; a.dk_bs = d->dk_bs; a.dk_rs = d->dk_rs;
  a.dV = (bf16_t*)d->dV; a.dv_bs = d->dv_bs; a.dv_rs = d->dv_rs;
  hipStream_t st = (hipStream_t)stream;
  if (d_->Tq == 1) {
    launch_q1_long<true>(a, d_->dh, st);
    DMT_CHECK_LAUNCH("dmt_attn_long_bwd(q1)");
    return DMT_OK;
  }
  DMT_LONG_DISPATCH(attn_long_bwd_kernel, a);
  DMT_CHECK_LAUNCH("dmt_attn_long_bwd");
  return DMT_OK;
}
