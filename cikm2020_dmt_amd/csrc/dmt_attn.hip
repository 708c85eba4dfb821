// Short-sequence multi-head attention core, forward and backward: one wavefront per (example, head).
// Lane k owns key k (Tk <= 64): scores are lane-local dot products against the query row broadcast from
// LDS, the softmax max / sum are wavefront shuffles, and P.V runs with lanes along the head dim.
// Backward recomputes P from Q, K (flash-style: no [B,H,T,T] tensor ever touches HBM) and keeps the
// per-key dK / dV rows in registers.  Masking follows the reference literally (see dmt_hip.h).
#include "dmt_common.h"
#include <stdlib.h>

namespace {

constexpr float PADDING_NUM = -4294967295.0f;   // -2**32 + 1 (TransformerModel_util.py:81)

struct AttnArgs {
  int B, H, dh, Tq, Tk;
  const void* Q; long long q_bs, q_rs;
  const void* K; long long k_bs, k_rs;
  const void* V; long long v_bs, v_rs;
  const int* q_lens;
  const int* k_lens;
  const void* resid; long long r_bs, r_rs;
  void* out; long long o_bs, o_rs;
  // backward only
  const void* dout; long long do_bs, do_rs;
  void* dQ; long long dq_bs, dq_rs;
  void* dK; long long dk_bs, dk_rs;
  void* dV; long long dv_bs, dv_rs;
  int lds_per_wave;    // floats
  // dropout on the attention weights (after the query mask): D = keep(idx) / keep_prob, idx = ((b*H+h)*Tq+q)*Tk+k
  unsigned drop_seed, drop_thr;
  float drop_inv;
  int drop_on;
  int vec16;   // bf16 MFMA path: every operand row is 16-byte aligned and dh % 8 == 0
  // packed rows (include/dmt_hip.h, attn_bwd_co_kernel only): example b's rows at rows row_off[b] + t, it has k_lens[b] of them;
  // ex_list: the n_list examples this launch covers
  const int* row_off;
  const int* ex_list;
  int n_list;
};

__device__ __forceinline__ float drop_factor(const AttnArgs& a, int b, int h, int q, int key) {
  if (!a.drop_on) return 1.f;
  const unsigned idx = (unsigned)((((long long)b * a.H + h) * a.Tq + q) * a.Tk + key);
  return dmt_drop_keep(a.drop_seed, idx, a.drop_thr) ? a.drop_inv : 0.f;
}

template <typename T, int DHT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int dh = DHT ? DHT : a.dh;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const long long wid = (long long)blockIdx.x * nw + wave;
  if (wid >= (long long)a.B * a.H) return;
  const int b = (int)(wid / a.H), h = (int)(wid % a.H);
  const int Tk = a.Tk, ldh = dh + 1;
  float* Ks = smem + (long long)wave * a.lds_per_wave;
  float* Vs = Ks + Tk * ldh;
  float* qbuf = Vs + Tk * ldh;
  float* pbuf = qbuf + dh;

  const T* Kg = reinterpret_cast<const T*>(a.K) + (long long)b * a.k_bs + h * dh;
  const T* Vg = reinterpret_cast<const T*>(a.V) + (long long)b * a.v_bs + h * dh;
  for (int i = lane; i < Tk * dh; i += 64) {
    const int k = i / dh, j = i - k * dh;
    Ks[k * ldh + j] = ldf<T>(Kg + (long long)k * a.k_rs + j);
    Vs[k * ldh + j] = ldf<T>(Vg + (long long)k * a.v_rs + j);
  }
  int klen = a.k_lens ? a.k_lens[b] : Tk;
  klen = klen < 0 ? 0 : (klen > Tk ? Tk : klen);
  const int qlen = a.q_lens ? a.q_lens[b] : a.Tq;
  const float sc = sqrtf((float)dh);
  const T* Qg = reinterpret_cast<const T*>(a.Q) + (long long)b * a.q_bs + h * dh;
  const T* Rg = a.resid ? reinterpret_cast<const T*>(a.resid) + (long long)b * a.r_bs + h * dh : nullptr;
  T* Og = reinterpret_cast<T*>(a.out) + (long long)b * a.o_bs + h * dh;
  __builtin_amdgcn_wave_barrier();

  for (int q = 0; q < a.Tq; ++q) {
    for (int j = lane; j < dh; j += 64) qbuf[j] = ldf<T>(Qg + (long long)q * a.q_rs + j);
    __builtin_amdgcn_wave_barrier();
    float s = 0.f;
    if (lane < Tk) {
      const float* kr = Ks + lane * ldh;
#pragma unroll 4
      for (int j = 0; j < dh; ++j) s = fmaf(qbuf[j], kr[j], s);
      s = s / sc;
      if (lane >= klen) s = PADDING_NUM;
    }
    const float m = wave_max(lane < Tk ? s : -3.0e38f);
    const float e = (lane < Tk) ? expf(s - m) : 0.f;
    const float sum = wave_sum(e);
    float p = e / sum;
    if (q >= qlen) p = PADDING_NUM;            // query mask applied AFTER the softmax (reference behaviour, F13)
    p *= drop_factor(a, b, h, q, lane);
    if (lane < Tk) pbuf[lane] = p;
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < dh; j += 64) {
      float o = 0.f;
#pragma unroll 4
      for (int k = 0; k < Tk; ++k) o = fmaf(pbuf[k], Vs[k * ldh + j], o);
      if (Rg) o += ldf<T>(Rg + (long long)q * a.r_rs + j);
      stf<T>(Og + (long long)q * a.o_rs + j, o);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename T, int DHT>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int DH = DHT;
  const int dh = DH;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const long long wid = (long long)blockIdx.x * nw + wave;
  if (wid >= (long long)a.B * a.H) return;
  const int b = (int)(wid / a.H), h = (int)(wid % a.H);
  const int Tk = a.Tk, ldh = dh + 1;
  float* Ks = smem + (long long)wave * a.lds_per_wave;
  float* Vs = Ks + Tk * ldh;
  float* qbuf = Vs + Tk * ldh;
  float* dobuf = qbuf + dh;
  float* dsbuf = dobuf + dh;

  const T* Kg = reinterpret_cast<const T*>(a.K) + (long long)b * a.k_bs + h * dh;
  const T* Vg = reinterpret_cast<const T*>(a.V) + (long long)b * a.v_bs + h * dh;
  for (int i = lane; i < Tk * dh; i += 64) {
    const int k = i / dh, j = i - k * dh;
    Ks[k * ldh + j] = ldf<T>(Kg + (long long)k * a.k_rs + j);
    Vs[k * ldh + j] = ldf<T>(Vg + (long long)k * a.v_rs + j);
  }
  int klen = a.k_lens ? a.k_lens[b] : Tk;
  klen = klen < 0 ? 0 : (klen > Tk ? Tk : klen);
  const int qlen = a.q_lens ? a.q_lens[b] : a.Tq;
  const float sc = sqrtf((float)dh);
  const T* Qg = reinterpret_cast<const T*>(a.Q) + (long long)b * a.q_bs + h * dh;
  const T* dOg = reinterpret_cast<const T*>(a.dout) + (long long)b * a.do_bs + h * dh;
  T* dQg = reinterpret_cast<T*>(a.dQ) + (long long)b * a.dq_bs + h * dh;
  float dKacc[DH], dVacc[DH];
#pragma unroll
  for (int j = 0; j < DH; ++j) { dKacc[j] = 0.f; dVacc[j] = 0.f; }
  __builtin_amdgcn_wave_barrier();

  for (int q = 0; q < a.Tq; ++q) {
    for (int j = lane; j < dh; j += 64) {
      qbuf[j] = ldf<T>(Qg + (long long)q * a.q_rs + j);
      dobuf[j] = ldf<T>(dOg + (long long)q * a.do_rs + j);
    }
    __builtin_amdgcn_wave_barrier();
    float s = 0.f, dP = 0.f;
    if (lane < Tk) {
      const float* kr = Ks + lane * ldh;
      const float* vr = Vs + lane * ldh;
#pragma unroll
      for (int j = 0; j < DH; ++j) { s = fmaf(qbuf[j], kr[j], s); dP = fmaf(dobuf[j], vr[j], dP); }
      s = s / sc;
      if (lane >= klen) s = PADDING_NUM;
    }
    const float m = wave_max(lane < Tk ? s : -3.0e38f);
    const float e = (lane < Tk) ? expf(s - m) : 0.f;
    const float sum = wave_sum(e);
    float p = e / sum;
    float ds = 0.f;
    const float Dk = drop_factor(a, b, h, q, lane);
    dP *= Dk;                                            // gradient w.r.t. the pre-dropout weights
    if (q < qlen) {
      const float dot = wave_sum(lane < Tk ? p * dP : 0.f);
      ds = (lane < klen) ? p * (dP - dot) / sc : 0.f;   // tf.where(key_mask, x, pad): no gradient into masked keys
    } else {
      p = PADDING_NUM;                                   // constant rows: gradient reaches V only
    }
    p *= Dk;                                             // dropped weights feed dV
    if (lane < Tk) {
#pragma unroll
      for (int j = 0; j < DH; ++j) {
        dVacc[j] = fmaf(p, dobuf[j], dVacc[j]);
        dKacc[j] = fmaf(ds, qbuf[j], dKacc[j]);
      }
      dsbuf[lane] = ds;
    }
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < dh; j += 64) {
      float g = 0.f;
#pragma unroll 4
      for (int k = 0; k < Tk; ++k) g = fmaf(dsbuf[k], Ks[k * ldh + j], g);
      stf<T>(dQg + (long long)q * a.dq_rs + j, g);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < Tk) {
    T* dKg = reinterpret_cast<T*>(a.dK) + (long long)b * a.dk_bs + (long long)lane * a.dk_rs + h * dh;
    T* dVg = reinterpret_cast<T*>(a.dV) + (long long)b * a.dv_bs + (long long)lane * a.dv_rs + h * dh;
#pragma unroll
    for (int j = 0; j < DH; ++j) { stf<T>(dKg + j, dKacc[j]); stf<T>(dVg + j, dVacc[j]); }
  }
}

// ------------------------------------------------------------------------------------------------------------
// bf16 MFMA forward.  One wavefront per (example, head), Tq, Tk <= 64.
//   S^T[key, q] = K Q^T      v_mfma_f32_32x32x16_bf16, A = K rows, B = Q rows, both read k-contiguous from global
//                            (swapped operands: each lane owns ONE query column, its 32+32 keys sit in the lane
//                            pair (l, l^32) -> softmax max / sum = in-register reductions + one cross-half exchange)
//   O^T[j, q]   = V^T P^T    B = P^T straight from the softmax registers (k-slot <-> key mapping below), A = V^T built
//                            from the LDS-staged V tile with 2-byte reads (lanes along j: conflict free)
// k-slot mapping of one k16 step u (keys 16u .. 16u+15): slot i of lane-half h  <->  key 16u + (i&3) + 8(i>>2) + 4h,
// which is exactly where the 32x32 accumulator layout leaves the scores, so P never moves between lanes.
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ bf16x8_t load_frag8(const bf16_t* __restrict__ rowp, int j0, bool row_ok, int dh, bool vec16 = false) {
  union { bf16x8_t v; uint2 h[2]; uint4 q; } u;
  u.h[0] = make_uint2(0u, 0u);
  u.h[1] = make_uint2(0u, 0u);
  if (vec16) {   // wave-uniform: dh % 8 == 0 and 16-byte aligned rows -> one request per lane instead of two
    if (row_ok && j0 + 8 <= dh) u.q = *reinterpret_cast<const uint4*>(rowp + j0);
    return u.v;
  }
  if (row_ok) {
    if (j0 + 4 <= dh) u.h[0] = *reinterpret_cast<const uint2*>(rowp + j0);
    if (j0 + 8 <= dh) u.h[1] = *reinterpret_cast<const uint2*>(rowp + j0 + 4);
  }
  return u.v;
}

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) { return dmt_pack_bf16(lo, hi); }

constexpr float LOG2E = 1.44269504088896340736f;

// Dropout keep bits of this lane's 32 score slots of query q: bit (16 kt + r) <-> key 32 kt + (r & 3) + 8 (r >> 2) + 4 half.
// Same counter as drop_factor(): flat index into the [B, H, Tq, Tk] weight tensor (low 32 bits).
template <int NTK = 2>
__device__ __forceinline__ unsigned drop_bits(const AttnArgs& a, int b, int h, int q, int half) {
  if (!a.drop_on) return 0xFFFFFFFFu;
  const unsigned base = (unsigned)((b * a.H + h) * a.Tq + q) * (unsigned)a.Tk + 4u * half;
  unsigned bits = 0;
#pragma unroll
  for (int e = 0; e < 16 * NTK; ++e) {
    const unsigned key = (e >> 4) * 32 + ((e & 15) & 3) + 8 * ((e & 15) >> 2);
    bits |= (dmt_drop_keep(a.drop_seed, base + key, a.drop_thr) ? 1u : 0u) << e;
  }
  return bits;
}

constexpr int TLD = 72;   // row stride (elements) of a transposed [dh][64 rows] bf16 LDS tile: 144 B = 9 x 16 B (odd)

// Stage X[row][j] (global, `n_rows` valid rows, row stride rs) as the TRANSPOSED tile Xt[j][row] (row < 64, zero filled).
// Each work item moves a 4(rows) x 4(dims) block: four 8-byte loads, a 4x4 bf16 transpose in registers, four 8-byte
// LDS stores.  Lanes run along rows first, so the stores of 16 lanes cover 128 contiguous bytes of one LDS row.
template <int DH>
__device__ __forceinline__ void stage_transposed(bf16_t* __restrict__ Xt, const bf16_t* __restrict__ Xg, long long rs, int n_rows, int lane) {
  constexpr int NJ = DH / 4;
  for (int item = lane; item < 16 * NJ; item += 64) {
    const int rg = item & 15, jg = item >> 4;
    const int r0 = rg * 4, j0 = jg * 4;
    uint2 a = make_uint2(0u, 0u), b = a, c = a, d = a;
    if (r0 + 0 < n_rows) a = *reinterpret_cast<const uint2*>(Xg + (long long)(r0 + 0) * rs + j0);
    if (r0 + 1 < n_rows) b = *reinterpret_cast<const uint2*>(Xg + (long long)(r0 + 1) * rs + j0);
    if (r0 + 2 < n_rows) c = *reinterpret_cast<const uint2*>(Xg + (long long)(r0 + 2) * rs + j0);
    if (r0 + 3 < n_rows) d = *reinterpret_cast<const uint2*>(Xg + (long long)(r0 + 3) * rs + j0);
    uint2 o0, o1, o2, o3;
    o0.x = (a.x & 0xFFFFu) | (b.x << 16);      o0.y = (c.x & 0xFFFFu) | (d.x << 16);
    o1.x = (a.x >> 16) | (b.x & 0xFFFF0000u);  o1.y = (c.x >> 16) | (d.x & 0xFFFF0000u);
    o2.x = (a.y & 0xFFFFu) | (b.y << 16);      o2.y = (c.y & 0xFFFFu) | (d.y << 16);
    o3.x = (a.y >> 16) | (b.y & 0xFFFF0000u);  o3.y = (c.y >> 16) | (d.y & 0xFFFF0000u);
    *reinterpret_cast<uint2*>(Xt + (j0 + 0) * TLD + r0) = o0;
    *reinterpret_cast<uint2*>(Xt + (j0 + 1) * TLD + r0) = o1;
    *reinterpret_cast<uint2*>(Xt + (j0 + 2) * TLD + r0) = o2;
    *reinterpret_cast<uint2*>(Xt + (j0 + 3) * TLD + r0) = o3;
  }
}

// A fragment (row j of X^T) in ACCUMULATOR slot order: element i <-> row 16u + (i&3) + 8(i>>2) + 4*half  (two 8-byte reads)
__device__ __forceinline__ bf16x8_t frag_T_slots(const bf16_t* __restrict__ Xt, int u, int half, int j) {
  union { bf16x8_t v; uint2 h[2]; } f;
  f.h[0] = *reinterpret_cast<const uint2*>(Xt + j * TLD + 16 * u + 4 * half);
  f.h[1] = *reinterpret_cast<const uint2*>(Xt + j * TLD + 16 * u + 8 + 4 * half);
  return f.v;
}

// A fragment (row j of X^T) in NATURAL order: element i <-> row 16u + 8*half + i  (one 16-byte read)
__device__ __forceinline__ bf16x8_t frag_T_rows(const bf16_t* __restrict__ Xt, int u, int half, int j) {
  return *reinterpret_cast<const bf16x8_t*>(Xt + j * TLD + 16 * u + 8 * half);
}

template <int DH>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const AttnArgs a) {
  constexpr int NK = (DH + 15) / 16;
  constexpr int NDT = (DH + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const long long wid = (long long)blockIdx.x * nw + wave;
  if (wid >= (long long)a.B * a.H) return;
  const int b = (int)(wid / a.H), h = (int)(wid % a.H);
  const int Tq = a.Tq, Tk = a.Tk;
  const int half = lane >> 5, l31 = lane & 31;
  bf16_t* Vt = reinterpret_cast<bf16_t*>(smem) + (long long)wave * DH * TLD;

  const bf16_t* Qg = reinterpret_cast<const bf16_t*>(a.Q) + (long long)b * a.q_bs + h * DH;
  const bf16_t* Kg = reinterpret_cast<const bf16_t*>(a.K) + (long long)b * a.k_bs + h * DH;
  const bf16_t* Vg = reinterpret_cast<const bf16_t*>(a.V) + (long long)b * a.v_bs + h * DH;

  // ---- stage V^T (keys >= Tk zero filled: 0 * garbage must not make NaN)
  stage_transposed<DH>(Vt, Vg, a.v_rs, Tk, lane);

  // ---- S^T = K Q^T
  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  {
    bf16x8_t aK[2][NK], bQ[2][NK];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = t * 32 + l31;
#pragma unroll
      for (int s2 = 0; s2 < NK; ++s2) {
        const int j0 = s2 * 16 + 8 * half;
        aK[t][s2] = load_frag8(Kg + (long long)row * a.k_rs, j0, row < Tk, DH, a.vec16 != 0);
        bQ[t][s2] = load_frag8(Qg + (long long)row * a.q_rs, j0, row < Tq, DH, a.vec16 != 0);
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < NK; ++s2)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          acc[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aK[kt][s2], bQ[qt][s2], acc[kt][qt], 0, 0, 0);
  }

  // ---- masked softmax over keys, per query column.  Scores are kept in log2 units (scale folded into one multiply,
  //      v_exp_f32 is exp2); masked keys get the reference's padding value, so an all-masked row is uniform as there.
  int klen = a.k_lens ? a.k_lens[b] : Tk;
  klen = klen < 0 ? 0 : (klen > Tk ? Tk : klen);
  const int qlen = a.q_lens ? a.q_lens[b] : Tq;
  const float kscale = LOG2E / sqrtf((float)DH);
  const int kl = klen - 4 * half, tk = Tk - 4 * half;   // slot constant c: key = c + 4 half
  bf16x8_t pB[2][4];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qt * 32 + l31;
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
        float x = acc[kt][qt][r] * kscale;
        x = (c >= kl) ? PADDING_NUM * LOG2E : x;
        x = (c >= tk) ? -3.0e38f : x;
        acc[kt][qt][r] = x;
        m = fmaxf(m, x);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(acc[kt][qt][r] - m);   // slots past Tk: exp2(-3e38 - m) = 0
        acc[kt][qt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv_sum = __builtin_amdgcn_rcpf(sum);
    const bool qpad = (q >= qlen);
    const unsigned keep = drop_bits(a, b, h, q, half);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      union { bf16x8_t v; unsigned w[4]; } f;
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const int r0 = 8 * (u & 1) + i;
        float p0 = acc[u >> 1][qt][r0] * inv_sum, p1 = acc[u >> 1][qt][r0 + 1] * inv_sum;
        const int c0 = (u >> 1) * 32 + (r0 & 3) + 8 * (r0 >> 2);
        if (qpad) {      // query mask applied AFTER the softmax (reference behaviour)
          p0 = (c0 < tk) ? PADDING_NUM : 0.f;
          p1 = (c0 + 1 < tk) ? PADDING_NUM : 0.f;
        }
        if (a.drop_on) {
          const int e0 = (u >> 1) * 16 + r0;
          p0 = ((keep >> e0) & 1u) ? p0 * a.drop_inv : 0.f;
          p1 = ((keep >> (e0 + 1)) & 1u) ? p1 * a.drop_inv : 0.f;
        }
        f.w[i >> 1] = pack_bf16(p0, p1);
      }
      pB[qt][u] = f.v;
    }
  }
  __builtin_amdgcn_wave_barrier();

  // ---- O^T = V^T P^T, residual, store
  const bf16_t* Rg = a.resid ? reinterpret_cast<const bf16_t*>(a.resid) + (long long)b * a.r_bs + h * DH : nullptr;
  bf16_t* Og = reinterpret_cast<bf16_t*>(a.out) + (long long)b * a.o_bs + h * DH;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) {
    f32x16_t o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    const int jj = (dt * 32 + l31 < DH) ? dt * 32 + l31 : DH - 1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bf16x8_t av = frag_T_slots(Vt, u, half, jj);
      o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, pB[0][u], o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, pB[1][u], o[1], 0, 0, 0);
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int q = qt * 32 + l31;
      if (q >= Tq) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = dt * 32 + 8 * g + 4 * half;
        if (j + 4 > DH) continue;
        float x0 = o[qt][4 * g + 0], x1 = o[qt][4 * g + 1], x2 = o[qt][4 * g + 2], x3 = o[qt][4 * g + 3];
        if (Rg) {
          union { uint2 v; bf16_t e[4]; } rv;
          rv.v = *reinterpret_cast<const uint2*>(Rg + (long long)q * a.r_rs + j);
          x0 += bf2f(rv.e[0]); x1 += bf2f(rv.e[1]); x2 += bf2f(rv.e[2]); x3 += bf2f(rv.e[3]);
        }
        uint2 ov;
        ov.x = pack_bf16(x0, x1);
        ov.y = pack_bf16(x2, x3);
        *reinterpret_cast<uint2*>(Og + (long long)q * a.o_rs + j) = ov;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Coalesced variant of the forward kernel (dh % 8 == 0, 16-byte aligned rows).  The kernel above reads every operand row by
// row PER LANE (64 lanes x 8/16 bytes from 64 different cache lines per instruction): rocprofv3 --pmc showed the vector memory
// pipe saturated (TCP busy 96 %, TD busy 89 %, 16x more L1 line accesses than the bytes need) while HBM, VALU and MFMA idled.
// Here every global access is a 16-byte chunk with consecutive lanes on consecutive chunks of a row (a head's row = dh*2
// contiguous bytes), tiles go through ONE wave-private LDS region that is reused for Q, K, V and the output:
//   Q, K : [64 rows][dh + 8]   16-byte fragment reads, rows 44 banks apart -> conflict free
//   V    : [64 keys][RSV]      untransposed; the A operand V^T comes from ds_read_b64_tr_b16 (4 keys x 16 dims per 16 lanes)
//   O    : [32 q][36] fp32 per (32 dims, query half) in a small second tile, read back as rows: residual add, one rounding,
//          16-byte stores (64-byte runs per row)
template <int DH> struct CoTile {
  static constexpr int CH = DH / 8;                 // 16-byte chunks per row
  static constexpr int RS = DH + 8;                 // Q / K row stride (elements)
  static constexpr int rsv() { int r = (DH + 7) / 8 * 8; while (r % 128 != 32 && r % 128 != 96) r += 8; return r; }
  static constexpr int RSV = rsv();                 // V row stride: 4 consecutive key rows x 64 B land in disjoint banks
  static constexpr int RO = 36;                     // fp32 staging row stride of one [32 q][32 dims] output sub-tile
  static constexpr int QKV_BYTES = ((64 * RS * 2 > 64 * RSV * 2 + 256 ? 64 * RS * 2 : 64 * RSV * 2 + 256) + 255) / 256 * 256;
  static constexpr int BYTES = QKV_BYTES + 32 * RO * 4;   // + the output staging tile
};

// Row-chunk mapping of a [64 rows][dh] tile onto the wave: RPI whole rows per pass (lane -> row lane / CH, chunk lane % CH),
// so the per-pass address is lane_offset + pass * constant (one VGPR per tensor instead of one 64-bit address per pass).
template <int DH> struct CoMap {
  static constexpr int CH = DH / 8;
  static constexpr int RPI = 64 / CH;                 // rows per pass
  static constexpr int NI = (64 + RPI - 1) / RPI;     // passes
};
template <int DH>
__device__ __forceinline__ void co_load(uint4 (&buf)[CoMap<DH>::NI], const bf16_t* __restrict__ Xg, long long rs, int n_rows, int lane) {
  typedef CoMap<DH> M;
  const int lr = lane / M::CH, lc = lane - lr * M::CH;
  const bool act = lr < M::RPI;
  const char* lp = reinterpret_cast<const char*>(Xg) + ((unsigned)lr * (unsigned)rs + (unsigned)lc * 8u) * 2u;
#pragma unroll
  for (int i = 0; i < M::NI; ++i) {
    const int r = i * M::RPI + lr;
    buf[i] = (act && r < n_rows && r < 64) ? *reinterpret_cast<const uint4*>(lp + (size_t)i * M::RPI * (size_t)rs * 2) : make_uint4(0u, 0u, 0u, 0u);
  }
}
template <int DH>
__device__ __forceinline__ void co_store(const uint4 (&buf)[CoMap<DH>::NI], bf16_t* __restrict__ Xt, int row_stride, int lane) {
  typedef CoMap<DH> M;
  const int lr = lane / M::CH, lc = lane - lr * M::CH;
  if (lr >= M::RPI) return;
#pragma unroll
  for (int i = 0; i < M::NI; ++i) {
    const int r = i * M::RPI + lr;
    if (r < 64) *reinterpret_cast<uint4*>(Xt + r * row_stride + lc * 8) = buf[i];
  }
}
// 16-byte MFMA fragment of row `row`, k = j0 .. j0+7 (zero past dh)
template <int DH>
__device__ __forceinline__ bf16x8_t co_frag(const bf16_t* __restrict__ Xt, int row, int j0) {
  union { bf16x8_t v; uint4 q; } u;
  u.q = make_uint4(0u, 0u, 0u, 0u);
  if (j0 + 8 <= DH) u.v = *reinterpret_cast<const bf16x8_t*>(Xt + row * CoTile<DH>::RS + j0);
  return u.v;
}

// NTQ / NTK = number of 32-row query / key tiles (1 when Tq resp. Tk <= 32: the 10-step cart sequence runs 1 x 1, the
// single-query decoder attention 1 x 2)
template <int DH, int NTQ, int NTK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_fwd_co_kernel(const AttnArgs a) {
  typedef CoTile<DH> CT;
  constexpr int NK = (DH + 15) / 16;
  constexpr int NDT = (DH + 31) / 32;
  constexpr int CH = CT::CH;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const long long wid = (long long)blockIdx.x * nw + wave;
  if (wid >= (long long)a.B * a.H) return;
  const int b = (int)(wid / a.H), h = (int)(wid % a.H);
  const int Tq = a.Tq, Tk = a.Tk;
  const int half = lane >> 5, l31 = lane & 31;
  bf16_t* R = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(smem) + (size_t)wave * CT::BYTES);

  const bf16_t* Qg = reinterpret_cast<const bf16_t*>(a.Q) + (long long)b * a.q_bs + h * DH;
  const bf16_t* Kg = reinterpret_cast<const bf16_t*>(a.K) + (long long)b * a.k_bs + h * DH;
  const bf16_t* Vg = reinterpret_cast<const bf16_t*>(a.V) + (long long)b * a.v_bs + h * DH;

  uint4 gq[CoMap<DH>::NI], gk[CoMap<DH>::NI];
  co_load<DH>(gq, Qg, a.q_rs, Tq, lane);
  co_load<DH>(gk, Kg, a.k_rs, Tk, lane);

  // ---- S^T = K Q^T
  f32x16_t acc[NTK][NTQ];
#pragma unroll
  for (int i = 0; i < NTK; ++i)
#pragma unroll
    for (int j = 0; j < NTQ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  {
    bf16x8_t bQ[NTQ][NK];
    co_store<DH>(gq, R, CT::RS, lane);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < NTQ; ++t)
#pragma unroll
      for (int s2 = 0; s2 < NK; ++s2) bQ[t][s2] = co_frag<DH>(R, t * 32 + l31, s2 * 16 + 8 * half);
    __builtin_amdgcn_wave_barrier();
    co_load<DH>(gq, Vg, a.v_rs, Tk, lane);          // V rides in Q's registers while K is consumed
    co_store<DH>(gk, R, CT::RS, lane);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s2 = 0; s2 < NK; ++s2)
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt) {
        const bf16x8_t kf = co_frag<DH>(R, kt * 32 + l31, s2 * 16 + 8 * half);
#pragma unroll
        for (int qt = 0; qt < NTQ; ++qt) acc[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, bQ[qt][s2], acc[kt][qt], 0, 0, 0);
      }
    __builtin_amdgcn_wave_barrier();
  }
  co_store<DH>(gq, R, CT::RSV, lane);               // V tile (keys >= Tk zero filled: 0 * garbage must not make NaN)
  // the residual pieces this lane adds in the epilogue (row = 32 qt + (item >> 2), dims dt*32 + 8 (item & 3)), requested now so
  // that their latency hides behind the softmax
  const bf16_t* Rg = a.resid ? reinterpret_cast<const bf16_t*>(a.resid) + (long long)b * a.r_bs + h * DH : nullptr;
  uint4 rres[NDT][NTQ][2];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int qt = 0; qt < NTQ; ++qt)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int item = lane + 64 * i, q = qt * 32 + (item >> 2), j = dt * 32 + 8 * (item & 3);
        rres[dt][qt][i] = (Rg && q < Tq && j + 8 <= DH) ? *reinterpret_cast<const uint4*>(Rg + (long long)q * a.r_rs + j) : make_uint4(0u, 0u, 0u, 0u);
      }

  // ---- masked softmax over keys, per query column (same arithmetic as attn_fwd_mfma_kernel)
  int klen = a.k_lens ? a.k_lens[b] : Tk;
  klen = klen < 0 ? 0 : (klen > Tk ? Tk : klen);
  const int qlen = a.q_lens ? a.q_lens[b] : Tq;
  const float kscale = LOG2E / sqrtf((float)DH);
  const int kl = klen - 4 * half, tk = Tk - 4 * half;
  bf16x8_t pB[NTQ][2 * NTK];
#pragma unroll
  for (int qt = 0; qt < NTQ; ++qt) {
    const int q = qt * 32 + l31;
    float m = -3.0e38f;
    int klv = kl, tkv = tk;          // (opaque copies: see attn_bwd_co_kernel)
    asm volatile("" : "+v"(klv), "+v"(tkv));
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
        float x = acc[kt][qt][r] * kscale;
        x = (c >= klv) ? PADDING_NUM * LOG2E : x;
        x = (c >= tkv) ? -3.0e38f : x;
        acc[kt][qt][r] = x;
        m = fmaxf(m, x);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(acc[kt][qt][r] - m);
        acc[kt][qt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv_sum = __builtin_amdgcn_rcpf(sum);
    const bool qpad = (q >= qlen);
    const unsigned keep = drop_bits<NTK>(a, b, h, q, half);
#pragma unroll
    for (int u = 0; u < 2 * NTK; ++u) {
      union { bf16x8_t v; unsigned w[4]; } f;
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const int r0 = 8 * (u & 1) + i;
        float p0 = acc[u >> 1][qt][r0] * inv_sum, p1 = acc[u >> 1][qt][r0 + 1] * inv_sum;
        const int c0 = (u >> 1) * 32 + (r0 & 3) + 8 * (r0 >> 2);
        if (qpad) {
          p0 = (c0 < tk) ? PADDING_NUM : 0.f;
          p1 = (c0 + 1 < tk) ? PADDING_NUM : 0.f;
        }
        if (a.drop_on) {
          const int e0 = (u >> 1) * 16 + r0;
          p0 = ((keep >> e0) & 1u) ? p0 * a.drop_inv : 0.f;
          p1 = ((keep >> (e0 + 1)) & 1u) ? p1 * a.drop_inv : 0.f;
        }
        f.w[i >> 1] = pack_bf16(p0, p1);
      }
      pB[qt][u] = f.v;
    }
  }
  __builtin_amdgcn_wave_barrier();

  // ---- O^T = V^T P^T per 32 output dims: A = V^T through transposing LDS reads (slot i of lane-half h <-> key
  //      16u + (i&3) + 8(i>>2) + 4h); epilogue per query half: registers -> fp32 rows in LDS -> (+ residual) -> bf16 rows
  bf16_t* Og = reinterpret_cast<bf16_t*>(a.out) + (long long)b * a.o_bs + h * DH;
  float* Of = reinterpret_cast<float*>(reinterpret_cast<char*>(R) + CT::QKV_BYTES);
  typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
  typedef __attribute__((address_space(3))) bf16x4_t* lds_p4;
  const int i16 = lane & 15, jgrp = (lane >> 4) & 1;
  const bf16_t* vbase = R + (4 * half + (i16 >> 2)) * CT::RSV + 16 * jgrp + 4 * (i16 & 3);
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) {
    f32x16_t o[NTQ];
#pragma unroll
    for (int qt = 0; qt < NTQ; ++qt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qt][r] = 0.f;
#pragma unroll
    for (int u = 0; u < 2 * NTK; ++u) {
      const bf16_t* vp = vbase + 16 * u * CT::RSV + dt * 32;
      const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p4)(vp));
      const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p4)(vp + 8 * CT::RSV));
      const bf16x8_t av = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int qt = 0; qt < NTQ; ++qt) o[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, pB[qt][u], o[qt], 0, 0, 0);
    }
#pragma unroll
    for (int qt = 0; qt < NTQ; ++qt) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(Of + l31 * CT::RO + 8 * g + 4 * half) = make_float4(o[qt][4 * g], o[qt][4 * g + 1], o[qt][4 * g + 2], o[qt][4 * g + 3]);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int item = lane + 64 * i, qr = item >> 2, q = qt * 32 + qr, j = dt * 32 + 8 * (item & 3);
        if (q >= Tq || j + 8 > DH) continue;
        const float4 x0 = *reinterpret_cast<const float4*>(Of + qr * CT::RO + 8 * (item & 3));
        const float4 x1 = *reinterpret_cast<const float4*>(Of + qr * CT::RO + 8 * (item & 3) + 4);
        float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const unsigned w[4] = {rres[dt][qt][i].x, rres[dt][qt][i].y, rres[dt][qt][i].z, rres[dt][qt][i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[2 * e] += __uint_as_float(w[e] << 16); f[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
        uint4 ov;
        ov.x = pack_bf16(f[0], f[1]); ov.y = pack_bf16(f[2], f[3]); ov.z = pack_bf16(f[4], f[5]); ov.w = pack_bf16(f[6], f[7]);
        *reinterpret_cast<uint4*>(Og + (long long)q * a.o_rs + j) = ov;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// Store an O^T-style accumulator pair (rows = head-dim in registers, column = lane's row of the output matrix).
template <int DH>
__device__ __forceinline__ void store_rows_T(const f32x16_t (&o)[2], bf16_t* __restrict__ base, long long rs, int dt, int l31,
                                             int half, int n_rows) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = t * 32 + l31;
    if (row >= n_rows) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int j = dt * 32 + 8 * g + 4 * half;
      if (j + 4 > DH) continue;
      uint2 ov;
      ov.x = pack_bf16(o[t][4 * g + 0], o[t][4 * g + 1]);
      ov.y = pack_bf16(o[t][4 * g + 2], o[t][4 * g + 3]);
      *reinterpret_cast<uint2*>(base + (long long)row * rs + j) = ov;
    }
  }
}

// bf16 MFMA backward (flash style: P recomputed).  Same wave-per-(example, head) decomposition as the forward.
//   S^T = K Q^T, dP^T = V dO^T                (operands k-contiguous from global)
//   dS  = P * (dP - rowsum(P*dP)) / sqrt(dh)  in the accumulator layout (lane = query column)
//   dQ^T = K^T dS^T                           (B = dS from registers, A = K^T from the LDS tile)
//   dV^T = dO^T P,  dK^T = Q^T dS             (P, dS transposed through LDS as [key][q]; A = dO^T / Q^T from LDS tiles)
template <int DH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_mfma_kernel(const AttnArgs a) {
  constexpr int NK = (DH + 15) / 16;
  constexpr int NDT = (DH + 31) / 32;
  constexpr int PLD = 72;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const long long wid = blockIdx.x;
  const int b = (int)(wid / a.H), h = (int)(wid % a.H);
  const int Tq = a.Tq, Tk = a.Tk;
  const int half = lane >> 5, l31 = lane & 31;
  // TWO LDS regions (20.7 KB per wave => 7 waves per CU): Tt holds ONE transposed operand tile at a time (K^T for dQ, then
  // dO^T for dV, then Q^T for dK -- each is staged right before its product), PD holds P and later dS as [key][q].
  constexpr int REG = (DH > 64 ? DH : 64) * TLD;
  bf16_t* Tt = reinterpret_cast<bf16_t*>(smem);
  bf16_t* PD = Tt + REG;
  bf16_t* Kt = Tt;
  bf16_t* dOt = Tt;
  bf16_t* Qt = Tt;
  bf16_t* PL = PD;
  bf16_t* DL = PD;

  const bf16_t* Qg = reinterpret_cast<const bf16_t*>(a.Q) + (long long)b * a.q_bs + h * DH;
  const bf16_t* Kg = reinterpret_cast<const bf16_t*>(a.K) + (long long)b * a.k_bs + h * DH;
  const bf16_t* Vg = reinterpret_cast<const bf16_t*>(a.V) + (long long)b * a.v_bs + h * DH;
  const bf16_t* dOg = reinterpret_cast<const bf16_t*>(a.dout) + (long long)b * a.do_bs + h * DH;

  stage_transposed<DH>(Kt, Kg, a.k_rs, Tk, lane);

  int klen = a.k_lens ? a.k_lens[b] : Tk;
  klen = klen < 0 ? 0 : (klen > Tk ? Tk : klen);
  const int qlen = a.q_lens ? a.q_lens[b] : Tq;
  const float kscale = LOG2E / sqrtf((float)DH), inv_sc = 1.0f / sqrtf((float)DH);
  const int kl = klen - 4 * half, tk = Tk - 4 * half;   // slot constant c: key = c + 4 half
  bf16x8_t dsB[2][4];     // dS  (B operand of dQ^T, later copied to LDS as [key][q])
  unsigned pP[2][16];     // P as it feeds dV (query mask and dropout applied), packed pairs of accumulator slots
  // one query tile (32 queries x 64 keys) at a time: 64 accumulator registers live instead of 128
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qt * 32 + l31;
    f32x16_t acc[2], dp[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[kt][r] = 0.f; dp[kt][r] = 0.f; }
    {
      // K / V row fragments are re-read per query tile (L1/L2 hits): keeping them live costs 80 registers and the second wave per SIMD
      bf16x8_t aK[2][NK], aV[2][NK];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = t * 32 + l31;
#pragma unroll
        for (int s2 = 0; s2 < NK; ++s2) {
          const int j0 = s2 * 16 + 8 * half;
          aK[t][s2] = load_frag8(Kg + (long long)row * a.k_rs, j0, row < Tk, DH, a.vec16 != 0);
          aV[t][s2] = load_frag8(Vg + (long long)row * a.v_rs, j0, row < Tk, DH, a.vec16 != 0);
        }
      }
      bf16x8_t bQ[NK], bD[NK];
#pragma unroll
      for (int s2 = 0; s2 < NK; ++s2) {
        const int j0 = s2 * 16 + 8 * half;
        bQ[s2] = load_frag8(Qg + (long long)q * a.q_rs, j0, q < Tq, DH, a.vec16 != 0);
        bD[s2] = load_frag8(dOg + (long long)q * a.do_rs, j0, q < Tq, DH, a.vec16 != 0);
      }
#pragma unroll
      for (int s2 = 0; s2 < NK; ++s2)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aK[kt][s2], bQ[s2], acc[kt], 0, 0, 0);   // S^T = K Q^T
          dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aV[kt][s2], bD[s2], dp[kt], 0, 0, 0);     // dP^T = V dO^T
        }
    }
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
        float x = acc[kt][r] * kscale;
        x = (c >= kl) ? PADDING_NUM * LOG2E : x;
        x = (c >= tk) ? -3.0e38f : x;
        acc[kt][r] = x;
        m = fmaxf(m, x);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(acc[kt][r] - m);
        acc[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv_sum = __builtin_amdgcn_rcpf(sum);
    const unsigned keep = drop_bits(a, b, h, q, half);
    float dot = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = acc[kt][r] * inv_sum;
        acc[kt][r] = pv;
        float gq = dp[kt][r];
        if (a.drop_on) gq = ((keep >> (kt * 16 + r)) & 1u) ? gq * a.drop_inv : 0.f;   // gradient w.r.t. the pre-dropout weights
        dp[kt][r] = gq;
        dot += pv * gq;
      }
    dot += __shfl_xor(dot, 32, 64);
    const bool qpad = (q >= qlen);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
        float pv = acc[kt][r];
        float ds = (c < kl) ? pv * (dp[kt][r] - dot) * inv_sc : 0.f;      // no gradient into masked keys
        if (qpad) { ds = 0.f; pv = (c < tk) ? PADDING_NUM : 0.f; }        // constant rows: gradient reaches V only
        if (a.drop_on) pv = ((keep >> (kt * 16 + r)) & 1u) ? pv * a.drop_inv : 0.f;   // dropped weights feed dV
        dp[kt][r] = ds;
        acc[kt][r] = pv;
      }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      union { bf16x8_t v; unsigned w[4]; } f;
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const int r0 = 8 * (u & 1) + i;
        f.w[i >> 1] = pack_bf16(dp[u >> 1][r0], dp[u >> 1][r0 + 1]);
      }
      dsB[qt][u] = f.v;
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) pP[qt][kt * 8 + (r >> 1)] = pack_bf16(acc[kt][r], acc[kt][r + 1]);
  }
  __builtin_amdgcn_wave_barrier();

  // ---- dQ^T = K^T dS^T
  bf16_t* dQg = reinterpret_cast<bf16_t*>(a.dQ) + (long long)b * a.dq_bs + h * DH;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) {
    f32x16_t o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    const int jj = (dt * 32 + l31 < DH) ? dt * 32 + l31 : DH - 1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bf16x8_t av = frag_T_slots(Kt, u, half, jj);
      o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, dsB[0][u], o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, dsB[1][u], o[1], 0, 0, 0);
    }
    store_rows_T<DH>(o, dQg, a.dq_rs, dt, l31, half, Tq);
  }

  // ---- dV^T = dO^T P   (reduction over queries, natural k slots).  dO^T replaces the dead K^T tile; P goes to LDS as [key][q].
  bf16_t* dKg = reinterpret_cast<bf16_t*>(a.dK) + (long long)b * a.dk_bs + h * DH;
  bf16_t* dVg = reinterpret_cast<bf16_t*>(a.dV) + (long long)b * a.dv_bs + h * DH;
  __builtin_amdgcn_wave_barrier();
  stage_transposed<DH>(dOt, dOg, a.do_rs, Tq, lane);
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        PL[(kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * PLD + qt * 32 + l31] = (bf16_t)(pP[qt][kt * 8 + (r >> 1)] >> (16 * (r & 1)));
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) {
    f32x16_t ov[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ov[0][r] = 0.f; ov[1][r] = 0.f; }
    const int jj = (dt * 32 + l31 < DH) ? dt * 32 + l31 : DH - 1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bf16x8_t ad = frag_T_rows(dOt, u, half, jj);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const bf16x8_t bp = *reinterpret_cast<const bf16x8_t*>(PL + (kt * 32 + l31) * PLD + 16 * u + 8 * half);
        ov[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, bp, ov[kt], 0, 0, 0);
      }
    }
    store_rows_T<DH>(ov, dVg, a.dv_rs, dt, l31, half, Tk);
  }
  // ---- dK^T = Q^T dS.  Q^T replaces the dead dO^T tile; dS replaces P.
  __builtin_amdgcn_wave_barrier();
  stage_transposed<DH>(Qt, Qg, a.q_rs, Tq, lane);
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        union { bf16x8_t v; unsigned w[4]; } f;
        f.v = dsB[qt][2 * kt + (r >> 3)];
        DL[(kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * PLD + qt * 32 + l31] = (bf16_t)(f.w[(r & 7) >> 1] >> (16 * (r & 1)));
      }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) {
    f32x16_t ok[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ok[0][r] = 0.f; ok[1][r] = 0.f; }
    const int jj = (dt * 32 + l31 < DH) ? dt * 32 + l31 : DH - 1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bf16x8_t aq = frag_T_rows(Qt, u, half, jj);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const bf16x8_t bd = *reinterpret_cast<const bf16x8_t*>(DL + (kt * 32 + l31) * PLD + 16 * u + 8 * half);
        ok[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, bd, ok[kt], 0, 0, 0);
      }
    }
    store_rows_T<DH>(ok, dKg, a.dk_rs, dt, l31, half, Tk);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Coalesced variant of the backward kernel (dh % 16 == 0, 16-byte aligned rows): same math and register layouts as
// attn_bwd_mfma_kernel, but every global access is a 16-byte chunk of a row with consecutive lanes on consecutive chunks
// (see attn_fwd_co_kernel for the measurement behind this).  Wave-private LDS:
//   X  [64][dh+8] bf16   one raw row tile at a time: K, V (fragments -> registers), Q (phase A), then K again (K^T for dQ), dO
//                        (dO^T for dV), Q (Q^T for dK) -- transposed operands come from ds_read_b64_tr_b16, no register transposes
//   Y  = PD [64][72] + ST: holds the raw dO tile during phase A, afterwards P / dS as [key][q] (PD) and the 32-row bf16 output
//                        staging tile (ST) from which dQ / dV / dK leave as coalesced 16-byte stores
template <int DH> struct CoBwd {
  static constexpr int RS = DH + 8;
  static constexpr int PLD = 72;
  static constexpr int SLD = 40;                       // staging row stride (elements): [32 rows][32 dims]
  static constexpr int X_BYTES = 64 * RS * 2;
  static constexpr int PD_BYTES = 64 * PLD * 2;
  static constexpr int ST_BYTES = 32 * SLD * 2;
  // The raw dO tile of phase A aliases PD + ST when P's first 32 rows (written while dO rows 32.. are still needed) end below
  // dO row 32, i.e. PLD <= RS; otherwise (small head dims, LDS is not tight there) it gets its own region.
  static constexpr bool Y_ALIASES = (PLD <= RS) && (PD_BYTES + ST_BYTES >= X_BYTES);
  static constexpr int Y_OFFSET = Y_ALIASES ? X_BYTES : X_BYTES + PD_BYTES + ST_BYTES;
  static constexpr int BYTES = ((Y_ALIASES ? X_BYTES + PD_BYTES + ST_BYTES : Y_OFFSET + X_BYTES) + 64 + 255) / 256 * 256;
};

// one [32 rows][32 dims] accumulator half (rows = lane's output row, registers = dims) -> bf16 -> LDS rows -> 16-byte stores
template <int DH>
__device__ __forceinline__ void co_store_half(const f32x16_t& o, bf16_t* __restrict__ ST, bf16_t* __restrict__ base, long long rs, int dt,
                                              int t, int lane, int n_rows) {
  constexpr int SLD = CoBwd<DH>::SLD;
  const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    uint2 ov;
    ov.x = pack_bf16(o[4 * g + 0], o[4 * g + 1]);
    ov.y = pack_bf16(o[4 * g + 2], o[4 * g + 3]);
    *reinterpret_cast<uint2*>(ST + l31 * SLD + 8 * g + 4 * half) = ov;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int item = lane + 64 * i, rr = item >> 2, row = t * 32 + rr, j = dt * 32 + 8 * (item & 3);
    if (row < n_rows && j + 8 <= DH)
      *reinterpret_cast<uint4*>(base + (long long)row * rs + j) = *reinterpret_cast<const uint4*>(ST + rr * SLD + 8 * (item & 3));
  }
  __builtin_amdgcn_wave_barrier();
}

template <int DH, int NTQ, int NTK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_co_kernel(const AttnArgs a) {
  typedef CoBwd<DH> CB;
  constexpr int NK = (DH + 15) / 16;
  constexpr int NDT = (DH + 31) / 32;
  constexpr int CH = DH / 8;
  constexpr int RS = CB::RS, PLD = CB::PLD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  // The H heads of an example read 2 * dh-byte slices of the SAME rows (Q | K | V, dO) and write slices of the same dQ | dK | dV rows:
  // 160-byte pieces at 160-byte steps for dh = 80, i.e. every 128-byte line is shared by two heads.  Workgroup i runs on XCD i % 8 (each
  // XCD has its own L2): with (example, head) = (i / H, i % H) the heads of one example sat on H different XCDs and every shared line was
  // fetched -- and written back as a partial line -- twice (counters: 1.8 GB per launch against 0.92 GB algorithmic at B = 4096, T = 50).
  // Here the H heads of an example are H consecutive workgroups OF ONE XCD.
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const int bi = (int)((slot / (unsigned)a.H) * 8u + xcd), h = (int)(slot % (unsigned)a.H);
  if (bi >= (a.ex_list ? a.n_list : a.B)) return;
  const int b = a.ex_list ? a.ex_list[bi] : bi;
  // packed rows: the example has k_lens[b] rows, queries and keys alike (rows past them do not exist: in the dense layout they are
  // masked keys -- softmax weight exactly 0 -- and queries whose output gradient is zero, so both layouts give the same dQ / dK / dV on
  // the rows that exist); a.Tq / a.Tk stay the dense lengths for the dropout index
  int Tq = a.Tq, Tk = a.Tk;
  long long bq = (long long)b * a.q_bs, bk = (long long)b * a.k_bs, bv = (long long)b * a.v_bs, bdo = (long long)b * a.do_bs;
  long long bdq = (long long)b * a.dq_bs, bdk = (long long)b * a.dk_bs, bdv = (long long)b * a.dv_bs;
  if (a.row_off) {
    int n = a.k_lens[b];
    n = n < 1 ? 1 : (n > a.Tk ? a.Tk : n);
    Tq = n; Tk = n;
    const long long r0 = a.row_off[b];
    bq = r0 * a.q_rs; bk = r0 * a.k_rs; bv = r0 * a.v_rs; bdo = r0 * a.do_rs; bdq = r0 * a.dq_rs; bdk = r0 * a.dk_rs; bdv = r0 * a.dv_rs;
  }
  const int half = lane >> 5, l31 = lane & 31;
  bf16_t* X = reinterpret_cast<bf16_t*>(smem);
  bf16_t* PD = X + CB::X_BYTES / 2;                // P, then dS, as [q][key]
  bf16_t* ST = PD + CB::PD_BYTES / 2;              // output staging
  bf16_t* Y = X + CB::Y_OFFSET / 2;                // raw dO tile during phase A

  const bf16_t* Qg = reinterpret_cast<const bf16_t*>(a.Q) + bq + h * DH;
  const bf16_t* Kg = reinterpret_cast<const bf16_t*>(a.K) + bk + h * DH;
  const bf16_t* Vg = reinterpret_cast<const bf16_t*>(a.V) + bv + h * DH;
  const bf16_t* dOg = reinterpret_cast<const bf16_t*>(a.dout) + bdo + h * DH;

  // all four operand tiles are requested up front: ONE exposed global round trip instead of four
  uint4 g0[CoMap<DH>::NI], g1[CoMap<DH>::NI];
  {
    uint4 g2[CoMap<DH>::NI], g3[CoMap<DH>::NI];
    co_load<DH>(g0, Kg, a.k_rs, Tk, lane);
    co_load<DH>(g1, Vg, a.v_rs, Tk, lane);
    co_load<DH>(g2, Qg, a.q_rs, Tq, lane);
    co_load<DH>(g3, dOg, a.do_rs, Tq, lane);
    __builtin_amdgcn_sched_barrier(0);
    co_store<DH>(g0, X, RS, lane);                 // K rows
    co_store<DH>(g3, Y, RS, lane);                 // dO rows (stay for the whole of phase A)
#pragma unroll
    for (int i = 0; i < CoMap<DH>::NI; ++i) g0[i] = g2[i];    // Q rows wait in registers until K and V have passed through X
  }

  int klen = a.k_lens ? a.k_lens[b] : Tk;
  klen = klen < 0 ? 0 : (klen > Tk ? Tk : klen);
  const int qlen = a.q_lens ? a.q_lens[b] : Tq;
  const float kscale = LOG2E / sqrtf((float)DH), inv_sc = 1.0f / sqrtf((float)DH);
  const int kl = klen - 4 * half, tk = Tk - 4 * half;   // slot constant c: key = c + 4 half
  bf16x8_t dsB[NTQ][2 * NTK];     // dS  (B operand of dQ^T, later copied to LDS as [key][q])
  {
    bf16x8_t aK[NTK][NK], aV[NTK][NK];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < NTK; ++t)
#pragma unroll
      for (int s2 = 0; s2 < NK; ++s2) aK[t][s2] = co_frag<DH>(X, t * 32 + l31, s2 * 16 + 8 * half);
    __builtin_amdgcn_wave_barrier();
    co_store<DH>(g1, X, RS, lane);                 // V rows
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < NTK; ++t)
#pragma unroll
      for (int s2 = 0; s2 < NK; ++s2) aV[t][s2] = co_frag<DH>(X, t * 32 + l31, s2 * 16 + 8 * half);
    __builtin_amdgcn_wave_barrier();
    co_store<DH>(g0, X, RS, lane);                 // Q rows
    __builtin_amdgcn_wave_barrier();

    // ---- phase A, one query tile (32 queries x 64 keys) at a time
#pragma unroll
    for (int qt = 0; qt < NTQ; ++qt) {
      const int q = qt * 32 + l31;
      f32x16_t acc[NTK], dp[NTK];
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[kt][r] = 0.f; dp[kt][r] = 0.f; }
#pragma unroll
      for (int s2 = 0; s2 < NK; ++s2) {
        const bf16x8_t bQ = co_frag<DH>(X, q, s2 * 16 + 8 * half), bD = co_frag<DH>(Y, q, s2 * 16 + 8 * half);
#pragma unroll
        for (int kt = 0; kt < NTK; ++kt) {
          acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aK[kt][s2], bQ, acc[kt], 0, 0, 0);   // S^T = K Q^T
          dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aV[kt][s2], bD, dp[kt], 0, 0, 0);     // dP^T = V dO^T
        }
      }
      if (qt == NTQ - 1) {       // K / V fragments are dead: their registers take the K rows that phase B transposes
        co_load<DH>(g0, Kg, a.k_rs, Tk, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
      float m = -3.0e38f;
      // (opaque copies: hipcc would otherwise hoist the 2 x 32 x NTQ lane masks of these compares out of the query-tile loop and
      //  keep them in SGPR pairs -- 153 spilled SGPRs)
      int klv = kl, tkv = tk;
      asm volatile("" : "+v"(klv), "+v"(tkv));
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
          float x = acc[kt][r] * kscale;
          x = (c >= klv) ? PADDING_NUM * LOG2E : x;
          x = (c >= tkv) ? -3.0e38f : x;
          acc[kt][r] = x;
          m = fmaxf(m, x);
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
      const unsigned keep = drop_bits<NTK>(a, b, h, q, half);
      float dot = 0.f;
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = acc[kt][r] * inv_sum;
          acc[kt][r] = pv;
          float gq = dp[kt][r];
          if (a.drop_on) gq = ((keep >> (kt * 16 + r)) & 1u) ? gq * a.drop_inv : 0.f;   // gradient w.r.t. the pre-dropout weights
          dp[kt][r] = gq;
          dot += pv * gq;
        }
      dot += __shfl_xor(dot, 32, 64);
      const bool qpad = (q >= qlen);
      asm volatile("" : "+v"(klv), "+v"(tkv));
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = kt * 32 + (r & 3) + 8 * (r >> 2);
          float pv = acc[kt][r];
          float ds = (c < klv) ? pv * (dp[kt][r] - dot) * inv_sc : 0.f;     // no gradient into masked keys
          if (qpad) { ds = 0.f; pv = (c < tkv) ? PADDING_NUM : 0.f; }       // constant rows: gradient reaches V only
          if (a.drop_on) pv = ((keep >> (kt * 16 + r)) & 1u) ? pv * a.drop_inv : 0.f;   // dropped weights feed dV
          dp[kt][r] = ds;
          acc[kt][r] = pv;
        }
#pragma unroll
      for (int u = 0; u < 2 * NTK; ++u) {
        union { bf16x8_t v; unsigned w[4]; } f;
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          const int r0 = 8 * (u & 1) + i;
          f.w[i >> 1] = pack_bf16(dp[u >> 1][r0], dp[u >> 1][r0 + 1]);
        }
        dsB[qt][u] = f.v;
      }
      // P as it feeds dV (query mask and dropout applied) goes straight to PD row q: 4 consecutive keys per 8-byte store
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 pv;
          pv.x = pack_bf16(acc[kt][4 * g + 0], acc[kt][4 * g + 1]);
          pv.y = pack_bf16(acc[kt][4 * g + 2], acc[kt][4 * g + 3]);
          *reinterpret_cast<uint2*>(PD + q * PLD + kt * 32 + 8 * g + 4 * half) = pv;
        }
      __builtin_amdgcn_sched_barrier(0);           // one query tile at a time: interleaving both doubles the live accumulators
    }
  }
  __builtin_amdgcn_wave_barrier();

  typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
  typedef __attribute__((address_space(3))) bf16x4_t* lds_p4;
  const int i16 = lane & 15, jgrp = (lane >> 4) & 1;
  // A = X^T fragments through transposing reads.  slot order: k-slot i of lane-half h <-> row 16u + (i&3) + 8(i>>2) + 4h;
  // natural order: k-slot i <-> row 16u + 8h + i
  const bf16_t* xs = X + (4 * half + (i16 >> 2)) * RS + 16 * jgrp + 4 * (i16 & 3);
  const bf16_t* xn = X + (8 * half + (i16 >> 2)) * RS + 16 * jgrp + 4 * (i16 & 3);
  auto fragT = [&](const bf16_t* base, int u, int dt, int hi_rows) -> bf16x8_t {
    const bf16_t* vp = base + 16 * u * RS + dt * 32;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p4)(vp));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p4)(vp + hi_rows * RS));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // B fragment of PD^T: lane n = key kt*32 + l31, k-slot i <-> query 16u + 8h + i
  const bf16_t* pn = PD + (8 * half + (i16 >> 2)) * PLD + 16 * jgrp + 4 * (i16 & 3);
  auto fragPD = [&](int u, int kt) -> bf16x8_t {
    const bf16_t* vp = pn + 16 * u * PLD + kt * 32;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p4)(vp));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p4)(vp + 4 * PLD));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  // ---- dQ^T = K^T dS^T
  bf16_t* dQg = reinterpret_cast<bf16_t*>(a.dQ) + bdq + h * DH;
  bf16_t* dKg = reinterpret_cast<bf16_t*>(a.dK) + bdk + h * DH;
  bf16_t* dVg = reinterpret_cast<bf16_t*>(a.dV) + bdv + h * DH;
  co_store<DH>(g0, X, RS, lane);
  co_load<DH>(g1, dOg, a.do_rs, Tq, lane);         // for phase C, in flight during phase B
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) {
    f32x16_t o[NTQ];
#pragma unroll
    for (int qt = 0; qt < NTQ; ++qt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qt][r] = 0.f;
#pragma unroll
    for (int u = 0; u < 2 * NTK; ++u) {
      const bf16x8_t av = fragT(xs, u, dt, 8);
#pragma unroll
      for (int qt = 0; qt < NTQ; ++qt) o[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, dsB[qt][u], o[qt], 0, 0, 0);
    }
#pragma unroll
    for (int qt = 0; qt < NTQ; ++qt) co_store_half<DH>(o[qt], ST, dQg, a.dq_rs, dt, qt, lane, Tq);
  }

  // ---- dV^T = dO^T P   (reduction over queries, natural k slots).  dO replaces the dead K tile; P goes to LDS as [key][q].
  __builtin_amdgcn_wave_barrier();
  co_load<DH>(g0, Qg, a.q_rs, Tq, lane);
  co_store<DH>(g1, X, RS, lane);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) {
    f32x16_t ov[NTK];
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) ov[kt][r] = 0.f;
#pragma unroll
    for (int u = 0; u < 2 * NTQ; ++u) {
      const bf16x8_t ad = fragT(xn, u, dt, 4);
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt) {
        const bf16x8_t bp = fragPD(u, kt);
        ov[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, bp, ov[kt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt) co_store_half<DH>(ov[kt], ST, dVg, a.dv_rs, dt, kt, lane, Tk);
  }

  // ---- dK^T = Q^T dS.  Q replaces the dead dO tile; dS replaces P.
  __builtin_amdgcn_wave_barrier();
  co_store<DH>(g0, X, RS, lane);
#pragma unroll
  for (int qt = 0; qt < NTQ; ++qt)
#pragma unroll
    for (int u = 0; u < 2 * NTK; ++u) {      // slots 0..3 of dsB[qt][u] <-> keys 16u + 4h + {0..3}, slots 4..7 <-> 8 keys further
      union { bf16x8_t v; uint2 h2[2]; } f;
      f.v = dsB[qt][u];
      *reinterpret_cast<uint2*>(PD + (qt * 32 + l31) * PLD + 16 * u + 4 * half) = f.h2[0];
      *reinterpret_cast<uint2*>(PD + (qt * 32 + l31) * PLD + 16 * u + 8 + 4 * half) = f.h2[1];
    }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) {
    f32x16_t ok[NTK];
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) ok[kt][r] = 0.f;
#pragma unroll
    for (int u = 0; u < 2 * NTQ; ++u) {
      const bf16x8_t aq = fragT(xn, u, dt, 4);
#pragma unroll
      for (int kt = 0; kt < NTK; ++kt) {
        const bf16x8_t bd = fragPD(u, kt);
        ok[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, bd, ok[kt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt) co_store_half<DH>(ok[kt], ST, dKg, a.dk_rs, dt, kt, lane, Tk);
  }
}

// ------------------------------------------------------------------------------------------------------------
// Single-query attention (the decoder: the target item attends over the encoded sequence, Tq == 1).  Memory bound:
// one wavefront per (example, head), lane k owns key k and streams its K / V rows straight from global in 4-element
// chunks; q / dO are wave-uniform (broadcast) loads; P.V and dQ run with lanes along the head dim over coalesced rows.
template <typename T> struct Chunk4;
template <> struct Chunk4<float> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) { const float4 x = *reinterpret_cast<const float4*>(p); v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Chunk4<bf16_t> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[4]) {
    const uint2 x = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(x.x << 16); v[1] = __uint_as_float(x.x & 0xFFFF0000u); v[2] = __uint_as_float(x.y << 16); v[3] = __uint_as_float(x.y & 0xFFFF0000u);
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[4]) {
    uint2 o; o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16); o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = o;
  }
};

template <typename T, int DH, bool BWD>
__global__ __launch_bounds__(256) void attn_q1_kernel(const AttnArgs a) {
  __shared__ float s_buf[4][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long long wid = (long long)blockIdx.x * 4 + wave;
  if (wid >= (long long)a.B * a.H) return;
  const int b = (int)(wid / a.H), h = (int)(wid % a.H);
  const int Tk = a.Tk;
  int klen = a.k_lens ? a.k_lens[b] : Tk;
  klen = klen < 0 ? 0 : (klen > Tk ? Tk : klen);
  const int qlen = a.q_lens ? a.q_lens[b] : 1;
  const float sc = sqrtf((float)DH);
  const T* Qg = reinterpret_cast<const T*>(a.Q) + (long long)b * a.q_bs + h * DH;
  const T* Kg = reinterpret_cast<const T*>(a.K) + (long long)b * a.k_bs + h * DH;
  const T* Vg = reinterpret_cast<const T*>(a.V) + (long long)b * a.v_bs + h * DH;
  const T* Kr = Kg + (long long)lane * a.k_rs;
  const T* Vr = Vg + (long long)lane * a.v_rs;
  const T* dOg = BWD ? reinterpret_cast<const T*>(a.dout) + (long long)b * a.do_bs + h * DH : nullptr;

  float s = 0.f, dP = 0.f;
  if (lane < Tk) {
#pragma unroll 5
    for (int c = 0; c < DH; c += 4) {
      float kq[4], qq[4];
      Chunk4<T>::ld(Kr + c, kq);
      Chunk4<T>::ld(Qg + c, qq);
      s = fmaf(qq[0], kq[0], s); s = fmaf(qq[1], kq[1], s); s = fmaf(qq[2], kq[2], s); s = fmaf(qq[3], kq[3], s);
      if constexpr (BWD) {
        float vv[4], dd[4];
        Chunk4<T>::ld(Vr + c, vv);
        Chunk4<T>::ld(dOg + c, dd);
        dP = fmaf(dd[0], vv[0], dP); dP = fmaf(dd[1], vv[1], dP); dP = fmaf(dd[2], vv[2], dP); dP = fmaf(dd[3], vv[3], dP);
      }
    }
    s = s / sc;
    if (lane >= klen) s = PADDING_NUM;
  }
  const float m = wave_max(lane < Tk ? s : -3.0e38f);
  const float e = (lane < Tk) ? expf(s - m) : 0.f;
  const float sum = wave_sum(e);
  float p = e / sum;
  const bool qpad = (0 >= qlen);
  const float Dk = drop_factor(a, b, h, 0, lane);
  if constexpr (!BWD) {
    if (qpad) p = PADDING_NUM;
    p *= Dk;
    if (lane < Tk) s_buf[wave][lane] = p;
    __builtin_amdgcn_wave_barrier();
    const T* Rg = a.resid ? reinterpret_cast<const T*>(a.resid) + (long long)b * a.r_bs + h * DH : nullptr;
    T* Og = reinterpret_cast<T*>(a.out) + (long long)b * a.o_bs + h * DH;
    for (int j = lane; j < DH; j += 64) {
      float o = 0.f;
#pragma unroll 5
      for (int k = 0; k < Tk; ++k) o = fmaf(s_buf[wave][k], ldf<T>(Vg + (long long)k * a.v_rs + j), o);
      if (Rg) o += ldf<T>(Rg + j);
      stf<T>(Og + j, o);
    }
  } else {
    float ds = 0.f;
    dP *= Dk;
    if (!qpad) {
      const float dot = wave_sum(lane < Tk ? p * dP : 0.f);
      ds = (lane < klen) ? p * (dP - dot) / sc : 0.f;
    } else {
      p = PADDING_NUM;
    }
    p *= Dk;
    if (lane < Tk) {
      s_buf[wave][lane] = ds;
      T* dKr = reinterpret_cast<T*>(a.dK) + (long long)b * a.dk_bs + (long long)lane * a.dk_rs + h * DH;
      T* dVr = reinterpret_cast<T*>(a.dV) + (long long)b * a.dv_bs + (long long)lane * a.dv_rs + h * DH;
#pragma unroll 5
      for (int c = 0; c < DH; c += 4) {
        float qq[4], dd[4], ok[4], ov[4];
        Chunk4<T>::ld(Qg + c, qq);
        Chunk4<T>::ld(dOg + c, dd);
#pragma unroll
        for (int i = 0; i < 4; ++i) { ok[i] = ds * qq[i]; ov[i] = p * dd[i]; }
        Chunk4<T>::st(dKr + c, ok);
        Chunk4<T>::st(dVr + c, ov);
      }
    }
    __builtin_amdgcn_wave_barrier();
    T* dQg = reinterpret_cast<T*>(a.dQ) + (long long)b * a.dq_bs + h * DH;
    for (int j = lane; j < DH; j += 64) {
      float g = 0.f;
#pragma unroll 5
      for (int k = 0; k < Tk; ++k) g = fmaf(s_buf[wave][k], ldf<T>(Kg + (long long)k * a.k_rs + j), g);
      stf<T>(dQg + j, g);
    }
  }
}

// ---- bf16 single-query attention with lanes ALONG the head dim: 16 lanes x 16 bytes cover one key row (dh <= 128), four
// rows per wave instruction, so every K / V (and dK / dV) row moves in coalesced 16-byte pieces and is read exactly ONCE:
// the rows stay in registers between the score pass and the P.V / dQ pass (Tk <= 64: 16 iterations of 4 rows).
//   s[k]  : 8-term partial dot per lane, reduced over the 16 lanes of the row group
//   P.V   : lane accumulates its 8 columns over its rows; the 4 row groups are summed at the end
// Same arithmetic as attn_q1_kernel (IEEE division, expf): it is memory bound either way.
__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xFFFF0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xFFFF0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xFFFF0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xFFFF0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack_bf16(f[0], f[1]), pack_bf16(f[2], f[3]), pack_bf16(f[4], f[5]), pack_bf16(f[6], f[7]));
}
__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}
__device__ __forceinline__ float cross_group_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float cross_group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }

template <int DH, bool BWD, int NIT>      // NIT: 4-row passes, Tk <= 4 * NIT (the row registers are sized by it)
__global__ __launch_bounds__(256) void attn_q1v_kernel(const AttnArgs a) {
  constexpr int CPR = DH / 8;      // 16-byte chunks per row
  const int lane = threadIdx.x & 63;
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (long long)a.B * a.H) return;
  const int b = (int)(wid / a.H), h = (int)(wid % a.H);
  const int g = lane >> 4, c = lane & 15;
  const bool cact = c < CPR;
  const int Tk = a.Tk;
  int klen = a.k_lens ? a.k_lens[b] : Tk;
  klen = klen < 0 ? 0 : (klen > Tk ? Tk : klen);
  const int qlen = a.q_lens ? a.q_lens[b] : 1;
  const float sc = sqrtf((float)DH);
  const bf16_t* Qg = reinterpret_cast<const bf16_t*>(a.Q) + (long long)b * a.q_bs + h * DH + c * 8;
  const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
  // All K / V row requests go out back to back and BRANCH-FREE: a predicated load is waited for before the next one is
  // issued (the first version paid 16 serial round trips per wave).  Lanes past the row (c >= CPR) and rows past Tk re-read a
  // valid chunk / the last row instead; what they load is kept out of every sum below.
  const int cc = cact ? c : CPR - 1;
  const bf16_t* Kg = reinterpret_cast<const bf16_t*>(a.K) + (long long)b * a.k_bs + h * DH + cc * 8;
  const bf16_t* Vg = reinterpret_cast<const bf16_t*>(a.V) + (long long)b * a.v_bs + h * DH + cc * 8;
  uint4 Kr[NIT], Vr[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int k = it * 4 + g, kc = k < Tk ? k : Tk - 1;
    Kr[it] = *reinterpret_cast<const uint4*>(Kg + (long long)kc * a.k_rs);
    Vr[it] = *reinterpret_cast<const uint4*>(Vg + (long long)kc * a.v_rs);
  }
  float q[8], dO[8];
  {
    const uint4 qu = cact ? *reinterpret_cast<const uint4*>(Qg) : z4;
    unpack8(qu, q);
    if constexpr (BWD) {
      const bf16_t* dOg = reinterpret_cast<const bf16_t*>(a.dout) + (long long)b * a.do_bs + h * DH + c * 8;
      const uint4 du = cact ? *reinterpret_cast<const uint4*>(dOg) : z4;
      unpack8(du, dO);
    }
  }
  float s[NIT], dP[NIT];
  float m = -3.0e38f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    s[it] = -3.0e38f; dP[it] = 0.f;
    if (it * 4 < Tk) {     // wave-uniform
      float kf[8];
      unpack8(Kr[it], kf);
      float d0 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) d0 = fmaf(q[e], kf[e], d0);
      d0 = group16_sum(cact ? d0 : 0.f);
      const int k = it * 4 + g;
      float x = d0 / sc;
      if (k >= klen) x = PADDING_NUM;
      if (k >= Tk) x = -3.0e38f;
      s[it] = x;
      m = fmaxf(m, x);
      if constexpr (BWD) {
        float vf[8];
        unpack8(Vr[it], vf);
        float d1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d1 = fmaf(dO[e], vf[e], d1);
        dP[it] = group16_sum(cact ? d1 : 0.f);
      }
    }
  }
  m = cross_group_max(m);
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int k = it * 4 + g;
    const float e = (it * 4 < Tk && k < Tk) ? expf(s[it] - m) : 0.f;
    s[it] = e;
    sum += e;
  }
  sum = cross_group_sum(sum);       // every lane of a group carries the same e: one copy per row, four groups
  const bool qpad = (0 >= qlen);

  if constexpr (!BWD) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (it * 4 < Tk) {
        const int k = it * 4 + g;
        float pk = s[it] / sum;
        if (qpad) pk = PADDING_NUM;                 // query mask applied AFTER the softmax (reference behaviour, F13)
        pk *= drop_factor(a, b, h, 0, k);
        if (k >= Tk) pk = 0.f;
        float vf[8];
        unpack8(Vr[it], vf);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(pk, vf[e], o[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = cross_group_sum(o[e]);
    if (g == 0 && cact) {
      if (a.resid) {
        float rf[8];
        unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.resid) + (long long)b * a.r_bs + h * DH + c * 8), rf);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += rf[e];
      }
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.out) + (long long)b * a.o_bs + h * DH + c * 8) = pack8(o);
    }
  } else {
    // dot = sum_k p[k] dP[k] (after the dropout factor), ds, then dK / dV rows and dQ
    float pd[NIT];
    float dot = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int k = it * 4 + g;
      const float Dk = (it * 4 < Tk) ? drop_factor(a, b, h, 0, k) : 0.f;
      const float pk = s[it] / sum;
      dP[it] *= Dk;                                  // gradient w.r.t. the pre-dropout weights
      pd[it] = Dk;
      s[it] = pk;
      if (it * 4 < Tk && k < Tk) dot += pk * dP[it];
    }
    dot = cross_group_sum(dot);
    float dq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dq[e] = 0.f;
    bf16_t* dKg = reinterpret_cast<bf16_t*>(a.dK) + (long long)b * a.dk_bs + h * DH + c * 8;
    bf16_t* dVg = reinterpret_cast<bf16_t*>(a.dV) + (long long)b * a.dv_bs + h * DH + c * 8;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (it * 4 < Tk) {
        const int k = it * 4 + g;
        float pk = s[it], ds = 0.f;
        if (!qpad) ds = (k < klen) ? pk * (dP[it] - dot) / sc : 0.f;   // no gradient into masked keys
        else pk = PADDING_NUM;                                          // constant rows: gradient reaches V only
        pk *= pd[it];                                                   // dropped weights feed dV
        if (cact && k < Tk) {
          float ok[8], ov[8], kf[8];
          unpack8(Kr[it], kf);
#pragma unroll
          for (int e = 0; e < 8; ++e) { ok[e] = ds * q[e]; ov[e] = pk * dO[e]; dq[e] = fmaf(ds, kf[e], dq[e]); }
          *reinterpret_cast<uint4*>(dKg + (long long)k * a.dk_rs) = pack8(ok);
          *reinterpret_cast<uint4*>(dVg + (long long)k * a.dv_rs) = pack8(ov);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) dq[e] = cross_group_sum(dq[e]);
    if (g == 0 && cact) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.dQ) + (long long)b * a.dq_bs + h * DH + c * 8) = pack8(dq);
  }
}

template <bool BWD>
int launch_q1v(const AttnArgs& a, hipStream_t st) {
  const unsigned nb = (unsigned)cdiv64((long long)a.B * a.H, 4);
  if (a.Tk < 1 || a.Tk > 64) return -1;
#define DMT_Q1V(DHV) do { \
    if (a.Tk <= 12) hipLaunchKernelGGL((attn_q1v_kernel<DHV, BWD, 3>), dim3(nb), dim3(256), 0, st, a); \
    else if (a.Tk <= 32) hipLaunchKernelGGL((attn_q1v_kernel<DHV, BWD, 8>), dim3(nb), dim3(256), 0, st, a); \
    else if (a.Tk <= 52) hipLaunchKernelGGL((attn_q1v_kernel<DHV, BWD, 13>), dim3(nb), dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((attn_q1v_kernel<DHV, BWD, 16>), dim3(nb), dim3(256), 0, st, a); } while (0)
  switch (a.dh) {
    case 16: DMT_Q1V(16); break;
    case 32: DMT_Q1V(32); break;
    case 64: DMT_Q1V(64); break;
    case 80: DMT_Q1V(80); break;
    default: return -1;
  }
#undef DMT_Q1V
  return 0;
}

// 16-byte pieces: base, batch and row strides 16-byte aligned
static bool q1v_aligned(const void* q, long long s0, long long s1) {
  return q == nullptr || (((uintptr_t)q) % 16 == 0 && s0 % 8 == 0 && s1 % 8 == 0);
}

template <typename T, bool BWD>
int launch_q1(const AttnArgs& a, hipStream_t st) {
  const unsigned nb = (unsigned)cdiv64((long long)a.B * a.H, 4);
  switch (a.dh) {
    case 16: hipLaunchKernelGGL((attn_q1_kernel<T, 16, BWD>), dim3(nb), dim3(256), 0, st, a); break;
    case 20: hipLaunchKernelGGL((attn_q1_kernel<T, 20, BWD>), dim3(nb), dim3(256), 0, st, a); break;
    case 32: hipLaunchKernelGGL((attn_q1_kernel<T, 32, BWD>), dim3(nb), dim3(256), 0, st, a); break;
    case 64: hipLaunchKernelGGL((attn_q1_kernel<T, 64, BWD>), dim3(nb), dim3(256), 0, st, a); break;
    case 80: hipLaunchKernelGGL((attn_q1_kernel<T, 80, BWD>), dim3(nb), dim3(256), 0, st, a); break;
    default: return -1;
  }
  return 0;
}

// 4-element chunks: 8-byte (bf16) / 16-byte (fp32) aligned rows
static bool q1_aligned(int dtype, const void* q, long long s0, long long s1, int dh) {
  const int esz = dtype == DMT_F32 ? 4 : 2;
  return q == nullptr || (((uintptr_t)q) % (4 * esz) == 0 && s0 % 4 == 0 && s1 % 4 == 0 && dh % 4 == 0);
}

int fill_args(AttnArgs& a, const dmt_attn_desc* d) {
  a.B = d->B; a.H = d->H; a.dh = d->dh; a.Tq = d->Tq; a.Tk = d->Tk;
  a.Q = d->Q; a.q_bs = d->q_bs; a.q_rs = d->q_rs;
  a.K = d->K; a.k_bs = d->k_bs; a.k_rs = d->k_rs;
  a.V = d->V; a.v_bs = d->v_bs; a.v_rs = d->v_rs;
  a.q_lens = d->q_lens; a.k_lens = d->k_lens;
  a.resid = d->resid; a.r_bs = d->r_bs; a.r_rs = d->r_rs;
  a.out = d->out; a.o_bs = d->o_bs; a.o_rs = d->o_rs;
  a.drop_on = (d->drop_keep > 0.f && d->drop_keep < 1.f) ? 1 : 0;
  a.drop_seed = d->drop_seed;
  a.drop_thr = a.drop_on ? (unsigned)(d->drop_keep * 16777216.0f) : 0u;
  a.drop_inv = a.drop_on ? 1.f / d->drop_keep : 1.f;
  a.dout = nullptr; a.dQ = a.dK = a.dV = nullptr;
  a.vec16 = 0;
  a.do_bs = a.do_rs = a.dq_bs = a.dq_rs = a.dk_bs = a.dk_rs = a.dv_bs = a.dv_rs = 0;
  a.row_off = nullptr; a.ex_list = nullptr; a.n_list = 0;      // (packed rows: set by dmt_attn_bwd on the one route that takes them)
  return 0;
}

int check_desc(const dmt_attn_desc* d, const char* who) {
  DMT_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  DMT_CHECK_ARG(d->dtype == DMT_F32 || d->dtype == DMT_BF16, "%s: bad dtype", who);
  DMT_CHECK_ARG(d->B > 0 && d->H > 0 && d->dh > 0 && d->Tq > 0 && d->Tk > 0, "%s: bad dims", who);
  DMT_CHECK_ARG(d->Q && d->K && d->V, "%s: null Q/K/V", who);
  if (d->Tk > 64 || d->Tq > 64) {
    dmt_set_error("%s: Tq=%d, Tk=%d: sequences over 64 need bf16, dh in 16/32/64/80 and T <= 256 (dmt_attn_long_*)", who, d->Tq, d->Tk);
    return DMT_ERR_UNSUPPORTED;
  }
  return DMT_OK;
}

template <typename T>
int launch_fwd(const AttnArgs& a, int nw, size_t lds, hipStream_t st) {
  const unsigned nb = (unsigned)cdiv64((long long)a.B * a.H, nw);
  if (a.dh == 20) hipLaunchKernelGGL((attn_fwd_kernel<T, 20>), dim3(nb), dim3(nw * 64), lds, st, a);
  else if (a.dh == 80) hipLaunchKernelGGL((attn_fwd_kernel<T, 80>), dim3(nb), dim3(nw * 64), lds, st, a);
  else hipLaunchKernelGGL((attn_fwd_kernel<T, 0>), dim3(nb), dim3(nw * 64), lds, st, a);
  return 0;
}

template <typename T>
int launch_bwd(const AttnArgs& a, int nw, size_t lds, hipStream_t st) {
  const unsigned nb = (unsigned)cdiv64((long long)a.B * a.H, nw);
  switch (a.dh) {
    case 4: hipLaunchKernelGGL((attn_bwd_kernel<T, 4>), dim3(nb), dim3(nw * 64), lds, st, a); break;
    case 8: hipLaunchKernelGGL((attn_bwd_kernel<T, 8>), dim3(nb), dim3(nw * 64), lds, st, a); break;
    case 16: hipLaunchKernelGGL((attn_bwd_kernel<T, 16>), dim3(nb), dim3(nw * 64), lds, st, a); break;
    case 20: hipLaunchKernelGGL((attn_bwd_kernel<T, 20>), dim3(nb), dim3(nw * 64), lds, st, a); break;
    case 32: hipLaunchKernelGGL((attn_bwd_kernel<T, 32>), dim3(nb), dim3(nw * 64), lds, st, a); break;
    case 64: hipLaunchKernelGGL((attn_bwd_kernel<T, 64>), dim3(nb), dim3(nw * 64), lds, st, a); break;
    case 80: hipLaunchKernelGGL((attn_bwd_kernel<T, 80>), dim3(nb), dim3(nw * 64), lds, st, a); break;
    default: return -1;
  }
  return 0;
}

}  // namespace

extern "C" int dmt_attn_fwd(const dmt_attn_desc* d, void* stream) {
  if (d && (d->Tq > 64 || d->Tk > 64) && dmt_attn_long_supported(d->dtype, d->dh, d->Tq, d->Tk)) return dmt_attn_long_fwd(d, stream);
  int rc = check_desc(d, "dmt_attn_fwd");
  if (rc != DMT_OK) return rc;
  DMT_CHECK_ARG(d->out != nullptr, "dmt_attn_fwd: null out");
  AttnArgs a;
  fill_args(a, d);
  const int per_wave = 2 * d->Tk * (d->dh + 1) + d->dh + d->Tk + 8;
  a.lds_per_wave = (per_wave + 3) & ~3;
  int nw = 4;
  while (nw > 1 && (size_t)nw * a.lds_per_wave * 4 > 64 * 1024) nw >>= 1;
  const size_t lds = (size_t)nw * a.lds_per_wave * 4;
  DMT_CHECK_ARG(lds <= 160 * 1024, "dmt_attn_fwd: Tk*dh too large for LDS");
  hipStream_t st = (hipStream_t)stream;
  if (d->Tq == 1 && q1_aligned(d->dtype, d->Q, d->q_bs, d->q_rs, d->dh) && q1_aligned(d->dtype, d->K, d->k_bs, d->k_rs, d->dh) &&
      q1_aligned(d->dtype, d->V, d->v_bs, d->v_rs, d->dh)) {
    // bf16 with 16-byte aligned rows: the lanes-along-dh kernel (coalesced row chunks, all K / V requests in flight at once:
    // 73 us forward / 116 us backward at B=4096, T=50 against 65-85 / 240 us before its loads were made branch-free).  Routing
    // Tq = 1 through the coalesced MFMA kernels (NTQ = 1, NTK = 2) measured no better, so the simple kernels stay.
    if (d->dtype == DMT_BF16 && d->Tk <= 64 && q1v_aligned(d->Q, d->q_bs, d->q_rs) && q1v_aligned(d->K, d->k_bs, d->k_rs) &&
        q1v_aligned(d->V, d->v_bs, d->v_rs) && q1v_aligned(d->out, d->o_bs, d->o_rs) && q1v_aligned(d->resid, d->r_bs, d->r_rs) &&
        launch_q1v<false>(a, st) == 0) {
      DMT_CHECK_LAUNCH("dmt_attn_fwd(q1v)");
      return DMT_OK;
    }
    const int r1 = (d->dtype == DMT_F32) ? launch_q1<float, false>(a, st) : launch_q1<bf16_t, false>(a, st);
    if (r1 == 0) { DMT_CHECK_LAUNCH("dmt_attn_fwd(q1)"); return DMT_OK; }
  }
  // bf16 MFMA path: 8-byte aligned rows (all strides multiples of 4 elements) and an instantiated head dim
  auto al8 = [](const void* q, long long s0, long long s1) { return q == nullptr || (((uintptr_t)q) % 8 == 0 && s0 % 4 == 0 && s1 % 4 == 0); };
  const bool mfma_ok = d->dtype == DMT_BF16 && d->Tq >= 8 && (d->dh == 20 || d->dh == 80 || d->dh == 16 || d->dh == 32 || d->dh == 64) &&
                       al8(d->Q, d->q_bs, d->q_rs) && al8(d->K, d->k_bs, d->k_rs) && al8(d->V, d->v_bs, d->v_rs) &&
                       al8(d->resid, d->r_bs, d->r_rs) && al8(d->out, d->o_bs, d->o_rs);
  if (mfma_ok) {
    a.vec16 = (d->dh % 8 == 0 && q1v_aligned(d->Q, d->q_bs, d->q_rs) && q1v_aligned(d->K, d->k_bs, d->k_rs) && q1v_aligned(d->V, d->v_bs, d->v_rs)) ? 1 : 0;
    const int nwm = 4;
    const unsigned nb = (unsigned)cdiv64((long long)d->B * d->H, nwm);
    // all-coalesced variant: 16-byte aligned rows everywhere (incl. residual and output)
    if (a.vec16 && d->dh % 16 == 0 && q1v_aligned(d->out, d->o_bs, d->o_rs) && (d->resid == nullptr || q1v_aligned(d->resid, d->r_bs, d->r_rs))) {
      const int ntq = d->Tq <= 32 ? 1 : 2, ntk = d->Tk <= 32 ? 1 : 2;       // 32-row tiles actually needed
#define DMT_FWD_CO(DHV) do { const size_t lb = (size_t)nwm * CoTile<DHV>::BYTES; \
    if (ntq == 1 && ntk == 1) hipLaunchKernelGGL((attn_fwd_co_kernel<DHV, 1, 1>), dim3(nb), dim3(nwm * 64), lb, st, a); \
    else if (ntq == 1) hipLaunchKernelGGL((attn_fwd_co_kernel<DHV, 1, 2>), dim3(nb), dim3(nwm * 64), lb, st, a); \
    else hipLaunchKernelGGL((attn_fwd_co_kernel<DHV, 2, 2>), dim3(nb), dim3(nwm * 64), lb, st, a); } while (0)
      switch (d->dh) {
        case 16: DMT_FWD_CO(16); break;
        case 32: DMT_FWD_CO(32); break;
        case 64: DMT_FWD_CO(64); break;
        default: DMT_FWD_CO(80); break;
      }
#undef DMT_FWD_CO
      DMT_CHECK_LAUNCH("dmt_attn_fwd(mfma, coalesced)");
      return DMT_OK;
    }
    const size_t ldsm = (size_t)nwm * d->dh * 72 * 2;
    switch (d->dh) {
      case 16: hipLaunchKernelGGL((attn_fwd_mfma_kernel<16>), dim3(nb), dim3(nwm * 64), ldsm, st, a); break;
      case 20: hipLaunchKernelGGL((attn_fwd_mfma_kernel<20>), dim3(nb), dim3(nwm * 64), ldsm, st, a); break;
      case 32: hipLaunchKernelGGL((attn_fwd_mfma_kernel<32>), dim3(nb), dim3(nwm * 64), ldsm, st, a); break;
      case 64: hipLaunchKernelGGL((attn_fwd_mfma_kernel<64>), dim3(nb), dim3(nwm * 64), ldsm, st, a); break;
      default: hipLaunchKernelGGL((attn_fwd_mfma_kernel<80>), dim3(nb), dim3(nwm * 64), ldsm, st, a); break;
    }
    DMT_CHECK_LAUNCH("dmt_attn_fwd(mfma)");
    return DMT_OK;
  }
  if (d->dtype == DMT_F32) launch_fwd<float>(a, nw, lds, st); else launch_fwd<bf16_t>(a, nw, lds, st);
  DMT_CHECK_LAUNCH("dmt_attn_fwd");
  return DMT_OK;
}

extern "C" int dmt_attn_bwd(const dmt_attn_bwd_desc* d, void* stream) {
  DMT_CHECK_ARG(d != nullptr, "dmt_attn_bwd: null descriptor");
  if ((d->f.Tq > 64 || d->f.Tk > 64) && dmt_attn_long_supported(d->f.dtype, d->f.dh, d->f.Tq, d->f.Tk)) return dmt_attn_long_bwd(d, stream);
  int rc = check_desc(&d->f, "dmt_attn_bwd");
  if (rc != DMT_OK) return rc;
  DMT_CHECK_ARG(d->dout && d->dQ && d->dK && d->dV, "dmt_attn_bwd: null gradient buffer");
  const bool packed = d->f.row_off != nullptr || d->f.ex_list != nullptr;
  if (packed) {
    DMT_CHECK_ARG(d->f.dtype == DMT_BF16 && d->f.Tq == d->f.Tk && d->f.Tq > 1 && d->f.k_lens != nullptr && (d->f.ex_list == nullptr || d->f.n_list > 0),
                  "dmt_attn_bwd: packed rows / example lists are taken by the bf16 self-attention form only (Tq == Tk > 1, k_lens given)");
  }
  AttnArgs a;
  fill_args(a, &d->f);
  a.dout = d->dout; a.do_bs = d->do_bs; a.do_rs = d->do_rs;
  a.dQ = d->dQ; a.dq_bs = d->dq_bs; a.dq_rs = d->dq_rs;
  a.dK = d->dK; a.dk_bs = d->dk_bs; a.dk_rs = d->dk_rs;
  a.dV = d->dV; a.dv_bs = d->dv_bs; a.dv_rs = d->dv_rs;
  const int per_wave = 2 * d->f.Tk * (d->f.dh + 1) + 2 * d->f.dh + d->f.Tk + 8;
  a.lds_per_wave = (per_wave + 3) & ~3;
  int nw = 4;
  while (nw > 1 && (size_t)nw * a.lds_per_wave * 4 > 64 * 1024) nw >>= 1;
  const size_t lds = (size_t)nw * a.lds_per_wave * 4;
  DMT_CHECK_ARG(lds <= 160 * 1024, "dmt_attn_bwd: Tk*dh too large for LDS");
  hipStream_t st = (hipStream_t)stream;
  {
    const dmt_attn_desc& f1 = d->f;
    if (f1.Tq == 1 && q1_aligned(f1.dtype, f1.Q, f1.q_bs, f1.q_rs, f1.dh) && q1_aligned(f1.dtype, f1.K, f1.k_bs, f1.k_rs, f1.dh) &&
        q1_aligned(f1.dtype, f1.V, f1.v_bs, f1.v_rs, f1.dh) && q1_aligned(f1.dtype, d->dout, d->do_bs, d->do_rs, f1.dh) &&
        q1_aligned(f1.dtype, d->dK, d->dk_bs, d->dk_rs, f1.dh) && q1_aligned(f1.dtype, d->dV, d->dv_bs, d->dv_rs, f1.dh)) {
      if (f1.dtype == DMT_BF16 && f1.Tk <= 64 && q1v_aligned(f1.Q, f1.q_bs, f1.q_rs) && q1v_aligned(f1.K, f1.k_bs, f1.k_rs) &&
          q1v_aligned(f1.V, f1.v_bs, f1.v_rs) && q1v_aligned(d->dout, d->do_bs, d->do_rs) && q1v_aligned(d->dQ, d->dq_bs, d->dq_rs) &&
          q1v_aligned(d->dK, d->dk_bs, d->dk_rs) && q1v_aligned(d->dV, d->dv_bs, d->dv_rs) && launch_q1v<true>(a, st) == 0) {
        DMT_CHECK_LAUNCH("dmt_attn_bwd(q1v)");
        return DMT_OK;
      }
      const int r1 = (f1.dtype == DMT_F32) ? launch_q1<float, true>(a, st) : launch_q1<bf16_t, true>(a, st);
      if (r1 == 0) { DMT_CHECK_LAUNCH("dmt_attn_bwd(q1)"); return DMT_OK; }
    }
  }
  {
    auto al8 = [](const void* q, long long s0, long long s1) { return q == nullptr || (((uintptr_t)q) % 8 == 0 && s0 % 4 == 0 && s1 % 4 == 0); };
    const dmt_attn_desc& f = d->f;
    const bool mfma_ok = f.dtype == DMT_BF16 && f.Tq >= 8 && (f.dh == 20 || f.dh == 80 || f.dh == 16 || f.dh == 32 || f.dh == 64) &&
                         al8(f.Q, f.q_bs, f.q_rs) && al8(f.K, f.k_bs, f.k_rs) && al8(f.V, f.v_bs, f.v_rs) &&
                         al8(d->dout, d->do_bs, d->do_rs) && al8(d->dQ, d->dq_bs, d->dq_rs) && al8(d->dK, d->dk_bs, d->dk_rs) &&
                         al8(d->dV, d->dv_bs, d->dv_rs);
    if (mfma_ok) {
      a.vec16 = (f.dh % 8 == 0 && q1v_aligned(f.Q, f.q_bs, f.q_rs) && q1v_aligned(f.K, f.k_bs, f.k_rs) && q1v_aligned(f.V, f.v_bs, f.v_rs) &&
                 q1v_aligned(d->dout, d->do_bs, d->do_rs)) ? 1 : 0;
      const size_t ldsm = ((size_t)(f.dh > 64 ? f.dh : 64) * 72 + 64 * 72) * 2;   // one transposed tile + one [key][q] tile
      const unsigned nbm = (unsigned)((long long)f.B * f.H);
      if (a.vec16 && f.dh % 16 == 0 && q1v_aligned(d->dQ, d->dq_bs, d->dq_rs) && q1v_aligned(d->dK, d->dk_bs, d->dk_rs) &&
          q1v_aligned(d->dV, d->dv_bs, d->dv_rs)) {
        a.row_off = f.row_off; a.ex_list = f.ex_list; a.n_list = f.n_list;
        const int tmax = (packed && f.max_len > 0 && f.max_len < f.Tq) ? f.max_len : f.Tq;
        const int ntq = tmax <= 32 ? 1 : 2, ntk = (packed ? tmax : f.Tk) <= 32 ? 1 : 2;
        const long long n_ex = f.ex_list ? f.n_list : f.B;
        size_t lds_extra = 0;      // timing experiments only (make EXPERIMENTS=1): LDS a wavefront reserves on top of its own -- fewer wavefronts per CU
#ifdef DMT_TIMING_EXPERIMENTS
        { const char* e = getenv("DMT_ATTN_BWD_LDS_EXTRA"); lds_extra = e ? (size_t)atoi(e) : 0; }
#endif
#define DMT_BWD_CO(DHV) do { const size_t lb = (size_t)CoBwd<DHV>::BYTES + lds_extra; \
    const unsigned nbx = (unsigned)((n_ex + 7) / 8 * 8 * f.H);       /* (examples in groups of 8: one per XCD) */ \
    if (ntq == 1 && ntk == 1) hipLaunchKernelGGL((attn_bwd_co_kernel<DHV, 1, 1>), dim3(nbx), dim3(64), lb, st, a); \
    else if (ntq == 1) hipLaunchKernelGGL((attn_bwd_co_kernel<DHV, 1, 2>), dim3(nbx), dim3(64), lb, st, a); \
    else hipLaunchKernelGGL((attn_bwd_co_kernel<DHV, 2, 2>), dim3(nbx), dim3(64), lb, st, a); } while (0)
        switch (f.dh) {
          case 16: DMT_BWD_CO(16); break;
          case 32: DMT_BWD_CO(32); break;
          case 64: DMT_BWD_CO(64); break;
          default: DMT_BWD_CO(80); break;
        }
#undef DMT_BWD_CO
        DMT_CHECK_LAUNCH("dmt_attn_bwd(mfma, coalesced)");
        return DMT_OK;
      }
      DMT_CHECK_ARG(!packed, "dmt_attn_bwd: packed rows need 16-byte aligned rows and dh %% 16 == 0 (the coalesced MFMA kernel)");
      switch (f.dh) {
        case 16: hipLaunchKernelGGL((attn_bwd_mfma_kernel<16>), dim3(nbm), dim3(64), ldsm, st, a); break;
        case 20: hipLaunchKernelGGL((attn_bwd_mfma_kernel<20>), dim3(nbm), dim3(64), ldsm, st, a); break;
        case 32: hipLaunchKernelGGL((attn_bwd_mfma_kernel<32>), dim3(nbm), dim3(64), ldsm, st, a); break;
        case 64: hipLaunchKernelGGL((attn_bwd_mfma_kernel<64>), dim3(nbm), dim3(64), ldsm, st, a); break;
        default: hipLaunchKernelGGL((attn_bwd_mfma_kernel<80>), dim3(nbm), dim3(64), ldsm, st, a); break;
      }
      DMT_CHECK_LAUNCH("dmt_attn_bwd(mfma)");
      return DMT_OK;
    }
  }
  DMT_CHECK_ARG(!packed, "dmt_attn_bwd: packed rows are taken by the coalesced MFMA kernel only (bf16, dh in 16/32/64/80, 16-byte aligned rows, T >= 8)");
  int r = (d->f.dtype == DMT_F32) ? launch_bwd<float>(a, nw, lds, st) : launch_bwd<bf16_t>(a, nw, lds, st);
  if (r != 0) { dmt_set_error("dmt_attn_bwd: head dim %d not instantiated (4,8,16,20,32,64,80)", d->f.dh); return DMT_ERR_UNSUPPORTED; }
  DMT_CHECK_LAUNCH("dmt_attn_bwd");
  return DMT_OK;
}
