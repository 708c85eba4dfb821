// Target-as-query cross attention WITHOUT the K / V projections of the memory (decoder of every behaviour sequence:
// TransformerModel.decode -> multihead_attention(queries = target [B, 1, d], keys = values = memory [B, T, d]),
// TransformerModel_util.py:160-209 with 11-56, 80-108).
//
// With ONE query per example the projections re-associate.  Per head h (dh = d / H columns hc of the packed kernels):
//     score_k = Q_h . K_k,h / sqrt(dh),   K_k,h = mem_k Wk[:, hc] + bk_h      =  (Wk[:, hc] Q_h) . mem_k / sqrt(dh)  + const
//     out_h   = sum_k P_k V_k,h,          V_k,h = mem_k Wv[:, hc] + bv_h      =  (sum_k P_k mem_k) Wv[:, hc] + (sum_k P_k) bv_h
// (the constant Q_h . bk_h shifts every unmasked score alike: the softmax does not see it, and masked keys carry the reference's
// padding value either way).  So instead of projecting T memory rows to K and V (a [B*T, d] x [d, 2d] GEMM forward, its input
// gradient and its weight gradient -- ~1 ms of the 12 ms step -- plus a kernel that reads the 2d-wide K|V back) the query is
// projected INTO memory space (q' = Q_h Wk[:, hc]^T, a [B, dh] x [dh, d] GEMM), these kernels attend over the raw memory rows, and
// the V projection is applied to the d-wide context (ctx_h | sum P) afterwards ([B, d+1] x [d+1, dh]).  Every GEMM left has B rows.
//
//   forward   ctx [B][H][d+8]: cols 0..d-1 = sum_k Pd_k mem_k, col d = sum_k Pd_k (Pd = weights after the dropout), rest 0
//   backward  d ctx, d out (for d (sum P) = d out_h . bv_h) -> d q' [B][H][d], d mem [B][T][d]
// One workgroup per example, one wavefront per head; the memory rows pass through LDS in chunks of 64 keys (one chunk at T <= 64:
// read once).  Scores: lane = key, a d-long dot product against the row in LDS; context / d q': lane = 8 columns, walking the keys.
// Same arithmetic as the single-query kernels of dmt_attn.hip (IEEE division, expf; key mask before the softmax; dropout counter
// ((b*H + h)*1 + 0)*T + k).  The query is always valid on this path (the decoder passes no query lengths).
#include "dmt_common.h"

namespace {

constexpr float PADDING_NUM = -4294967295.0f;
constexpr int D = 320;                 // memory row width
constexpr int RS = D + 8;              // LDS row stride (elements): 16-byte reads, rows 4 banks apart
constexpr int CK = 64;                 // keys per chunk
constexpr int MAXT = 256;
constexpr int MAXH = 4;
constexpr int NCH = D / 8;             // 16-byte chunks per row (40)

struct Q1mArgs {
  int B, T, H;
  float inv_sc;                        // 1 / sqrt(dh)
  const bf16_t* mem; long long m_bs, m_rs;
  const int* k_lens;
  const bf16_t* qp;                    // [B][H][D] bf16
  bf16_t* ctx; long long ctx_hs;       // [B][H][ctx_hs]
  unsigned drop_seed, drop_thr;
  float drop_inv;
  int drop_on;
  const bf16_t* dctx;                  // [B][H][D] bf16
  const bf16_t* dout; long long do_bs; // [B][H*dh]
  const float* bv;                     // [H*dh]
  int dh;
  bf16_t* dqp;                         // [B][H][D]
  bf16_t* dmem; long long dm_bs, dm_rs;
  const int* row_off;                  // packed rows (include/dmt_hip.h): example b's memory rows at rows row_off[b] + k, k < k_lens[b]; null: dense
};

__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xFFFF0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xFFFF0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xFFFF0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xFFFF0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(dmt_pack_bf16(f[0], f[1]), dmt_pack_bf16(f[2], f[3]), dmt_pack_bf16(f[4], f[5]), dmt_pack_bf16(f[6], f[7]));
}

// packed bf16 dot product with fp32 accumulate: acc + a.lo * b.lo + a.hi * b.hi   (v_dot2c_f32_bf16)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2(unsigned a, unsigned b, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), acc, false);
}
__device__ __forceinline__ float dot8(const uint4& m, const uint4& v, float acc) {
  acc = dot2(m.x, v.x, acc); acc = dot2(m.y, v.y, acc); acc = dot2(m.z, v.z, acc); acc = dot2(m.w, v.w, acc);
  return acc;
}
// acc[0..7] += s * (the 8 bf16 of m), s given as bf16 bits: two dot2 per dword against (s, 0) and (0, s) -- no unpacking
__device__ __forceinline__ void axpy8(float (&acc)[8], const uint4& m, unsigned sb) {
  const unsigned lo = sb & 0xFFFFu, hi = sb << 16;
  acc[0] = dot2(m.x, lo, acc[0]); acc[1] = dot2(m.x, hi, acc[1]);
  acc[2] = dot2(m.y, lo, acc[2]); acc[3] = dot2(m.y, hi, acc[3]);
  acc[4] = dot2(m.z, lo, acc[4]); acc[5] = dot2(m.z, hi, acc[5]);
  acc[6] = dot2(m.w, lo, acc[6]); acc[7] = dot2(m.w, hi, acc[7]);
}

// dot products of LDS row `row` (bf16) with one or two bf16 vectors in LDS (wave-uniform addresses: broadcast reads)
template <bool TWO>
__device__ __forceinline__ void row_dots(const bf16_t* __restrict__ row, const bf16_t* __restrict__ v0, const bf16_t* __restrict__ v1, float& a0, float& a1) {
  a0 = 0.f; a1 = 0.f;
#pragma unroll 8
  for (int ch = 0; ch < NCH; ++ch) {
    const uint4 m = *reinterpret_cast<const uint4*>(row + ch * 8);
    a0 = dot8(m, *reinterpret_cast<const uint4*>(v0 + ch * 8), a0);
    if constexpr (TWO) a1 = dot8(m, *reinterpret_cast<const uint4*>(v1 + ch * 8), a1);
  }
}

// masked softmax over the T scores of one head (in sc, natural units, already divided by sqrt(dh)); leaves P in sc
__device__ __forceinline__ void softmax_row(float* __restrict__ sc, int T, int klen, int lane) {
  float m = -3.0e38f;
  for (int k = lane; k < T; k += 64) {
    const float x = (k < klen) ? sc[k] : PADDING_NUM;
    sc[k] = x;
    m = fmaxf(m, x);
  }
  m = wave_max(m);
  float sum = 0.f;
  for (int k = lane; k < T; k += 64) {
    const float e = expf(sc[k] - m);
    sc[k] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  for (int k = lane; k < T; k += 64) sc[k] = sc[k] / sum;
}

// LDS carve-up (dynamic; the host sizes it from T so that short histories leave room for four workgroups per CU):
//   s_mem [MR][RS] bf16 | s_q [H][D] bf16 | (bwd) s_dc [H][D] bf16 | s_sc [H][TP] f32 | s_g [H][TP] f32 | (bwd) s_cq, s_pds
struct Q1mLds {
  bf16_t* mem; bf16_t* q; bf16_t* dc; float* sc; float* g; unsigned* pb; unsigned* cq; unsigned* pds; float* dS;
};
__host__ __device__ inline int q1m_mem_rows(int T) { const int r = T < CK ? T : CK; return (r + 1) & ~1; }
__host__ __device__ inline int q1m_tp(int T) { return (T + 63) & ~63; }
__host__ __device__ inline size_t q1m_lds_bytes(int T, int H, bool bwd) {
  size_t b = (size_t)q1m_mem_rows(T) * RS * 2 + (size_t)H * D * 2 + (size_t)H * q1m_tp(T) * 4 * 2 + 64;
  if (bwd) b += (size_t)H * D * 2 + (size_t)H * D * 4 + (size_t)H * q1m_tp(T) * 4;
  return (b + 255) & ~(size_t)255;
}
__device__ __forceinline__ Q1mLds q1m_carve(unsigned char* base, int T, int H, bool bwd) {
  Q1mLds L;
  unsigned char* p = base;
  L.mem = (bf16_t*)p; p += (size_t)q1m_mem_rows(T) * RS * 2;
  L.q = (bf16_t*)p; p += (size_t)H * D * 2;
  L.dc = (bf16_t*)p; if (bwd) p += (size_t)H * D * 2;
  L.sc = (float*)p; p += (size_t)H * q1m_tp(T) * 4;
  L.g = (float*)p; L.pb = (unsigned*)p; p += (size_t)H * q1m_tp(T) * 4;
  L.cq = (unsigned*)p; if (bwd) p += (size_t)H * D * 4;
  L.pds = (unsigned*)p; if (bwd) p += (size_t)H * q1m_tp(T) * 4;
  L.dS = (float*)p;
  return L;
}

// rows [k0, k0 + rows) of example b's memory -> LDS (rows past T: zeros).  All of a thread's 16-byte requests (at most ten: rows <= CK)
// are in flight together, branch-free (rows past T re-read row T - 1 and are replaced by zeros): the rolled load -> wait -> store loop
// paid one memory latency per piece, ten in a row at T = 64 -- most of a workgroup's lifetime.
__device__ __forceinline__ void stage_rows(bf16_t* __restrict__ s_mem, const bf16_t* __restrict__ mem_b, long long m_rs, int k0, int rows, int T, int tid) {
  constexpr int IT = (CK * NCH + 255) / 256;
  uint4 v[IT];
  const int n = rows * NCH;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = tid + i * 256;
    const int cc = c < n ? c : n - 1;
    const int r = cc / NCH, ch = cc - r * NCH;
    const int rr = (k0 + r < T) ? k0 + r : T - 1;
    v[i] = *reinterpret_cast<const uint4*>(mem_b + (long long)rr * m_rs + ch * 8);
  }
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = tid + i * 256;
    if (c < n) {
      const int r = c / NCH, ch = c - r * NCH;
      *reinterpret_cast<uint4*>(s_mem + r * RS + ch * 8) = (k0 + r < T) ? v[i] : make_uint4(0u, 0u, 0u, 0u);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void q1m_fwd_kernel(const Q1mArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, Td = a.T, H = a.H;       // Td: the dense length (LDS carve-up, dropout index)
  const Q1mLds L = q1m_carve(smem, Td, H, false);
  const int MR = q1m_mem_rows(Td), TP = q1m_tp(Td);
  int klen = a.k_lens ? a.k_lens[b] : Td;
  klen = klen < 0 ? 0 : (klen > Td ? Td : klen);
  // packed rows: the example HAS klen rows (T = klen: a key past it does not exist -- in the dense layout it is masked to the padding
  // value, whose softmax weight is exactly 0 beside any valid key: the same P, context and gradients on the rows that exist)
  const int T = a.row_off ? (klen > 0 ? klen : 1) : Td;
  const bf16_t* mem_b = a.mem + (a.row_off ? (long long)a.row_off[b] * a.m_rs : (long long)b * a.m_bs);
  for (int i = tid; i < H * D / 8; i += 256) reinterpret_cast<uint4*>(L.q)[i] = reinterpret_cast<const uint4*>(a.qp + (long long)b * H * D)[i];
  const int nchunk = (T + CK - 1) / CK;
  const int h = wave;                  // one head per wavefront (H <= 4)
  const bool act = h < H;
  // ---- scores: lane = key
  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();
    stage_rows(L.mem, mem_b, a.m_rs, c * CK, MR, T, tid);
    __syncthreads();
    if (act && lane < MR) {
      float s0, s1;
      row_dots<false>(L.mem + lane * RS, L.q + h * D, nullptr, s0, s1);
      if (c * CK + lane < T) L.sc[h * TP + c * CK + lane] = s0 * a.inv_sc;
    }
  }
  __syncthreads();
  // ---- softmax, dropout; the weights as bf16 bits for the context sums, their sum S from the same rounded values
  float ssum = 0.f;
  if (act) {
    softmax_row(L.sc + h * TP, T, klen, lane);
    for (int k = lane; k < T; k += 64) {
      float p = L.sc[h * TP + k];
      if (a.drop_on) p = dmt_drop_keep(a.drop_seed, (unsigned)((b * H + h) * Td + k), a.drop_thr) ? p * a.drop_inv : 0.f;
      const bf16_t pb = f2bf(p);
      L.pb[h * TP + k] = (unsigned)pb;
      ssum += bf2f(pb);
    }
    ssum = wave_sum(ssum);
  }
  // ---- context: lane = 8 columns
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int c = 0; c < nchunk; ++c) {
    if (nchunk > 1) {
      __syncthreads();
      stage_rows(L.mem, mem_b, a.m_rs, c * CK, MR, T, tid);
    }
    __syncthreads();
    const int kn = (T - c * CK) < CK ? (T - c * CK) : CK;
    if (act && lane < NCH) {
#pragma unroll 4
      for (int k = 0; k < kn; ++k) axpy8(acc, *reinterpret_cast<const uint4*>(L.mem + k * RS + lane * 8), L.pb[h * TP + c * CK + k]);
    }
  }
  if (act) {
    bf16_t* cp = a.ctx + ((long long)b * H + h) * a.ctx_hs;
    if (lane < NCH) *reinterpret_cast<uint4*>(cp + lane * 8) = pack8(acc);
    if (lane == NCH) {
      const float z[8] = {ssum, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<uint4*>(cp + D) = pack8(z);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------- backward
__global__ __launch_bounds__(256) void q1m_bwd_kernel(const Q1mArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, Td = a.T, H = a.H;
  const Q1mLds L = q1m_carve(smem, Td, H, true);
  const int MR = q1m_mem_rows(Td), TP = q1m_tp(Td);
  int klen = a.k_lens ? a.k_lens[b] : Td;
  klen = klen < 0 ? 0 : (klen > Td ? Td : klen);
  const int T = a.row_off ? (klen > 0 ? klen : 1) : Td;     // (packed rows: see the forward kernel)
  const bf16_t* mem_b = a.mem + (a.row_off ? (long long)a.row_off[b] * a.m_rs : (long long)b * a.m_bs);
  for (int i = tid; i < H * D / 8; i += 256) {
    reinterpret_cast<uint4*>(L.q)[i] = reinterpret_cast<const uint4*>(a.qp + (long long)b * H * D)[i];
    reinterpret_cast<uint4*>(L.dc)[i] = reinterpret_cast<const uint4*>(a.dctx + (long long)b * H * D)[i];
  }
  // (d ctx_h[i], q'_h[i]) interleaved as bf16 pairs: the d mem sums take them as one dot2 operand
  for (int i = tid; i < H * D; i += 256) {
    const unsigned c = a.dctx[(long long)b * H * D + i], q = a.qp[(long long)b * H * D + i];
    L.cq[i] = c | (q << 16);
  }
  const int h = wave;                  // one head per wavefront (H <= 4)
  const bool act = h < H;
  // d (sum_k Pd_k) of head h = d out_h . bv_h
  if (act) {
    float p = 0.f;
    for (int n = lane; n < a.dh; n += 64) p = fmaf(bf2f(a.dout[(long long)b * a.do_bs + h * a.dh + n]), a.bv[h * a.dh + n], p);
    p = wave_sum(p);
    if (lane == 0) L.dS[h] = p;
  }
  const int nchunk = (T + CK - 1) / CK;
  // ---- scores and d Pd: lane = key
  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();
    stage_rows(L.mem, mem_b, a.m_rs, c * CK, MR, T, tid);
    __syncthreads();
    if (act && lane < MR) {
      float s0, g0;
      row_dots<true>(L.mem + lane * RS, L.q + h * D, L.dc + h * D, s0, g0);
      if (c * CK + lane < T) { L.sc[h * TP + c * CK + lane] = s0 * a.inv_sc; L.g[h * TP + c * CK + lane] = g0 + L.dS[h]; }
    }
  }
  __syncthreads();
  // ---- softmax backward: dS_k = P_k (dP_k - sum P dP) / sqrt(dh) on unmasked keys; (Pd_k, dS_k) as a bf16 pair per (head, key)
  if (act) {
    float* sc = L.sc + h * TP;
    float* g = L.g + h * TP;
    softmax_row(sc, T, klen, lane);
    float dot = 0.f;
    for (int k = lane; k < T; k += 64) {
      float keep = 1.f;
      if (a.drop_on) keep = dmt_drop_keep(a.drop_seed, (unsigned)((b * H + h) * Td + k), a.drop_thr) ? a.drop_inv : 0.f;
      const float dp = g[k] * keep;
      g[k] = dp;
      dot += sc[k] * dp;
    }
    dot = wave_sum(dot);
    for (int k = lane; k < T; k += 64) {
      float keep = 1.f;
      if (a.drop_on) keep = dmt_drop_keep(a.drop_seed, (unsigned)((b * H + h) * Td + k), a.drop_thr) ? a.drop_inv : 0.f;
      const float p = sc[k];
      const float ds = (k < klen) ? p * (g[k] - dot) * a.inv_sc : 0.f;     // (carries the 1 / sqrt(dh) of the score)
      L.pds[h * TP + k] = (unsigned)f2bf(p * keep) | ((unsigned)f2bf(ds) << 16);
    }
  }
  // ---- d q'_h = sum_k dS_k mem_k: lane = 8 columns
  {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int c = 0; c < nchunk; ++c) {
      if (nchunk > 1) {
        __syncthreads();
        stage_rows(L.mem, mem_b, a.m_rs, c * CK, MR, T, tid);
      }
      __syncthreads();
      const int kn = (T - c * CK) < CK ? (T - c * CK) : CK;
      if (act && lane < NCH) {
#pragma unroll 4
        for (int k = 0; k < kn; ++k) axpy8(acc, *reinterpret_cast<const uint4*>(L.mem + k * RS + lane * 8), L.pds[h * TP + c * CK + k] >> 16);
      }
    }
    if (act && lane < NCH) *reinterpret_cast<uint4*>(a.dqp + ((long long)b * H + h) * D + lane * 8) = pack8(acc);
  }
  __syncthreads();
  // ---- d mem_k = sum_h Pd_k,h d ctx_h + dS_k,h q'_h: one (key, 8 columns) item per thread, one dot2 per (head, column)
  bf16_t* dm_b = a.dmem + (a.row_off ? (long long)a.row_off[b] * a.dm_rs : (long long)b * a.dm_bs);
  const int n_out = (a.row_off && klen == 0) ? 0 : T;      // (an example without rows owns no row of the packed gradient)
  for (int it = tid; it < n_out * NCH; it += 256) {
    const int k = it / NCH, ch = it - k * NCH;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int hh = 0; hh < H; ++hh) {
      const unsigned pd = L.pds[hh * TP + k];
      const uint4 c0 = *reinterpret_cast<const uint4*>(L.cq + hh * D + ch * 8), c1 = *reinterpret_cast<const uint4*>(L.cq + hh * D + ch * 8 + 4);
      acc[0] = dot2(c0.x, pd, acc[0]); acc[1] = dot2(c0.y, pd, acc[1]); acc[2] = dot2(c0.z, pd, acc[2]); acc[3] = dot2(c0.w, pd, acc[3]);
      acc[4] = dot2(c1.x, pd, acc[4]); acc[5] = dot2(c1.y, pd, acc[5]); acc[6] = dot2(c1.z, pd, acc[6]); acc[7] = dot2(c1.w, pd, acc[7]);
    }
    *reinterpret_cast<uint4*>(dm_b + (long long)k * a.dm_rs + ch * 8) = pack8(acc);
  }
}

int fill(Q1mArgs& a, const dmt_q1mem_desc* d, const char* who) {
  DMT_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  DMT_CHECK_ARG(d->B > 0 && d->T > 0 && d->T <= MAXT && d->H > 0 && d->H <= MAXH && d->d == D && d->dh > 0 && d->dh * d->H == D,
                "%s: built for memory width %d, H <= %d, T <= %d (got d %d, H %d, dh %d, T %d)", who, D, MAXH, MAXT, d->d, d->H, d->dh, d->T);
  DMT_CHECK_ARG(d->mem && d->qp, "%s: null operand", who);
  DMT_CHECK_ARG((((uintptr_t)d->mem) & 15) == 0 && d->m_bs % 8 == 0 && d->m_rs % 8 == 0 && (((uintptr_t)d->qp) & 15) == 0,
                "%s: memory rows and q' must be 16-byte aligned", who);
  a.B = d->B; a.T = d->T; a.H = d->H; a.dh = d->dh;
  a.inv_sc = 1.0f / sqrtf((float)d->dh);
  a.mem = (const bf16_t*)d->mem; a.m_bs = d->m_bs; a.m_rs = d->m_rs;
  a.k_lens = d->k_lens;
  a.qp = (const bf16_t*)d->qp;
  a.ctx = (bf16_t*)d->ctx; a.ctx_hs = d->ctx_hs;
  a.drop_on = (d->drop_keep > 0.f && d->drop_keep < 1.f) ? 1 : 0;
  a.drop_seed = d->drop_seed;
  a.drop_thr = a.drop_on ? (unsigned)(d->drop_keep * 16777216.0f) : 0u;
  a.drop_inv = a.drop_on ? 1.f / d->drop_keep : 1.f;
  a.dctx = (const bf16_t*)d->dctx; a.dout = (const bf16_t*)d->dout; a.do_bs = d->do_bs; a.bv = d->bv;
  a.dqp = (bf16_t*)d->dqp; a.dmem = (bf16_t*)d->dmem; a.dm_bs = d->dm_bs; a.dm_rs = d->dm_rs;
  a.row_off = d->row_off;
  DMT_CHECK_ARG(d->row_off == nullptr || d->k_lens != nullptr, "%s: packed rows (row_off) need k_lens", who);
  return DMT_OK;
}

}  // namespace

extern "C" int dmt_q1mem_supported(int32_t dtype, int32_t d, int32_t H, int32_t T) {
  return (dtype == DMT_BF16 && d == D && H >= 1 && H <= MAXH && D % H == 0 && T >= 1 && T <= MAXT) ? 1 : 0;
}

extern "C" int dmt_q1mem_fwd(const dmt_q1mem_desc* d, void* stream) {
  Q1mArgs a;
  if (fill(a, d, "dmt_q1mem_fwd") != DMT_OK) return DMT_ERR_ARG;
  DMT_CHECK_ARG(d->ctx && d->ctx_hs >= D + 8 && d->ctx_hs % 8 == 0 && (((uintptr_t)d->ctx) & 15) == 0, "dmt_q1mem_fwd: ctx rows need >= d + 8 columns, 16-byte aligned");
  hipLaunchKernelGGL(q1m_fwd_kernel, dim3((unsigned)d->B), dim3(256), q1m_lds_bytes(d->T, d->H, false), (hipStream_t)stream, a);
  DMT_CHECK_LAUNCH("dmt_q1mem_fwd");
  return DMT_OK;
}

extern "C" int dmt_q1mem_bwd(const dmt_q1mem_desc* d, void* stream) {
  Q1mArgs a;
  if (fill(a, d, "dmt_q1mem_bwd") != DMT_OK) return DMT_ERR_ARG;
  DMT_CHECK_ARG(d->dctx && d->dout && d->bv && d->dqp && d->dmem, "dmt_q1mem_bwd: null argument");
  DMT_CHECK_ARG((((uintptr_t)d->dctx) & 15) == 0 && (((uintptr_t)d->dqp) & 15) == 0 && (((uintptr_t)d->dmem) & 15) == 0 && d->dm_bs % 8 == 0 && d->dm_rs % 8 == 0,
                "dmt_q1mem_bwd: rows must be 16-byte aligned");
  hipLaunchKernelGGL(q1m_bwd_kernel, dim3((unsigned)d->B), dim3(256), q1m_lds_bytes(d->T, d->H, true), (hipStream_t)stream, a);
  DMT_CHECK_LAUNCH("dmt_q1mem_bwd");
  return DMT_OK;
}
