// Row-wise kernels: LayerNorm fwd/bwd, MMoE gate softmax + mixture fwd/bwd, unbias loss fwd+bwd,
// relu gradient mask, column sums, bf16 shadow casts, AUC confusion histogram.
#include "dmt_common.h"

#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------ LayerNorm
// One wavefront per row; lane owns columns lane, lane+64, ...  (coalesced for any alignment / ld).
template <typename T, int NE>
__global__ __launch_bounds__(256) void ln_fwd_kernel(long long rows, int d, const T* __restrict__ x, long long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, T* __restrict__ y, long long ldy, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const long long w0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nw = (long long)gridDim.x * 4;
  float gm[NE], bt[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int c = lane + 64 * i;
    gm[i] = (c < d) ? gamma[c] : 0.f;
    bt[i] = (c < d) ? beta[c] : 0.f;
  }
  for (long long r = w0; r < rows; r += nw) {
    float v[NE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int c = lane + 64 * i;
      v[i] = (c < d) ? ldf<T>(x + r * ldx + c) : 0.f;
      s += v[i];
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int c = lane + 64 * i;
      const float t = (c < d) ? (v[i] - mean) : 0.f;
      q += t * t;
    }
    const float var = wave_sum(q) / (float)d;
    const float den = sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int c = lane + 64 * i;
      if (c < d) stf<T>(y + r * ldy + c, gm[i] * ((v[i] - mean) / den) + bt[i]);
    }
    if (stats && lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = 1.f / den; }
  }
}

template <typename T, int NE>
__global__ __launch_bounds__(256) void ln_bwd_kernel(long long rows, int d, const T* __restrict__ x, long long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ stats,
                                                     const T* __restrict__ dy, long long lddy, T* __restrict__ dx, long long lddx,
                                                     float* __restrict__ partials) {
  __shared__ float s_part[4][2][NE * 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long long w0 = (long long)blockIdx.x * 4 + wave;
  const long long nw = (long long)gridDim.x * 4;
  float gm[NE], dg[NE], db[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int c = lane + 64 * i;
    gm[i] = (c < d) ? gamma[c] : 0.f;
    dg[i] = 0.f;
    db[i] = 0.f;
  }
  for (long long r = w0; r < rows; r += nw) {
    const float mean = stats[2 * r], rstd = stats[2 * r + 1];
    float xh[NE], g[NE];
    float a = 0.f, bq = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int c = lane + 64 * i;
      const float xv = (c < d) ? ldf<T>(x + r * ldx + c) : 0.f;
      const float dv = (c < d) ? ldf<T>(dy + r * lddy + c) : 0.f;
      xh[i] = (c < d) ? (xv - mean) * rstd : 0.f;
      g[i] = dv * gm[i];
      a += g[i];
      bq += g[i] * xh[i];
      dg[i] += dv * xh[i];
      db[i] += dv;
    }
    a = wave_sum(a) / (float)d;
    bq = wave_sum(bq) / (float)d;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int c = lane + 64 * i;
      if (c < d) stf<T>(dx + r * lddx + c, rstd * (g[i] - a - xh[i] * bq));
    }
  }
#pragma unroll
  for (int i = 0; i < NE; ++i) { s_part[wave][0][i * 64 + lane] = dg[i]; s_part[wave][1][i * 64 + lane] = db[i]; }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * NE * 64; c += 256) {
    const int which = c / (NE * 64), cc = c % (NE * 64);
    if (cc < d) {
      const float s = s_part[0][which][cc] + s_part[1][which][cc] + s_part[2][which][cc] + s_part[3][which][cc];
      partials[(long long)blockIdx.x * 2 * d + which * d + cc] = s;
    }
  }
}

// ---- bf16, d % 8 == 0, d <= 512, 16-byte aligned rows: lane owns the 8 consecutive columns 8*lane .. 8*lane+7 (lanes past
// d/8 idle), so a row costs ONE 16-byte access per array instead of d/64 two-byte ones (the kernels are HBM bound; the
// request count was the limit: 3.9 -> ~5 TB/s).
__device__ __forceinline__ void ln_unpack8(const uint4& u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xFFFF0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xFFFF0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xFFFF0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xFFFF0000u);
}
__device__ __forceinline__ uint4 ln_pack8(const float (&f)[8]) {
  return make_uint4((unsigned)f2bf(f[0]) | ((unsigned)f2bf(f[1]) << 16), (unsigned)f2bf(f[2]) | ((unsigned)f2bf(f[3]) << 16),
                    (unsigned)f2bf(f[4]) | ((unsigned)f2bf(f[5]) << 16), (unsigned)f2bf(f[6]) | ((unsigned)f2bf(f[7]) << 16));
}

__global__ __launch_bounds__(256) void ln_fwd_v8_kernel(long long rows, int d, const bf16_t* __restrict__ x, long long ldx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        bf16_t* __restrict__ y, long long ldy, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const long long w0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nw = (long long)gridDim.x * 4;
  const int c0 = lane * 8;
  const bool act = c0 < d;
  float gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { gm[e] = act ? gamma[c0 + e] : 0.f; bt[e] = act ? beta[c0 + e] : 0.f; }
  for (long long r = w0; r < rows; r += nw) {
    float v[8];
    const uint4 xu = act ? *reinterpret_cast<const uint4*>(x + r * ldx + c0) : make_uint4(0u, 0u, 0u, 0u);
    ln_unpack8(xu, v);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float t = act ? (v[e] - mean) : 0.f; q += t * t; }
    const float var = wave_sum(q) / (float)d;
    const float den = sqrtf(var + eps);
    if (act) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = gm[e] * ((v[e] - mean) / den) + bt[e];
      *reinterpret_cast<uint4*>(y + r * ldy + c0) = ln_pack8(o);
    }
    if (stats && lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = 1.f / den; }
  }
}

__global__ __launch_bounds__(256) void ln_bwd_v8_kernel(long long rows, int d, const bf16_t* __restrict__ x, long long ldx,
                                                        const float* __restrict__ gamma, const float* __restrict__ stats,
                                                        const bf16_t* __restrict__ dy, long long lddy, bf16_t* __restrict__ dx, long long lddx,
                                                        float* __restrict__ partials) {
  __shared__ float s_part[4][2][512];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long long w0 = (long long)blockIdx.x * 4 + wave;
  const long long nw = (long long)gridDim.x * 4;
  const int c0 = lane * 8;
  const bool act = c0 < d;
  float gm[8], dg[8], db[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { gm[e] = act ? gamma[c0 + e] : 0.f; dg[e] = 0.f; db[e] = 0.f; }
  const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
  for (long long r = w0; r < rows; r += nw) {
    const float mean = stats[2 * r], rstd = stats[2 * r + 1];
    float xv[8], dv[8], xh[8], g[8];
    ln_unpack8(act ? *reinterpret_cast<const uint4*>(x + r * ldx + c0) : z4, xv);
    ln_unpack8(act ? *reinterpret_cast<const uint4*>(dy + r * lddy + c0) : z4, dv);
    float a = 0.f, bq = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xh[e] = act ? (xv[e] - mean) * rstd : 0.f;
      g[e] = dv[e] * gm[e];
      a += g[e];
      bq += g[e] * xh[e];
      dg[e] += dv[e] * xh[e];
      db[e] += dv[e];
    }
    a = wave_sum(a) / (float)d;
    bq = wave_sum(bq) / (float)d;
    if (act) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = rstd * (g[e] - a - xh[e] * bq);
      *reinterpret_cast<uint4*>(dx + r * lddx + c0) = ln_pack8(o);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { s_part[wave][0][c0 + e] = dg[e]; s_part[wave][1][c0 + e] = db[e]; }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * 512; c += 256) {
    const int which = c >> 9, cc = c & 511;
    if (cc < d) {
      const float s = s_part[0][which][cc] + s_part[1][which][cc] + s_part[2][which][cc] + s_part[3][which][cc];
      partials[(long long)blockIdx.x * 2 * d + which * d + cc] = s;
    }
  }
}

static bool ln_v8_ok(int32_t dtype, int d, const void* p0, long long ld0, const void* p1, long long ld1, const void* p2, long long ld2) {
  auto al = [](const void* p, long long ld) { return p == nullptr || ((((uintptr_t)p) & 15) == 0 && (ld & 7) == 0); };
  return dtype == DMT_BF16 && (d & 7) == 0 && d <= 512 && al(p0, ld0) && al(p1, ld1) && al(p2, ld2);
}

// One wavefront per column: lanes stride over the per-block partials, then a shuffle reduction (deterministic order).
__global__ __launch_bounds__(256) void ln_bwd_finish_kernel(int nblk, int d, const float* __restrict__ partials,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= 2 * d) return;
  float s = 0.f;
  for (int b = lane; b < nblk; b += 64) s += partials[(long long)b * 2 * d + c];
  s = wave_sum(s);
  if (lane == 0) { if (c < d) dgamma[c] += s; else dbeta[c - d] += s; }
}

// the same for a list of LayerNorm gradients in one launch: a step's dgamma / dbeta all finish together.  blockIdx.y = a GROUP of jobs
// with one destination (the encoder's and the decoder's feed-forward LayerNorm share gamma / beta: tie_ffn), summed one after the
// other by the same wavefront -- a fixed order, and no two blocks add to one address.
struct LnFinishJobs { int n_groups; int first[DMT_LN_FINISH_MAX + 1]; dmt_ln_finish_job job[DMT_LN_FINISH_MAX]; };
__global__ __launch_bounds__(256) void ln_bwd_finish_jobs_kernel(const LnFinishJobs jobs) {
  const int g = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int d = jobs.job[jobs.first[g]].d;
  if (c >= 2 * d) return;
  float s = 0.f;
  for (int i = jobs.first[g]; i < jobs.first[g + 1]; ++i) {
    const dmt_ln_finish_job& j = jobs.job[i];
    float sj = 0.f;
    for (int b = lane; b < j.n_part; b += 64) sj += j.partials[(long long)b * 2 * d + c];
    s += wave_sum(sj);
  }
  const dmt_ln_finish_job& j0 = jobs.job[jobs.first[g]];
  if (lane == 0) { if (c < d) j0.dgamma[c] += s; else j0.dbeta[c - d] += s; }
}

// ------------------------------------------------------------------------------------------ MMoE mix
template <typename T>
__global__ __launch_bounds__(128) void mix_fwd_kernel(int B, int E, int U, int nt, const T* __restrict__ expert, long long lde,
                                                      const T* __restrict__ glogit, long long ldg, float* __restrict__ gates,
                                                      T* __restrict__ mix) {
  __shared__ float s_g[64];
  const int b = blockIdx.x;
  if (threadIdx.x < nt) {
    const int t = threadIdx.x;
    float m = -3.0e38f;
    for (int e = 0; e < E; ++e) m = fmaxf(m, ldf<T>(glogit + (long long)b * ldg + t * E + e));
    float s = 0.f;
    for (int e = 0; e < E; ++e) { const float v = expf(ldf<T>(glogit + (long long)b * ldg + t * E + e) - m); s_g[t * E + e] = v; s += v; }
    for (int e = 0; e < E; ++e) { const float v = s_g[t * E + e] / s; s_g[t * E + e] = v; gates[((long long)t * B + b) * E + e] = v; }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < U; j += blockDim.x) {
    for (int t = 0; t < nt; ++t) {
      float acc = 0.f;
      for (int e = 0; e < E; ++e) acc = fmaf(s_g[t * E + e], ldf<T>(expert + (long long)b * lde + e * U + j), acc);
      stf<T>(mix + ((long long)t * B + b) * U + j, acc);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(128) void mix_bwd_kernel(int B, int E, int U, int nt, const T* __restrict__ expert, long long lde,
                                                      const float* __restrict__ gates, const T* __restrict__ dmix,
                                                      T* __restrict__ dexpert, long long ldde, T* __restrict__ dglogit, long long lddg,
                                                      int relu_mask) {
  __shared__ float s_g[64];
  __shared__ float s_dg[2][64];
  const int b = blockIdx.x;
  const int ng = nt * E;
  if (threadIdx.x < ng) s_g[threadIdx.x] = gates[((long long)(threadIdx.x / E) * B + b) * E + (threadIdx.x % E)];
  __syncthreads();
  float part[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) part[i] = 0.f;
  for (int j = threadIdx.x; j < U; j += blockDim.x) {
    float dm[4];
    for (int t = 0; t < nt; ++t) dm[t] = ldf<T>(dmix + ((long long)t * B + b) * U + j);
    for (int e = 0; e < E; ++e) {
      const float ev = ldf<T>(expert + (long long)b * lde + e * U + j);
      float de = 0.f;
      for (int t = 0; t < nt; ++t) {
        de = fmaf(s_g[t * E + e], dm[t], de);
        if (t * E + e < 16) part[t * E + e] = fmaf(ev, dm[t], part[t * E + e]);
      }
      if (relu_mask && !(ev > 0.f)) de = 0.f;
      stf<T>(dexpert + (long long)b * ldde + e * U + j, de);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float s = wave_sum(part[i]);
    if (lane == 0 && i < ng) s_dg[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < nt) {
    const int t = threadIdx.x;
    float dot = 0.f;
    for (int e = 0; e < E; ++e) dot += s_g[t * E + e] * (s_dg[0][t * E + e] + s_dg[1][t * E + e]);
    for (int e = 0; e < E; ++e) {
      const float dg = s_dg[0][t * E + e] + s_dg[1][t * E + e];
      stf<T>(dglogit + (long long)b * lddg + t * E + e, s_g[t * E + e] * (dg - dot));
    }
  }
}

// ------------------------------------------------------------------------------------------ loss
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// keras sparse_categorical_crossentropy on [1-p, p] with clipping (inference_mlp.py:162-168):
//   x = log(q0 + q1) - log(q_y),  q = clip([1-p, p], eps, 1-eps);  returns x and dx/dp
__device__ __forceinline__ void xent_clip(float p, int y, float& x, float& dxdp) {
  const float eps = 1e-7f, hi = 1.f - 1e-7f;
  const float r0 = 1.f - p, r1 = p;
  const float q0 = fminf(fmaxf(r0, eps), hi), q1 = fminf(fmaxf(r1, eps), hi);
  const float d0 = (r0 > eps && r0 < hi) ? -1.f : 0.f;   // dq0/dp
  const float d1 = (r1 > eps && r1 < hi) ? 1.f : 0.f;    // dq1/dp
  const float qs = q0 + q1;
  const float qy = y ? q1 : q0;
  const float dy = y ? d1 : d0;
  x = logf(qs) - logf(qy);
  dxdp = (d0 + d1) / qs - dy / qy;
}

__global__ __launch_bounds__(1024) void loss_unbias_kernel(int B, const float* __restrict__ click, const float* __restrict__ order,
                                                           const float* __restrict__ ybias, const float* __restrict__ mask5,
                                                           const float* __restrict__ w_ctr, const float* __restrict__ w_ecvr,
                                                           float lw_clk, float lw_ord, int method, int ctr_rel, float gscale,
                                                           float* __restrict__ loss, float* __restrict__ p_ctr_o,
                                                           float* __restrict__ p_cvr_o, float* __restrict__ d_click,
                                                           float* __restrict__ d_order, float* __restrict__ d_bias) {
  __shared__ float s_red[16];
  float lsum = 0.f;
  const float invB = 1.f / (float)B;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float c = click[b], o = order[b], yb = ybias[b];
    const float* mk = mask5 + (long long)b * 5;
    float wc = 0.f, wo = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) { wc += mk[k] * w_ctr[k]; wo += mk[k] * w_ecvr[k]; }
    const int y_clk = (int)(mk[1] + mk[2] + mk[3] + mk[4]);
    const int y_ord = (int)(mk[3] + mk[4]);
    float p_ctr, p_cvr, dpc_dc, dpc_db, dpv_do, dpv_db;
    const float sc = sigmoidf_(c), so = sigmoidf_(o);
    if (method == 2) {
      // logit_loss: tf.nn.sigmoid_cross_entropy_with_logits = max(x,0) - x*z + log(1 + exp(-|x|)), no bias tower
      const float xc = fmaxf(c, 0.f) - c * (float)y_clk + log1pf(expf(-fabsf(c)));
      const float xo = fmaxf(o, 0.f) - o * (float)y_ord + log1pf(expf(-fabsf(o)));
      lsum += lw_clk * wc * xc + lw_ord * wo * xo;
      if (p_ctr_o) p_ctr_o[b] = sc;
      if (p_cvr_o) p_cvr_o[b] = so;
      if (d_click) d_click[b] = gscale * invB * lw_clk * wc * (sc - (float)y_clk);
      if (d_order) d_order[b] = gscale * invB * lw_ord * wo * (so - (float)y_ord);
      if (d_bias) d_bias[b] = 0.f;
      continue;
    }
    if (method == 1) {
      const float sb = sigmoidf_(yb);
      p_ctr = sc * sb; p_cvr = so * sb;
      dpc_dc = sc * (1.f - sc) * sb; dpc_db = sc * sb * (1.f - sb);
      dpv_do = so * (1.f - so) * sb; dpv_db = so * sb * (1.f - sb);
    } else {
      p_ctr = sigmoidf_(c + yb); p_cvr = sigmoidf_(o + yb);
      dpc_dc = dpc_db = p_ctr * (1.f - p_ctr);
      dpv_do = dpv_db = p_cvr * (1.f - p_cvr);
    }
    float x1, g1, x2, g2;
    xent_clip(p_ctr, y_clk, x1, g1);
    xent_clip(p_cvr, y_ord, x2, g2);
    float xc = x1, xo = x2;
    float dc = g1 * dpc_dc, dbb = wc * lw_clk * g1 * dpc_db + wo * lw_ord * g2 * dpv_db, dd = g2 * dpv_do;
    if (ctr_rel) {
      float x3, g3, x4, g4;
      xent_clip(sc, y_clk, x3, g3);
      xent_clip(so, y_ord, x4, g4);
      xc += x3; xo += x4;
      dc += g3 * sc * (1.f - sc);
      dd += g4 * so * (1.f - so);
    }
    lsum += lw_clk * wc * xc + lw_ord * wo * xo;
    if (p_ctr_o) p_ctr_o[b] = p_ctr;
    if (p_cvr_o) p_cvr_o[b] = p_cvr;
    if (d_click) d_click[b] = gscale * invB * lw_clk * wc * dc;
    if (d_order) d_order[b] = gscale * invB * lw_ord * wo * dd;
    if (d_bias) d_bias[b] = gscale * invB * dbb;
  }
  lsum = wave_sum(lsum);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_red[wave] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += s_red[w];
    loss[0] = s * invB;
  }
}

// ------------------------------------------------------------------------------------------ misc
// y[b,t,:] = scale * x[b,t,:] + pos[t,:]   (TransformerModel.encode: enc *= d_model**0.5; enc += P[0:T])
template <typename T>
__global__ __launch_bounds__(256) void scale_add_pos_kernel(long long n, int Tlen, int d, const T* __restrict__ x, float scale,
                                                            const float* __restrict__ pos, T* __restrict__ y) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % d);
  const int t = (int)((i / d) % Tlen);
  stf<T>(y + i, scale * ldf<T>(x + i) + (pos ? pos[(long long)t * d + c] : 0.f));
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(long long n, const T* __restrict__ x, T* __restrict__ y, uint32_t seed, uint32_t thr24,
                                                      float inv_keep) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  stf<T>(y + i, dmt_drop_keep(seed, (uint32_t)i, thr24) ? ldf<T>(x + i) * inv_keep : 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(long long rows, long long cols, const T* __restrict__ dy, long long lddy,
                                                       const T* __restrict__ y, long long ldy, T* __restrict__ dz, long long lddz) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols) return;
  const long long r = i / cols, c = i - r * cols;
  const float g = ldf<T>(dy + r * lddy + c);
  stf<T>(dz + r * lddz + c, (ldf<T>(y + r * ldy + c) > 0.f) ? g : 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(long long rows, long long cols, const T* __restrict__ x, long long ldx,
                                                     float scale, float* __restrict__ out, int rows_per_block) {
  const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  const long long r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  float s = 0.f;
  for (long long r = r0; r < r1; ++r) s += ldf<T>(x + r * ldx + c);
  atomicAdd(out + c, s * scale);
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_drop_kernel(long long rows, long long cols, const T* __restrict__ x, float scale,
                                                          float* __restrict__ out, int rows_per_block, uint32_t seed, uint32_t thr24) {
  const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  const long long r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  float s = 0.f;
  for (long long r = r0; r < r1; ++r)
    if (dmt_drop_keep(seed, (uint32_t)(r * cols + c), thr24)) s += ldf<T>(x + r * cols + c);
  atomicAdd(out + c, s * scale);
}

// ORDERED column sums (bit-reproducible: the order of every addition is fixed by the shape alone): a block owns 64 columns, wavefront w
// sums the rows r = w (mod 4) in increasing order -- eight independent row loads in flight, added in row order -- and the four partial
// sums are combined as ((p0 + p1) + p2) + p3; one plain read-add-write per column (no other block touches it).  The one-block-per-
// column-strip-over-ALL-rows form this replaces was latency-bound: 1.8 ms for the 4096 x 16000 position gradient of the E64 step.
template <typename T, bool DROP>
__global__ __launch_bounds__(256) void colsum_fixed_kernel(long long rows, long long cols, const T* __restrict__ x, long long ldx, float scale,
                                                           float* __restrict__ out, uint32_t seed, uint32_t thr24) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long c = (long long)blockIdx.x * 64 + lane;
  const long long cc = c < cols ? c : cols - 1;            // (branch-free loads; the column is masked at the store)
  float s = 0.f;
  for (long long r0 = wave; r0 < rows; r0 += 32) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long long r = r0 + 4 * i;
      const long long rr = r < rows ? r : rows - 1;
      v[i] = ldf<T>(x + rr * ldx + cc);
      if (DROP) { if (!dmt_drop_keep(seed, (uint32_t)(rr * cols + cc), thr24)) v[i] = 0.f; }
      if (r >= rows) v[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < cols) out[c] += (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) * scale;
}

// bf16, cols % 8 == 0: a block owns 512 columns (64 lanes x 16 bytes); its 4 waves take interleaved rows, their partial sums
// meet in LDS and leave as ONE atomic per column per block
__global__ __launch_bounds__(256) void colsum_drop_v8_kernel(long long rows, long long cols, const bf16_t* __restrict__ x, float scale,
                                                             float* __restrict__ out, int rows_per_block, uint32_t seed, uint32_t thr24) {
  __shared__ float part[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long c = ((long long)blockIdx.x * 64 + lane) * 8;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  const long long r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < cols) {
    for (long long r = r0 + wave; r < r1; r += 4) {
      const uint4 v = *(const uint4*)(x + r * cols + c);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      const uint32_t base = (uint32_t)(r * cols + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (dmt_drop_keep(seed, base + 2 * i, thr24)) s[2 * i] += __uint_as_float(w[i] << 16);
        if (dmt_drop_keep(seed, base + 2 * i + 1, thr24)) s[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) part[wave][i * 64 + lane] = s[i];     // [i][lane]: conflict-free
  __syncthreads();
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int i = e >> 6, l = e & 63;
    const long long cc = ((long long)blockIdx.x * 64 + l) * 8 + i;
    if (cc < cols) atomicAdd(out + cc, (part[0][e] + part[1][e] + part[2][e] + part[3][e]) * scale);
  }
}

// Learned-position gradient of a PACKED sequence (include/dmt_hip.h "PACKED ROWS"): out[t, c] += scale * sum over the examples b with
// lens[b] > t of mask((b * T + t) * d + c) * x[row_off[b] + t, c] -- the dense kernel above with the rows looked up instead of strided;
// the dropout index is the DENSE one, so both layouts draw the same mask.  bf16, d % 8 == 0.  Workgroup (t, chunk of examples): the
// chunk's row numbers (or -1) are staged in LDS once (coalesced reads of lens / row_off), then a lane owns 8 columns, the four wavefronts
// take interleaved examples with eight row requests in flight each; partial sums meet in LDS and leave as one atomic per column
// (ordered: one chunk, plain add, fixed order).
constexpr int CSP_STAGE = 1024;
__global__ __launch_bounds__(256) void colsum_packed_kernel(int B, int T, int d, const bf16_t* __restrict__ x, const int* __restrict__ row_off,
                                                            const int* __restrict__ lens, float scale, float* __restrict__ out, int ex_per_block,
                                                            uint32_t seed, uint32_t thr24, int drop_on, int atomic) {
  __shared__ float part[4][512];
  __shared__ int s_row[CSP_STAGE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x;
  const int b0 = blockIdx.y * ex_per_block;
  const int b1 = (b0 + ex_per_block < B) ? b0 + ex_per_block : B;
  const int nch = d / 8;
  for (int c0 = 0; c0 < nch; c0 += 64) {
    const int ch = c0 + lane;
    const int chc = ch < nch ? ch : nch - 1;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int bs = b0; bs < b1; bs += CSP_STAGE) {
      const int nb = (b1 - bs) < CSP_STAGE ? (b1 - bs) : CSP_STAGE;
      __syncthreads();
      for (int i = threadIdx.x; i < nb; i += 256) s_row[i] = lens[bs + i] > t ? row_off[bs + i] + t : -1;
      __syncthreads();
      for (int i0 = wave; i0 < nb; i0 += 32) {
        uint4 v[8];
        int rw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + 4 * u;
          rw[u] = i < nb ? s_row[i] : -1;
          v[u] = *reinterpret_cast<const uint4*>(x + (long long)(rw[u] >= 0 ? rw[u] : 0) * d + chc * 8);      // (branch-free: row 0 when there is none)
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (rw[u] < 0) continue;                     // (wave-uniform)
          const int b = bs + i0 + 4 * u;
          const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
          const uint32_t base = (uint32_t)(((long long)b * T + t) * d + chc * 8);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (!drop_on || dmt_drop_keep(seed, base + 2 * i, thr24)) s[2 * i] += __uint_as_float(w[i] << 16);
            if (!drop_on || dmt_drop_keep(seed, base + 2 * i + 1, thr24)) s[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) part[wave][i * 64 + lane] = s[i];
    __syncthreads();
    for (int e = threadIdx.x; e < 512; e += 256) {
      const int i = e >> 6, l = e & 63;
      const int col = (c0 + l) * 8 + i;
      if (c0 + l < nch) {
        const float v2 = (((part[0][e] + part[1][e]) + part[2][e]) + part[3][e]) * scale;
        if (atomic) atomicAdd(out + (long long)t * d + col, v2); else out[(long long)t * d + col] += v2;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(long long n, const float* __restrict__ src, bf16_t* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = f2bf(src[i]);
}

// 32x32 LDS-tiled transpose: dst_plain[r][c] = dst_t[c][r] = bf16(src[r][c])
__global__ __launch_bounds__(256) void cast_transpose_kernel(int rows, int cols, const float* __restrict__ src, long long ld,
                                                             bf16_t* __restrict__ dp, long long ldp, bf16_t* __restrict__ dt, long long ldt) {
  __shared__ bf16_t tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 8 rows per pass
  for (int rr = ty; rr < 32; rr += 8) {
    const int r = r0 + rr, c = c0 + tx;
    bf16_t v = 0;
    if (r < rows && c < cols) {
      v = f2bf(src[(long long)r * ld + c]);
      if (dp) dp[(long long)r * ldp + c] = v;
    }
    tile[rr][tx] = v;
  }
  __syncthreads();
  if (dt) {
    for (int cc = ty; cc < 32; cc += 8) {
      const int c = c0 + cc, r = r0 + tx;
      if (r < rows && c < cols) dt[(long long)c * ldt + r] = tile[tx][cc];
    }
  }
}

// every 2-D weight of the model in ONE launch: block -> (job, 32x32 tile) through the jobs' tile prefix (n_jobs is a few dozen)
__global__ __launch_bounds__(256) void cast_transpose_batched_kernel(int n_jobs, const dmt_cast_job* __restrict__ jobs) {
  __shared__ bf16_t tile[32][33];
  int j = 0;
  while (j + 1 < n_jobs && (int)blockIdx.x >= jobs[j + 1].tile_begin) ++j;
  const dmt_cast_job jb = jobs[j];
  const int t = (int)blockIdx.x - jb.tile_begin;
  const int c0 = (t % jb.tiles_x) * 32, r0 = (t / jb.tiles_x) * 32;
  const float* __restrict__ src = jb.src;
  bf16_t* __restrict__ dp = (bf16_t*)jb.dst_plain;
  bf16_t* __restrict__ dt = (bf16_t*)jb.dst_t;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int rr = ty; rr < 32; rr += 8) {
    const int r = r0 + rr, c = c0 + tx;
    bf16_t v = 0;
    if (r < jb.rows && c < jb.cols) {
      v = f2bf(src[(long long)r * jb.ld_src + c]);
      if (dp) dp[(long long)r * jb.ld_plain + c] = v;
    }
    tile[rr][tx] = v;
  }
  __syncthreads();
  if (dt) {
    for (int cc = ty; cc < 32; cc += 8) {
      const int c = c0 + cc, r = r0 + tx;
      if (r < jb.rows && c < jb.cols) dt[(long long)c * jb.ld_t + r] = tile[tx][cc];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Row softmax of the UNFUSED attention used for sequences longer than 64 keys (BASELINE "long-seq variant", L = 200): the
// scores S = Q K^T and the products P V, dO V^T, dS K, dS^T Q, P^T dO run as batched dmt_gemm launches, these two kernels do
// what lies between them (TransformerModel_util.py:11-56, 80-108): scale, key mask (padding value -2^32+1), softmax over
// keys, query mask applied AFTER the softmax, attention-weight dropout with the library's counter mask (same flat index
// ((b*H + h)*Tq + q)*Tk + k as the fused kernels).  One wavefront per (example, head, query) row; rows are a few hundred
// bytes, the three passes over them stay in cache.
constexpr float SM_PADDING_NUM = -4294967295.0f;   // -2**32 + 1

template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(int B, int H, int Tq, int Tk, T* __restrict__ S, long long ld,
                                                          const int* __restrict__ q_lens, const int* __restrict__ k_lens, float scale,
                                                          int drop_on, uint32_t seed, uint32_t thr24, float inv_keep, T* __restrict__ P, int causal) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)B * H * Tq) return;
  const int q = (int)(row % Tq);
  const int b = (int)(row / ((long long)H * Tq));
  int klen = k_lens ? k_lens[b] : Tk;
  klen = klen < 0 ? 0 : (klen > Tk ? Tk : klen);
  // future blinding (TransformerModel_util.py:34-36, 99-105: mask(type="future") after the key mask): keys k > q get the padding value too
  if (causal && klen > q + 1) klen = q + 1;
  const int qlen = q_lens ? q_lens[b] : Tq;
  T* s = S + row * ld;
  T* p = P + row * ld;
  float m = -3.0e38f;
  for (int k = lane; k < Tk; k += 64) {
    const float x = (k < klen) ? ldf<T>(s + k) * scale : SM_PADDING_NUM;
    m = fmaxf(m, x);
  }
  m = wave_max(m);
  float sum = 0.f;
  for (int k = lane; k < Tk; k += 64) {
    const float x = (k < klen) ? ldf<T>(s + k) * scale : SM_PADDING_NUM;
    sum += expf(x - m);
  }
  sum = wave_sum(sum);
  const bool qpad = q >= qlen;
  const uint32_t base = (uint32_t)(row * Tk);
  for (int k = lane; k < Tk; k += 64) {
    const float x = (k < klen) ? ldf<T>(s + k) * scale : SM_PADDING_NUM;
    float pv = expf(x - m) / sum;
    if (qpad) pv = SM_PADDING_NUM;                                      // query mask after the softmax (reference behaviour)
    stf<T>(p + k, pv);
    if (drop_on) pv = dmt_drop_keep(seed, base + k, thr24) ? pv * inv_keep : 0.f;
    stf<T>(s + k, pv);                                                  // the A operand of P.V
  }
  for (int k = Tk + lane; k < ld; k += 64) { stf<T>(s + k, 0.f); stf<T>(p + k, 0.f); }   // row padding up to the leading dimension
}

// dP (gradient w.r.t. the dropped weights, in place -> dS), P (saved) -> Pd (dropped weights, the A^T operand of dV)
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(int B, int H, int Tq, int Tk, const T* __restrict__ P, T* __restrict__ dPS,
                                                          T* __restrict__ Pd, long long ld, const int* __restrict__ q_lens,
                                                          const int* __restrict__ k_lens, float scale, int drop_on, uint32_t seed,
                                                          uint32_t thr24, float inv_keep, int causal) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)B * H * Tq) return;
  const int q = (int)(row % Tq);
  const int b = (int)(row / ((long long)H * Tq));
  int klen = k_lens ? k_lens[b] : Tk;
  klen = klen < 0 ? 0 : (klen > Tk ? Tk : klen);
  if (causal && klen > q + 1) klen = q + 1;
  const int qlen = q_lens ? q_lens[b] : Tq;
  const bool qpad = q >= qlen;
  const T* p = P + row * ld;
  T* g = dPS + row * ld;
  T* pd = Pd + row * ld;
  const uint32_t base = (uint32_t)(row * Tk);
  float dot = 0.f;
  for (int k = lane; k < Tk; k += 64) {
    float gq = ldf<T>(g + k);
    if (drop_on) gq = dmt_drop_keep(seed, base + k, thr24) ? gq * inv_keep : 0.f;     // gradient w.r.t. the pre-dropout weights
    dot += ldf<T>(p + k) * gq;
  }
  dot = wave_sum(dot);
  for (int k = lane; k < Tk; k += 64) {
    const float pv = ldf<T>(p + k);
    float gq = ldf<T>(g + k);
    bool keep = true;
    if (drop_on) { keep = dmt_drop_keep(seed, base + k, thr24); gq = keep ? gq * inv_keep : 0.f; }
    float ds = (k < klen && !qpad) ? pv * (gq - dot) * scale : 0.f;     // no gradient into masked keys / through constant rows
    stf<T>(g + k, ds);
    stf<T>(pd + k, drop_on ? (keep ? pv * inv_keep : 0.f) : pv);
  }
  for (int k = Tk + lane; k < ld; k += 64) { stf<T>(g + k, 0.f); stf<T>(pd + k, 0.f); }
}

__global__ __launch_bounds__(256) void auc_hist_kernel(int B, const float* __restrict__ pred, const float* __restrict__ label, int n_thr,
                                                       unsigned long long* __restrict__ hist) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float p = pred[b];
  // thresholds: t_0 = -1e-7, t_i = i/(n_thr-1) (0 < i < n_thr-1), t_last = 1 + 1e-7;  bin = #{i : p > t_i}
  int lo = 0;
  if (p > -1e-7f) {
    lo = 1;
    const float step = 1.0f / (float)(n_thr - 1);
    int guess = (int)floorf(p / step);
    guess = guess < 0 ? 0 : (guess > n_thr - 2 ? n_thr - 2 : guess);
    // exact comparison against the float32 thresholds around the guess
    int cnt = 1;
    for (int i = (guess > 2 ? guess - 2 : 1); i <= n_thr - 2 && i <= guess + 2; ++i) {
      if (i < 1) continue;
      const float ti = (float)((double)i / (double)(n_thr - 1));
      if (p > ti) cnt = i + 1;
    }
    lo = cnt;
    if (p > 1.0f + 1e-7f) lo = n_thr;
  }
  const int pos = label[b] > 0.5f ? 1 : 0;
  atomicAdd(&hist[(long long)pos * (n_thr + 1) + lo], 1ULL);
}


// tf.metrics.precision / recall (run_dnn.py:221-227, 230-238): predictions and labels are cast to bool (non-zero);
// counts = [true positives, false positives, false negatives, true negatives]
__global__ __launch_bounds__(256) void confusion_kernel(int B, const float* __restrict__ pred, const float* __restrict__ label, float thr,
                                                        unsigned long long* __restrict__ counts) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  int cls = -1;
  if (b < B) {
    const bool p = pred[b] > thr, y = label[b] != 0.f;
    cls = p ? (y ? 0 : 1) : (y ? 2 : 3);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const unsigned long long m = __ballot(cls == c);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&counts[c], (unsigned long long)__popcll(m));
  }
}

// sum over the DISTINCT ids of one feature of ||E[id]||^2 / 2  (tf.nn.l2_loss(tf.gather(E, tf.unique(ids)))):
// a row is counted by the first entry that sets its bit in `seen` (rows / 32 words, zeroed by the caller)
__global__ __launch_bounds__(256) void l2_unique_kernel(int B, int T, const int* __restrict__ idx, const int* __restrict__ lens,
                                                        const float* __restrict__ table, int rows, int dim, unsigned* __restrict__ seen,
                                                        float* __restrict__ out, int* __restrict__ mult) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  if (e < (long long)B * T) {
    const int b = (int)(e / T), t = (int)(e % T);
    if (t < lens[b]) {
      const int r = idx[e];
      if (r >= 0 && r < rows) {
        const unsigned bit = 1u << (r & 31);
        const unsigned old = atomicOr(&seen[r >> 5], bit);
        if (!(old & bit)) {
          const float* row = table + (long long)r * dim;
          for (int c = 0; c < dim; ++c) acc += row[c] * row[c];
          if (mult) atomicAdd(&mult[r], 1);        // one more embedding_list entry whose batch holds this row (the term's gradient)
        }
      }
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0 && acc != 0.f) atomicAdd(out, 0.5f * acc);
}

template <typename T, typename F>
int ln_dispatch(int d, F&& f) {
  const int ne = (d + 63) / 64;
  if (ne <= 2) return f(std::integral_constant<int, 2>());
  if (ne <= 5) return f(std::integral_constant<int, 5>());
  if (ne <= 8) return f(std::integral_constant<int, 8>());
  if (ne <= 16) return f(std::integral_constant<int, 16>());
  return -1;
}

}  // namespace

extern "C" int dmt_ln_fwd(int32_t dtype, int64_t rows, int32_t d, const void* x, int64_t ldx, const float* gamma,
                          const float* beta, float eps, void* y, int64_t ldy, float* stats, void* stream) {
  DMT_CHECK_ARG(rows > 0 && d > 0 && x && y && gamma && beta, "dmt_ln_fwd: bad argument");
  DMT_CHECK_ARG(dtype == DMT_F32 || dtype == DMT_BF16, "dmt_ln_fwd: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  long long nb = cdiv64(rows, 4);
  if (nb > 4096) nb = 4096;
  if (ln_v8_ok(dtype, d, x, ldx, y, ldy, nullptr, 0)) {
    hipLaunchKernelGGL(ln_fwd_v8_kernel, dim3((unsigned)nb), dim3(256), 0, st, (long long)rows, d, (const bf16_t*)x, (long long)ldx, gamma, beta,
                       eps, (bf16_t*)y, (long long)ldy, stats);
    DMT_CHECK_LAUNCH("dmt_ln_fwd(v8)");
    return DMT_OK;
  }
  int rc;
  if (dtype == DMT_F32)
    rc = ln_dispatch<float>(d, [&](auto ne) {
      hipLaunchKernelGGL((ln_fwd_kernel<float, decltype(ne)::value>), dim3((unsigned)nb), dim3(256), 0, st, (long long)rows, d,
                         (const float*)x, (long long)ldx, gamma, beta, eps, (float*)y, (long long)ldy, stats);
      return 0; });
  else
    rc = ln_dispatch<bf16_t>(d, [&](auto ne) {
      hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, decltype(ne)::value>), dim3((unsigned)nb), dim3(256), 0, st, (long long)rows, d,
                         (const bf16_t*)x, (long long)ldx, gamma, beta, eps, (bf16_t*)y, (long long)ldy, stats);
      return 0; });
  if (rc != 0) { dmt_set_error("dmt_ln_fwd: d=%d > 1024 unsupported", d); return DMT_ERR_UNSUPPORTED; }
  DMT_CHECK_LAUNCH("dmt_ln_fwd");
  return DMT_OK;
}

extern "C" int32_t dmt_ln_bwd_partials(int64_t rows) {
  long long nb = cdiv64(rows, 4);
  if (nb > 1024) nb = 1024;
  return (int32_t)(nb < 1 ? 1 : nb);
}

extern "C" int dmt_ln_bwd(int32_t dtype, int64_t rows, int32_t d, const void* x, int64_t ldx, const float* gamma,
                          const float* stats, const void* dy, int64_t lddy, void* dx, int64_t lddx, float* dgamma,
                          float* dbeta, float* partials, void* stream) {
  DMT_CHECK_ARG(rows > 0 && d > 0 && x && gamma && stats && dy && dx && partials, "dmt_ln_bwd: bad argument");
  DMT_CHECK_ARG((dgamma != nullptr) == (dbeta != nullptr), "dmt_ln_bwd: dgamma and dbeta are given together or not at all");
  DMT_CHECK_ARG(dtype == DMT_F32 || dtype == DMT_BF16, "dmt_ln_bwd: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  const int nb = dmt_ln_bwd_partials(rows);
  const bool finish = dgamma != nullptr;        // null: the caller reduces `partials` later (dmt_ln_bwd_finish_batched)
  if (ln_v8_ok(dtype, d, x, ldx, dy, lddy, dx, lddx)) {
    hipLaunchKernelGGL(ln_bwd_v8_kernel, dim3(nb), dim3(256), 0, st, (long long)rows, d, (const bf16_t*)x, (long long)ldx, gamma, stats,
                       (const bf16_t*)dy, (long long)lddy, (bf16_t*)dx, (long long)lddx, partials);
    if (finish) hipLaunchKernelGGL(ln_bwd_finish_kernel, dim3((2 * d + 3) / 4), dim3(256), 0, st, nb, d, partials, dgamma, dbeta);
    DMT_CHECK_LAUNCH("dmt_ln_bwd(v8)");
    return DMT_OK;
  }
  int rc;
  if (dtype == DMT_F32)
    rc = ln_dispatch<float>(d, [&](auto ne) {
      hipLaunchKernelGGL((ln_bwd_kernel<float, decltype(ne)::value>), dim3(nb), dim3(256), 0, st, (long long)rows, d, (const float*)x,
                         (long long)ldx, gamma, stats, (const float*)dy, (long long)lddy, (float*)dx, (long long)lddx, partials);
      return 0; });
  else
    rc = ln_dispatch<bf16_t>(d, [&](auto ne) {
      hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, decltype(ne)::value>), dim3(nb), dim3(256), 0, st, (long long)rows, d, (const bf16_t*)x,
                         (long long)ldx, gamma, stats, (const bf16_t*)dy, (long long)lddy, (bf16_t*)dx, (long long)lddx, partials);
      return 0; });
  if (rc != 0) { dmt_set_error("dmt_ln_bwd: d=%d > 1024 unsupported", d); return DMT_ERR_UNSUPPORTED; }
  if (finish) hipLaunchKernelGGL(ln_bwd_finish_kernel, dim3((2 * d + 3) / 4), dim3(256), 0, st, nb, d, partials, dgamma, dbeta);
  DMT_CHECK_LAUNCH("dmt_ln_bwd");
  return DMT_OK;
}

extern "C" int dmt_ln_bwd_finish_batched(const dmt_ln_finish_job* jobs, int32_t n, void* stream) {
  DMT_CHECK_ARG(jobs && n > 0 && n <= DMT_LN_FINISH_MAX, "dmt_ln_bwd_finish_batched: 1 .. DMT_LN_FINISH_MAX jobs");
  LnFinishJobs pack;
  int dmax = 0, ng = 0, filled = 0;
  bool taken[DMT_LN_FINISH_MAX] = {false};
  for (int i = 0; i < n; ++i) {
    DMT_CHECK_ARG(jobs[i].partials && jobs[i].dgamma && jobs[i].dbeta && jobs[i].n_part > 0 && jobs[i].d > 0, "dmt_ln_bwd_finish_batched: bad job");
    if (jobs[i].d > dmax) dmax = jobs[i].d;
  }
  for (int i = 0; i < n; ++i) {              // groups of equal destination, in order of first appearance
    if (taken[i]) continue;
    pack.first[ng++] = filled;
    for (int k = i; k < n; ++k)
      if (!taken[k] && jobs[k].dgamma == jobs[i].dgamma) {
        DMT_CHECK_ARG(jobs[k].dbeta == jobs[i].dbeta && jobs[k].d == jobs[i].d, "dmt_ln_bwd_finish_batched: jobs share dgamma but not dbeta / d");
        pack.job[filled++] = jobs[k];
        taken[k] = true;
      }
  }
  pack.first[ng] = filled;
  pack.n_groups = ng;
  hipLaunchKernelGGL(ln_bwd_finish_jobs_kernel, dim3((2 * dmax + 3) / 4, ng), dim3(256), 0, (hipStream_t)stream, pack);
  DMT_CHECK_LAUNCH("dmt_ln_bwd_finish_batched");
  return DMT_OK;
}

extern "C" int dmt_mmoe_mix_fwd(int32_t dtype, int32_t B, int32_t E, int32_t U, int32_t n_tasks, const void* expert,
                                int64_t ld_expert, const void* glogit, int64_t ld_glogit, float* gates, void* mix, void* stream) {
  DMT_CHECK_ARG(B > 0 && E > 0 && U > 0 && n_tasks > 0 && n_tasks <= 4 && n_tasks * E <= 16, "dmt_mmoe_mix_fwd: bad dims (n_tasks*E <= 16)");
  DMT_CHECK_ARG(expert && glogit && gates && mix, "dmt_mmoe_mix_fwd: null argument");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DMT_F32)
    hipLaunchKernelGGL((mix_fwd_kernel<float>), dim3(B), dim3(128), 0, st, B, E, U, n_tasks, (const float*)expert, (long long)ld_expert,
                       (const float*)glogit, (long long)ld_glogit, gates, (float*)mix);
  else
    hipLaunchKernelGGL((mix_fwd_kernel<bf16_t>), dim3(B), dim3(128), 0, st, B, E, U, n_tasks, (const bf16_t*)expert, (long long)ld_expert,
                       (const bf16_t*)glogit, (long long)ld_glogit, gates, (bf16_t*)mix);
  DMT_CHECK_LAUNCH("dmt_mmoe_mix_fwd");
  return DMT_OK;
}

extern "C" int dmt_mmoe_mix_bwd(int32_t dtype, int32_t B, int32_t E, int32_t U, int32_t n_tasks, const void* expert,
                                int64_t ld_expert, const float* gates, const void* dmix, void* dexpert, int64_t ld_dexpert,
                                void* dglogit, int64_t ld_dglogit, int32_t relu_mask, void* stream) {
  DMT_CHECK_ARG(B > 0 && E > 0 && U > 0 && n_tasks > 0 && n_tasks <= 4 && n_tasks * E <= 16, "dmt_mmoe_mix_bwd: bad dims (n_tasks*E <= 16)");
  DMT_CHECK_ARG(expert && gates && dmix && dexpert && dglogit, "dmt_mmoe_mix_bwd: null argument");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DMT_F32)
    hipLaunchKernelGGL((mix_bwd_kernel<float>), dim3(B), dim3(128), 0, st, B, E, U, n_tasks, (const float*)expert, (long long)ld_expert, gates,
                       (const float*)dmix, (float*)dexpert, (long long)ld_dexpert, (float*)dglogit, (long long)ld_dglogit, relu_mask);
  else
    hipLaunchKernelGGL((mix_bwd_kernel<bf16_t>), dim3(B), dim3(128), 0, st, B, E, U, n_tasks, (const bf16_t*)expert, (long long)ld_expert, gates,
                       (const bf16_t*)dmix, (bf16_t*)dexpert, (long long)ld_dexpert, (bf16_t*)dglogit, (long long)ld_dglogit, relu_mask);
  DMT_CHECK_LAUNCH("dmt_mmoe_mix_bwd");
  return DMT_OK;
}

extern "C" int dmt_loss_unbias(int32_t B, const float* click, const float* order, const float* ybias, const float* mask5,
                               const float* w_ctr, const float* w_ecvr, float lw_clk, float lw_ord, int32_t method,
                               int32_t ctr_rel, float grad_scale, float* loss, float* p_ctr, float* p_cvr, float* d_click,
                               float* d_order, float* d_bias, void* stream) {
  DMT_CHECK_ARG(B > 0 && click && order && ybias && mask5 && w_ctr && w_ecvr && loss, "dmt_loss_unbias: null argument");
  hipLaunchKernelGGL(loss_unbias_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, B, click, order, ybias, mask5, w_ctr, w_ecvr,
                     lw_clk, lw_ord, method, ctr_rel, grad_scale, loss, p_ctr, p_cvr, d_click, d_order, d_bias);
  DMT_CHECK_LAUNCH("dmt_loss_unbias");
  return DMT_OK;
}

extern "C" int dmt_scale_add_pos(int32_t dtype, int64_t B, int32_t T, int32_t d, const void* x, float scale, const float* pos,
                                 void* y, void* stream) {
  DMT_CHECK_ARG(B > 0 && T > 0 && d > 0 && x && y, "dmt_scale_add_pos: bad argument");
  const long long n = (long long)B * T * d;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DMT_F32)
    hipLaunchKernelGGL((scale_add_pos_kernel<float>), dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, n, T, d, (const float*)x, scale, pos, (float*)y);
  else
    hipLaunchKernelGGL((scale_add_pos_kernel<bf16_t>), dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, n, T, d, (const bf16_t*)x, scale, pos, (bf16_t*)y);
  DMT_CHECK_LAUNCH("dmt_scale_add_pos");
  return DMT_OK;
}

extern "C" int dmt_dropout(int32_t dtype, int64_t n, const void* x, void* y, uint32_t seed, float keep_prob, void* stream) {
  DMT_CHECK_ARG(n > 0 && n < 0xFFFFFFFFll && x && y, "dmt_dropout: bad argument (n must fit 32 bits)");
  DMT_CHECK_ARG(keep_prob > 0.f && keep_prob <= 1.f, "dmt_dropout: keep_prob must be in (0, 1]");
  const uint32_t thr = (uint32_t)(keep_prob * 16777216.0f);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DMT_F32)
    hipLaunchKernelGGL((dropout_kernel<float>), dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, (long long)n, (const float*)x, (float*)y, seed, thr, 1.f / keep_prob);
  else
    hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, (long long)n, (const bf16_t*)x, (bf16_t*)y, seed, thr, 1.f / keep_prob);
  DMT_CHECK_LAUNCH("dmt_dropout");
  return DMT_OK;
}

extern "C" int dmt_relu_bwd(int32_t dtype, int64_t rows, int64_t cols, const void* dy, int64_t lddy, const void* y, int64_t ldy,
                            void* dz, int64_t lddz, void* stream) {
  DMT_CHECK_ARG(rows > 0 && cols > 0 && dy && y && dz, "dmt_relu_bwd: bad argument");
  const unsigned nb = (unsigned)cdiv64(rows * cols, 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DMT_F32)
    hipLaunchKernelGGL((relu_bwd_kernel<float>), dim3(nb), dim3(256), 0, st, (long long)rows, (long long)cols, (const float*)dy,
                       (long long)lddy, (const float*)y, (long long)ldy, (float*)dz, (long long)lddz);
  else
    hipLaunchKernelGGL((relu_bwd_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (long long)rows, (long long)cols, (const bf16_t*)dy,
                       (long long)lddy, (const bf16_t*)y, (long long)ldy, (bf16_t*)dz, (long long)lddz);
  DMT_CHECK_LAUNCH("dmt_relu_bwd");
  return DMT_OK;
}

extern "C" int dmt_colsum_rows_packed(int32_t dtype, int32_t B, int32_t T, int32_t d, const void* x, const int32_t* row_off, const int32_t* lens,
                                      float scale, float* out, uint32_t seed, float keep_prob, int32_t ordered, void* stream) {
  DMT_CHECK_ARG(dtype == DMT_BF16 && B > 0 && T > 0 && d > 0 && d % 8 == 0 && x && row_off && lens && out, "dmt_colsum_rows_packed: bf16 rows, d %% 8 == 0, non-null arguments");
  DMT_CHECK_ARG(((uintptr_t)x & 15) == 0, "dmt_colsum_rows_packed: rows must be 16-byte aligned");
  DMT_CHECK_ARG((long long)B * T * d < (1ll << 32), "dmt_colsum_rows_packed: dropout counter range");
  const bool drop = keep_prob > 0.f && keep_prob < 1.f;
  const uint32_t thr = drop ? (uint32_t)(keep_prob * 16777216.0f) : 0u;
  const int epb = ordered ? B : (B + 15) / 16;
  dim3 grid((unsigned)T, (unsigned)((B + epb - 1) / epb));
  hipLaunchKernelGGL(colsum_packed_kernel, grid, dim3(256), 0, (hipStream_t)stream, B, T, d, (const bf16_t*)x, row_off, lens, drop ? scale / keep_prob : scale, out,
                     epb, seed, thr, drop ? 1 : 0, ordered ? 0 : 1);
  DMT_CHECK_LAUNCH("dmt_colsum_rows_packed");
  return DMT_OK;
}

extern "C" int dmt_colsum_drop(int32_t dtype, int64_t rows, int64_t cols, const void* x, float scale, float* out, uint32_t seed,
                               float keep_prob, int32_t ordered, void* stream) {
  DMT_CHECK_ARG(rows > 0 && cols > 0 && x && out, "dmt_colsum_drop: bad argument");
  DMT_CHECK_ARG(keep_prob > 0.f && keep_prob <= 1.f, "dmt_colsum_drop: keep_prob must be in (0, 1]");
  const bool det = ordered != 0;
  const int rpb = 64;
  dim3 grid((unsigned)cdiv64(cols, 256), (unsigned)cdiv64(rows, rpb));
  DMT_CHECK_ARG(grid.y <= 65535, "dmt_colsum_drop: too many rows");
  const uint32_t thr = (uint32_t)(keep_prob * 16777216.0f);
  hipStream_t st = (hipStream_t)stream;
  if (det) {
    const unsigned nb = (unsigned)cdiv64(cols, 64);
    if (dtype == DMT_F32)
      hipLaunchKernelGGL((colsum_fixed_kernel<float, true>), dim3(nb), dim3(256), 0, st, (long long)rows, (long long)cols, (const float*)x, (long long)cols, scale / keep_prob, out, seed, thr);
    else
      hipLaunchKernelGGL((colsum_fixed_kernel<bf16_t, true>), dim3(nb), dim3(256), 0, st, (long long)rows, (long long)cols, (const bf16_t*)x, (long long)cols, scale / keep_prob, out, seed, thr);
    DMT_CHECK_LAUNCH("dmt_colsum_drop(ordered)");
    return DMT_OK;
  }
  if (!det && dtype == DMT_BF16 && cols % 8 == 0 && ((uintptr_t)x & 15) == 0 && rows >= 256) {
    const int rpb8 = 64;
    dim3 g8((unsigned)cdiv64(cols / 8, 64), (unsigned)cdiv64(rows, rpb8));
    DMT_CHECK_ARG(g8.y <= 65535, "dmt_colsum_drop: too many rows");
    hipLaunchKernelGGL(colsum_drop_v8_kernel, g8, dim3(256), 0, st, (long long)rows, (long long)cols, (const bf16_t*)x, scale / keep_prob, out, rpb8, seed, thr);
    DMT_CHECK_LAUNCH("dmt_colsum_drop");
    return DMT_OK;
  }
  if (dtype == DMT_F32)
    hipLaunchKernelGGL((colsum_drop_kernel<float>), grid, dim3(256), 0, st, (long long)rows, (long long)cols, (const float*)x, scale / keep_prob, out, rpb, seed, thr);
  else
    hipLaunchKernelGGL((colsum_drop_kernel<bf16_t>), grid, dim3(256), 0, st, (long long)rows, (long long)cols, (const bf16_t*)x, scale / keep_prob, out, rpb, seed, thr);
  DMT_CHECK_LAUNCH("dmt_colsum_drop");
  return DMT_OK;
}

extern "C" int dmt_colsum(int32_t dtype, int64_t rows, int64_t cols, const void* x, int64_t ldx, float scale, float* out,
                          int32_t ordered, void* stream) {
  DMT_CHECK_ARG(rows > 0 && cols > 0 && x && out, "dmt_colsum: bad argument");
  const int rpb = 64;
  dim3 grid((unsigned)cdiv64(cols, 256), (unsigned)cdiv64(rows, rpb));
  DMT_CHECK_ARG(grid.y <= 65535, "dmt_colsum: too many rows");
  hipStream_t st = (hipStream_t)stream;
  if (ordered) {
    const unsigned nb = (unsigned)cdiv64(cols, 64);
    if (dtype == DMT_F32)
      hipLaunchKernelGGL((colsum_fixed_kernel<float, false>), dim3(nb), dim3(256), 0, st, (long long)rows, (long long)cols, (const float*)x, (long long)ldx, scale, out, 0u, 0u);
    else
      hipLaunchKernelGGL((colsum_fixed_kernel<bf16_t, false>), dim3(nb), dim3(256), 0, st, (long long)rows, (long long)cols, (const bf16_t*)x, (long long)ldx, scale, out, 0u, 0u);
    DMT_CHECK_LAUNCH("dmt_colsum(ordered)");
    return DMT_OK;
  }
  if (dtype == DMT_F32)
    hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, st, (long long)rows, (long long)cols, (const float*)x, (long long)ldx, scale, out, rpb);
  else
    hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(256), 0, st, (long long)rows, (long long)cols, (const bf16_t*)x, (long long)ldx, scale, out, rpb);
  DMT_CHECK_LAUNCH("dmt_colsum");
  return DMT_OK;
}

extern "C" int dmt_cast_bf16(int64_t n, const float* src, void* dst, void* stream) {
  DMT_CHECK_ARG(n > 0 && src && dst, "dmt_cast_bf16: bad argument");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, (long long)n, src, (bf16_t*)dst);
  DMT_CHECK_LAUNCH("dmt_cast_bf16");
  return DMT_OK;
}

extern "C" int dmt_cast_transpose_bf16(int32_t rows, int32_t cols, const float* src, int64_t ld_src, void* dst_plain,
                                       int64_t ld_plain, void* dst_t, int64_t ld_t, void* stream) {
  DMT_CHECK_ARG(rows > 0 && cols > 0 && src && (dst_plain || dst_t), "dmt_cast_transpose_bf16: bad argument");
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  hipLaunchKernelGGL(cast_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, rows, cols, src, (long long)ld_src,
                     (bf16_t*)dst_plain, (long long)ld_plain, (bf16_t*)dst_t, (long long)ld_t);
  DMT_CHECK_LAUNCH("dmt_cast_transpose_bf16");
  return DMT_OK;
}

extern "C" int dmt_cast_transpose_bf16_batched(int32_t n_jobs, const dmt_cast_job* jobs_dev, int32_t total_tiles, void* stream) {
  DMT_CHECK_ARG(n_jobs > 0 && jobs_dev && total_tiles > 0, "dmt_cast_transpose_bf16_batched: bad argument");
  hipLaunchKernelGGL(cast_transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, n_jobs, jobs_dev);
  DMT_CHECK_LAUNCH("dmt_cast_transpose_bf16_batched");
  return DMT_OK;
}

extern "C" int dmt_softmax_fwd(int32_t dtype, int32_t B, int32_t H, int32_t Tq, int32_t Tk, void* S, int64_t ld, const int32_t* q_lens,
                               const int32_t* k_lens, float scale, uint32_t drop_seed, float drop_keep, void* P, int32_t causal, void* stream) {
  DMT_CHECK_ARG(B > 0 && H > 0 && Tq > 0 && Tk > 0 && S && P && ld >= Tk, "dmt_softmax_fwd: bad argument");
  DMT_CHECK_ARG(dtype == DMT_F32 || dtype == DMT_BF16, "dmt_softmax_fwd: bad dtype");
  DMT_CHECK_ARG((long long)B * H * Tq * Tk < 0xFFFFFFFFll, "dmt_softmax_fwd: B*H*Tq*Tk exceeds the 32-bit dropout counter");
  const int drop_on = (drop_keep > 0.f && drop_keep < 1.f) ? 1 : 0;
  const uint32_t thr = (uint32_t)(drop_keep * 16777216.0f);
  const float inv = drop_on ? 1.f / drop_keep : 1.f;
  const unsigned nb = (unsigned)cdiv64((long long)B * H * Tq, 4);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DMT_F32)
    hipLaunchKernelGGL((softmax_fwd_kernel<float>), dim3(nb), dim3(256), 0, st, B, H, Tq, Tk, (float*)S, (long long)ld, q_lens, k_lens, scale, drop_on, drop_seed, thr, inv, (float*)P, causal);
  else
    hipLaunchKernelGGL((softmax_fwd_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, B, H, Tq, Tk, (bf16_t*)S, (long long)ld, q_lens, k_lens, scale, drop_on, drop_seed, thr, inv, (bf16_t*)P, causal);
  DMT_CHECK_LAUNCH("dmt_softmax_fwd");
  return DMT_OK;
}

extern "C" int dmt_softmax_bwd(int32_t dtype, int32_t B, int32_t H, int32_t Tq, int32_t Tk, const void* P, void* dP_dS, void* Pd, int64_t ld,
                               const int32_t* q_lens, const int32_t* k_lens, float scale, uint32_t drop_seed, float drop_keep, int32_t causal, void* stream) {
  DMT_CHECK_ARG(B > 0 && H > 0 && Tq > 0 && Tk > 0 && P && dP_dS && Pd && ld >= Tk, "dmt_softmax_bwd: bad argument");
  DMT_CHECK_ARG(dtype == DMT_F32 || dtype == DMT_BF16, "dmt_softmax_bwd: bad dtype");
  const int drop_on = (drop_keep > 0.f && drop_keep < 1.f) ? 1 : 0;
  const uint32_t thr = (uint32_t)(drop_keep * 16777216.0f);
  const float inv = drop_on ? 1.f / drop_keep : 1.f;
  const unsigned nb = (unsigned)cdiv64((long long)B * H * Tq, 4);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DMT_F32)
    hipLaunchKernelGGL((softmax_bwd_kernel<float>), dim3(nb), dim3(256), 0, st, B, H, Tq, Tk, (const float*)P, (float*)dP_dS, (float*)Pd, (long long)ld, q_lens, k_lens, scale, drop_on, drop_seed, thr, inv, causal);
  else
    hipLaunchKernelGGL((softmax_bwd_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, B, H, Tq, Tk, (const bf16_t*)P, (bf16_t*)dP_dS, (bf16_t*)Pd, (long long)ld, q_lens, k_lens, scale, drop_on, drop_seed, thr, inv, causal);
  DMT_CHECK_LAUNCH("dmt_softmax_bwd");
  return DMT_OK;
}

extern "C" int dmt_auc_hist(int32_t B, const float* pred, const float* label, int32_t n_thr, long long* hist, void* stream) {
  DMT_CHECK_ARG(B > 0 && pred && label && hist && n_thr >= 3, "dmt_auc_hist: bad argument");
  hipLaunchKernelGGL(auc_hist_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, pred, label, n_thr,
                     (unsigned long long*)hist);
  DMT_CHECK_LAUNCH("dmt_auc_hist");
  return DMT_OK;
}

extern "C" int dmt_confusion_counts(int32_t B, const float* pred, const float* label, float threshold, long long* counts, void* stream) {
  DMT_CHECK_ARG(B > 0 && pred && label && counts, "dmt_confusion_counts: bad argument");
  hipLaunchKernelGGL(confusion_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, pred, label, threshold,
                     (unsigned long long*)counts);
  DMT_CHECK_LAUNCH("dmt_confusion_counts");
  return DMT_OK;
}

extern "C" int dmt_l2_unique_rows(int32_t B, int32_t T, const int32_t* idx, const int32_t* lens, const float* table, int32_t rows, int32_t dim,
                                  uint32_t* seen, float* out, void* stream) {
  DMT_CHECK_ARG(B > 0 && T > 0 && idx && lens && table && seen && out && rows > 0 && dim > 0, "dmt_l2_unique_rows: bad argument");
  const long long n = (long long)B * T;
  hipLaunchKernelGGL(l2_unique_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, T, idx, lens, table, rows, dim,
                     seen, out, (int*)nullptr);
  DMT_CHECK_LAUNCH("dmt_l2_unique_rows");
  return DMT_OK;
}

extern "C" int dmt_l2_unique_rows_count(int32_t B, int32_t T, const int32_t* idx, const int32_t* lens, const float* table, int32_t rows,
                                        int32_t dim, uint32_t* seen, float* out, int32_t* mult, void* stream) {
  DMT_CHECK_ARG(B > 0 && T > 0 && idx && lens && table && seen && out && mult && rows > 0 && dim > 0, "dmt_l2_unique_rows_count: bad argument");
  const long long n = (long long)B * T;
  hipLaunchKernelGGL(l2_unique_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, T, idx, lens, table, rows, dim,
                     seen, out, mult);
  DMT_CHECK_LAUNCH("dmt_l2_unique_rows_count");
  return DMT_OK;
}
