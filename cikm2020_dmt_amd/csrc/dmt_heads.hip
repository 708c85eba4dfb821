// The heads of the model in two launches (bf16 mode, reference widths): the T task towers (mix[t] [128] -> fc 32, relu -> 1 logit:
// build_tower, mmoe_transformer_unbias.py:107-126) and the position-bias tower (20 pooled inputs -> 32 -> 16 -> 1 with dropout 0.5
// after each hidden layer: embedding_mlp_bias, :259-289).  Per row this is 10 K FLOPs; as separate dense layers it was ~30 launches of
// 10-40 us (forward, relu / dropout gates, input gradients, N = 1 weight gradients) -- more than the MMoE experts cost.
//   forward   logits [T+1][B] fp32 (click, order, y_bias) + the hidden activations the backward pass needs.
//   backward  d logits -> d mix [T][B][128], d (bias input) [B][20], the pre-activation gradients dz of every hidden layer (operands
//             of the remaining weight-gradient GEMMs dW = x^T dz), and -- summed in LDS, one fp32 atomicAdd per element and workgroup --
//             the weight / bias gradients of the three 1-wide output layers.
// No MFMA: four lanes share a row, each owning a quarter of the hidden units; weights sit in LDS as fp32 (converted from the bf16
// shadows, so the products are the ones the MFMA path forms); activations are rounded to bf16 where the layer-per-launch path stores
// them.  Dropout is the library's counter mask with the flat index row * width + unit of the [B, width] activation (dmt_dropout).
#include "dmt_common.h"

namespace {

constexpr int UI = 128, UF = 32;          // tower input / hidden width
constexpr int BI = 20, BH0 = 32, BH1 = 16;    // bias tower widths
constexpr int MAXT = 2;

struct HeadsArgs {
  int B, T;
  const bf16_t* mix;                        // [T][B][UI]
  const bf16_t* zb; long long ld_zb;        // [B][>= BI]
  const bf16_t* fc_w[MAXT]; const float* fc_b[MAXT];      // [UI][UF] bf16 (plain shadow), [UF]
  const bf16_t* out_w[MAXT]; const float* out_b[MAXT];    // [UF][1], [1]
  const bf16_t* bw[3]; const float* bb[3];                // [BI][BH0], [BH0][BH1], [BH1][1]
  unsigned drop_seed[2], drop_thr[2];
  float drop_inv[2];
  int drop_on[2];
  float* logits;                            // [T+1][B]
  bf16_t* h_fc;                             // [T][B][UF]
  bf16_t* h0; bf16_t* h1;                   // [B][BH0], [B][BH1] (after dropout: the next layer's input)
  // backward
  const float* dlogits;                     // [T+1][B]
  bf16_t* dmix;                             // [T][B][UI]
  bf16_t* dzb; long long ld_dzb;            // [B][>= BI]
  bf16_t* dz_fc; bf16_t* dz0; bf16_t* dz1;  // [T][B][UF], [B][BH0], [B][BH1]
  float* g_out_w[MAXT]; float* g_out_b[MAXT];   // gradient views of the 1-wide output layers (accumulated into)
  float* g_bw2; float* g_bb2;
};

struct HeadsLds {
  float wfc[MAXT][UI * UF];
  float bfc[MAXT][UF];
  float wout[MAXT][UF];
  float bout[MAXT];
  float w0[BI * BH0], b0[BH0], w1[BH0 * BH1], b1[BH1], w2[BH1], b2;
  float gsum[MAXT][UF + 1];
  float gsum2[BH1 + 1];
};

__device__ __forceinline__ float bfround(float x) { return bf2f(f2bf(x)); }

__device__ __forceinline__ void load_weights(HeadsLds& s, const HeadsArgs& a, int tid) {
  for (int t = 0; t < a.T; ++t) {
    for (int i = tid; i < UI * UF; i += 256) s.wfc[t][i] = bf2f(a.fc_w[t][i]);
    if (tid < UF) { s.bfc[t][tid] = a.fc_b[t][tid]; s.wout[t][tid] = bf2f(a.out_w[t][tid]); }
    if (tid == 0) s.bout[t] = a.out_b[t][0];
  }
  for (int i = tid; i < BI * BH0; i += 256) s.w0[i] = bf2f(a.bw[0][i]);
  for (int i = tid; i < BH0 * BH1; i += 256) s.w1[i] = bf2f(a.bw[1][i]);
  if (tid < BH0) s.b0[tid] = a.bb[0][tid];
  if (tid < BH1) { s.b1[tid] = a.bb[1][tid]; s.w2[tid] = bf2f(a.bw[2][tid]); }
  if (tid == 0) s.b2 = a.bb[2][0];
}

__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xFFFF0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xFFFF0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xFFFF0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xFFFF0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(dmt_pack_bf16(f[0], f[1]), dmt_pack_bf16(f[2], f[3]), dmt_pack_bf16(f[4], f[5]), dmt_pack_bf16(f[6], f[7]));
}
__device__ __forceinline__ float quad_sum(float v) { v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); return v; }

// ------------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void heads_fwd_kernel(const HeadsArgs a) {
  __shared__ HeadsLds s;
  const int tid = threadIdx.x, g = tid & 3, lane = tid & 63;
  load_weights(s, a, tid);
  __syncthreads();
  const long long row = (long long)blockIdx.x * 64 + (tid >> 2);
  const bool ok = row < a.B;
  const long long r = ok ? row : 0;
  // ---- task towers
  for (int t = 0; t < a.T; ++t) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = s.bfc[t][8 * g + i];
    const bf16_t* mp = a.mix + ((long long)t * a.B + r) * UI;
#pragma unroll 4
    for (int k0 = 0; k0 < UI; k0 += 8) {
      float m[8];
      unpack8(*reinterpret_cast<const uint4*>(mp + k0), m);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const float4 wa = *reinterpret_cast<const float4*>(&s.wfc[t][(k0 + kk) * UF + 8 * g]);
        const float4 wb = *reinterpret_cast<const float4*>(&s.wfc[t][(k0 + kk) * UF + 8 * g + 4]);
        acc[0] = fmaf(m[kk], wa.x, acc[0]); acc[1] = fmaf(m[kk], wa.y, acc[1]); acc[2] = fmaf(m[kk], wa.z, acc[2]); acc[3] = fmaf(m[kk], wa.w, acc[3]);
        acc[4] = fmaf(m[kk], wb.x, acc[4]); acc[5] = fmaf(m[kk], wb.y, acc[5]); acc[6] = fmaf(m[kk], wb.z, acc[6]); acc[7] = fmaf(m[kk], wb.w, acc[7]);
      }
    }
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = bfround(fmaxf(acc[i], 0.f)); part = fmaf(acc[i], s.wout[t][8 * g + i], part); }
    if (ok) *reinterpret_cast<uint4*>(a.h_fc + ((long long)t * a.B + row) * UF + 8 * g) = pack8(acc);
    part = quad_sum(part);
    if (ok && g == 0) a.logits[(long long)t * a.B + row] = part + s.bout[t];
  }
  // ---- position-bias tower
  {
    float x[BI];
    const bf16_t* zp = a.zb + r * a.ld_zb;
#pragma unroll
    for (int k = 0; k < BI; k += 4) {
      const uint2 u = *reinterpret_cast<const uint2*>(zp + k);
      x[k] = __uint_as_float(u.x << 16); x[k + 1] = __uint_as_float(u.x & 0xFFFF0000u);
      x[k + 2] = __uint_as_float(u.y << 16); x[k + 3] = __uint_as_float(u.y & 0xFFFF0000u);
    }
    float h0[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) h0[i] = s.b0[8 * g + i];
#pragma unroll
    for (int k = 0; k < BI; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) h0[i] = fmaf(x[k], s.w0[k * BH0 + 8 * g + i], h0[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = bfround(fmaxf(h0[i], 0.f));
      if (a.drop_on[0]) v = dmt_drop_keep(a.drop_seed[0], (uint32_t)(row * BH0 + 8 * g + i), a.drop_thr[0]) ? bfround(v * a.drop_inv[0]) : 0.f;
      h0[i] = v;
    }
    if (ok) *reinterpret_cast<uint4*>(a.h0 + row * BH0 + 8 * g) = pack8(h0);
    float h1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h1[i] = s.b1[4 * g + i];
#pragma unroll
    for (int src = 0; src < 4; ++src)
#pragma unroll
      for (int ii = 0; ii < 8; ++ii) {
        const float v = __shfl(h0[ii], (lane & ~3) | src, 64);
        const float4 w = *reinterpret_cast<const float4*>(&s.w1[(8 * src + ii) * BH1 + 4 * g]);
        h1[0] = fmaf(v, w.x, h1[0]); h1[1] = fmaf(v, w.y, h1[1]); h1[2] = fmaf(v, w.z, h1[2]); h1[3] = fmaf(v, w.w, h1[3]);
      }
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = bfround(fmaxf(h1[i], 0.f));
      if (a.drop_on[1]) v = dmt_drop_keep(a.drop_seed[1], (uint32_t)(row * BH1 + 4 * g + i), a.drop_thr[1]) ? bfround(v * a.drop_inv[1]) : 0.f;
      h1[i] = v;
      part = fmaf(v, s.w2[4 * g + i], part);
    }
    if (ok) *reinterpret_cast<uint2*>(a.h1 + row * BH1 + 4 * g) = make_uint2(dmt_pack_bf16(h1[0], h1[1]), dmt_pack_bf16(h1[2], h1[3]));
    part = quad_sum(part);
    if (ok && g == 0) a.logits[(long long)a.T * a.B + row] = part + s.b2;
  }
}

// ----------------------------------------------------------------------------------------------------------- backward
__global__ __launch_bounds__(256) void heads_bwd_kernel(const HeadsArgs a) {
  __shared__ HeadsLds s;
  const int tid = threadIdx.x, g = tid & 3, lane = tid & 63;
  load_weights(s, a, tid);
  for (int i = tid; i < MAXT * (UF + 1); i += 256) (&s.gsum[0][0])[i] = 0.f;
  if (tid < BH1 + 1) s.gsum2[tid] = 0.f;
  __syncthreads();
  const long long row = (long long)blockIdx.x * 64 + (tid >> 2);
  const bool ok = row < a.B;
  const long long r = ok ? row : 0;
  for (int t = 0; t < a.T; ++t) {
    const float dl = ok ? bfround(a.dlogits[(long long)t * a.B + r]) : 0.f;      // (the layer path hands the logit gradient on as bf16)
    float h[8], dz[8];
    unpack8(*reinterpret_cast<const uint4*>(a.h_fc + ((long long)t * a.B + r) * UF + 8 * g), h);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dz[i] = (h[i] > 0.f) ? bfround(dl * s.wout[t][8 * g + i]) : 0.f;
      atomicAdd(&s.gsum[t][8 * g + i], h[i] * dl);                 // d W_out[j] = sum_rows h[j] dl
    }
    if (g == 0) atomicAdd(&s.gsum[t][UF], dl);
    if (ok) *reinterpret_cast<uint4*>(a.dz_fc + ((long long)t * a.B + row) * UF + 8 * g) = pack8(dz);
    // d mix[k] = sum_j dz[j] Wfc[k][j]: a quarter of the j per lane, summed over the four lanes of the row; lane g keeps k in [32g, 32g+32)
    float out[32];
#pragma unroll
    for (int k = 0; k < UI; ++k) {
      const float4 wa = *reinterpret_cast<const float4*>(&s.wfc[t][k * UF + 8 * g]);
      const float4 wb = *reinterpret_cast<const float4*>(&s.wfc[t][k * UF + 8 * g + 4]);
      float p = dz[0] * wa.x;
      p = fmaf(dz[1], wa.y, p); p = fmaf(dz[2], wa.z, p); p = fmaf(dz[3], wa.w, p);
      p = fmaf(dz[4], wb.x, p); p = fmaf(dz[5], wb.y, p); p = fmaf(dz[6], wb.z, p); p = fmaf(dz[7], wb.w, p);
      p = quad_sum(p);
      if ((k >> 5) == g) out[k & 31] = p;
    }
    if (ok) {
      bf16_t* dp = a.dmix + ((long long)t * a.B + row) * UI + 32 * g;
#pragma unroll
      for (int c = 0; c < 32; c += 8) {
        const float f[8] = {out[c], out[c + 1], out[c + 2], out[c + 3], out[c + 4], out[c + 5], out[c + 6], out[c + 7]};
        *reinterpret_cast<uint4*>(dp + c) = pack8(f);
      }
    }
  }
  // ---- position-bias tower
  {
    const float dy = ok ? bfround(a.dlogits[(long long)a.T * a.B + r]) : 0.f;
    float h1[4], dz1[4];
    {
      const uint2 u = *reinterpret_cast<const uint2*>(a.h1 + r * BH1 + 4 * g);
      h1[0] = __uint_as_float(u.x << 16); h1[1] = __uint_as_float(u.x & 0xFFFF0000u);
      h1[2] = __uint_as_float(u.y << 16); h1[3] = __uint_as_float(u.y & 0xFFFF0000u);
    }
    const float inv1 = a.drop_on[1] ? a.drop_inv[1] : 1.f, inv0 = a.drop_on[0] ? a.drop_inv[0] : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // h1 here is the dropped, rescaled activation: > 0 iff the unit fired AND was kept
      const float gd = bfround(dy * s.w2[4 * g + i]);              // gradient w.r.t. the dropped activation (a bf16 tensor in the layer path)
      dz1[i] = (h1[i] > 0.f) ? bfround(gd * inv1) : 0.f;
      atomicAdd(&s.gsum2[4 * g + i], h1[i] * dy);
    }
    if (g == 0) atomicAdd(&s.gsum2[BH1], dy);
    if (ok) *reinterpret_cast<uint2*>(a.dz1 + row * BH1 + 4 * g) = make_uint2(dmt_pack_bf16(dz1[0], dz1[1]), dmt_pack_bf16(dz1[2], dz1[3]));
    // d h0d[k] = sum_j dz1[j] W1[k][j], k = 8g .. 8g+7 on this lane; all 16 dz1 come from the row's four lanes
    float dh0[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dh0[i] = 0.f;
#pragma unroll
    for (int src = 0; src < 4; ++src)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float v = __shfl(dz1[jj], (lane & ~3) | src, 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) dh0[i] = fmaf(v, s.w1[(8 * g + i) * BH1 + 4 * src + jj], dh0[i]);
      }
    float h0[8], dz0[8];
    unpack8(*reinterpret_cast<const uint4*>(a.h0 + r * BH0 + 8 * g), h0);
#pragma unroll
    for (int i = 0; i < 8; ++i) dz0[i] = (h0[i] > 0.f) ? bfround(bfround(dh0[i]) * inv0) : 0.f;
    if (ok) *reinterpret_cast<uint4*>(a.dz0 + row * BH0 + 8 * g) = pack8(dz0);
    // d zb[k] = sum_j dz0[j] W0[k][j], k = 5g .. 5g+4 on this lane
    float dx[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int src = 0; src < 4; ++src)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const float v = __shfl(dz0[jj], (lane & ~3) | src, 64);
#pragma unroll
        for (int i = 0; i < 5; ++i) dx[i] = fmaf(v, s.w0[(5 * g + i) * BH0 + 8 * src + jj], dx[i]);
      }
    if (ok) {
      bf16_t* dp = a.dzb + row * a.ld_dzb + 5 * g;
#pragma unroll
      for (int i = 0; i < 5; ++i) dp[i] = f2bf(dx[i]);
    }
  }
  __syncthreads();
  // ---- the 1-wide layers' weight / bias gradients: one atomicAdd per element and workgroup onto the gradient arena
  for (int t = 0; t < a.T; ++t) {
    if (tid < UF) atomicAdd(a.g_out_w[t] + tid, s.gsum[t][tid]);
    if (tid == UF) atomicAdd(a.g_out_b[t], s.gsum[t][UF]);
  }
  if (tid < BH1) atomicAdd(a.g_bw2 + tid, s.gsum2[tid]);
  if (tid == BH1) atomicAdd(a.g_bb2, s.gsum2[BH1]);
}

int fill(HeadsArgs& a, const dmt_heads_desc* d, const char* who) {
  DMT_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  DMT_CHECK_ARG(d->B > 0 && d->T >= 1 && d->T <= MAXT, "%s: bad dims (T <= %d)", who, MAXT);
  DMT_CHECK_ARG(d->u_in == UI && d->u_fc == UF && d->b_in == BI && d->b_h0 == BH0 && d->b_h1 == BH1,
                "%s: built for tower widths %d/%d and bias-tower widths %d/%d/%d", who, UI, UF, BI, BH0, BH1);
  DMT_CHECK_ARG(d->mix && d->zb && d->logits && d->h_fc && d->h0 && d->h1, "%s: null buffer", who);
  DMT_CHECK_ARG((((uintptr_t)d->mix) & 15) == 0 && (((uintptr_t)d->zb) & 7) == 0 && d->ld_zb % 4 == 0, "%s: mix rows 16-byte, bias-input rows 8-byte aligned", who);
  a.B = d->B; a.T = d->T;
  a.mix = (const bf16_t*)d->mix; a.zb = (const bf16_t*)d->zb; a.ld_zb = d->ld_zb;
  for (int t = 0; t < d->T; ++t) {
    DMT_CHECK_ARG(d->fc_w[t] && d->fc_b[t] && d->out_w[t] && d->out_b[t], "%s: null tower weights", who);
    a.fc_w[t] = (const bf16_t*)d->fc_w[t]; a.fc_b[t] = d->fc_b[t]; a.out_w[t] = (const bf16_t*)d->out_w[t]; a.out_b[t] = d->out_b[t];
    a.g_out_w[t] = d->g_out_w[t]; a.g_out_b[t] = d->g_out_b[t];
  }
  for (int l = 0; l < 3; ++l) {
    DMT_CHECK_ARG(d->bias_w[l] && d->bias_b[l], "%s: null bias-tower weights", who);
    a.bw[l] = (const bf16_t*)d->bias_w[l]; a.bb[l] = d->bias_b[l];
  }
  for (int l = 0; l < 2; ++l) {
    a.drop_on[l] = (d->drop_keep[l] > 0.f && d->drop_keep[l] < 1.f) ? 1 : 0;
    a.drop_seed[l] = d->drop_seed[l];
    a.drop_thr[l] = a.drop_on[l] ? (unsigned)(d->drop_keep[l] * 16777216.0f) : 0u;
    a.drop_inv[l] = a.drop_on[l] ? 1.f / d->drop_keep[l] : 1.f;
  }
  a.logits = d->logits; a.h_fc = (bf16_t*)d->h_fc; a.h0 = (bf16_t*)d->h0; a.h1 = (bf16_t*)d->h1;
  a.dlogits = d->dlogits; a.dmix = (bf16_t*)d->dmix; a.dzb = (bf16_t*)d->dzb; a.ld_dzb = d->ld_dzb;
  a.dz_fc = (bf16_t*)d->dz_fc; a.dz0 = (bf16_t*)d->dz0; a.dz1 = (bf16_t*)d->dz1;
  a.g_bw2 = d->g_bias_w2; a.g_bb2 = d->g_bias_b2;
  return DMT_OK;
}

}  // namespace

extern "C" int dmt_heads_supported(int32_t u_in, int32_t u_fc, int32_t b_in, int32_t b_h0, int32_t b_h1, int32_t T) {
  return (u_in == UI && u_fc == UF && b_in == BI && b_h0 == BH0 && b_h1 == BH1 && T >= 1 && T <= MAXT) ? 1 : 0;
}

extern "C" int dmt_heads_fwd(const dmt_heads_desc* d, void* stream) {
  HeadsArgs a;
  if (fill(a, d, "dmt_heads_fwd") != DMT_OK) return DMT_ERR_ARG;
  hipLaunchKernelGGL(heads_fwd_kernel, dim3((unsigned)cdiv64(d->B, 64)), dim3(256), 0, (hipStream_t)stream, a);
  DMT_CHECK_LAUNCH("dmt_heads_fwd");
  return DMT_OK;
}

extern "C" int dmt_heads_bwd(const dmt_heads_desc* d, void* stream) {
  HeadsArgs a;
  if (fill(a, d, "dmt_heads_bwd") != DMT_OK) return DMT_ERR_ARG;
  DMT_CHECK_ARG(d->dlogits && d->dmix && d->dzb && d->dz_fc && d->dz0 && d->dz1 && d->g_bias_w2 && d->g_bias_b2, "dmt_heads_bwd: null argument");
  for (int t = 0; t < d->T; ++t) DMT_CHECK_ARG(d->g_out_w[t] && d->g_out_b[t], "dmt_heads_bwd: null gradient view");
  DMT_CHECK_ARG((((uintptr_t)d->dmix) & 15) == 0, "dmt_heads_bwd: dmix rows must be 16-byte aligned");
  hipLaunchKernelGGL(heads_bwd_kernel, dim3((unsigned)cdiv64(d->B, 64)), dim3(256), 0, (hipStream_t)stream, a);
  DMT_CHECK_LAUNCH("dmt_heads_bwd");
  return DMT_OK;
}
