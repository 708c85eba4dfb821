// Fused expert-MLP + gate kernels of the MMoE bottom (mmoe_transformer_unbias.py:63-105, expert_gate), bf16, reference widths
// 512 -> 256 -> 128 per expert.  Input of both kernels is the output of the ONE concatenated layer-0 GEMM:
//     g1 [B, E*512 + T*E] = relu(z W0_e + b0_e) for the E experts side by side | the T gates' logits.
//   forward   expert layers 1 and 2 (bias + relu), the T gate softmaxes over the experts, and the mixtures
//             mix[t] = sum_e gate[t][:, e] * h2_e                      -- one launch instead of two batched GEMMs + a mix kernel.
//   backward  d mix -> d gates -> d gate logits; d h2 (relu gate) -> d h1 = d h2 W2^T (relu gate) -> d x = d h1 W1^T, written as ONE
//             [B, E*512 + T*E] gradient of g1                           -- one launch instead of mix-bwd + 2 relu-bwd + 2 batched GEMMs.
//             (The weight gradients dW = x^T dh are reductions over the batch: they stay batched dmt_gemm launches over the saved
//             dh1 / dh2.)
// One workgroup (4 wavefronts) per 32 rows, the experts one after the other.  All products are computed TRANSPOSED
// (D^T = W^T X^T: A = weight rows from global memory -- every workgroup streams the same 1.3 MB of weights, an L2 stream --, B =
// activation rows), so a lane owns one example row and 4 consecutive output columns per register group: biases / relu gates are
// register-indexed, the gate of a row is a per-lane scalar, the hidden tile goes to LDS with 8-byte stores and comes back as the
// B operand of the next layer with 16-byte reads.  FLOPs are tiny (1.3 MFLOP per row); what the fusion buys is launches and the
// round trips of h1 / h2 / their gradients between them.
#include "dmt_common.h"

extern "C" int64_t dmt_mmoe_experts_ws_bytes(int32_t B);

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int U0 = 512, U1 = 256, U2 = 128;
constexpr int MT = 1;                  // 32-row tiles per workgroup
constexpr int MR = 32 * MT;            // rows per workgroup (B = 4096: 128 workgroups)
constexpr int H1S = U1 + 8;            // LDS row strides (elements): 16-byte fragment reads, rows 4 banks apart
constexpr int H2S = U2 + 8;
constexpr int MAXG = 16;               // T * E

struct MmoeArgs {
  int B, E, T;
  const bf16_t* g1; long long ldg;                 // [B, >= E*U0 + T*E]
  const bf16_t* w1t; long long w1_es, w1_ld;       // layer-1 weights, transposed shadows [E][U1][ld >= U0]  (k-contiguous rows)
  const bf16_t* w2t; long long w2_es, w2_ld;       // layer-2 weights, transposed shadows [E][U2][ld >= U1]
  const bf16_t* w1p; long long w1p_es;             // plain shadows [E][U0][U1] (backward)
  const bf16_t* w2p; long long w2p_es;             // plain shadows [E][U1][U2]
  const float* b1; long long b1_es;                // biases [E][U1], [E][U2]
  const float* b2; long long b2_es;
  bf16_t* h1; bf16_t* h2;                          // saved activations [B, E*U1], [B, E*U2]
  float* gates;                                    // [T, B, E]
  bf16_t* mix;                                     // [T, B, U2]
  // backward
  const bf16_t* dmix;                              // [T, B, U2]
  bf16_t* dh1; bf16_t* dh2;                        // [B, E*U1], [B, E*U2]: operands of the weight-gradient GEMMs
  bf16_t* dg1; long long lddg;                     // [B, >= E*U0 + T*E]
  // split form (one workgroup per (row tile, expert)): the experts' d gate partials
  int n_tiles;
  float* dgs;                                      // [B][MAXG]
  int gate_dx;                                     // backward: d x *= (g1 > 0)
};

__device__ __forceinline__ f32x16_t zero16() {
  f32x16_t z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
__device__ __forceinline__ bf16x8_t zero8() {
  union { bf16x8_t v; uint4 q; } u;
  u.q = make_uint4(0u, 0u, 0u, 0u);
  return u.v;
}
__device__ __forceinline__ bf16x8_t ld8(const bf16_t* p) { return *reinterpret_cast<const bf16x8_t*>(p); }
__device__ __forceinline__ f32x16_t mma(const bf16x8_t& a, const bf16x8_t& b, const f32x16_t& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float bfround(float x) { return bf2f(f2bf(x)); }

// accumulator group g of a D^T tile (lane = example row, registers 4g..4g+3 = output columns n0 + 8g + 4 half + {0..3}) -> 4 bf16
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) { return make_uint2(dmt_pack_bf16(a, b), dmt_pack_bf16(c, d)); }

// ------------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void mmoe_experts_fwd_kernel(const MmoeArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t s_h1[MR * H1S];
  __shared__ float s_gate[MR][MAXG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const long long r0 = (long long)blockIdx.x * MR;
  const int E = a.E, T = a.T;
  // ---- gates: softmax over the experts' logits, per (row, task)
  if (tid < MR * T) {
    const int m = tid / T, t = tid - m * T;
    const long long row = r0 + m;
    if (row < a.B) {
      const bf16_t* gl = a.g1 + row * a.ldg + E * U0 + t * E;
      float mx = -3.0e38f;
      for (int e = 0; e < E; ++e) mx = fmaxf(mx, bf2f(gl[e]));
      float s = 0.f;
      for (int e = 0; e < E; ++e) { const float v = expf(bf2f(gl[e]) - mx); s_gate[m][t * E + e] = v; s += v; }
      for (int e = 0; e < E; ++e) { const float v = s_gate[m][t * E + e] / s; s_gate[m][t * E + e] = v; a.gates[((long long)t * a.B + row) * E + e] = v; }
    } else {
      for (int e = 0; e < E; ++e) s_gate[m][t * E + e] = 0.f;
    }
  }
  f32x16_t mixacc[4][MT];       // [task][row tile]: lane = row, registers = this wave's 32 output columns
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) mixacc[t][mt] = zero16();
  long long rowm[MT];
  bool rok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { rowm[mt] = r0 + mt * 32 + l31; rok[mt] = rowm[mt] < a.B; }

  for (int e = 0; e < E; ++e) {
    // ---- layer 1: h1^T [U1][64] = W1^T [U1][U0] x^T; this wave: hidden units 64 wave .. +63
    {
      f32x16_t acc[2][MT];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[i][mt] = zero16();
      const bf16_t* wp[2];
      const bf16_t* xp[MT];
#pragma unroll
      for (int i = 0; i < 2; ++i) wp[i] = a.w1t + e * a.w1_es + (long long)(64 * wave + 32 * i + l31) * a.w1_ld + 8 * half;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xp[mt] = a.g1 + (rok[mt] ? rowm[mt] : 0) * a.ldg + e * U0 + 8 * half;
#pragma unroll 8
      for (int k0 = 0; k0 < U0; k0 += 16) {
        const bf16x8_t w0 = ld8(wp[0] + k0), w1 = ld8(wp[1] + k0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const bf16x8_t x = ld8(xp[mt] + k0);
          acc[0][mt] = mma(w0, x, acc[0][mt]);
          acc[1][mt] = mma(w1, x, acc[1][mt]);
        }
      }
      const float* bias = a.b1 + e * a.b1_es;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = 64 * wave + 32 * nt + 8 * g + 4 * half;
          const float4 bb = *reinterpret_cast<const float4*>(bias + n);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const float v0 = fmaxf(acc[nt][mt][4 * g + 0] + bb.x, 0.f), v1 = fmaxf(acc[nt][mt][4 * g + 1] + bb.y, 0.f);
            const float v2 = fmaxf(acc[nt][mt][4 * g + 2] + bb.z, 0.f), v3 = fmaxf(acc[nt][mt][4 * g + 3] + bb.w, 0.f);
            const uint2 pk = pack4(v0, v1, v2, v3);
            *reinterpret_cast<uint2*>(s_h1 + (mt * 32 + l31) * H1S + n) = pk;
            if (rok[mt]) *reinterpret_cast<uint2*>(a.h1 + rowm[mt] * (long long)(E * U1) + e * U1 + n) = pk;
          }
        }
    }
    __syncthreads();
    // ---- layer 2: h2^T [U2][64] = W2^T [U2][U1] h1^T; this wave: output units 32 wave .. +31
    {
      f32x16_t acc[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = zero16();
      const bf16_t* wp = a.w2t + e * a.w2_es + (long long)(32 * wave + l31) * a.w2_ld + 8 * half;
      const bf16_t* hp = s_h1 + l31 * H1S + 8 * half;
#pragma unroll 8
      for (int k0 = 0; k0 < U1; k0 += 16) {
        const bf16x8_t w = ld8(wp + k0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = mma(w, ld8(hp + mt * 32 * H1S + k0), acc[mt]);
      }
      const float* bias = a.b2 + e * a.b2_es;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 32 * wave + 8 * g + 4 * half;
        const float4 bb = *reinterpret_cast<const float4*>(bias + n);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          // (rounded to bf16 first: the mixture then sees exactly the h2 the backward pass reads)
          const float v0 = bfround(fmaxf(acc[mt][4 * g + 0] + bb.x, 0.f)), v1 = bfround(fmaxf(acc[mt][4 * g + 1] + bb.y, 0.f));
          const float v2 = bfround(fmaxf(acc[mt][4 * g + 2] + bb.z, 0.f)), v3 = bfround(fmaxf(acc[mt][4 * g + 3] + bb.w, 0.f));
          if (rok[mt]) *reinterpret_cast<uint2*>(a.h2 + rowm[mt] * (long long)(E * U2) + e * U2 + n) = pack4(v0, v1, v2, v3);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (t >= T) break;
            const float gt = s_gate[mt * 32 + l31][t * E + e];
            mixacc[t][mt][4 * g + 0] = fmaf(gt, v0, mixacc[t][mt][4 * g + 0]);
            mixacc[t][mt][4 * g + 1] = fmaf(gt, v1, mixacc[t][mt][4 * g + 1]);
            mixacc[t][mt][4 * g + 2] = fmaf(gt, v2, mixacc[t][mt][4 * g + 2]);
            mixacc[t][mt][4 * g + 3] = fmaf(gt, v3, mixacc[t][mt][4 * g + 3]);
          }
        }
      }
    }
    __syncthreads();       // s_h1 is rewritten by the next expert
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t >= T) break;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (!rok[mt]) continue;
      bf16_t* mp = a.mix + ((long long)t * a.B + rowm[mt]) * U2 + 32 * wave + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<uint2*>(mp + 8 * g) = pack4(mixacc[t][mt][4 * g], mixacc[t][mt][4 * g + 1], mixacc[t][mt][4 * g + 2], mixacc[t][mt][4 * g + 3]);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------- backward
__global__ __launch_bounds__(256) void mmoe_experts_bwd_kernel(const MmoeArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t s_dh2[MR * H2S];
  __shared__ __attribute__((aligned(16))) bf16_t s_dh1[MR * H1S];
  __shared__ float s_gate[MR][MAXG];
  __shared__ float s_dg[MR][MAXG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const long long r0 = (long long)blockIdx.x * MR;
  const int E = a.E, T = a.T, ng = T * E;
  for (int i = tid; i < MR * MAXG; i += 256) {
    const int m = i / MAXG, j = i - m * MAXG;
    const long long row = r0 + m;
    s_gate[m][j] = (j < ng && row < a.B) ? a.gates[((long long)(j / E) * a.B + row) * E + (j % E)] : 0.f;
    s_dg[m][j] = 0.f;
  }
  __syncthreads();
  long long rowm[MT];
  bool rok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { rowm[mt] = r0 + mt * 32 + l31; rok[mt] = rowm[mt] < a.B; }
  // element-wise stage mapping: EPR threads per row, ECW consecutive columns each
  constexpr int EPR = 256 / MR, ECW = U2 / EPR;
  const int em = tid / EPR, ec = (tid % EPR) * ECW;
  const long long erow = r0 + em;
  const bool eok = erow < a.B;

  for (int e = 0; e < E; ++e) {
    // ---- d h2 = relu'(h2) * sum_t gate[t][e] d mix[t];  d gate[t][e] = d mix[t] . h2     (element-wise, 32 columns per thread)
    {
      float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c8 = 0; c8 < ECW; c8 += 8) {
        float hv[8], dv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) dv[i] = 0.f;
        union { uint4 q; bf16_t h[8]; } hx, dm;
        hx.q = eok ? *reinterpret_cast<const uint4*>(a.h2 + erow * (long long)(E * U2) + e * U2 + ec + c8) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int i = 0; i < 8; ++i) hv[i] = bf2f(hx.h[i]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (t >= T) break;
          dm.q = eok ? *reinterpret_cast<const uint4*>(a.dmix + ((long long)t * a.B + erow) * U2 + ec + c8) : make_uint4(0u, 0u, 0u, 0u);
          const float gt = s_gate[em][t * E + e];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float d = bf2f(dm.h[i]);
            dv[i] = fmaf(gt, d, dv[i]);
            part[t] = fmaf(hv[i], d, part[t]);
          }
        }
        uint4 o;
        o.x = dmt_pack_bf16(hv[0] > 0.f ? dv[0] : 0.f, hv[1] > 0.f ? dv[1] : 0.f);
        o.y = dmt_pack_bf16(hv[2] > 0.f ? dv[2] : 0.f, hv[3] > 0.f ? dv[3] : 0.f);
        o.z = dmt_pack_bf16(hv[4] > 0.f ? dv[4] : 0.f, hv[5] > 0.f ? dv[5] : 0.f);
        o.w = dmt_pack_bf16(hv[6] > 0.f ? dv[6] : 0.f, hv[7] > 0.f ? dv[7] : 0.f);
        *reinterpret_cast<uint4*>(s_dh2 + em * H2S + ec + c8) = o;
        if (eok) *reinterpret_cast<uint4*>(a.dh2 + erow * (long long)(E * U2) + e * U2 + ec + c8) = o;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t >= T) break;
        float p = part[t];
#pragma unroll
        for (int o = 1; o < EPR; o <<= 1) p += __shfl_xor(p, o, 64);
        if ((tid % EPR) == 0) s_dg[em][t * E + e] = p;
      }
    }
    __syncthreads();
    // ---- d h1^T [U1][64] = W2 [U1][U2] d h2^T, relu gate of h1; this wave: hidden units 64 wave .. +63
    {
      f32x16_t acc[2][MT];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[i][mt] = zero16();
      const bf16_t* wp[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) wp[i] = a.w2p + e * a.w2p_es + (long long)(64 * wave + 32 * i + l31) * U2 + 8 * half;
      const bf16_t* dp = s_dh2 + l31 * H2S + 8 * half;
#pragma unroll
      for (int k0 = 0; k0 < U2; k0 += 16) {
        const bf16x8_t w0 = ld8(wp[0] + k0), w1 = ld8(wp[1] + k0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const bf16x8_t dd = ld8(dp + mt * 32 * H2S + k0);
          acc[0][mt] = mma(w0, dd, acc[0][mt]);
          acc[1][mt] = mma(w1, dd, acc[1][mt]);
        }
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = 64 * wave + 32 * nt + 8 * g + 4 * half;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            union { uint2 u; bf16_t h[4]; } hv;
            hv.u = rok[mt] ? *reinterpret_cast<const uint2*>(a.h1 + rowm[mt] * (long long)(E * U1) + e * U1 + n) : make_uint2(0u, 0u);
            const uint2 pk = pack4(bf2f(hv.h[0]) > 0.f ? acc[nt][mt][4 * g + 0] : 0.f, bf2f(hv.h[1]) > 0.f ? acc[nt][mt][4 * g + 1] : 0.f,
                                   bf2f(hv.h[2]) > 0.f ? acc[nt][mt][4 * g + 2] : 0.f, bf2f(hv.h[3]) > 0.f ? acc[nt][mt][4 * g + 3] : 0.f);
            *reinterpret_cast<uint2*>(s_dh1 + (mt * 32 + l31) * H1S + n) = pk;
            if (rok[mt]) *reinterpret_cast<uint2*>(a.dh1 + rowm[mt] * (long long)(E * U1) + e * U1 + n) = pk;
          }
        }
    }
    __syncthreads();
    // ---- d x^T [U0][64] = W1 [U0][U1] d h1^T; this wave: inputs 128 wave .. +127  (the layer-0 relu gate belongs to the GEMM that made g1)
    {
      f32x16_t acc[4][MT];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[i][mt] = zero16();
      const bf16_t* wp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wp[i] = a.w1p + e * a.w1p_es + (long long)(128 * wave + 32 * i + l31) * U1 + 8 * half;
      const bf16_t* dp = s_dh1 + l31 * H1S + 8 * half;
#pragma unroll 4
      for (int k0 = 0; k0 < U1; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bf16x8_t w = ld8(wp[i] + k0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mma(w, ld8(dp + mt * 32 * H1S + k0), acc[i][mt]);
        }
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (!rok[mt]) continue;
          bf16_t* xp = a.dg1 + rowm[mt] * a.lddg + e * U0 + 128 * wave + 32 * nt + 4 * half;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<uint2*>(xp + 8 * g) = pack4(acc[nt][mt][4 * g], acc[nt][mt][4 * g + 1], acc[nt][mt][4 * g + 2], acc[nt][mt][4 * g + 3]);
        }
    }
    __syncthreads();       // s_dh2 / s_dh1 are rewritten by the next expert
  }
  // ---- d gate logits: softmax backward per (row, task)
  if (tid < MR * T) {
    const int m = tid / T, t = tid - m * T;
    const long long row = r0 + m;
    if (row < a.B) {
      float dot = 0.f;
      for (int e = 0; e < E; ++e) dot += s_gate[m][t * E + e] * s_dg[m][t * E + e];
      bf16_t* dl = a.dg1 + row * a.lddg + E * U0 + t * E;
      for (int e = 0; e < E; ++e) dl[e] = f2bf(s_gate[m][t * E + e] * (s_dg[m][t * E + e] - dot));
    }
  }
}


// ---------------------------------------------------------------------------------------------- split form (round 4)
// B = 4096 rows are 128 row tiles: the kernels above fill half the chip, and a workgroup walks its four experts one after the other
// -- a chain of L2 round trips (every MFMA operand of the weights comes straight from global memory) four experts long: 85 / 75 us
// in the middle of the step's junction, where nothing else runs.  Here one workgroup owns (row tile, expert): 512 workgroups, a chain one
// expert long.  What needs all experts of a tile -- the mixtures, the softmax gradient of the gate logits -- is done by a small second
// launch (mmoe_mix_finish_kernel / mmoe_dgate_finish_kernel: one workgroup per row tile) that walks the experts in index order over the
// values the first one stored (h2 rounded to bf16, the fp32 gates, the fp32 d gate partials): the same operations in the same order as
// above, bit-identical results.  (First built with the tile's last-arriving workgroup doing that step behind a device-scope counter:
// correct, and SLOWER than the one-workgroup form -- 128 against 80 us per launch -- because a device-scope release on this chip is a
// write-back of the XCD's whole L2, once per workgroup.  The kernel boundary is the cheap release.)  Workgroup i runs on XCD i % 8:
// the E workgroups of a tile are E consecutive workgroups of one XCD (they share the tile's g1 / d mix rows in its L2).
__device__ __forceinline__ bool split_map(const MmoeArgs& a, int& tile, int& e) {
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  tile = (int)((slot / (unsigned)a.E) * 8u + xcd);
  e = (int)(slot % (unsigned)a.E);
  return tile < a.n_tiles;
}

__global__ __launch_bounds__(256) void mmoe_split_fwd_kernel(const MmoeArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t s_h1[MR * H1S];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  int tile, e;
  if (!split_map(a, tile, e)) return;
  const long long r0 = (long long)tile * MR;
  const int E = a.E, T = a.T;
  // ---- gates (expert 0's workgroup): softmax over the experts' logits, per (row, task)
  if (e == 0 && tid < MR * T) {
    const int m = tid / T, t = tid - m * T;
    const long long row = r0 + m;
    if (row < a.B) {
      const bf16_t* gl = a.g1 + row * a.ldg + E * U0 + t * E;
      float ex[MAXG];
      float mx = -3.0e38f;
      for (int q = 0; q < E; ++q) mx = fmaxf(mx, bf2f(gl[q]));
      float s = 0.f;
      for (int q = 0; q < E; ++q) { const float v = expf(bf2f(gl[q]) - mx); ex[q] = v; s += v; }
      for (int q = 0; q < E; ++q) a.gates[((long long)t * a.B + row) * E + q] = ex[q] / s;
    }
  }
  const long long rowm = r0 + l31;
  const bool rok = rowm < a.B;
  // ---- layer 1: h1^T [U1][32] = W1^T [U1][U0] x^T; this wave: hidden units 64 wave .. +63
  {
    f32x16_t acc[2] = {zero16(), zero16()};
    const bf16_t* wp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) wp[i] = a.w1t + e * a.w1_es + (long long)(64 * wave + 32 * i + l31) * a.w1_ld + 8 * half;
    const bf16_t* xp = a.g1 + (rok ? rowm : 0) * a.ldg + e * U0 + 8 * half;
#pragma unroll 8
    for (int k0 = 0; k0 < U0; k0 += 16) {
      const bf16x8_t w0 = ld8(wp[0] + k0), w1 = ld8(wp[1] + k0);
      const bf16x8_t x = ld8(xp + k0);
      acc[0] = mma(w0, x, acc[0]);
      acc[1] = mma(w1, x, acc[1]);
    }
    const float* bias = a.b1 + e * a.b1_es;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 64 * wave + 32 * nt + 8 * g + 4 * half;
        const float4 bb = *reinterpret_cast<const float4*>(bias + n);
        const uint2 pk = pack4(fmaxf(acc[nt][4 * g + 0] + bb.x, 0.f), fmaxf(acc[nt][4 * g + 1] + bb.y, 0.f),
                               fmaxf(acc[nt][4 * g + 2] + bb.z, 0.f), fmaxf(acc[nt][4 * g + 3] + bb.w, 0.f));
        *reinterpret_cast<uint2*>(s_h1 + l31 * H1S + n) = pk;
        if (rok) *reinterpret_cast<uint2*>(a.h1 + rowm * (long long)(E * U1) + e * U1 + n) = pk;
      }
  }
  __syncthreads();
  // ---- layer 2: h2^T [U2][32] = W2^T [U2][U1] h1^T; this wave: output units 32 wave .. +31
  {
    f32x16_t acc = zero16();
    const bf16_t* wp = a.w2t + e * a.w2_es + (long long)(32 * wave + l31) * a.w2_ld + 8 * half;
    const bf16_t* hp = s_h1 + l31 * H1S + 8 * half;
#pragma unroll 8
    for (int k0 = 0; k0 < U1; k0 += 16) acc = mma(ld8(wp + k0), ld8(hp + k0), acc);
    const float* bias = a.b2 + e * a.b2_es;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = 32 * wave + 8 * g + 4 * half;
      const float4 bb = *reinterpret_cast<const float4*>(bias + n);
      if (rok)
        *reinterpret_cast<uint2*>(a.h2 + rowm * (long long)(E * U2) + e * U2 + n) =
            pack4(fmaxf(acc[4 * g + 0] + bb.x, 0.f), fmaxf(acc[4 * g + 1] + bb.y, 0.f), fmaxf(acc[4 * g + 2] + bb.z, 0.f), fmaxf(acc[4 * g + 3] + bb.w, 0.f));
    }
  }
}

// mix[t] = sum_e gate[t][e] * h2_e, e ascending (an fmaf chain from 0, as in the one-workgroup form); one workgroup per row tile
__global__ __launch_bounds__(256) void mmoe_mix_finish_kernel(const MmoeArgs a) {
  const int tid = threadIdx.x;
  const long long r0 = (long long)blockIdx.x * MR;
  const int E = a.E, T = a.T;
  const int m = tid >> 3, c0 = (tid & 7) * 16;
  const long long row = r0 + m;
  if (row >= a.B) return;
  for (int t = 0; t < T; ++t) {
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int q = 0; q < E; ++q) {
      const float gt = a.gates[((long long)t * a.B + row) * E + q];
      union { uint4 u[2]; bf16_t h[16]; } hv;
      const uint4* hp = reinterpret_cast<const uint4*>(a.h2 + row * (long long)(E * U2) + q * U2 + c0);
      hv.u[0] = hp[0]; hv.u[1] = hp[1];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fmaf(gt, bf2f(hv.h[i]), acc[i]);
    }
    uint4 o[2];
    o[0] = make_uint4(dmt_pack_bf16(acc[0], acc[1]), dmt_pack_bf16(acc[2], acc[3]), dmt_pack_bf16(acc[4], acc[5]), dmt_pack_bf16(acc[6], acc[7]));
    o[1] = make_uint4(dmt_pack_bf16(acc[8], acc[9]), dmt_pack_bf16(acc[10], acc[11]), dmt_pack_bf16(acc[12], acc[13]), dmt_pack_bf16(acc[14], acc[15]));
    uint4* mp = reinterpret_cast<uint4*>(a.mix + ((long long)t * a.B + row) * U2 + c0);
    mp[0] = o[0]; mp[1] = o[1];
  }
}

__global__ __launch_bounds__(256) void mmoe_split_bwd_kernel(const MmoeArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t s_dh2[MR * H2S];
  __shared__ __attribute__((aligned(16))) bf16_t s_dh1[MR * H1S];
  __shared__ float s_gate[MR][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  int tile, e;
  if (!split_map(a, tile, e)) return;
  const long long r0 = (long long)tile * MR;
  const int E = a.E, T = a.T;
  if (tid < MR * 4) {
    const int m = tid >> 2, t = tid & 3;
    const long long row = r0 + m;
    s_gate[m][t] = (t < T && row < a.B) ? a.gates[((long long)t * a.B + row) * E + e] : 0.f;
  }
  __syncthreads();
  const long long rowm = r0 + l31;
  const bool rok = rowm < a.B;
  constexpr int EPR = 256 / MR, ECW = U2 / EPR;
  const int em = tid / EPR, ec = (tid % EPR) * ECW;
  const long long erow = r0 + em;
  const bool eok = erow < a.B;
  // ---- d h2 = relu'(h2) * sum_t gate[t][e] d mix[t];  d gate[t][e] = d mix[t] . h2
  {
    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c8 = 0; c8 < ECW; c8 += 8) {
      float hv[8], dv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) dv[i] = 0.f;
      union { uint4 q; bf16_t h[8]; } hx, dm;
      hx.q = eok ? *reinterpret_cast<const uint4*>(a.h2 + erow * (long long)(E * U2) + e * U2 + ec + c8) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int i = 0; i < 8; ++i) hv[i] = bf2f(hx.h[i]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t >= T) break;
        dm.q = eok ? *reinterpret_cast<const uint4*>(a.dmix + ((long long)t * a.B + erow) * U2 + ec + c8) : make_uint4(0u, 0u, 0u, 0u);
        const float gt = s_gate[em][t];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float d = bf2f(dm.h[i]);
          dv[i] = fmaf(gt, d, dv[i]);
          part[t] = fmaf(hv[i], d, part[t]);
        }
      }
      uint4 o;
      o.x = dmt_pack_bf16(hv[0] > 0.f ? dv[0] : 0.f, hv[1] > 0.f ? dv[1] : 0.f);
      o.y = dmt_pack_bf16(hv[2] > 0.f ? dv[2] : 0.f, hv[3] > 0.f ? dv[3] : 0.f);
      o.z = dmt_pack_bf16(hv[4] > 0.f ? dv[4] : 0.f, hv[5] > 0.f ? dv[5] : 0.f);
      o.w = dmt_pack_bf16(hv[6] > 0.f ? dv[6] : 0.f, hv[7] > 0.f ? dv[7] : 0.f);
      *reinterpret_cast<uint4*>(s_dh2 + em * H2S + ec + c8) = o;
      if (eok) *reinterpret_cast<uint4*>(a.dh2 + erow * (long long)(E * U2) + e * U2 + ec + c8) = o;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t >= T) break;
      float p = part[t];
#pragma unroll
      for (int o = 1; o < EPR; o <<= 1) p += __shfl_xor(p, o, 64);
      if ((tid % EPR) == 0 && eok) a.dgs[erow * MAXG + t * E + e] = p;
    }
  }
  __syncthreads();
  // ---- d h1^T [U1][32] = W2 [U1][U2] d h2^T, relu gate of h1; this wave: hidden units 64 wave .. +63
  {
    f32x16_t acc[2] = {zero16(), zero16()};
    const bf16_t* wp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) wp[i] = a.w2p + e * a.w2p_es + (long long)(64 * wave + 32 * i + l31) * U2 + 8 * half;
    const bf16_t* dp = s_dh2 + l31 * H2S + 8 * half;
#pragma unroll
    for (int k0 = 0; k0 < U2; k0 += 16) {
      const bf16x8_t w0 = ld8(wp[0] + k0), w1 = ld8(wp[1] + k0);
      const bf16x8_t dd = ld8(dp + k0);
      acc[0] = mma(w0, dd, acc[0]);
      acc[1] = mma(w1, dd, acc[1]);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 64 * wave + 32 * nt + 8 * g + 4 * half;
        union { uint2 u; bf16_t h[4]; } hv;
        hv.u = rok ? *reinterpret_cast<const uint2*>(a.h1 + rowm * (long long)(E * U1) + e * U1 + n) : make_uint2(0u, 0u);
        const uint2 pk = pack4(bf2f(hv.h[0]) > 0.f ? acc[nt][4 * g + 0] : 0.f, bf2f(hv.h[1]) > 0.f ? acc[nt][4 * g + 1] : 0.f,
                               bf2f(hv.h[2]) > 0.f ? acc[nt][4 * g + 2] : 0.f, bf2f(hv.h[3]) > 0.f ? acc[nt][4 * g + 3] : 0.f);
        *reinterpret_cast<uint2*>(s_dh1 + l31 * H1S + n) = pk;
        if (rok) *reinterpret_cast<uint2*>(a.dh1 + rowm * (long long)(E * U1) + e * U1 + n) = pk;
      }
  }
  __syncthreads();
  // ---- d x^T [U0][32] = W1 [U0][U1] d h1^T; this wave: inputs 128 wave .. +127 (gate_dx: times the relu gradient of the layer that made g1)
  {
    f32x16_t acc[4] = {zero16(), zero16(), zero16(), zero16()};
    const bf16_t* wp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wp[i] = a.w1p + e * a.w1p_es + (long long)(128 * wave + 32 * i + l31) * U1 + 8 * half;
    const bf16_t* dp = s_dh1 + l31 * H1S + 8 * half;
#pragma unroll 4
    for (int k0 = 0; k0 < U1; k0 += 16) {
      const bf16x8_t dd = ld8(dp + k0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = mma(ld8(wp[i] + k0), dd, acc[i]);
    }
    if (rok) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const long long col = e * U0 + 128 * wave + 32 * nt + 4 * half;
        bf16_t* xp = a.dg1 + rowm * a.lddg + col;
        const bf16_t* gp = a.g1 + rowm * a.ldg + col;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v0 = acc[nt][4 * g], v1 = acc[nt][4 * g + 1], v2 = acc[nt][4 * g + 2], v3 = acc[nt][4 * g + 3];
          if (a.gate_dx) {
            union { uint2 u; bf16_t h[4]; } gv;
            gv.u = *reinterpret_cast<const uint2*>(gp + 8 * g);
            v0 = bf2f(gv.h[0]) > 0.f ? v0 : 0.f; v1 = bf2f(gv.h[1]) > 0.f ? v1 : 0.f;
            v2 = bf2f(gv.h[2]) > 0.f ? v2 : 0.f; v3 = bf2f(gv.h[3]) > 0.f ? v3 : 0.f;
          }
          *reinterpret_cast<uint2*>(xp + 8 * g) = pack4(v0, v1, v2, v3);
        }
      }
    }
  }
}

// d gate logits: softmax backward per (row, task) over the experts' partials, experts ascending; one workgroup per row tile
__global__ __launch_bounds__(256) void mmoe_dgate_finish_kernel(const MmoeArgs a) {
  const int tid = threadIdx.x;
  const long long r0 = (long long)blockIdx.x * MR;
  const int E = a.E, T = a.T;
  if (tid < MR * T) {
    const int m = tid / T, t = tid - m * T;
    const long long row = r0 + m;
    if (row < a.B) {
      float gt[MAXG], dg[MAXG];
      float dot = 0.f;
      for (int q = 0; q < E; ++q) {
        gt[q] = a.gates[((long long)t * a.B + row) * E + q];
        dg[q] = a.dgs[row * MAXG + t * E + q];
        dot += gt[q] * dg[q];
      }
      bf16_t* dl = a.dg1 + row * a.lddg + E * U0 + t * E;
      for (int q = 0; q < E; ++q) dl[q] = f2bf(gt[q] * (dg[q] - dot));
    }
  }
}

int fill(MmoeArgs& a, const dmt_mmoe_desc* d, const char* who) {
  DMT_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  DMT_CHECK_ARG(d->B > 0 && d->E > 0 && d->T > 0 && d->T <= 4 && d->T * d->E <= MAXG, "%s: bad dims (T <= 4, T*E <= 16)", who);
  DMT_CHECK_ARG(d->u0 == U0 && d->u1 == U1 && d->u2 == U2, "%s: built for expert widths %d/%d/%d (got %d/%d/%d)", who, U0, U1, U2, d->u0, d->u1, d->u2);
  DMT_CHECK_ARG(d->g1 && d->gates && d->h1 && d->h2, "%s: null buffer", who);
  DMT_CHECK_ARG(d->ldg % 8 == 0 && (((uintptr_t)d->g1) & 15) == 0, "%s: g1 rows must be 16-byte aligned", who);
  a.B = d->B; a.E = d->E; a.T = d->T;
  a.g1 = (const bf16_t*)d->g1; a.ldg = d->ldg;
  a.w1t = (const bf16_t*)d->w1t; a.w1_es = d->w1t_expert_stride; a.w1_ld = d->w1t_ld;
  a.w2t = (const bf16_t*)d->w2t; a.w2_es = d->w2t_expert_stride; a.w2_ld = d->w2t_ld;
  a.w1p = (const bf16_t*)d->w1; a.w1p_es = d->w1_expert_stride;
  a.w2p = (const bf16_t*)d->w2; a.w2p_es = d->w2_expert_stride;
  a.b1 = d->b1; a.b1_es = d->b1_expert_stride;
  a.b2 = d->b2; a.b2_es = d->b2_expert_stride;
  a.h1 = (bf16_t*)d->h1; a.h2 = (bf16_t*)d->h2; a.gates = d->gates; a.mix = (bf16_t*)d->mix;
  a.dmix = (const bf16_t*)d->dmix; a.dh1 = (bf16_t*)d->dh1; a.dh2 = (bf16_t*)d->dh2; a.dg1 = (bf16_t*)d->dg1; a.lddg = d->lddg;
  a.n_tiles = (int)cdiv64(d->B, MR);
  a.dgs = nullptr;
  a.gate_dx = d->gate_dx ? 1 : 0;
  if (d->ws != nullptr) {
    DMT_CHECK_ARG(d->ws_bytes >= dmt_mmoe_experts_ws_bytes(d->B) && (((uintptr_t)d->ws) & 15) == 0, "%s: workspace too small (%lld < %lld bytes) or misaligned", who,
                  (long long)d->ws_bytes, (long long)dmt_mmoe_experts_ws_bytes(d->B));
    a.dgs = (float*)d->ws;
  }
  return DMT_OK;
}

}  // namespace

extern "C" int64_t dmt_mmoe_experts_ws_bytes(int32_t B) {
  if (B <= 0) return 0;
  return (int64_t)((size_t)B * MAXG * sizeof(float));
}

extern "C" int dmt_mmoe_experts_supported(int32_t u0, int32_t u1, int32_t u2, int32_t E, int32_t T) {
  return (u0 == U0 && u1 == U1 && u2 == U2 && E >= 1 && T >= 1 && T <= 4 && T * E <= MAXG) ? 1 : 0;
}

extern "C" int dmt_mmoe_experts_fwd(const dmt_mmoe_desc* d, void* stream) {
  MmoeArgs a;
  if (fill(a, d, "dmt_mmoe_experts_fwd") != DMT_OK) return DMT_ERR_ARG;
  DMT_CHECK_ARG(d->w1t && d->w2t && d->b1 && d->b2 && d->mix, "dmt_mmoe_experts_fwd: null argument");
  DMT_CHECK_ARG(d->w1t_ld % 8 == 0 && d->w2t_ld % 8 == 0 && d->w1t_expert_stride % 8 == 0 && d->w2t_expert_stride % 8 == 0 &&
                ((((uintptr_t)d->w1t) | ((uintptr_t)d->w2t)) & 15) == 0 && ((((uintptr_t)d->b1) | ((uintptr_t)d->b2)) & 15) == 0 &&
                d->b1_expert_stride % 4 == 0 && d->b2_expert_stride % 4 == 0, "dmt_mmoe_experts_fwd: weight / bias rows must be 16-byte aligned");
  if (a.dgs != nullptr) {
    const unsigned nb = (unsigned)((a.n_tiles + 7) / 8 * 8 * a.E);
    hipLaunchKernelGGL(mmoe_split_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(mmoe_mix_finish_kernel, dim3((unsigned)a.n_tiles), dim3(256), 0, (hipStream_t)stream, a);
    DMT_CHECK_LAUNCH("dmt_mmoe_experts_fwd(split)");
    return DMT_OK;
  }
  hipLaunchKernelGGL(mmoe_experts_fwd_kernel, dim3((unsigned)cdiv64(d->B, MR)), dim3(256), 0, (hipStream_t)stream, a);
  DMT_CHECK_LAUNCH("dmt_mmoe_experts_fwd");
  return DMT_OK;
}

extern "C" int dmt_mmoe_experts_bwd(const dmt_mmoe_desc* d, void* stream) {
  MmoeArgs a;
  if (fill(a, d, "dmt_mmoe_experts_bwd") != DMT_OK) return DMT_ERR_ARG;
  DMT_CHECK_ARG(d->w1 && d->w2 && d->dmix && d->dh1 && d->dh2 && d->dg1, "dmt_mmoe_experts_bwd: null argument");
  DMT_CHECK_ARG(d->lddg % 4 == 0 && (((uintptr_t)d->dg1) & 7) == 0 && d->w1_expert_stride % 8 == 0 && d->w2_expert_stride % 8 == 0 &&
                ((((uintptr_t)d->w1) | ((uintptr_t)d->w2)) & 15) == 0, "dmt_mmoe_experts_bwd: rows must be aligned (weights 16 B, dg1 8 B)");
  if (a.dgs != nullptr) {
    const unsigned nb = (unsigned)((a.n_tiles + 7) / 8 * 8 * a.E);
    hipLaunchKernelGGL(mmoe_split_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(mmoe_dgate_finish_kernel, dim3((unsigned)a.n_tiles), dim3(256), 0, (hipStream_t)stream, a);
    DMT_CHECK_LAUNCH("dmt_mmoe_experts_bwd(split)");
    return DMT_OK;
  }
  DMT_CHECK_ARG(!a.gate_dx, "dmt_mmoe_experts_bwd: gate_dx needs the workspace form");
  hipLaunchKernelGGL(mmoe_experts_bwd_kernel, dim3((unsigned)cdiv64(d->B, MR)), dim3(256), 0, (hipStream_t)stream, a);
  DMT_CHECK_LAUNCH("dmt_mmoe_experts_bwd");
  return DMT_OK;
}
