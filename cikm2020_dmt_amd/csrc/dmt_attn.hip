// Short-sequence multi-head attention core, forward and backward: one wavefront per (example, head).
// Lane k owns key k (Tk <= 64): scores are lane-local dot products against the query row broadcast from
// LDS, the softmax max / sum are wavefront shuffles, and P.V runs with lanes along the head dim.
// Backward recomputes P from Q, K (flash-style: no [B,H,T,T] tensor ever touches HBM) and keeps the
// per-key dK / dV rows in registers.  Masking follows the reference literally (see dmt_hip.h).
#include "dmt_common.h"

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
};

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
    if (q < qlen) {
      const float dot = wave_sum(lane < Tk ? p * dP : 0.f);
      ds = (lane < klen) ? p * (dP - dot) / sc : 0.f;   // tf.where(key_mask, x, pad): no gradient into masked keys
    } else {
      p = PADDING_NUM;                                   // constant rows: gradient reaches V only
    }
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

int fill_args(AttnArgs& a, const dmt_attn_desc* d) {
  a.B = d->B; a.H = d->H; a.dh = d->dh; a.Tq = d->Tq; a.Tk = d->Tk;
  a.Q = d->Q; a.q_bs = d->q_bs; a.q_rs = d->q_rs;
  a.K = d->K; a.k_bs = d->k_bs; a.k_rs = d->k_rs;
  a.V = d->V; a.v_bs = d->v_bs; a.v_rs = d->v_rs;
  a.q_lens = d->q_lens; a.k_lens = d->k_lens;
  a.resid = d->resid; a.r_bs = d->r_bs; a.r_rs = d->r_rs;
  a.out = d->out; a.o_bs = d->o_bs; a.o_rs = d->o_rs;
  a.dout = nullptr; a.dQ = a.dK = a.dV = nullptr;
  a.do_bs = a.do_rs = a.dq_bs = a.dq_rs = a.dk_bs = a.dk_rs = a.dv_bs = a.dv_rs = 0;
  return 0;
}

int check_desc(const dmt_attn_desc* d, const char* who) {
  DMT_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  DMT_CHECK_ARG(d->dtype == DMT_F32 || d->dtype == DMT_BF16, "%s: bad dtype", who);
  DMT_CHECK_ARG(d->B > 0 && d->H > 0 && d->dh > 0 && d->Tq > 0 && d->Tk > 0, "%s: bad dims", who);
  DMT_CHECK_ARG(d->Q && d->K && d->V, "%s: null Q/K/V", who);
  if (d->Tk > 64) { dmt_set_error("%s: Tk=%d > 64 keys per wavefront is not supported yet", who, d->Tk); return DMT_ERR_UNSUPPORTED; }
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
  if (d->dtype == DMT_F32) launch_fwd<float>(a, nw, lds, st); else launch_fwd<bf16_t>(a, nw, lds, st);
  DMT_CHECK_LAUNCH("dmt_attn_fwd");
  return DMT_OK;
}

extern "C" int dmt_attn_bwd(const dmt_attn_bwd_desc* d, void* stream) {
  DMT_CHECK_ARG(d != nullptr, "dmt_attn_bwd: null descriptor");
  int rc = check_desc(&d->f, "dmt_attn_bwd");
  if (rc != DMT_OK) return rc;
  DMT_CHECK_ARG(d->dout && d->dQ && d->dK && d->dV, "dmt_attn_bwd: null gradient buffer");
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
  int r = (d->f.dtype == DMT_F32) ? launch_bwd<float>(a, nw, lds, st) : launch_bwd<bf16_t>(a, nw, lds, st);
  if (r != 0) { dmt_set_error("dmt_attn_bwd: head dim %d not instantiated (4,8,16,20,32,64,80)", d->f.dh); return DMT_ERR_UNSUPPORTED; }
  DMT_CHECK_LAUNCH("dmt_attn_bwd");
  return DMT_OK;
}
