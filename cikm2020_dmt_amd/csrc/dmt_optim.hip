// tf.train.AdamOptimizer arithmetic: dense sweep and exact lazy (catch-up) sparse-row update.
// Both paths go through adam_update() with explicit fmaf so the lazy replay of zero-gradient steps is
// bit-identical to the dense sweep it replaces (tests/test_gpu_adam.py checks this bitwise).
#include "dmt_common.h"

namespace {

// var -= lr_t * m / (sqrt(v) + eps) with the hardware's square root and reciprocal (v_sqrt_f32, v_rcp_f32: 1 ulp each) and one fused
// multiply-add, instead of IEEE sqrtf and division (~10 instructions each on this ISA: scaling, Newton steps, fix-ups).  The replay of
// the lazy rows is a serial chain of these updates per element -- VALU-bound on exactly those two sequences (round 4: 0.78 GB moved in
// 0.51 ms) -- and the dense sweep, the sparse-row update and the replay all call THIS function, so lazy == dense stays a bitwise
// identity.  Against exact arithmetic the step differs by <= ~2 ulp of the UPDATE (not of p): 1e-7 of a step of size ~lr.
// (v below the normal range -- after ~88 000 zero-gradient steps -- : v_sqrt_f32 returns 0 for it; sqrt(v) < 1.1e-19 there, which
//  vanishes beside eps = 1e-8 in fp32 either way.)
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, float a, float c1, float c2, float eps) {
  m = fmaf(g - m, c1, m);               // m += (g - m) * (1 - beta1)
  v = fmaf(fmaf(g, g, -v), c2, v);      // v += (g*g - v) * (1 - beta2)
  const float r = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + eps);
  p = fmaf(-(m * a), r, p);
}

__global__ void adam_begin_kernel(float* state, float* lr_hist, int cap, float lr, float b1, float b2) {
  const float b1p = state[0], b2p = state[1];
  const float a = lr * sqrtf(1.f - b2p) / (1.f - b1p);
  int* istate = reinterpret_cast<int*>(state);
  const int step = istate[3] + 1;
  state[2] = a;
  istate[3] = step;
  if (lr_hist && step < cap) lr_hist[step] = a;
  (void)b1; (void)b2;
}

// restart the device-side step counter (the index into lr_hist) after a flush: every row is up to date at local step 0
__global__ __launch_bounds__(256) void adam_rebase_kernel(float* state, int* __restrict__ last_step, long long rows) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < rows) last_step[i] = 0;
  if (i == 0) reinterpret_cast<int*>(state)[3] = 0;
}

__global__ void adam_end_kernel(float* state, float b1, float b2) {
  state[0] = state[0] * b1;
  state[1] = state[1] * b2;
}

__global__ __launch_bounds__(256) void adam_dense_kernel(long long n, float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                         const float* __restrict__ g, float gscale, const float* __restrict__ state,
                                                         float b1, float b2, float eps, bf16_t* __restrict__ lp) {
  const float a = state[2];
  const float c1 = 1.f - b1, c2 = 1.f - b2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float pv = p[i], mv = m[i], vv = v[i];
    adam_update(pv, mv, vv, g[i] * gscale, a, c1, c2, eps);
    p[i] = pv; m[i] = mv; v[i] = vv;
    if (lp) lp[i] = f2bf(pv);
  }
}

// Replay of the zero-gradient steps from_step .. to_step of one element (what the dense sweep would have done to it meanwhile).
// With g = 0 the first moment shrinks by beta1 per step and the step size |lr_t m / (sqrt(v) + eps)| by ~0.9 per step, so after
// 100-200 steps an update no longer changes p in fp32 -- and never will again (the size keeps falling).  From there on only m and
// v move, geometrically:
//   * the steps up to that point are replayed with adam_update(), bit-identical to the dense sweep;
//   * at most DMT_ADAM_EXACT_TAIL further steps are replayed as the same fp32 products (still bit-identical);
//   * a longer tail is applied in closed form, m *= beta1^k, v *= beta2^k by repeated squaring: p is exactly what the dense sweep
//     leaves, m and v agree with its serial product to ~1e-5 relative (k roundings there, ~2 log2 k here).
// Without the bound a row unseen for 100 k steps cost 900 + 90 000 serial iterations per element.
constexpr int DMT_ADAM_EXACT_TAIL = 64;

__device__ __forceinline__ float pow_int(float b, int k) {
  float r = 1.f;
  while (k > 0) {
    if (k & 1) r *= b;
    b *= b;
    k >>= 1;
  }
  return r;
}

__device__ __forceinline__ void catch_up(float& pv, float& mv, float& vv, int from_step, int to_step, const float* __restrict__ lr_hist,
                                         float c1, float c2, float eps) {
  int s = from_step;
  for (; s <= to_step; ++s) {
    const float p0 = pv;
    adam_update(pv, mv, vv, 0.f, lr_hist[s], c1, c2, eps);
    if (pv == p0) { ++s; break; }               // p has stopped moving for good
  }
  const int rem = to_step - s + 1;
  if (rem <= 0) return;
  if (rem <= DMT_ADAM_EXACT_TAIL) {
    for (; s <= to_step; ++s) {
      mv = fmaf(0.f - mv, c1, mv);
      vv = fmaf(fmaf(0.f, 0.f, -vv), c2, vv);
    }
  } else {
    mv *= pow_int(1.f - c1, rem);
    vv *= pow_int(1.f - c2, rem);
  }
}

// The same replay for one ROW per wavefront (lane = element; `act`: this lane holds an element with non-zero moments).  The step sizes
// lr_t of 64 steps arrive in ONE coalesced load (lane i holds step base + i) and reach the loop by v_readlane with a scalar index: the
// element-wise form fetched lr_hist[s] inside the loop, one dependent trip to the L2 per replayed step.  Same operations per element,
// in the same order, as catch_up(): bit-identical results.  Must be called by all 64 lanes.
__device__ __forceinline__ void catch_up_wave(float& pv, float& mv, float& vv, bool act, int from_step, int to_step,
                                              const float* __restrict__ lr_hist, float c1, float c2, float eps, int lane) {
  int s_exit = to_step + 1;                       // first step this lane does NOT replay with adam_update()
  bool moving = act;
  for (int base = from_step; base <= to_step; base += 64) {
    if (__builtin_amdgcn_ballot_w64(moving) == 0ull) break;
    const int n = (to_step - base + 1) < 64 ? (to_step - base + 1) : 64;
    const float lrv = lane < n ? lr_hist[base + lane] : 0.f;
    for (int i = 0; i < n; ++i) {
      const float lr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lrv), i));
      if (moving) {
        const float p0 = pv;
        adam_update(pv, mv, vv, 0.f, lr, c1, c2, eps);
        if (pv == p0) { moving = false; s_exit = base + i + 1; }      // p has stopped moving for good
      }
      if (__builtin_amdgcn_ballot_w64(moving) == 0ull) break;
    }
  }
  if (!act) return;
  int s = s_exit;
  const int rem = to_step - s + 1;
  if (rem <= 0) return;
  if (rem <= DMT_ADAM_EXACT_TAIL) {
    for (; s <= to_step; ++s) {
      mv = fmaf(0.f - mv, c1, mv);
      vv = fmaf(fmaf(0.f, 0.f, -vv), c2, vv);
    }
  } else {
    mv *= pow_int(1.f - c1, rem);
    vv *= pow_int(1.f - c2, rem);
  }
}

// Row-sharded tables (tm.shard_w > 1): this rank holds the rows with global id % shard_w == shard_r, densely: local row
// (row - row_base[t]) / shard_w of table t (every row_base is a multiple of shard_w), last_step index row / shard_w.
__device__ __forceinline__ int tm_sw(const dmt_table_map& tm) { return tm.shard_w > 1 ? tm.shard_w : 1; }
__device__ __forceinline__ long long tm_elem(const dmt_table_map& tm, int t, long long row, int dim) {
  return tm.elem_off[t] + ((row - tm.row_base[t]) / tm_sw(tm)) * dim;
}
__device__ __forceinline__ bool tm_owned(const dmt_table_map& tm, long long row) { return tm.shard_w <= 1 || (row % tm.shard_w) == tm.shard_r; }

__device__ __forceinline__ int find_table(const dmt_table_map& tm, int row) {
  int t = 0;
  while (t + 1 < tm.n_tables && row >= tm.row_base[t + 1]) ++t;
  return t;
}

// Sparse-row kernels: grid-stride over the distinct rows (the count is device-side, so a capacity-sized grid would be
// mostly empty blocks), FOUR rows per wavefront: a 16-lane group owns one row and moves it in 16-byte pieces (dim % 4 == 0;
// other widths take the scalar tail), so a wave has 4 x (p, m, v, grad) row reads in flight instead of one.
constexpr int SPARSE_GRID = 256 * 8;

__device__ __forceinline__ float4 ld4(const float* q) { return *reinterpret_cast<const float4*>(q); }
__device__ __forceinline__ void st4(float* q, const float4& x) { *reinterpret_cast<float4*>(q) = x; }

// gradient rows as fp32 or (data-parallel wire format) bf16
__device__ __forceinline__ float4 ldg4(const float* q) { return ld4(q); }
__device__ __forceinline__ float4 ldg4(const bf16_t* q) {
  const uint2 u = *reinterpret_cast<const uint2*>(q);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
}
__device__ __forceinline__ float ldg1(const float* q) { return *q; }
__device__ __forceinline__ float ldg1(const bf16_t* q) { return bf2f(*q); }

template <typename GT>
__global__ __launch_bounds__(256) void adam_sparse_kernel(const dmt_table_map tm, float* __restrict__ p, float* __restrict__ m,
                                                          float* __restrict__ v, int* __restrict__ last_step,
                                                          const uint32_t* __restrict__ uniq, const int* __restrict__ n_uniq,
                                                          const GT* __restrict__ grad_rows, int max_dim, float gscale,
                                                          const float* __restrict__ state, const float* __restrict__ lr_hist,
                                                          float b1, float b2, float eps) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, c = lane & 15;
  const long long n = n_uniq[0];
  const int step = reinterpret_cast<const int*>(state)[3];
  const float a = state[2];
  const float c1 = 1.f - b1, c2 = 1.f - b2;
  const long long groups = (long long)gridDim.x * 16;   // 16-lane groups in the grid
  for (long long u = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + grp; u < n; u += groups) {
    const int row = (int)uniq[u];
    if ((uint32_t)row >= (uint32_t)tm.row_base[tm.n_tables]) continue;   // padding slot of a gathered (rank-major) row list
    const int t = find_table(tm, row);
    const int dim = tm.dim[t];
    if (!tm_owned(tm, row)) continue;
    const int lsi = row / tm_sw(tm);
    const int last = last_step[lsi];
    const long long base = tm_elem(tm, t, row, dim);
    const GT* gr = grad_rows + u * max_dim;
    if ((dim & 3) == 0 && (max_dim & 3) == 0) {
      for (int j = c * 4; j < dim; j += 64) {
        float4 pv = ld4(p + base + j), mv = ld4(m + base + j), vv = ld4(v + base + j);
        const float4 gv = ldg4(gr + j);
        if (last + 1 <= step - 1) {
          catch_up(pv.x, mv.x, vv.x, last + 1, step - 1, lr_hist, c1, c2, eps);
          catch_up(pv.y, mv.y, vv.y, last + 1, step - 1, lr_hist, c1, c2, eps);
          catch_up(pv.z, mv.z, vv.z, last + 1, step - 1, lr_hist, c1, c2, eps);
          catch_up(pv.w, mv.w, vv.w, last + 1, step - 1, lr_hist, c1, c2, eps);
        }
        adam_update(pv.x, mv.x, vv.x, gv.x * gscale, a, c1, c2, eps);
        adam_update(pv.y, mv.y, vv.y, gv.y * gscale, a, c1, c2, eps);
        adam_update(pv.z, mv.z, vv.z, gv.z * gscale, a, c1, c2, eps);
        adam_update(pv.w, mv.w, vv.w, gv.w * gscale, a, c1, c2, eps);
        st4(p + base + j, pv); st4(m + base + j, mv); st4(v + base + j, vv);
      }
    } else {
      for (int j = c; j < dim; j += 16) {
        float pv = p[base + j], mv = m[base + j], vv = v[base + j];
        catch_up(pv, mv, vv, last + 1, step - 1, lr_hist, c1, c2, eps);
        adam_update(pv, mv, vv, ldg1(gr + j) * gscale, a, c1, c2, eps);
        p[base + j] = pv; m[base + j] = mv; v[base + j] = vv;
      }
    }
    if (c == 0) last_step[lsi] = step;   // only this group touches the row (rows are distinct): the read above is long done
  }
}

// One WAVEFRONT per row, one lane per element (dim <= 64: one pass; wider rows in 64-element strides).  The replay length is a
// property of the ROW (its last-touch step), so the loop is wave-uniform up to the per-element early exit -- the four-rows-per-wave
// form paid 4 x max(age of its 4 rows) serial iterations per wavefront (each lane walked 4 elements one after the other); with real
// id gaps (bench.py --age-tables: median 6, tail > 100 steps) that was 0.9 ms of a 10.4 ms step.
// to_step < 0: replay through the last completed step (state.step).  stamp != null: rows whose stamp equals stamp_skip are left alone
// (rows the optimizer step IN FLIGHT is going to update itself: the early catch-up of the next batch, Trainer.train_step).
__global__ __launch_bounds__(256) void adam_catchup_kernel(const dmt_table_map tm, float* __restrict__ p, float* __restrict__ m,
                                                           float* __restrict__ v, int* __restrict__ last_step,
                                                           const uint32_t* __restrict__ uniq, const int* __restrict__ n_uniq,
                                                           const float* __restrict__ state, const float* __restrict__ lr_hist,
                                                           float b1, float b2, float eps, int to_step, const int* __restrict__ stamp,
                                                           int stamp_skip) {
  const int lane = threadIdx.x & 63;
  const long long n = n_uniq[0];
  const int step = to_step >= 0 ? to_step : reinterpret_cast<const int*>(state)[3];
  const float c1 = 1.f - b1, c2 = 1.f - b2;
  const long long waves = (long long)gridDim.x * 4;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // (scalar: the row loop and the replay loop are wave-uniform)
  for (long long u = (long long)blockIdx.x * 4 + wv; u < n; u += waves) {
    const int row = (int)uniq[u];
    if ((uint32_t)row >= (uint32_t)tm.row_base[tm.n_tables] || !tm_owned(tm, row)) continue;
    const int lsi = row / tm_sw(tm);
    const int last = last_step[lsi];
    if (last >= step) continue;
    if (stamp != nullptr && stamp[lsi] == stamp_skip) continue;
    const int t = find_table(tm, row);
    const int dim = tm.dim[t];
    const long long base = tm_elem(tm, t, row, dim);
    for (int j0 = 0; j0 < dim; j0 += 64) {
      const int j = j0 + lane;
      float pv = 0.f, mv = 0.f, vv = 0.f;
      if (j < dim) { pv = p[base + j]; mv = m[base + j]; vv = v[base + j]; }
      const bool act = mv != 0.f || vv != 0.f;            // (zero moments: p does not move and the moments stay zero)
      catch_up_wave(pv, mv, vv, act, last + 1, step, lr_hist, c1, c2, eps, lane);
      if (act) { p[base + j] = pv; m[base + j] = mv; v[base + j] = vv; }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) last_step[lsi] = step;
  }
}

// stamp[row] = value for every listed row (Trainer: "the optimizer step in flight updates this row itself")
__global__ __launch_bounds__(256) void rows_stamp_kernel(const dmt_table_map tm, const uint32_t* __restrict__ uniq, const int* __restrict__ n_uniq,
                                                         int* __restrict__ stamp, int value) {
  const long long n = n_uniq[0];
  for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < n; u += (long long)gridDim.x * 256) {
    const int row = (int)uniq[u];
    if ((uint32_t)row >= (uint32_t)tm.row_base[tm.n_tables] || !tm_owned(tm, row)) continue;
    stamp[row / tm_sw(tm)] = value;
  }
}

__global__ __launch_bounds__(256) void adam_flush_kernel(const dmt_table_map tm, float* __restrict__ p, float* __restrict__ m,
                                                         float* __restrict__ v, int* __restrict__ last_step,
                                                         const float* __restrict__ state, const float* __restrict__ lr_hist,
                                                         float b1, float b2, float eps, long long n_local) {
  const int lane = threadIdx.x & 63;
  const int step = reinterpret_cast<const int*>(state)[3];
  const float c1 = 1.f - b1, c2 = 1.f - b2;
  // grid-stride over the local rows: a 100 M-row table (BASELINE configs[3]) has 25 M four-row blocks = 6.4e9 threads, more than one
  // launch may carry (2^32); tests/test_gpu_configs.py::test_config3_* caught the one-block-per-four-rows form skipping rows
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (long long lsi = (long long)blockIdx.x * 4 + wv; lsi < n_local; lsi += (long long)gridDim.x * 4) {   // local row (= global row when not sharded)
    const long long row = tm.shard_w > 1 ? lsi * tm.shard_w + tm.shard_r : lsi;
    if (row >= tm.row_base[tm.n_tables]) break;
    // (a sharded layout pads every table to a multiple of shard_w rows; the padding rows have storage -- zeros -- and are skipped below)
    const int last = last_step[lsi];
    if (last >= step) continue;
    const int t = find_table(tm, (int)row);
    const int dim = tm.dim[t];
    for (int j0 = 0; j0 < dim; j0 += 64) {
      const int j = j0 + lane;
      const long long off = tm_elem(tm, t, row, dim) + j;
      float pv = 0.f, mv = 0.f, vv = 0.f;
      if (j < dim) { pv = p[off]; mv = m[off]; vv = v[off]; }
      const bool act = mv != 0.f || vv != 0.f;
      catch_up_wave(pv, mv, vv, act, last + 1, step, lr_hist, c1, c2, eps, lane);
      if (act) { p[off] = pv; m[off] = mv; v[off] = vv; }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) last_step[lsi] = step;
  }
}

}  // namespace

extern "C" int dmt_adam_begin_step(float* state, float* lr_hist, int32_t lr_hist_cap, float lr, float beta1, float beta2,
                                   void* stream) {
  DMT_CHECK_ARG(state != nullptr, "dmt_adam_begin_step: null state");
  hipLaunchKernelGGL(adam_begin_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, lr_hist, lr_hist_cap, lr, beta1, beta2);
  DMT_CHECK_LAUNCH("dmt_adam_begin_step");
  return DMT_OK;
}

extern "C" int dmt_adam_end_step(float* state, float beta1, float beta2, void* stream) {
  DMT_CHECK_ARG(state != nullptr, "dmt_adam_end_step: null state");
  hipLaunchKernelGGL(adam_end_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, beta1, beta2);
  DMT_CHECK_LAUNCH("dmt_adam_end_step");
  return DMT_OK;
}

extern "C" int dmt_adam_dense(int64_t n, float* p, float* m, float* v, const float* g, float grad_scale, const float* state,
                              float beta1, float beta2, float eps, void* lp_bf16, void* stream) {
  DMT_CHECK_ARG(n > 0 && p && m && v && g && state, "dmt_adam_dense: bad argument");
  long long nb = cdiv64(n, 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(adam_dense_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (long long)n, p, m, v, g, grad_scale,
                     state, beta1, beta2, eps, (bf16_t*)lp_bf16);
  DMT_CHECK_LAUNCH("dmt_adam_dense");
  return DMT_OK;
}

extern "C" int dmt_adam_sparse_rows(const dmt_table_map* tm, float* p, float* m, float* v, int32_t* last_step,
                                    const uint32_t* uniq_keys, const int32_t* n_uniq, int32_t max_uniq, const float* grad_rows,
                                    int32_t max_dim, float grad_scale, const float* state, const float* lr_hist, float beta1,
                                    float beta2, float eps, void* stream) {
  DMT_CHECK_ARG(tm && p && m && v && last_step && uniq_keys && n_uniq && grad_rows && state && lr_hist, "dmt_adam_sparse_rows: null argument");
  DMT_CHECK_ARG(tm->n_tables > 0 && tm->n_tables <= DMT_MAX_TABLES && max_uniq > 0, "dmt_adam_sparse_rows: bad table map / max_uniq");
  const long long need = cdiv64(max_uniq, 16);
  const unsigned nb = (unsigned)(need < SPARSE_GRID ? need : SPARSE_GRID);
  hipLaunchKernelGGL(adam_sparse_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *tm, p, m, v, last_step, uniq_keys, n_uniq,
                     grad_rows, max_dim, grad_scale, state, lr_hist, beta1, beta2, eps);
  DMT_CHECK_LAUNCH("dmt_adam_sparse_rows");
  return DMT_OK;
}

extern "C" int dmt_adam_sparse_rows_bf16(const dmt_table_map* tm, float* p, float* m, float* v, int32_t* last_step,
                                         const uint32_t* uniq_keys, const int32_t* n_uniq, int32_t max_uniq, const void* grad_rows_bf16,
                                         int32_t max_dim, float grad_scale, const float* state, const float* lr_hist, float beta1,
                                         float beta2, float eps, void* stream) {
  DMT_CHECK_ARG(tm && p && m && v && last_step && uniq_keys && n_uniq && grad_rows_bf16 && state && lr_hist, "dmt_adam_sparse_rows_bf16: null argument");
  DMT_CHECK_ARG(tm->n_tables > 0 && tm->n_tables <= DMT_MAX_TABLES && max_uniq > 0, "dmt_adam_sparse_rows_bf16: bad table map / max_uniq");
  DMT_CHECK_ARG(max_dim % 4 == 0 && ((uintptr_t)grad_rows_bf16 & 7) == 0, "dmt_adam_sparse_rows_bf16: rows must be 8-byte aligned, max_dim % 4 == 0");
  const long long need = cdiv64(max_uniq, 16);
  const unsigned nb = (unsigned)(need < SPARSE_GRID ? need : SPARSE_GRID);
  hipLaunchKernelGGL(adam_sparse_kernel<bf16_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *tm, p, m, v, last_step, uniq_keys, n_uniq,
                     reinterpret_cast<const bf16_t*>(grad_rows_bf16), max_dim, grad_scale, state, lr_hist, beta1, beta2, eps);
  DMT_CHECK_LAUNCH("dmt_adam_sparse_rows_bf16");
  return DMT_OK;
}

static int catchup_launch(const dmt_table_map* tm, float* p, float* m, float* v, int32_t* last_step, const uint32_t* uniq_keys,
                          const int32_t* n_uniq, int32_t max_uniq, const float* state, const float* lr_hist, float beta1, float beta2,
                          float eps, int to_step, const int32_t* stamp, int stamp_skip, void* stream, const char* who) {
  DMT_CHECK_ARG(tm && p && m && v && last_step && uniq_keys && n_uniq && state && lr_hist, "%s: null argument", who);
  DMT_CHECK_ARG(tm->n_tables > 0 && tm->n_tables <= DMT_MAX_TABLES && max_uniq > 0, "%s: bad table map / max_uniq", who);
  const long long need = cdiv64(max_uniq, 4);            // one wavefront per row, four per block
  const unsigned nb = (unsigned)(need < SPARSE_GRID ? need : SPARSE_GRID);
  hipLaunchKernelGGL(adam_catchup_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *tm, p, m, v, last_step, uniq_keys, n_uniq,
                     state, lr_hist, beta1, beta2, eps, to_step, stamp, stamp_skip);
  DMT_CHECK_LAUNCH(who);
  return DMT_OK;
}

extern "C" int dmt_adam_catchup_rows(const dmt_table_map* tm, float* p, float* m, float* v, int32_t* last_step,
                                     const uint32_t* uniq_keys, const int32_t* n_uniq, int32_t max_uniq, const float* state,
                                     const float* lr_hist, float beta1, float beta2, float eps, void* stream) {
  return catchup_launch(tm, p, m, v, last_step, uniq_keys, n_uniq, max_uniq, state, lr_hist, beta1, beta2, eps, -1, nullptr, 0, stream,
                        "dmt_adam_catchup_rows");
}

extern "C" int dmt_adam_catchup_rows_to(const dmt_table_map* tm, float* p, float* m, float* v, int32_t* last_step,
                                        const uint32_t* uniq_keys, const int32_t* n_uniq, int32_t max_uniq, const float* state,
                                        const float* lr_hist, float beta1, float beta2, float eps, int32_t to_step,
                                        const int32_t* stamp, int32_t stamp_skip, void* stream) {
  DMT_CHECK_ARG(to_step >= 0, "dmt_adam_catchup_rows_to: to_step must be >= 0");
  return catchup_launch(tm, p, m, v, last_step, uniq_keys, n_uniq, max_uniq, state, lr_hist, beta1, beta2, eps, to_step, stamp, stamp_skip,
                        stream, "dmt_adam_catchup_rows_to");
}

extern "C" int dmt_rows_stamp(const dmt_table_map* tm, const uint32_t* uniq_keys, const int32_t* n_uniq, int32_t max_uniq,
                              int32_t* stamp, int32_t value, void* stream) {
  DMT_CHECK_ARG(tm && uniq_keys && n_uniq && stamp && max_uniq > 0, "dmt_rows_stamp: bad argument");
  const long long need = cdiv64(max_uniq, 256);
  const unsigned nb = (unsigned)(need < 4096 ? need : 4096);
  hipLaunchKernelGGL(rows_stamp_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *tm, uniq_keys, n_uniq, stamp, value);
  DMT_CHECK_LAUNCH("dmt_rows_stamp");
  return DMT_OK;
}

extern "C" int dmt_adam_rebase(float* state, int32_t* last_step, int64_t rows, void* stream) {
  DMT_CHECK_ARG(state && last_step && rows > 0, "dmt_adam_rebase: bad argument");
  hipLaunchKernelGGL(adam_rebase_kernel, dim3((unsigned)cdiv64(rows, 256)), dim3(256), 0, (hipStream_t)stream, state, last_step, (long long)rows);
  DMT_CHECK_LAUNCH("dmt_adam_rebase");
  return DMT_OK;
}

extern "C" int dmt_adam_flush_rows(const dmt_table_map* tm, float* p, float* m, float* v, int32_t* last_step,
                                   const float* state, const float* lr_hist, float beta1, float beta2, float eps, void* stream) {
  DMT_CHECK_ARG(tm && p && m && v && last_step && state && lr_hist, "dmt_adam_flush_rows: null argument");
  const long long rows = cdiv64(tm->row_base[tm->n_tables], tm->shard_w > 1 ? tm->shard_w : 1);
  const long long nb_all = cdiv64(rows, 4);
  const unsigned nb = (unsigned)(nb_all < (1ll << 20) ? nb_all : (1ll << 20));          // (2^20 blocks x 256 threads < 2^32 threads per launch)
  hipLaunchKernelGGL(adam_flush_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *tm, p, m, v, last_step, state, lr_hist, beta1,
                     beta2, eps, rows);
  DMT_CHECK_LAUNCH("dmt_adam_flush_rows");
  return DMT_OK;
}

// ---- row-sharded tables: the owner's side of the forward exchange (BASELINE configs[3])
namespace {
// out[u, 0:dim] = p[row keys[u]] (fp32, row stride max_dim; columns >= dim are zeroed): the rows this rank owns, in request order
__global__ __launch_bounds__(256) void rows_gather_kernel(const dmt_table_map tm, const float* __restrict__ p, const uint32_t* __restrict__ keys,
                                                          long long n, float* __restrict__ out, int max_dim) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, c = lane & 15;
  const long long groups = (long long)gridDim.x * 16;
  for (long long u = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + grp; u < n; u += groups) {
    const uint32_t key = keys[u];
    float* o = out + u * max_dim;
    const bool ok = key < (uint32_t)tm.row_base[tm.n_tables] && tm_owned(tm, key);
    const int t = ok ? find_table(tm, (int)key) : 0;
    const int dim = ok ? tm.dim[t] : 0;
    const float* src = p + (ok ? tm_elem(tm, t, key, dim) : 0);
    for (int j = c; j < max_dim; j += 16) o[j] = j < dim ? src[j] : 0.f;
  }
}
}  // namespace

extern "C" int dmt_rows_gather(const dmt_table_map* tm, const float* p, const uint32_t* keys, int64_t n, float* out, int32_t max_dim,
                               void* stream) {
  DMT_CHECK_ARG(tm && p && (n == 0 || (keys && out)) && max_dim > 0, "dmt_rows_gather: bad argument");
  if (n == 0) return DMT_OK;
  long long nb = cdiv64(n, 16);
  if (nb > SPARSE_GRID) nb = SPARSE_GRID;
  hipLaunchKernelGGL(rows_gather_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, *tm, p, keys, (long long)n, out, max_dim);
  DMT_CHECK_LAUNCH("dmt_rows_gather");
  return DMT_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// The other five optimizers get_optimizer can return (model/inference_mlp.py:264-280), with the constructor defaults TF 1.12 gives
// them (the reference passes the learning rate only).  One update function per kind, shared by the dense sweep, the sparse-row
// update and the replay of zero-gradient steps, as for Adam above.  What a zero-gradient step does to an element, per kind:
//   sgd, adagrad       nothing (the accumulator only grows by g*g);
//   ftrl               recomputes var from (accum, linear, lr): the value it already holds for a row updated before at the same
//                      learning rate, a RESCALED one after the schedule has changed lr, and ZERO for a row that never has been
//                      updated (linear == 0): the reference's dense ApplyFtrl wipes every table row the first step does not touch
//                      -- dmt_opt_flush_rows after the first step, and after every step whose lr differs from the one before, does
//                      the same;
//   rmsprop            ms decays by `decay`, mom = momentum * mom = 0 (the default momentum 0.0 is the only one built): var unchanged;
//   adadelta           accum and accum_update decay by rho: var unchanged.
// So var never needs a replay before it is READ (no catch-up in front of the gather); rmsprop / adadelta replay the decay of their
// slots when the row is next UPDATED (or flushed): serially for up to DMT_OPT_EXACT_TAIL steps (bit-identical to the dense sweep),
// in closed form beyond.
namespace {

struct OptHP { float lr, h0, h1, h2; };
constexpr int DMT_OPT_EXACT_TAIL = 64;

template <int KIND>
__device__ __forceinline__ void opt_update(float& p, float& s0, float& s1, float g, const OptHP& hp) {
  if constexpr (KIND == DMT_OPT_SGD) {                  // ApplyGradientDescent: var -= grad * lr
    p = fmaf(-hp.lr, g, p);
  } else if constexpr (KIND == DMT_OPT_ADAGRAD) {       // ApplyAdagrad: accum += g*g; var -= g * lr * rsqrt(accum)
    s0 = fmaf(g, g, s0);
    p = fmaf(-(g * hp.lr), __builtin_amdgcn_rsqf(s0), p);
  } else if constexpr (KIND == DMT_OPT_ADADELTA) {      // ApplyAdadelta (rho = h0, epsilon = h1)
    s0 = fmaf(s0, hp.h0, g * g * (1.f - hp.h0));
    const float upd = __builtin_amdgcn_sqrtf(s1 + hp.h1) * __builtin_amdgcn_rsqf(s0 + hp.h1) * g;
    p = fmaf(-hp.lr, upd, p);
    s1 = fmaf(s1, hp.h0, upd * upd * (1.f - hp.h0));
  } else if constexpr (KIND == DMT_OPT_RMSPROP) {       // ApplyRMSProp (decay = h0, momentum = 0, epsilon = h2): s1 is `mom`
    s0 = fmaf(fmaf(g, g, -s0), 1.f - hp.h0, s0);
    s1 = (g * hp.lr) * __builtin_amdgcn_rsqf(s0 + hp.h2);
    p -= s1;
  } else {                                              // ApplyFtrl, learning_rate_power = -0.5 (l1 = h0, l2 = h1): s0 accum, s1 linear
    const float na = fmaf(g, g, s0);
    const float rs = __builtin_amdgcn_sqrtf(na);
    const float sigma = (rs - __builtin_amdgcn_sqrtf(s0)) / hp.lr;
    s1 += g - sigma * p;
    const float quad = rs / hp.lr + 2.f * hp.h1;
    const float x = (s1 > 0.f ? hp.h0 : -hp.h0) - s1;
    p = fabsf(s1) > hp.h0 ? x / quad : 0.f;
    s0 = na;
  }
}

__device__ __forceinline__ float decay_k(float x, float r, int k) {
  int i = 0;
  for (; i < k && i < DMT_OPT_EXACT_TAIL; ++i) x *= r;
  if (i < k) x *= pow_int(r, k - i);
  return x;
}

// k zero-gradient steps on the slots of one element (var is not touched by any of them, see above)
template <int KIND>
__device__ __forceinline__ void opt_idle(float& s0, float& s1, int k, const OptHP& hp) {
  if (k <= 0) return;
  if constexpr (KIND == DMT_OPT_ADADELTA) {
    s0 = decay_k(s0, hp.h0, k);
    s1 = decay_k(s1, hp.h0, k);
  } else if constexpr (KIND == DMT_OPT_RMSPROP) {
    int i = 0;
    for (; i < k && i < DMT_OPT_EXACT_TAIL; ++i) s0 = fmaf(-s0, 1.f - hp.h0, s0);    // ms += (0 - ms) * (1 - decay)
    if (i < k) s0 *= pow_int(hp.h0, k - i);
    s1 = 0.f;
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void opt_dense_kernel(long long n, float* __restrict__ p, float* __restrict__ s0, float* __restrict__ s1,
                                                        const float* __restrict__ g, float gscale, OptHP hp, bf16_t* __restrict__ lp) {
  constexpr bool U0 = KIND != DMT_OPT_SGD, U1 = KIND == DMT_OPT_ADADELTA || KIND == DMT_OPT_RMSPROP || KIND == DMT_OPT_FTRL;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float pv = p[i], a = U0 ? s0[i] : 0.f, b = U1 ? s1[i] : 0.f;
    opt_update<KIND>(pv, a, b, g[i] * gscale, hp);
    p[i] = pv;
    if (U0) s0[i] = a;
    if (U1) s1[i] = b;
    if (lp) lp[i] = f2bf(pv);
  }
}

// as adam_sparse_kernel: a 16-lane group per distinct row, 16-byte pieces
template <int KIND, typename GT>
__global__ __launch_bounds__(256) void opt_sparse_kernel(const dmt_table_map tm, float* __restrict__ p, float* __restrict__ s0,
                                                         float* __restrict__ s1, int* __restrict__ last_step,
                                                         const uint32_t* __restrict__ uniq, const int* __restrict__ n_uniq,
                                                         const GT* __restrict__ grad_rows, int max_dim, float gscale, int step, OptHP hp) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, c = lane & 15;
  const long long n = n_uniq[0];
  const long long groups = (long long)gridDim.x * 16;
  for (long long u = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + grp; u < n; u += groups) {
    const int row = (int)uniq[u];
    if ((uint32_t)row >= (uint32_t)tm.row_base[tm.n_tables]) continue;
    const int t = find_table(tm, row);
    const int dim = tm.dim[t];
    if (!tm_owned(tm, row)) continue;
    const int lsi = row / tm_sw(tm);
    const int idle = step - 1 - last_step[lsi];
    const long long base = tm_elem(tm, t, row, dim);
    const GT* gr = grad_rows + u * max_dim;
    if ((dim & 3) == 0 && (max_dim & 3) == 0) {
      for (int j = c * 4; j < dim; j += 64) {
        float4 pv = ld4(p + base + j), a = ld4(s0 + base + j), b = ld4(s1 + base + j);
        const float4 gv = ldg4(gr + j);
        opt_idle<KIND>(a.x, b.x, idle, hp); opt_idle<KIND>(a.y, b.y, idle, hp); opt_idle<KIND>(a.z, b.z, idle, hp); opt_idle<KIND>(a.w, b.w, idle, hp);
        opt_update<KIND>(pv.x, a.x, b.x, gv.x * gscale, hp);
        opt_update<KIND>(pv.y, a.y, b.y, gv.y * gscale, hp);
        opt_update<KIND>(pv.z, a.z, b.z, gv.z * gscale, hp);
        opt_update<KIND>(pv.w, a.w, b.w, gv.w * gscale, hp);
        st4(p + base + j, pv); st4(s0 + base + j, a); st4(s1 + base + j, b);
      }
    } else {
      for (int j = c; j < dim; j += 16) {
        float pv = p[base + j], a = s0[base + j], b = s1[base + j];
        opt_idle<KIND>(a, b, idle, hp);
        opt_update<KIND>(pv, a, b, ldg1(gr + j) * gscale, hp);
        p[base + j] = pv; s0[base + j] = a; s1[base + j] = b;
      }
    }
    if (c == 0) last_step[lsi] = step;
  }
}

// every local row brought to `step` (a wavefront per row): slots decayed (rmsprop / adadelta); ftrl: var recomputed from the slots
template <int KIND>
__global__ __launch_bounds__(256) void opt_flush_kernel(const dmt_table_map tm, float* __restrict__ p, float* __restrict__ s0,
                                                        float* __restrict__ s1, int* __restrict__ last_step, int step, OptHP hp,
                                                        long long n_local) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (long long lsi = (long long)blockIdx.x * 4 + wv; lsi < n_local; lsi += (long long)gridDim.x * 4) {
    const long long row = tm.shard_w > 1 ? lsi * tm.shard_w + tm.shard_r : lsi;
    if (row >= tm.row_base[tm.n_tables]) break;
    const int last = last_step[lsi];
    if (KIND != DMT_OPT_FTRL && last >= step) continue;
    const int t = find_table(tm, (int)row);
    const int dim = tm.dim[t];
    for (int j = lane; j < dim; j += 64) {
      const long long off = tm_elem(tm, t, row, dim) + j;
      if constexpr (KIND == DMT_OPT_FTRL) {
        // what a zero-gradient ApplyFtrl leaves: var as a function of (accum, linear) and THIS step's learning rate (the same
        // expression, on the same values, as the tail of opt_update: a row that is current is rewritten with what it holds)
        const float a = s0[off], b = s1[off];
        const float quad = __builtin_amdgcn_sqrtf(a) / hp.lr + 2.f * hp.h1;
        const float x = (b > 0.f ? hp.h0 : -hp.h0) - b;
        p[off] = fabsf(b) > hp.h0 ? x / quad : 0.f;
      } else {
        float a = s0[off], b = s1[off];
        opt_idle<KIND>(a, b, step - last, hp);
        s0[off] = a; s1[off] = b;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) last_step[lsi] = step;
  }
}

template <int KIND>
int opt_dense_launch(int64_t n, float* p, float* s0, float* s1, const float* g, float gscale, OptHP hp, void* lp, hipStream_t st) {
  long long nb = cdiv64(n, 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(opt_dense_kernel<KIND>, dim3((unsigned)nb), dim3(256), 0, st, (long long)n, p, s0, s1, g, gscale, hp, (bf16_t*)lp);
  DMT_CHECK_LAUNCH("dmt_opt_dense");
  return DMT_OK;
}

template <int KIND>
int opt_sparse_launch(const dmt_table_map* tm, float* p, float* s0, float* s1, int32_t* last_step, const uint32_t* uniq, const int32_t* n_uniq,
                      int32_t max_uniq, const void* rows, int bf16, int32_t max_dim, float gscale, int32_t step, OptHP hp, hipStream_t st) {
  const long long need = cdiv64(max_uniq, 16);
  const unsigned nb = (unsigned)(need < SPARSE_GRID ? need : SPARSE_GRID);
  if (bf16)
    hipLaunchKernelGGL((opt_sparse_kernel<KIND, bf16_t>), dim3(nb), dim3(256), 0, st, *tm, p, s0, s1, last_step, uniq, n_uniq,
                       reinterpret_cast<const bf16_t*>(rows), max_dim, gscale, step, hp);
  else
    hipLaunchKernelGGL((opt_sparse_kernel<KIND, float>), dim3(nb), dim3(256), 0, st, *tm, p, s0, s1, last_step, uniq, n_uniq,
                       reinterpret_cast<const float*>(rows), max_dim, gscale, step, hp);
  DMT_CHECK_LAUNCH("dmt_opt_sparse_rows");
  return DMT_OK;
}

template <int KIND>
int opt_flush_launch(const dmt_table_map* tm, float* p, float* s0, float* s1, int32_t* last_step, int32_t step, OptHP hp, hipStream_t st) {
  const long long total = tm->row_base[tm->n_tables];
  const long long n_local = tm->shard_w > 1 ? cdiv64(total, tm->shard_w) : total;
  long long nb = cdiv64(n_local, 4);
  if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(opt_flush_kernel<KIND>, dim3((unsigned)nb), dim3(256), 0, st, *tm, p, s0, s1, last_step, step, hp, n_local);
  DMT_CHECK_LAUNCH("dmt_opt_flush_rows");
  return DMT_OK;
}

bool opt_kind_ok(int kind) { return kind >= DMT_OPT_SGD && kind <= DMT_OPT_FTRL; }

}  // namespace

#define DMT_OPT_DISPATCH(kind, CALL)                                  \
  switch (kind) {                                                     \
    case DMT_OPT_SGD: return CALL(DMT_OPT_SGD);                       \
    case DMT_OPT_ADAGRAD: return CALL(DMT_OPT_ADAGRAD);               \
    case DMT_OPT_ADADELTA: return CALL(DMT_OPT_ADADELTA);             \
    case DMT_OPT_RMSPROP: return CALL(DMT_OPT_RMSPROP);               \
    default: return CALL(DMT_OPT_FTRL);                               \
  }

extern "C" int dmt_opt_dense(int32_t kind, int64_t n, float* p, float* s0, float* s1, const float* g, float grad_scale, float lr, float h0,
                             float h1, float h2, void* lp_bf16, void* stream) {
  DMT_CHECK_ARG(opt_kind_ok(kind), "dmt_opt_dense: unknown optimizer kind");
  DMT_CHECK_ARG(n > 0 && p && s0 && s1 && g, "dmt_opt_dense: bad argument");
  const OptHP hp{lr, h0, h1, h2};
#define CALL(K) opt_dense_launch<K>(n, p, s0, s1, g, grad_scale, hp, lp_bf16, (hipStream_t)stream)
  DMT_OPT_DISPATCH(kind, CALL)
#undef CALL
}

extern "C" int dmt_opt_sparse_rows(int32_t kind, const dmt_table_map* tm, float* p, float* s0, float* s1, int32_t* last_step,
                                   const uint32_t* uniq_keys, const int32_t* n_uniq, int32_t max_uniq, const void* grad_rows,
                                   int32_t grad_is_bf16, int32_t max_dim, float grad_scale, int32_t step, float lr, float h0, float h1,
                                   float h2, void* stream) {
  DMT_CHECK_ARG(opt_kind_ok(kind), "dmt_opt_sparse_rows: unknown optimizer kind");
  DMT_CHECK_ARG(tm && p && s0 && s1 && last_step && uniq_keys && n_uniq && grad_rows, "dmt_opt_sparse_rows: null argument");
  DMT_CHECK_ARG(tm->n_tables > 0 && tm->n_tables <= DMT_MAX_TABLES && max_uniq > 0 && step > 0, "dmt_opt_sparse_rows: bad table map / max_uniq / step");
  DMT_CHECK_ARG(!grad_is_bf16 || (max_dim % 4 == 0 && ((uintptr_t)grad_rows & 7) == 0), "dmt_opt_sparse_rows: bf16 rows must be 8-byte aligned, max_dim % 4 == 0");
  const OptHP hp{lr, h0, h1, h2};
#define CALL(K) opt_sparse_launch<K>(tm, p, s0, s1, last_step, uniq_keys, n_uniq, max_uniq, grad_rows, grad_is_bf16, max_dim, grad_scale, step, hp, (hipStream_t)stream)
  DMT_OPT_DISPATCH(kind, CALL)
#undef CALL
}

extern "C" int dmt_opt_flush_rows(int32_t kind, const dmt_table_map* tm, float* p, float* s0, float* s1, int32_t* last_step, int32_t step,
                                  float lr, float h0, float h1, float h2, void* stream) {
  DMT_CHECK_ARG(opt_kind_ok(kind), "dmt_opt_flush_rows: unknown optimizer kind");
  DMT_CHECK_ARG(tm && p && s0 && s1 && last_step && step >= 0, "dmt_opt_flush_rows: bad argument");
  DMT_CHECK_ARG(kind != DMT_OPT_FTRL || lr > 0.f, "dmt_opt_flush_rows: ftrl needs the learning rate of the last step");
  const OptHP hp{lr, h0, h1, h2};
#define CALL(K) opt_flush_launch<K>(tm, p, s0, s1, last_step, step, hp, (hipStream_t)stream)
  DMT_OPT_DISPATCH(kind, CALL)
#undef CALL
}

// ------------------------------------------------------------------------------------------------------------------------------
// Gradient of l2_norm (model/net/mmoe_transformer_unbias.py:42-60) on the reduced embedding-gradient rows: every embedding_list
// entry whose batch holds a row adds  coef * E[row]  to that row's gradient (coef = dLoss/dl2 * l2_emb_lambda / batch_size;
// mult[row] = the number of such entries, counted by dmt_l2_unique_rows_count).
namespace {
__global__ __launch_bounds__(256) void l2_rows_add_kernel(const dmt_table_map tm, const float* __restrict__ p, const uint32_t* __restrict__ uniq,
                                                          const int* __restrict__ n_uniq, const int* __restrict__ mult,
                                                          const float* __restrict__ coef, float* __restrict__ grad_rows, int max_dim) {
  const int lane = threadIdx.x & 63;
  const long long n = n_uniq[0];
  const float c = coef[0];
  for (long long u = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); u < n; u += (long long)gridDim.x * 4) {
    const int row = (int)uniq[u];
    if ((uint32_t)row >= (uint32_t)tm.row_base[tm.n_tables]) continue;
    const int k = mult[row];
    if (k == 0) continue;
    const int t = find_table(tm, row);
    const int dim = tm.dim[t];
    const long long base = tm_elem(tm, t, row, dim);
    for (int j = lane; j < dim; j += 64) grad_rows[u * max_dim + j] += c * (float)k * p[base + j];
  }
}
}  // namespace

extern "C" int dmt_l2_rows_add(const dmt_table_map* tm, const float* p, const uint32_t* uniq_keys, const int32_t* n_uniq, int32_t max_uniq,
                               const int32_t* mult, const float* coef, float* grad_rows, int32_t max_dim, void* stream) {
  DMT_CHECK_ARG(tm && p && uniq_keys && n_uniq && mult && coef && grad_rows && max_uniq > 0 && max_dim > 0, "dmt_l2_rows_add: bad argument");
  DMT_CHECK_ARG(tm->n_tables > 0 && tm->n_tables <= DMT_MAX_TABLES && tm->shard_w <= 1, "dmt_l2_rows_add: bad table map (replicated tables only)");
  const long long need = cdiv64(max_uniq, 4);
  hipLaunchKernelGGL(l2_rows_add_kernel, dim3((unsigned)(need < 4096 ? need : 4096)), dim3(256), 0, (hipStream_t)stream, *tm, p, uniq_keys,
                     n_uniq, mult, coef, grad_rows, max_dim);
  DMT_CHECK_LAUNCH("dmt_l2_rows_add");
  return DMT_OK;
}
