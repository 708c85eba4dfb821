// Issue rate of v_mfma_f32_32x32x16_bf16 as a function of the number of independent accumulator chains per wave and of the
// waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip && ./mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int NCH>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
  f32x16_t acc[NCH];
  for (int c = 0; c < NCH; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8 / NCH; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < NCH; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NCH> void run(int threads, float* d) {
  const int iters = 20000, blocks = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NCH><<<blocks, threads>>>(d, 10);
  hipEventRecord(e0);
  k<NCH><<<blocks, threads>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * threads / 64, mf = waves * iters * 8.0;
  const double flops = mf * 2.0 * 32 * 32 * 16;
  printf("chains %d  waves/WG %d : %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", NCH, threads / 64, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (mf / 1024.0));
}

int main() {
  float* d; hipMalloc(&d, 256 * 4 * 512 * 4);
  for (int threads : {256, 512}) { run<1>(threads, d); run<2>(threads, d); run<4>(threads, d); }
  return 0;
}
