// How fast can a CU fill its LDS from an L2-resident buffer?  (a) LDS-DMA (buffer_load_dwordx4 ... lds), (b) global_load_dwordx4 to
// registers + ds_write_b128.  Every workgroup (one per CU, NW wavefronts) streams the same `bytes`-sized buffer REP times into a ring
// of LDS; nothing reads the LDS (pure fill rate).   build: hipcc --offload-arch=gfx950 -O3 lds_fill.hip -o lds_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int NW, int MODE, int INFLIGHT>
__global__ __launch_bounds__(NW * 64) void fill_kernel(const unsigned char* __restrict__ src, long long bytes, int rep, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[96 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, (int)bytes, 0x00020000);
  const long long per_round = (long long)NW * 1024;          // bytes one round of the workgroup moves
  const long long rounds = bytes / per_round;
  unsigned acc = 0;
  for (int r = 0; r < rep; ++r) {
    for (long long i = 0; i < rounds; i += INFLIGHT) {
      if constexpr (MODE == 0) {
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) {
          const long long off = ((i + k) % rounds) * per_round + wave * 1024;
          unsigned char* dst = smem + ((off) % (96 * 1024));
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vp)dst, 16, lane * 16, (int)off, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        u32x4_t v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) {
          const long long off = ((i + k) % rounds) * per_round + wave * 1024 + lane * 16;
          v[k] = *reinterpret_cast<const u32x4_t*>(src + off);
        }
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) {
          const long long off = ((i + k) % rounds) * per_round + wave * 1024 + lane * 16;
          *reinterpret_cast<u32x4_t*>(smem + (off % (96 * 1024))) = v[k];
        }
      }
    }
  }
  __syncthreads();
  acc = *reinterpret_cast<unsigned*>(smem + (threadIdx.x * 4) % (96 * 1024));
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int NW, int MODE, int INFLIGHT>
void run(const unsigned char* src, long long bytes, unsigned* sink, const char* name) {
  const int rep = 200, grid = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((fill_kernel<NW, MODE, INFLIGHT>), dim3(grid), dim3(NW * 64), 0, 0, src, bytes, 5, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((fill_kernel<NW, MODE, INFLIGHT>), dim3(grid), dim3(NW * 64), 0, 0, src, bytes, rep, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double tot = (double)bytes * rep * grid;
  printf("%-44s waves %d inflight %d: %8.1f GB/s per CU, %6.2f TB/s chip (%.3f ms)\n", name, NW, INFLIGHT, tot / grid / ms / 1e6, tot / ms / 1e9, ms);
}

int main() {
  const long long bytes = 2 * 1024 * 1024;      // the size of a dmt_chain2 weight image (L2-resident)
  unsigned char* src; unsigned* sink;
  hipMalloc(&src, bytes); hipMalloc(&sink, 64);
  hipMemset(src, 1, bytes);
  run<4, 0, 6>(src, bytes, sink, "LDS-DMA buffer_load_dwordx4 lds");
  run<8, 0, 6>(src, bytes, sink, "LDS-DMA buffer_load_dwordx4 lds");
  run<8, 0, 12>(src, bytes, sink, "LDS-DMA buffer_load_dwordx4 lds");
  run<12, 0, 6>(src, bytes, sink, "LDS-DMA buffer_load_dwordx4 lds");
  run<4, 1, 6>(src, bytes, sink, "global_load_dwordx4 + ds_write_b128");
  run<8, 1, 6>(src, bytes, sink, "global_load_dwordx4 + ds_write_b128");
  run<8, 1, 12>(src, bytes, sink, "global_load_dwordx4 + ds_write_b128");
  run<12, 1, 6>(src, bytes, sink, "global_load_dwordx4 + ds_write_b128");
  run<16, 1, 8>(src, bytes, sink, "global_load_dwordx4 + ds_write_b128");
  return 0;
}
