// LDS float atomic (ds_add_f32) cost under address / bank conflicts, one wave and eight waves per CU.
// build: hipcc -O3 --offload-arch=gfx950 lds_atomic.hip -o lds_atomic ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(512) ki(int same, int bankstride, int iters, long long* out, float* sink) {
  __shared__ unsigned acc[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) acc[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int slot = (lane / same) * bankstride;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 16; ++c) atomicAdd(acc + ((slot + c * 67) & 4095), 1u + c);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; sink[blockIdx.x] = (float) acc[5]; }
}

template <bool ATOMIC>
__global__ void __launch_bounds__(512) k(int same, int bankstride, int iters, long long* out, float* sink) {
  __shared__ float acc[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // `same` lanes share one address; distinct addresses are bankstride words apart
  const int slot = (lane / same) * bankstride;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      float* p = acc + ((slot + c * 67) & 4095);
      if (ATOMIC) atomicAdd(p, 1.0f + c);
      else *p += 1.0f + c;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; sink[blockIdx.x] = acc[5]; }
}

int main() {
  long long* d; float* s;
  hipMalloc(&d, 4096 * sizeof(long long)); hipMalloc(&s, 4096 * sizeof(float));
  const int iters = 64;
  for (int threads : {64, 512}) {
    for (int atomic = 1; atomic >= 0; --atomic) {
      for (int bankstride : {1, 32}) {
        for (int same : {1, 2, 4, 8, 16, 64}) {
          if (bankstride == 32 && same != 1) continue;
          if (atomic) k<true><<<256, threads>>>(same, bankstride, iters, d, s);
          else k<false><<<256, threads>>>(same, bankstride, iters, d, s);
          hipDeviceSynchronize();
          std::vector<long long> h(256);
          hipMemcpy(h.data(), d, 256 * sizeof(long long), hipMemcpyDeviceToHost);
          double avg = 0; for (auto v : h) avg += v; avg /= 256;
          printf("threads %3d %s stride %2d lanes/address %2d : %8.1f cycles per wave-instruction\n", threads,
                 atomic ? "ds_add_f32" : "rd+add+wr ", bankstride, same, avg / (iters * 16));
        }
      }
    }
  }
  for (int threads : {64, 512})
    for (int same : {1, 4, 64}) {
      ki<<<256, threads>>>(same, 1, iters, d, s);
      hipDeviceSynchronize();
      std::vector<long long> h(256);
      hipMemcpy(h.data(), d, 256 * sizeof(long long), hipMemcpyDeviceToHost);
      double avg = 0; for (auto v : h) avg += v; avg /= 256;
      printf("threads %3d ds_add_u32 lanes/address %2d : %8.1f cycles per wave-instruction\n", threads, same, avg / (iters * 16));
    }
  return 0;
}
