// Microbenchmark: cost of one vector-memory instruction per CU for dword / dwordx2 / dwordx4
// loads that hit in L1/L2 (small table), full occupancy.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <typename V, int N>
__global__ void __launch_bounds__(256) k(const V* __restrict__ tab, float* out, int mask, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  int idx = tid & mask;
  for (int it = 0; it < iters; ++it) {
    V v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = tab[(idx + i * 4099) & mask];
#pragma unroll
    for (int i = 0; i < N; ++i) acc += reinterpret_cast<const float*>(&v[i])[0];
    idx = (idx + 64 * 17) & mask;
  }
  if (acc == 12345.678f) out[tid] = acc;
}

template <typename V, int N>
void run(const char* name, const void* tab, float* out, int elems_mask) {
  const int blocks = 256 * 8 * 4, iters = 64;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<V, N><<<blocks, 256>>>((const V*) tab, out, elems_mask, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<V, N><<<blocks, 256>>>((const V*) tab, out, elems_mask, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double instr = (double) blocks * 4 * iters * N;          // wave-level load instructions
  const double per_cu = instr / 256.0;
  printf("%-10s N=%2d  %.3f ms  %.1f ns per wave-instr per CU (%.1f clk @2.4GHz)  %.2f TB/s\n", name, N, ms,
         ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.4, instr * 64 * sizeof(V) / ms / 1e9);
}

int main() {
  const size_t bytes = 1 << 20;                                 // 1 MiB table: L2-resident
  void* tab; float* out;
  hipMalloc(&tab, bytes); hipMemset(tab, 0, bytes); hipMalloc(&out, 256 * 8 * 4 * 256 * 4);
  run<float, 8>("dword", tab, out, (int) (bytes / 4 - 1));
  run<float, 16>("dword", tab, out, (int) (bytes / 4 - 1));
  run<float2, 8>("dwordx2", tab, out, (int) (bytes / 8 - 1));
  run<float4, 8>("dwordx4", tab, out, (int) (bytes / 16 - 1));
  run<float4, 4>("dwordx4", tab, out, (int) (bytes / 16 - 1));
  // tiny table: L1-resident (16 KiB)
  run<float, 8>("dword-L1", tab, out, 4095);
  run<float4, 8>("dwx4-L1", tab, out, 1023);
  return 0;
}
