// Microbenchmark: what a dwordx4 gather instruction costs per CU as a function of (a) how many
// lanes are active, (b) how the lanes' addresses are spread (coalesced / random 16 B / random
// 96-byte rows read as 6 consecutive dwordx4).  Table is L2-resident (4 MiB) unless stated.
// hipcc --offload-arch=gfx950 -O3 vmem_gather.hip -o vmem_gather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// MODE 0: coalesced (lane i reads element base + i); 1: random 16-byte element per lane;
// 2: random 96-byte row per lane, N = 6 consecutive float4 of it
template <int MODE, int N>
__global__ void __launch_bounds__(256) k(const float4* __restrict__ tab, float* out, unsigned mask,
                                         int iters, int active) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  if (lane >= active) return;
  unsigned h = tid * 2654435761u;
  unsigned seq = (blockIdx.x * 256 + (threadIdx.x & ~63)) * 7u;
  for (int it = 0; it < iters; ++it) {
    float4 v[N];
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = tab[(seq + i * 4099u + lane) & mask];
      seq += 64 * 17;
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        h = h * 1664525u + 1013904223u;
        v[i] = tab[(h >> 8) & mask];
      }
    } else if (MODE >= 10) {
      // lane stride of (MODE - 10) float4: lane i reads element base + i * stride
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = tab[(seq * (MODE - 10) + i * 4099u + lane * (MODE - 10)) & mask];
      seq += 64 * 17;
    } else if (MODE == 3) {
      // random 128-byte-aligned row per lane, float4 number i of it: all lanes share the offset in the line
      h = h * 1664525u + 1013904223u;
      const unsigned row = ((h >> 8) & mask) / 8 * 8;
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = tab[row + i];
    } else {
      h = h * 1664525u + 1013904223u;
      const unsigned row = min(((h >> 8) & mask) / 6 * 6, mask - 5);
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = tab[row + i];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) acc += v[i].x + v[i].w;
  }
  if (acc == 12345.678f) out[tid] = acc;
}

template <int MODE, int N>
void run(const char* name, const float4* tab, float* out, unsigned mask, int active) {
  const int blocks = 256 * 8 * 2, iters = 64;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, N><<<blocks, 256>>>(tab, out, mask, iters, active);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<MODE, N><<<blocks, 256>>>(tab, out, mask, iters, active);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double instr = (double) blocks * 4 * iters * N;
  const double per_cu = instr / 256.0;
  printf("%-22s lanes %2d  %7.3f ms  %6.1f clk per wave-instr per CU  %6.2f TB/s useful\n", name, active, ms,
         ms * 1e6 / per_cu * 2.4, instr * active * 16 / ms / 1e9);
}

int main() {
  const size_t bytes = 4 << 20;
  float4* tab; float* out;
  hipMalloc(&tab, 64 << 20); hipMemset(tab, 0, 64 << 20); hipMalloc(&out, 256 * 8 * 2 * 256 * 4);
  const unsigned m4 = (unsigned) (bytes / 16 - 1), m64 = (unsigned) ((64u << 20) / 16 - 1);
  for (int act : {64, 32, 16, 8}) run<0, 8>("coalesced dwordx4", tab, out, m4, act);
  for (int act : {64, 32, 16, 8}) run<1, 8>("random 16 B (4 MiB)", tab, out, m4, act);
  for (int act : {64, 32, 16}) run<2, 6>("random 96 B rows (4 MiB)", tab, out, m4, act);
  run<3, 6>("random 128 B-aligned rows", tab, out, m4, 64);
  run<10 + 1, 8>("lane stride 16 B", tab, out, m4, 64);
  run<10 + 2, 8>("lane stride 32 B", tab, out, m4, 64);
  run<10 + 4, 8>("lane stride 64 B", tab, out, m4, 64);
  run<10 + 8, 8>("lane stride 128 B", tab, out, m4, 64);
  run<10 + 6, 8>("lane stride 96 B", tab, out, m4, 64);
  for (int act : {64, 32}) run<1, 8>("random 16 B (64 MiB)", tab, out, m64, act);
  for (int act : {64, 32}) run<2, 6>("random 96 B rows (64 MiB)", tab, out, m64, act);
  return 0;
}
