// How long does the GPU take to dispatch N workgroups that leave after one flag load?  (The camera gather's
// x-run workgroups in accumulate mode: 22 400 at cfg-B, one in 25 with anything to add.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/empty_dispatch tools/microbench/empty_dispatch.hip && /tmp/empty_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) leave(const int* __restrict__ flags, int* __restrict__ out) {
  if (flags[blockIdx.x] == 0) return;
  out[blockIdx.x] = 1;
}
int main() {
  int *flags, *out;
  const int nmax = 1 << 18;
  hipMalloc(&flags, nmax * sizeof(int)); hipMalloc(&out, nmax * sizeof(int));
  hipMemset(flags, 0, nmax * sizeof(int));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int n : {1280, 2800, 5600, 11200, 22400, 44800, 89600}) {
    for (int i = 0; i < 5; ++i) leave<<<n, 256>>>(flags, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 50; ++i) leave<<<n, 256>>>(flags, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%6d workgroups of 256: %7.2f us per launch (%.2f ns per workgroup)\n", n, ms * 1e3 / 50, ms * 1e6 / 50 / n);
  }
  return 0;
}
