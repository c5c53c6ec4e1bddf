// Microbenchmark (GPU box): replay the sub-brick boxes dumped from the real camera forward
// (gpurun_out/dbg/boxes.bin: int32 [n][8] = ox oy oz ex ey ez block wave) with the same
// block / wave assignment: only LDS-DMA + wait per box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ int div_small(int n, float inv_d) { return (int) (((float) n + 0.5f) * inv_d); }

template <int MODE>
__global__ void __launch_bounds__(256, 3) replay(const float4* __restrict__ vol4, const int* __restrict__ boxes,
                                                 const int* __restrict__ first, float* __restrict__ out,
                                                 int X, int Y, unsigned long long* cyc) {
  __shared__ float4 smem[4 * 96 * 7 + 128];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float4* lds = smem + wave * 96 * 7;
  const int w = blockIdx.x * 4 + wave;
  const int b0 = first[w], b1 = first[w + 1];
  const int g = lane / 7, q = lane - g * 7;
  const bool act = q < 6 && g < 9;
  float acc = 0.f;
  unsigned long long t0 = clock64(), wsum = 0;
  for (int k = b0; k < b1; ++k) {
    const int* e = boxes + k * 8;
    const int ox = e[0], oy = e[1], oz = e[2], ex = e[3], ey = e[4], ez = e[5];
    const int rows = ex * ey * ez;
    const float inv_ex = 1.0f / (float) ex, inv_ey = 1.0f / (float) ey;
    const float4* src = vol4 + ((oz * Y + oy) * X + ox) * 6 + q;
    const int yst = X * 6, zst = Y * X * 6;
    for (int r0 = 0; r0 < rows; r0 += 9) {
      const int r = r0 + g;
      if (act && r < rows) {
        const int l = div_small(r, inv_ex), dx = r - l * ex;
        const int dz = div_small(l, inv_ey), dy = l - dz * ey;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (src + dz * zst + dy * yst + dx * 6),
                                          (__attribute__((address_space(3))) void*) (lds + r0 * 7), 16, 0, 0);
      }
    }
    unsigned long long ta = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long tb = clock64();
    wsum += tb - ta;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    acc += lds[(lane * 5 + k) % (rows * 7)].x;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 1) {      // compute between bricks, like the march
      float t = acc;
      for (int k2 = 0; k2 < 300; ++k2) t = __builtin_fmaf(t, 1.0001f, 0.5f);
      acc = t;
    }
  }
  unsigned long long t1 = clock64();
  if (lane == 0) { atomicAdd(cyc, t1 - t0); atomicAdd(cyc + 1, wsum); }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  FILE* f = fopen(argc > 1 ? argv[1] : "gpurun_out/dbg/boxes.bin", "rb");
  if (!f) { printf("no boxes file\n"); return 1; }
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  const int n = (int) (sz / 32);
  std::vector<int> raw(n * 8);
  if (fread(raw.data(), 32, n, f) != (size_t) n) return 1;
  fclose(f);
  const int nblock = 1056, nw = nblock * 4;
  const int shuffle = argc > 2 ? atoi(argv[2]) : 0;
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  if (shuffle == 1) {            // same boxes, dealt to random waves
    unsigned s = 12345;
    for (int i = n - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; int j = (s >> 8) % (i + 1); std::swap(order[i], order[j]); }
    for (int i = 0; i < n; ++i) { raw[order[i] * 8 + 6] = i % nblock; raw[order[i] * 8 + 7] = (i / nblock) % 4; }
  }
  if (shuffle == 2) {            // blocks renumbered without the XCD permutation: consecutive tiles round-robin over XCDs
    for (int i = 0; i < n; ++i) { int b = raw[i * 8 + 6]; int t = (b % 8) * (nblock / 8) + b / 8; raw[i * 8 + 6] = t; }
  }
  if (shuffle == 3) {            // each wave walks its boxes in reverse order (far to near)
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    return raw[a * 8 + 6] * 4 + raw[a * 8 + 7] < raw[b * 8 + 6] * 4 + raw[b * 8 + 7]; });
  std::vector<int> boxes(n * 8), first(nw + 1, 0);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < 8; ++j) boxes[i * 8 + j] = raw[order[i] * 8 + j];
    first[boxes[i * 8 + 6] * 4 + boxes[i * 8 + 7] + 1]++;
  }
  for (int i = 0; i < nw; ++i) first[i + 1] += first[i];
  const int X = 200, Y = 200, Z = 16;
  float4* vol; int *dboxes, *dfirst; float* out; unsigned long long* cyc;
  CK(hipMalloc(&vol, (size_t) X * Y * Z * 6 * 16)); CK(hipMemset(vol, 0, (size_t) X * Y * Z * 6 * 16));
  CK(hipMalloc(&dboxes, n * 32)); CK(hipMemcpy(dboxes, boxes.data(), n * 32, hipMemcpyHostToDevice));
  CK(hipMalloc(&dfirst, (nw + 1) * 4)); CK(hipMemcpy(dfirst, first.data(), (nw + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, nblock * 256 * 4)); CK(hipMalloc(&cyc, 16));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(cyc, 0, 16));
      CK(hipEventRecord(a));
      if (mode == 0) replay<0><<<nblock, 256>>>(vol, dboxes, dfirst, out, X, Y, cyc);
      else replay<1><<<nblock, 256>>>(vol, dboxes, dfirst, out, X, Y, cyc);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      unsigned long long c[2]; CK(hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost));
      if (rep == 2) printf("mode %d: %d boxes, kernel %.1f us, %.0f cycles per box (wave time), vmcnt wait %.0f cycles per box\n",
                           mode, n, ms * 1e3, (double) c[0] / n, (double) c[1] / n);
    }
  return 0;
}
