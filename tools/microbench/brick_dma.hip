// Microbenchmark (GPU box): cost of staging voxel bricks global -> LDS, per wave-instruction.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/brick_dma.hip -o tools/microbench/brick_dma && ./brick_dma
// Variants: 0 LDS-DMA 9 rows x 6 float4 (7th lane of a row masked), brick-shaped addresses
//           1 same addresses through registers (global_load_dwordx4 + ds_write_b128)
//           2 LDS-DMA, 64 lanes x 16 B contiguous (1 KiB per instruction)
//           3 LDS-DMA 8 rows x 8 float4, 128-B rows (aligned)
//           4 as 0 but rows padded to 128 B in memory (6 of 8 lanes active)
//           5 as 1 but one row per lane-sextet replaced by dwordx3+... (skip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VAR>
__global__ void __launch_bounds__(256) k(const float4* __restrict__ vol, float* __restrict__ out, int iters,
                                         int X, int Y, int Z, unsigned long long* cyc) {
  __shared__ float4 lds[4 * 96 * 8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4* my = lds + wave * 96 * 8;
  unsigned seed = (blockIdx.x * 4 + wave) * 2654435761u + 12345u;
  float acc = 0.f;
  const int ex = 5, ey = 4, ez = 3;                 // 60 rows
  const int RW = (VAR == 3 || VAR == 4) ? 8 : 6;     // float4 per row in memory
  unsigned long long wsum = 0;
  unsigned long long t0 = clock64();
  if (VAR == 9 && (wave & 1)) {
    // the odd waves run dense independent VALU work instead (the other waves stage bricks as in 5)
    float a0 = seed, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    for (int k2 = 0; k2 < iters * 900; ++k2) {
      a0 = __builtin_fmaf(a0, 1.0001f, 0.5f); a1 = __builtin_fmaf(a1, 1.0001f, 0.5f);
      a2 = __builtin_fmaf(a2, 1.0001f, 0.5f); a3 = __builtin_fmaf(a3, 1.0001f, 0.5f);
      a4 = __builtin_fmaf(a4, 1.0001f, 0.5f); a5 = __builtin_fmaf(a5, 1.0001f, 0.5f);
      a6 = __builtin_fmaf(a6, 1.0001f, 0.5f); a7 = __builtin_fmaf(a7, 1.0001f, 0.5f);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    return;
  }
  for (int it = 0; it < iters; ++it) {
    seed = seed * 1664525u + 1013904223u;
    const int ox = (seed >> 8) % (X - ex), oy = (seed >> 16) % (Y - ey), oz = (seed >> 24) % (Z - ez);
    const float4* src = vol + (long) ((oz * Y + oy) * X + ox) * RW;
    if (VAR == 5 || VAR == 6 || VAR == 7 || VAR == 8 || VAR == 9 || VAR == 10) {
      // variable brick shapes, per-XCD locality (blocks b, b + 8, ... share a region), dummy work between bricks
      const int ex2 = 2 + (seed >> 3) % 5, ey2 = 2 + (seed >> 6) % 5, ez2 = 2 + (seed >> 9) % 3;
      const int reg = blockIdx.x & 7;
      const int oy2 = reg * (Y / 8) + (seed >> 16) % (Y / 8 - ey2);
      const int ox2 = (seed >> 8) % (X - ex2), oz2 = (seed >> 24) % (Z - ez2);
      const float4* src2 = vol + (long) ((oz2 * Y + oy2) * X + ox2) * RW;
      const int rows = ex2 * ey2 * ez2;
      const int g = lane / 7, q = lane % 7;
      for (int r0 = 0; r0 < rows; r0 += 9) {
        const int r = r0 + g;
        if (q < 6 && g < 9 && r < rows) {
          const int l = r / ex2, dx = r % ex2, dz = l / ey2, dy = l % ey2;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (src2 + ((dz * Y + dy) * X + dx) * RW + q),
                                            (__attribute__((address_space(3))) void*) (my + r0 * 7), 16, 0, 0);
        }
      }
      if (VAR == 6) {
        float t = acc;
        for (int k2 = 0; k2 < 600; ++k2) t = __builtin_fmaf(t, 1.0001f, 0.5f);
        acc = t;
      }
      if (VAR == 10) {
        // wait at once (measured), THEN a long compute phase: the memory pipe idles most of the time
        unsigned long long ta = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long tb = clock64();
        wsum += tb - ta;
        float t = acc;
        for (int k2 = 0; k2 < 2500; ++k2) t = __builtin_fmaf(t, 1.0001f, 0.5f);
        acc = t;
      }
      if (VAR == 7 || VAR == 8) {
        // LDS gather traffic of the same wave beside the DMA in flight: 48 ds_read_b128 from a second region
        const float4* other = lds + ((wave + (VAR == 8 ? 0 : 2)) & 3) * 96 * 8 + (VAR == 8 ? 400 : 0);
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k2 = 0; k2 < 48; ++k2) {
          const float4 f = other[((lane * 7 + k2 * 13 + it) % 300)];
          t.x += f.x; t.y += f.y; t.z += f.z; t.w += f.w;
        }
        acc += t.x + t.y + t.z + t.w;
      }
    } else if (VAR == 0 || VAR == 4) {
      const int g = lane / 7, q = lane % 7;
      for (int r0 = 0; r0 < 60; r0 += 9) {
        const int r = r0 + g;
        if (q < 6 && g < 9 && r < 60) {
          const int l = r / ex, dx = r % ex, dz = l / ey, dy = l % ey;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (src + ((dz * Y + dy) * X + dx) * RW + q),
                                            (__attribute__((address_space(3))) void*) (my + r0 * 7), 16, 0, 0);
        }
      }
    } else if (VAR == 1) {
      const int g = lane / 6, q = lane % 6;
      for (int r0 = 0; r0 < 60; r0 += 10) {
        const int r = r0 + g;
        if (g < 10 && r < 60) {
          const int l = r / ex, dx = r % ex, dz = l / ey, dy = l % ey;
          my[r * 7 + q] = src[((dz * Y + dy) * X + dx) * RW + q];
        }
      }
    } else if (VAR == 2) {
      for (int p = 0; p < 6; ++p)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (src + p * 4096 + lane),
                                          (__attribute__((address_space(3))) void*) (my + p * 64), 16, 0, 0);
    } else if (VAR == 3) {
      const int g = lane / 8, q = lane % 8;
      for (int r0 = 0; r0 < 60; r0 += 8) {
        const int r = r0 + g;
        if (r < 60) {
          const int l = r / ex, dx = r % ex, dz = l / ey, dy = l % ey;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (src + ((dz * Y + dy) * X + dx) * RW + q),
                                            (__attribute__((address_space(3))) void*) (my + r0 * 8), 16, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    acc += my[(lane * 5 + it) % 400].x;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = clock64();
  if (lane == 0) { atomicAdd(cyc, t1 - t0); atomicAdd(cyc + 1, wsum); }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  const int X = 200, Y = 200, Z = 16;
  const size_t n4 = (size_t) X * Y * Z * 8;
  float4* vol; float* out; unsigned long long* cyc;
  CK(hipMalloc(&vol, n4 * 16)); CK(hipMemset(vol, 0, n4 * 16));
  CK(hipMalloc(&out, 3072 * 256 * 4)); CK(hipMalloc(&cyc, 16));
  const int iters = 200, grid = 768;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int var = 9; var < 11; ++var) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(cyc, 0, 16));
      CK(hipEventRecord(a));
      switch (var) {
        case 0: k<0><<<grid, 256>>>(vol, out, iters, X, Y, Z, cyc); break;
        case 1: k<1><<<grid, 256>>>(vol, out, iters, X, Y, Z, cyc); break;
        case 2: k<2><<<grid, 256>>>(vol, out, iters, X, Y, Z, cyc); break;
        case 3: k<3><<<grid, 256>>>(vol, out, iters, X, Y, Z, cyc); break;
        case 4: k<4><<<grid, 256>>>(vol, out, iters, X, Y, Z, cyc); break;
        case 5: k<5><<<grid, 256>>>(vol, out, iters, X, Y, Z, cyc); break;
        case 6: k<6><<<grid, 256>>>(vol, out, iters, X, Y, Z, cyc); break;
        case 7: k<7><<<grid, 256>>>(vol, out, iters, X, Y, Z, cyc); break;
        case 8: k<8><<<grid, 256>>>(vol, out, iters, X, Y, Z, cyc); break;
        case 9: k<9><<<grid, 256>>>(vol, out, iters, X, Y, Z, cyc); break;
        case 10: k<10><<<grid, 256>>>(vol, out, iters, X, Y, Z, cyc); break;
      }
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      unsigned long long cc[2]; CK(hipMemcpy(cc, cyc, 16, hipMemcpyDeviceToHost));
      unsigned long long c = cc[0];
      if (rep == 1 && var == 10) printf("   variant 10: vmcnt wait right after issue: %.0f cycles per brick\n", (double) cc[1] / ((double) grid * 4 * iters));
      const double bricks = (double) grid * (var == 9 ? 2 : 4) * iters;
      if (rep == 1)
        printf("variant %d: %.1f us, %.0f cycles per brick per wave (60 rows), %.2f TB/s of brick bytes (60 x 96 B)\n", var,
               ms * 1e3, (double) c / bricks, bricks * 60 * 96 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
