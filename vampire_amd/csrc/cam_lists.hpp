// The two work lists of the camera backward's scatter stage (render_bwd_cell.hip), and the constants its kernels share.
//   cells  the cells with more than kCellHeavy records (cam_cell_splat_kernel's items), as chunks.  A workgroup takes a
//          scan tile of 2 048 cells, eight per thread, compacts its heavy cells' chunks and appends them with ONE atomic.
//   runs   the gather's x-runs (32 voxels) that hold at least one record.  When the gather adds on top of the BEV
//          branch's gradient (the default) it visits those only -- with early ray termination most of the volume
//          lies behind terminated rays (cfg-B: 925 of 22 400 runs), and a workgroup per run that leaves at once
//          still costs its dispatch.  Thread = run: the records in reach of a run's voxels are four ranges of 33
//          x-neighbouring cells, i.e. eight start offsets; one append per wave.
// (Rounds 3 - 5 walked the voxels, a thread each, for run FLAGS: 0.7 M threads and 16 offset loads per thread; an
// append per run took 125 us when nothing terminates.)
// The lists need the scanned cells and are needed by the kernels BEHIND the per-ray pass, so they are built by that
// pass's launch: its workgroups past the ray tiles (render_bwd_ray.hip).  As a launch of their own between the scan
// and the per-ray pass (through the first builds of round 6) their 409 small workgroups cost a replayed step 13 us:
// 9 of kernel and 4 - 5 of hand-over behind a kernel that short.
#pragma once
#include "render_common.hpp"

namespace vamp {

#ifndef VAMP_CELL_HEAVY
#define VAMP_CELL_HEAVY 32
#endif
constexpr int kCellHeavy = VAMP_CELL_HEAVY;   // records per cell beyond which the cell is summed once per corner (cam_cell_splat_kernel)
#ifndef VAMP_SPLAT_CHUNK
#define VAMP_SPLAT_CHUNK 256
#endif
// A heavy cell of n records is summed in ceil(n / kSplatChunk) chunks of n / chunks records (the last one takes the
// remainder), each a list entry and a workgroup of the splat with a row set of its own: slot (first record of the
// chunk) / kCellHeavy -- chunks are at least kSplatChunk / 2 >= kCellHeavy records long, so no two share a slot.
constexpr int kSplatChunk = VAMP_SPLAT_CHUNK;
static_assert(kSplatChunk / 2 >= kCellHeavy, "chunks must not share slots");
__host__ __device__ inline int splat_chunks(int n) { return (n + kSplatChunk - 1) / kSplatChunk; }
#ifndef VAMP_SPLAT_NW
#define VAMP_SPLAT_NW 2
#endif
#ifndef VAMP_GATHER_GRID
#define VAMP_GATHER_GRID 20480
#endif
constexpr int kGatherGrid = VAMP_GATHER_GRID;   // workgroups of the gather at most (a workgroup takes every kGatherGrid-th listed x-run)
constexpr int kRunVox = 32;                  // voxels (an x-run) per gather workgroup
constexpr int kListCells = kScanTile;        // cells per list-building workgroup of the cell part

struct CamListArgs {
  const int *off, *boff;   // cell start = off[c] + boff[c / kScanTile]
  int2* hcells;            // out: {first record, records} per heavy-cell chunk
  int* nhcells;            // out: their number (zero on entry: the scan's duty)
  int* runs;               // out: the x-runs with records
  int* nruns;              // out: their number (zero on entry)
  long ncell_b, total_runs, ncell;
  int runs_x;
  unsigned cell_blocks;    // list-building workgroups of the cell part
  unsigned nblocks;        // ... and in all
  unsigned first_block;    // blockIdx.x of the first of them in the launch that hosts them
};
// render_bwd_cell.hip
CamListArgs cam_list_args(const VampRenderDesc* d, void* cell_scratch);

__device__ __forceinline__ int cell_start(const int* __restrict__ off, const int* __restrict__ boff, long c) {
  return off[c] + boff[c / kScanTile];
}

// one list-building workgroup (256 threads); bid in [0, A.nblocks)
__device__ __forceinline__ void cam_lists_block(const RenderParams& P, const CamListArgs& A, unsigned bid) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (bid < A.cell_blocks) {
    // cells c0 .. c0 + 7 of this thread (the last cell with a successor is A.ncell - 2: cell_count_padded)
    const long c0 = (long) bid * kListCells + 8 * threadIdx.x;
    int st[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) st[k] = cell_start(A.off, A.boff, min(c0 + k, A.ncell - 1));
    unsigned hv = 0;
    int mine = 0;                       // list entries of this thread: the chunks of its heavy cells
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool heavy = c0 + k < A.ncell - 1 && st[k + 1] - st[k] > kCellHeavy;
      hv |= heavy ? 1u << k : 0u;
      mine += heavy ? splat_chunks(st[k + 1] - st[k]) : 0;
    }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    __shared__ int wsum[4], base_s;
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int n = wsum[0] + wsum[1] + wsum[2] + wsum[3];
      base_s = n ? atomicAdd(A.nhcells, n) : 0;
    }
    __syncthreads();
    int at = base_s + incl - mine;
    for (int k = 0; k < wv; ++k) at += wsum[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (!(hv & (1u << k))) continue;
      const int nrec = st[k + 1] - st[k], nchunk = splat_chunks(nrec), len = nrec / nchunk;
      for (int j = 0; j < nchunk; ++j)
        A.hcells[at++] = make_int2(st[k] + j * len, j == nchunk - 1 ? nrec - j * len : len);
    }
    return;
  }
  const long run = (long) (bid - A.cell_blocks) * 256 + threadIdx.x;
  const bool run_ok = run < A.total_runs;
  const long rc = run_ok ? run : A.total_runs - 1;
  const int bx = (int) (rc % A.runs_x);
  const long rest = rc / A.runs_x;
  const int iy = (int) (rest % P.Y);
  const long zb = rest / P.Y;
  const int iz = (int) (zb % P.Z);
  const long b = zb / P.Z;
  const int ix0 = bx * kRunVox, span = min(kRunVox, P.X - ix0) + 1;      // cells ix0 .. ix0 + span - 1
  int tot = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long c = b * A.ncell_b + ((long) (iz + (r >> 1)) * (P.Y + 1) + (iy + (r & 1))) * (P.X + 1) + ix0;
    tot += cell_start(A.off, A.boff, c + span) - cell_start(A.off, A.boff, c);
  }
  const bool act = run_ok && tot > 0;
  const unsigned long long m = __ballot(act);
  if (m == 0ull) return;
  const int lead = __ffsll((long long) m) - 1;
  int base = 0;
  if (lane == lead) base = atomicAdd(A.nruns, __popcll(m));
  base = __shfl(base, lead, 64);
  if (act) A.runs[base + __popcll(m & ((1ull << lane) - 1ull))] = (int) run;
}

}  // namespace vamp
