// Library-wide state: last-error string, ABI version, and the optional HIP-event
// kernel timer used by bench.py's roofline leg.
#include "common.hpp"

#include <algorithm>
#include <mutex>
#include <vector>

namespace vamp {

thread_local char g_err[512] = "";

namespace {
struct Pair {
  hipEvent_t a, b;
  int slot;
};
std::mutex g_mu;
bool g_on = false;
int g_only = -1;                   // >= 0: time this slot only (keeps the timed region light)
std::vector<Pair> g_pairs;        // recorded since enable(1)
std::vector<Pair> g_free;         // recycled events
const char* g_names[kProfSlots] = {
    "feat_to_channel_last", "lift_fwd", "lift_bwd_gather", "feat_to_channel_first", "lift_fwd_dense",
    "lift_bwd_dense", "pack_volume", "render_cam_fwd", "render_bev_fwd", "render_cam_bwd_ray",
    "unpack_grad", "render_bev_bwd_scan", "memset", "aux", "render_cam_bwd_gather", "render_bev_fwd_channels",
    "render_bev_bwd_q", "render_bev_bwd_gather", "lift_bwd_v1", "lift_bwd_count",
    "lift_bwd_fill", "render_cam_bwd_rank", "render_cam_bwd_fill", "render_cam_bwd_heavy", "render_cam_bwd_v1",
    "depth_softmax", "density_gate", "upsample_trilinear",
    "conv3d_fwd", "conv3d_dgrad", "conv3d_wgrad", "render_cam_term", "render_fwd_merged"};
}  // namespace

bool prof_enabled() { return g_on; }

void prof_begin(int slot, hipStream_t s, ProfScope* sc) {
  sc->idx = -1;
  if (!g_on || (g_only >= 0 && slot != g_only)) return;
  std::lock_guard<std::mutex> lk(g_mu);
  Pair p;
  if (!g_free.empty()) {
    p = g_free.back();
    g_free.pop_back();
  } else {
    if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
  }
  p.slot = slot;
  (void) hipEventRecord(p.a, s);
  g_pairs.push_back(p);
  sc->idx = (int) g_pairs.size() - 1;
}

void prof_end(hipStream_t s, ProfScope* sc) {
  if (sc->idx < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (sc->idx < (int) g_pairs.size()) (void) hipEventRecord(g_pairs[sc->idx].b, s);
}


// ---------------------------------------------------------------------------
// Exclusive prefix sum of per-owner counts (bin sizes -> list offsets) in one 1024-thread
// workgroup; also zeroes the fill cursors and publishes the total.  n is a few thousand to a
// few hundred thousand: one pass per 1024 elements, wave scans + LDS.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
exclusive_scan_kernel(const int* __restrict__ cnt, int* __restrict__ off, int* __restrict__ fill,
                      int n, int* __restrict__ total) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const int v = i < n ? cnt[i] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int wbase = 0;
    for (int k = 0; k < wv; ++k) wbase += wsum[k];
    const int c0 = carry;
    if (i < n) {
      off[i] = c0 + wbase + incl - v;
      fill[i] = 0;
    }
    __syncthreads();
    if (tid == 1023) carry = c0 + wbase + incl;
    __syncthreads();
  }
  // total[1], total[2] are list counters of the cell-list users (aux[ntile + 1], aux[ntile + 2]): the lists are built
  // right after this scan, so they are zeroed here instead of by a launch of their own
  if (tid == 0) { total[0] = carry; total[1] = 0; total[2] = 0; }
}

// The cell-list scan in ONE launch.  Level 1: every workgroup scans its kScanTile-cell tile (256
// threads x 8 cells) exclusively and publishes the tile total.  Level 2: the workgroup that takes the
// last arrival ticket scans the tile totals (a few hundred values) -- consumers add off[c] +
// boff[c / kScanTile].  The hand-off goes through agent-scope atomics on both sides (see below).
// (Two launches before round 3: the second one, a single workgroup, cost a dependent kernel boundary
// four times per step.)
__device__ __forceinline__ void cell_scan_tile(int* __restrict__ cnt, int* __restrict__ off, int* __restrict__ bsum,
                                               int* __restrict__ boff, int* __restrict__ fill, int* __restrict__ total,
                                               int* __restrict__ ticket, int ntile, unsigned bid) {
  __shared__ int wsum[4];
  __shared__ int is_last;
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long base = (long) bid * kScanTile + tid * 8;
  int4* c4 = reinterpret_cast<int4*>(cnt + base);
  const int4 a = c4[0], b = c4[1];
  // the counters are spent: left at zero they are the fill cursors of the list's user (the lift
  // backward hands out record slots with them), without a zero fill of their own
  c4[0] = make_int4(0, 0, 0, 0);
  c4[1] = make_int4(0, 0, 0, 0);
  const int v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  int tsum = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) tsum += v[k];
  int incl = tsum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int wbase = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < wv) wbase += wsum[k];
  int run = wbase + incl - tsum;
  int o8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    o8[k] = run;
    run += v[k];
  }
  int4* o4 = reinterpret_cast<int4*>(off + base);
  o4[0] = make_int4(o8[0], o8[1], o8[2], o8[3]);
  o4[1] = make_int4(o8[4], o8[5], o8[6], o8[7]);
  if (tid == 255) {
    // Publish the tile total, then take a ticket.  Both, and the last workgroup's reads of the totals,
    // are agent-scope read-modify-write atomics: they execute at the memory side, past every cache, so
    // the hand-off needs no release / acquire fence (a release fence here writes back the 16 KB of
    // offsets the workgroup has just dirtied: measured slower than the second launch it was to
    // replace).  The exchange has returned before the ticket is taken.
    const int old = __hip_atomic_exchange(bsum + bid, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(old) : "memory");
    const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == ntile - 1) ? 1 : 0;
  }
  if (tid == 0) carry = 0;
  __syncthreads();
  if (!is_last) return;
  // ---- level 2, last workgroup only: exclusive scan of the ntile totals
  for (int b0 = 0; b0 < ntile; b0 += 256) {
    const int i = b0 + tid;
    const int vv = i < ntile ? __hip_atomic_fetch_add(bsum + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    int inc2 = vv;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc2, o, 64);
      if (lane >= o) inc2 += up;
    }
    __syncthreads();                                  // wsum of the previous round is consumed
    if (lane == 63) wsum[wv] = inc2;
    __syncthreads();
    int wb = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < wv) wb += wsum[k];
    const int c0 = carry;
    if (i < ntile) {
      boff[i] = c0 + wb + inc2 - vv;
      fill[i] = 0;
    }
    __syncthreads();
    if (tid == 255) carry = c0 + wb + inc2;
    __syncthreads();
  }
  // total[1], total[2] are list counters of the cell-list users (aux[ntile + 1], aux[ntile + 2]): the lists are built
  // right after this scan, so they are zeroed here instead of by a launch of their own
  if (tid == 0) {
    total[0] = carry;
    total[1] = 0;
    total[2] = 0;
    *ticket = 0;                                      // ready for the next scan without a zero fill
  }
}

// ScanDuty (common.hpp): with early ray termination a few ray tiles hold a ray that never saturates and march 85
// samples where the others march 8: started last they were the per-ray pass's tail (23 tiles of 1 056, 42 us each,
// in a kernel of 56 us).
// The duty has a workgroup of its own, behind the scan's (in front of workgroup 0's tile it made that workgroup the last to
// publish its total: the scan's second level, and with it the launch, waited for the sort).
__device__ __forceinline__ void scan_duty(const ScanDuty& d) {
  __shared__ int cls[34];
  if (threadIdx.x < 34) cls[threadIdx.x] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < d.n; t += 256) atomicAdd(cls + (32 - __clz(max(d.key[t], 0))), 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int k = 32; k >= 0; --k) { const int n = cls[k]; cls[k] = acc; acc += n; }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < d.n; t += 256) d.order[atomicAdd(cls + (32 - __clz(max(d.key[t], 0))), 1)] = t;
  __syncthreads();
}

__global__ void __launch_bounds__(256)
cell_scan_kernel(int* __restrict__ cnt, int* __restrict__ off, int* __restrict__ bsum, int* __restrict__ boff,
                 int* __restrict__ fill, int* __restrict__ total, int* __restrict__ ticket, int ntile, ScanDuty duty) {
  if (blockIdx.x == (unsigned) ntile) { scan_duty(duty); return; }
  cell_scan_tile(cnt, off, bsum, boff, fill, total, ticket, ntile, blockIdx.x);
}

// Two independent cell lists scanned by ONE launch (workgroups [0, a.ntile) the first, the rest the second; each has
// its own ticket word and level 2): the lift's pair cells and the camera backward's sample cells of a training
// forward are both due between the render forward and the backward, and a launch of a few hundred small workgroups
// costs what its slowest workgroup costs -- 5 + 10 us as two launches, 10 as one.
__global__ void __launch_bounds__(256)
cell_scan_pair_kernel(ScanJob a, ScanJob b, ScanDuty duty) {
  if (blockIdx.x == (unsigned) (a.ntile + b.ntile)) { scan_duty(duty); return; }
  if (blockIdx.x < (unsigned) a.ntile)
    cell_scan_tile(a.cnt, a.off, a.bsum, a.boff, a.fill, a.total, a.ticket, a.ntile, blockIdx.x);
  else
    cell_scan_tile(b.cnt, b.off, b.bsum, b.boff, b.fill, b.total, b.ticket, b.ntile, blockIdx.x - (unsigned) a.ntile);
}

int launch_cell_scan(int* cnt, int* off, int* bsum, int* boff, int* aux, long ncell,
                     hipStream_t s, const ScanDuty* duty) {
  const long ntile = ncell / kScanTile;
  if (ncell % kScanTile != 0 || ntile > 0x7fffffffL)
    return fail(VAMP_EINVAL, "%s: cell count must be a multiple of the scan tile", __func__);
  VAMP_TIMED(kProfAux, s, (cell_scan_kernel<<<(unsigned) ntile + (duty && duty->key ? 1u : 0u), 256, 0, s>>>(
      cnt, off, bsum, boff, aux, aux + ntile, cnt + ncell, (int) ntile, duty ? *duty : ScanDuty{nullptr, nullptr, 0})));
  return check_launch("cell_scan_kernel");
}

int make_scan_job(int* cnt, int* off, int* bsum, int* boff, int* aux, long ncell, ScanJob* job) {
  const long ntile = ncell / kScanTile;
  if (ncell % kScanTile != 0 || ntile > 0x3fffffffL)
    return fail(VAMP_EINVAL, "%s: cell count must be a multiple of the scan tile", __func__);
  *job = ScanJob{cnt, off, bsum, boff, aux, aux + ntile, cnt + ncell, (int) ntile};
  return VAMP_OK;
}

int launch_cell_scan_pair(const ScanJob& a, const ScanJob& b, hipStream_t s, const ScanDuty* duty) {
  VAMP_TIMED(kProfAux, s, (cell_scan_pair_kernel<<<(unsigned) (a.ntile + b.ntile) + (duty && duty->key ? 1u : 0u), 256, 0, s>>>(
      a, b, duty ? *duty : ScanDuty{nullptr, nullptr, 0})));
  return check_launch("cell_scan_pair_kernel");
}

// grad_beta += sign(beta_raw) * sum(part[0..n)): the per-workgroup partial sums of d loss / d beta_eff
// are added up in a fixed order by one workgroup (no float atomics on one address, same bits every run)
__global__ void __launch_bounds__(1024)
beta_reduce_kernel(const float* __restrict__ part, int n, const float* __restrict__ beta_raw,
                   float* __restrict__ grad_beta) {
  __shared__ float red[16];
  float v = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) v += part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < 16; ++i) tot += red[i];
    const float sgn = (beta_raw[0] > 0.f) ? 1.f : ((beta_raw[0] < 0.f) ? -1.f : 0.f);
    // one atomic per launch: the camera and BEV branches add to the same word from two streams
    // (two addends commute, so the sum has the same bits every run)
    atomicAdd(grad_beta, sgn * tot);
  }
}

int launch_beta_reduce(const float* part, int n, const float* beta_raw, float* grad_beta, hipStream_t s) {
  beta_reduce_kernel<<<1, 1024, 0, s>>>(part, n, beta_raw, grad_beta);
  return check_launch("beta_reduce_kernel");
}

// Zero fill as an ordinary kernel.  hipMemsetAsync is avoided on purpose: a captured step whose
// single-stream graph held memset nodes faulted on replay once any eager kernel had run on the
// null stream in between (ROCm 7.2, tools/debug/graph_abort.py); kernel nodes do not.
__global__ void __launch_bounds__(256)
zero_fill_kernel(uint4* __restrict__ p16, size_t n16, uint32_t* __restrict__ tail, int ntail) {
  const size_t stride = (size_t) gridDim.x * 256;
  for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
    p16[i] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0 && (int) threadIdx.x < ntail) tail[threadIdx.x] = 0u;
}

// ---------------------------------------------------------------------------
// vamp_debug_checks(1): the promises a caller makes with the *_CLEAN / *_VALID flags are verified before
// they are relied on -- synchronously (the stream is drained per check: a debugging mode, never on by default).
// ---------------------------------------------------------------------------
bool g_debug_checks = false;

__global__ void __launch_bounds__(256)
debug_range_kernel(const int* __restrict__ p, size_t n, int lo, int hi, unsigned long long* __restrict__ bad) {
  size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  unsigned long long c = 0;
  for (; i < n; i += (size_t) gridDim.x * 256) c += (p[i] < lo || p[i] > hi) ? 1ull : 0ull;
  if (c) atomicAdd(bad, c);
}

int debug_expect_range(const int* p, size_t n, int lo, int hi, hipStream_t s, const char* what) {
  if (!g_debug_checks || !p || n == 0) return VAMP_OK;
  unsigned long long* bad = nullptr;
  if (hipHostMalloc(reinterpret_cast<void**>(&bad), sizeof(*bad), 0) != hipSuccess) return fail(VAMP_EHIP, "debug check: hipHostMalloc failed");
  *bad = 0;
  const unsigned grid = (unsigned) std::min<size_t>((n + 255) / 256, 4096);
  debug_range_kernel<<<grid, 256, 0, s>>>(p, n, lo, hi, bad);
  const hipError_t e = hipStreamSynchronize(s);
  const unsigned long long nbad = *bad;
  (void) hipHostFree(bad);
  if (e != hipSuccess) return fail(VAMP_EHIP, "debug check of (%s) failed to run", what);
  if (nbad) return fail(VAMP_EINVAL, "promise broken: %s -- %ld of %ld entries are not", what, (long) nbad, (long) n);
  return VAMP_OK;
}

int launch_zero(void* ptr, size_t bytes, hipStream_t s) {
  if (bytes == 0) return VAMP_OK;
  const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
  if ((a & 3u) != 0 || (bytes & 3u) != 0) return fail(VAMP_EINVAL, "%s: buffer must be 4-byte granular", __func__);
  // head words up to the first 16-byte boundary go with the tail words
  char* p = static_cast<char*>(ptr);
  const size_t head = std::min(bytes, (size_t) ((16 - (a & 15u)) & 15u));
  if (head) {
    zero_fill_kernel<<<1, 256, 0, s>>>(nullptr, 0, reinterpret_cast<uint32_t*>(p), (int) (head / 4));
    if (int e = check_launch("zero_fill_kernel")) return e;
    p += head;
    bytes -= head;
    if (bytes == 0) return VAMP_OK;
  }
  const size_t n16 = bytes / 16;
  const int ntail = (int) ((bytes - n16 * 16) / 4);
  const unsigned grid = (unsigned) std::min<size_t>(std::max<size_t>(1, (n16 + 1023) / 1024), 256 * 16);
  VAMP_TIMED(kProfMemset, s, (zero_fill_kernel<<<grid, 256, 0, s>>>(
      reinterpret_cast<uint4*>(p), n16, reinterpret_cast<uint32_t*>(p + n16 * 16), ntail)));
  return check_launch("zero_fill_kernel");
}

int launch_exclusive_scan(const int* cnt, int* off, int* fill, int n, int* total, hipStream_t s) {
  VAMP_TIMED(kProfAux, s, (exclusive_scan_kernel<<<1, 1024, 0, s>>>(cnt, off, fill, n, total)));
  return check_launch("exclusive_scan_kernel");
}

}  // namespace vamp

using namespace vamp;

extern "C" {

int vamp_debug_checks(int on) {
  vamp::g_debug_checks = on != 0;
  return VAMP_OK;
}

int vamp_abi_version(void) { return VAMP_ABI_VERSION; }
const char* vamp_last_error(void) { return g_err; }

int vamp_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (on) {
    for (auto& p : g_pairs) g_free.push_back(p);
    g_pairs.clear();
  }
  g_on = on != 0;
  return VAMP_OK;
}

int vamp_profile_slots(void) { return kProfSlots; }

int vamp_profile_select(int slot) {
  if (slot >= kProfSlots) return fail(VAMP_EINVAL, "%s: bad slot", __func__);
  std::lock_guard<std::mutex> lk(g_mu);
  g_only = slot < 0 ? -1 : slot;
  return VAMP_OK;
}

int vamp_profile_read(int slot, const char** name, int* launches, double* total_ms) {
  if (slot < 0 || slot >= kProfSlots || !name || !launches || !total_ms)
    return fail(VAMP_EINVAL, "%s: bad argument", __func__);
  std::lock_guard<std::mutex> lk(g_mu);
  *name = g_names[slot];
  *launches = 0;
  *total_ms = 0.0;
  for (auto& p : g_pairs) {
    if (p.slot != slot) continue;
    if (hipEventSynchronize(p.b) != hipSuccess) return fail(VAMP_EHIP, "%s: hipEventSynchronize", __func__);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) != hipSuccess) return fail(VAMP_EHIP, "%s: hipEventElapsedTime", __func__);
    *launches += 1;
    *total_ms += ms;
  }
  return VAMP_OK;
}

}  // extern "C"
