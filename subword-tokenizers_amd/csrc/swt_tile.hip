// swt_tile.hip -- the skeleton's own kernels: plan (tile -> first sentence), scan of tile totals, gather.
#include "swt_tile.h"

namespace swt {

// plan[t] = first sentence whose first byte is >= t * tile  (lower bound; plan[n_tiles] = n_sent)
__global__ void plan_kernel(const uint64_t *__restrict__ sent_off, uint64_t n_sent, uint64_t n_tiles, uint32_t tile,
                            uint64_t *__restrict__ plan) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > n_tiles) return;
  if (t == n_tiles) { plan[t] = n_sent; return; }
  const uint64_t target = t * (uint64_t)tile;
  uint64_t lo = 0, hi = n_sent;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (sent_off[mid] < target) lo = mid + 1; else hi = mid;
  }
  plan[t] = lo;
}

// Exclusive scan of the tile totals, one launch: every workgroup scans its 1024 tiles locally and publishes its total;
// the last one to arrive (ticket) scans the workgroup totals.  Global base of tile t = blk_base[t >> 10] + tile_base[t].
template <class T>
__global__ __launch_bounds__(1024) void tile_scan_kernel(const T *__restrict__ tile_tok, uint64_t n_tiles,
                                                         T *__restrict__ tile_base, unsigned long long *__restrict__ blk_tot,
                                                         unsigned long long *__restrict__ blk_base, unsigned int *__restrict__ ticket,
                                                         uint64_t *__restrict__ n_tokens) {
  __shared__ unsigned long long wsum[16];
  __shared__ unsigned long long carry_s;
  __shared__ bool is_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t i = (uint64_t)blockIdx.x * 1024 + tid;
  const unsigned long long v = i < n_tiles ? tile_tok[i] : 0;
  unsigned long long x = v;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long y = __shfl_up(x, d);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  unsigned long long wb = 0;
  for (int w = 0; w < wave; w++) wb += wsum[w];
  if (i < n_tiles) tile_base[i] = (T)(wb + x - v);
  if (tid == 1023) {
    // the total travels in a device-scope store and is read back by device-scope loads below, so the ticket only has to wait until
    // the store has been performed -- no device-scope FENCE (which writes the XCD's L2 back and invalidates it, once per workgroup)
    __hip_atomic_store(&blk_tot[blockIdx.x], wb + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    is_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < gridDim.x; base += 1024) {
    const uint32_t j = base + tid;
    const unsigned long long u = j < gridDim.x ? __hip_atomic_load(&blk_tot[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    unsigned long long z = u;
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long y = __shfl_up(z, d);
      if (lane >= d) z += y;
    }
    if (lane == 63) wsum[wave] = z;
    __syncthreads();
    unsigned long long wb2 = 0;
    for (int w = 0; w < wave; w++) wb2 += wsum[w];
    const unsigned long long carry = carry_s;
    if (j < gridDim.x) blk_base[j] = carry + wb2 + z - u;
    __syncthreads();
    if (tid == 1023) carry_s = carry + wb2 + z;
    __syncthreads();
  }
  if (tid == 0) {
    *n_tokens = carry_s;
    *ticket = 0;  // ready for the next call
  }
}

// A tile's tokens are contiguous in the output: copy its run and turn local sentence offsets into global ones.  A tile holds
// a few dozen tokens: one WAVE per tile (a workgroup per tile spent most of the launch on dispatching 66 k workgroups).
__global__ __launch_bounds__(kThreads) void gather_kernel(const uint64_t *__restrict__ sent_off, const uint64_t *__restrict__ plan,
                                                          uint64_t n_tiles, uint64_t n_sent, const uint32_t *__restrict__ scratch,
                                                          const uint32_t *__restrict__ sent_local, const uint32_t *__restrict__ tile_tok,
                                                          const uint32_t *__restrict__ tile_base, const unsigned long long *__restrict__ blk_base,
                                                          const uint64_t *__restrict__ n_tokens, uint32_t *__restrict__ out_ids,
                                                          uint64_t *__restrict__ out_off) {
  const int lane = threadIdx.x & 63;
  const uint64_t t = (uint64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (t >= n_tiles) return;
  const uint64_t s_lo = plan[t], s_hi = plan[t + 1];
  if (t == n_tiles - 1 && lane == 0) out_off[n_sent] = *n_tokens;
  if (s_lo == s_hi) return;
  const uint64_t base = blk_base[t >> 10] + tile_base[t];
  const uint32_t n = tile_tok[t];
  const uint32_t *src = scratch + sent_off[s_lo];
  for (uint32_t i = lane; i < n; i += 64) out_ids[base + i] = src[i];
  for (uint64_t s = s_lo + lane; s < s_hi; s += 64) out_off[s] = base + sent_local[s];
}

int TileWorkspace::reserve(uint64_t n_bytes, uint64_t n_sent, uint64_t n_tiles) {
  int rc;
  if ((rc = plan.reserve((n_tiles + 1) * 8))) return rc;
  if ((rc = scratch.reserve((n_bytes + 64) * 4))) return rc;
  if ((rc = sent_local.reserve((n_sent + 1) * 4))) return rc;
  if ((rc = tile_tok.reserve((n_tiles + 1) * 4))) return rc;
  if ((rc = tile_base.reserve((n_tiles + 2) * 4))) return rc;
  const uint64_t nb = (n_tiles + 1023) / 1024;
  const void *before = blk.p;
  if ((rc = blk.reserve((2 * nb + 4) * 8))) return rc;
  if (blk.p != before) {  // a new buffer: zero the ticket once (the scan kernel resets it itself afterwards)
    SWT_HIP(hipMemset(blk.p, 0, 8));
    SWT_HIP(hipDeviceSynchronize());  // rare path; callers launch on streams that need not order with the null stream
  }
  return SWT_OK;
}

void TileWorkspace::release() {
  plan.release(); scratch.release(); sent_local.release(); tile_tok.release(); tile_base.release(); blk.release();
}

void launch_plan(const uint64_t *d_sent_off, uint64_t n_sent, uint64_t n_tiles, uint32_t tile, uint64_t *d_plan, hipStream_t st) {
  hipLaunchKernelGGL(plan_kernel, dim3((unsigned)((n_tiles + 1 + 255) / 256)), dim3(256), 0, st, d_sent_off, n_sent, n_tiles, tile, d_plan);
}

void launch_scan_only(uint64_t n_tiles, const TileWorkspace &ws, uint64_t *d_n_tokens, hipStream_t st) {
  const uint64_t nb = (n_tiles + 1023) / 1024;
  unsigned long long *b = ws.blk.as<unsigned long long>();
  hipLaunchKernelGGL(tile_scan_kernel<uint32_t>, dim3((unsigned)nb), dim3(1024), 0, st, ws.tile_tok.as<uint32_t>(), n_tiles,
                     ws.tile_base.as<uint32_t>(), b + 1, b + 1 + nb, reinterpret_cast<unsigned int *>(b), d_n_tokens);
}

void launch_scan_u64(uint64_t n, const unsigned long long *d_in, unsigned long long *d_local, unsigned long long *blk,
                     uint64_t *d_total, hipStream_t st) {
  const uint64_t nb = (n + 1023) / 1024;
  hipLaunchKernelGGL(tile_scan_kernel<unsigned long long>, dim3((unsigned)nb), dim3(1024), 0, st, d_in, n, d_local, blk + 1,
                     blk + 1 + nb, reinterpret_cast<unsigned int *>(blk), d_total);
}

void launch_scan_gather(const uint64_t *d_sent_off, uint64_t n_sent, uint64_t n_tiles, const TileWorkspace &ws,
                        uint32_t *d_out_ids, uint64_t *d_out_off, uint64_t *d_n_tokens, hipStream_t st) {
  const uint64_t nb = (n_tiles + 1023) / 1024;
  // blk layout: [0] ticket (8 bytes), then blk_tot[nb], then blk_base[nb]
  unsigned long long *b = ws.blk.as<unsigned long long>();
  unsigned int *ticket = reinterpret_cast<unsigned int *>(b);
  unsigned long long *blk_tot = b + 1, *blk_base = b + 1 + nb;
  hipLaunchKernelGGL(tile_scan_kernel<uint32_t>, dim3((unsigned)nb), dim3(1024), 0, st, ws.tile_tok.as<uint32_t>(), n_tiles,
                     ws.tile_base.as<uint32_t>(), blk_tot, blk_base, ticket, d_n_tokens);
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((n_tiles + kThreads / 64 - 1) / (kThreads / 64))), dim3(kThreads), 0, st, d_sent_off, ws.plan.as<uint64_t>(), n_tiles,
                     n_sent, ws.scratch.as<uint32_t>(), ws.sent_local.as<uint32_t>(), ws.tile_tok.as<uint32_t>(),
                     ws.tile_base.as<uint32_t>(), blk_base, d_n_tokens, d_out_ids, d_out_off);
}

}  // namespace swt
