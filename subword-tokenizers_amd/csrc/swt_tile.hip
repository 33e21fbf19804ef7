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

// Exclusive scan of the tile totals (one workgroup; n_tiles is small next to the text).
__global__ __launch_bounds__(1024) void tile_scan_kernel(const uint32_t *__restrict__ tile_tok, uint64_t n_tiles,
                                                         uint64_t *__restrict__ tile_base, uint64_t *__restrict__ n_tokens) {
  __shared__ unsigned long long wsum[16];
  __shared__ unsigned long long carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (uint64_t base = 0; base < n_tiles; base += 1024) {
    const uint64_t i = base + tid;
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
    const unsigned long long carry = carry_s;
    if (i < n_tiles) tile_base[i] = carry + wb + x - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + wb + x;
    __syncthreads();
  }
  if (tid == 0) {
    tile_base[n_tiles] = carry_s;
    *n_tokens = carry_s;
  }
}

// A tile's tokens are contiguous in the output: copy its run and turn local sentence offsets into global ones.
__global__ __launch_bounds__(kThreads) void gather_kernel(const uint64_t *__restrict__ sent_off, const uint64_t *__restrict__ plan,
                                                          uint64_t n_tiles, uint64_t n_sent, const uint32_t *__restrict__ scratch,
                                                          const uint32_t *__restrict__ sent_local, const uint32_t *__restrict__ tile_tok,
                                                          const uint64_t *__restrict__ tile_base, uint32_t *__restrict__ out_ids,
                                                          uint64_t *__restrict__ out_off) {
  const uint64_t t = blockIdx.x;
  const uint64_t s_lo = plan[t], s_hi = plan[t + 1];
  if (t == n_tiles - 1 && threadIdx.x == 0) out_off[n_sent] = tile_base[n_tiles];
  if (s_lo == s_hi) return;
  const uint64_t base = tile_base[t];
  const uint32_t n = tile_tok[t];
  const uint32_t *src = scratch + sent_off[s_lo];
  for (uint32_t i = threadIdx.x; i < n; i += kThreads) out_ids[base + i] = src[i];
  for (uint64_t s = s_lo + threadIdx.x; s < s_hi; s += kThreads) out_off[s] = base + sent_local[s];
}

int TileWorkspace::reserve(uint64_t n_bytes, uint64_t n_sent, uint64_t n_tiles) {
  int rc;
  if ((rc = plan.reserve((n_tiles + 1) * 8))) return rc;
  if ((rc = scratch.reserve((n_bytes + 64) * 4))) return rc;
  if ((rc = sent_local.reserve((n_sent + 1) * 4))) return rc;
  if ((rc = tile_tok.reserve((n_tiles + 1) * 4))) return rc;
  if ((rc = tile_base.reserve((n_tiles + 2) * 8))) return rc;
  return SWT_OK;
}

void TileWorkspace::release() {
  plan.release(); scratch.release(); sent_local.release(); tile_tok.release(); tile_base.release();
}

void launch_plan(const uint64_t *d_sent_off, uint64_t n_sent, uint64_t n_tiles, uint32_t tile, uint64_t *d_plan, hipStream_t st) {
  hipLaunchKernelGGL(plan_kernel, dim3((unsigned)((n_tiles + 1 + 255) / 256)), dim3(256), 0, st, d_sent_off, n_sent, n_tiles, tile, d_plan);
}

void launch_scan_gather(const uint64_t *d_sent_off, uint64_t n_sent, uint64_t n_tiles, const TileWorkspace &ws,
                        uint32_t *d_out_ids, uint64_t *d_out_off, uint64_t *d_n_tokens, hipStream_t st) {
  hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, ws.tile_tok.as<uint32_t>(), n_tiles,
                     ws.tile_base.as<uint64_t>(), d_n_tokens);
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)n_tiles), dim3(kThreads), 0, st, d_sent_off, ws.plan.as<uint64_t>(), n_tiles,
                     n_sent, ws.scratch.as<uint32_t>(), ws.sent_local.as<uint32_t>(), ws.tile_tok.as<uint32_t>(),
                     ws.tile_base.as<uint64_t>(), d_out_ids, d_out_off);
}

}  // namespace swt
