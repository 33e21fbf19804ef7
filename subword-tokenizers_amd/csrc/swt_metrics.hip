// swt_metrics.hip -- token-id histogram on the device: what the reference's benchmark suite counts with a Python Counter
// over token STRINGS (/root/reference/source/benchmarks.py:240-253, zipf_distribution) counted over token IDS, which are
// in bijection with the strings (a token with the '##' prefix and the same token without it are different strings and
// different ids: SWT_BPE_CONT).  SURVEY.md section 8f-3: so that the quality metrics run on 10 M-sentence outputs that never
// leave the device as strings.
//
// Token frequencies are Zipfian, so a global atomicAdd per token would serialise on the few hot ids.  Every wave stages its
// counters in LDS (a 1,024-slot table per wave, linear probing, the same pattern as the trainer's hist_build_kernel) and
// only the distinct ids of a wave's stretch reach the global array, one atomicAdd each.
#include "swt_common.h"

namespace swt {

constexpr int kMhThreads = 256;
constexpr int kMhSlots = 1024;
constexpr int kMhPerLane = 64;  // ids per lane: a wave flushes after 4,096 ids

__global__ __launch_bounds__(kMhThreads) void token_hist_kernel(const uint32_t *__restrict__ ids, uint64_t n, uint32_t id_cap,
                                                                unsigned long long *__restrict__ counts, unsigned long long *__restrict__ oor) {
  __shared__ uint32_t lk[kMhThreads / 64][kMhSlots];
  __shared__ uint32_t lc[kMhThreads / 64][kMhSlots];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t wave_id = (uint64_t)blockIdx.x * (kMhThreads / 64) + wave, n_waves = (uint64_t)gridDim.x * (kMhThreads / 64);
  for (uint64_t base = wave_id * 64 * kMhPerLane; base < n; base += n_waves * 64 * kMhPerLane) {
    for (int i = lane; i < kMhSlots; i += 64) { lk[wave][i] = 0xFFFFFFFFu; lc[wave][i] = 0; }
    __builtin_amdgcn_wave_barrier();
    for (int u = 0; u < kMhPerLane; u++) {
      const uint64_t i = base + (uint64_t)u * 64 + lane;  // coalesced: the wave reads 256 contiguous bytes per step
      if (i >= n) break;
      const uint32_t id = ids[i];
      const uint32_t sym = id & 0x7FFFFFFFu;
      if (sym >= id_cap) { atomicAdd(oor, 1ull); continue; }
      const uint32_t idx = sym + ((id >> 31) ? id_cap : 0u);
      uint32_t h = (idx * 2654435761u) >> 22;  // 10 bits
      bool done = false;
      for (int probe = 0; probe < 8 && !done; probe++) {
        uint32_t k = lk[wave][h];
        if (k == 0xFFFFFFFFu) {
          k = atomicCAS(&lk[wave][h], 0xFFFFFFFFu, idx);
          if (k == 0xFFFFFFFFu) k = idx;
        }
        if (k == idx) { atomicAdd(&lc[wave][h], 1u); done = true; }
        h = (h + 1) & (kMhSlots - 1);
      }
      if (!done) atomicAdd(&counts[idx], 1ull);  // the wave's table is crowded: straight to global
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < kMhSlots; i += 64)
      if (lk[wave][i] != 0xFFFFFFFFu) atomicAdd(&counts[lk[wave][i]], (unsigned long long)lc[wave][i]);
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace swt

using namespace swt;

extern "C" {

int swt_token_histogram_dev(const uint32_t *d_ids, uint64_t n, uint32_t id_cap, uint64_t *d_counts, uint64_t *d_out_of_range, void *stream) try {
  if ((n && !d_ids) || !d_counts || !d_out_of_range || !id_cap) return fail(SWT_ERR_INVALID, "null argument");
  int rc = ensure_device();
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  SWT_HIP(hipMemsetAsync(d_counts, 0, (size_t)2 * id_cap * 8, st));
  SWT_HIP(hipMemsetAsync(d_out_of_range, 0, 8, st));
  if (!n) return SWT_OK;
  uint64_t blocks = (n + (uint64_t)kMhThreads * kMhPerLane - 1) / ((uint64_t)kMhThreads * kMhPerLane);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(token_hist_kernel, dim3((unsigned)blocks), dim3(kMhThreads), 0, st, d_ids, n, id_cap,
                     reinterpret_cast<unsigned long long *>(d_counts), reinterpret_cast<unsigned long long *>(d_out_of_range));
  SWT_HIP(hipGetLastError());
  return SWT_OK;
} SWT_API_CATCH

int swt_token_histogram(const uint32_t *ids, uint64_t n, uint32_t id_cap, uint64_t *counts, uint64_t *out_of_range) try {
  if ((n && !ids) || !counts || !out_of_range || !id_cap) return fail(SWT_ERR_INVALID, "null argument");
  int rc = ensure_device();
  if (rc) return rc;
  DevBuf d_ids, d_counts;
  struct Guard { DevBuf &a, &b; ~Guard() { a.release(); b.release(); } } guard{d_ids, d_counts};
  if ((rc = d_ids.reserve(n * 4 + 16)) || (rc = d_counts.reserve((size_t)2 * id_cap * 8 + 16))) return rc;
  if (n) SWT_HIP(hipMemcpy(d_ids.p, ids, n * 4, hipMemcpyHostToDevice));
  uint64_t *d_c = d_counts.as<uint64_t>();
  if ((rc = swt_token_histogram_dev(d_ids.as<uint32_t>(), n, id_cap, d_c + 1, d_c, nullptr))) return rc;
  SWT_HIP(hipStreamSynchronize(nullptr));
  SWT_HIP(hipMemcpy(out_of_range, d_c, 8, hipMemcpyDeviceToHost));
  SWT_HIP(hipMemcpy(counts, d_c + 1, (size_t)2 * id_cap * 8, hipMemcpyDeviceToHost));
  return SWT_OK;
} SWT_API_CATCH

}  // extern "C"
