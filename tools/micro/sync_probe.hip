// What one launch + wait costs from the host (diagnostics): hipStreamSynchronize against spinning on a word the kernel
// writes into pinned host memory behind a system-scope fence.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
__global__ void k_plain(uint32_t *out) { if (threadIdx.x == 0) out[0] = 1; }
__global__ void k_flag(uint32_t *payload, volatile uint64_t *flag, uint64_t v) {
  payload[threadIdx.x] = (uint32_t)v + threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence_system(); *flag = v; }
}
int main() {
  uint32_t *d; hipMalloc(&d, 256);
  uint8_t *h; hipHostMalloc((void **)&h, 4096, hipHostMallocDefault);
  volatile uint64_t *flag = (volatile uint64_t *)h; uint32_t *payload = (uint32_t *)(h + 64);
  const int n = 5000;
  for (int rep = 0; rep < 2; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) { hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, 0, d); hipStreamSynchronize(0); }
    auto t1 = std::chrono::steady_clock::now();
    int bad = 0;
    for (int i = 0; i < n; i++) {
      const uint64_t v = (uint64_t)rep * n + i + 1;
      hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, 0, payload, flag, v);
      while (*flag != v) { }
      if (payload[63] != (uint32_t)v + 63) bad++;
    }
    auto t2 = std::chrono::steady_clock::now();
    hipStreamSynchronize(0);
    printf("launch + hipStreamSynchronize %.1f us, launch + spin on host word %.1f us (stale payloads: %d)\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / n, std::chrono::duration<double, std::micro>(t2 - t1).count() / n, bad);
  }
  return 0;
}
