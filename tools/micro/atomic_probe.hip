// atomic_probe: what same-address device atomics cost a short kernel.  H lanes spread evenly over a grid of 128 x 256 issue one
// atomic each at ONE address (or at H addresses); the kernel's on-device span (first start -> last end, s_memrealtime) is the cost.
// Build: hipcc --offload-arch=gfx950 -O3 -o atomic_probe atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ unsigned long long g_t0, g_t1;
template <int MODE>  // 0 atomicMin 64 same address (no return), 1 returning, 2 distinct addresses, 3 plain store same address, 4 atomicAdd same
__global__ void probe(unsigned long long *cell, unsigned int every, unsigned long long *sink) {
  if (threadIdx.x == 0) atomicMin(&g_t0, __builtin_amdgcn_s_memrealtime());
  const unsigned int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid % every == 0) {
    const unsigned long long v = 1000000ull - gid;
    if (MODE == 0) atomicMin(cell, v);
    if (MODE == 1) { const unsigned long long o = atomicMin(cell, v); if (o == 12345) sink[gid] = o; }
    if (MODE == 2) atomicMin(cell + 16 * (gid / every), v);
    if (MODE == 3) __hip_atomic_store(cell, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 4) atomicAdd(cell, v);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) atomicMax(&g_t1, __builtin_amdgcn_s_memrealtime());
}
template <int MODE>
void run(const char *name, unsigned long long *cell, unsigned long long *sink) {
  for (unsigned int H : {0u, 1u, 16u, 128u, 512u, 2048u, 32768u}) {
    const unsigned int total = 128 * 256, every = H ? total / H : total * 2 + 1;
    double sum = 0; int n = 0;
    for (int it = 0; it < 60; it++) {
      unsigned long long a = ~0ull, b = 0, big = ~0ull;
      hipMemcpyToSymbol(HIP_SYMBOL(g_t0), &a, 8); hipMemcpyToSymbol(HIP_SYMBOL(g_t1), &b, 8);
      hipMemcpy(cell, &big, 8, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(probe<MODE>, dim3(128), dim3(256), 0, 0, cell, every, sink);
      hipDeviceSynchronize();
      hipMemcpyFromSymbol(&a, HIP_SYMBOL(g_t0), 8); hipMemcpyFromSymbol(&b, HIP_SYMBOL(g_t1), 8);
      if (it >= 10) { sum += (double)(b - a) / 100.0; n++; }
    }
    printf("%-28s H=%6u lanes: span %.2f us\n", name, H, sum / n);
  }
}
int main() {
  unsigned long long *cell, *sink;
  hipMalloc(&cell, 16 * 8 * 40000); hipMalloc(&sink, 8 * 40000);
  hipMemset(cell, 0xFF, 16 * 8 * 40000);
  run<0>("atomicMin same address", cell, sink);
  run<1>("atomicMin same, returning", cell, sink);
  run<2>("atomicMin distinct lines", cell, sink);
  run<3>("store same address", cell, sink);
  run<4>("atomicAdd same address", cell, sink);
  return 0;
}
