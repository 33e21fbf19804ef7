// clock_probe: shader clock seen by short dependent-chain kernels launched back to back (DVFS under launch-bound load), and
// the latency of a dependent global-load chain.  Build: hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned long long *out, const unsigned int *chain, int hops, int idx) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned int p = idx & 1023;
  for (int i = 0; i < hops; i++) p = chain[p];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[3 * idx] = t1 - t0;
  out[3 * idx + 1] = r1 - r0;
  out[3 * idx + 2] = p;
}
int main() {
  const int N = 4000, M = 1 << 22;
  std::vector<unsigned int> h(M);
  for (int i = 0; i < M; i++) h[i] = (unsigned int)(((unsigned long long)i * 2654435761ull + 12345) % M);
  unsigned int *d_chain; unsigned long long *d_out;
  hipMalloc(&d_chain, M * 4); hipMalloc(&d_out, N * 24);
  hipMemcpy(d_chain, h.data(), M * 4, hipMemcpyHostToDevice);
  for (int hops : {16, 64}) {
    for (int grid : {1, 128}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      for (int i = 0; i < N; i++) hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, d_out, d_chain, hops, i);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> o(3 * N);
      hipMemcpy(o.data(), d_out, N * 24, hipMemcpyDeviceToHost);
      double st = 0, rt = 0;
      for (int i = N / 2; i < N; i++) { st += o[3 * i]; rt += o[3 * i + 1]; }
      printf("hops %d grid %d: %.2f us per launch, shader clock %.0f MHz, %.0f shader cycles = %.0f ns per dependent load\n", hops, grid,
             ms * 1e3 / N, st / rt * 100.0, st / (N / 2) / hops, rt / (N / 2) / hops * 10.0);
    }
  }
  return 0;
}
