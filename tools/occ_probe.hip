#include <hip/hip_runtime.h>
#include <cstdio>
template <int KB> __global__ __launch_bounds__(256) void k(int *o) { __shared__ int a[KB * 256]; a[threadIdx.x] = threadIdx.x; __syncthreads(); o[threadIdx.x] = a[(threadIdx.x * 7) % (KB * 256)]; }
template <int KB> void q() { int n = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<KB>, 256, 0); printf("LDS %3d KB -> %d blocks/CU\n", KB, n); }
int main() { hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
 q<8>(); q<16>(); q<20>(); q<32>(); q<40>(); q<51>(); q<64>(); return 0; }
