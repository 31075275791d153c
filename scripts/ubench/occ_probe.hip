// occupancy vs dynamic LDS size for a 256-thread / 320-thread workgroup (hipOccupancyMaxActiveBlocksPerMultiprocessor)
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ unsigned smem[];
__global__ void __launch_bounds__(256) k256(unsigned* out) { smem[threadIdx.x] = threadIdx.x; __syncthreads(); out[threadIdx.x] = smem[255 - threadIdx.x]; }
__global__ void __launch_bounds__(320) k320(unsigned* out) { smem[threadIdx.x] = threadIdx.x; __syncthreads(); out[threadIdx.x] = smem[319 - threadIdx.x]; }
__global__ void __launch_bounds__(640) k640(unsigned* out) { smem[threadIdx.x] = threadIdx.x; __syncthreads(); out[threadIdx.x] = smem[639 - threadIdx.x]; }
int main() {
    hipFuncSetAttribute((const void*)k256, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k320, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k640, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int prev = -1;
    for (int b = 20000; b <= 84000; b += 64) {
        int n = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k256, 256, b);
        if (n != prev) { printf("k256 lds %d -> %d blocks\n", b, n); prev = n; }
    }
    prev = -1;
    for (int b = 20000; b <= 84000; b += 64) {
        int n = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k320, 320, b);
        if (n != prev) { printf("k320 lds %d -> %d blocks\n", b, n); prev = n; }
    }
    prev = -1;
    for (int b = 20000; b <= 164000; b += 64) {
        int n = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k640, 640, b);
        if (n != prev) { printf("k640 lds %d -> %d blocks\n", b, n); prev = n; }
    }
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu perMultiprocessor %zu maxThreadsPerMP %d\n", p.sharedMemPerBlock, p.sharedMemPerMultiprocessor, p.maxThreadsPerMultiProcessor);
    return 0;
}
