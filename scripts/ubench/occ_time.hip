// measured workgroups-per-CU vs dynamic LDS size: every workgroup spins a fixed number of clocks, so the
// launch time is proportional to 1 / (concurrent workgroups per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ unsigned smem[];
__global__ void __launch_bounds__(256) spin(unsigned* out, long long clocks) {
    smem[threadIdx.x] = threadIdx.x;
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < clocks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && smem[1] == 12345u) out[0] = 1;
}
int main() {
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    unsigned* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = 256 * 60;
    for (int b = 26000; b <= 34000; b += 256) {
        spin<<<wgs, 256, b>>>(d, 2000); hipDeviceSynchronize();
        hipEventRecord(e0); spin<<<wgs, 256, b>>>(d, 2000); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("lds %d ms %.3f\n", b, ms);
    }
    return 0;
}
