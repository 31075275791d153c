// micro-benchmark: LDS gather / scatter issue rates per CU (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t tab[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + seed, acc = 0;
    const uint32_t lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        uint32_t a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            x = x * 1664525u + 1013904223u;
            if (MODE == 0) a[i] = tab[(x >> 20) & 255];                 // random gather in a 1 KB table (b32)
            if (MODE == 1) a[i] = tab[(lane + i * 64 + it) & 4095];     // conflict-free b32
            if (MODE == 2) a[i] = ((volatile uint16_t*)tab)[(lane + i * 80 + it) & 8191];   // conflict-free u16
            if (MODE == 3) a[i] = tab[((x >> 20) & 7) * 33];            // telomere-like: 8 hot entries
            if (MODE == 4) a[i] = tab[((x >> 24) & 255) + (((x >> 16) & 3) << 8)];   // 4 replicated 1 KB tables
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += a[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE> void run(const char* name, uint32_t* d, int wg_per_cu) {
    const int iters = 4000, blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 10, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters, 3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * iters * 8;
    printf("%-34s WG/CU=%d  %.3f ms  %.3f LDS wave-instr/cycle/CU (at 2.4 GHz)  [VALU per LDS op ~4]\n", name, wg_per_cu, ms, insts / 256.0 / (ms * 1e-3 * 2.4e9));
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {2, 4}) {
        run<0>("random gather 1KB b32", d, w); run<1>("conflict-free b32", d, w); run<2>("conflict-free u16", d, w);
        run<3>("8 hot entries", d, w); run<4>("random gather, 4 replicas", d, w);
    }
    return 0;
}
