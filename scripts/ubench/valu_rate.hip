// micro-benchmark: issue rate of the integer VALU ops the scan kernel is made of (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 7 + i;
    uint32_t m = seed | 0xFF;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_alignbit(a[i], a[(i + 1) & 7], 6);
            if (OP == 1) a[i] = (a[i] & m) + it;
            if (OP == 2) a[i] = __builtin_popcount(a[i]) + a[(i + 1) & 7];          // v_bcnt accumulates
            if (OP == 3) a[i] = (a[i] << 2) + m;                                     // v_lshl_add
            if (OP == 4) a[i] = a[i] | a[(i + 3) & 7] | m;                            // v_or3
            if (OP == 5) a[i] = __builtin_amdgcn_udot4(a[i], m, a[(i + 1) & 7], false);
            if (OP == 6) a[i] = __builtin_amdgcn_perm(a[i], m, a[(i + 1) & 7]);
            if (OP == 7) a[i] = a[i] * m;                                            // v_mul_lo_u32
            if (OP == 8) a[i] = __umulhi(a[i], a[i]) + 1u;                            // v_mul_hi_u32 (+ v_add)
            if (OP == 10) a[i] = (a[i] << (a[(i + 1) & 7] & 31)) + 1u;                 // v_lshlrev + v_add
            if (OP == 11) a[i] = __builtin_amdgcn_sad_u8(a[i], m, a[(i + 1) & 7]);      // v_sad_u8
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, uint32_t* d, int waves_per_simd) {
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd;      // 256 CUs x (waves_per_simd WGs of 4 waves) -> waves_per_simd per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 100, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, iters, 3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * iters * 8;               // wave-instructions
    double per_simd_per_cycle = insts / 1024.0 / (ms * 1e-3 * 2.4e9);
    printf("%-14s waves/SIMD=%d  %.3f ms  %.3f wave-instr/cycle/SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms, per_simd_per_cycle);
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_alignbit", d, w); run<1>("v_and+v_add", d, w); run<2>("v_bcnt", d, w); run<3>("v_lshl_add", d, w);
        run<4>("v_or3", d, w); run<5>("v_dot4_u32_u8", d, w); run<6>("v_perm", d, w); run<7>("v_mul_lo_u32", d, w);
        run<8>("v_mul_hi+add", d, w); run<10>("v_lshl+add", d, w); run<11>("v_sad_u8", d, w);
    }
    return 0;
}
