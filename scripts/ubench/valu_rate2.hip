// micro-benchmark 2: issue rate of the "other" instructions the scan kernel uses (gfx950):
// 64-bit integer ops, multiplies, DPP, lane moves, f64, and LDS gathers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t tab[1024 + 64 * 9];
    for (int i = threadIdx.x; i < 1024 + 64 * 9; i += 256) tab[i] = (i * 2654435761u) >> 7;
    __syncthreads();
    uint32_t a[8];
    uint64_t q[8];
    double f[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 7 + i; q[i] = a[i] * 0x100000001ull; f[i] = 1.0 + a[i] * 1e-9; }
    uint32_t m = seed | 0xFF;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) q[i] += 1ull << (a[i] & 63);                                   // v_lshlrev_b64 + 64-bit add
            if (OP == 1) q[i] = (q[i] << 2) + q[(i + 1) & 7];                            // v_lshl_add_u64
            if (OP == 2) q[i] = (uint64_t)a[i] * m + q[i];                               // v_mad_u64_u32
            if (OP == 3) a[i] = __umulhi(a[i], m) + it;                                  // v_mul_hi_u32
            if (OP == 4) a[i] = __umul24(a[i], m) + it;                                  // v_mul_u32_u24 / mad
            if (OP == 5) a[i] += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a[i], 0x111, 0xf, 0xf, false);   // v_add_dpp row_shr:1
            if (OP == 6) a[i] = (uint32_t)__builtin_amdgcn_readlane((int)a[i], (i * 5) & 63) + a[i];
            if (OP == 7) a[i] = (uint32_t)__builtin_clz(a[i] | 1u) + a[(i + 1) & 7];     // v_ffbh
            if (OP == 8) a[i] = a[i] > m ? a[(i + 1) & 7] : a[i] + 1;                    // v_cmp + v_cndmask
            if (OP == 9) f[i] = f[i] * 1.0000001 + 1e-9;                                 // v_fma_f64
            if (OP == 10) f[i] = (double)a[i] + f[i];                                    // v_cvt_f64_u32 + add
            if (OP == 11) f[i] = 1.0 / f[i] + 1.0;                                       // f64 divide
            if (OP == 12) a[i] = tab[(a[i] + it) & 1023];                                // random LDS gather (dependent)
            if (OP == 13) a[i] = tab[1024 + (threadIdx.x & 63) * 9 + i] + a[i];          // conflict-free LDS read
            if (OP == 14) a[i] = ((const uint8_t*)tab)[(threadIdx.x & 63) * 16 + i + (it & 3) * 1024] + a[i];   // ds_read_u8
            if (OP == 15) a[i] = __builtin_amdgcn_perm(a[i], a[(i + 1) & 7], 0x07060100u) ^ (a[i] >> 31 << i);
            if (OP == 16) { float x = __builtin_amdgcn_rcpf((float)a[i]); a[i] += (uint32_t)(x * 1e6f); }   // cvt, rcp, mul, cvt, add
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32) ^ (uint32_t)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, uint32_t* d, int waves_per_simd) {
    const int iters = 4000;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 100, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, iters, 3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double groups = (double)blocks * 4 * iters * 8;               // statement executions per wave
    double cyc = (ms * 1e-3 * 2.4e9) * 1024.0 / groups;           // SIMD cycles per statement per wave
    printf("%-28s waves/SIMD=%d  %.3f ms  %.1f SIMD-cycles per statement\n", name, waves_per_simd, ms, cyc);
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {4}) {
        run<0>("shl64(1,x)+add64", d, w); run<1>("lshl_add_u64", d, w); run<2>("mad_u64_u32", d, w); run<3>("mul_hi_u32+add", d, w);
        run<4>("mul_u24+add", d, w); run<5>("add_dpp row_shr", d, w); run<6>("readlane+add", d, w); run<7>("ffbh(or)+add", d, w);
        run<8>("cmp+cndmask+add", d, w); run<9>("fma_f64", d, w); run<10>("cvt_f64_u32+add_f64", d, w); run<11>("f64 divide+add", d, w);
        run<12>("LDS random gather", d, w); run<13>("LDS stride-9 read + add", d, w); run<14>("ds_read_u8 + add", d, w); run<15>("perm+shifts+xor", d, w);
        run<16>("cvt,rcp_f32,mul,cvt,add", d, w);
    }
    return 0;
}
