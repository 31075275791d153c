// tps_kernels.h -- the scan kernels of libtopsicle_hip.so: the kernel body (one macro) and the list of instantiations, in
// GROUPS so that the library can be compiled as several translation units in parallel (the self-overlap kernels alone are
// two thirds of the machine code).  topsicle_hip.hip includes this with TPS_KGROUP undefined (all kernels in one unit: the
// scripts that dump resource usage or ISA do that) or = 0 (host code + the small kernels, the rest declared only);
// tps_kernels.hip is compiled once per TPS_KGROUP = 1 .. TPS_KGROUPS - 1 (__graft_entry__.build_hip).
#pragma once
#include <hip/hip_runtime.h>

#include "tps_device.h"

// One wave per read, tps::WPG waves per workgroup.  The lookup table is loaded once per workgroup
// (the only workgroup barrier in the kernel); after that every wave runs its own read with
// wave-level synchronisation only.
// diagnostics build: clock stamps at kernel entry (13: shader clock, 15: the 100 MHz device-wide real-time counter) and
// behind the table barrier (14), per read like the stamps inside scan_read
#ifdef TPS_STAMPS
#define TPS_KSTAMP(i) do { int64_t r_ = (int64_t)blockIdx.x * a.wpg + (int)(threadIdx.x >> 6);                                \
        if (a.order && r_ < a.n_reads) r_ = a.order[r_];                                                                      \
        if (a.stamps && (threadIdx.x & 63u) == 0 && r_ < a.n_reads) { a.stamps[r_ * 16 + (i)] = __builtin_readcyclecounter(); \
            if ((i) == 13) a.stamps[r_ * 16 + 15] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define TPS_KSTAMP(i) ((void)0)
#endif
#define TPS_SCAN_KERNEL_F(NAME, SV, SO, PAIR, RAW, MINW, FULL, DCLASS)                                                           \
    extern "C" __global__ void __launch_bounds__(tps::NT * tps::WPG_MAX, MINW) NAME(tps::ScanArgs a) {     \
        extern __shared__ __attribute__((aligned(16))) uint32_t lds[];                                     \
        TPS_KSTAMP(13);                                                                                    \
        /* workgroup-shared tables: [pair table (PAIR kernels)][single table], both aligned to their size */ \
        uint32_t* lut = lds + ((PAIR) ? a.pair_n : 0);                                                     \
        const int nthr_ = tps::NT * a.wpg;            /* = blockDim.x */                                   \
        if ((SV) != 0) {   /* fused kernels: the host keeps the table in its LDS format (lut_img): a copy in 16-byte pieces */ \
            const int ndw_ = (int)tps::lut_dw(a);                                                           \
            for (int c = 4 * (int)threadIdx.x; c < ndw_; c += 4 * nthr_)                                    \
                *(uint4*)(lut + c) = *(const uint4*)(a.lut_img + c);                                        \
        } else {                                                                                           \
            for (int i = (int)threadIdx.x; i < a.lut_n; i += nthr_) lut[i] = a.lut[i];                      \
        }                                                                                                  \
        if (PAIR) {   /* host-built pair table, stored right behind the plain table */                      \
            for (int c = 4 * (int)threadIdx.x; c < a.pair_n; c += 4 * nthr_)                               \
                *(uint4*)(lds + c) = *(const uint4*)(a.pair_img + c);                                 \
        }                                                                                                  \
        __syncthreads();                                                                                   \
        TPS_KSTAMP(14);                                                                                    \
        /* One read per wave; the hardware dispatcher balances the workgroups.  (Persistent waves were      \
           tried: a shared device counter sustains only ~50 M same-address atomics/s -- too slow for the  \
           claim rate -- and a static stride loses the dispatcher's dynamic balancing: 10 % slower on    \
           25k-read batches.)                                                                             \
           readfirstlane: the wave index is uniform -> everything per read lives in SGPRs */              \
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));                          \
        const int64_t wave_dw = (tps::lds_dwords(a) + 3) & ~3ll;                                           \
        uint32_t* slice = lut + tps::lut_dw(a) + wave * wave_dw;                                           \
        /* (several reads per wave -- a grid-strided loop here, so that a big table is loaded once per 8 or 16 reads -- was built \
           in round 4 and dropped: with a loop around it the optimiser hoists the read-invariant arithmetic of scan_read out of  \
           the loop and keeps it live: _s6so 88 -> 96 VGPRs + scratch, 146 -> 227 spilled SGPRs.  So was requesting the read's      \
           descriptor and its step-1 heads BEFORE the table load: config 2 57.1 against 57.0 us, k = 6 146.9 against 144.9 --    \
           with 20 - 24 waves per CU in flight a read's own latency is hidden already) */                                         \
        /* round 5: which read a wave slot takes is the host's choice (ScanArgs::order): a workgroup's LDS and wave slots come   \
           free when its LAST read ends, so reads of similar length share a workgroup and the longest go first -- a batch of     \
           log-normal read lengths (what an ONT file holds) takes a fifth less time; batches of equal reads keep file order */   \
        const int64_t slot = (int64_t)blockIdx.x * a.wpg + wave;                                           \
        if (slot < a.n_reads) {                                                                            \
            const int64_t r = a.order ? (int64_t)__builtin_amdgcn_readfirstlane(a.order[slot]) : slot;     \
            tps::scan_read<SV, SO, PAIR, RAW, FULL, DCLASS>(a, r, slice, lut);                             \
        }                                                                                                  \
    }
#ifndef TPS_R_MINW
#define TPS_R_MINW 5      // waves per SIMD the raw-row kernels of tables without self-overlap are compiled for: 96 VGPRs with 7 spilled and 32 B
#endif                    // of scratch instead of 106 and none -- k = 4 with raw rows 171.2 -> 165.8 us (143.4 against 149.6 per batch on two streams);
                          // the fifth wave is worth more than the spills cost, as in the self-overlap raw kernels.  Slide 8 would spill 64: it stays at 3
#ifndef TPS_S8SOR_MINW
#define TPS_S8SOR_MINW 4  // ... of the slide-8 self-overlap raw-row kernel: 113 VGPRs; compiled for 5 it spills 94 VGPRs to 68 B of scratch (k = 5 at slide 8: 168.6 -> 163.8 us)
#endif
#ifndef TPS_SOR_MINW
#define TPS_SOR_MINW 5
#endif
#ifndef TPS_SOL_MINW
#define TPS_SOL_MINW 6     // waves per SIMD the sums kernels of self-overlap periods 2 .. 4 (k = 5) are compiled for: 80 VGPRs without a spill, and their LDS (25 472 B per workgroup) allows the sixth: k = 5 sums 106.0 -> 102.0 us, 97 -> 90 per batch on two streams
#endif
#ifndef TPS_SO_MINW
#define TPS_SO_MINW 6     // waves per SIMD the sums-only self-overlap kernels are compiled for (round 5: 80 VGPRs, no VGPR spill; with the lane totals in the
                          // pad words a k = 6 wave slice is 5 536 B: 8 192 + 8 x 5 536 = 52 480 B -> three 8-wave workgroups = 24 waves per CU instead of 20)
#endif
#define TPS_SCAN_KERNEL(NAME, SV, SO, PAIR, RAW, MINW) TPS_SCAN_KERNEL_F(NAME, SV, SO, PAIR, RAW, MINW, tps::tile_full_default(SV), 0)
#define TPS_SCAN_KERNEL_D(NAME, SV, SO, PAIR, RAW, MINW, DCLASS) TPS_SCAN_KERNEL_F(NAME, SV, SO, PAIR, RAW, MINW, tps::tile_full_default(SV), DCLASS)

#define TPS_KGROUPS 18
#ifdef TPS_KGROUP
#define TPS_IN_GROUP(g) (TPS_KGROUP == (g))
#else
#define TPS_IN_GROUP(g) 1
#endif
#define TPS_SCAN_KERNEL_DECL(NAME) extern "C" __global__ void NAME(tps::ScanArgs a);

TPS_SCAN_KERNEL_DECL(tps_scan_kernel)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s5)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s6)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s7)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s8)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s5p)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s6p)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s7p)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s8p)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s5r)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s6r)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s7r)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s8r)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s5so)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s6so)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s7so)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s8so)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s5sol)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s6sol)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s7sol)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s8sol)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s5sor)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s6sor)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s7sor)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s8sor)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s5sorh)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s6sorh)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s7sorh)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s8sorh)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s3)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s3p)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s4)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s4p)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s9)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s9p)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s10)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s10p)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s11)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s11p)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s12)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s12p)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s5q)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s6q)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s7q)
TPS_SCAN_KERNEL_DECL(tps_scan_kernel_s8q)

#if TPS_IN_GROUP(0)
TPS_SCAN_KERNEL(tps_scan_kernel_s5, 5, false, false, false, 5)       // specialised: compile-time slide, <= 15 patterns
TPS_SCAN_KERNEL(tps_scan_kernel_s6, 6, false, false, false, 5)
TPS_SCAN_KERNEL(tps_scan_kernel_s7, 7, false, false, false, 5)
TPS_SCAN_KERNEL(tps_scan_kernel_s8, 8, false, false, false, 5)
#endif
#if TPS_IN_GROUP(1)
TPS_SCAN_KERNEL(tps_scan_kernel_s5p, 5, false, true, false, 5)       // ... k <= 4: two positions per table lookup
TPS_SCAN_KERNEL(tps_scan_kernel_s6p, 6, false, true, false, 5)
TPS_SCAN_KERNEL(tps_scan_kernel_s7p, 7, false, true, false, 5)
TPS_SCAN_KERNEL(tps_scan_kernel_s8p, 8, false, true, false, 5)
#endif
#if TPS_IN_GROUP(2)
TPS_SCAN_KERNEL(tps_scan_kernel, 0, false, false, true, 4)          // generic: any slide, up to 31 patterns
// (_s6r: PAIR = a pair table of FIELDS for k <= 4 when the planner grants it, ScanArgs::pair_n -- tile_pp_s<.., PAIRF>; at slides 5 and 7 the window's partial block
// ends inside a pair (r = 96 % S is odd), at slide 8 a lane can hold 16 occurrences of a 4-mer: no per-pattern tile there at all)
TPS_SCAN_KERNEL(tps_scan_kernel_s5r, 5, false, false, true, TPS_R_MINW)       // ... with the per-pattern raw counts (TPS_F_STORE_RAW)
TPS_SCAN_KERNEL(tps_scan_kernel_s6r, 6, false, true, true, TPS_R_MINW)
TPS_SCAN_KERNEL(tps_scan_kernel_s7r, 7, false, false, true, TPS_R_MINW)
TPS_SCAN_KERNEL(tps_scan_kernel_s8r, 8, false, false, true, 3)
#endif
#if TPS_IN_GROUP(3)
TPS_SCAN_KERNEL_D(tps_scan_kernel_s5so, 5, true, false, false, TPS_SO_MINW, 2)      // ... self-overlapping k-mers in the table, sums only (tile_lc_s<.., CD>: plain counts, chains corrected), periods 5 and 6
TPS_SCAN_KERNEL_D(tps_scan_kernel_s6so, 6, true, false, false, TPS_SO_MINW, 2)
#endif
#if TPS_IN_GROUP(4)
TPS_SCAN_KERNEL_D(tps_scan_kernel_s7so, 7, true, false, false, TPS_SO_MINW, 2)
TPS_SCAN_KERNEL_D(tps_scan_kernel_s8so, 8, true, false, false, TPS_SO_MINW, 2)
#endif
#if TPS_IN_GROUP(5)
TPS_SCAN_KERNEL(tps_scan_kernel_s5sor, 5, true, false, true, 5)      // ... the same with the per-pattern raw counts (tile_pp_s)
#endif
#if TPS_IN_GROUP(6)
TPS_SCAN_KERNEL(tps_scan_kernel_s6sor, 6, true, false, true, TPS_SOR_MINW)
#endif
#if TPS_IN_GROUP(7)
TPS_SCAN_KERNEL(tps_scan_kernel_s7sor, 7, true, false, true, 5)
#endif
#if TPS_IN_GROUP(8)
TPS_SCAN_KERNEL(tps_scan_kernel_s8sor, 8, true, false, true, TPS_S8SOR_MINW)
#endif
// ... the same for tables of 4^6 and more k-mers: the LDS table holds 16-bit field indices (LUT_F16: 8 KB instead of 16 KB at k = 6, one
// multiply per position more) -- with the slim exchange region five 4-wave workgroups fit a CU instead of two 8-wave ones
#if TPS_IN_GROUP(11)
TPS_SCAN_KERNEL_D(tps_scan_kernel_s5sorh, 5, true, false, true, 5, 3)
#endif
#if TPS_IN_GROUP(12)
TPS_SCAN_KERNEL_D(tps_scan_kernel_s6sorh, 6, true, false, true, 5, 3)
#endif
#if TPS_IN_GROUP(13)
TPS_SCAN_KERNEL_D(tps_scan_kernel_s7sorh, 7, true, false, true, 5, 3)
#endif
#if TPS_IN_GROUP(14)
TPS_SCAN_KERNEL_D(tps_scan_kernel_s8sorh, 8, true, false, true, 5, 3)      // (compiled for 4 waves per SIMD: 185.6 -> 191.0 us, k = 6 at slide 8)
#endif
// ... k = 5 tables without self-overlap, sums only: pair AND single table as 16-bit pattern masks (ScanArgs::pair16) -- the 4^6-entry
// pair table is 8 KB, shared by the 8 waves of a workgroup (three per CU: the same 24 waves as the k = 4 pair kernels)
#if TPS_IN_GROUP(15)
TPS_SCAN_KERNEL_D(tps_scan_kernel_s5q, 5, false, true, false, 5, 4)
TPS_SCAN_KERNEL_D(tps_scan_kernel_s6q, 6, false, true, false, 5, 4)
TPS_SCAN_KERNEL_D(tps_scan_kernel_s7q, 7, false, true, false, 5, 4)
TPS_SCAN_KERNEL_D(tps_scan_kernel_s8q, 8, false, true, false, 5, 4)
#endif
// ... the default (sums only, no self-overlap) kernels for the other slides a window of 100 allows (3, 4, 9 .. 12; round 4: slides outside 5 .. 8 took
// the generic kernel, three to five times slower per window); the raw-row and self-overlap families keep slides 5 .. 8
#if TPS_IN_GROUP(16)
TPS_SCAN_KERNEL(tps_scan_kernel_s3, 3, false, false, false, 5)
TPS_SCAN_KERNEL(tps_scan_kernel_s4, 4, false, false, false, 5)
TPS_SCAN_KERNEL(tps_scan_kernel_s9, 9, false, false, false, 5)
TPS_SCAN_KERNEL(tps_scan_kernel_s10, 10, false, false, false, 5)
TPS_SCAN_KERNEL(tps_scan_kernel_s11, 11, false, false, false, 5)
TPS_SCAN_KERNEL(tps_scan_kernel_s12, 12, false, false, false, 5)
#endif
#if TPS_IN_GROUP(17)
TPS_SCAN_KERNEL(tps_scan_kernel_s3p, 3, false, true, false, 5)
TPS_SCAN_KERNEL(tps_scan_kernel_s4p, 4, false, true, false, 5)
TPS_SCAN_KERNEL(tps_scan_kernel_s9p, 9, false, true, false, 5)
TPS_SCAN_KERNEL(tps_scan_kernel_s10p, 10, false, true, false, 5)
TPS_SCAN_KERNEL(tps_scan_kernel_s11p, 11, false, true, false, 5)
TPS_SCAN_KERNEL(tps_scan_kernel_s12p, 12, false, true, false, 5)
#endif
#if TPS_IN_GROUP(9)
TPS_SCAN_KERNEL_D(tps_scan_kernel_s5sol, 5, true, false, false, TPS_SOL_MINW, 1)     // ... the same for self-overlap periods 2 .. 4
TPS_SCAN_KERNEL_D(tps_scan_kernel_s6sol, 6, true, false, false, TPS_SOL_MINW, 1)
#endif
#if TPS_IN_GROUP(10)
TPS_SCAN_KERNEL_D(tps_scan_kernel_s7sol, 7, true, false, false, TPS_SOL_MINW, 1)
TPS_SCAN_KERNEL_D(tps_scan_kernel_s8sol, 8, true, false, false, TPS_SOL_MINW, 1)
#endif
