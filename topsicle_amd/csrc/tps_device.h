// tps_device.h -- per-read scan logic of the telomere k-mer scanner (MI355X / gfx950).
//
// One WAVE (64 lanes) owns one read and runs, in one launch (4 or 8 independent waves per workgroup):
//   step 1  TRC counts of the first / reversed-last no_bp bases      (allsteps.py:152-204)
//   step 2  sliding-window k-mer counts S_w of the chosen tail        (allsteps.py:257-297)
//   step 3  single-split l2 change-point on S_w                      (allsteps.py:300-333)
//
// Data flow inside the wave (everything between HBM and the result lives in registers and LDS):
//   HBM packed bases (tps_pack.h: 2 bits per base) --one aligned 16-byte quad (64 bases) per lane--> tile in LDS (seq2)
//   seq2 --k-mer code per position--> LDS lookup table (4^k entries over the pattern list; a pair table
//          for k <= 4; a perfect hash of the pattern codes for k > 7)
//   per block of `slide` positions: OR of the masks + running match count
//   per window: S_w = matches + #patterns absent  (OR over the window's blocks, count differences) -> HBM (16-bit)
//   prefix sums of S_w at the change-point candidates --> arg-max of the split gain (f64 fractions compared by
//   cross-multiplication, exact 128-bit integer tournament when float64 cannot separate the best).
//
// Block/window code paths that share everything else:
//   * default (tile_lc_s, template <S>): slide S in {5..8} known at compile time, <= 15 patterns, k <= 7, sums only.  A lane
//     keeps its 8 blocks of the packed read in registers (immediate shifts), publishes one word per block (prefix-OR | count)
//     and computes its OWN 8 windows from registers + one LDS read each; prefix scan in registers; tables with
//     self-overlapping k-mers count plainly and take the chains' skipped occurrences back (CD).
//   * per-pattern tiles (tile_pp_s): the exact count of every pattern in every window (raw rows).
//   * tile_fused_s: the round-1/2 lane-strided tile, kept for tiles with non-ACGT letters and as recount fallback.
//   * generic: any slide, up to 31 patterns, k up to 15; per-block masks in LDS, q+1 reads per window.
//
// The file is written against a tiny portability layer so that the SAME source also builds
// as a sequential host emulation (tests/emu, -DTPS_EMU) for logic tests without a GPU.
// The emulation is test infrastructure only; the product library contains device code only.
#pragma once
#include <stdint.h>
#include "../../include/topsicle_hip.h"

#ifdef TPS_EMU
#define TPS_DEV static inline
#define TPS_HD static inline
#define TPS_PHASE for (int tid = 0; tid < tps::NT; ++tid)
#define TPS_SYNC() ((void)0)
#define TPS_UNROLL
#define TPS_NOVEC
#define TPS_PIN_S(x) ((void)0)
#define TPS_PIN_V(x) ((void)0)
#else
#define TPS_DEV __device__ __forceinline__
#define TPS_HD __host__ __device__ inline
// Every phase gets a FRESH, opaque copy of the lane id (an empty asm the optimiser cannot see through).
// Without it LLVM hoists all lane-dependent address arithmetic of every phase out of the tile loop
// and keeps it live across the whole kernel: measured 99 -> 42 VGPRs on the fused tile alone.
__device__ __forceinline__ int tps_fresh_lane() {
    int t = (int)(threadIdx.x & 63u);
    asm volatile("" : "+v"(t));
    return t;
}
#define TPS_PHASE for (int tid = tps_fresh_lane(), once_ = 1; once_; once_ = 0)
// wave-level synchronisation: a wave's LDS operations execute in issue order, so making earlier LDS
// writes visible to the other lanes of the SAME wave only needs the compiler not to reorder / cache
// across this point (no s_barrier, no cross-wave skew)
#define TPS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define TPS_UNROLL _Pragma("unroll")
// short runtime-bounded loops: keep them as plain scalar loops (the vectoriser turns a 1-2 iteration loop
// into prologue / vector body / epilogue control flow that costs more than the loop)
#define TPS_NOVEC _Pragma("clang loop vectorize(disable) interleave(disable) unroll(disable)")
// (__builtin_amdgcn_sched_barrier(0) was used here to bound register pressure; with ROCm 7.2 it made the
// self-overlap + invalid-base instance of the fused tile nondeterministic on gfx950, and it is no longer
// needed once every phase launders its lane id)
// zero-cost "redefinition" of a wave-uniform value: it stays in an SGPR (or a VGPR lane) instead of being re-loaded from the
// kernel-argument segment inside a loop (an s_load + s_waitcnt that also drains the LDS queue)
#define TPS_PIN_S(x) asm volatile("" : "+s"(x))
// the same for a per-lane value: it is computed HERE (the compiler otherwise sinks pure arithmetic past the wave barrier
// of the next phase and keeps all its inputs alive across it)
#define TPS_PIN_V(x) asm volatile("" : "+v"(x))
#endif

namespace tps {

constexpr int NT = 64;                            // lanes that cooperate on one read: one wave
constexpr int WPG = 4;                            // independent waves (reads) per workgroup: the default, and the fixed size of the small kernels
constexpr int WPG_MAX = 8;                        // the scan kernels take 4 .. 8 waves per workgroup (ScanArgs::wpg): a big table (k >= 6) is shared by more waves
constexpr uint32_t FLAG_CONFLICT = 0x80000000u;   // generic path: bit 31 of a block mask
constexpr uint32_t FLAG16 = 0x8000u;              // specialised path: bit 15 of a 16-bit mask
constexpr int HIST_COPIES = 8;                    // private step-1 histograms (lane % 8)
constexpr int WIN_U = 4;                          // windows per lane and group in the window phase
constexpr int COOP_MAX = 24;                      // fused tiles: up to this many windows are recounted by the whole wave, one at a time

typedef unsigned __int128 u128;

// ------------------------------------------------------------------ portability layer
#ifdef TPS_EMU
TPS_DEV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31));
}
TPS_DEV uint32_t udot4(uint32_t a, uint32_t b) {
    uint32_t s = 0;
    for (int i = 0; i < 4; ++i) s += ((a >> (8 * i)) & 255u) * ((b >> (8 * i)) & 255u);
    return s;
}
TPS_DEV uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) {
    uint32_t out = 0;
    for (int i = 0; i < 4; ++i) {
        uint32_t c = (sel >> (8 * i)) & 255u, byte;
        if (c < 4) byte = (s1 >> (8 * c)) & 255u;
        else if (c < 8) byte = (s0 >> (8 * (c - 4))) & 255u;
        else byte = (c >= 13) ? 255u : 0u;
        out |= byte << (8 * i);
    }
    return out;
}
TPS_DEV int popc(uint32_t x) { return __builtin_popcount(x); }
TPS_DEV int ffs0(uint32_t x) { return __builtin_ctz(x); }
TPS_DEV uint32_t uniform(uint32_t x) { return x; }
TPS_DEV void lds_add(uint32_t* p, uint32_t v) { *p += v; }
TPS_DEV void lds_or(uint32_t* p, uint32_t v) { *p |= v; }
TPS_DEV uint32_t lds_add_ret(uint32_t* p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
TPS_DEV void lds_max_u64(uint64_t* p, uint64_t v) { if (v > *p) *p = v; }
TPS_DEV void lds_max_i32(int32_t* p, int32_t v) { if (v > *p) *p = v; }
struct u32x4 { uint32_t x, y, z, w; };
struct u32x2 { uint32_t x, y; };
TPS_DEV u32x4 load16(const uint8_t* p) { return *(const u32x4*)p; }
TPS_DEV u32x2 load8(const uint8_t* p) { return *(const u32x2*)p; }
TPS_DEV uint32_t bitrev32(uint32_t x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
}
#else
TPS_DEV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
TPS_DEV uint32_t udot4(uint32_t a, uint32_t b) { return __builtin_amdgcn_udot4(a, b, 0u, false); }
TPS_DEV uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
TPS_DEV int popc(uint32_t x) { return __builtin_popcount(x); }
TPS_DEV int ffs0(uint32_t x) { return __builtin_ctz(x); }
// a value every lane of the wave agrees on (e.g. read from LDS): tell the compiler it is scalar
TPS_DEV uint32_t uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
TPS_DEV void lds_add(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
TPS_DEV void lds_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
TPS_DEV uint32_t lds_add_ret(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
TPS_DEV void lds_max_u64(uint64_t* p, uint64_t v) { atomicMax((unsigned long long*)p, (unsigned long long)v); }
TPS_DEV void lds_max_i32(int32_t* p, int32_t v) { atomicMax(p, v); }
typedef uint4 u32x4;
typedef uint2 u32x2;
// 16-byte load that is KNOWN to hit global memory: the address was rebuilt from an integer (aligned
// down), which makes the compiler fall back to FLAT loads -- those also count on lgkmcnt and would
// stall every LDS wait behind the prefetch.  An explicit global address space keeps them on vmcnt.
TPS_DEV u32x4 load16(const uint8_t* p) {
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) v4u* gptr_t;
    const v4u t = *(gptr_t)(uintptr_t)p;
    u32x4 r;
    r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w;
    return r;
}
TPS_DEV u32x2 load8(const uint8_t* p) {
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) v2u* gptr_t;
    const v2u t = *(gptr_t)(uintptr_t)p;
    u32x2 r;
    r.x = t.x; r.y = t.y;
    return r;
}
TPS_DEV uint32_t bitrev32(uint32_t x) { return __builtin_bitreverse32(x); }      // v_bfrev_b32
#endif

// table entry at byte offset `off` (already masked to the table size).  The table starts at the
// workgroup's LDS offset 0, i.e. it is aligned to any power of two, so base | off == base + off and the
// mask + base fold into one v_and_or_b32 -- where the base is a known 0 (every kernel's first table) into one v_and; with mask and
// base both in SGPRs (the single table of the pair-table kernels: 8 lookups per tile at odd r, step 1's) it comes out as v_and +
// v_or, a VOP3 instruction reading one scalar register only on gfx9.  The base pinned into a VGPR gives the one instruction, 6 VALU
// per tile less -- and measures nothing at config 2 (55.0 / 54.8 against 54.8 / 55.0 us), +1 % on the kernels whose base is 0: not kept.
// Nor does a compile-time table address help (tried: a fixed-size pair region, so that the single table starts at LDS byte 4096): the
// base of the dynamic LDS array is resolved after instruction selection -- `| base` of the array's own start is folded late, any
// other constant offset stays an instruction, and written as `+` even the start costs a `v_add_u32 0` per lookup.
#ifdef TPS_EMU
TPS_DEV uint32_t lut_at(const uint32_t* lut, uint32_t v4, uint32_t amask) { return *(const uint32_t*)((const char*)lut + (v4 & amask)); }
#define lut_at_tile lut_at
#else
TPS_DEV uint32_t lut_at(const uint32_t* lut, uint32_t v4, uint32_t amask) {
    typedef const __attribute__((address_space(3))) uint32_t* lptr_t;
    const uint32_t base = (uint32_t)(uintptr_t)(lptr_t)lut;
    return *(lptr_t)(uintptr_t)((v4 & amask) | base);
}
#ifdef TPS_DIAG_NOCONF
/* diagnostic builds only (WRONG window sums, same control flow): the default tile's table gathers with every lane on its own
   bank -- what do the gathers' bank conflicts cost? */
TPS_DEV uint32_t lut_at_tile(const uint32_t* lut, uint32_t v4, uint32_t amask) {
    typedef const __attribute__((address_space(3))) uint32_t* lptr_t;
    const uint32_t base = (uint32_t)(uintptr_t)(lptr_t)lut;
    return *(lptr_t)(uintptr_t)((((v4 & amask) >> 30) | ((threadIdx.x & 63u) << 2)) | base);
}
#else
#define lut_at_tile lut_at
#endif
#endif

#ifndef TPS_WIDE16_MODE
#define TPS_WIDE16_MODE 1         // (0: A/B builds without the pin of lut16_at_wide)
#endif
// 16-bit table entry (LUT_M16 tables: one pattern mask per k-mer code) at byte offset `off2` (already masked to the table size)
#ifdef TPS_EMU
TPS_DEV uint32_t lut16_at(const uint32_t* lut, uint32_t v2, uint32_t amask1) { return *(const uint16_t*)((const char*)lut + (v2 & amask1)); }
#define lut16_at_wide lut16_at
#else
TPS_DEV uint32_t lut16_at(const uint32_t* lut, uint32_t v2, uint32_t amask1) {
    typedef const __attribute__((address_space(3))) uint16_t* lptr16_t;
    const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)lut;
    return *(lptr16_t)(uintptr_t)((v2 & amask1) | base);      // ds_read_u16: zero-extended
}
// ... for the sums tiles of the self-overlap tables (tile_lc_s<.., CD>), opaque to the optimiser: it otherwise narrows everything
// computed from these values to 16-bit arithmetic and legalises that with one `v_and_b32 0xffff` per entry -- 6 of a block's 57 VALU
// instructions (k = 5 sums 107.4 -> 105.7 us, k = 6 142.5 -> 140.6; the raw-row tiles' look-ups are better off without:
// `_s6sorh` 211.6 -> 219.2 us with it, the waits move up to the loads)
TPS_DEV uint32_t lut16_at_wide(const uint32_t* lut, uint32_t v2, uint32_t amask1) {
    uint32_t h = lut16_at(lut, v2, amask1);
#if TPS_WIDE16_MODE == 1
    asm("" : "+v"(h));
#endif
    return h;
}
#endif

// 16-bit candidate sums kept off-chip: explicit global address space (a generic pointer would become FLAT
// instructions, which also count on the LDS counter)
#ifdef TPS_EMU
TPS_DEV void g16_store(uint64_t base, uint32_t i, uint32_t v) { ((uint16_t*)(uintptr_t)base)[i] = (uint16_t)v; }
TPS_DEV uint32_t g16_load(uint64_t base, uint32_t i) { return ((const uint16_t*)(uintptr_t)base)[i]; }
#else
TPS_DEV void g16_store(uint64_t base, uint32_t i, uint32_t v) {
    typedef __attribute__((address_space(1))) uint16_t* gp_t;
    ((gp_t)(uintptr_t)base)[i] = (uint16_t)v;
}
TPS_DEV uint32_t g16_load(uint64_t base, uint32_t i) {
    typedef const __attribute__((address_space(1))) uint16_t* gp_t;
    return ((gp_t)(uintptr_t)base)[i];
}
#endif

#ifdef TPS_EMU
TPS_DEV void g32_store(uint64_t base, uint32_t i, uint32_t v) { ((uint32_t*)(uintptr_t)base)[i] = v; }
TPS_DEV uint32_t g32_load(uint64_t base, uint32_t i) { return ((const uint32_t*)(uintptr_t)base)[i]; }
#else
TPS_DEV void g32_store(uint64_t base, uint32_t i, uint32_t v) {
    typedef __attribute__((address_space(1))) uint32_t* gp_t;
    ((gp_t)(uintptr_t)base)[i] = v;
}
TPS_DEV uint32_t g32_load(uint64_t base, uint32_t i) {
    typedef const __attribute__((address_space(1))) uint32_t* gp_t;
    return ((gp_t)(uintptr_t)base)[i];
}
#endif

// high half of a 32 x 32-bit product (v_mul_hi_u32, full rate) and the sum of the four bytes of a word (v_sad_u8)
TPS_DEV uint32_t mulhi32(uint32_t x, uint32_t y) { return (uint32_t)(((uint64_t)x * (uint64_t)y) >> 32); }
#ifdef TPS_EMU
TPS_DEV uint32_t mul24(uint32_t x, uint32_t y) { return (x & 0xFFFFFFu) * (y & 0xFFFFFFu); }
#else
TPS_DEV uint32_t mul24(uint32_t x, uint32_t y) { return (x & 0xFFFFFFu) * (y & 0xFFFFFFu); }      // (the masks let the compiler pick v_mul_u32_u24)
#endif
#ifdef TPS_EMU
TPS_DEV uint32_t add_bytes(uint32_t v, uint32_t acc) { return acc + (v & 255u) + ((v >> 8) & 255u) + ((v >> 16) & 255u) + (v >> 24); }
#else
TPS_DEV uint32_t add_bytes(uint32_t v, uint32_t acc) { return __builtin_amdgcn_sad_u8(v, 0u, acc); }
#endif

constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }
constexpr int clog2(int x) { return x <= 1 ? 0 : 1 + clog2(x / 2); }

// diagnostics: thread 0 stores the shader clock at phase boundaries when ScanArgs::stamps is set
#if defined(TPS_EMU) || !defined(TPS_STAMPS)
#define TPS_STAMP(i) ((void)0)
#else
#define TPS_STAMP(i) do { if (a.stamps && (threadIdx.x & 63u) == 0) a.stamps[r * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#endif
// ... inside the per-pattern tiles (first tile of a read only): 6 = phase 1 done, 7 = windows done, 11 = rows out, 12 = candidates done
#if defined(TPS_EMU) || !defined(TPS_STAMPS)
#define TPS_PP_STAMP(i) ((void)0)
#else
#define TPS_PP_STAMP(i) do { if (w0 == 0 && a.stamps && (threadIdx.x & 63u) == 0) a.stamps[tc.rd * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#endif

// ------------------------------------------------------------------ kernel arguments
struct PatInfo {
    int32_t P, k;
    uint32_t kmask;          // (1 << 2k) - 1
    uint32_t all_mask;       // (1 << P) - 1
    uint32_t so_mask;        // list patterns that can overlap themselves (have a period < k)
    uint32_t dup_mask;       // list patterns whose k-mer appears more than once in the list
    int32_t n_periods;
    int32_t period[8];       // union of those periods d (1 <= d < k)
    uint32_t period_pat[8];  // list patterns having period d
    // k > TPS_DIRECT_K: the table is a collision-free hash of the P k-mer codes, (key, mask) pairs;
    // slot = (code * hash_mul) >> hash_shift.  hash_shift = 0 means the direct 4^k table.
    uint32_t hash_mul, hash_shift;
};

struct ScanArgs {
    // the packed batch (tps_pack.h): 16 bases per word, reads start on 16-byte quads
    const uint32_t* seq2;        // 2-bit codes
    const uint16_t* inv;         // bit j of inv[w] = base j of word w is not acgtACGT; only read for reads flagged in desc
    const tps_read_desc* desc;   // n: word offset, length, flags of every read
    const uint8_t* tails_in;     // n, or nullptr
    const uint32_t* lut;         // 4^k masks over the pattern list (+ the pair table behind them)
    const uint32_t* lut_img;     // fused kernels: the table as it sits in LDS (mask << 16 | count, one-hot fields or 16-bit masks: lut_dw(a) dwords)
    const uint32_t* pair_img;    // pair-table kernels: the pair table as it sits in LDS (pair_n dwords: mask << 16 | count per (k+1)-mer; pair16: 16-bit masks;
                                 // raw-row kernels with lut_fields: the two positions' one-hot fields added up)
    tps_read_result* results;    // n
    int32_t* c_start;            // n*P or nullptr
    int32_t* c_end;              // n*P or nullptr
    const int64_t* win_off;      // n+1 (window layout of sums/raw)
    int32_t* sums;               // generic kernel: S_w as int32 at win_off[r]; or nullptr
    // fused kernels (round 3): S_w leaves as 16-bit values (S_w <= 15 patterns x 225 occurrences there) -- the window sums are the
    // kernel's largest output and HBM writes its scarcest resource (config 2: 98.7 MB of int32 per launch cost 10 of 68 us).  Every
    // read's region starts at win_off16[r], a multiple of 8 windows (16-byte aligned); the library widens on download.
    uint16_t* sums16;
    const int64_t* win_off16;    // n+1
    uint8_t* raw;                // or nullptr
    uint64_t* stamps;            // diagnostics: 16 clock stamps per read, or nullptr
    int32_t raw_m;               // 0, or 2 (strided scans, tps_plan.h: stride_base): the per-pattern tiles store only every 2nd window's raw row, straight into the
    const int64_t* raw_win_off;  //    layout of the requested slide -- raw = that scan's buffer, raw_win_off = its window layout (tables without self-overlap, clean batches)
    const int32_t* order;        // n, or nullptr: the read wave slot i of the launch takes (plan_dispatch_order: reads in classes of equal work, longest first;
                                 // nullptr = file order, also whenever every read of the batch is in one class)
    int64_t n_reads;
    PatInfo pat;
    tps_params prm;
    // LDS plan (host-computed: plan_geometry() in tps_plan.h)
    int32_t variant;             // 0 = generic path, otherwise the compile-time slide of the kernel
    int32_t lut_n;               // 4^k
    int32_t seq_dw;              // dwords of seq2 (and of val)
    int32_t head_dw;             // dwords reserved per step-1 head inside seq2
    int32_t nblk_cap;            // blocks per tile = spans_per_tile << blk_log2
    int32_t spans_per_tile;
    int32_t span_dw;             // dwords (16 positions each) one span covers = slide / gcd(slide,16)
    int32_t blk_log2;            // log2(blocks per span), blocks per span = 16 / gcd(slide,16)
    int32_t rec_rs;              // specialised path: row stride (records) of the block-record table
    int32_t tot_dw;              // dwords of the Tot array (>= spans_per_tile + 1 and >= NT, even)
    int32_t blk_dw;              // dwords of the block region (also step-1 histograms, Binseg scratch)
    int32_t lc_cap;              // capacity of the candidate-prefix array Lc (u32 entries) = max n_win / jump + 1
    uint32_t jump_magic;         // ceil(2^32 / jump): w / jump == mulhi(w, jump_magic) for w < 2^20; 0 for jump == 1 (see div_jump)
    int32_t q, r, lw;            // window = q full blocks + r positions; lw = W - k start positions
    // fused path, 16-bit candidate sums: Lc16[c] = left sum of candidate c counted from its tile's first
    // window, Tc[t] = sum of S_w before tile t; tile of window w = mulhi(w, tw_magic)
    int32_t lc16;                // 1 = Lc16 + Tc instead of the u32 Lc
    int32_t lc_global;           // 1 = the candidate sums (Lc16, or the generic kernel's 32-bit Lc) live off-chip in lc_scratch (lc_stride entries per read): 1 kB of L2 traffic per
    int32_t lc_stride;           //        read instead of 1 kB of LDS per wave -> one more resident workgroup per CU
    uint16_t* lc_scratch;
    int32_t pair_n;              // 0, or 4^(k+1): entries of the pair table (two adjacent positions per lookup)
    int32_t tile_cap;            // entries of Tc (tiles of the longest read)
    int32_t tw;                  // windows per fused tile
    uint32_t tw_magic;           // ceil(2^32 / tw)
    int32_t tile_full;           // fused tiles: 1 = all 64 lanes hold windows' blocks (8 more windows per tile), 0 = the last lane is halo
    int32_t wpg;                 // waves (reads) per workgroup of this launch, 4 .. WPG_MAX
    int32_t pp_d;                // per-pattern tiles (tile_pp_s): -1 = not eligible, 0 = no self-overlapping k-mer,
                                 // d > 0 = the one self-overlap period of the table
    int32_t so_fast;             // sums only, pp_d > 0: tiles without a chained occurrence take the plain tile (chain test inside it)
    int32_t val_on;              // 1 = some read of the batch has a non-ACGT letter: the invalid masks get their LDS staging area
    int32_t seq_alias;           // fused sums-only kernels without self-overlap: the staged bases share LDS with the tail of row[]
    int32_t lut_fields;          // raw-count kernels on a table the per-pattern tiles take: the LDS table holds ready-made one-hot
                                 // 2-bit fields (1 << 2 p for pattern p) instead of mask << 16 | count
    int32_t lut16;               // sums-only kernels of self-overlap tables: the LDS table holds 16-bit pattern masks (LUT_M16), half the
                                 // bytes of mask << 16 | count -- at k = 6 that is 8 KB instead of 16 KB per workgroup; counts come from v_bcnt
    int32_t xt_alias;            // 1: ... and their XT words (written behind a tile's window phase) share LDS with the head of the staged bases;
                                 // 2: the raw-row kernels -- the per-pattern tiles keep their lane totals in END's pad words and need no XF / XT
    int32_t xt_own;              // xt_alias kernels whose fallback tile may run (non-ACGT letters in the batch, TPS_NO_SO_FAST): XT gets its own words
    int32_t pair16;              // k = 5 tables without self-overlap, sums only (kernels _s*q): BOTH tables as 16-bit pattern masks -- the pair table of
                                 // 4^6 (k+1)-mers is 8 KB instead of 16, shared by the 8 waves of a workgroup (lut16 is set as well)
};

struct BinsegArgs {
    const int32_t* sums;
    const int64_t* win_off;
    int32_t* bkp;
    double* gain;
    int64_t n_reads;
    int32_t n_patterns, jump, min_size;
    uint8_t* tie;                // per read: 1 = the exact tournament decided (TPS_RES_TIE), or nullptr
};

// ------------------------------------------------------------------ LDS carve
constexpr int MISC_DW = 20;
constexpr int XS_DW = 64 + NT + 4 * NT + NT + 16 * 5 + 8;   // scratch of the exact Binseg tournament (aliases the block region)
constexpr int HIST_STRIDE = 33;          // odd stride: the copies of one pattern sit on different LDS banks
constexpr int HIST_DW = 2 * HIST_COPIES * HIST_STRIDE;   // step-1 private histograms (alias the block region)
struct Lds {
    uint32_t* lut;     // generic kernel: mask over the pattern list; fused kernels: mask << 16 | popcount(mask)
    uint32_t* lut2;    // pair kernels: entry of the (k+1)-mer at p = entries of the k-mers at p and p+1 combined (OR | sum)
    int lshift;        // 0 or 16: lut[code] >> lshift is the mask; LUT_FIELDS: one-hot 2-bit fields (lut_mask)
    uint32_t* seq2;    // 2-bit packed bases, 16 per dword
    uint16_t* val;     // bit j of val[c] set = position 16c+j is NOT one of acgtACGT
    uint32_t* blk;     // block region
    // generic path views of blk
    uint32_t* G;       // per block: OR of masks over its `slide` positions (+FLAG_CONFLICT)
    uint32_t* Gp;      // per block: OR over its first r positions
    uint16_t* C0;      // per block: matches before the block (span-local running count)
    uint16_t* C1;      // per block: matches before position r of the block
    // specialised (fused) path views of blk: what a lane publishes about its 8 blocks.  Every word is
    // mask << 16 | count, like the table entries, indexed by the padded block number b + b / 8
    // (lane L's blocks sit at 9 L .. 9 L + 7: conflict-free for lane-contiguous AND lane-strided access)
    uint32_t* XPC;     // per block: OR of the lane's earlier blocks and r positions of this one | lane-local count there
    uint32_t* XF;      // [XLANES] OR of all the lane's block masks (high half)
    uint32_t* XT;      // [XLANES] matches in the lane's span, then exclusive prefix over lanes
    // after phase 1b XF / XT hold, per START lane: OR of the whole lanes a window skips | matches they add,
    // for the near (XF) and the far (XT) end lane
    uint32_t* Tot;     // per span: matches in the span, then exclusive prefix over spans
    uint32_t* Lc;      // Lc[c] = sum of S_w over w < c * jump: left sums of the change-point candidates
    uint16_t* Lc16;    // (lc16) the same, counted from the first window of the candidate's tile
    uint32_t* Tc;      // (lc16) sum of S_w before each tile
    uint32_t* row;     // WIN_U * NT dwords: one group of window sums, scanned in place
    uint32_t* misc;
};
constexpr int XLANES = NT + 16;                  // most lanes an exchange row can hold: NT + halo lanes read past the tile end
TPS_HD int64_t xchg_dw(const ScanArgs& a) {      // fused path: XPC (9 words per lane), XF, XT (xt_alias 1: XT lives elsewhere, 2: both do)
    return 9ll * NT + (a.xt_alias == 1 ? 1ll : a.xt_alias == 2 ? 0ll : 2ll) * XLANES;
}
// dwords of the workgroup's LDS table: 4^k entries of 4 bytes, or of 2 (lut16)
TPS_HD int64_t lut_dw(const ScanArgs& a) { return a.lut16 ? ((((int64_t)a.lut_n + 1) / 2 + 3) & ~3ll) : (((int64_t)a.lut_n + 3) & ~3ll); }
TPS_HD int64_t blk_region_dw(const ScanArgs& a) {
    // generic kernel: G, Gp (u32) and C0, C1 (u16) per block; fused kernels: the exchange arrays
    int64_t need = a.variant ? xchg_dw(a) : 2ll * a.nblk_cap + 2ll * ((a.nblk_cap + 1) / 2);
    if (need < XS_DW) need = XS_DW;
    if (need < HIST_DW) need = HIST_DW;
    return (need + 3) & ~3ll;
}
TPS_HD int64_t val_dw(const ScanArgs& a) { return a.val_on ? ((a.seq_dw + 4 + 3) / 4) * 2 : 0; }   // u16 per 16 positions (+ look-ahead), even (seq_dw is a multiple of 4); none for a batch without invalid letters
TPS_HD int64_t lc_dw(const ScanArgs& a) {          // even dword counts keep misc 8-byte aligned
    if (a.lc16) return (a.lc_global ? 0 : ((a.lc_cap + 3) / 4) * 2) + ((a.tile_cap + 1) / 2) * 2;
    return a.lc_global ? 0 : ((a.lc_cap + 1) / 2) * 2;        // generic kernel: absolute 32-bit sums, also off-chip by default
}
TPS_HD int64_t row_dw(const ScanArgs& a) {
    const int64_t fused = a.variant ? ((int64_t)NT << a.blk_log2) + NT : 0;      // + one pad word per lane
    return fused > WIN_U * NT ? fused : WIN_U * NT;
}
TPS_DEV Lds carve(uint32_t* base, uint32_t* lut, const ScanArgs& a) {
    Lds l;
    l.lut2 = lut - a.pair_n;                   // the pair table precedes the single table (both size-aligned)
    uint32_t* p = base;
    l.blk = p;  p += a.blk_dw;                 // first: 16-byte aligned for the 8-byte records
    l.lut = lut;                               // one table per workgroup, shared by its waves
    l.lshift = a.variant ? 16 : 0;
    l.seq2 = p; p += a.seq_dw;
    l.val = (uint16_t*)p;  p += val_dw(a);
    l.Tot = p;  p += a.tot_dw;
    l.Lc = p;
    l.Lc16 = (uint16_t*)p;
    l.Tc = p + ((a.lc16 && a.lc_global) ? 0 : ((a.lc_cap + 3) / 4) * 2);
    p += lc_dw(a);
    l.row = p;  p += row_dw(a);
    l.misc = p;
    l.G = l.blk;
    l.Gp = l.G + a.nblk_cap;
    l.C0 = (uint16_t*)(l.Gp + a.nblk_cap);
    l.C1 = l.C0 + ((a.nblk_cap + 1) / 2) * 2;
    l.XPC = l.blk;
    l.XF = l.XPC + 9 * NT;
    l.XT = l.XF + XLANES;
    return l;
}
TPS_HD int64_t lds_dwords(const ScanArgs& a) {
    return (int64_t)a.blk_dw + (a.seq_alias ? 0 : a.seq_dw) + val_dw(a) + a.tot_dw + lc_dw(a) + row_dw(a) + MISC_DW +
           ((a.xt_alias && a.xt_own) ? (a.xt_alias == 2 ? 2 : 1) * XLANES : 0);   // per wave; + the table per workgroup
}
// LDS dwords of a whole workgroup: the shared table + WPG wave slices (each rounded to 16 bytes)
TPS_HD int64_t wg_lds_dwords(const ScanArgs& a) { return a.pair_n + lut_dw(a) + (int64_t)(a.wpg > 0 ? a.wpg : WPG) * ((lds_dwords(a) + 3) & ~3ll); }
// misc layout (dwords)
constexpr int M_BEST = 0;        // 2: step-1 arg-max keys (count << 5 | 31 - pattern) of the two sides
constexpr int M_CMASK = 2;       // 2: conflict masks of step 1 (start, end)
constexpr int M_INVALID = 4;     // any non-ACGT base in the staged range
constexpr int M_SCAN = 10;       // 8: workgroup scan scratch (wave totals, grand total)
constexpr int M_MAXSC = 6;       // u64 (8-byte aligned): best f64 score bits
constexpr int M_BESTB = 8;       // i32: largest b among the candidates with the best score
constexpr int M_NTIE = 9;        // candidates within float noise of the best score
// exact Binseg tournament scratch (dwords, relative to its base)
constexpr int X_Q = 0;           // 64: scan partials (NT/16 groups + total)
constexpr int X_BS = 64;         // NT: chunk sums
constexpr int X_CD = X_BS + NT;          // NT x u64: candidate |D|
constexpr int X_CDEN = X_CD + 2 * NT;    // NT x u64: candidate b(n-b)
constexpr int X_CB = X_CDEN + 2 * NT;    // NT x i32: candidate b
constexpr int X_R = X_CB + NT;           // 16 x (u64, u64, i32): second-level reduction

// ------------------------------------------------------------------ workgroup exclusive scan (in place)
// arr[0..n) in LDS -> exclusive prefix sums; returns the grand total.  Called by every lane of
// the wave outside TPS_PHASE.
// Logical entry i lives at arr[i + (i >> pad_log2)]: one pad word per 2^pad_log2 entries keeps the
// per-lane runs (lane l owns entries l*per .. l*per+per-1) on different LDS banks.
TPS_DEV int padded(int i, int pad_log2) { return i + (i >> pad_log2); }
#ifdef TPS_EMU
TPS_DEV uint32_t wg_exclusive_scan(uint32_t* arr, int n, uint32_t* scratch, int pad_log2 = 30) {
    (void)scratch;
    uint32_t run = 0;
    for (int i = 0; i < n; ++i) { uint32_t t = arr[padded(i, pad_log2)]; arr[padded(i, pad_log2)] = run; run += t; }
    return run;
}
#else
TPS_DEV uint32_t wg_exclusive_scan(uint32_t* arr, int n, uint32_t* scratch, int pad_log2 = 30) {
    (void)scratch;
    const int lane = (int)(threadIdx.x & 63u);
    const int per = (n + NT - 1) / NT;
    const int lo = lane * per, hi = (lo + per < n) ? lo + per : n;
    uint32_t s = 0;
    for (int i = lo; i < hi; ++i) s += arr[padded(i, pad_log2)];
    // inclusive wave scan with DPP row shifts / broadcasts: six VALU adds, no LDS round trips
    uint32_t inc = s;
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);   // row_shr:1
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);   // row_shr:2
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);   // row_shr:4
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false);   // row_shr:8
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    uint32_t run = inc - s;
    for (int i = lo; i < hi; ++i) { uint32_t t = arr[padded(i, pad_log2)]; arr[padded(i, pad_log2)] = run; run += t; }
    TPS_SYNC();
    return total;
}
#endif

// wave-wide maximum of an unsigned value (device only: DPP row shifts / broadcasts, no LDS round trip)
#ifndef TPS_EMU
TPS_DEV uint32_t wave_max_u32(uint32_t v) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
#endif

// ------------------------------------------------------------------ staging: packed HBM -> LDS
// The batch is resident in the packed format of tps_pack.h (2 bits per base, 16 bases per word, reads on 16-byte
// boundaries), so staging is a copy: one aligned 16-byte "quad" (64 bases) per lane and load.  A staged range holds
// s-indices [i0, i0+n) of a tail string.  Forward tail: s[i] = seq[t + i]; reverse tail: s[i] = seq[L-1-t-i]
// (allsteps.py:267-271, 176-177) -- quads are then taken in descending address order and each word's sixteen 2-bit
// fields are reversed on the way (v_bfrev_b32 + a swap of the two bits of every field: 5 VALU per 16 bases).  No
// complementing: complement k-mers are already in the table.  Quad c of the range goes to LDS words 4c .. 4c+3 of the
// destination; `delta` (0..63) is the position of s-index i0 inside quad 0.  Every quad touched lies inside the read's
// own allocation (reads are padded to whole quads with zeros), so no bounds masking is needed.
struct Stage {
    uint64_t q0;             // byte address of quad 0 in seq2 (forward: ascending by 16, reverse: descending)
    uint64_t v0;             // byte address of quad 0's invalid words in inv (8 bytes per quad)
    int32_t delta;
    int32_t nq;              // quads that hold staged data
    int32_t n;               // staged s-indices
    bool reverse;
};
TPS_DEV Stage stage_plan(const uint32_t* seq2_base, const uint16_t* inv_base, int64_t word_off, int64_t L, bool reverse, int64_t t, int64_t i0, int32_t n) {
    Stage st;
    st.reverse = reverse;
    st.n = n;
    const int64_t pos = reverse ? (L - 1 - t - i0) : (t + i0);     // read position of s-index i0
    const int64_t w = word_off + 4 * (pos >> 6);
    st.delta = reverse ? 63 - (int32_t)(pos & 63) : (int32_t)(pos & 63);
    st.q0 = (uint64_t)(uintptr_t)seq2_base + (uint64_t)w * 4u;
    st.v0 = (uint64_t)(uintptr_t)inv_base + (uint64_t)w * 2u;
    st.nq = n > 0 ? (st.delta + n + 63) >> 6 : 0;
    return st;
}
// address of quad c of a staged range = a wave-uniform base + an unsigned 32-bit per-lane offset (one scalar-base load
// address instead of 64-bit per-lane arithmetic with a select for the direction)
constexpr int STAGE_KMAX = 1 << 16;              // > quads of any staged range (everything staged fits the 160 KB of LDS)
TPS_DEV const uint8_t* stage_addr(const Stage& st, int c) {
    const uint8_t* base = (const uint8_t*)(uintptr_t)(st.reverse ? st.q0 - 16ull * STAGE_KMAX : st.q0);      // uniform
    const uint32_t off = st.reverse ? 16u * (uint32_t)(STAGE_KMAX - c) : 16u * (uint32_t)c;
    return base + off;
}
TPS_DEV const uint8_t* stage_inv_addr(const Stage& st, int c) {
    const uint8_t* base = (const uint8_t*)(uintptr_t)(st.reverse ? st.v0 - 8ull * STAGE_KMAX : st.v0);
    const uint32_t off = st.reverse ? 8u * (uint32_t)(STAGE_KMAX - c) : 8u * (uint32_t)c;
    return base + off;
}
// the sixteen 2-bit fields of a word in reverse order
TPS_DEV uint32_t rev2(uint32_t x) {
    const uint32_t y = bitrev32(x);               // fields reversed, the two bits of each field swapped
    return ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
}
// a loaded quad in LDS order
TPS_DEV u32x4 stage_orient(const Stage& st, const u32x4& v) {
    if (!st.reverse) return v;
    u32x4 r;
    r.x = rev2(v.w); r.y = rev2(v.z); r.z = rev2(v.y); r.w = rev2(v.x);
    return r;
}
// the four 16-bit invalid words of a quad (two per dword) in LDS order
TPS_DEV u32x2 stage_orient_inv(const Stage& st, const u32x2& v) {
    if (!st.reverse) return v;
    u32x2 r;
    r.x = bitrev32(v.y); r.y = bitrev32(v.x);
    return r;
}
#ifdef TPS_EMU
TPS_DEV void lds_store16(uint32_t* p, const u32x4& v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; }
TPS_DEV void lds_store8(uint32_t* p, const u32x2& v) { p[0] = v.x; p[1] = v.y; }
#else
TPS_DEV void lds_store16(uint32_t* p, const u32x4& v) { *(u32x4*)p = v; }        // ds_write_b128 (p is 16-byte aligned)
TPS_DEV void lds_store8(uint32_t* p, const u32x2& v) { *(u32x2*)p = v; }
#endif

// lane `tid` stages quads tid, tid+NT, ... of the `nqd` quads at seq2 / val (zeros past the staged range); up to four
// 16-byte loads are in flight per lane before the first one is consumed.  The invalid words are only touched for reads
// that have any (has_inv, wave-uniform).
TPS_DEV void stage_thread(const Stage& st, bool has_inv, uint32_t* seq2, uint16_t* val, int nqd, uint32_t* invalid_flag, int tid) {
    for (int base = tid; base < nqd; base += 4 * NT) {
        u32x4 v[4];
        u32x2 b[4];
        TPS_UNROLL
        for (int u = 0; u < 4; ++u) {
            const int c = base + u * NT;
            v[u].x = v[u].y = v[u].z = v[u].w = 0;
            b[u].x = b[u].y = 0;
            if (c < nqd && c < st.nq) {
                v[u] = load16(stage_addr(st, c));
                if (has_inv) b[u] = load8(stage_inv_addr(st, c));
            }
        }
        TPS_UNROLL
        for (int u = 0; u < 4; ++u) {
            const int c = base + u * NT;
            if (c < nqd) {
                lds_store16(seq2 + 4 * c, stage_orient(st, v[u]));
                if (has_inv) {
                    const u32x2 o = stage_orient_inv(st, b[u]);
                    if (o.x | o.y) *invalid_flag = 1u;       // benign race: every writer stores 1
                    lds_store8((uint32_t*)val + 2 * c, o);
                }
            }
        }
    }
}

// 32-bit window (16 bases) starting at position q of a packed array
TPS_DEV uint32_t v_at(const uint32_t* seq2, int q) {
    int idx = q >> 4;
    return alignbit(seq2[idx + 1], seq2[idx], (uint32_t)(q & 15) * 2u);
}
// 1 if any of the k positions q..q+k-1 is not ACGT
TPS_DEV bool invalid_at(const uint16_t* val, int q, int k) {
    int idx = q >> 4;
    uint64_t v = (uint64_t)val[idx] | ((uint64_t)val[idx + 1] << 16) | ((uint64_t)val[idx + 2] << 32);
    return ((v >> (q & 15)) & ((1ull << k) - 1ull)) != 0;
}
// Table formats in LDS (Lds::lshift): 0 = mask over the pattern list (generic kernel), 16 = mask << 16 | popcount (fused
// kernels), LUT_FIELDS = the one-hot 2-bit field 1 << 2 p of THE pattern p the k-mer belongs to (raw-count kernels on tables
// without duplicate k-mers: what the per-pattern tiles and the packed step 1 add up, without the squaring).
constexpr int LUT_FIELDS = 32;
constexpr int LUT_M16 = 48;                       // 16-bit entries: the pattern mask alone (ScanArgs::lut16)
constexpr int LUT_F16 = 64;                       // 16-bit entries: 1 << (field index of THE pattern), pp_field -- squared, the one-hot 2-bit field of
                                                  // LUT_FIELDS (the raw-row kernels of big self-overlap tables, k >= 6: half the LDS table)
// Which of the sixteen 2-bit fields belongs to list pattern p.  The per-pattern tiles widen fields to nibbles (even / odd
// fields: two words) and nibbles to bytes (four words, pp_expand); with THIS assignment the four byte words come out in ROW
// order -- word p / 4, byte p % 4 holds pattern p -- so a raw row is the words as they are (no byte transposition: 8 v_perm per
// window before round 3), and a list of at most 12 patterns never touches the fourth word.
TPS_HD int pp_field(int p) { return 4 * (p & 3) + 2 * ((p >> 2) & 1) + (p >> 3); }
TPS_HD int pp_pattern(int f) { return (f >> 2) + 4 * (((f >> 1) & 1) + 2 * (f & 1)); }
TPS_HD uint32_t mask_to_fields(uint32_t m) {          // pattern mask -> the same patterns as 2-bit fields (bit 2 pp_field(p))
    uint32_t f = 0;
    for (int p = 0; p < 16; ++p) f |= ((m >> p) & 1u) << (2 * pp_field(p));
    return f;
}
TPS_DEV uint32_t field_to_mask(uint32_t f) { return f ? 1u << pp_pattern(ffs0(f) >> 1) : 0u; }                // one-hot field -> pattern mask
TPS_DEV uint32_t field_to_entry(uint32_t f) { return f ? ((1u << (16 + pp_pattern(ffs0(f) >> 1))) | 1u) : 0u; } // ... -> mask << 16 | 1
template <int N> struct IntC { static constexpr int value = N; };
// mask of list patterns whose k-mer is the low 2k bits of v (direct table or perfect hash)
TPS_DEV uint32_t lut_mask(const uint32_t* lut, int lshift, const PatInfo& pat, uint32_t v) {
    const uint32_t code = v & pat.kmask;
    if (pat.hash_shift) {
        const uint32_t slot = (code * pat.hash_mul) >> pat.hash_shift;
        return lut[2 * slot] == code ? lut[2 * slot + 1] : 0u;
    }
    if (lshift == LUT_FIELDS) return field_to_mask(lut[code]);
    if (lshift == LUT_M16) return ((const uint16_t*)lut)[code];
    if (lshift == LUT_F16) { const uint32_t m = ((const uint16_t*)lut)[code]; return m ? 1u << pp_pattern(ffs0(m)) : 0u; }
    return lut[code] >> lshift;
}
// mask of list patterns whose k-mer starts at position q
TPS_DEV uint32_t h_at(const uint32_t* lut, int lshift, const uint32_t* seq2, const uint16_t* val, const PatInfo& pat, int q, bool any_invalid) {
    uint32_t h = lut_mask(lut, lshift, pat, v_at(seq2, q));
    if (any_invalid && h && invalid_at(val, q, pat.k)) h = 0;
    return h;
}
// patterns p (subset of `h`) that occur again d < k positions later (d a period of p)
TPS_DEV uint32_t conflict_bits(const PatInfo& pat, uint32_t v, uint32_t h) {
    // long k-mers (hashed table): k + d bases do not fit the 16-base word, so every occurrence of a
    // self-overlapping pattern is reported (conservative: the exact recount then decides)
    if (pat.hash_shift) return h & pat.so_mask;
    uint32_t c = 0;
    TPS_NOVEC
    for (int i = 0; i < pat.n_periods; ++i) {
        uint32_t hp = h & pat.period_pat[i];
        if (hp && (((v ^ (v >> (2 * pat.period[i]))) & pat.kmask) == 0)) c |= hp;
    }
    return c;
}

// Leftmost non-overlapping count of list pattern `bit` over `npos` start positions from
// position q0 -- exactly what len(list(re.finditer(p, text))) gives (allsteps.py:182, 281).
TPS_DEV void greedy_count(const uint32_t* lut, int lshift, const uint32_t* seq2, const uint16_t* val, const PatInfo& pat, int q0,
                          int npos, int bit, bool any_invalid, int& occ, int& greedy) {
    occ = 0; greedy = 0;
    int cursor = 0;
    for (int p = 0; p < npos; ++p) {
        uint32_t h = h_at(lut, lshift, seq2, val, pat, q0 + p, any_invalid);
        if ((h >> bit) & 1u) {
            ++occ;
            if (p >= cursor) { ++greedy; cursor = p + pat.k; }
        }
    }
}

// ------------------------------------------------------------------ step 1: TRC counts of both tails
// Lanes 0-31 take the first-bases head, 32-63 the reversed-last-bases head.  A lane handles
// groups of 8 consecutive start positions: two LDS reads give the 16 packed bases that hold all
// eight k-mers (k <= 7), the eight table lookups are independent.  Matches go to one of 16
// private histograms (lane % 16) so that LDS atomics rarely collide.
template <bool PLAIN>
TPS_DEV void trc_count_thread(const ScanArgs& a, const Lds& l, const Stage& st_s, const Stage& st_e, int tid) {
    // PLAIN: no non-ACGT letter staged, no self-overlapping k-mer, no duplicate in the pattern list --
    // the common case, one predicated LDS atomic per matching position and nothing else.
    const PatInfo& pat = a.pat;
    const int side = tid >> 5, t = tid & 31;
    const int delta = side ? st_e.delta : st_s.delta;
    const uint32_t* seq2 = l.seq2 + side * a.head_dw;
    const uint16_t* val = l.val + side * a.head_dw;
    const bool inv = !PLAIN && l.misc[M_INVALID] != 0;
    const bool so = !PLAIN && pat.so_mask != 0;
    const int npos = st_s.n - pat.k + 1;
    uint32_t* hist = l.blk + (side * HIST_COPIES + (tid & (HIST_COPIES - 1))) * HIST_STRIDE;
    uint32_t cm = 0;
    for (int p0 = t * 8; p0 < npos; p0 += 32 * 8) {
        const int q0 = delta + p0, idx = q0 >> 4;
        const uint32_t sh = (uint32_t)(q0 & 15) * 2u;
        const uint32_t d0 = seq2[idx], d1 = seq2[idx + 1], d2 = seq2[idx + 2];
        const uint32_t w0 = alignbit(d1, d0, sh), w1 = alignbit(d2, d1, sh);
        uint32_t h[8], v[8];
        TPS_UNROLL
        for (int j = 0; j < 8; ++j) {
            v[j] = j ? alignbit(w1, w0, 2u * j) : w0;
            h[j] = lut_mask(l.lut, l.lshift, pat, v[j]);
        }
        TPS_UNROLL
        for (int j = 0; j < 8; ++j) {
            uint32_t hj = h[j];
            if (p0 + j >= npos) hj = 0;
            if (PLAIN) {
                if (hj) lds_add(&hist[ffs0(hj)], 1u);
            } else {
                if (hj && inv && invalid_at(val, q0 + j, pat.k)) hj = 0;
                if (hj) {
                    if (so && (hj & pat.so_mask)) cm |= conflict_bits(pat, v[j], hj);
                    do {
                        int b = ffs0(hj);
                        hj &= hj - 1;
                        lds_add(&hist[b], 1u);
                    } while (hj);
                }
            }
        }
    }
    if (!PLAIN && cm) lds_or(&l.misc[M_CMASK + side], cm);
}
// Step 1, packed path (fused kernels, PLAIN case: every position matches at most one pattern and
// occurrences of one pattern never overlap).  Lane (side, t) scans the 16-position chunks t, t+32, ...
// of its side.  A table entry is e = 1 << (16 + p) | 1 for a position where pattern p starts (no duplicate
// k-mers), so mulhi(e, e) = 1 << 2p: a one-hot 2-bit field per pattern (v_mul_hi_u32 + one add per position; no
// predication, no LDS atomics).  Eight consecutive positions hold at most ceil(8 / min period) <= 3 occurrences of a
// pattern, so two half-chunks are added up in 2-bit fields each and then widened to nibbles (even / odd patterns:
// two words); a lane sees `iters` chunks = at most iters * ceil(16 / k) <= 15 occurrences per pattern.
// The lane then widens its nibbles to bytes (side totals <= npos / k <= 255) and parks them in LDS:
// 16 bytes per lane = [patterns 0,4,8,12 | 2,6,10,14 | 1,5,9,13 | 3,7,11,15].
// Tables with self-overlapping k-mers (no invalid letter, no duplicate) take the same path: occurrences are
// counted as they are, and a pattern that matches at p and again d < k bases later (d one of its periods;
// one AND per position and period on the table entries, six look-ahead entries per chunk) is reported in
// the side's conflict mask -- only those patterns are then recounted leftmost-non-overlapping.  An
// occurrence can repeat every min-period bases, which bounds the 4-bit and 8-bit fields.
TPS_DEV int min_period(const PatInfo& pat) {
    int m = pat.k;
    TPS_NOVEC
    for (int i = 0; i < pat.n_periods; ++i) m = pat.period[i] < m ? pat.period[i] : m;
    return m;
}
TPS_DEV bool trc_packed_ok(const ScanArgs& a, int npos) {
    const int nchunks = (npos + 15) >> 4, iters = (nchunks + 31) >> 5;
    const int mp = min_period(a.pat);
    int maxd = 0;
    TPS_NOVEC
    for (int i = 0; i < a.pat.n_periods; ++i) maxd = a.pat.period[i] > maxd ? a.pat.period[i] : maxd;
    return a.pat.P <= 15 && maxd <= 6 && mp >= 3 && npos <= 255 * mp && iters * ((16 + mp - 1) / mp) <= 15;
}
// FMT 1: the table holds one-hot 2-bit fields (LUT_FIELDS): no squaring; FMT 2: 16-bit pattern masks (LUT_M16): the one-hot mask
// 1 << p squares to the field 1 << 2 p by a plain 24-bit multiply.
// SO_ (round 4): tables with self-overlapping k-mers.  re.finditer counts leftmost non-overlapping occurrences
// (allsteps.py:182): along a chain of occurrences d apart (d the k-mers' one period) it takes every other one, so a chain of n
// counts ceil(n / 2) = n - pairs + triples for n <= 3, where a pair / triple = occurrences at j, j + d (, j + 2 d).  Two
// occurrences d apart ARE k + d bases of period d, so pairs and triples are found on the packed bases themselves, 16 positions at a
// time: one XOR of the chunk's registers against themselves d bases on, an AND-fold to runs of k equal comparisons, two more ANDs
// for the triples and the chains of four -- ~30 instructions per chunk and no extra lookups.  Only a chunk that holds a pair
// (a deleted base inside the telomere: ~15 % of the chunks at ONT error rates) walks its 16 positions again to take the
// pairs off the lane's per-pattern counts and add the triples back; a chain of FOUR or more (two per thousand bases even there)
// flags its pattern for the exact recount (trc_publish_occ / trc_walk_occ), as does any overlapping pair in a table with more
// than one period.  Round 3 flagged every pair: at k = 6 every ONT read took the recount, 60 of a read's 132 thousand clocks.
template <bool SO_, int FMT = 0>
TPS_DEV void trc_count_packed(const ScanArgs& a, const Lds& l, const Stage& st_s, const Stage& st_e, int tid, uint32_t* keep = nullptr) {
    const PatInfo& pat = a.pat;
    const int side = tid >> 5, t = tid & 31;
    const int delta = side ? st_e.delta : st_s.delta;
    const uint32_t* seq2 = l.seq2 + side * a.head_dw;
    constexpr bool FLD = FMT == 1, M16 = FMT == 2 || FMT == 3, F16 = FMT == 3;   // (3: LUT_F16 -- 16-bit entries that square to fields in ROW order)
    constexpr int LS = M16 ? 1 : 2;                 // log2(bytes per table entry)
    const uint32_t amask = pat.kmask << LS;
    const int npos = st_s.n - pat.k + 1;
    const int nchunks = (npos + 15) >> 4;
    const int k = pat.k;
    auto fieldsq = [&](uint32_t h) -> uint32_t { return FLD ? h : M16 ? mul24(h, h) : mulhi32(h, h); };   // table entry -> one-hot 2-bit field
    uint32_t ne = 0, no = 0;                        // per-pattern counts of this lane, nibbles: even / odd patterns
    uint32_t pe = 0, po = 0, te = 0, to = 0;        // SO_: the same for pairs and triples (counted at their first element)
    uint32_t cf = 0;                                // SO_: entries of the patterns that need the exact recount
    for (int c0 = 0; c0 < nchunks; c0 += 32) {      // uniform trip count
        const int c = c0 + t;
        // the base BEFORE the chunk's first one goes to bit 0 (4-byte entries; 2-byte entries: its high bit): alignbit(.., 2 j) &
        // (kmask << LS) is then the table's byte offset of position j (the bits below it are masked away)
        const int bo = 2 * (delta + 16 * c) - LS;
        const int idx = bo >> 5;                     // -1 for the very first chunk of an aligned head: harmless
        const uint32_t sh = (uint32_t)(bo & 31);
        const uint32_t d0 = seq2[idx], d1 = seq2[idx + 1], d2 = seq2[idx + 2];
        const uint32_t w0 = alignbit(d1, d0, sh), w1 = alignbit(d2, d1, sh);
        uint32_t h[16];
        TPS_UNROLL
        for (int j = 0; j < 16; ++j) {
            const uint32_t v_ = j == 0 ? w0 : alignbit(w1, w0, 2u * j);
            h[j] = M16 ? lut16_at(l.lut, v_, amask) : lut_at(l.lut, v_, amask);
        }
        const bool last_pass = 16 * (c0 + 32) + 32 > npos;      // uniform: the pass that holds the end of the head
        if (last_pass) {
            int nv = npos - 16 * c;                  // start positions of the head in this lane's chunk (<= 0: none)
            TPS_PIN_V(nv);                           // (one subtraction, then compares against constants -- not an add per position)
            TPS_UNROLL
            for (int j = 0; j < 16; ++j)
                if (j >= nv) h[j] = 0;
        }
        TPS_UNROLL
        for (int half = 0; half < 2; ++half) {
            uint32_t x2 = 0;                         // 2-bit fields: <= 3 occurrences of a pattern in 8 positions
            TPS_UNROLL
            for (int j = 8 * half; j < 8 * half + 8; ++j) x2 += fieldsq(h[j]);
            ne += x2 & 0x33333333u;
            no += (x2 >> 2) & 0x33333333u;
        }
        if (SO_) {
            const uint32_t w2 = alignbit(seq2[idx + 3], d2, sh);
            // the chunk's bases from bit 0 on: positions 0 .. 15 | 16 .. 31 | 32 .. 47 (the last word a fraction of a base short)
            const uint32_t a0 = alignbit(w1, w0, (uint32_t)LS), a1 = alignbit(w2, w1, (uint32_t)LS), a2 = w2 >> LS;
            TPS_NOVEC
            for (int pi = 0; pi < pat.n_periods; ++pi) {            // (one period: every telomere motif tried)
                const int d = pat.period[pi];
                const uint32_t x0 = a0 ^ alignbit(a1, a0, 2u * (uint32_t)d), x1 = a1 ^ alignbit(a2, a1, 2u * (uint32_t)d);
                const uint32_t z0 = ~(x0 | (x0 >> 1)) & 0x55555555u, z1 = ~(x1 | (x1 >> 1)) & 0x55555555u;   // bit 2 i: base i == base i + d
                uint32_t r0 = z0, r1 = z1;           // bit 2 j: the k-mers at j and j + d are the same (k equal comparisons from j on)
                TPS_NOVEC
                for (int u = 1; u < k; ++u) { r0 &= alignbit(z1, z0, 2u * (uint32_t)u); r1 &= z1 >> (2 * u); }
                // r is known up to position 32 - k; pairs, triples and chains of four by their FIRST element j = 0 .. 15
                uint32_t pr = r0, tr = 0, qd = 0;
                if (pat.n_periods == 1) {
                    tr = r0 & alignbit(r1, r0, 2u * (uint32_t)d);
                    qd = tr & alignbit(r1, r0, 4u * (uint32_t)d);
                    const int known = 32 - k - 2 * d;                 // first elements up to here see their whole chain of four
                    if (known < 15) qd |= tr & ~((known < 0) ? 0u : ((4u << (2 * known)) - 1u));      // (the others: a triple counts as one)
                } else {
                    qd = pr;                           // several periods: any overlapping pair sends its pattern to the recount
                }
                if (last_pass) {
                    // both (all three, four) elements are start positions of the head
                    auto below = [&](int n) -> uint32_t { return n <= 0 ? 0u : n >= 16 ? 0xFFFFFFFFu : ((1u << (2 * n)) - 1u); };
                    pr &= below(npos - 16 * c - d);
                    tr &= below(npos - 16 * c - 2 * d);
                    if (pat.n_periods == 1) qd &= below(npos - 16 * c - 2 * d); else qd &= below(npos - 16 * c - d);
                }
#ifndef TPS_TRC_PAIRWALK
#define TPS_TRC_PAIRWALK 1        // 0: A/B builds with round 4's walk over all 16 positions of a chunk that holds a pair
#endif
#if TPS_TRC_PAIRWALK != 0 && !defined(TPS_EMU_OLD_PAIRWALK)
                // a pair starts in this chunk (or the bases are periodic without a pattern): its pattern is looked up again, pair by pair --
                // a lane holds one or two (a deleted base inside the telomere), where walking all 16 positions of the chunk with three masks
                // each cost 160 instructions per pass whenever ANY lane of the wave had one: always, on telomeric heads at ONT error rates
                {
                    uint32_t m = pr;
                    while (m != 0u) {
                        const uint32_t j2 = (uint32_t)ffs0(m);              // = 2 j
                        m &= m - 1u;
                        const uint32_t hj = M16 ? lut16_at(l.lut, alignbit(w1, w0, j2), amask) : lut_at(l.lut, alignbit(w1, w0, j2), amask);
                        const uint32_t f = fieldsq(hj);
                        pe += f & 0x33333333u; po += (f >> 2) & 0x33333333u;
                        if ((tr >> j2) & 1u) { te += f & 0x33333333u; to += (f >> 2) & 0x33333333u; }
                        if ((qd >> j2) & 1u) cf |= hj;
                    }
                }
#else
                if (pr != 0u) {                        // a pair starts in this chunk (or the bases are periodic without a pattern)
                    uint32_t xp[2] = {0u, 0u}, xt[2] = {0u, 0u};
                    TPS_UNROLL
                    for (int j = 0; j < 16; ++j) {
                        const uint32_t hp = h[j] & (0u - ((pr >> (2 * j)) & 1u));
                        const uint32_t ht = h[j] & (0u - ((tr >> (2 * j)) & 1u));
                        xp[j >> 3] += fieldsq(hp);
                        xt[j >> 3] += fieldsq(ht);
                        cf |= h[j] & (0u - ((qd >> (2 * j)) & 1u));
                    }
                    TPS_UNROLL
                    for (int half = 0; half < 2; ++half) {
                        pe += xp[half] & 0x33333333u; po += (xp[half] >> 2) & 0x33333333u;
                        te += xt[half] & 0x33333333u; to += (xt[half] >> 2) & 0x33333333u;
                    }
                }
#endif
            }
        }
    }
    if (SO_ && cf) {
        uint32_t cm = M16 ? cf : cf >> 16;
        if (FLD || F16) {                           // fields / field indices -> pattern mask (rare: only a lane that saw a long chain)
            cm = 0;
            while (cf) { cm |= 1u << pp_pattern(F16 ? ffs0(cf) : ffs0(cf) >> 1); cf &= cf - 1u; }
        }
        lds_or(&l.misc[M_CMASK + side], cm);
    }
    uint32_t* dst = keep ? keep : l.blk + 4 * tid;  // (keep: the caller sums the lanes' words itself -- trc_decide_packed)
    // bytes per pattern: occurrences - pairs + triples (a lane's pairs are among its occurrences, its triples among its pairs:
    // no byte borrows)
    dst[0] = (ne & 0x0F0F0F0Fu) - (pe & 0x0F0F0F0Fu) + (te & 0x0F0F0F0Fu);
    dst[1] = ((ne >> 4) & 0x0F0F0F0Fu) - ((pe >> 4) & 0x0F0F0F0Fu) + ((te >> 4) & 0x0F0F0F0Fu);
    dst[2] = (no & 0x0F0F0F0Fu) - (po & 0x0F0F0F0Fu) + (to & 0x0F0F0F0Fu);
    dst[3] = ((no >> 4) & 0x0F0F0F0Fu) - ((po >> 4) & 0x0F0F0F0Fu) + ((to >> 4) & 0x0F0F0F0Fu);
}
// Recount of the (few) patterns with overlapping occurrences, cooperatively: every lane looks its chunks up
// again and publishes, per conflicting pattern, the 16 occurrence bits of each chunk (u16 per chunk: a
// side's head is <= 64 chunks = 32 dwords per pattern); the pattern's lane then walks the bits
// leftmost-non-overlapping.  Up to OCC_SLOTS conflicting patterns per side; more fall back to the
// sequential greedy_count.
constexpr int OCC_SLOTS = 4;
constexpr int OCC_BASE_DW = 256;                  // behind the 64 x 16 bytes of packed counters in the block region
TPS_DEV int occ_slot(uint32_t cmask, int p) {     // index of pattern p among the set bits of cmask
    return popc(cmask & ((1u << p) - 1u));
}
TPS_DEV void trc_publish_occ(const ScanArgs& a, const Lds& l, const Stage& st_s, const Stage& st_e, int tid) {
    const PatInfo& pat = a.pat;
    const int side = tid >> 5, t = tid & 31;
    const uint32_t cmask = l.misc[M_CMASK + side];
    if (!cmask || popc(cmask) > OCC_SLOTS || st_s.n - a.pat.k + 1 > 1024) return;
    const int delta = side ? st_e.delta : st_s.delta;
    const uint32_t* seq2 = l.seq2 + side * a.head_dw;
    const uint32_t amask = pat.kmask << 2;
    const int npos = st_s.n - pat.k + 1;
    const int nchunks = (npos + 15) >> 4;
    uint16_t* occ = (uint16_t*)(l.blk + OCC_BASE_DW + side * OCC_SLOTS * 32);
    for (int c = t; c < 64; c += 32) {
        uint32_t bits[OCC_SLOTS] = {0, 0, 0, 0};
        if (c < nchunks) {
            const int qm = delta + 16 * c - 1;
            const int idx = qm >> 4;
            const uint32_t sh = (uint32_t)(qm & 15) * 2u;
            const uint32_t d0 = seq2[idx], d1 = seq2[idx + 1], d2 = seq2[idx + 2];
            const uint32_t w0 = alignbit(d1, d0, sh), w1 = alignbit(d2, d1, sh);
            TPS_UNROLL
            for (int j = 0; j < 16; ++j) {
                // (w0 holds the base before the chunk at bit 0: two bits above it the k-mer code starts -- any table format)
                uint32_t h = lut_mask(l.lut, l.lshift, pat, (j ? alignbit(w1, w0, 2u * j) : w0) >> 2);
                if (16 * c + j >= npos) h = 0;
                h &= cmask;
                while (h) {
                    const int b = ffs0(h);
                    h &= h - 1;
                    const int sl = occ_slot(cmask, b);
                    TPS_UNROLL
                    for (int q = 0; q < OCC_SLOTS; ++q)
                        if (q == sl) bits[q] |= 1u << j;
                }
            }
        }
        TPS_UNROLL
        for (int q = 0; q < OCC_SLOTS; ++q) occ[q * 64 + c] = (uint16_t)bits[q];
    }
}
TPS_DEV int trc_walk_occ(const Lds& l, int side, int slot, int k) {
    const uint32_t* occ = l.blk + OCC_BASE_DW + (side * OCC_SLOTS + slot) * 32;
    int greedy = 0, cursor = 0;
    for (int wd = 0; wd < 32; ++wd) {
        uint32_t m = occ[wd];
        while (m) {
            const int p = 32 * wd + ffs0(m);
            m &= m - 1;
            if (p >= cursor) { ++greedy; cursor = p + k; }
        }
    }
    return greedy;
}
// Thread (side, p): add the 32 lanes' byte of pattern p (a pattern with overlapping occurrences is recounted
// leftmost-non-overlapping: sequential, rare), publish the count and bid for the arg-max.
TPS_DEV void trc_sum_packed(const ScanArgs& a, const Lds& l, const Stage& st_s, const Stage& st_e, int64_t r, int tid) {
    const int side = tid >> 5, p = tid & 31;
    if (p < a.pat.P) {
        // byte of pattern p in a lane's 16 parked bytes: fields in list order (mask tables, squared entries) put it in word
        // [0, 2, 1, 3][p & 3], byte p >> 2; ready-made fields (LUT_FIELDS tables: pp_field) in row order, byte p
        const int pbyte = (l.lshift == LUT_FIELDS || l.lshift == LUT_F16) ? p : 4 * (((p & 1) << 1) | ((p >> 1) & 1)) + (p >> 2);
        const uint8_t* src = (const uint8_t*)(l.blk + 4 * 32 * side) + pbyte;
        uint32_t sm = 0;
        TPS_UNROLL
        for (int t = 0; t < 32; ++t) sm += src[16 * t];
        const uint32_t cmask = l.misc[M_CMASK + side];
        if ((cmask >> p) & 1u) {
            if (popc(cmask) <= OCC_SLOTS && st_s.n - a.pat.k + 1 <= 1024) {
                sm = (uint32_t)trc_walk_occ(l, side, occ_slot(cmask, p), a.pat.k);
            } else {
                int occ, g;
                greedy_count(l.lut, l.lshift, l.seq2 + side * a.head_dw, l.val + side * a.head_dw, a.pat, side ? st_e.delta : st_s.delta,
                             st_s.n - a.pat.k + 1, p, false, occ, g);
                sm = (uint32_t)g;
            }
        }
        int32_t* dst = side ? a.c_end : a.c_start;
        if (dst) dst[r * a.pat.P + p] = (int32_t)sm;
        lds_max_i32((int32_t*)&l.misc[M_BEST + side], (int32_t)((sm << 5) | (uint32_t)(31 - p)));
    }
}

#ifndef TPS_EMU
// Plain tables (no self-overlap, no duplicates, clean heads) in the default kernels: count, sum and decide without leaving the
// registers.  The 32 lanes of a side add their four words of per-pattern bytes by DPP (a side's count of a pattern is at most
// 255: trc_packed_ok), lanes 31 and 63 hold the sides' totals, and the arg-max over the patterns runs on the scalar unit --
// instead of parking 16 bytes per lane in LDS, a barrier, 32 byte reads by each of P lanes, an LDS atomic max, another barrier
// and two uniform reads (a fifth of step 1's latency, which every read pays before its first tile can be requested).
// Returns the keys count << 5 | (31 - p) of the first pattern with the largest count, per side.
template <int FMT = 0>
TPS_DEV void trc_decide_packed(const ScanArgs& a, const Lds& l, const Stage& st_s, const Stage& st_e, int64_t r, uint32_t& ks, uint32_t& ke) {
    uint32_t x[4];
    TPS_PHASE { trc_count_packed<false, FMT>(a, l, st_s, st_e, tid, x); }
    uint32_t s0[4], s1[4];
    TPS_UNROLL
    for (int w = 0; w < 4; ++w) {
        uint32_t v = x[w];
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
        s0[w] = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
        s1[w] = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    }
    // pattern p's byte: word [0, 2, 1, 3][p & 3], byte p >> 2 (the layout trc_sum_packed reads from LDS)
    const int P = a.pat.P;
    uint32_t bs = 0, be = 0;
    TPS_UNROLL
    for (int p = 0; p < 15; ++p) {
        if (p < P) {                                  // uniform
            const int wi = ((p & 1) << 1) | ((p >> 1) & 1), sh = 8 * (p >> 2);
            const uint32_t c0 = (s0[wi] >> sh) & 255u, c1 = (s1[wi] >> sh) & 255u;
            const uint32_t k0 = (c0 << 5) | (uint32_t)(31 - p), k1 = (c1 << 5) | (uint32_t)(31 - p);
            bs = k0 > bs ? k0 : bs;
            be = k1 > be ? k1 : be;
        }
    }
    ks = bs;
    ke = be;
    if (a.c_start) {                                  // the per-pattern counts of both heads, if the caller asked for them
        const int lane = (int)(threadIdx.x & 63u), side = lane >> 5, p = lane & 31;
        if (p < P) {
            const int wi = ((p & 1) << 1) | ((p >> 1) & 1);
            const uint32_t w0 = wi == 0 ? s0[0] : wi == 1 ? s0[1] : wi == 2 ? s0[2] : s0[3];
            const uint32_t w1 = wi == 0 ? s1[0] : wi == 1 ? s1[1] : wi == 2 ? s1[2] : s1[3];
            const uint32_t c = ((side ? w1 : w0) >> (8 * (p >> 2))) & 255u;
            (side ? a.c_end : a.c_start)[r * P + p] = (int32_t)c;
        }
    }
}
#endif

// Thread (side, p): sum the private histograms; if pattern p has overlapping occurrences, recount
// it leftmost-non-overlapping (sequential, rare); publish the count and bid for the side's
// arg-max with key = count << 5 | (31 - p), so the FIRST pattern with the largest count wins.
TPS_DEV void trc_sum_thread(const ScanArgs& a, const Lds& l, const Stage& st_s, const Stage& st_e, int64_t r, int tid) {
    const int side = tid >> 5, p = tid & 31;
    if (p < a.pat.P) {
        uint32_t sm = 0;
        TPS_UNROLL
        for (int c = 0; c < HIST_COPIES; ++c) sm += l.blk[(side * HIST_COPIES + c) * HIST_STRIDE + p];
        if ((l.misc[M_CMASK + side] >> p) & 1u) {
            int occ, g;
            greedy_count(l.lut, l.lshift, l.seq2 + side * a.head_dw, l.val + side * a.head_dw, a.pat, side ? st_e.delta : st_s.delta,
                         st_s.n - a.pat.k + 1, p, l.misc[M_INVALID] != 0, occ, g);
            sm = (uint32_t)g;
        }
        int32_t* dst = side ? a.c_end : a.c_start;
        if (dst) dst[r * a.pat.P + p] = (int32_t)sm;
        lds_max_i32((int32_t*)&l.misc[M_BEST + side], (int32_t)((sm << 5) | (uint32_t)(31 - p)));
    }
}

// ------------------------------------------------------------------ step 2 (generic path), phase B: blocks
// Thread owns span `span` of the tile: span_dw dwords = (1<<blk_log2) blocks of `slide`
// positions, starting at a dword boundary + (delta & 15).
TPS_DEV void blocks_span(const ScanArgs& a, const Lds& l, int delta, int span) {
    const PatInfo& pat = a.pat;
    const int s = a.prm.slide, r = a.r, k = pat.k;
    const bool inv = l.misc[M_INVALID] != 0;
    const bool so = pat.so_mask != 0;
    const uint32_t sh2 = (uint32_t)(delta & 15) * 2u;
    const int d0 = (delta >> 4) + span * a.span_dw;
    int j = span << a.blk_log2;
    uint32_t lo = l.seq2[d0], hi = l.seq2[d0 + 1];
    uint32_t cur = alignbit(hi, lo, sh2);
    uint32_t g = 0, cnt = 0;
    int pib = 0;                                  // position in block: identical in every lane
    for (int dw = 0; dw < a.span_dw; ++dw) {
        uint32_t nx2 = l.seq2[d0 + dw + 2];
        uint32_t nxt = alignbit(nx2, hi, sh2);
        uint32_t v[16], h[16];
        TPS_UNROLL
        for (int i = 0; i < 16; ++i) {
            v[i] = i ? alignbit(nxt, cur, 2u * i) : cur;
            h[i] = lut_mask(l.lut, 0, pat, v[i]);
        }
        if (inv) {
            int q = ((d0 + dw) << 4) + (delta & 15);
            TPS_UNROLL
            for (int i = 0; i < 16; ++i)
                if (h[i] && invalid_at(l.val, q + i, k)) h[i] = 0;
        }
        uint32_t cf = 0;                          // bit i: position i starts an overlapping pair
        if (so) {
            TPS_UNROLL
            for (int i = 0; i < 16; ++i)
                if ((h[i] & pat.so_mask) && conflict_bits(pat, v[i], h[i])) cf |= 1u << i;
        }
        TPS_UNROLL
        for (int i = 0; i < 16; ++i) {
            if (pib == 0) {
                g = 0;
                l.C0[j] = (uint16_t)cnt;
                if (r == 0) { l.C1[j] = (uint16_t)cnt; l.Gp[j] = 0; }
            }
            g |= h[i];
            if ((cf >> i) & 1u) g |= FLAG_CONFLICT;
            cnt += (uint32_t)popc(h[i]);
            ++pib;
            if (pib == r) { l.C1[j] = (uint16_t)cnt; l.Gp[j] = g; }
            if (pib == s) { l.G[j] = g; ++j; pib = 0; }
        }
        hi = nx2;
        cur = nxt;
    }
    l.Tot[span] = cnt;
}

// ------------------------------------------------------------------ step 2, exact window recount
// Sequential scan of one window's `npos` start positions from LDS position q0 with a rolling 32-base
// register window (one LDS read per 16 positions): calls f(p, mask) for every position.
template <typename F>
TPS_DEV void scan_positions(const Lds& l, const PatInfo& pat, int q0, int npos, bool inv, F f) {
    const int idx = q0 >> 4;
    const uint32_t sh = (uint32_t)(q0 & 15) * 2u;
    uint32_t d1 = l.seq2[idx + 1];
    uint32_t cur = alignbit(d1, l.seq2[idx], sh), nxt = 0;
    for (int p = 0; p < npos; ++p) {
        const int j = p & 15;
        if (j == 0) {
            if (p) cur = nxt;
            const uint32_t d2 = l.seq2[idx + (p >> 4) + 2];
            nxt = alignbit(d2, d1, sh);
            d1 = d2;
        }
        const uint32_t v = j ? alignbit(nxt, cur, 2u * (uint32_t)j) : cur;
        uint32_t h = lut_mask(l.lut, l.lshift, pat, v);
        if (inv && h && invalid_at(l.val, q0 + p, pat.k)) h = 0;
        f(p, h);
    }
}

// Exact S_w of tile-local window wl (and, if raw_row is set, the per-pattern counts c'_p).
// Only self-overlapping k-mers can make the leftmost-non-overlapping count (what re.finditer yields,
// allsteps.py:281) differ from the plain occurrence count, so:
//   sums only : S_w = fast S_w - sum over present self-overlapping patterns of (occurrences - greedy)
//   raw counts: one pass accumulates all occurrence counts in packed bytes, then the present
//               self-overlapping patterns are recounted greedily; zeros are floored to 1.
//   full      : the raw-count procedure without a row to store (present = every pattern that may overlap).
TPS_DEV uint32_t window_exact(const ScanArgs& a, const Lds& l, int delta, int wl, uint32_t present, uint32_t fast_sw,
                              uint8_t* raw_row, bool full = false) {
    const PatInfo& pat = a.pat;
    const bool inv = l.misc[M_INVALID] != 0;
    const int q0 = delta + wl * a.prm.slide;
    uint32_t redo = present & pat.so_mask;
    if (!raw_row && !full) {
        uint32_t sum = fast_sw;
        while (redo) {
            const int bit = ffs0(redo);
            redo &= redo - 1;
            int occ = 0, greedy = 0, cursor = 0;
            scan_positions(l, pat, q0, a.lw, inv, [&](int p, uint32_t h) {
                if ((h >> bit) & 1u) {
                    ++occ;
                    if (p >= cursor) { ++greedy; cursor = p + pat.k; }
                }
            });
            sum -= (uint32_t)(occ - greedy);
        }
        return sum;
    }
    uint64_t clo = 0, chi = 0, c3 = 0, c4 = 0;          // packed byte counters: patterns 0-7, 8-15, 16-23, 24-31
    scan_positions(l, pat, q0, a.lw, inv, [&](int, uint32_t h) {
        while (h) {
            const int b = ffs0(h);
            h &= h - 1;
            const uint64_t inc = 1ull << (8 * (b & 7));
            const int g = b >> 3;
            clo += g == 0 ? inc : 0ull;
            chi += g == 1 ? inc : 0ull;
            c3 += g == 2 ? inc : 0ull;
            c4 += g == 3 ? inc : 0ull;
        }
    });
    uint32_t sum = 0;
    for (int b = 0; b < pat.P; ++b) {
        const uint64_t word = (b >> 3) == 0 ? clo : (b >> 3) == 1 ? chi : (b >> 3) == 2 ? c3 : c4;
        int c = (int)((word >> (8 * (b & 7))) & 255u);
        if ((redo >> b) & 1u) {
            int greedy = 0, cursor = 0;
            scan_positions(l, pat, q0, a.lw, inv, [&](int p, uint32_t h) {
                if (((h >> b) & 1u) && p >= cursor) { ++greedy; cursor = p + pat.k; }
            });
            c = greedy;
        }
        if (c == 0) c = 1;                         // `matches or 1` (allsteps.py:281, 288)
        if (raw_row) raw_row[b] = (uint8_t)c;
        sum += (uint32_t)c;
    }
    return sum;
}

// One group of WIN_U * NT consecutive windows of the tile, starting at tile-local window `base`:
// lane `tid` computes windows base + u*NT + tid, stores S_w to HBM and leaves it in row[u*NT + tid]
// (0 beyond the tile) for the prefix scan that follows.
TPS_DEV void windows_group(const ScanArgs& a, const Lds& l, int delta, int w0, int nw_tile, int64_t out_base, int base, int tid) {
    const PatInfo& pat = a.pat;
    const int q = a.q;
    for (int u = 0; u < WIN_U; ++u) {
        const int wl = base + u * NT + tid;
        uint32_t sw = 0;
        if (wl < nw_tile) {
            uint32_t m = l.Gp[wl + q];
            for (int i = 0; i < q; ++i) m |= l.G[wl + i];
            int je = wl + q;
            uint32_t cnt = ((uint32_t)l.C1[je] + l.Tot[je >> a.blk_log2]) - ((uint32_t)l.C0[wl] + l.Tot[wl >> a.blk_log2]);
            uint32_t present = m & pat.all_mask;
            sw = cnt + (uint32_t)(pat.P - popc(present));
            uint8_t* raw_row = a.raw ? a.raw + (out_base + w0 + wl) * (int64_t)pat.P : nullptr;
            if ((m & FLAG_CONFLICT) || raw_row) sw = window_exact(a, l, delta, wl, present, sw, raw_row);
            a.sums[out_base + w0 + wl] = (int32_t)sw;
        }
        l.row[u * NT + tid] = sw;
    }
}

// w / jump by the host-supplied magic multiplier (jump == 1 has no 32-bit magic: ceil(2^32 / 1) = 2^32)
TPS_DEV uint32_t div_jump(uint32_t w, uint32_t magic) { return magic ? (uint32_t)(((uint64_t)w * magic) >> 32) : w; }
// After the in-place exclusive scan of row[]: lane `tid` records the left sums of the change-point
// candidates among its windows (global window index divisible by jump).
TPS_DEV void candidates_group(const ScanArgs& a, const Lds& l, uint64_t lc_g, int w0, int nw_tile, int base, uint32_t carry, int tid) {
    const uint32_t jump = (uint32_t)a.prm.jump;
    TPS_UNROLL
    for (int u = 0; u < WIN_U; ++u) {
        const int wl = base + u * NT + tid;
        if (wl < nw_tile) {
            const uint32_t w = (uint32_t)(w0 + wl);
            const uint32_t c = div_jump(w, a.jump_magic);
            if (c * jump == w && (int)c < a.lc_cap) {
                if (lc_g) g32_store(lc_g, c, carry + l.row[u * NT + tid]);
                else l.Lc[c] = carry + l.row[u * NT + tid];
            }
        }
    }
}

// ------------------------------------------------------------------ step 2 (specialised path)
// Compile-time slide S: a span is SPAN = S/gcd(S,16) dwords = B = 16/gcd(S,16) blocks, kept in
// registers with immediate shifts.  Blocks are grouped in aligned chunks of C = min(B, 8): per
// block the record holds the OR over the rest of its chunk (suffix) and the OR over the chunk's
// earlier blocks plus this block's first r positions (prefix), so a window's presence mask is
// suffix[first block] | full chunks in between | prefix[partial block].
// per-read copies of the plan constants the fused tile needs (pinned: see TPS_PIN_S)
struct TileConst {
    int32_t q, r;
    uint32_t jump, jump_magic, lc_cap, lc16;
    uint64_t lc_g;               // this read's off-chip candidate sums: Lc16, or absolute 32-bit sums (0 = they live in LDS)
    uint16_t* sw16;              // this read's window sums (16-bit, ScanArgs::sums16)
    int64_t rd;                  // (diagnostics build: the read index, for the clock stamps of the per-pattern tiles)
};
TPS_DEV TileConst tile_const(const ScanArgs& a, int64_t r) {
    TileConst t;
    t.lc_g = a.lc_global ? (uint64_t)(uintptr_t)(a.lc_scratch + r * (int64_t)a.lc_stride) : 0ull;
    t.sw16 = a.sums16 + (a.win_off16 ? a.win_off16[r] : 0);
    t.rd = r;
    t.q = (int32_t)uniform((uint32_t)a.q); t.r = (int32_t)uniform((uint32_t)a.r);
    t.jump = uniform((uint32_t)a.prm.jump); t.jump_magic = uniform(a.jump_magic);
    t.lc_cap = uniform((uint32_t)a.lc_cap); t.lc16 = uniform((uint32_t)a.lc16);
    TPS_PIN_S(t.q); TPS_PIN_S(t.r); TPS_PIN_S(t.jump); TPS_PIN_S(t.jump_magic); TPS_PIN_S(t.lc_cap); TPS_PIN_S(t.lc16);
    return t;
}

#ifndef TPS_RAW_M
#define TPS_RAW_M 1           // 0: A/B builds without the every-second-row store of the strided scans (ScanArgs::raw_m)
#endif
template <int S>
struct Geo {
    static constexpr int B = 8;                   // blocks (= windows) per lane
    static constexpr int LOG2B = 3;
    static constexpr int C = 8;                   // chunk of the prefix/suffix ORs = the lane's blocks
    static constexpr int LOG2C = 3;
    static constexpr int POS = B * S;             // positions per lane
    // dwords (16 positions each) a lane needs after shifting its first position to bit 0: its
    // positions plus the look-ahead of the last k-mer and of the overlap test (k + d <= 13 bases)
    static constexpr int WDW = (POS + 13 + 15) / 16;
};
// Staging geometry of a tile: NT lanes of positions (+ look-ahead and the sub-quad start offset), staged as whole
// 16-byte quads (64 bases) of the packed batch, PF per lane (1 up to slide 7, a nearly empty second one at slide 8).
// All 64 lanes hold window blocks (FULL tiles: 512 - q - 1 windows); HALO tiles (last lane unused) are kept as a
// template option for experiments only -- with packed input a FULL tile costs no extra staging work at any slide.
constexpr int SEQ_LEAD = 4;                       // LDS words in front of a staged tile: lanes read the base before their first, the per-pattern tiles 16 more
template <int S, bool FULL>
struct TileGeo {
    static constexpr int LANES = FULL ? NT : NT - 1;
    static constexpr int BASES = LANES * Geo<S>::POS + 13 + 15;       // positions the lanes touch, look-ahead included
    static constexpr int NQ = (63 + BASES + 63) / 64;                 // quads of a tile whatever its sub-quad start offset
    static constexpr int PF = (NQ + NT - 1) / NT;
    static constexpr int SEQ_TILE = SEQ_LEAD + 4 * NQ + 4;            // dwords of seq2: lead, the tile, look-ahead reads of the last lane
    static constexpr int SEQ = SEQ_TILE < 144 ? 144 : SEQ_TILE;       // ... and never less than step 1's two heads of 1000 bases need (slide 3: 112 for the tile)
};
constexpr bool tile_full_default(int s) { return s >= 1; }

// LDS slice of a wave in the fused kernels: everything whose size is known at compile time comes first, at
// compile-time offsets from the slice base (one SGPR for all of it, offsets folded into the DS instructions);
// only the candidate / tile sums, whose size depends on the longest read, follow.  Sizes = plan_geometry's.
// XTA (the sums-only kernels of self-overlap tables, ScanArgs::xt_alias): XT -- written behind a tile's window phase, read by its
// candidate phase -- shares its words with the head of the staged bases, which nothing reads after the tile's first phase
// (tile_lc_s<.., CD>: the lanes' registers and the chain walks).  80 dwords per wave less: with the 16-bit table that is the fifth
// 4-wave workgroup per CU at k = 6 (8 192 + 4 x 5 856 B = 31 616 <= 32 000).  The fallback tile (tile_fused_s: recounts read the
// bases AFTER it has rewritten XT) gets XT words of its own behind everything else whenever it can run (ScanArgs::xt_own).
// XM = 2 (the raw-row kernels, ScanArgs::xt_alias == 2): no XF / XT at all in the exchange region -- tile_pp_s keeps the lanes'
// totals in the pad words of END; the fallback tile gets both behind everything else whenever it can run (xt_own).
template <int S, bool FULL, int XM = 0>
TPS_DEV Lds carve_fused(uint32_t* base, uint32_t* lut, const ScanArgs& a) {
    constexpr bool XTA = XM == 1;
    constexpr int BLK = 9 * NT + (XM == 0 ? 2 : XM == 1 ? 1 : 0) * XLANES, ROW = NT * Geo<S>::B + NT, SEQ = TileGeo<S, FULL>::SEQ;
    const int VAL = a.val_on ? ((SEQ + 4 + 3) / 4) * 2 : 0;
    static_assert(BLK % 4 == 0 && ROW % 4 == 0 && MISC_DW % 4 == 0 && SEQ % 4 == 0, "seq2 must be 16-byte aligned");
    // seq_alias: the staged bases live in the LAST SEQ dwords of row[].  A tile's lanes read them into registers at the very
    // start of phase 1 (and step 1 reads its heads there while it counts into the block region in front); everything a tile
    // writes to row[] afterwards (XS, then S_w and its scan) comes later in program order, and a wave's LDS operations
    // execute in order -- the next tile's staging overwrites a row[] that is no longer needed.  Only for the kernels that
    // never go back to the bases after phase 1 (no self-overlap recounts, no raw counts).
    static_assert(BLK + ROW - SEQ >= HIST_DW && BLK + ROW - SEQ >= 4 * NT, "step 1 counts in front of the aliased bases");
    Lds l;
    l.lut2 = lut - a.pair_n;
    l.lut = lut;
    l.lshift = (a.lut16 && !a.lut_fields && !a.pair16) ? LUT_M16 : (a.lut_fields && a.lut16) ? LUT_F16 : a.lut_fields ? LUT_FIELDS : 16;     // (P16K kernels: scan_read sets LUT_M16 itself)
    l.blk = base;
    l.XPC = base;
    l.XF = base + 9 * NT;
    l.XT = l.XF + XLANES;                      // (XTA: set below)
    l.G = l.Gp = base; l.C0 = l.C1 = (uint16_t*)base;      // generic-path views: unused here
    l.row = base + BLK;
    l.misc = l.row + ROW;
    l.Tot = l.misc + MISC_DW;
    uint32_t* p = l.Tot + 4;                   // 16-byte aligned: BLK, ROW, MISC_DW are multiples of 4 dwords
    if (a.seq_alias) {
        l.seq2 = base + (BLK + ROW - SEQ);
    } else {
        l.seq2 = p;
        p += SEQ;
    }
    l.val = (uint16_t*)p;
    p += VAL;
    l.Lc = p;
    l.Lc16 = (uint16_t*)p;
    l.Tc = p + ((a.lc16 && a.lc_global) ? 0 : ((a.lc_cap + 3) / 4) * 2);
    if (XTA) l.XT = a.xt_own ? p + lc_dw(a) : l.seq2;
    if (XM == 2) {                             // (only the fallback tile reads these; without xt_own it cannot run)
        l.XF = p + lc_dw(a);
        l.XT = l.XF + XLANES;
    }
    return l;
}

// One fused tile: NT lanes x B blocks.
//   phase 1  (lane-contiguous) a lane scans its B blocks with the packed bases in registers and
//            publishes two words per block, both mask << 16 | count like the table entries:
//              XS [b]  OR of this and the lane's later blocks | lane-local count before the block
//              XPC[b]  OR of the lane's earlier blocks and r positions of this one | count there
//            plus the OR / match total of the whole lane (XF, XT).
//   phase 1b (after the scan of XT) per start lane: OR and match count of the whole lanes a window
//            skips between its first and last lane, for both possible lane distances (FOA, FOB).
//   phase 2  (lane-strided) window w = 64 u + lane starts at block w and ends r positions into block
//            w + q, which lies in a LATER lane because q >= 8:  XS[w] | skipped lanes | XPC[w + q].
//            Three LDS reads at per-lane bases + immediates, S_w to HBM coalesced and into row[].
//   phase 3  (after the exclusive scan of row[]) candidate-strided: Lc[c] = left sum of window c * jump.
// XS aliases row[]: the only reader of XS[w] is the lane that then writes row[w].
// Windows beyond nw_tile (they need blocks of the next tile) are not produced.
#ifdef TPS_EMU
inline int& emu_counter(int i) { static int c[8] = {0, 0, 0, 0, 0, 0, 0, 0}; return c[i]; }   // tests: 0 = per-pattern tiles, 1 = windows recounted there, 4 = exact change-point tournaments, 5 = sums tiles of a self-overlap table with chains corrected, 6 = ... without a chain
#endif
TPS_DEV uint32_t pack_hi_lo(uint32_t hi_src, uint32_t lo_src) { return perm(hi_src, lo_src, 0x07060100u); }

// RAW: this instantiation can also produce the per-pattern counts (TPS_F_STORE_RAW); the kernels without it carry
// no recount code at all unless the table has self-overlapping k-mers.
// RPT: the window's partial block (r = (W - k) % slide positions) as a COMPILE-TIME constant, or -1 = read it at run time.
// A window ends r positions into a block, so every block publishes its prefix words at that position; with r known at
// compile time the capture is a plain copy at one unrolled position instead of two selects at every position (measured on
// the slide-7 kernel, r = 4: 112 of ~565 instructions per tile).  scan_read switches on r once per tile.
template <int S, bool SO, bool INV, int RPT, bool PAIR, bool RAW>
TPS_DEV void tile_fused_s(const ScanArgs& a, const TileConst& tc, const Lds& l, int delta, int w0, int tile, int nw_tile,
                          int64_t out_base, uint64_t& s_total, int64_t r) {
    // RZ: the window has no partial block (W - k divisible by the slide), so nothing is captured mid-block
    constexpr bool RZ = RPT == 0;
    typedef Geo<S> g_;
    constexpr int WDW = g_::WDW, B = g_::B, LOG2B = g_::LOG2B, POS = g_::POS;
    constexpr int RS = NT + NT / B;               // row stride of the padded layout between u and u + 1
    const PatInfo& pat = a.pat;
    const int rp = RPT >= 0 ? RPT : tc.r, q = tc.q;   // rp: positions of the partial block (a.r)
    const uint32_t amask = pat.kmask << 2;        // k-mer code as a byte offset into the 4-byte table
    TPS_PHASE {
        const int span = tid;
        // The lane's bases start at an arbitrary bit offset; one per-lane funnel shift aligns the base
        // BEFORE its first one to bit 0.  Then `alignbit(w[dw+1], w[dw], 2*(p%16)) & (kmask << 2)` is
        // the k-mer code of position p times 4 -- the table's byte offset -- with immediate shifts only.
        const int p0 = delta + span * POS;        // >= 16: fused tiles are staged one dword into seq2
        const uint32_t sh2 = (uint32_t)((p0 - 1) & 15) * 2u;
        const int d0 = (p0 - 1) >> 4;
        uint32_t w[WDW];
        {
            uint32_t prev = l.seq2[d0];
            TPS_UNROLL
            for (int i = 0; i < WDW; ++i) {
                uint32_t nx = l.seq2[d0 + i + 1];
                w[i] = alignbit(nx, prev, sh2);
                prev = nx;
            }
        }
        uint32_t cnt = 0, run_or = 0;             // table entries are mask << 16 | popcount: OR keeps the masks
        uint32_t gs[B], c0s[B];                   // in the high half, ADD the match count in the low half
        uint32_t* xs = l.row + span * (B + 1);
        uint32_t* xpc = l.XPC + span * (B + 1);
        if constexpr (PAIR) {
            // Pair table: one lookup covers positions p and p+1 (entry = OR of their masks | sum of their
            // counts), so a block costs S/2 lookups (+ one single for an odd slide).  Only tables
            // without self-overlapping k-mers on tiles without invalid letters take this path.
            static_assert(!SO && !INV, "pair lookups need per-position independence");
            constexpr int NP = S / 2, NH = NP + (S & 1);
            const uint32_t amask2 = (pat.kmask << 4) | 0xCu;     // (k+1)-mer code as a byte offset
            const int rpe = rp & ~1;                              // capture point rounded down to a pair boundary
            uint32_t hc[NH], hn[NH];
            auto v4_at = [&](int p) -> uint32_t {
                const int dw = p >> 4, bit = p & 15;
                return bit ? alignbit(dw + 1 < WDW ? w[dw + 1] : 0u, w[dw], 2u * bit) : w[dw];
            };
            auto fetch = [&](int blk, uint32_t* hh) {
                TPS_UNROLL
                for (int j = 0; j < NP; ++j) hh[j] = lut_at(l.lut2, v4_at(blk * S + 2 * j), amask2);
                if (S & 1) hh[NP] = lut_at(l.lut, v4_at(blk * S + S - 1), amask);
            };
            fetch(0, hc);
            TPS_UNROLL
            for (int blk = 0; blk < B; ++blk) {
                if (blk + 1 < B) fetch(blk + 1, hn);
                uint32_t g = 0;
                c0s[blk] = cnt;
                if (RZ) xpc[blk] = pack_hi_lo(run_or, cnt);
                TPS_UNROLL
                for (int j = 0; j < NH; ++j) {
                    if (!RZ) {
                        if (2 * j == rpe) {          // uniform: the window's partial block ends inside / before this pair
                            uint32_t c1 = cnt, pp = run_or | g;
                            if (rp & 1) {
                                const uint32_t h1 = lut_at(l.lut, v4_at(blk * S + 2 * j), amask);
                                c1 += h1;
                                pp |= h1;
                            }
                            xpc[blk] = pack_hi_lo(pp, c1);
                        }
                    }
                    g |= hc[j];
                    cnt += hc[j];
                }
                gs[blk] = g;
                run_or |= g;
                TPS_UNROLL
                for (int j = 0; j < NH; ++j) hc[j] = hn[j];
            }
        } else {
            // table lookups run one block ahead of their use (software pipeline, 2 S values in flight)
            uint32_t hc[S], hn[S];
            auto fetch = [&](int blk, uint32_t* hh, int cnt_) {
                TPS_UNROLL
                for (int i = 0; i < S; ++i) {
                    if (i < cnt_) {
                        const int p = blk * S + i;        // constant after unrolling
                        const int dw = p >> 4, bit = p & 15;
                        uint32_t v4 = bit ? alignbit(dw + 1 < WDW ? w[dw + 1] : 0u, w[dw], 2u * bit) : w[dw];
                        uint32_t h;
                        if constexpr (SO && !RAW) {       // these kernels' table holds 16-bit masks (LUT_M16): rebuild mask << 16 | count
                            const uint32_t m16 = lut16_at(l.lut, v4 >> 1, amask >> 1);
                            h = (m16 << 16) | (uint32_t)popc(m16);
                        } else {
                            h = lut_at(l.lut, v4, amask);
                        }
                        if constexpr (INV && RAW) {
                            if (a.lut_fields && a.lut16) {              // (16-bit field-index table: the entry read above is not one)
                                const uint32_t m16 = lut16_at(l.lut, v4 >> 1, amask >> 1);
                                h = m16 ? ((1u << (16 + pp_pattern(ffs0(m16)))) | 1u) : 0u;
                            } else if (a.lut_fields) {
                                h = field_to_entry(h);                  // (a tile with non-ACGT letters of a batch on the per-pattern tiles)
                            }
                        }
                        if (INV) {
                            if (h && invalid_at(l.val, p0 + p, pat.k)) h = 0;   // tiles with non-ACGT letters only
                        }
                        hh[i] = h;
                    } else {
                        hh[i] = 0;
                    }
                }
            };
            // Self-overlapping k-mers: pattern b has OVERLAPPING occurrences iff it matches at p and again at
            // p + d, d one of its periods (d < k).  With the entries of the next block at hand that is one AND per
            // position and period: ppd[d] = the patterns with period d (in the entries' mask half).  The planner
            // only picks these kernels when every period is <= the slide, so p + d lies in this or the next block.
            constexpr int MAXD = S < 6 ? S : 6;
            uint32_t ppd[MAXD + 1];
            if (SO) {
                TPS_UNROLL
                for (int d = 0; d <= MAXD; ++d) ppd[d] = 0;
                TPS_NOVEC
                for (int i = 0; i < pat.n_periods; ++i) {
                    TPS_UNROLL
                    for (int d = 1; d <= MAXD; ++d)
                        if (pat.period[i] == d) ppd[d] |= pat.period_pat[i] << 16;
                }
            }
            fetch(0, hc, S);
            TPS_UNROLL
            for (int blk = 0; blk < B; ++blk) {
                if (blk + 1 < B) fetch(blk + 1, hn, S);
                else if (SO) fetch(B, hn, MAXD);          // look-ahead past the lane's last block (w[] holds 13 extra bases)
                uint32_t g = 0;
                c0s[blk] = cnt;
                uint32_t c1 = cnt, pp = run_or;
                if (SO) {
                    uint32_t cf = 0;
                    TPS_UNROLL
                    for (int d = 1; d <= MAXD; ++d) {
                        if (ppd[d]) {                     // uniform
                            TPS_UNROLL
                            for (int i = 0; i < S; ++i) cf |= hc[i] & (i + d < S ? hc[i + d] : hn[i + d - S]) & ppd[d];
                        }
                    }
                    if (cf) g = FLAG16 << 16;
                }
                TPS_UNROLL
                for (int i = 0; i < S; ++i) {
                    const uint32_t h = hc[i];
                    g |= h;
                    cnt += h;
                    if (!RZ) {
                        if (i + 1 == rp) { c1 = cnt; pp = run_or | g; }
                    }
                }
                xpc[blk] = pack_hi_lo(pp, c1);
                gs[blk] = g;
                run_or |= g;
                TPS_UNROLL
                for (int i = 0; i < S; ++i) hc[i] = hn[i];
            }
        }
        uint32_t sfx = 0;
        TPS_UNROLL
        for (int j = B - 1; j >= 0; --j) {
            sfx |= gs[j];
            xs[j] = pack_hi_lo(sfx, c0s[j]);
        }
        l.XF[span] = pack_hi_lo(sfx, cnt);        // the lane's OR | the lane's matches
    }
    TPS_SYNC();
    if (w0 == 0) TPS_STAMP(6);
    if (w0 == 0) TPS_STAMP(7);
    const int rot = q & (B - 1), dl0 = q >> LOG2B;
    uint32_t fo_a = 0, fo_b = 0;
#ifdef TPS_EMU
    uint32_t fo_keep[NT][2];
#endif
    TPS_PHASE {
        // A window's last (partial) block lies dl0 or dl0+1 lanes ahead of its first.  What it skips: the OR of
        // the whole lanes strictly in between (the next dl0-1, or dl0, lanes) and the matches of every lane from
        // its own up to the one before the last (dl0, or dl0+1, lanes) -- sums of a few neighbours' words, so no
        // prefix scan over the lanes is needed.
        uint32_t orw = 0, sumw = l.XF[tid];
        TPS_NOVEC
        for (int t = 1; t < dl0; ++t) {
            const uint32_t v = l.XF[tid + t];
            orw |= v;
            sumw += v;
        }
        const uint32_t vb = l.XF[tid + dl0];
        fo_a = pack_hi_lo(orw, sumw);
        fo_b = pack_hi_lo(orw | vb, sumw + vb);
#ifdef TPS_EMU
        fo_keep[tid][0] = fo_a; fo_keep[tid][1] = fo_b;
#endif
    }
    TPS_SYNC();                                   // every lane has read its inputs: XF / XT are reused in place
    TPS_PHASE {
#ifdef TPS_EMU
        fo_a = fo_keep[tid][0]; fo_b = fo_keep[tid][1];
#endif
        l.XF[tid] = fo_a;
        l.XT[tid] = fo_b;
    }
    TPS_SYNC();
#ifdef TPS_EMU
    uint32_t redo_keep[NT][1 + Geo<S>::B / 2];
#else
    uint32_t redo_flags = 0, redo_present[Geo<S>::B / 2];
    TPS_UNROLL
    for (int t = 0; t < Geo<S>::B / 2; ++t) redo_present[t] = 0;
#endif
    TPS_PHASE {
        const uint32_t lane = (uint32_t)tid;
        const bool farl = (((lane & (B - 1)) + (uint32_t)rot) >> LOG2B) != 0;
        uint32_t* ps = l.row + (lane + (lane >> LOG2B));                         // XS in, S_w out
        const uint32_t* pe = l.XPC + ((lane + (uint32_t)q) + ((lane + (uint32_t)q) >> LOG2B));
        const uint32_t* pf = (farl ? l.XT : l.XF) + (lane >> LOG2B);
        uint16_t* out = tc.sw16 + w0;
        uint16_t* outl = out + lane;
        const uint32_t am = pat.all_mask << 16;
        const int nfull = nw_tile >> 6;                                          // uniform
        const uint32_t npart = (uint32_t)(nw_tile & 63);
        uint32_t flags = 0;                          // bit u: window 64 u + lane needs the exact recount
        uint32_t present[B / 2];                     // SO only: the windows' presence masks, two per word
        TPS_UNROLL
        for (int t = 0; t < B / 2; ++t) present[t] = 0;
        // all LDS reads of the lane's windows are issued before the first one is used: one LDS round trip
        // per tile instead of one per window (rows past the tile's windows read in-bounds garbage)
        uint32_t xv[B], ev[B], fv[B];
        TPS_UNROLL
        for (int u = 0; u < B; ++u) { xv[u] = ps[u * RS]; ev[u] = pe[u * RS]; fv[u] = pf[u * (NT / B)]; }   // unconditional: no branch, no zeroing
        TPS_UNROLL
        for (int u = 0; u < B; ++u) {
            uint32_t sw = 0;
            if (u <= nfull) {
                const bool valid = (u < nfull) || (lane < npart);
                const uint32_t x = xv[u], e = ev[u], f = fv[u];
                const uint32_t m = x | e | f;
                const uint32_t s_ = ((e - x + f) & 0xFFFFu) + (uint32_t)popc(~m & am);
                if (valid) {
                    sw = s_;
                    if (SO) {
                        flags |= (m >> 31) << u;
                        present[u / 2] |= (u & 1) ? (m & 0xFFFF0000u) : (m >> 16);
                    }
                    outl[u * NT] = (uint16_t)sw;       // per-lane base + immediate offset
                }
            }
            ps[u * RS] = sw;
        }
        // windows that need the exact recount (overlapping occurrences of a self-overlapping k-mer, or raw
        // counts wanted) are queued: entry = tile-local window | its presence mask << 16.  The queue lives in
        // the XPC area, which no window needs any more.
        if (SO || (RAW && a.raw)) {
            if (RAW && a.raw) {
                const int nv = nfull + ((lane < npart) ? 1 : 0);
                flags = nv >= B ? (1u << B) - 1u : (1u << nv) - 1u;
            }
#ifdef TPS_EMU
            redo_keep[tid][0] = flags;
            for (int t = 0; t < B / 2; ++t) redo_keep[tid][1 + t] = present[t];
#else
            redo_flags = flags;
            TPS_UNROLL
            for (int t = 0; t < B / 2; ++t) redo_present[t] = present[t];
#endif
        }
    }
    if (SO || (RAW && a.raw)) {
        TPS_SYNC();
        uint32_t* queue = l.XPC;                       // up to NT * B entries
        uint32_t* occ = l.XPC + NT * B;                // [P <= 15][4] occurrence bits of one window, + the correction
        TPS_PHASE {
            uint32_t flags;
            uint32_t present[B / 2];
#ifdef TPS_EMU
            flags = redo_keep[tid][0];
            for (int t = 0; t < B / 2; ++t) present[t] = redo_keep[tid][1 + t];
#else
            flags = redo_flags;
            TPS_UNROLL
            for (int t = 0; t < B / 2; ++t) present[t] = redo_present[t];
#endif
            while (flags) {
                const int u = ffs0(flags);
                flags &= flags - 1;
                uint32_t pm = 0;
                TPS_UNROLL
                for (int t = 0; t < B / 2; ++t)
                    if (t == (u >> 1)) pm = present[t];
                pm = (pm >> (16 * (u & 1))) & pat.all_mask;
                const uint32_t e = lds_add_ret(&l.misc[M_NTIE], 1u);
                queue[e] = (uint32_t)(u * NT + tid) | (pm << 16);
            }
            occ[tid] = 0;
        }
        TPS_SYNC();
        const int n_redo = (int)uniform(l.misc[M_NTIE]);
        uint16_t* out = tc.sw16 + w0;
        if (n_redo > 0 && !(RAW && a.raw) && n_redo <= COOP_MAX && a.lw <= 128) {
            // few windows: the whole wave recounts one window at a time -- lanes look up the window's positions
            // and publish per-pattern occurrence bits, one lane per pattern walks its bits greedily
            const bool inv = INV;
            for (int e = 0; e < n_redo; ++e) {
                const uint32_t ent = uniform(queue[e]);
                const int wl = (int)(ent & 0xFFFFu);
                const uint32_t redo = (ent >> 16) & pat.so_mask;
                const int q0 = delta + wl * S;
                TPS_PHASE {
                    for (int p = tid; p < a.lw; p += NT) {
                        uint32_t h = h_at(l.lut, l.lshift, l.seq2, l.val, pat, q0 + p, inv) & redo;
                        while (h) {
                            const int b = ffs0(h);
                            h &= h - 1;
                            lds_or(&occ[4 * b + (p >> 5)], 1u << (p & 31));
                        }
                    }
                }
                TPS_SYNC();
                TPS_PHASE {
                    if (tid < pat.P && ((redo >> tid) & 1u)) {
                        int n_occ = 0, greedy = 0, cursor = 0;
                        TPS_UNROLL
                        for (int wd = 0; wd < 4; ++wd) {
                            uint32_t m = occ[4 * tid + wd];
                            n_occ += popc(m);
                            while (m) {
                                const int p = 32 * wd + ffs0(m);
                                m &= m - 1;
                                if (p >= cursor) { ++greedy; cursor = p + pat.k; }
                            }
                            occ[4 * tid + wd] = 0;
                        }
                        if (n_occ != greedy) lds_add(&occ[60], (uint32_t)(n_occ - greedy));
                    }
                }
                TPS_SYNC();
                TPS_PHASE {
                    if (tid == 0) {
                        const uint32_t sw = l.row[padded(wl, LOG2B)] - occ[60];
                        l.row[padded(wl, LOG2B)] = sw;
                        out[wl] = (uint16_t)sw;
                        occ[60] = 0;
                    }
                }
                TPS_SYNC();
            }
        } else if (n_redo > 0) {
            // many windows (or raw counts): every lane recounts its share of the queue sequentially
            TPS_PHASE {
                for (int e = tid; e < n_redo; e += NT) {
                    const uint32_t ent = queue[e];
                    const int wl = (int)(ent & 0xFFFFu);
                    uint8_t* raw_row = (RAW && a.raw) ? a.raw + (out_base + w0 + wl) * (int64_t)pat.P : nullptr;
                    const uint32_t sw = window_exact(a, l, delta, wl, ent >> 16, l.row[padded(wl, LOG2B)], raw_row);
                    l.row[padded(wl, LOG2B)] = sw;
                    out[wl] = (uint16_t)sw;
                }
            }
        }
    }
    TPS_SYNC();
    if (w0 == 0) TPS_STAMP(11);
    const uint32_t gsum = wg_exclusive_scan(l.row, NT * B, &l.misc[M_SCAN], LOG2B);
    if (w0 == 0) TPS_STAMP(12);
    {
        // change-point candidates of this tile: c with w0 <= c * jump < w0 + nw_tile, 64 per pass
        const uint32_t jump = tc.jump;
        const uint32_t c_lo = div_jump((uint32_t)w0 + jump - 1u, tc.jump_magic);
        uint32_t c_hi = div_jump((uint32_t)(w0 + nw_tile) + jump - 1u, tc.jump_magic);
        if (c_hi > tc.lc_cap) c_hi = tc.lc_cap;
        const int passes = c_hi > c_lo ? (int)((c_hi - c_lo + NT - 1) / NT) : 0;
        TPS_PHASE {
            const uint32_t carry = (uint32_t)s_total;
            if (tid == 0) { l.misc[M_INVALID] = 0; l.misc[M_NTIE] = 0; }      // for the next tile's staging
            if (tc.lc16 && tid == 0) l.Tc[tile] = carry;
            uint32_t c = c_lo + (uint32_t)tid;
            uint32_t w = c * jump - (uint32_t)w0;         // tile-local window index of candidate c
            TPS_NOVEC
            for (int t = 0; t < passes; ++t) {
                if (c < c_hi) {
                    const uint32_t pre = l.row[w + (w >> LOG2B)];
                    if (tc.lc16) {
                        if (tc.lc_g) g16_store(tc.lc_g, c, pre);
                        else l.Lc16[c] = (uint16_t)pre;
                    } else if (tc.lc_g) {
                        g32_store(tc.lc_g, c, carry + pre);
                    } else {
                        l.Lc[c] = carry + pre;
                    }
                }
                c += NT;
                w += NT * jump;
            }
        }
    }
    s_total += gsum;
    TPS_SYNC();
}

// After the window phase of a fused tile (row[] = S_w of the tile's windows, padded layout, 0 beyond nw_tile):
// exclusive scan of row[] and the left sums of the tile's change-point candidates (phase 3 of tile_fused_s).
TPS_DEV void tile_candidates(const TileConst& tc, const Lds& l, int w0, int tile, int nw_tile, uint64_t& s_total) {
    constexpr int B = 8, LOG2B = 3;
    const uint32_t gsum = wg_exclusive_scan(l.row, NT * B, &l.misc[M_SCAN], LOG2B);
    const uint32_t jump = tc.jump;
    const uint32_t c_lo = div_jump((uint32_t)w0 + jump - 1u, tc.jump_magic);
    uint32_t c_hi = div_jump((uint32_t)(w0 + nw_tile) + jump - 1u, tc.jump_magic);
    if (c_hi > tc.lc_cap) c_hi = tc.lc_cap;
    const int passes = c_hi > c_lo ? (int)((c_hi - c_lo + NT - 1) / NT) : 0;
    TPS_PHASE {
        const uint32_t carry = (uint32_t)s_total;
        if (tid == 0) { l.misc[M_INVALID] = 0; l.misc[M_NTIE] = 0; }      // for the next tile's staging
        if (tc.lc16 && tid == 0) l.Tc[tile] = carry;
        uint32_t c = c_lo + (uint32_t)tid;
        uint32_t w = c * jump - (uint32_t)w0;
        TPS_NOVEC
        for (int t = 0; t < passes; ++t) {
            if (c < c_hi) {
                const uint32_t pre = l.row[w + (w >> LOG2B)];
                if (tc.lc16) {
                    if (tc.lc_g) g16_store(tc.lc_g, c, pre);
                    else l.Lc16[c] = (uint16_t)pre;
                } else if (tc.lc_g) {
                    g32_store(tc.lc_g, c, carry + pre);
                } else {
                    l.Lc[c] = carry + pre;
                }
            }
            c += NT;
            w += NT * jump;
        }
    }
    s_total += gsum;
    TPS_SYNC();
}

// ------------------------------------------------------------------ step 2, the default tile (round 3)
// Tables without self-overlapping k-mers, sums only (kernels _s5 .. _s8, _s5p .. _s8p): the same published words as
// tile_fused_s, but the windows are computed LANE-CONTIGUOUS -- lane L owns windows 8 L .. 8 L + 7, the ones that start at
// its own blocks -- so everything about a window's start side stays in the lane's registers:
//   phase 1  (lane-contiguous) as tile_fused_s; per block XPC[b] (prefix-OR | count r positions in) and per lane XF go to
//            LDS, the suffix words XS[b] stay in registers
//   phase 2  (lane-contiguous) window 8 L + j = XS[j] (registers) | whole lanes in between (2 - 3 neighbours' XF words,
//            summed in registers) | XPC[8 L + j + q] (one LDS read per window, conflict-free: stride 9 words between
//            lanes).  The lane's 8 S_w are added up in registers, a DPP scan over the 64 lane totals gives every window's
//            exclusive prefix inside the tile; S_w leaves for HBM as 32 contiguous bytes per lane, the prefixes go to row[]
//   phase 3  (candidate-strided) as before: Lc[c] = carry + prefix of window c * jump
// Against tile_fused_s per tile and lane: 17 LDS stores instead of 35 (no XS, no rewritten XF / XT, no second copy of
// S_w for the scan), 13 + the table gathers LDS loads instead of 37 + the gathers, three wave barriers instead of six,
// two 16-byte stores to HBM instead of eight dword stores.
#ifndef TPS_PP_RPT
#define TPS_PP_RPT 1        // (0: A/B builds -- the per-pattern tiles read the window's partial-block position at run time, as before round 5)
#endif
#ifndef TPS_PP_PAIRF
#define TPS_PP_PAIRF 1      // (0: A/B builds -- _s6r looks every position up by itself although the pair table of fields is loaded)
#endif
#ifndef TPS_XPAD
#define TPS_XPAD 1          // (0: the default kernels keep XF / XT arrays in the exchange region -- A/B builds, with TPS_NO_XPAD=1 in the environment)
#endif
#ifndef TPS_LC_TILE
#define TPS_LC_TILE 1
#endif
// Cache policy of the kernels' big output streams (S_w, raw rows: buffer stores): 2 = nt, non-temporal (gfx940+), 0 = default.
// Nothing on the device reads these bytes again (the exact change-point tournament aside); streamed past the caches they cost
// less HBM time: 10 000 x 25 kb reads with raw rows 192.7 -> 176.8 us per launch, config 2 57.4 -> 56.3 us (same box, A/B).
#ifndef TPS_STORE_AUX
#define TPS_STORE_AUX 2
#endif
// the lane's 8 window sums -> tile_out[8 lane .. 8 lane + 7] as 16-bit values, windows at or past nw_tile dropped
#ifdef TPS_EMU
TPS_DEV void g_store_sw8(uint16_t* tile_out, int lane, int nw_tile, const uint32_t* v) {
    for (int i = 0; i < 8; ++i)
        if (8 * lane + i < nw_tile) tile_out[8 * lane + i] = (uint16_t)v[i];
}
#else
TPS_DEV void g_store_sw8(uint16_t* tile_out, int lane, int nw_tile, const uint32_t* v) {
    // One buffer_store_dwordx4 (16 contiguous bytes per lane) through a raw buffer descriptor that ends behind the tile's last
    // window: the hardware range-checks every dword of a multi-dword store on its own and drops the ones past the end (GCN3 /
    // Vega ISA, "range checking": raw buffers, store_dword_x{2,3,4} per component) -- the lane that holds the tile's last windows
    // needs no exec masking and no scalar fallback.  A dword is two windows: tiles start at even windows (plan_geometry keeps
    // the windows per tile even, a read's region starts at a multiple of 8), and the odd last window of a read takes the padding
    // slot behind it along.  The descriptor is wave-uniform (SGPRs only).
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tile_out, 0, ((nw_tile + 1) & ~1) * 2, 0x00020000);
    v4u t;
    t.x = v[0] | (v[1] << 16); t.y = v[2] | (v[3] << 16); t.z = v[4] | (v[5] << 16); t.w = v[6] | (v[7] << 16);      // (v_lshl_or_b32; every S_w < 2^16)
    __builtin_amdgcn_raw_buffer_store_b128(t, rs, lane * 16, 0, TPS_STORE_AUX);
}
#endif
// 16 bytes -> base[cdw .. cdw + 3] (dwords), the dwords at or past n_dw dropped: one range-checked buffer_store_dwordx4 at a
// dword-aligned address instead of four dword stores and a tail case
#ifdef TPS_EMU
TPS_DEV void g_store16_clamped(uint32_t* base, int n_dw, int cdw, const u32x4& t) {
    const uint32_t v[4] = {t.x, t.y, t.z, t.w};
    for (int i = 0; i < 4; ++i)
        if (cdw + i < n_dw) base[cdw + i] = v[i];
}
#else
TPS_DEV void g_store16_clamped(uint32_t* base, int n_dw, int cdw, const u32x4& t) {
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, n_dw * 4, 0x00020000);
    v4u x;
    x.x = t.x; x.y = t.y; x.z = t.z; x.w = t.w;
#ifndef TPS_NO_RAW_STORE                              /* (diagnostic builds only: what do the raw-row stores cost?) */
    __builtin_amdgcn_raw_buffer_store_b128(x, rs, cdw * 4, 0, TPS_STORE_AUX);
#endif
}
#endif
// ROTZ: q (the window's whole blocks) is a multiple of 8 -- every window ends in the same block-in-lane it starts in, dl0
// lanes on: no per-window choice between the near and the far end lane (the default geometry: W = 100, k = 4, slide 6).
// CD > 0: the table's self-overlapping k-mers share the ONE period CD (2 CD >= k), sums only.  re.finditer counts leftmost
// non-overlapping occurrences (allsteps.py:281): along a CHAIN of occurrences CD apart it takes every other one, restarting
// at the window's first.  The tile counts every occurrence as a plain tile does and takes the difference back afterwards: a
// window that holds n consecutive elements of a chain counted n, finditer counts ceil(n / 2) -- the correction floor(n / 2)
// is piecewise constant in the window index, so the lane that finds a chain's HEAD (an occurrence at p and at p + CD, none
// at p - CD: one AND per position on table entries that are in registers anyway) adds its steps to a difference array in
// LDS (row[]: free until the prefixes are written) -- for a pair, the usual case (one deleted base makes one), +1 at the
// first window that holds both and -1 behind the last.  Tiles with a chain (most telomeric tiles at k = 6, few elsewhere)
// then prefix-sum the array and subtract.  This replaces the canonical-pick tile of round 2 (tile_so_s: one dependent pick
// per position in every lane of every chained tile, 2.3 x a plain tile) and its chain-parity repairs.
template <int S, bool INV, int RPT, bool PAIR, bool ROTZ, int CD_ = 0, bool P16_ = false>
TPS_DEV void tile_lc_s(const ScanArgs& a, const TileConst& tc, const Lds& l, int delta, int w0, int tile, int nw_tile,
                       int64_t out_base, uint64_t& s_total, int64_t r) {
    static_assert(!(PAIR && INV), "pair lookups need per-position independence");
    constexpr int CD = CD_;
    static_assert(CD == 0 || (!PAIR && !INV && CD <= S), "chain corrections: single lookups on clean tiles, period <= slide");
    static_assert(!(P16_ && CD_ > 0), "the 16-bit pair table is for tables without self-overlap");
    constexpr bool RZ = RPT == 0;
    // M16: the kernels that run the chain-corrected tiles keep their LDS table as 16-bit pattern masks (ScanArgs::lut16): a block's
    // OR is the masks' OR, its match count their popcounts added up (v_bcnt_u32_b32: one instruction, like the add it replaces);
    // the published words keep the layout mask << 16 | count
    // (P16_: the pair-table kernels of k = 5 tables -- pair and single table both in that format, ScanArgs::pair16)
    constexpr bool M16 = CD > 0 || P16_;
    constexpr int LS = M16 ? 1 : 2;               // log2(bytes per table entry)
    // XPAD (the default kernels, round 4): a lane's OR | matches (XF) and its exclusive prefix in the tile (XT) live in the pad words of
    // its XPC and row[] rows (9 words per lane, the ninth unused) -- no XF / XT arrays in the wave's slice (carve_fused<.., XM = 2>)
    // (round 5: the chain-corrected tiles too -- nothing of theirs touches the pad words: the window phase and the difference array
    // index row[] / XPC by PADDED window, the zeroing of the difference array comes before xt_at is written -- which frees the 80
    // dwords of XF per wave that stood between the k = 6 sums kernel and three 8-wave workgroups per CU: 6 waves per SIMD)
    constexpr bool XPAD = TPS_XPAD != 0;
    typedef Geo<S> g_;
    auto xf_at = [&](int lane_) -> uint32_t& { return XPAD ? l.XPC[lane_ * (g_::B + 1) + g_::B] : l.XF[lane_]; };
    auto xt_at = [&](int lane_) -> uint32_t& { return XPAD ? l.row[lane_ * (g_::B + 1) + g_::B] : l.XT[lane_]; };
    constexpr int B = g_::B, LOG2B = g_::LOG2B, POS = g_::POS;
    constexpr int LB = CD;                        // CD: the lane's registers start CD positions before its first one (look-back of the pair test)
    constexpr int WDW = (LB + POS + 13 + 15) / 16;
    static_assert(CD > 0 || WDW == g_::WDW, "the plain tiles keep Geo's register footprint");
    const PatInfo& pat = a.pat;
    int rp = RPT >= 0 ? RPT : tc.r, q = tc.q;
    // opaque per tile: what derives from q (the far-end choice of each of a lane's 8 windows, the lane distances) is recomputed on
    // the scalar unit per tile instead of being hoisted out of the tile loop and kept in -- spilled -- SGPR pairs across the read
    TPS_PIN_S(q);
    if (RPT < 0) TPS_PIN_S(rp);
    const uint32_t amask = pat.kmask << LS;
    // what a lane keeps about its blocks (table entries are mask << 16 | count: the OR of entries is right in its high half,
    // their sum in its low half -- the other halves are garbage that the window arithmetic never looks at)
    uint32_t sfx[B], c0s[B], xf_own = 0;         // OR of this and the lane's later blocks; matches before the block; the lane's OR | matches
#ifdef TPS_EMU
    uint32_t sfx_keep[NT][B], c0_keep[NT][B], xf_keep[NT];
#endif
    uint32_t lane_chain = 0;                      // CD: this lane found the head of a chain of three or more (it added steps to the difference array)
    uint32_t pair_lo = 0, pair_hi = 0;            // CD: bit e = some pattern occurs at the lane's position e and CD before it (a pair, filed under its SECOND element)
#ifdef TPS_EMU
    uint32_t pair_keep[NT][2];
#endif
    // (CD) a chain of three or more occurrences CD apart, first element at tile position tp (it may lie before the tile), pattern
    // mask pm: the pair count takes n - 1 off a window that holds n consecutive elements, finditer counts ceil(n / 2) = n - (n - 1)
    // + floor((n - 1) / 2) -- the steps of floor((n - 1) / 2), window by window, go to the difference array (any length, exact)
    // One chain at a time, its windows spread over the wave's lanes (round 4: the lane that found a chain used to walk its ~20
    // windows alone, ~400 instructions with 63 lanes masked off -- two such chains per telomeric tile at k = 6 and ONT error rates
    // made that tile's first phase twice as long as any other's): chain_len() is wave-uniform, chain_window() is lane `ln`'s share.
    auto chain_len = [&](int tp, uint32_t pm) -> int {
        constexpr int CD = CD_ > 0 ? CD_ : 1;     // (never called for CD_ = 0; keeps the divisions below well-formed)
        auto occ = [&](int pos) -> bool { return uniform(lut_mask(l.lut, l.lshift, pat, v_at(l.seq2, delta + pos)) & pm) != 0u; };
        const int tend = NT * POS;                                           // positions of the tile
        int c = 3;
        while (tp + c * CD < tend && occ(tp + c * CD)) ++c;
        return c;
    };
    auto chain_windows = [&](int tp, int c, int ln) {
        constexpr int CD = CD_ > 0 ? CD_ : 1;
        const int lw = a.lw;
        auto w_in = [&](int e) -> int { const int x = e - lw + 1; return x <= 0 ? 0 : (x + S - 1) / S; };     // first window that holds position e
        auto cur_of = [&](int w) -> int {
            // chain elements j with w S <= tp + j CD < w S + lw
            const int a0 = w * S - tp, a1 = w * S + lw - 1 - tp;
            int jlo = a0 <= 0 ? 0 : (a0 + CD - 1) / CD, jhi = a1 < 0 ? -1 : a1 / CD;
            if (jhi > c - 1) jhi = c - 1;
            return jhi > jlo ? (jhi - jlo) >> 1 : 0;
        };
        const int wf = w_in(tp < 0 ? 0 : tp);
        int wb = (tp + (c - 1) * CD) / S + 1;     // the first window behind the chain: the last step (back to 0)
        if (wb > NT * B - 1) wb = NT * B - 1;     // (steps at or behind the tile's last window slot are nobody's)
        for (int w = wf + ln; w <= wb; w += NT) {
            const int cur = cur_of(w), prev = w == wf ? 0 : cur_of(w - 1);
            if (cur != prev) lds_add(&l.row[w + (w >> LOG2B)], (uint32_t)(cur - prev));
        }
    };
    TPS_PHASE {
        const int span = tid;
        const int p0 = delta + span * POS;        // >= 16: fused tiles are staged behind SEQ_LEAD words
        // the lane's bases shifted so that position p's k-mer code sits LS bits above bit 2 p: code << LS is the table's byte offset
        const int bo = 2 * (p0 - LB) - LS;        // (CD: from LB positions before the lane's first one)
        const uint32_t sh2 = (uint32_t)(bo & 31);
        const int d0 = bo >> 5;
        uint32_t w[WDW];
        {
            uint32_t prev = l.seq2[d0];
            TPS_UNROLL
            for (int i = 0; i < WDW; ++i) {
                uint32_t nx = l.seq2[d0 + i + 1];
                w[i] = alignbit(nx, prev, sh2);
                prev = nx;
            }
        }
        auto v4_at = [&](int p) -> uint32_t {     // the base before position p at bit 0 (p constant after unrolling; p >= -LB)
            const int dw = (p + LB) >> 4, bit = (p + LB) & 15;
            return bit ? alignbit(dw + 1 < WDW ? w[dw + 1] : 0u, w[dw], 2u * bit) : w[dw];
        };
        uint32_t cnt = 0, run_or = 0;
        uint32_t gs[B];
        pair_lo = pair_hi = 0;                    // (per lane)
        uint32_t* xpc = l.XPC + span * (B + 1);
        if constexpr (PAIR) {
            constexpr int NP = S / 2, NH = NP + (S & 1);
            const uint32_t amask2 = M16 ? ((pat.kmask << 3) | 0x6u) : ((pat.kmask << 4) | 0xCu);     // the (k+1)-mer's code << LS
            const int rpe = rp & ~1;
            uint32_t hc[NH], hn[NH];
            auto fetch = [&](int blk, uint32_t* hh) {
                TPS_UNROLL
                for (int j = 0; j < NP; ++j) hh[j] = M16 ? lut16_at(l.lut2, v4_at(blk * S + 2 * j), amask2) : lut_at_tile(l.lut2, v4_at(blk * S + 2 * j), amask2);
                if (S & 1) hh[NP] = M16 ? lut16_at(l.lut, v4_at(blk * S + S - 1), amask) : lut_at_tile(l.lut, v4_at(blk * S + S - 1), amask);
            };
            fetch(0, hc);
            TPS_UNROLL
            for (int blk = 0; blk < B; ++blk) {
                if (blk + 1 < B) fetch(blk + 1, hn);
                uint32_t g = 0;
                c0s[blk] = cnt;
                if (RZ) xpc[blk] = M16 ? ((run_or << 16) | cnt) : pack_hi_lo(run_or, cnt);
                TPS_UNROLL
                for (int j = 0; j < NH; ++j) {
                    if (!RZ) {
                        if (2 * j == rpe) {
                            uint32_t c1 = cnt, pp = run_or | g;
                            if (rp & 1) {
                                const uint32_t h1 = M16 ? lut16_at(l.lut, v4_at(blk * S + 2 * j), amask) : lut_at(l.lut, v4_at(blk * S + 2 * j), amask);
                                c1 = M16 ? c1 + (uint32_t)popc(h1) : c1 + h1;
                                pp |= h1;
                            }
                            xpc[blk] = M16 ? ((pp << 16) | c1) : pack_hi_lo(pp, c1);
                        }
                    }
                    g |= hc[j];
                    cnt = M16 ? cnt + (uint32_t)popc(hc[j]) : cnt + hc[j];     // (a pair entry: the two positions' masks ORed -- never the same pattern twice)
                }
                gs[blk] = g;
                run_or |= g;
                TPS_UNROLL
                for (int j = 0; j < NH; ++j) hc[j] = hn[j];
            }
        } else {
            uint32_t hc[S], hn[S];
            auto fetch = [&](int blk, uint32_t* hh, int cnt_) {
                TPS_UNROLL
                for (int i = 0; i < S; ++i) {
                    if (i < cnt_) {
                        const int p = blk * S + i;
                        uint32_t h = M16 ? lut16_at_wide(l.lut, v4_at(p), amask) : lut_at(l.lut, v4_at(p), amask);
                        if (INV) {
                            if (h && invalid_at(l.val, p0 + p, pat.k)) h = 0;   // tiles with non-ACGT letters only
                        }
                        hh[i] = h;
                    } else {
                        hh[i] = 0;
                    }
                }
            };
            fetch(0, hc, S);
            constexpr int NB_ = CD > 0 ? CD : 1;
            uint32_t hb[NB_];                             // CD: the entries of the CD positions before the block
            if constexpr (CD > 0) {
                TPS_UNROLL
                for (int i = 0; i < CD; ++i) hb[i] = lut16_at_wide(l.lut, v4_at(i - CD), amask);
            }
            TPS_UNROLL
            for (int blk = 0; blk < B; ++blk) {
                if (blk + 1 < B) fetch(blk + 1, hn, S);
                if constexpr (CD > 0) {
                    uint32_t cfb = 0;                     // some pattern occurs at a position of this block and CD before it
                    TPS_UNROLL
                    for (int i = 0; i < S; ++i) cfb |= hc[i] & (i >= CD ? hc[i - CD] : hb[i]);
                    if (cfb != 0u) {                      // a deleted or inserted base inside a telomeric stretch
                        // remember WHERE (one bit per position of the lane): everything about pairs and chains happens behind
                        // the block loop, on these bits
                        TPS_UNROLL
                        for (int i = 0; i < S; ++i) {
                            const uint32_t pr = (hc[i] & (i >= CD ? hc[i - CD] : hb[i])) ? 1u : 0u;
                            if (blk * S + i < 32) pair_lo |= pr << ((blk * S + i) & 31);
                            else pair_hi |= pr << ((blk * S + i) & 31);
                        }
                    }
                    TPS_UNROLL
                    for (int i = 0; i < CD; ++i) hb[i] = hc[S - CD + i];
                }
                uint32_t g = 0;
                c0s[blk] = cnt;
                uint32_t c1 = cnt, pp = run_or;
                // CD >= S - 2 (slide 6: k = 6 with CD = 5, k = 5 with CD = 4): two occurrences of a pattern are CD apart (the table's one
                // period) or at least k > CD, so a block of S <= CD + 2 positions holds a pattern at most twice -- CD apart, or at its
                // positions 0 and S - 1 -- and every position matches at most one pattern (no duplicate k-mers in these tables): the
                // block's matches are the popcount of its OR plus those few possible pairs (whose ANDs the pair test below needs
                // anyway), instead of a popcount and an add per position
                constexpr bool CNT_OR = CD > 0 && CD >= S - 2;
                if constexpr (CNT_OR) {
                    uint32_t g_rp = 0, dup = 0, dup_rp = 0;
                    TPS_UNROLL
                    for (int i = 0; i < S; ++i) {
                        g |= hc[i];
                        if (!RZ) {
                            if (i + 1 == rp) g_rp = g;
                        }
                        if (i >= CD) {
                            const uint32_t t = (uint32_t)popc(hc[i] & hc[i - CD]);
                            dup += t;
                            if (!RZ) {
                                if (i < rp) dup_rp += t;
                            }
                        }
                    }
                    if (CD == S - 2) dup += (uint32_t)popc(hc[S - 1] & hc[0]);
                    if (!RZ) { pp = run_or | g_rp; c1 = cnt + (uint32_t)popc(g_rp) + dup_rp; }
                    cnt += (uint32_t)popc(g) + dup;
                } else {
                TPS_UNROLL
                for (int i = 0; i < S; ++i) {
                    const uint32_t h = hc[i];
                    g |= h;
                    cnt = M16 ? cnt + (uint32_t)popc(h) : cnt + h;
                    if (!RZ) {
                        if (i + 1 == rp) { c1 = cnt; pp = run_or | g; }
                    }
                }
                }
                xpc[blk] = M16 ? ((pp << 16) | c1) : pack_hi_lo(pp, c1);
                gs[blk] = g;
                run_or |= g;
                TPS_UNROLL
                for (int i = 0; i < S; ++i) hc[i] = hn[i];
            }
        }
        uint32_t sf = 0;
        TPS_UNROLL
        for (int j = B - 1; j >= 0; --j) {
            sf = M16 ? ((gs[j] << 16) | sf) : (sf | gs[j]);      // (the window phase wants the masks in the high half)
            sfx[j] = sf;
        }
        xf_own = M16 ? (sf | cnt) : pack_hi_lo(sf, cnt);
        xf_at(span) = xf_own;
        if constexpr (CD > 0) { TPS_PIN_V(pair_lo); TPS_PIN_V(pair_hi); }
#ifdef TPS_EMU
        for (int j = 0; j < B; ++j) { sfx_keep[tid][j] = sfx[j]; c0_keep[tid][j] = c0s[j]; }
        xf_keep[tid] = xf_own;
        pair_keep[tid][0] = pair_lo; pair_keep[tid][1] = pair_hi;
#endif
    }
    if constexpr (CD > 0) {
        // Pairs (round 4).  A pair -- the same pattern at e - CD and e, filed under e -- lies inside a window iff
        // window start + CD <= e < window end: the END is the matches' own, so the pairs ride in the count half of the published
        // words as a second field (matches in bits 0 .. 7, pairs in bits 8 .. 15: the window phase's differences are exact modulo 2^16
        // as long as both results fit their fields -- a window has at most lw <= 255 of either, plan_geometry checks), and the START
        // side is the owning lane's registers:
        // c0s[j] also takes the pairs before position CD of block j.  A window then counts matches - pairs: exact for chains of
        // two, the usual case (one deleted base inside a telomeric stretch makes one); chains of three or more get the
        // difference back through the difference array (long_chain_steps).  Everything here is bit arithmetic on the lanes'
        // 64-bit pair masks -- no walk over the bases, no table lookups -- and only runs in tiles that hold a pair at all.
        // (Round 3 walked every chain from its head: five dependent LDS round trips and two atomics per pair, ~30 pairs per
        // telomeric tile at ONT error rates -- a third of the k = 6 kernel's instructions.)
        constexpr int PS = 8;
#ifdef TPS_EMU
        bool tile_pairs = false;
        for (int t = 0; t < NT; ++t) tile_pairs = tile_pairs || (pair_keep[t][0] | pair_keep[t][1]) != 0u;
#else
        const bool tile_pairs = __builtin_amdgcn_ballot_w64((pair_lo | pair_hi) != 0u) != 0;
#endif
        if (tile_pairs) {
            TPS_SYNC();                               // (every lane's words are published: the atomics below add to them)
            uint32_t head3_lo = 0, head3_hi = 0;      // second elements of the first pair of a chain of three or more
#ifdef TPS_EMU
            uint32_t head3_keep[NT][2];
#endif
            TPS_PHASE {
#ifdef TPS_EMU
                pair_lo = pair_keep[tid][0]; pair_hi = pair_keep[tid][1];
                for (int j = 0; j < B; ++j) c0s[j] = c0_keep[tid][j];
                xf_own = xf_keep[tid];
#endif
                const int lane = tid;
                auto below = [&](int n) -> uint32_t {     // pairs at positions < n (n constant after unrolling, 0 <= n <= 64)
                    if (n <= 0) return 0u;
                    if (n >= 64) return (uint32_t)(popc(pair_lo) + popc(pair_hi));
                    if (n <= 32) return (uint32_t)popc(n == 32 ? pair_lo : (pair_lo & ((1u << n) - 1u)));
                    return (uint32_t)(popc(pair_lo) + popc(pair_hi & ((1u << (n - 32)) - 1u)));
                };
                if ((pair_lo | pair_hi) != 0u) {
                    uint32_t* xpc = l.XPC + lane * (B + 1);
                    TPS_UNROLL
                    for (int blk = 0; blk < B; ++blk) {
                        const uint32_t pe = below(blk * S + (RZ ? 0 : RPT));      // (RPT is a compile-time constant in these tiles)
                        if (pe) lds_add(&xpc[blk], pe << PS);
                        c0s[blk] += below(blk * S + CD) << PS;
                    }
                    const uint32_t pt = below(64) << PS;
                    xf_own += pt;
                    lds_add(&xf_at(lane), pt);
                }
                // chains of three or more: pair bits CD apart.  The neighbours' bits close the lane's ends (nothing before the
                // tile's first lane: what lies there is in none of its windows; nothing behind the last: nor is that)
                uint32_t prev_hi, next_lo;
#ifdef TPS_EMU
                prev_hi = lane > 0 ? pair_keep[lane - 1][1] : 0u;
                const uint32_t prev_lo_ = lane > 0 ? pair_keep[lane - 1][0] : 0u;
                next_lo = lane + 1 < NT ? pair_keep[lane + 1][0] : 0u;
#else
                prev_hi = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + NT - 1) & (NT - 1)) << 2, (int)pair_hi);
                const uint32_t prev_lo_ = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + NT - 1) & (NT - 1)) << 2, (int)pair_lo);
                next_lo = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + 1) & (NT - 1)) << 2, (int)pair_lo);
                if (lane == 0) prev_hi = 0u;
                if (lane == NT - 1) next_lo = 0u;
#endif
                const uint64_t pm64 = ((uint64_t)pair_hi << 32) | pair_lo;
                const uint64_t pv64 = lane > 0 ? (((uint64_t)prev_hi << 32) | prev_lo_) : 0ull;
                // bit e of `before`: a pair at e - CD (the previous lane's last CD positions for e < CD); of `after`: one at e + CD
                const uint64_t before = (pm64 << CD) | (pv64 >> (POS - CD));
                const uint64_t after = (pm64 >> CD) | ((uint64_t)(next_lo & ((1u << CD) - 1u)) << (POS - CD));
                const uint64_t lim = POS >= 64 ? ~0ull : ((1ull << POS) - 1ull);
                const uint64_t h3 = pm64 & ~before & after & lim;
                head3_lo = (uint32_t)h3; head3_hi = (uint32_t)(h3 >> 32);
#ifdef TPS_EMU
                for (int j = 0; j < B; ++j) c0_keep[tid][j] = c0s[j];
                xf_keep[tid] = xf_own;
                head3_keep[tid][0] = head3_lo; head3_keep[tid][1] = head3_hi;
#endif
            }
#ifdef TPS_EMU
            bool tile_long = false;
            for (int t = 0; t < NT; ++t) tile_long = tile_long || (head3_keep[t][0] | head3_keep[t][1]) != 0u;
#else
            const bool tile_long = __builtin_amdgcn_ballot_w64((head3_lo | head3_hi) != 0u) != 0;
#endif
            if (tile_long) {
                TPS_PHASE {
                    TPS_UNROLL
                    for (int i = 0; i < B + 1; ++i) l.row[tid + i * NT] = 0;      // the difference array, indexed by padded window
                }
                TPS_SYNC();
                // every chain in turn (owner lane by owner lane, bit by bit: all of it wave-uniform), its windows lane-parallel
                auto one_owner = [&](int src, uint32_t h_lo, uint32_t h_hi, auto&& each_lane) {
                    while (h_lo | h_hi) {
                        int e;
                        if (h_lo) { e = ffs0(h_lo); h_lo &= h_lo - 1u; }
                        else { e = 32 + ffs0(h_hi); h_hi &= h_hi - 1u; }
                        const int te = src * POS + e;                             // tile position of the chain's SECOND element
                        const uint32_t pm = uniform(lut_mask(l.lut, l.lshift, pat, v_at(l.seq2, delta + te)));
                        const int c = chain_len(te - CD, pm);
                        each_lane(te - CD, c);
                    }
                };
#ifdef TPS_EMU
                for (int src = 0; src < NT; ++src)
                    one_owner(src, head3_keep[src][0], head3_keep[src][1], [&](int tp, int c) { for (int ln = 0; ln < NT; ++ln) chain_windows(tp, c, ln); });
#else
                {
                    const int ln = tps_fresh_lane();
                    uint64_t owners = __builtin_amdgcn_ballot_w64((head3_lo | head3_hi) != 0u);
                    while (owners) {
                        const int src = __builtin_ctzll(owners);
                        owners &= owners - 1ull;
                        const uint32_t h_lo = (uint32_t)__builtin_amdgcn_readlane((int)head3_lo, src), h_hi = (uint32_t)__builtin_amdgcn_readlane((int)head3_hi, src);
                        one_owner(src, h_lo, h_hi, [&](int tp, int c) { chain_windows(tp, c, ln); });
                    }
                }
#endif
                lane_chain = 1u;                      // (uniform: some lane added steps)
            }
        }
    }
    TPS_SYNC();
    if (w0 == 0) TPS_STAMP(6);
    const int rot = ROTZ ? 0 : (q & (B - 1)), dl0 = q >> LOG2B;
    const int brk = B - rot;                      // windows j >= brk end one lane further on
    uint32_t sw[B], ltot = 0;
#ifdef TPS_EMU
    uint32_t sw_keep[NT][B], tot_keep[NT];
#endif
    TPS_PHASE {
#ifdef TPS_EMU
        for (int j = 0; j < B; ++j) { sfx[j] = sfx_keep[tid][j]; c0s[j] = c0_keep[tid][j]; }
        xf_own = xf_keep[tid];
#endif
        const int lane = tid;
        // the far-end words first: one LDS round trip for the lane's 8 windows (rows past the tile's windows read in-bounds
        // garbage: what they turn into is never stored and never counted)
        const uint32_t* pe = l.XPC + (lane + dl0) * (B + 1) + rot;
        uint32_t ev[B];
        TPS_UNROLL
        for (int j = 0; j < B; ++j) ev[j] = pe[j + ((!ROTZ && j >= brk) ? 1 : 0)];       // (the pad word between two lanes' blocks)
        // whole lanes a window skips: OR of the lanes strictly in between, matches of every lane from its own up to the
        // one before the last -- for both possible lane distances
        uint32_t orw = 0, sumw = xf_own;
        TPS_NOVEC
        for (int t = 1; t < dl0; ++t) {
            const uint32_t v = xf_at(lane + t);
            orw |= v;
            sumw += v;
        }
        uint32_t orb = orw, sumb = sumw;
        if (!ROTZ) {
            const uint32_t vb = xf_at(lane + dl0);
            orb |= vb;
            sumb += vb;
        }
        const uint32_t am = pat.all_mask << 16;
        uint32_t run = 0;
        TPS_UNROLL
        for (int j = 0; j < B; ++j) {
            const bool far_ = !ROTZ && j >= brk;
            const uint32_t e = ev[j], fo = far_ ? orb : orw, fs = far_ ? sumb : sumw;
            const uint32_t m = sfx[j] | e | fo;                 // presence: high halves
            const uint32_t c = (e - c0s[j] + fs) & 0xFFFFu;     // matches: low halves (CD: matches | pairs << 8)
            sw[j] = (CD > 0 ? (c & 0xFFu) - (c >> 8) : c) + (uint32_t)popc(~m & am);
            run += sw[j];
        }
        ltot = run;
#ifdef TPS_EMU
        for (int j = 0; j < B; ++j) sw_keep[tid][j] = sw[j];
        tot_keep[tid] = ltot;
#endif
    }
    if constexpr (CD > 0) {
        // a chain somewhere in the tile: the windows give back what the chains' skipped occurrences added -- the prefix sum of
        // the difference array, lane-contiguous like the windows themselves
#ifdef TPS_EMU
        const bool tile_chain = lane_chain != 0;
        if (tile_chain) {
            ++emu_counter(5);
            uint32_t acc = 0;
            for (int t = 0; t < NT; ++t) {
                uint32_t tot = 0;
                for (int j = 0; j < B; ++j) { acc += l.row[t * (B + 1) + j]; sw_keep[t][j] += acc; tot += sw_keep[t][j]; }
                tot_keep[t] = tot;
            }
        } else {
            ++emu_counter(6);
        }
#else
        const bool tile_chain = lane_chain != 0;          // (uniform)
        if (tile_chain) {
            uint32_t pd[B], dt = 0;
            TPS_PHASE {
                const uint32_t* pr = l.row + tid * (B + 1);
                TPS_UNROLL
                for (int j = 0; j < B; ++j) { dt += pr[j]; pd[j] = dt; }
            }
            uint32_t inc = dt;
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);   // row_shr:1
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);   // row_shr:2
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);   // row_shr:4
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false);   // row_shr:8
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
            const uint32_t dbase = inc - dt;
            uint32_t run = 0;
            TPS_UNROLL
            for (int j = 0; j < B; ++j) { sw[j] += dbase + pd[j]; run += sw[j]; }
            ltot = run;
        }
#endif
    }
    // exclusive scan of the lane totals over the wave (lanes past the tile's last window add garbage behind every valid window)
    uint32_t lexc = 0;
#ifdef TPS_EMU
    uint32_t exc_keep[NT];
    { uint32_t acc = 0; for (int t = 0; t < NT; ++t) { exc_keep[t] = acc; acc += tot_keep[t]; } }
#else
    {
        uint32_t inc = ltot;
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);   // row_shr:1
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);   // row_shr:2
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);   // row_shr:4
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false);   // row_shr:8
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
        inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
        lexc = inc - ltot;
    }
#endif
    TPS_PHASE {
#ifdef TPS_EMU
        for (int j = 0; j < B; ++j) sw[j] = sw_keep[tid][j];
        lexc = exc_keep[tid];
#endif
        const int lane = tid;
#ifndef TPS_NO_SW_STORE                               /* (diagnostic builds only: what do the S_w stores cost?) */
        g_store_sw8(tc.sw16 + w0, lane, nw_tile, sw);
#endif
        // what the candidate phase reads: every window's exclusive prefix inside its LANE (padded layout) and every lane's
        // exclusive prefix inside the tile
        uint32_t* pr = l.row + lane * (B + 1);
        uint32_t run = 0;
        TPS_UNROLL
        for (int j = 0; j < B; ++j) { pr[j] = run; run += sw[j]; }
        xt_at(lane) = lexc;
    }
    TPS_SYNC();
    if (w0 == 0) TPS_STAMP(12);
    // the tile's total = the prefix at its first window that is NOT part of it (nw_tile <= 512 - q - 1: a written entry)
    const uint32_t gsum = uniform(l.row[nw_tile + (nw_tile >> LOG2B)] + xt_at(nw_tile >> LOG2B));
    {
        // change-point candidates of this tile: c with w0 <= c * jump < w0 + nw_tile, 64 per pass
        const uint32_t jump = tc.jump;
        const uint32_t c_lo = div_jump((uint32_t)w0 + jump - 1u, tc.jump_magic);
        uint32_t c_hi = div_jump((uint32_t)(w0 + nw_tile) + jump - 1u, tc.jump_magic);
        if (c_hi > tc.lc_cap) c_hi = tc.lc_cap;
        const int passes = c_hi > c_lo ? (int)((c_hi - c_lo + NT - 1) / NT) : 0;
        TPS_PHASE {
            const uint32_t carry = (uint32_t)s_total;
            if (tid == 0) { l.misc[M_INVALID] = 0; l.misc[M_NTIE] = 0; }      // for the next tile's staging
            if (tc.lc16 && tid == 0) l.Tc[tile] = carry;
            uint32_t c = c_lo + (uint32_t)tid;
            uint32_t w = c * jump - (uint32_t)w0;         // tile-local window index of candidate c
            TPS_NOVEC
            for (int t = 0; t < passes; ++t) {
                if (c < c_hi) {
                    const uint32_t pre = l.row[w + (w >> LOG2B)] + xt_at((int)(w >> LOG2B));
                    if (tc.lc16) {
                        if (tc.lc_g) g16_store(tc.lc_g, c, pre);
                        else l.Lc16[c] = (uint16_t)pre;
                    } else if (tc.lc_g) {
                        g32_store(tc.lc_g, c, carry + pre);
                    } else {
                        l.Lc[c] = carry + pre;
                    }
                }
                c += NT;
                w += NT * jump;
            }
        }
    }
    s_total += gsum;
    TPS_SYNC();
    (void)r;
}

// ------------------------------------------------------------------ step 2, per-pattern tiles
// The exact per-pattern counts c_p of every window (rawCountPattern, allsteps.py:398-411) -- and S_w = sum of
// max(c_p, 1) from them -- without recounting windows, also for tables whose k-mers can overlap themselves.
//
// Counting: a table entry e = 1 << (16 + p) | 1 squares to mulhi(e, e) = 1 << 2p: a one-hot 2-bit field per
// pattern (no duplicate k-mers in the list).  A block (S <= 8 positions, k >= 4) holds at most 2 non-overlapping
// occurrences of a pattern, a lane's 8 blocks at most 14: block counts add up in 2-bit fields, lane-local
// prefixes in nibbles (even / odd patterns in two words), window totals in bytes.
//
// Self-overlap (D > 0: the table's k-mers have exactly one period D < k, so 2 D >= k): re.finditer counts
// leftmost non-overlapping occurrences.  CANONICAL picks are the greedy choice from the head of every chain of
// overlapping occurrences: pick(p) = occ(p) & ~pick(p - D).  A window that does not start inside a chain sees
// exactly the canonical picks among its start positions (cutting a chain at the window's END keeps its head).
// A window whose first occurrence of pattern p is a canonically SKIPPED one (blocked by a pick before the window's
// first position) picks it instead: one more than canonical if the chain ends there -- the block publishes these
// "start skips" and the window adds them -- and only if the chain goes on (three or more chained occurrences,
// e.g. a (CCTAA)n run at k = 6) the window is recounted exactly.
// A lane learns the picks before its first position from a look-back over the previous lane's last 16 positions
// (recursion from "nothing picked"; exact unless one pattern chains through the whole look-back: then the
// windows that touch this lane are recounted).  The first lane of a tile takes the state the previous tile left.
//
//   phase 1  (lane-contiguous) picks of the lane's 8 blocks; publishes per block END[b] = lane-local nibble counts
//            before block b + its first r positions, per lane the nibble totals; keeps V[b] = 14 - counts before
//            block b + start skips (nibbles) and the recount flags in registers
//   phase 2  (lane-contiguous) window 8 L + j = own lane from block j on + whole lanes in between + END[8 L + j + q]:
//            bytes, floored to 1, summed (S_w), transposed into pattern order, stored
//   phase 3  tile_candidates
template <int S>
struct GeoPP {
    static constexpr int B = 8, POS = B * S, LBK = 16;
    // base before the look-back + look-back + positions + chain look-ahead (<= 6) + last k-mer (<= 7 bases)
    static constexpr int WDW = (1 + LBK + POS + 6 + 7 + 15) / 16;
};
// the raw-count kernels load their LDS table as one-hot 2-bit fields whenever the per-pattern tiles will run (plan_geometry
// sets ScanArgs::lut_fields): a position then costs lookup + add, without the v_mul_hi_u32 that squares mask << 16 | 1 into
// 1 << 2 p (p: the pattern's field, pp_field).
constexpr uint32_t PP_BIAS = 14;                  // V nibbles = PP_BIAS - prefix + start skips, all in 0..15
TPS_DEV void pp_expand(uint32_t ne, uint32_t no, uint32_t* b) {   // nibble words (even / odd patterns) -> 4 byte words
    b[0] = ne & 0x0F0F0F0Fu;                      // patterns 0, 4, 8, 12
    b[1] = (ne >> 4) & 0x0F0F0F0Fu;               // patterns 2, 6, 10, 14
    b[2] = no & 0x0F0F0F0Fu;                      // patterns 1, 5, 9, 13
    b[3] = (no >> 4) & 0x0F0F0F0Fu;               // patterns 3, 7, 11, 15
}

// CD > 0 (with D = 0): CHAIN DETECTION on a table whose self-overlapping k-mers share the period CD.  The tile runs as if the
// table had no self-overlap (no look-back, no skips, no repairs) and ANDs the fields CD positions apart on the way; if any
// lane saw a pattern occur at p and again at p + CD inside the tile it returns true right after phase 1 (nothing but its own
// exchange words written) and the caller runs tile_pp_s<S, CD> on the tile; otherwise every window's per-pattern count is
// the plain prefix difference and the tile completes here (see tile_fused_s<.., CD> for the argument and the hand-over).
// RPT >= 0: the window's partial-block position r as a compile-time constant (the capture of the far-end block's prefix is one
// select per position otherwise: 8 S of a tile-lane's ~800 instructions); scan_read instantiates the r of the kernel's home k at the
// default window (k = 4 / 5 / 6 at W = 100) beside the run-time variant.
// PAIRF (D = 0, even RPT: k = 4 at the default window): two positions per lookup from a pair table of FIELDS -- the entry of the
// (k+1)-mer at p is field(p) + field(p + 1) (ScanArgs::pair_n with lut_fields; a table without self-overlap never matches one
// pattern at two adjacent positions, so the 2-bit fields still hold a block's counts): half the gathers and adds of phase 1.
template <int S, int D, int CD = 0, bool F16 = false, int RPT = -1, bool PAIRF = false>
TPS_DEV bool tile_pp_s(const ScanArgs& a, const TileConst& tc, const Lds& l, int delta, int w0, int tile, int nw_tile,
                       int64_t out_base, uint64_t& s_total) {
    static_assert(CD == 0 || (D == 0 && CD <= 6), "chain detection runs on the tile without self-overlap logic");
    static_assert(RPT < S, "r is a position inside a block");
    static_assert(!PAIRF || (D == 0 && CD == 0 && !F16 && RPT >= 0 && (RPT & 1) == 0), "pair lookups: plain tiles, the partial block ends between two pairs");
    // look-back over the previous lane's last LBK positions (tables with a self-overlap period only): enough for a chain that
    // starts inside it; a chain through the whole look-back takes the walk below
    constexpr int B = 8, POS = B * S, LBK = D > 0 ? 2 * D + 2 : 0;
    constexpr int WDW = (1 + LBK + POS + 6 + 7 + 15) / 16;
    constexpr int DH = D > 0 ? D : 1;
    constexpr int AHEAD = (D > 0 && 2 * D > S) ? 2 * D - S : 0;   // positions past the lane whose occurrences close a start-skip chain
    constexpr uint32_t M3 = 0x33333333u;
    const PatInfo& pat = a.pat;
    int rp = RPT >= 0 ? RPT : tc.r, q = tc.q;
    if constexpr (RPT < 0) TPS_PIN_S(rp);
    TPS_PIN_S(q);                                 // opaque per tile: what derives from them is recomputed (scalar) per tile, not kept in SGPRs across the read
    constexpr int LS = F16 ? 1 : 2;               // log2(bytes per table entry): F16 = the 16-bit field-index table (LUT_F16)
    const uint32_t amask = pat.kmask << LS;
    uint32_t* ende = l.XPC;                       // END, even patterns (padded block index)
    uint32_t* endo = l.row;                       // END, odd patterns; S_w takes the place after phase 2
    // per lane: its nibble totals, kept in the PAD word of the lane's 9-word group of END (round 4: the per-pattern tiles need no
    // XF / XT words of their own any more -- 160 dwords per wave less in the raw-row kernels' LDS slices)
    uint32_t* tne = ende + B;                     // lane t's total: tne[t * (B + 1)]
    uint32_t* tno = endo + B;
    uint32_t* carry = l.misc + M_SCAN;            // [0, 6): picks of the D positions before the next tile's first; [6]: uncertain
    const int cblk = a.tw & (B - 1), clane = a.tw >> 3;
    // the state at THIS tile's first position (one lane overwrites `carry` for the next tile during phase 1)
    uint32_t cold[7];
    TPS_UNROLL
    for (int i = 0; i < 7; ++i) cold[i] = (D > 0) ? uniform(carry[i]) : 0u;
    // was pattern pidx picked at LDS position pos, one of the D positions before the tile (delta - D <= pos < delta)?
    auto picked_before_tile = [&](int pos, int fidx) -> bool {      // (fidx: the pattern's FIELD, pp_field)
        uint32_t cv = 0;
        TPS_UNROLL
        for (int i = 0; i < DH; ++i) cv = (pos - (delta - DH) == i) ? cold[i] : cv;
        return w0 != 0 && ((cv >> (2 * fidx)) & 1u) != 0;
    };
#ifdef TPS_EMU
    uint32_t keep[NT][2 * B + 3];
    if (CD == 0) ++emu_counter(0);
#endif
    uint32_t ve[B], vo[B], chm = 0, chw = 0, unc = 0;
    uint32_t chain_any = 0;                       // CD: (field at p) & (field at p + CD), OR over the lane's positions
    TPS_PHASE {
        const int span = tid;
        const int p0 = delta + span * POS;        // >= 16
        const int bo = 2 * (p0 - LBK) - LS;       // the lane's registers start LBK positions before its first one (p0 >= 64)
        const uint32_t sh2 = (uint32_t)(bo & 31);
        const int d0 = bo >> 5;
        uint32_t w[WDW];
        {
            uint32_t prev = l.seq2[d0];
            TPS_UNROLL
            for (int i = 0; i < WDW; ++i) {
                uint32_t nx = l.seq2[d0 + i + 1];
                w[i] = alignbit(nx, prev, sh2);
                prev = nx;
            }
        }
        // one-hot field of the pattern that starts at position p (relative to the lane's first, -LBK <= p)
        auto look = [&](int p) -> uint32_t {
            const int idx = p + LBK, dw = idx >> 4, bit = idx & 15;
            const uint32_t v4 = bit ? alignbit(dw + 1 < WDW ? w[dw + 1] : 0u, w[dw], 2u * bit) : w[dw];
            if constexpr (F16) {
                const uint32_t m = lut16_at(l.lut, v4, amask);
                return mul24(m, m);                     // 1 << field index -> the one-hot 2-bit field
            }
            const uint32_t h = lut_at(l.lut, v4, amask);
            return h;                                   // (the table of these kernels holds the fields ready-made)
        };
        uint32_t pk[LBK + POS + AHEAD + 1];       // picks, index p + LBK (registers: only the last D are live)
        uint32_t tv[POS + 1];                     // skips
        if constexpr (D > 0) {
            uint32_t ca[LBK > 0 ? LBK : 1];       // same pattern at p, p - D, p - 2 D, ... down to the look-back's start
            TPS_UNROLL
            for (int i = 0; i < LBK; ++i) {
                const uint32_t h = look(i - LBK);
                if (i >= DH) {
                    pk[i] = h ^ (h & pk[i - DH]);
                    ca[i] = h & ca[i - DH];
                } else {
                    pk[i] = h;
                    ca[i] = h;
                }
            }
            TPS_UNROLL
            for (int i = LBK - DH; i < LBK; ++i) unc |= ca[i];
            if (unc && span > 0) {
                // One pattern chains through the whole look-back: walk the chain further back (rare, a few dependent
                // lookups).  An odd number of earlier links flips the picks of this chain; a chain that leaves the
                // staged tile stays unresolved unless the tile is the first (nothing precedes the region).
                unc = 0;
                TPS_UNROLL
                for (int i = LBK - DH; i < LBK; ++i) {
                    if (ca[i]) {
                        const int fidx = ffs0(ca[i]) >> 1, pidx = pp_pattern(fidx);       // the pattern's field / its place in the list
                        int n = 0, pw = p0 - LBK + (i % DH) - DH;
                        while (pw >= delta && ((lut_mask(l.lut, l.lshift, pat, v_at(l.seq2, pw)) >> pidx) & 1u)) { ++n; pw -= DH; }
                        // the chain's earliest element in the tile is skipped iff the same pattern was picked D before it,
                        // which for an element within D of the tile's start is what the previous tile left
                        const bool blocked = pw < delta && picked_before_tile(pw, fidx);
                        if (pw < delta && w0 != 0 && cold[6]) unc |= ca[i];
                        else if (((n & 1) != 0) != blocked) pk[i] ^= ca[i];
                    }
                }
            }
            if (span == 0) {                      // the state the previous tile left (nothing before the first tile)
                TPS_UNROLL
                for (int i = 0; i < DH; ++i) pk[LBK - DH + i] = cold[i];
                unc = cold[6];
            }
        }
        uint32_t pe = 0, po = 0;                  // lane-local counts so far, nibbles
        uint32_t ch3[B];
        TPS_UNROLL
        for (int blk = 0; blk < B; ++blk) ch3[blk] = 0;
        TPS_UNROLL
        for (int blk = 0; blk < B; ++blk) {
            if (D > 0) {
                if (blk == cblk && span == clane) {   // this block is the next tile's first
                    TPS_UNROLL
                    for (int i = 0; i < DH; ++i) carry[i] = pk[LBK + blk * S - DH + i];
                    carry[6] = unc;
                }
            }
            uint32_t acc = 0, pb = 0, sfw = 0;
            if constexpr (PAIRF) {
                const uint32_t amask2 = (pat.kmask << 4) | 0xCu;          // the (k+1)-mer's code << 2
                TPS_UNROLL
                for (int i = 0; i + 1 < S; i += 2) {
                    const int idx = blk * S + i, dw = idx >> 4, bit = idx & 15;          // (LBK = 0)
                    const uint32_t v4 = bit ? alignbit(dw + 1 < WDW ? w[dw + 1] : 0u, w[dw], 2u * bit) : w[dw];
                    acc += lut_at(l.lut2, v4, amask2);
                    if (i + 2 == RPT) pb = acc;
                }
                if (S & 1) acc += look(blk * S + S - 1);
            } else {
            TPS_UNROLL
            for (int i = 0; i < S; ++i) {
                const int p = blk * S + i;
                const uint32_t h = look(p);
                uint32_t t = 0;
                if (D > 0) {
                    t = h & pk[LBK + p - DH];
                    if (i < D) sfw |= t;
                    const int pm = p - D;         // an occurrence D after a start skip: the chain goes on
                    if (pm >= 0 && pm % S < D) ch3[pm / S] |= h & tv[pm];
                }
                tv[p] = t;
                const uint32_t pick = h ^ t;
                pk[LBK + p] = pick;
                if constexpr (CD > 0) {
                    if (p >= CD) chain_any |= h & pk[p - CD];
                }
                acc += pick;
                if constexpr (RPT >= 0) { if (i + 1 == RPT) pb = acc; }
                else pb = (i + 1 == rp) ? acc : pb;
            }
            }
            const uint32_t ee = pe + (pb & M3), eo = po + ((pb >> 2) & M3);
            ende[span * (B + 1) + blk] = ee;
            endo[span * (B + 1) + blk] = eo;
            ve[blk] = PP_BIAS * 0x11111111u - pe + (sfw & M3);
            vo[blk] = PP_BIAS * 0x11111111u - po + ((sfw >> 2) & M3);
            pe += acc & M3;
            po += (acc >> 2) & M3;
            if (D > 0 && blk > 0) {               // the chain flags of the previous block are complete (2 D <= 2 S positions on)
                chm |= ch3[blk - 1] ? (1u << (blk - 1)) : 0u;
                chw |= ch3[blk - 1];
                TPS_PIN_V(chm); TPS_PIN_V(chw);
            }
        }
        if (D > 0) {
            TPS_UNROLL
            for (int i = 0; i < AHEAD; ++i) {     // occurrences just past the lane that continue a start-skip chain of its last block
                const int p = POS + i, pm = p - D;
                if (pm % S < D) ch3[pm / S] |= look(p) & tv[pm];
            }
            chm |= ch3[B - 1] ? (1u << (B - 1)) : 0u;
            chw |= ch3[B - 1];
        }
        if constexpr (CD > 0) {
            TPS_UNROLL
            for (int i = 0; i < CD; ++i) chain_any |= look(POS + i) & pk[POS + i - CD];   // pairs that reach into the next lane's positions
        }
        tne[span * (B + 1)] = pe;
        tno[span * (B + 1)] = po;
        TPS_PIN_V(chm); TPS_PIN_V(chw); TPS_PIN_V(unc);
        TPS_UNROLL
        for (int i = 0; i < B; ++i) { TPS_PIN_V(ve[i]); TPS_PIN_V(vo[i]); }
#ifdef TPS_EMU
        for (int i = 0; i < B; ++i) { keep[tid][i] = ve[i]; keep[tid][B + i] = vo[i]; }
        keep[tid][2 * B] = chm; keep[tid][2 * B + 1] = unc; keep[tid][2 * B + 2] = chw;
        chm = 0; unc = 0; chw = 0;
#endif
    }
    TPS_SYNC();
    TPS_PP_STAMP(6);
    if constexpr (CD > 0) {
#ifdef TPS_EMU
        const bool chained = chain_any != 0;      // (the emulation's phase loop has OR-ed every lane into the one variable)
#else
        const bool chained = __builtin_amdgcn_ballot_w64(chain_any != 0) != 0;
#endif
        if (chained) return true;
#ifdef TPS_EMU
        ++emu_counter(0);
        ++emu_counter(7);
#endif
    }
    // lanes whose look-back could not fix their state: every window that touches one of them is recounted
    uint64_t unc_mask = 0;
    if (D > 0) {
#ifdef TPS_EMU
        for (int t = 0; t < NT; ++t) unc_mask |= (uint64_t)(keep[t][2 * B + 1] != 0) << t;
        emu_counter(2) += __builtin_popcountll(unc_mask);
#else
        unc_mask = __builtin_amdgcn_ballot_w64(unc != 0);
#endif
    }
    const int rot = q & (B - 1), dl0 = q >> 3;
#ifdef TPS_EMU
    uint32_t sw_keep[NT][B];
    uint32_t rows_keep[NT][B][4];
#endif
    uint32_t swv[B], rows[B][3], rx[B / 2], todo = 0;     // rx: bytes 12-13 of the rows (14 patterns), two rows per word
#ifdef TPS_EMU
    uint32_t todo_keep[NT];
    for (int t = 0; t < NT; ++t) todo_keep[t] = 0;
#endif
    // raw rows (P <= 14 bytes: plan_geometry) always leave through LDS, coalesced: rows of 4, 8 or 12 bytes packed and in
    // whole cache lines; the others padded to 16 bytes in LDS and copied out in 16-bit units (P even) or bytes (P odd)
    const bool staged = a.raw != nullptr && (pat.P & 3) == 0;
    const bool staged16 = a.raw != nullptr && (pat.P & 3) != 0;
    TPS_PHASE {
#ifdef TPS_EMU
        for (int i = 0; i < B; ++i) { ve[i] = keep[tid][i]; vo[i] = keep[tid][B + i]; }
        chm = keep[tid][2 * B]; chw = keep[tid][2 * B + 2];
#endif
        const int lane = tid;
        bool redo_all = false;
        if (D > 0) redo_all = ((unc_mask >> lane) & ((2ull << (dl0 + 1)) - 1ull)) != 0;
        // the far-end words of the lane's 8 windows: one LDS round trip (windows past the tile read in-bounds garbage: their rows
        // are never copied out, their sums are zeroed below)
        const uint32_t* pee = ende + lane * (B + 1);
        const uint32_t* peo = endo + lane * (B + 1);
        uint32_t ee[B], eo[B];
        TPS_UNROLL
        for (int j = 0; j < B; ++j) {
            const int eb = j + q;
            ee[j] = pee[eb + (eb >> 3)];
            eo[j] = peo[eb + (eb >> 3)];
        }
        if (D > 0) {
            // windows to repair after the fast pass: bits 0-7 recount (the lane's state before its first position is not
            // known), bits 8-15 chain parity
            todo = redo_all ? 0xFFu : (chm << 8);
#ifdef TPS_EMU
            todo_keep[tid] = todo;
#endif
        }
        const int nv = nw_tile - lane * B;         // windows of this lane inside the tile (<= 0: none)
        // NW byte words per row: the fields are laid out so that word i holds patterns 4 i .. 4 i + 3 (pp_field) -- a list of at
        // most 12 patterns (every 6-mer motif) never touches the fourth word
        auto windows = [&](auto nw_c) {
            constexpr int NW = decltype(nw_c)::value;
            // whole lanes a window covers after its own block: near end dl0 lanes (own lane included), far end one more
            uint32_t fa[NW], fb[NW];
            TPS_UNROLL
            for (int i = 0; i < NW; ++i) fa[i] = 0;
            TPS_NOVEC
            for (int t = 0; t < dl0; ++t) {
                uint32_t x[4];
                pp_expand(tne[(lane + t) * (B + 1)], tno[(lane + t) * (B + 1)], x);
                TPS_UNROLL
                for (int i = 0; i < NW; ++i) fa[i] += x[i];
            }
            {
                uint32_t x[4];
                pp_expand(tne[(lane + dl0) * (B + 1)], tno[(lane + dl0) * (B + 1)], x);
                TPS_UNROLL
                for (int i = 0; i < NW; ++i) {
                    fa[i] -= PP_BIAS * 0x01010101u;   // the bias of V; the running sums below are whole-word arithmetic
                    fb[i] = fa[i] + x[i];
                }
            }
            // windows j >= B - rot end one lane further on (far end): the running totals switch from fa to fb there, once
            // (a scalar branch around the copies instead of a select per word and window)
            uint32_t fc[NW];
            TPS_UNROLL
            for (int i = 0; i < NW; ++i) fc[i] = fa[i];
            const int brk = B - rot;
            TPS_UNROLL
            for (int j = 0; j < B; ++j) {
                if (j == brk) {                        // uniform
                    TPS_UNROLL
                    for (int i = 0; i < NW; ++i) { fc[i] = fb[i]; TPS_PIN_V(fc[i]); }
                }
                uint32_t v[4], e[4], c[4] = {0, 0, 0, 0};
                pp_expand(ve[j], vo[j], v);
                pp_expand(ee[j], eo[j], e);
                uint32_t sw = (uint32_t)pat.P - 4u * NW;      // the fields no pattern owns are floored to 1 like the others
                TPS_UNROLL
                for (int i = 0; i < NW; ++i) {
                    c[i] = fc[i] + v[i] + e[i];
                    c[i] |= ((0x80808080u - c[i]) >> 7) & 0x01010101u;     // `matches or 1` per byte (counts <= 127): bit 0 set where the byte is 0
                    sw = add_bytes(c[i], sw);
                }
                swv[j] = j < nv ? sw : 0u;
                TPS_UNROLL
                for (int i = 0; i < 3; ++i) rows[j][i] = c[i];        // the row as it is: word i = patterns 4 i .. 4 i + 3
                if (j & 1) rx[j >> 1] = pack_hi_lo(c[3] << 16, rx[j >> 1]);
                else rx[j >> 1] = c[3] & 0xFFFFu;
#ifdef TPS_EMU
                sw_keep[tid][j] = swv[j];
                for (int i = 0; i < 4; ++i) rows_keep[tid][j][i] = c[i];
#endif
            }
        };
        if (pat.P <= 12) windows(IntC<3>{});
        else windows(IntC<4>{});
        g_store_sw8(tc.sw16 + w0, lane, nw_tile, swv);     // (0 for the windows past the tile: dropped by the range check)
    }
    TPS_SYNC();                                   // every END word has been read: both halves of END are free
    TPS_PP_STAMP(7);
    bool rows_done = false;
    if constexpr (TPS_RAW_M != 0) {
        if (staged && a.raw_m == 2) {
            // Strided scans (round 5): this scan runs at HALF the requested slide and only the even windows are wanted -- tiles start at
            // even windows, so a lane keeps its rows 0, 2, 4, 6: 4 P contiguous bytes per lane in the requested slide's layout, all 64
            // lanes in ONE pass through the (free) END area, copied out in 16-byte pieces like the full rows below.
            rows_done = true;
            const int pd = pat.P >> 2;            // dwords per row: 1, 2 or 3
            uint32_t* buf = l.XPC;
            uint32_t* gout = (uint32_t*)(a.raw + (a.raw_win_off[tc.rd] + (w0 >> 1)) * (int64_t)pat.P);
            TPS_PHASE {
                uint32_t* dst = buf + tid * 4 * pd;
#ifdef TPS_EMU
                for (int j = 0; j < B; ++j) for (int i = 0; i < 3; ++i) rows[j][i] = rows_keep[tid][j][i];
#endif
                TPS_UNROLL
                for (int j = 0; j < 4; ++j) {
                    TPS_UNROLL
                    for (int i = 0; i < 3; ++i)
                        if (i < pd) dst[j * pd + i] = rows[2 * j][i];
                }
            }
            TPS_SYNC();
            TPS_PHASE {
                const int nvalid = ((nw_tile + 1) >> 1) * pd;       // dwords of the tile's even windows
                const int npass = (nvalid + 4 * NT - 1) / (4 * NT);  // uniform
                TPS_NOVEC
                for (int it = 0; it < npass; ++it) {
                    const int cdw = 4 * tid + it * 4 * NT;
                    g_store16_clamped(gout, nvalid, cdw, *(const u32x4*)(buf + cdw));
                }
            }
            TPS_SYNC();
        }
    }
    if (rows_done) {
    } else if (staged) {
        // Raw rows leave through LDS: a lane's 8 rows are 8 P contiguous bytes in HBM, LPP lanes per pass mirror a
        // contiguous stretch of the output in the (now free) END / totals area, and all 64 lanes copy it out in
        // 16-byte pieces -- full cache lines instead of 64 scattered 12-byte writes per store instruction.
        // (round 4: the buffer is END's even AND odd half -- XPC and row[], contiguous in the raw-row kernels' slices, 1152 dwords --
        // and S_w moves into row[] only afterwards: 48 lanes of 12-byte rows per pass instead of 22, two passes per tile instead of three)
        const int pd = pat.P >> 2;                // dwords per row: 1, 2 or 3
        const int LPP = pd == 3 ? 48 : NT;
        uint32_t* buf = l.XPC;
        uint32_t* gout = (uint32_t*)(a.raw + (out_base + w0) * (int64_t)pat.P);
        for (int l0 = 0; l0 < NT; l0 += LPP) {
            TPS_PHASE {
                if (tid >= l0 && tid < l0 + LPP) {
                    uint32_t* dst = buf + (tid - l0) * B * pd;
#ifdef TPS_EMU
                    for (int j = 0; j < B; ++j) for (int i = 0; i < 3; ++i) rows[j][i] = rows_keep[tid][j][i];
#endif
                    if (pd == 3) {
                        TPS_UNROLL
                        for (int g = 0; g < 6; ++g) {
                            u32x4 t;
                            t.x = rows[(4 * g) / 3][(4 * g) % 3]; t.y = rows[(4 * g + 1) / 3][(4 * g + 1) % 3];
                            t.z = rows[(4 * g + 2) / 3][(4 * g + 2) % 3]; t.w = rows[(4 * g + 3) / 3][(4 * g + 3) % 3];
                            *(u32x4*)(dst + 4 * g) = t;
                        }
                    } else if (pd == 2) {
                        TPS_UNROLL
                        for (int j = 0; j < B; ++j) { dst[2 * j] = rows[j][0]; dst[2 * j + 1] = rows[j][1]; }
                    } else {
                        TPS_UNROLL
                        for (int j = 0; j < B; ++j) dst[j] = rows[j][0];
                    }
                }
            }
            TPS_SYNC();
            TPS_PHASE {
                int nvalid = (nw_tile - l0 * B) * pd;          // dwords of this pass that belong to the tile's windows
                const int ncap = LPP * B * pd;
                if (nvalid > ncap) nvalid = ncap;
                uint32_t* g = gout + (int64_t)l0 * B * pd;
                const int npass = (nvalid + 4 * NT - 1) / (4 * NT);      // uniform
                TPS_NOVEC
                for (int it = 0; it < npass; ++it) {
                    const int cdw = 4 * tid + it * 4 * NT;
                    g_store16_clamped(g, nvalid, cdw, *(const u32x4*)(buf + cdw));      // (reads past nvalid stay inside the wave's LDS slice; what they fetch is dropped)
                }
            }
            TPS_SYNC();
        }
    } else if (staged16) {
        constexpr int LPP = 36;                   // 36 lanes x 8 rows x 16 bytes = 4608 bytes = XPC + row[] (see above)
        const int ph = pat.P >> 1;                // 16-bit units per row: 1, 3, 5 or 7
        const uint32_t ph_magic = ph ? (65536u + (uint32_t)ph - 1u) / (uint32_t)ph : 0u;    // u / ph = (u * magic) >> 16 for u < 2^15
        uint32_t* buf = l.XPC;
        uint16_t* gout = (uint16_t*)(a.raw + (out_base + w0) * (int64_t)pat.P);
        for (int l0 = 0; l0 < NT; l0 += LPP) {
            TPS_PHASE {
                if (tid >= l0 && tid < l0 + LPP) {
                    uint32_t* dst = buf + (tid - l0) * B * 4;
                    TPS_UNROLL
                    for (int j = 0; j < B; ++j) {
                        u32x4 t;
#ifdef TPS_EMU
                        t.x = rows_keep[tid][j][0]; t.y = rows_keep[tid][j][1]; t.z = rows_keep[tid][j][2]; t.w = rows_keep[tid][j][3] & 0xFFFFu;
#else
                        t.x = rows[j][0]; t.y = rows[j][1]; t.z = rows[j][2]; t.w = (rx[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
#endif
                        *(u32x4*)(dst + 4 * j) = t;
                    }
                }
            }
            TPS_SYNC();
            if ((pat.P & 1) == 0) {
                TPS_PHASE {
                    int nvalid = (nw_tile - l0 * B) * ph;          // 16-bit units of this pass that belong to the tile's windows
                    const int ncap = LPP * B * ph;
                    if (nvalid > ncap) nvalid = ncap;
                    uint16_t* g = gout + (int64_t)l0 * B * ph;
                    const uint16_t* b16 = (const uint16_t*)buf;
                    for (int u = tid; u < nvalid; u += NT) {
                        const uint32_t row = ((uint32_t)u * ph_magic) >> 16;
                        g[u] = b16[row * 8u + ((uint32_t)u - row * (uint32_t)ph)];
                    }
                }
            } else {
                // odd row length (a hand-made pattern list): the same, byte by byte
                const uint32_t pb = (uint32_t)pat.P, pb_magic = (65536u + pb - 1u) / pb;      // u / P = (u * magic) >> 16 for u < 2^15
                TPS_PHASE {
                    int nvalid = (nw_tile - l0 * B) * (int)pb;
                    const int ncap = LPP * B * (int)pb;
                    if (nvalid > ncap) nvalid = ncap;
                    uint8_t* g = (uint8_t*)gout + (int64_t)l0 * B * pb;
                    const uint8_t* b8 = (const uint8_t*)buf;
                    for (int u = tid; u < nvalid; u += NT) {
                        const uint32_t row = ((uint32_t)u * pb_magic) >> 16;
                        g[u] = b8[row * 16u + ((uint32_t)u - row * pb)];
                    }
                }
            }
            TPS_SYNC();
        }
    }
    // S_w takes the place of END's odd half (the candidate phase and the repairs read it there)
    TPS_PHASE {
        TPS_UNROLL
        for (int j = 0; j < B; ++j) {
#ifdef TPS_EMU
            swv[j] = sw_keep[tid][j];
#endif
            l.row[tid * (B + 1) + j] = swv[j];
        }
        if constexpr (CD > 0) {
            // what a chained successor wants from this (chain-free) tile: no pick before its first position matters -- a pair
            // across that position would lie inside this tile -- and nothing is uncertain
            if (tid == clane) {
                TPS_UNROLL
                for (int i = 0; i < 7; ++i) carry[i] = 0u;
            }
        }
    }
    TPS_SYNC();
    if (D > 0) {
        // Repairs (rare, lane-divergent): results are in memory by now -- S_w in row[] and HBM, raw rows in HBM.
#ifndef TPS_EMU
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // this wave's own raw-row stores before its byte updates
#endif
        TPS_PHASE {
#ifdef TPS_EMU
            todo = todo_keep[tid];
            chw = keep[tid][2 * B + 2];
#endif
            const int lane = tid;
            while (todo) {
                const int bit = ffs0(todo);
                todo &= todo - 1u;
                const int j = bit & 7, wl = lane * B + j;
                if (wl >= nw_tile) continue;
                uint8_t* raw_row = a.raw ? a.raw + (out_base + w0 + wl) * (int64_t)pat.P : nullptr;
                if (TPS_RAW_M != 0 && a.raw && a.raw_m == 2)      // strided scans: only the even windows have a row, in the requested slide's layout
                    raw_row = ((w0 + wl) & 1) ? nullptr : a.raw + (a.raw_win_off[tc.rd] + ((w0 + wl) >> 1)) * (int64_t)pat.P;
                uint32_t sw = l.row[lane * (B + 1) + j];
                if (bit < 8) {
#ifdef TPS_EMU
                    ++emu_counter(1);
#endif
                    sw = window_exact(a, l, delta, wl, pat.all_mask, 0u, raw_row, true);
                } else {
                    // The window starts on a canonically skipped occurrence x whose chain goes on: it picks x, x + 2D, ...
                    // = one more than canonical iff the chain has an odd number of elements from x (inside the window).
                    // The start skip was added; take it back for an even count.
#ifdef TPS_EMU
                    ++emu_counter(3);
#endif
                    // chw is the union over the lane's flagged blocks, so x's own place in its chain is walked too: n earlier
                    // links (odd = canonically skipped = a start skip of this window), m elements from x inside the window.
                    const int a0 = delta + wl * S;
                    uint32_t fw = chw;
                    bool lost = false;
                    while (fw) {
                        const int fidx = ffs0(fw) >> 1, pidx = pp_pattern(fidx);
                        fw &= fw - 1u;
                        int x = -1;
                        TPS_NOVEC
                        for (int i = 0; i < D; ++i)
                            if ((lut_mask(l.lut, l.lshift, pat, v_at(l.seq2, a0 + i)) >> pidx) & 1u) x = a0 + i;
                        if (x >= 0) {
                            int n = 0, pb = x - D;
                            while (pb >= delta && ((lut_mask(l.lut, l.lshift, pat, v_at(l.seq2, pb)) >> pidx) & 1u)) { ++n; pb -= D; }
                            const bool blocked = pb < delta && picked_before_tile(pb, fidx);
                            if (pb < delta && w0 != 0 && cold[6]) lost = true;      // the chain leaves the tile and the state there is unknown
                            int m = 1;
                            for (int pw = x + D; pw < a0 + a.lw && ((lut_mask(l.lut, l.lshift, pat, v_at(l.seq2, pw)) >> pidx) & 1u); pw += D) ++m;
                            if ((((n & 1) != 0) != blocked) && (m & 1) == 0) {
                                sw -= 1u;
                                if (raw_row) raw_row[pidx] = (uint8_t)(raw_row[pidx] - 1u);
                            }
                        }
                    }
                    if (lost) {
#ifdef TPS_EMU
                        ++emu_counter(1);
#endif
                        sw = window_exact(a, l, delta, wl, pat.all_mask, 0u, raw_row, true);
                    }
                }
                l.row[lane * (B + 1) + j] = sw;
                tc.sw16[w0 + wl] = (uint16_t)sw;
            }
        }
        TPS_SYNC();
    }
    TPS_PP_STAMP(11);
    tile_candidates(tc, l, w0, tile, nw_tile, s_total);
    TPS_PP_STAMP(12);
    return false;
}

// ------------------------------------------------------------------ step 3: single-split Binseg (l2)
// gain(b) = cost(0,n) - cost(0,b) - cost(b,n) = (n L_b - T b)^2 / (n b (n-b))   [y = S / P]
// so the arg-max over b in {0, jump, 2 jump, ...}, b >= min_size, n-b >= min_size is that of
// D_b^2 / (b (n-b)), D_b = n L_b - T b (an exact integer).  Scores are compared in float64; if
// more than one candidate lies within float noise of the best score, an exact 128-bit integer
// tournament decides (ties -> larger b, as max() over (gain, bkp) tuples does in ruptures'
// Binseg._single_bkp).
struct Cand { uint64_t d; uint64_t den; int32_t b; };
TPS_DEV bool cand_better(const Cand& x, const Cand& y) {
    if (x.b < 0) return false;
    if (y.b < 0) return true;
    u128 lhs = (u128)x.d * x.d * y.den;
    u128 rhs = (u128)y.d * y.d * x.den;
    if (lhs != rhs) return lhs > rhs;
    return x.b > y.b;
}
TPS_DEV bool binseg_admissible(int n, int jump, int min_size) {
    if (n / jump < 1) return false;
    int need = ((min_size + jump - 1) / jump) * jump + min_size;
    return need <= n;
}
template <typename ST>
TPS_DEV Cand binseg_chunk(const ST* S, int n, int jump, int min_size, int lo, int hi, uint64_t prefix, uint64_t total) {
    Cand best{0, 1, -1};
    uint64_t run = prefix;
    for (int b = lo; b < hi; ++b) {
        if (b % jump == 0 && b >= min_size && n - b >= min_size) {
            int64_t d = (int64_t)n * (int64_t)run - (int64_t)total * (int64_t)b;
            Cand c{(uint64_t)(d < 0 ? -d : d), (uint64_t)b * (uint64_t)(n - b), b};
            if (cand_better(c, best)) best = c;
        }
        run += (uint64_t)S[b];
    }
    return best;
}
TPS_DEV double score_f64(int64_t d, uint64_t den) {
    double x = (double)d;
    return x * x / (double)den;
}
TPS_DEV double gain_from(int64_t d, uint64_t den, int n, int n_patterns) {
    double x = (double)d;
    return x * x / ((double)n * (double)den) / ((double)n_patterns * (double)n_patterns);
}

// max of a non-negative f64 (as its bit pattern) over the workgroup into *slot (LDS, pre-zeroed):
// wave-level butterfly first so that only one LDS atomic per wave is issued.
TPS_DEV void wg_max_bits(uint64_t bits, uint64_t* slot) {
#ifndef TPS_EMU
    TPS_UNROLL
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)bits, d), hi = __shfl_xor((uint32_t)(bits >> 32), d);
        uint64_t o = ((uint64_t)hi << 32) | lo;
        bits = o > bits ? o : bits;
    }
    if ((threadIdx.x & 63) != 0) return;
#endif
    lds_max_u64(slot, bits);
}

// Exact tournament over all candidates (slow path), scratch xs (XS_DW dwords).  Called by the
// whole workgroup outside TPS_PHASE.
template <typename ST>
TPS_DEV Cand binseg_exact_wg(const ST* S, int n, int jump, int min_size, uint32_t* xs) {
    const int cl = (n + NT - 1) / NT;
    TPS_PHASE {
        int lo = tid * cl, hi = lo + cl < n ? lo + cl : n;
        uint32_t sm = 0;
        for (int i = lo; i < hi; ++i) sm += (uint32_t)S[i];
        xs[X_BS + tid] = sm;
    }
    TPS_SYNC();
    constexpr int NG = NT / 16;                   // groups of 16 lanes
    TPS_PHASE {
        if (tid < NG) {
            uint32_t sm = 0;
            for (int i = 0; i < 16; ++i) sm += xs[X_BS + tid * 16 + i];
            xs[X_Q + tid] = sm;
        }
    }
    TPS_SYNC();
    TPS_PHASE {
        if (tid == 0) {
            uint32_t run = 0;
            for (int i = 0; i < NG; ++i) { uint32_t t = xs[X_Q + i]; xs[X_Q + i] = run; run += t; }
            xs[X_Q + 16] = run;
        }
    }
    TPS_SYNC();
    TPS_PHASE {
        uint64_t pre = xs[X_Q + (tid >> 4)];
        for (int i = (tid & ~15); i < tid; ++i) pre += xs[X_BS + i];
        int lo = tid * cl, hi = lo + cl < n ? lo + cl : n;
        Cand c = binseg_chunk(S, n, jump, min_size, lo, hi, pre, (uint64_t)xs[X_Q + 16]);
        ((uint64_t*)&xs[X_CD])[tid] = c.d;
        ((uint64_t*)&xs[X_CDEN])[tid] = c.den;
        ((int32_t*)&xs[X_CB])[tid] = c.b;
    }
    TPS_SYNC();
    TPS_PHASE {
        if (tid < NG) {
            Cand best{0, 1, -1};
            for (int i = 0; i < 16; ++i) {
                int t = tid * 16 + i;
                Cand c{((uint64_t*)&xs[X_CD])[t], ((uint64_t*)&xs[X_CDEN])[t], ((int32_t*)&xs[X_CB])[t]};
                if (cand_better(c, best)) best = c;
            }
            ((uint64_t*)&xs[X_R])[tid] = best.d;
            ((uint64_t*)&xs[X_R + 32])[tid] = best.den;
            ((int32_t*)&xs[X_R + 64])[tid] = best.b;
        }
    }
    TPS_SYNC();
    Cand best{0, 1, -1};
    for (int i = 0; i < NG; ++i) {
        Cand c{((uint64_t*)&xs[X_R])[i], ((uint64_t*)&xs[X_R + 32])[i], ((int32_t*)&xs[X_R + 64])[i]};
        if (cand_better(c, best)) best = c;
    }
    TPS_SYNC();
    return best;
}

// Workgroup-wide Binseg over S[0..n): float64 scores, exact fallback when float noise cannot
// separate the best candidates.  bs: NT dwords, misc: MISC_DW region, xs: XS_DW scratch.
// A thread owns a chunk of cl = jump * ceil(n / (jump NT)) consecutive windows; chunks of up to
// CHUNK_REGS windows are read from LDS/HBM once and kept in registers.
constexpr int CHUNK_REGS = 16;
template <typename ST>
TPS_DEV void binseg_wg(const ST* S, int n, int jump, int min_size, int n_patterns, uint32_t* bs, uint32_t* misc,
                       uint32_t* xs, int& bkp, double& gain, bool& tie) {
    const int per = (n + jump * NT - 1) / (jump * NT);
    const int cl = per * jump;
    const bool in_regs = cl <= CHUNK_REGS;
    TPS_PHASE {
        const int lo = tid * cl, hi = lo + cl < n ? lo + cl : n;
        uint32_t sm = 0;
        if (in_regs) {
            TPS_UNROLL
            for (int i = 0; i < CHUNK_REGS; ++i)
                if (lo + i < hi) sm += (uint32_t)S[lo + i];
        } else {
            for (int i = lo; i < hi; ++i) sm += (uint32_t)S[i];
        }
        bs[tid] = sm;
        if (tid == 0) { *(uint64_t*)&misc[M_MAXSC] = 0ull; misc[M_BESTB] = (uint32_t)-1; misc[M_NTIE] = 0u; }
    }
    TPS_SYNC();
    const uint64_t tot = wg_exclusive_scan(bs, NT, &misc[M_SCAN]);
    // per-thread best and runner-up scores over its candidates; they stay in registers on the
    // device and in bs-adjacent scratch (xs) in the emulation, where phases are separate loops
#ifdef TPS_EMU
    double* keep = (double*)xs;                   // 3 doubles per thread: best, second, (double)best_b
#endif
    double best = -1.0, second = -1.0;
    int best_b = -1;
    TPS_PHASE {
        const int lo = tid * cl, hi = lo + cl < n ? lo + cl : n;
        uint64_t run = bs[tid];
        best = -1.0; second = -1.0; best_b = -1;
        auto visit = [&](int b, uint32_t sv, int& j) {
            if (j == 0 && b >= min_size && n - b >= min_size) {
                int64_t d = (int64_t)n * (int64_t)run - (int64_t)tot * (int64_t)b;
                double sc = score_f64(d, (uint64_t)b * (uint64_t)(n - b));
                if (sc >= best) { second = best; best = sc; best_b = b; }
                else if (sc > second) second = sc;
            }
            run += (uint64_t)sv;
            if (++j == jump) j = 0;
        };
        int j = 0;
        if (in_regs) {
            TPS_UNROLL
            for (int i = 0; i < CHUNK_REGS; ++i)
                if (lo + i < hi) visit(lo + i, (uint32_t)S[lo + i], j);
        } else {
            for (int b = lo; b < hi; ++b) visit(b, (uint32_t)S[b], j);
        }
        uint64_t bits = 0;
        if (best >= 0.0) __builtin_memcpy(&bits, &best, 8);   // non-negative doubles order like integers
        wg_max_bits(bits, (uint64_t*)&misc[M_MAXSC]);
#ifdef TPS_EMU
        keep[3 * tid] = best; keep[3 * tid + 1] = second; keep[3 * tid + 2] = (double)best_b;
#endif
    }
    TPS_SYNC();
    TPS_PHASE {
#ifdef TPS_EMU
        best = keep[3 * tid]; second = keep[3 * tid + 1]; best_b = (int)keep[3 * tid + 2];
#endif
        double m;
        __builtin_memcpy(&m, &misc[M_MAXSC], 8);
        const double thr = m * (1.0 - 1e-14);
        uint32_t near = (best >= thr && best >= 0.0 ? 1u : 0u) + (second >= thr && second >= 0.0 ? 1u : 0u);
        if (near) lds_add(&misc[M_NTIE], near);
        if (best == m && best_b >= 0) lds_max_i32((int32_t*)&misc[M_BESTB], best_b);
    }
    TPS_SYNC();
    tie = misc[M_NTIE] > 1u;
    if (tie) {                                    // float noise cannot separate them: exact integers decide
        Cand ex = binseg_exact_wg(S, n, jump, min_size, xs);
        bkp = ex.b;
        gain = ex.b < 0 ? 0.0 : gain_from((int64_t)ex.d, ex.den, n, n_patterns);
    } else {
        bkp = (int32_t)misc[M_BESTB];
        double m;
        __builtin_memcpy(&m, &misc[M_MAXSC], 8);
        gain = bkp < 0 ? 0.0 : m / (double)n / ((double)n_patterns * (double)n_patterns);
        TPS_SYNC();
    }
}

// Fused Binseg from the candidate left sums Lc[c] = sum_{w < c*jump} S_w (c = 1 .. (n-1)/jump) and
// the total T: float64 scores, wave arg-max; if more than one candidate lies within float noise of
// the best score the exact integer tournament re-reads S_w from HBM (rare).
template <typename ST>
TPS_DEV void binseg_from_lc(const ScanArgs& a, const Lds& l, uint64_t lc_g, const ST* S_global, int n, uint64_t tot, int jump,
                            int min_size, int n_patterns, uint32_t* misc, uint32_t* xs, int& bkp, double& gain, bool& tie) {
    const int ncand = (n - 1) / jump;              // candidates b = c*jump, 1 <= c <= ncand  (b < n)
#ifdef TPS_EMU
    double* keep = (double*)xs;
#endif
    // Per lane: the best candidate as a FRACTION (num = D^2, den = b (n - b)); candidates are compared by
    // cross-multiplication, so the f64 division happens once per lane instead of once per candidate.
    // `amb` = another candidate of this lane lies within 1e-13 (relative) of the lane's best: if that
    // best is also the global one, float64 cannot be trusted to separate them -> exact path.
    double best = -1.0;
    int best_b = -1;
    bool amb = false;
    uint64_t bits = 0;
    // admissible candidates: b = c * jump with b >= min_size and n - b >= min_size
    const int c_min = (min_size + jump - 1) / jump > 1 ? (min_size + jump - 1) / jump : 1;
    const int c_max = (n - min_size) / jump < ncand ? (n - min_size) / jump : ncand;
    // D = n L_b - T b as an exact float64 integer when both products stay below 2^53 (always, for the
    // fused geometry's window sizes); otherwise through int64 like the standalone path
    const bool exact53 = (double)n * 4294967296.0 < 9007199254740992.0 && (double)tot * (double)n < 9007199254740992.0;
    const double nf = (double)n, totf = (double)tot;
    // Float32 prefilter (off-chip 16-bit sums, the fused kernels' normal case): the score D^2 / (b (n - b)) of every
    // candidate in single precision from the EXACT D (float64 fma, then rounded once), the wave maximum, and only
    // candidates within 1e-4 of it -- two orders of magnitude more than single precision can be off -- go through the
    // float64 fraction comparison below (41 instructions per candidate; the prefilter costs about a third of that).
    constexpr int LCV = 16;
    const bool prefilter = !a.lc16 && lc_g && exact53 && c_max - c_min < LCV * NT;
    int nslot = c_max >= c_min ? (c_max - c_min + NT) / NT : 0;          // candidate slots per lane that any lane uses
    TPS_PIN_S(nslot);
    auto tile_sum = [&](int c) { return l.Tc[(uint32_t)(((uint64_t)(uint32_t)(c * jump) * a.tw_magic) >> 32)]; };
    // per lane: its best single-precision score with that candidate's index and left sum, and its second-best score
    float p_s1 = -1.0f, p_s2 = -1.0f, thr32 = -1.0f;
    uint32_t p_lc = 0;
    int p_c = -1;
    bool crowded = false;                          // some lane holds a second candidate within the prefilter's margin
#ifdef TPS_EMU
    static thread_local float ps1_keep[NT], ps2_keep[NT];
    static thread_local uint32_t plc_keep[NT];
    static thread_local int pc_keep[NT];
    float emu_m32 = 0.0f;
#endif
    if (prefilter) {
        TPS_PHASE {
            p_s1 = -1.0f; p_s2 = -1.0f; p_lc = 0; p_c = -1;
            // four slots at a time behind ONE uniform test (nslot lives in an SGPR): slots no lane uses cost nothing;
            // the four loads of a group are requested before the first one is used
            auto group = [&](int g) {
                uint32_t lcv[4];
                TPS_UNROLL
                for (int i = 0; i < 4; ++i) {
                    const int c = c_min + tid + (4 * g + i) * NT;
                    lcv[i] = c <= c_max ? g32_load(lc_g, (uint32_t)c) : 0u;
                }
                TPS_UNROLL
                for (int i = 0; i < 4; ++i) {
                    const int c = c_min + tid + (4 * g + i) * NT;
                    const int b = c * jump;
                    const double bf = (double)b;
                    const float d32 = (float)__builtin_fma(-totf, bf, nf * (double)lcv[i]);
                    const float den32 = (float)b * (float)(n - b);
#ifdef TPS_EMU
                    float s_ = d32 * d32 * (1.0f / den32);
#else
                    float s_ = d32 * d32 * __builtin_amdgcn_rcpf(den32);
#endif
                    s_ = c <= c_max ? s_ : -1.0f;
                    const bool gt = s_ > p_s1;                    // (an exact tie becomes the runner-up: conservative)
                    const float lose = gt ? p_s1 : s_;
                    p_s2 = lose > p_s2 ? lose : p_s2;
                    p_lc = gt ? lcv[i] : p_lc;
                    p_c = gt ? c : p_c;
                    p_s1 = gt ? s_ : p_s1;
                }
            };
            if (nslot > 0) group(0);
            if (nslot > 4) group(1);
            if (nslot > 8) group(2);
            if (nslot > 12) group(3);
#ifdef TPS_EMU
            ps1_keep[tid] = p_s1; ps2_keep[tid] = p_s2; plc_keep[tid] = p_lc; pc_keep[tid] = p_c;
            emu_m32 = p_s1 > emu_m32 ? p_s1 : emu_m32;
#endif
        }
        float m32;
#ifdef TPS_EMU
        m32 = emu_m32;
#else
        {
            const float nn = p_s1 > 0.0f ? p_s1 : 0.0f;
            uint32_t mb;
            __builtin_memcpy(&mb, &nn, 4);          // non-negative floats order like integers
            mb = wave_max_u32(mb);
            __builtin_memcpy(&m32, &mb, 4);
        }
#endif
        thr32 = m32 * (1.0f - 1e-4f);
#ifdef TPS_EMU
        for (int t = 0; t < NT; ++t) crowded = crowded || (ps2_keep[t] >= thr32 && ps2_keep[t] >= 0.0f);
#else
        crowded = __builtin_amdgcn_ballot_w64(p_s2 >= thr32 && p_s2 >= 0.0f) != 0;
#endif
    }
    TPS_PHASE {
        double bn = -1.0, bd = 1.0;
        best_b = -1;
        amb = false;
        auto offer = [&](int c, uint32_t lc) {
            const int b = c * jump;
            const double bf = (double)b;
            double dd;
            if (exact53) dd = __builtin_fma(-totf, bf, nf * (double)lc);
            else dd = (double)((int64_t)n * (int64_t)lc - (int64_t)tot * (int64_t)b);
            const double num = dd * dd, den = bf * (nf - bf);
            const double t1 = num * bd, t2 = bn * den;         // num / den  vs  bn / bd
            const double diff = t1 - t2;
            const bool near = __builtin_fabs(diff) <= 1e-13 * t2;   // false while bn < 0
            const bool take = diff >= 0.0;                     // ties -> the later (larger) candidate
            amb = near || (amb && !take);
            if (take) { bn = num; bd = den; best_b = b; }
        };
        if (prefilter) {
#ifdef TPS_EMU
            p_s1 = ps1_keep[tid]; p_lc = plc_keep[tid]; p_c = pc_keep[tid];
#endif
            if (p_c >= 0 && p_s1 >= thr32) offer(p_c, p_lc);
        } else if (a.lc16 && lc_g && c_max - c_min < LCV * NT) {
            // off-chip 16-bit sums: all of the lane's values are requested before the first one is used
            uint32_t lcv[LCV];
            TPS_UNROLL
            for (int i = 0; i < LCV; ++i) {
                const int c = c_min + tid + i * NT;
                lcv[i] = c <= c_max ? g16_load(lc_g, (uint32_t)c) : 0u;
            }
            TPS_UNROLL
            for (int i = 0; i < LCV; ++i) {
                const int c = c_min + tid + i * NT;
                if (c <= c_max) offer(c, tile_sum(c) + lcv[i]);
            }
        } else {
            TPS_NOVEC
            for (int c = c_min + tid; c <= c_max; c += NT) {
                uint32_t lc;
                if (a.lc16) lc = tile_sum(c) + (lc_g ? g16_load(lc_g, (uint32_t)c) : (uint32_t)l.Lc16[c]);
                else lc = lc_g ? g32_load(lc_g, (uint32_t)c) : l.Lc[c];
                offer(c, lc);
            }
        }
        best = best_b >= 0 ? bn / bd : -1.0;
        bits = 0;
        if (best >= 0.0) __builtin_memcpy(&bits, &best, 8);   // non-negative doubles order like integers
#ifdef TPS_EMU
        keep[3 * tid] = best; keep[3 * tid + 1] = amb ? 1.0 : 0.0; keep[3 * tid + 2] = (double)best_b;
#endif
    }
    // wave-wide: the best score, how many candidates float64 cannot separate from it, the largest b holding it
    double m;
    uint32_t ntie = 0;
    int32_t bestb = -1;
#ifdef TPS_EMU
    m = 0.0;
    for (int t = 0; t < NT; ++t) if (keep[3 * t] > m) m = keep[3 * t];
    for (int t = 0; t < NT; ++t) {
        const double bt = keep[3 * t];
        if (bt >= m * (1.0 - 1e-14) && bt >= 0.0) ntie += keep[3 * t + 1] != 0.0 ? 2u : 1u;
        if (bt == m && (int)keep[3 * t + 2] >= 0 && (int)keep[3 * t + 2] > bestb) bestb = (int)keep[3 * t + 2];
    }
#else
    {
        const uint32_t hi = (uint32_t)(bits >> 32);
        const uint32_t mh = wave_max_u32(hi);
        const uint32_t ml = wave_max_u32(hi == mh ? (uint32_t)bits : 0u);
        const uint64_t mbits = ((uint64_t)mh << 32) | ml;
        __builtin_memcpy(&m, &mbits, 8);
        const bool n1 = best >= m * (1.0 - 1e-14) && best >= 0.0;
        ntie = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(n1)) +
               (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(n1 && amb));
        bestb = (int32_t)wave_max_u32((best == m && best_b >= 0) ? (uint32_t)best_b + 1u : 0u) - 1;
    }
#endif
    (void)misc;
    if (crowded) ntie = 2u;                        // the prefilter kept one candidate per lane: a second one that close needs the full comparison
    tie = ntie > 1u;
    if (ntie > 1u) {                               // float noise cannot separate them: exact integers decide
#ifdef TPS_EMU
        ++emu_counter(4);
#endif
        Cand ex = binseg_exact_wg(S_global, n, jump, min_size, xs);
        bkp = ex.b;
        gain = ex.b < 0 ? 0.0 : gain_from((int64_t)ex.d, ex.den, n, n_patterns);
    } else {
        bkp = bestb;
        gain = bkp < 0 ? 0.0 : m / (double)n / ((double)n_patterns * (double)n_patterns);
    }
}

// ------------------------------------------------------------------ the per-read program
// `lds_base` is this wave's LDS slice, `lut` the workgroup's lookup table (already loaded), `r` the read index.  SV = 0 runs the
// generic block/window path, SV > 0 the path specialised for slide == SV; SO = the pattern table
// holds self-overlapping k-mers (only meaningful for SV > 0; the generic path tests it at run time).
// In the device build every lane of the wave executes this function; TPS_PHASE bodies run once
// per lane and TPS_SYNC() is a wave-level fence.  In the emulation TPS_PHASE loops over the 64
// lane ids, so phases run in program order.
// DCLASS (sums-only self-overlap kernels): which self-overlap periods this instantiation carries -- 0 all, 1 periods 2 .. 4,
// 2 periods 5 and 6 (one tile_lc_s<.., CD> per period and partial-block position: the host picks the kernel by ScanArgs::pp_d).
template <int SV, bool SO, bool PAIR = false, bool RAW = true, bool FULL = tile_full_default(SV), int DCLASS = 0>
TPS_DEV void scan_read(const ScanArgs& a, int64_t r, uint32_t* lds_base, uint32_t* lut) {
    constexpr bool M16K = SV != 0 && SO && !RAW;       // sums-only kernels of self-overlap tables: 16-bit table, XT aliased (lut16 / xt_alias)
    // (XM = 2 also for the default kernels: every tile of theirs is a tile_lc_s<.., CD = 0>, which keeps XF / XT in the pad words too)
    constexpr int XM = (SV != 0 && (RAW || ((M16K || !SO) && TPS_LC_TILE != 0 && TPS_XPAD != 0))) ? 2 : 0;
    constexpr bool F16K = SV != 0 && SO && RAW && DCLASS == 3;     // raw rows of a big self-overlap table: 16-bit field-index table (_s*sorh)
    constexpr bool P16K = SV != 0 && !SO && !RAW && PAIR && DCLASS == 4;      // pair-table kernels of k = 5 tables: 16-bit pair + single table (_s*q)
    Lds l_ = SV ? carve_fused<SV ? SV : 5, FULL, XM>(lds_base, lut, a) : carve(lds_base, lut, a);
    if constexpr (P16K) l_.lshift = LUT_M16;
    const Lds l = l_;
    const PatInfo& pat = a.pat;
    const tps_params& prm = a.prm;
    // one 16-byte descriptor per read (wave-uniform: a scalar load)
    const int64_t woff = a.desc[r].word_off;
    const int64_t L = a.desc[r].len;
    const bool has_inv = (a.desc[r].flags & TPS_RD_HAS_INVALID) != 0;
    const bool step1 = (prm.flags & TPS_F_STEP1) != 0;

    const int n1 = (int)(L < prm.no_bp ? L : prm.no_bp);
    const Stage st_s = stage_plan(a.seq2, a.inv, woff, L, false, 0, 0, n1);     // first n1 bases
    const Stage st_e = stage_plan(a.seq2, a.inv, woff, L, true, 0, 0, n1);      // last n1 bases, reversed

    TPS_STAMP(0);
    TPS_PHASE { for (int i = tid; i < MISC_DW; i += NT) l.misc[i] = 0; }
    TPS_SYNC();
    TPS_STAMP(1);
    TPS_PHASE {
        if (step1) {
            // the heads are staged side by side: words [0, head_dw) and [head_dw, 2 head_dw), head_dw / 4 quads each.
            // 1000 bases are at most 17 quads per head: lanes 0-31 take the first head, lanes 32-63 the reversed last
            // one, every lane at most one 16-byte load in the common case -- one HBM round trip for both heads.
            const int hq = a.head_dw >> 2;
            const int side = tid >> 5;
            const Stage& st = side ? st_e : st_s;
            uint32_t* dst = l.seq2 + side * a.head_dw;
            uint32_t* dstv = (uint32_t*)l.val + side * (a.head_dw >> 1);
            for (int c = tid & 31; c < hq; c += 32) {
                u32x4 v;
                u32x2 b;
                v.x = v.y = v.z = v.w = 0;
                b.x = b.y = 0;
                if (c < st.nq) {
                    v = load16(stage_addr(st, c));
                    if (has_inv) b = load8(stage_inv_addr(st, c));
                }
                lds_store16(dst + 4 * c, stage_orient(st, v));
                if (has_inv) {
                    const u32x2 o = stage_orient_inv(st, b);
                    if (o.x | o.y) l.misc[M_INVALID] = 1u;
                    lds_store8(dstv + 2 * c, o);
                }
            }
        }
    }
    TPS_SYNC();

    int tail = 0, pass = 1;
    tps_read_result res;
    res.best_start = res.best_end = 0;
    res.best_start_idx = res.best_end_idx = 0;
    res.n_win = 0; res.bkp = -1; res.gain = 0.0;
    res.flags = 0; res.reserved = 0;

    TPS_STAMP(2);
    if (step1) {
        const bool clean = uniform(l.misc[M_INVALID]) == 0 && pat.dup_mask == 0;
        const bool plain = clean && pat.so_mask == 0;
        bool packed1 = false;
        if constexpr (SV != 0) packed1 = clean && trc_packed_ok(a, st_s.n - pat.k + 1);
        uint32_t ks = 0, ke = 0;
        bool decided = false;
#if !defined(TPS_EMU) && !defined(TPS_NO_DECIDE_FAST)           /* (TPS_NO_DECIDE_FAST: A/B builds) */
        if constexpr (SV != 0 && !SO && !RAW) {
            if (packed1) {
                trc_decide_packed<P16K ? 2 : 0>(a, l, st_s, st_e, r, ks, ke);
                decided = true;
                TPS_STAMP(3);
            }
        }
#endif
        if (decided) {
        } else if (packed1) {
            bool fld = false;
            if constexpr (RAW) fld = a.lut_fields != 0;
            if (fld) {
                if constexpr (RAW) { TPS_PHASE { trc_count_packed<SO, F16K ? 3 : 1>(a, l, st_s, st_e, tid); } }
            } else if (SO) {
                TPS_PHASE { trc_count_packed<true, M16K ? 2 : 0>(a, l, st_s, st_e, tid); }
            } else {
                TPS_PHASE { trc_count_packed<false, P16K ? 2 : 0>(a, l, st_s, st_e, tid); }
            }
            TPS_SYNC();
            if (SO && (uniform(l.misc[M_CMASK]) | uniform(l.misc[M_CMASK + 1]))) {
                TPS_PHASE { trc_publish_occ(a, l, st_s, st_e, tid); }
                TPS_SYNC();
            }
            TPS_STAMP(3);
            TPS_PHASE { trc_sum_packed(a, l, st_s, st_e, r, tid); }
        } else {
            TPS_PHASE { for (int i = tid; i < HIST_DW; i += NT) l.blk[i] = 0; }
            TPS_SYNC();
            if (plain) {
                TPS_PHASE { trc_count_thread<true>(a, l, st_s, st_e, tid); }
            } else {
                TPS_PHASE { trc_count_thread<false>(a, l, st_s, st_e, tid); }
            }
            TPS_SYNC();
            TPS_STAMP(3);
            TPS_PHASE { trc_sum_thread(a, l, st_s, st_e, r, tid); }
        }
        if (!decided) {
            TPS_SYNC();
            ks = uniform(l.misc[M_BEST]);
            ke = uniform(l.misc[M_BEST + 1]);
        }
        res.best_start = (int32_t)(ks >> 5); res.best_start_idx = 31 - (int32_t)(ks & 31u);
        res.best_end = (int32_t)(ke >> 5); res.best_end_idx = 31 - (int32_t)(ke & 31u);
        // forward only if strictly larger (allsteps.py:193); strict cutoff and length tests
        tail = res.best_start > res.best_end ? 0 : 1;
        int best = tail ? res.best_end : res.best_start;
        pass = (L > prm.min_len && best > prm.min_count) ? 1 : 0;
    } else if (a.tails_in) {
        uint8_t tv = a.tails_in[r];
        tail = tv & 1;
        pass = (tv & 2) ? 0 : 1;                   // bit 1 set = skip this read
    }
    res.tail = tail;
    res.pass = pass;
    TPS_STAMP(4);

    int n_win = 0;
    uint64_t s_total = 0;                           // sum of S_w so far (uniform across the wave)
    if (pass && (prm.flags & TPS_F_WINDOWS)) {
        const int64_t m = L < prm.maxlen ? L : prm.maxlen;
        const int64_t n_s = m - prm.trimfirst;
        if (n_s >= prm.window) n_win = (int)((n_s - prm.window) / prm.slide) + 1;
        if (n_win / prm.jump + 1 > a.lc_cap) n_win = 0;   // host plans lc_cap from the longest read
        const int64_t out_base = a.win_off ? a.win_off[r] : 0;
        if constexpr (SV == 0) {
            // ---------------- generic tiles: spans_per_tile spans of span_dw dwords each
            const uint64_t lc_gen = a.lc_global ? (uint64_t)(uintptr_t)(a.lc_scratch + r * (int64_t)a.lc_stride) : 0ull;
            const int blk_per_tile = a.spans_per_tile << a.blk_log2;
            const int tw = blk_per_tile - a.q - 1;     // windows per tile
            for (int w0 = 0; w0 < n_win; w0 += tw) {
                const int nw_tile = (n_win - w0) < tw ? (n_win - w0) : tw;
                const int64_t i0 = (int64_t)w0 * prm.slide;
                int64_t n_stage = n_s - i0;
                const int64_t cap = (int64_t)blk_per_tile * prm.slide + 32;
                if (n_stage > cap) n_stage = cap;
                const Stage st = stage_plan(a.seq2, a.inv, woff, L, tail == 1, prm.trimfirst, i0, (int)n_stage);
                // spans needed for this tile's blocks 0 .. nw_tile-1+q (+ the partial block)
                const int blk_need = nw_tile + a.q + 1;
                const int spans = (blk_need + (1 << a.blk_log2) - 1) >> a.blk_log2;
                // words the spans read: the tile starts up to 63 positions into its first quad, + the look-ahead words
                const int ndw = spans * a.span_dw + 8 < a.seq_dw ? spans * a.span_dw + 8 : a.seq_dw;
                TPS_PHASE { if (tid == 0) l.misc[M_INVALID] = 0; }
                TPS_SYNC();
                TPS_PHASE { stage_thread(st, has_inv, l.seq2, l.val, (ndw + 3) >> 2, &l.misc[M_INVALID], tid); }
                TPS_SYNC();
                if (w0 == 0) TPS_STAMP(5);
                TPS_PHASE {
                    for (int sp = tid; sp < spans; sp += NT) blocks_span(a, l, st.delta, sp);
                }
                TPS_SYNC();
                if (w0 == 0) TPS_STAMP(6);
                wg_exclusive_scan(l.Tot, spans, &l.misc[M_SCAN]);
                if (w0 == 0) TPS_STAMP(7);
                for (int base = 0; base < nw_tile; base += WIN_U * NT) {
                    TPS_PHASE { windows_group(a, l, st.delta, w0, nw_tile, out_base, base, tid); }
                    TPS_SYNC();
                    const uint32_t gsum = wg_exclusive_scan(l.row, WIN_U * NT, &l.misc[M_SCAN]);
                    TPS_PHASE { candidates_group(a, l, lc_gen, w0, nw_tile, base, (uint32_t)s_total, tid); }
                    s_total += gsum;
                    TPS_SYNC();
                }
                if (w0 == 0) TPS_STAMP(8);
            }
        } else {
            // ---------------- fused tiles: NT lanes x 8 blocks
            typedef TileGeo<SV ? SV : 1, FULL> t_;
            constexpr int PF = t_::PF;
            const TileConst tc = tile_const(a, r);
            int tw = (int)uniform((uint32_t)a.tw);     // windows per tile: NT * B - q - 1
            TPS_PIN_S(tw);
            bool pp = false;
            if constexpr (RAW) pp = a.pp_d >= 0 && (SO ? a.pp_d > 0 : (a.pp_d == 0 && a.raw != nullptr));
            bool pp_expect = a.so_fast != 2;       // per-pattern tiles of a self-overlap table: the previous tile held a chain (the first one is expected to)
            auto tile_stage = [&](int w0_) {
                const int64_t i0 = (int64_t)w0_ * prm.slide;
                int64_t n_stage = n_s - i0;
                int64_t cap = (int64_t)t_::BASES;
                // the last window ends r positions into block tw - 1 + q; + its last k-mer, the pair lookup's extra base and
                // the look-ahead of the self-overlap tests (<= 13 bases in all)
                const int64_t need = (int64_t)(tw - 1 + tc.q) * prm.slide + tc.r + 13;
                if (cap > need) cap = need;
                if (n_stage > cap) n_stage = cap;
                return stage_plan(a.seq2, a.inv, woff, L, tail == 1, prm.trimfirst, i0, (int)n_stage);
            };
            // Software prefetch: the 16-byte load(s) of the NEXT tile are issued right after the current tile has been
            // copied into LDS and complete while it is being scanned, so a wave pays the HBM latency once per read
            // instead of once per tile.  One quad (64 bases) per lane covers a tile up to slide 7.
            u32x4 pf[PF];
            u32x2 pv[PF];
#ifdef TPS_EMU
            u32x4 pf_keep[NT][PF];
            u32x2 pv_keep[NT][PF];
#endif
            auto pf_load = [&](const Stage& stn, int tid_) {
                TPS_UNROLL
                for (int u = 0; u < PF; ++u) {
                    const int c = tid_ + u * NT;
                    pf[u].x = pf[u].y = pf[u].z = pf[u].w = 0;
                    pv[u].x = pv[u].y = 0;
                    if (c < stn.nq) {
                        pf[u] = load16(stage_addr(stn, c));
                        if (has_inv) pv[u] = load8(stage_inv_addr(stn, c));
                    }
#ifdef TPS_EMU
                    pf_keep[tid_][u] = pf[u];
                    pv_keep[tid_][u] = pv[u];
#endif
                }
            };
            if (n_win > 0) {
                const Stage st0 = tile_stage(0);
                TPS_PHASE {
                    if (tid == 0) { l.misc[M_INVALID] = 0; l.misc[M_NTIE] = 0; }   // step 1 may have flagged its heads
                    pf_load(st0, tid);
                }
            }
            for (int w0 = 0, tile = 0; w0 < n_win; w0 += tw, ++tile) {
                const int nw_tile = (n_win - w0) < tw ? (n_win - w0) : tw;
                const Stage st = tile_stage(w0);
                // (misc[M_INVALID] / [M_NTIE] are zero here: cleared with the rest of misc at the start of the read and again
                // by the last phase of every tile -- no extra phase and barrier per tile for that)
                TPS_PHASE {
                    TPS_UNROLL
                    for (int u = 0; u < PF; ++u) {
                        const int c = tid + u * NT;
#ifdef TPS_EMU
                        pf[u] = pf_keep[tid][u];
                        pv[u] = pv_keep[tid][u];
#endif
                        if (c < t_::NQ) {              // every quad of the tile buffer is written (zeros past the staged range)
                            lds_store16(l.seq2 + SEQ_LEAD + 4 * c, stage_orient(st, pf[u]));
                            if (has_inv) {
                                const u32x2 o = stage_orient_inv(st, pv[u]);
                                if (o.x | o.y) l.misc[M_INVALID] = 1u;
                                lds_store8((uint32_t*)l.val + (SEQ_LEAD >> 1) + 2 * c, o);
                            }
                        }
                    }
                    if (tid == NT - 1) {               // the lead words: lanes read the base before their first one
                        u32x4 z;
                        z.x = z.y = z.z = z.w = 0;
                        lds_store16(l.seq2, z);
                        if (has_inv) { u32x2 z2; z2.x = z2.y = 0; lds_store8((uint32_t*)l.val, z2); }
                    }
                }
                TPS_SYNC();
                if (w0 + tw < n_win) {
                    const Stage stn = tile_stage(w0 + tw);
                    TPS_PHASE { pf_load(stn, tid); }
                }
                if (w0 == 0) TPS_STAMP(5);
                const int fdelta = st.delta + 16 * SEQ_LEAD;      // LDS position of the tile's first base
                if constexpr (RAW) {
                    // per-pattern tiles: raw counts wanted, or a table with one self-overlap period (exact without recounts)
                    constexpr int SP = SV ? SV : 5;
                    if (pp && uniform(l.misc[M_INVALID]) == 0) {
                        if constexpr (SO) {
                            // tiles without a chained occurrence complete as tiles of a table without self-overlap; a tile that
                            // follows a chained one skips that attempt (round 4: the telomere is a run of tiles at the start of the
                            // scanned tail, and at ONT error rates every telomeric tile at k = 6 holds a chain -- the detecting
                            // pass was phase 1 run twice for them)
                            // (the HOME shape of the kernel -- the period and partial-block position of the reference's human motif at the
                            // default window: CCCTAA at k = 5 (period 4, r = 95 % S; _s*sor) and k = 6 (period 5, r = 94 % S; _s*sorh) -- takes
                            // instantiations with r as a compile-time constant; every other period / window the run-time ones)
                            constexpr int HD = F16K ? 5 : 4, HR = (F16K ? 94 : 95) % SP;
                            const bool home = TPS_PP_RPT != 0 && a.pp_d == HD && tc.r == HR;
                            bool chained = true;
                            if (a.so_fast && !pp_expect) {
                                if (home) chained = tile_pp_s<SP, 0, HD, F16K, HR>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total);
                                else switch (a.pp_d) {
                                    case 2: chained = tile_pp_s<SP, 0, 2, F16K>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total); break;
                                    case 3: chained = tile_pp_s<SP, 0, 3, F16K>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total); break;
                                    case 4: chained = tile_pp_s<SP, 0, 4, F16K>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total); break;
                                    case 5: chained = tile_pp_s<SP, 0, 5, F16K>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total); break;
                                    default: chained = tile_pp_s<SP, 0, (SP < 6 ? SP : 6), F16K>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total); break;
                                }
                            }
                            if (!chained) { pp_expect = false; continue; }
                            const uint64_t s_before = s_total;
                            if (home) tile_pp_s<SP, HD, 0, F16K, HR>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total);
                            else switch (a.pp_d) {
                                case 2: tile_pp_s<SP, 2, 0, F16K>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total); break;
                                case 3: tile_pp_s<SP, 3, 0, F16K>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total); break;
                                case 4: tile_pp_s<SP, 4, 0, F16K>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total); break;
                                case 5: tile_pp_s<SP, 5, 0, F16K>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total); break;
                                default: tile_pp_s<SP, (SP < 6 ? SP : 6), 0, F16K>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total); break;
                            }
                            // the next tile comes straight here if this one looks telomeric (a mean S_w well above the P of a window
                            // without matches; a guess that only decides which exact tile code runs first)
                            pp_expect = a.so_fast != 2 && (s_total - s_before) > (uint64_t)nw_tile * (uint64_t)(pat.P + 4);
                        } else {
                            constexpr int HR4 = 96 % SP;       // k = 4 at the default window
                            if constexpr (PAIR && (HR4 & 1) == 0 && TPS_PP_PAIRF != 0) {
                                if (tc.r == HR4 && a.pair_n != 0) { tile_pp_s<SP, 0, 0, false, HR4, true>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total); continue; }
                            }
                            if (TPS_PP_RPT != 0 && tc.r == HR4) tile_pp_s<SP, 0, 0, false, HR4>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total);
                            else tile_pp_s<SP, 0>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total);
                        }
                        continue;
                    }
                    if (pp && SO) {                    // a tile with non-ACGT letters takes the recount path: the next tile starts blind
                        TPS_PHASE { if (tid == 0) l.misc[M_SCAN + 6] = 1u; }
                        TPS_SYNC();
                    }
                }
                if constexpr (SO && !RAW) {
                    // sums only, a table with one self-overlap period
                    constexpr int SP = SV ? SV : 5;
                    if (a.pp_d > 0 && uniform(l.misc[M_INVALID]) == 0) {
                        // every occurrence counted as in a plain tile, the chains' skipped ones taken back (tile_lc_s<.., CD>)
                        bool done = false;
#define TPS_LC_CD_RP(D_, N) case N: if constexpr (N < SP && D_ <= SP) { tile_lc_s<SP, false, (N < SP ? N : 0), false, false, (D_ <= SP ? D_ : 0)>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total, r); done = true; } break;
#define TPS_LC_CD(D_) case D_: switch (tc.r) { TPS_LC_CD_RP(D_, 0) TPS_LC_CD_RP(D_, 1) TPS_LC_CD_RP(D_, 2) TPS_LC_CD_RP(D_, 3) TPS_LC_CD_RP(D_, 4) TPS_LC_CD_RP(D_, 5) TPS_LC_CD_RP(D_, 6) TPS_LC_CD_RP(D_, 7) default: break; } break;
                        if (a.so_fast) {
                            if constexpr (DCLASS == 1) { switch (a.pp_d) { TPS_LC_CD(2) TPS_LC_CD(3) TPS_LC_CD(4) default: break; } }
                            else if constexpr (DCLASS == 2) { switch (a.pp_d) { TPS_LC_CD(5) TPS_LC_CD(6) default: break; } }
                            else { switch (a.pp_d) { TPS_LC_CD(2) TPS_LC_CD(3) TPS_LC_CD(4) TPS_LC_CD(5) TPS_LC_CD(6) default: break; } }
                        }
#undef TPS_LC_CD
#undef TPS_LC_CD_RP
                        if (done) continue;
                        // (TPS_NO_SO_FAST: the flag-and-recount tile below; so does a tile with non-ACGT letters)
                    }
                }
                constexpr int SF = SV ? SV : 1;
                constexpr bool LC = TPS_LC_TILE != 0 && !SO && !RAW;     // the default kernels: lane-contiguous windows (tile_lc_s)
                if (uniform(l.misc[M_INVALID]) != 0) {
                    if constexpr (LC) tile_lc_s<SF, true, -1, false, false, 0, P16K>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total, r);
                    else tile_fused_s<SF, SO, true, -1, false, RAW>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total, r);
                } else if constexpr (SO || RAW) {
                    // (these kernels reach the plain tile only as a fallback: one instantiation with r read at run time)
                    // (the raw-row kernels' pair table holds FIELDS, for tile_pp_s only: their fallback tile looks positions up one by one)
                    if (tc.r == 0) tile_fused_s<SF, SO, false, 0, PAIR && !RAW, RAW>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total, r);
                    else tile_fused_s<SF, SO, false, -1, PAIR && !RAW, RAW>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total, r);
                } else {
#define TPS_TILE_RP(N) case N: if constexpr (N < SF) { if constexpr (LC) { if ((tc.q & 7) == 0) tile_lc_s<SF, false, (N < SF ? N : 0), PAIR, true, 0, P16K>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total, r); \
                                                                          else tile_lc_s<SF, false, (N < SF ? N : 0), PAIR, false, 0, P16K>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total, r); } \
                                                       else tile_fused_s<SF, SO, false, (N < SF ? N : 0), PAIR, RAW>(a, tc, l, fdelta, w0, tile, nw_tile, out_base, s_total, r); } break;
                    switch (tc.r) {
                        TPS_TILE_RP(0) TPS_TILE_RP(1) TPS_TILE_RP(2) TPS_TILE_RP(3) TPS_TILE_RP(4) TPS_TILE_RP(5) TPS_TILE_RP(6) TPS_TILE_RP(7)
                        TPS_TILE_RP(8) TPS_TILE_RP(9) TPS_TILE_RP(10) TPS_TILE_RP(11)
                        default: break;
                    }
#undef TPS_TILE_RP
                }
                if (w0 == 0) TPS_STAMP(8);
            }
        }
    }
    res.n_win = n_win;
    TPS_STAMP(9);

    if (n_win > 0 && (prm.flags & TPS_F_BINSEG) && binseg_admissible(n_win, prm.jump, prm.min_size)) {
        int bkp;
        double gain;
        bool tie = false;
#ifndef TPS_EMU
        // same wave, same CU: workgroup scope orders this wave's S_w / off-chip candidate-sum stores before its own
        // loads (an agent-scope fence writes the L2 back: measured 5x slower)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#endif
        const uint64_t lc_g = a.lc_global ? (uint64_t)(uintptr_t)(a.lc_scratch + r * (int64_t)a.lc_stride) : 0ull;
        if constexpr (SV != 0)
            binseg_from_lc(a, l, lc_g, (const uint16_t*)(a.sums16 + (a.win_off16 ? a.win_off16[r] : 0)), n_win, s_total, prm.jump, prm.min_size, pat.P,
                           l.misc, l.blk, bkp, gain, tie);
        else
            binseg_from_lc(a, l, lc_g, (const int32_t*)(a.sums + (a.win_off ? a.win_off[r] : 0)), n_win, s_total, prm.jump, prm.min_size, pat.P,
                           l.misc, l.blk, bkp, gain, tie);
        res.bkp = bkp;
        res.gain = gain;
        res.flags = tie ? TPS_RES_TIE : 0u;
    }
    TPS_PHASE { if (tid == 0) a.results[r] = res; }
    TPS_STAMP(10);
}

// standalone Binseg over window sums in global memory (tps_binseg_l2); smem = this wave's slice
constexpr int BINSEG_SMEM_DW = ((XS_DW + 2 + MISC_DW + NT + 3) / 4) * 4;
TPS_DEV void binseg_read(const BinsegArgs& a, int64_t r, uint32_t* smem) {
    const int64_t lo = a.win_off[r];
    const int n = (int)(a.win_off[r + 1] - lo);
    const int32_t* S = a.sums + lo;
    uint32_t* xs = smem;
    uint32_t* misc = smem + ((XS_DW + 1) / 2) * 2;
    uint32_t* bs = misc + MISC_DW;
    int bkp = -1;
    double gain = 0.0;
    bool tie = false;
    if (binseg_admissible(n, a.jump, a.min_size)) {
        binseg_wg(S, n, a.jump, a.min_size, a.n_patterns, bs, misc, xs, bkp, gain, tie);
    }
    TPS_PHASE {
        if (tid == 0) {
            a.bkp[r] = bkp;
            if (a.gain) a.gain[r] = gain;
            if (a.tie) a.tie[r] = tie ? 1 : 0;
        }
    }
}

// ------------------------------------------------------------------ k-mer followers (overview heat map)
// Topsicle/descriptive_plot.py:259-291 (patterns_vs_match_heatmap): in bases [lo, hi) of a read and of its reverse
// complement, the non-overlapping matches of the regex  kmer(.{f})  for every k-mer of the doubled motif: which k-mer,
// and which f bases follow it.  One wave per read.  The pattern table is the scan's own (k-mers, then their complements:
// allsteps.py:104-120): on the read itself the first n_fwd patterns are looked for; on the reverse complement the
// staged tail is the REVERSED read and a k-mer matches there iff its complement matches the reversed bases, i.e.
// patterns n_fwd .. 2 n_fwd - 1 -- no complementing of the sequence, as in the scan.
//   phase 1  every lane looks up positions lane, lane + 64, ... and sets the bit of (pattern, position)
//   phase 2  lane p walks pattern p's bits leftmost-first, a match consumes k + f positions (re.finditer); the picks
//            replace the occurrence bits and go to HBM (one bit per position: the host builds the reference's rows)
//   phase 3  every pick adds one to hist[strand][pattern][code of the f following bases] (bin 4^f: a non-ACGT letter
//            among them; on the reverse complement the codes are complemented: code ^ 2 per base)
// ------------------------------------------------------------------ every m-th window of a scan at a base slide (tps_plan.h: stride_base)
struct StrideArgs {
    const tps_read_result* base_results;   // what the base-slide scan left (tail, pass, step-1 counts; n_win of the base slide)
    const uint16_t* base_sums16;           // its S_w, 16-bit, win_off16 layout
    const int64_t* base_win_off16;
    const uint8_t* base_raw;               // its raw rows (win_off layout of the base slide), or nullptr
    const int64_t* base_win_off;
    const int64_t* win_off;                // n + 1: the layout of the requested slide
    int32_t* sums;                         // S_w of the requested slide (int32, like the generic kernel's)
    uint8_t* raw;                          // raw rows of the requested slide, or nullptr
    tps_read_result* results;
    int64_t n_reads;
    int32_t m, P, n_patterns, jump, min_size, binseg;
    int32_t s16_dw;                        // LDS words per wave for the compacted S_w as 16-bit values (0 = none)
                                           // (raw == nullptr with raw rows wanted: the base scan stored every m-th row itself, ScanArgs::raw_m)
};
// `s16`: this wave's LDS copy of the compacted series (host: StrideArgs::s16_dw words per wave, 0 = the series is too long for LDS and
// the change point reads it back from HBM -- the lane-contiguous chunks of binseg_wg are 20 dependent-latency loads per pass there)
TPS_DEV void stride_read(const StrideArgs& a, int64_t r, uint32_t* smem, uint16_t* s16) {
    tps_read_result res = a.base_results[r];
    const int64_t lo = a.win_off[r];
    const int n = res.n_win > 0 ? (int)(a.win_off[r + 1] - lo) : 0;      // (the base scan had windows <=> this one has: window w is base window w m)
    const int64_t lo16 = a.base_win_off16[r], lob = a.base_win_off[r];
    const int m = a.m, P = a.P;
    TPS_PHASE {
        for (int w = tid; w < n; w += NT) {
            const uint32_t v = a.base_sums16[lo16 + (int64_t)w * m];
            a.sums[lo + w] = (int32_t)v;
            if (s16) s16[w] = (uint16_t)v;
        }
        if (a.raw) {
            if ((P & 3) == 0) {                    // rows of whole dwords (both layouts start dword-aligned then)
                const int pd = P >> 2;
                const uint32_t* src = (const uint32_t*)a.base_raw + lob * pd;
                uint32_t* dst = (uint32_t*)a.raw + lo * pd;
                for (int w = tid; w < n; w += NT) {
                    const uint32_t* sr = src + (int64_t)w * m * pd;
                    uint32_t* dr = dst + (int64_t)w * pd;
                    TPS_NOVEC
                    for (int d = 0; d < pd; ++d) dr[d] = sr[d];
                }
            } else {
                const uint8_t* src = a.base_raw + lob * P;
                uint8_t* dst = a.raw + lo * P;
                for (int w = tid; w < n; w += NT) {
                    const uint8_t* sr = src + (int64_t)w * m * P;
                    uint8_t* dr = dst + (int64_t)w * P;
                    TPS_NOVEC
                    for (int d = 0; d < P; ++d) dr[d] = sr[d];
                }
            }
        }
    }
    TPS_SYNC();
    res.n_win = n;
    res.bkp = -1;
    res.gain = 0.0;
    res.flags = 0;
    if (a.binseg && n > 0 && binseg_admissible(n, a.jump, a.min_size)) {
        uint32_t* xs = smem;
        uint32_t* misc = smem + ((XS_DW + 1) / 2) * 2;
        uint32_t* bs = misc + MISC_DW;
        int bkp = -1;
        double gain = 0.0;
        bool tie = false;
        if (s16) {
            binseg_wg((const uint16_t*)s16, n, a.jump, a.min_size, a.n_patterns, bs, misc, xs, bkp, gain, tie);
        } else {
#ifndef TPS_EMU
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // this wave's own stores above, read back below
#endif
            binseg_wg(a.sums + lo, n, a.jump, a.min_size, a.n_patterns, bs, misc, xs, bkp, gain, tie);
        }
        res.bkp = bkp;
        res.gain = gain;
        res.flags = tie ? TPS_RES_TIE : 0u;
    }
    TPS_PHASE { if (tid == 0) a.results[r] = res; }
}

struct FollowArgs {
    const uint32_t* seq2;
    const uint16_t* inv;
    const tps_read_desc* desc;
    const uint32_t* lut;         // the table in global memory: masks over the pattern list (direct or hashed)
    uint32_t* picks;             // [n_reads][2][n_fwd][pw]
    unsigned long long* hist;    // [2][n_fwd][nbins] or nullptr
    int64_t n_reads;
    PatInfo pat;
    int32_t n_fwd, follow, lo, hi, min_len, pw, nbins;
};
constexpr int FOLLOW_MAX_SPAN = 4096;                      // hi - lo
constexpr int FOLLOW_SEQ_DW = 4 * ((63 + FOLLOW_MAX_SPAN + 63) / 64) + 8;
constexpr int FOLLOW_PW = FOLLOW_MAX_SPAN / 32;
constexpr int FOLLOW_LDS_DW = FOLLOW_SEQ_DW + FOLLOW_SEQ_DW / 2 + 4 + 15 * FOLLOW_PW + 4;   // seq2, val, flag, occurrence / pick bits
#ifdef TPS_EMU
TPS_DEV void hist_add(unsigned long long* p) { *p += 1ull; }
#else
TPS_DEV void hist_add(unsigned long long* p) { atomicAdd(p, 1ull); }
#endif
TPS_DEV void followers_read(const FollowArgs& a, int64_t r, uint32_t* lds) {
    uint32_t* seq2 = lds;
    uint16_t* val = (uint16_t*)(lds + FOLLOW_SEQ_DW);
    uint32_t* flag = lds + FOLLOW_SEQ_DW + FOLLOW_SEQ_DW / 2 + 2;
    uint32_t* occ = flag + 2;
    const PatInfo& pat = a.pat;
    const int64_t woff = a.desc[r].word_off;
    const int64_t L = a.desc[r].len;
    const bool has_inv = (a.desc[r].flags & TPS_RD_HAS_INVALID) != 0;
    if (L <= a.min_len) return;                    // (the host zeroed the picks)
    const int64_t m = L < a.hi ? L : a.hi;
    const int n = (int)(m - a.lo);
    const int need = pat.k + a.follow;
    if (n < need) return;
    const int npos = n - need + 1;
    const int pw = a.pw;
    const uint32_t fmask = a.follow >= 16 ? 0xFFFFFFFFu : ((1u << (2 * a.follow)) - 1u);
    for (int strand = 0; strand < 2; ++strand) {
        const Stage st = stage_plan(a.seq2, a.inv, woff, L, strand == 1, a.lo, 0, n);
        TPS_PHASE {
            for (int i = tid; i < a.n_fwd * pw; i += NT) occ[i] = 0;
            if (tid == 0) flag[0] = 0;
        }
        TPS_SYNC();
        TPS_PHASE { stage_thread(st, has_inv, seq2, val, (st.delta + n + 63 + 64) >> 6, flag, tid); }
        TPS_SYNC();
        const bool any_inv = has_inv && uniform(flag[0]) != 0;
        const uint32_t sel = (1u << a.n_fwd) - 1u;
        TPS_PHASE {
            for (int p = tid; p < npos; p += NT) {
                uint32_t h = h_at(a.lut, 0, seq2, val, pat, st.delta + p, any_inv);
                h = (h >> (strand ? a.n_fwd : 0)) & sel;
                while (h) {
                    const int b = ffs0(h);
                    h &= h - 1;
                    lds_or(&occ[b * pw + (p >> 5)], 1u << (p & 31));
                }
            }
        }
        TPS_SYNC();
        TPS_PHASE {
            if (tid < a.n_fwd) {
                uint32_t* bits = occ + tid * pw;
                uint32_t* out = a.picks + ((r * 2 + strand) * a.n_fwd + tid) * (int64_t)pw;
                int cursor = 0;
                for (int w = 0; w < pw; ++w) {
                    uint32_t mm = bits[w], picked = 0;
                    while (mm) {
                        const int bit = ffs0(mm);
                        mm &= mm - 1;
                        const int pos = 32 * w + bit;
                        if (pos >= cursor) { picked |= 1u << bit; cursor = pos + need; }
                    }
                    bits[w] = picked;
                    out[w] = picked;
                }
            }
        }
        TPS_SYNC();
        if (a.hist) {
            TPS_PHASE {
                for (int i = tid; i < a.n_fwd * pw; i += NT) {
                    uint32_t mm = occ[i];
                    const int pj = i / pw, w = i - pj * pw;
                    while (mm) {
                        const int bit = ffs0(mm);
                        mm &= mm - 1;
                        const int q = st.delta + 32 * w + bit + pat.k;
                        uint32_t code = v_at(seq2, q) & fmask;
                        if (strand) code ^= 0xAAAAAAAAu & fmask;        // complement: A <-> T, C <-> G is code ^ 2
                        const bool bad = any_inv && a.follow > 0 && invalid_at(val, q, a.follow);
                        hist_add(&a.hist[((int64_t)strand * a.n_fwd + pj) * a.nbins + (bad ? (uint32_t)(a.nbins - 1) : code)]);
                    }
                }
            }
            TPS_SYNC();
        }
    }
}

}  // namespace tps
