// tps_device.h -- per-read scan logic of the telomere k-mer scanner (MI355X / gfx950).
//
// One 256-thread workgroup owns one read and runs, in one launch:
//   step 1  TRC counts of the first / reversed-last no_bp bases      (allsteps.py:152-204)
//   step 2  sliding-window k-mer counts S_w of the chosen tail        (allsteps.py:257-297)
//   step 3  single-split l2 change-point on S_w                      (allsteps.py:300-333)
//
// Data flow inside the workgroup (everything between HBM and the result lives in LDS):
//   HBM ASCII bases --16 B/lane coalesced loads--> 2-bit packed tile in LDS (seq2)
//   seq2 --k-mer code per position--> LDS lookup table (4^k masks over the pattern list)
//   per block of `slide` positions: OR of masks (G), OR over the first r positions (Gp),
//   running match counts (C0/C1)  -->  per window: S_w = matches + #patterns absent
//   S_w (u16, LDS) --prefix sums--> exact integer arg-max of the split gain.
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
#else
#define TPS_DEV __device__ __forceinline__
#define TPS_HD __host__ __device__ inline
#define TPS_PHASE for (int tid = (int)threadIdx.x, once_ = 1; once_; once_ = 0)
#define TPS_SYNC() __syncthreads()
#endif

namespace tps {

constexpr int NT = 256;                 // threads per workgroup (4 waves of 64)
constexpr uint32_t FLAG_CONFLICT = 0x80000000u;   // bit 31 of a block mask

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
TPS_DEV void lds_add(uint32_t* p, uint32_t v) { *p += v; }
TPS_DEV void lds_or(uint32_t* p, uint32_t v) { *p |= v; }
struct u32x4 { uint32_t x, y, z, w; };
TPS_DEV u32x4 load16(const uint8_t* p) { return *(const u32x4*)p; }
#else
TPS_DEV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
TPS_DEV uint32_t udot4(uint32_t a, uint32_t b) { return __builtin_amdgcn_udot4(a, b, 0u, false); }
TPS_DEV uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
TPS_DEV int popc(uint32_t x) { return __builtin_popcount(x); }
TPS_DEV int ffs0(uint32_t x) { return __builtin_ctz(x); }
TPS_DEV void lds_add(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
TPS_DEV void lds_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
typedef uint4 u32x4;
TPS_DEV u32x4 load16(const uint8_t* p) { return *reinterpret_cast<const uint4*>(p); }
#endif

// ------------------------------------------------------------------ kernel arguments
struct PatInfo {
    int32_t P, k;
    uint32_t kmask;          // (1 << 2k) - 1
    uint32_t all_mask;       // (1 << P) - 1
    uint32_t so_mask;        // list patterns that can overlap themselves (have a period < k)
    int32_t n_periods;
    int32_t period[8];       // union of those periods d (1 <= d < k)
    uint32_t period_pat[8];  // list patterns having period d
};

struct ScanArgs {
    const uint8_t* bases;        // device; >= 64 readable bytes before and after the data
    const int64_t* offsets;      // n+1
    const uint8_t* tails_in;     // n, or nullptr
    const uint32_t* lut;         // 4^k masks over the pattern list
    tps_read_result* results;    // n
    int32_t* c_start;            // n*P or nullptr
    int32_t* c_end;              // n*P or nullptr
    const int64_t* win_off;      // n+1 (window layout of sums/raw)
    int32_t* sums;               // or nullptr
    uint8_t* raw;                // or nullptr
    int64_t n_reads;
    PatInfo pat;
    tps_params prm;
    // LDS plan (host-computed, see plan_lds() in topsicle_hip.hip)
    int32_t lut_n;               // 4^k
    int32_t seq_dw;              // dwords of seq2 (and of val)
    int32_t nblk_cap;            // blocks per tile = spans_per_tile << blk_log2
    int32_t spans_per_tile;
    int32_t span_dw;             // dwords (16 positions each) one span covers = slide / gcd(slide,16)
    int32_t blk_log2;            // log2(blocks per span), blocks per span = 16 / gcd(slide,16)
    int32_t s_cap;               // capacity of the S array (u16 entries)
    int32_t q, r, lw;            // window = q full blocks + r positions; lw = W - k start positions
};

struct BinsegArgs {
    const int32_t* sums;
    const int64_t* win_off;
    int32_t* bkp;
    double* gain;
    int64_t n_reads;
    int32_t n_patterns, jump, min_size;
};

// ------------------------------------------------------------------ LDS carve
constexpr int MISC_DW = 2560;
struct Lds {
    uint32_t* lut;
    uint32_t* seq2;    // 2-bit packed bases, 16 per dword
    uint32_t* val;     // bit j of val[c] set = position 16c+j is NOT one of acgtACGT
    uint32_t* G;       // per block: OR of masks over its `slide` positions (+FLAG_CONFLICT)
    uint32_t* Gp;      // per block: OR over its first r positions
    uint16_t* C0;      // per block: matches before the block (span-local running count)
    uint16_t* C1;      // per block: matches before position r of the block
    uint32_t* Tot;     // per span: matches in the span, then exclusive prefix over spans
    uint16_t* S;       // window sums of the whole read
    uint32_t* misc;    // hist[32], cmask, flags, scan/reduction scratch
};
TPS_DEV Lds carve(uint32_t* base, const ScanArgs& a) {
    Lds l;
    uint32_t* p = base;
    l.lut = p;  p += a.lut_n;
    l.seq2 = p; p += a.seq_dw;
    l.val = p;  p += a.seq_dw;
    l.G = p;    p += a.nblk_cap;
    l.Gp = p;   p += a.nblk_cap;
    l.C0 = (uint16_t*)p; p += (a.nblk_cap + 1) / 2;
    l.C1 = (uint16_t*)p; p += (a.nblk_cap + 1) / 2;
    l.Tot = p;  p += a.spans_per_tile + 1;
    l.S = (uint16_t*)p;  p += (a.s_cap + 1) / 2;
    l.misc = p;
    return l;
}
TPS_HD int64_t lds_dwords(const ScanArgs& a) {
    return (int64_t)a.lut_n + 2ll * a.seq_dw + 2ll * a.nblk_cap + 2ll * ((a.nblk_cap + 1) / 2) +
           a.spans_per_tile + 1 + (a.s_cap + 1) / 2 + MISC_DW;
}
// misc layout (dwords)
constexpr int M_HIST = 0;        // 32
constexpr int M_CMASK = 32;      // conflict mask of step 1
constexpr int M_INVALID = 33;    // any non-ACGT base in the staged range
constexpr int M_BEST = 34;       // best_start, idx, best_end, idx  (4)
constexpr int M_TAIL = 38;       // tail, pass
constexpr int M_Q = 64;          // 64 dwords: scan partials
constexpr int M_BS = 128;        // NT dwords: binseg chunk sums
constexpr int M_CD = 384;        // NT x u64 : candidate |D|      (512 dwords)
constexpr int M_CDEN = 896;      // NT x u64 : candidate b(n-b)   (512 dwords)
constexpr int M_CB = 1408;       // NT x i32 : candidate b
constexpr int M_R = 1664;        // 16 x (u64,u64,i32) second-level reduction (reserve 128)

// ------------------------------------------------------------------ staging: HBM ASCII -> LDS 2-bit
// Stages s-indices [i0, i0+n) of a tail string into LDS.  Forward tail: s[i] = seq[t + i];
// reverse tail: s[i] = seq[L-1-t-i] (allsteps.py:267-271, 176-177).  LDS position of s-index
// i0 is `delta` (0..15): global 16-byte chunks are loaded whole and aligned, chunk c -> dword c.
struct Stage {
    const uint8_t* chunk0;   // address of chunk 0 (forward: ascending, reverse: descending by 16)
    int32_t delta;
    int32_t nch;             // chunks that hold staged data
    int32_t n;               // staged s-indices
    bool reverse;
};
TPS_DEV Stage stage_plan(const uint8_t* seq, int64_t L, bool reverse, int64_t t, int64_t i0, int32_t n) {
    Stage st;
    st.reverse = reverse;
    st.n = n;
    if (!reverse) {
        uintptr_t a0 = (uintptr_t)(seq + t + i0);
        uintptr_t lo = a0 & ~(uintptr_t)15;
        st.delta = (int32_t)(a0 - lo);
        st.chunk0 = (const uint8_t*)lo;
    } else {
        uintptr_t e = (uintptr_t)(seq + (L - 1 - t - i0));
        uintptr_t top = e | (uintptr_t)15;
        st.delta = (int32_t)(top - e);
        st.chunk0 = (const uint8_t*)(top - 15);
    }
    st.nch = n > 0 ? (st.delta + n + 15) >> 4 : 0;
    return st;
}

TPS_DEV uint32_t bad_bits16(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, bool reverse) {
    uint32_t w[4] = {b0, b1, b2, b3};
    uint32_t m = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if ((w[i] >> (8 * j)) & 255u) m |= 1u << (4 * i + j);
    if (reverse) {      // position j <-> byte 15-j
        uint32_t rv = 0;
        for (int j = 0; j < 16; ++j) rv |= ((m >> j) & 1u) << (15 - j);
        m = rv;
    }
    return m;
}

// one thread stages chunks tid, tid+NT, ...
TPS_DEV void stage_thread(const Stage& st, const Lds& l, int seq_dw, int tid) {
    const uint32_t wfwd = 0x40100401u, wrev = 0x01041040u;
    const uint32_t w = st.reverse ? wrev : wfwd;
    for (int c = tid; c < seq_dw; c += NT) {
        uint32_t packed = 0, bad = 0;
        if (c < st.nch) {
            const uint8_t* p = st.reverse ? st.chunk0 - 16 * (intptr_t)c : st.chunk0 + 16 * (intptr_t)c;
            u32x4 v = load16(p);
            uint32_t y0 = v.x & 0x06060606u, y1 = v.y & 0x06060606u, y2 = v.z & 0x06060606u, y3 = v.w & 0x06060606u;
            uint32_t d0 = udot4(y0, w), d1 = udot4(y1, w), d2 = udot4(y2, w), d3 = udot4(y3, w);
            // d_i = 2 * (8-bit packed codes of 4 bases); halve per 16-bit half so bit 32 is never needed
            uint32_t lo2 = st.reverse ? (d3 + (d2 << 8)) : (d0 + (d1 << 8));
            uint32_t hi2 = st.reverse ? (d1 + (d0 << 8)) : (d2 + (d3 << 8));
            packed = (lo2 >> 1) | ((hi2 >> 1) << 16);
            // expected lower-case letter for each 2-bit code: selector 0,2,4,6 -> a,c,t,g
            const uint32_t s1 = 0x00630061u, s0 = 0x00670074u;
            uint32_t b0 = (v.x | 0x20202020u) ^ perm(s0, s1, y0);
            uint32_t b1 = (v.y | 0x20202020u) ^ perm(s0, s1, y1);
            uint32_t b2 = (v.z | 0x20202020u) ^ perm(s0, s1, y2);
            uint32_t b3 = (v.w | 0x20202020u) ^ perm(s0, s1, y3);
            if (b0 | b1 | b2 | b3) {
                bad = bad_bits16(b0, b1, b2, b3, st.reverse);
                // keep only positions inside the staged range [delta, delta+n)
                int lo = st.delta - 16 * c, hi = st.delta + st.n - 16 * c;
                uint32_t keep = 0xFFFFu;
                if (lo > 0) keep &= (lo >= 16) ? 0u : (0xFFFFu << lo);
                if (hi < 16) keep &= (hi <= 0) ? 0u : ((1u << hi) - 1u);
                bad &= keep & 0xFFFFu;
                if (bad) l.misc[M_INVALID] = 1u;      // benign race: every writer stores 1
            }
        }
        l.seq2[c] = packed;
        l.val[c] = bad;
    }
}

// 32-bit window (16 bases) starting at LDS position q
TPS_DEV uint32_t v_at(const Lds& l, int q) {
    int idx = q >> 4;
    return alignbit(l.seq2[idx + 1], l.seq2[idx], (uint32_t)(q & 15) * 2u);
}
// 1 if any of the k positions q..q+k-1 is not ACGT
TPS_DEV bool invalid_at(const Lds& l, int q, int k) {
    int idx = q >> 4;
    uint64_t v = (uint64_t)l.val[idx] | ((uint64_t)l.val[idx + 1] << 16) | ((uint64_t)l.val[idx + 2] << 32);
    return ((v >> (q & 15)) & ((1ull << k) - 1ull)) != 0;
}
// mask of list patterns whose k-mer starts at LDS position q
TPS_DEV uint32_t h_at(const Lds& l, const PatInfo& pat, int q, bool any_invalid) {
    uint32_t h = l.lut[v_at(l, q) & pat.kmask];
    if (any_invalid && h && invalid_at(l, q, pat.k)) h = 0;
    return h;
}
// patterns p (subset of `h`) that occur again d < k positions later (d a period of p)
TPS_DEV uint32_t conflict_bits(const PatInfo& pat, uint32_t v, uint32_t h) {
    uint32_t c = 0;
    for (int i = 0; i < pat.n_periods; ++i) {
        uint32_t hp = h & pat.period_pat[i];
        if (hp && (((v ^ (v >> (2 * pat.period[i]))) & pat.kmask) == 0)) c |= hp;
    }
    return c;
}

// Leftmost non-overlapping count of list pattern `bit` over `npos` start positions from LDS
// position q0 -- exactly what len(list(re.finditer(p, text))) gives (allsteps.py:182, 281).
TPS_DEV void greedy_count(const Lds& l, const PatInfo& pat, int q0, int npos, int bit, bool any_invalid,
                          int& occ, int& greedy) {
    occ = 0; greedy = 0;
    int cursor = 0;
    for (int p = 0; p < npos; ++p) {
        uint32_t h = h_at(l, pat, q0 + p, any_invalid);
        if ((h >> bit) & 1u) {
            ++occ;
            if (p >= cursor) { ++greedy; cursor = p + pat.k; }
        }
    }
}

// ------------------------------------------------------------------ step 1: TRC counts of one tail
// Counts every list pattern over the staged string of n1 bases (LDS positions delta..).
// hist[] must be zeroed; phases are separated by the caller's TPS_SYNC().
TPS_DEV void trc_count_thread(const Lds& l, const PatInfo& pat, const Stage& st, int tid) {
    const bool inv = l.misc[M_INVALID] != 0;
    const int npos = st.n - pat.k + 1;
    for (int p = tid; p < npos; p += NT) {
        uint32_t v = v_at(l, st.delta + p);
        uint32_t h = l.lut[v & pat.kmask];
        if (!h) continue;
        if (inv && invalid_at(l, st.delta + p, pat.k)) continue;
        if (h & pat.so_mask) {
            uint32_t c = conflict_bits(pat, v, h);
            if (c) lds_or(&l.misc[M_CMASK], c);
        }
        while (h) {
            int b = ffs0(h);
            h &= h - 1;
            lds_add(&l.misc[M_HIST + b], 1u);
        }
    }
}
TPS_DEV void trc_fix_thread(const Lds& l, const PatInfo& pat, const Stage& st, int tid) {
    // patterns with overlapping occurrences: recount leftmost-non-overlapping, sequentially
    if (tid < pat.P && ((l.misc[M_CMASK] >> tid) & 1u)) {
        int occ, g;
        greedy_count(l, pat, st.delta, st.n - pat.k + 1, tid, l.misc[M_INVALID] != 0, occ, g);
        l.misc[M_HIST + tid] = (uint32_t)g;
    }
}

// ------------------------------------------------------------------ step 2, phase B: blocks
// Thread `tid` owns span `span` of the tile: span_dw dwords = (1<<blk_log2) blocks of `slide`
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
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v[i] = i ? alignbit(nxt, cur, 2u * i) : cur;
            h[i] = l.lut[v[i] & pat.kmask];
        }
        if (inv) {
            int q = ((d0 + dw) << 4) + (delta & 15);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (h[i] && invalid_at(l, q + i, k)) h[i] = 0;
        }
        uint32_t cf = 0;                          // bit i: position i starts an overlapping pair
        if (so) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if ((h[i] & pat.so_mask) && conflict_bits(pat, v[i], h[i])) cf |= 1u << i;
        }
#pragma unroll
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

// ------------------------------------------------------------------ step 2, phase C: windows
// Exact count of every pattern in tile-local window wl (slow path: overlapping occurrences of a
// self-overlapping k-mer inside the window, or raw output requested).
TPS_DEV uint32_t window_exact(const ScanArgs& a, const Lds& l, int delta, int wl, uint32_t present,
                              uint8_t* raw_row) {
    const PatInfo& pat = a.pat;
    const bool inv = l.misc[M_INVALID] != 0;
    uint32_t sum = 0;
    for (int b = 0; b < pat.P; ++b) {
        int c = 0;
        if ((present >> b) & 1u) {
            int occ;
            greedy_count(l, pat, delta + wl * a.prm.slide, a.lw, b, inv, occ, c);
        }
        if (c == 0) c = 1;                         // `matches or 1` (allsteps.py:281, 288)
        if (raw_row) raw_row[b] = (uint8_t)c;
        sum += (uint32_t)c;
    }
    return sum;
}

TPS_DEV void windows_thread(const ScanArgs& a, const Lds& l, int delta, int w0, int nw_tile,
                            int64_t out_base, int tid) {
    const PatInfo& pat = a.pat;
    const int q = a.q;
    for (int wl = tid; wl < nw_tile; wl += NT) {
        uint32_t m = l.Gp[wl + q];
        for (int i = 0; i < q; ++i) m |= l.G[wl + i];
        int je = wl + q;
        uint32_t cnt = ((uint32_t)l.C1[je] + l.Tot[je >> a.blk_log2]) - ((uint32_t)l.C0[wl] + l.Tot[wl >> a.blk_log2]);
        uint32_t present = m & pat.all_mask;
        uint32_t sw = cnt + (uint32_t)(pat.P - popc(present));
        uint8_t* raw_row = a.raw ? a.raw + (out_base + w0 + wl) * (int64_t)pat.P : nullptr;
        if ((m & FLAG_CONFLICT) || raw_row) sw = window_exact(a, l, delta, wl, present, raw_row);
        l.S[w0 + wl] = (uint16_t)sw;
        if (a.sums) a.sums[out_base + w0 + wl] = (int32_t)sw;
    }
}

// ------------------------------------------------------------------ step 3: single-split Binseg (l2)
// gain(b) = cost(0,n) - cost(0,b) - cost(b,n) = (n L_b - T b)^2 / (n b (n-b))   [y = S / P]
// so the arg-max over b in {0, jump, 2 jump, ...}, b >= min_size, n-b >= min_size is decided in
// exact integer arithmetic on D_b = n L_b - T b (ties -> larger b, as max() over (gain, bkp)
// tuples does in ruptures' Binseg._single_bkp).
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
TPS_DEV double cand_gain(const Cand& c, int n, int n_patterns) {
    if (c.b < 0) return 0.0;
    double d = (double)c.d;
    return d * d / ((double)n * (double)c.den) / ((double)n_patterns * (double)n_patterns);
}

// Workgroup-wide Binseg over S[0..n): phases separated by TPS_SYNC(); scratch in misc.
#define TPS_BINSEG_BODY(S, n, jump, min_size, misc, RESULT)                                        \
    {                                                                                              \
        const int cl_ = ((n) + tps::NT - 1) / tps::NT;                                             \
        TPS_PHASE {                                                                                \
            int lo_ = tid * cl_, hi_ = lo_ + cl_ < (n) ? lo_ + cl_ : (n);                          \
            uint32_t s_ = 0;                                                                       \
            for (int i_ = lo_; i_ < hi_; ++i_) s_ += (uint32_t)(S)[i_];                            \
            (misc)[tps::M_BS + tid] = s_;                                                          \
        }                                                                                          \
        TPS_SYNC();                                                                                \
        TPS_PHASE {                                                                                \
            if (tid < 16) {                                                                        \
                uint32_t s_ = 0;                                                                   \
                for (int i_ = 0; i_ < 16; ++i_) s_ += (misc)[tps::M_BS + tid * 16 + i_];           \
                (misc)[tps::M_Q + tid] = s_;                                                       \
            }                                                                                      \
        }                                                                                          \
        TPS_SYNC();                                                                                \
        TPS_PHASE {                                                                                \
            if (tid == 0) {                                                                        \
                uint32_t run_ = 0;                                                                 \
                for (int i_ = 0; i_ < 16; ++i_) { uint32_t t_ = (misc)[tps::M_Q + i_]; (misc)[tps::M_Q + i_] = run_; run_ += t_; } \
                (misc)[tps::M_Q + 16] = run_;                                                      \
            }                                                                                      \
        }                                                                                          \
        TPS_SYNC();                                                                                \
        TPS_PHASE {                                                                                \
            uint64_t pre_ = (misc)[tps::M_Q + (tid >> 4)];                                         \
            for (int i_ = (tid & ~15); i_ < tid; ++i_) pre_ += (misc)[tps::M_BS + i_];             \
            int lo_ = tid * cl_, hi_ = lo_ + cl_ < (n) ? lo_ + cl_ : (n);                          \
            tps::Cand c_ = tps::binseg_chunk((S), (n), (jump), (min_size), lo_, hi_, pre_, (uint64_t)(misc)[tps::M_Q + 16]); \
            ((uint64_t*)&(misc)[tps::M_CD])[tid] = c_.d;                                           \
            ((uint64_t*)&(misc)[tps::M_CDEN])[tid] = c_.den;                                       \
            ((int32_t*)&(misc)[tps::M_CB])[tid] = c_.b;                                            \
        }                                                                                          \
        TPS_SYNC();                                                                                \
        TPS_PHASE {                                                                                \
            if (tid < 16) {                                                                        \
                tps::Cand b_{0, 1, -1};                                                            \
                for (int i_ = 0; i_ < 16; ++i_) {                                                  \
                    int t_ = tid * 16 + i_;                                                        \
                    tps::Cand c_{((uint64_t*)&(misc)[tps::M_CD])[t_], ((uint64_t*)&(misc)[tps::M_CDEN])[t_], ((int32_t*)&(misc)[tps::M_CB])[t_]}; \
                    if (tps::cand_better(c_, b_)) b_ = c_;                                         \
                }                                                                                  \
                ((uint64_t*)&(misc)[tps::M_R])[tid] = b_.d;                                        \
                ((uint64_t*)&(misc)[tps::M_R + 32])[tid] = b_.den;                                 \
                ((int32_t*)&(misc)[tps::M_R + 64])[tid] = b_.b;                                    \
            }                                                                                      \
        }                                                                                          \
        TPS_SYNC();                                                                                \
        {                                                                                          \
            tps::Cand b_{0, 1, -1};                                                                \
            for (int i_ = 0; i_ < 16; ++i_) {                                                      \
                tps::Cand c_{((uint64_t*)&(misc)[tps::M_R])[i_], ((uint64_t*)&(misc)[tps::M_R + 32])[i_], ((int32_t*)&(misc)[tps::M_R + 64])[i_]}; \
                if (tps::cand_better(c_, b_)) b_ = c_;                                             \
            }                                                                                      \
            RESULT = b_;                                                                           \
        }                                                                                          \
        TPS_SYNC();                                                                                \
    }

// ------------------------------------------------------------------ the per-read program
// `lds_base` is the workgroup's LDS (dynamic shared memory); `r` the read index.
// In the device build every thread of the workgroup executes this function; TPS_PHASE bodies
// run once per thread and TPS_SYNC() is __syncthreads().  In the emulation TPS_PHASE loops
// over the 256 thread ids, so phases run in program order.
TPS_DEV void scan_read(const ScanArgs& a, int64_t r, uint32_t* lds_base) {
    const Lds l = carve(lds_base, a);
    const PatInfo& pat = a.pat;
    const tps_params& prm = a.prm;
    const int64_t off = a.offsets[r];
    const int64_t L = a.offsets[r + 1] - off;
    const uint8_t* seq = a.bases + off;

    TPS_PHASE {
        for (int i = tid; i < a.lut_n; i += NT) l.lut[i] = a.lut[i];
        if (tid < 64) l.misc[tid] = 0;
    }
    TPS_SYNC();

    int tail = 0, pass = 1;
    tps_read_result res;
    res.best_start = res.best_end = 0;
    res.best_start_idx = res.best_end_idx = 0;
    res.n_win = 0; res.bkp = -1; res.gain = 0.0;

    if (prm.flags & TPS_F_STEP1) {
        const int n1 = (int)(L < prm.no_bp ? L : prm.no_bp);
        for (int side = 0; side < 2; ++side) {
            const Stage st = stage_plan(seq, L, side == 1, 0, 0, n1);
            TPS_PHASE {
                if (tid < 34) l.misc[tid] = 0;           // hist, cmask, invalid
            }
            TPS_SYNC();
            TPS_PHASE { stage_thread(st, l, a.seq_dw, tid); }
            TPS_SYNC();
            TPS_PHASE { trc_count_thread(l, pat, st, tid); }
            TPS_SYNC();
            if (l.misc[M_CMASK]) {
                TPS_PHASE { trc_fix_thread(l, pat, st, tid); }
                TPS_SYNC();
            }
            TPS_PHASE {
                if (tid < pat.P) {
                    int32_t* dst = side ? a.c_end : a.c_start;
                    if (dst) dst[r * pat.P + tid] = (int32_t)l.misc[M_HIST + tid];
                }
                if (tid == 0) {
                    uint32_t best = 0; int idx = 0;
                    for (int p = 0; p < pat.P; ++p)
                        if (l.misc[M_HIST + p] > best) { best = l.misc[M_HIST + p]; idx = p; }
                    l.misc[M_BEST + 2 * side] = best;
                    l.misc[M_BEST + 2 * side + 1] = (uint32_t)idx;
                }
            }
            TPS_SYNC();
        }
        res.best_start = (int32_t)l.misc[M_BEST];
        res.best_start_idx = (int32_t)l.misc[M_BEST + 1];
        res.best_end = (int32_t)l.misc[M_BEST + 2];
        res.best_end_idx = (int32_t)l.misc[M_BEST + 3];
        // forward only if strictly larger (allsteps.py:193); strict cutoff and length tests
        tail = res.best_start > res.best_end ? 0 : 1;
        int best = tail ? res.best_end : res.best_start;
        pass = (L > prm.min_len && best > prm.min_count) ? 1 : 0;
    } else if (a.tails_in) {
        uint8_t tv = a.tails_in[r];
        tail = tv & 1;
        pass = (tv & 2) ? 0 : 1;                          // bit 1 set = skip this read
    }
    res.tail = tail;
    res.pass = pass;

    int n_win = 0;
    if (pass && (prm.flags & TPS_F_WINDOWS)) {
        const int64_t m = L < prm.maxlen ? L : prm.maxlen;
        const int64_t n_s = m - prm.trimfirst;
        if (n_s >= prm.window) n_win = (int)((n_s - prm.window) / prm.slide) + 1;
        if (n_win > a.s_cap) n_win = 0;                   // host plans s_cap >= max n_win
        const int blk_per_tile = a.spans_per_tile << a.blk_log2;
        const int tw = blk_per_tile - a.q - 1;            // windows per tile
        const int64_t out_base = a.win_off ? a.win_off[r] : 0;
        for (int w0 = 0; w0 < n_win; w0 += tw) {
            const int nw_tile = (n_win - w0) < tw ? (n_win - w0) : tw;
            const int64_t i0 = (int64_t)w0 * prm.slide;
            int64_t n_stage = n_s - i0;
            const int64_t cap = (int64_t)blk_per_tile * prm.slide + 32;
            if (n_stage > cap) n_stage = cap;
            const Stage st = stage_plan(seq, L, tail == 1, prm.trimfirst, i0, (int)n_stage);
            TPS_PHASE { if (tid == 0) l.misc[M_INVALID] = 0; }
            TPS_SYNC();
            TPS_PHASE { stage_thread(st, l, a.seq_dw, tid); }
            TPS_SYNC();
            // spans needed for this tile's blocks 0 .. nw_tile-1+q (+ the partial block)
            const int blk_need = nw_tile + a.q + 1;
            const int spans = (blk_need + (1 << a.blk_log2) - 1) >> a.blk_log2;
            TPS_PHASE {
                for (int sp = tid; sp < spans; sp += NT) blocks_span(a, l, st.delta, sp);
            }
            TPS_SYNC();
            // exclusive prefix of Tot over spans (two-level; spans <= NT)
            const int per = (spans + 63) / 64;
            TPS_PHASE {
                if (tid < 64) {
                    uint32_t s_ = 0;
                    for (int i = tid * per; i < (tid + 1) * per && i < spans; ++i) s_ += l.Tot[i];
                    l.misc[M_Q + tid] = s_;
                }
            }
            TPS_SYNC();
            TPS_PHASE {
                if (tid == 0) {
                    uint32_t run = 0;
                    for (int i = 0; i < 64; ++i) { uint32_t t_ = l.misc[M_Q + i]; l.misc[M_Q + i] = run; run += t_; }
                }
            }
            TPS_SYNC();
            TPS_PHASE {
                if (tid < 64) {
                    uint32_t run = l.misc[M_Q + tid];
                    for (int i = tid * per; i < (tid + 1) * per && i < spans; ++i) { uint32_t t_ = l.Tot[i]; l.Tot[i] = run; run += t_; }
                }
            }
            TPS_SYNC();
            TPS_PHASE { windows_thread(a, l, st.delta, w0, nw_tile, out_base, tid); }
            TPS_SYNC();
        }
    }
    res.n_win = n_win;

    if (n_win > 0 && (prm.flags & TPS_F_BINSEG) && binseg_admissible(n_win, prm.jump, prm.min_size)) {
        Cand best;
        TPS_BINSEG_BODY(l.S, n_win, prm.jump, prm.min_size, l.misc, best);
        res.bkp = best.b;
        res.gain = cand_gain(best, n_win, pat.P);
    }
    TPS_PHASE { if (tid == 0) a.results[r] = res; }
}

// standalone Binseg over window sums in global memory (tps_binseg_l2)
TPS_DEV void binseg_read(const BinsegArgs& a, int64_t r, uint32_t* misc) {
    const int64_t lo = a.win_off[r];
    const int n = (int)(a.win_off[r + 1] - lo);
    const int32_t* S = a.sums + lo;
    Cand best{0, 1, -1};
    if (binseg_admissible(n, a.jump, a.min_size)) {
        TPS_BINSEG_BODY(S, n, a.jump, a.min_size, misc, best);
    }
    TPS_PHASE {
        if (tid == 0) {
            a.bkp[r] = best.b;
            if (a.gain) a.gain[r] = cand_gain(best, n, a.n_patterns);
        }
    }
}

}  // namespace tps
