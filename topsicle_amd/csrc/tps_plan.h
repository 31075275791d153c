// tps_plan.h -- host-side planning shared by the library (topsicle_hip.hip) and the test
// emulation (tests/emu): pattern table -> lookup table + self-overlap info, and the LDS
// geometry of one scan.  Plain C++, no HIP calls.
#pragma once
#include <cstdlib>
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "tps_device.h"

namespace tps {

inline int64_t window_count(int64_t L, int W, int s, int t, int M) {
    // seq_cut_windows over seq[t:min(L,M)]: range(0, n_s - W + 1, s)   (allsteps.py:219, 263-271)
    int64_t m = std::min<int64_t>(L, M);
    int64_t ns = m - t;
    if (W < 1 || s < 1 || ns < W) return 0;
    return (ns - W) / s + 1;
}

// Which read each wave slot of a launch takes (ScanArgs::order).  A workgroup's LDS and wave slots are free again when the LAST of
// its reads ends, and a launch ends with its longest read: reads are dispatched in classes of equal work -- 64 classes of the
// windows a read will have scanned (none if the length filter drops it: such a read leaves after step 1) -- the class of the longest
// reads first, file order inside a class (a counting sort: O(n)).  Measured on one MI355X (scripts/order_probe.py, 20 000 reads of
// log-normal length, median 11 kb): k = 4 sums 88.8 -> 71.1 us per launch, k = 6 sums 166.1 -> 131.1, k = 4 with raw rows 179.2 -> 139.4;
// a batch of equal reads has one class and keeps file order (order.clear(): no indirection in the kernel).
// n_win[i] = windows of read i (window_count), passes[i] = the read is longer than min_len.
inline void plan_dispatch_order(const int64_t* n_win, const uint8_t* passes, int64_t n, std::vector<int32_t>& order) {
    order.clear();
    if (n < 2 || n > 0x7FFFFFFFll) return;
    constexpr int NCLS = 64;
    int64_t mx = 0;
    for (int64_t i = 0; i < n; ++i) if (passes[i] && n_win[i] > mx) mx = n_win[i];
    if (mx == 0) return;
    // 0: step 1 only; 1 .. NCLS - 1 by windows, rounded UP: the top class holds the reads within 1 / 62 of the longest
    auto cls = [&](int64_t i) -> int { return passes[i] ? (int)(1 + (n_win[i] * (NCLS - 2) + mx - 1) / mx) : 0; };
    int64_t cnt[NCLS + 1] = {0};
    for (int64_t i = 0; i < n; ++i) ++cnt[cls(i)];
    int used = 0;
    for (int c = 0; c < NCLS; ++c) used += cnt[c] != 0;
    if (used < 2) return;
    int64_t start[NCLS];
    int64_t acc = 0;
    for (int c = NCLS - 1; c >= 0; --c) { start[c] = acc; acc += cnt[c]; }
    order.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) order[(size_t)start[cls(i)]++] = (int32_t)i;
}

// Device layout of the fused kernels' 16-bit window sums (ScanArgs::sums16): a read with nw windows owns this many slots, so
// that every read's region starts 16-byte aligned and the dword that holds an odd last window ends in padding.
inline int64_t sums16_slots(int64_t nw) { return (nw + 7) & ~7ll; }

inline int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

// Pattern list (P strings of k letters, reference order) -> 4^k lookup table of list masks and
// the self-overlap (period) information.  Returns "" or an error message.
inline std::string build_patterns(const char* pats, int P, int k, std::vector<uint32_t>& lut, PatInfo& pi) {
    if (k < 1 || k > TPS_MAX_K) return "k=" + std::to_string(k) + " not supported (1.." + std::to_string(TPS_MAX_K) + ")";
    if (P < 1 || P > TPS_MAX_PATTERNS) return std::to_string(P) + " patterns not supported (1.." + std::to_string(TPS_MAX_PATTERNS) + ")";
    const bool hashed = k > TPS_DIRECT_K;
    if (!hashed) lut.assign((size_t)1 << (2 * k), 0u);
    pi = PatInfo{};
    pi.P = P;
    pi.k = k;
    pi.kmask = (1u << (2 * k)) - 1u;
    pi.all_mask = (1u << P) - 1u;
    uint32_t per_pat[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<uint32_t> code_mask((size_t)P, 0u);
    std::vector<uint32_t> codes((size_t)P);
    for (int p = 0; p < P; ++p) {
        uint32_t code = 0;
        char up[16];
        for (int i = 0; i < k; ++i) {
            char ch = pats[p * k + i];
            if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
            up[i] = ch;
            uint32_t v;
            switch (ch) {          // the 2-bit code the kernel derives from ASCII: (c >> 1) & 3
                case 'A': v = 0; break;
                case 'C': v = 1; break;
                case 'T': v = 2; break;
                case 'G': v = 3; break;
                default: return "pattern " + std::to_string(p) + " has a non-ACGT letter";
            }
            code |= v << (2 * i);
        }
        if (!hashed) lut[code] |= 1u << p;
        codes[(size_t)p] = code;
        for (int d = 1; d < k; ++d) {             // proper periods -> the k-mer can overlap itself
            bool periodic = true;
            for (int i = 0; i + d < k; ++i) periodic = periodic && (up[i] == up[i + d]);
            if (periodic) { per_pat[d] |= 1u << p; pi.so_mask |= 1u << p; }
        }
    }
    for (int p = 0; p < P; ++p)                   // mask of all list patterns sharing pattern p's k-mer
        for (int o = 0; o < P; ++o)
            if (codes[(size_t)o] == codes[(size_t)p]) code_mask[(size_t)p] |= 1u << o;
    for (int p = 0; p < P; ++p)
        if (code_mask[(size_t)p] & (code_mask[(size_t)p] - 1)) pi.dup_mask |= 1u << p;
    if (hashed) {
        // perfect hash of the distinct codes into 256 (key, mask) slots: try odd multipliers until none collide
        const int log2h = 8;
        const uint32_t H = 1u << log2h;
        uint32_t mul = 0;
        for (uint32_t cand = 0x9E3779B1u, tries = 0; tries < 200000 && !mul; ++tries, cand += 0xC657CB56u) {
            const uint32_t m = cand | 1u;
            std::vector<int> owner(H, -1);
            bool ok = true;
            for (int p = 0; p < P && ok; ++p) {
                const uint32_t slot = (codes[(size_t)p] * m) >> (32 - log2h);
                if (owner[slot] >= 0 && codes[(size_t)owner[slot]] != codes[(size_t)p]) ok = false;
                else owner[slot] = p;
            }
            if (ok) mul = m;
        }
        if (!mul) return "no collision-free hash for this pattern table";
        lut.assign((size_t)2 * H, 0u);
        for (uint32_t i = 0; i < H; ++i) lut[2 * i] = 0xFFFFFFFFu;           // no k-mer code has all bits set (k <= 15)
        for (int p = 0; p < P; ++p) {
            const uint32_t slot = (codes[(size_t)p] * mul) >> (32 - log2h);
            lut[2 * slot] = codes[(size_t)p];
            lut[2 * slot + 1] = code_mask[(size_t)p];
        }
        pi.hash_mul = mul;
        pi.hash_shift = 32 - log2h;
    }
    for (int d = 1; d < k; ++d)
        if (per_pat[d]) {
            if (pi.n_periods == 8) return "pattern table has more than 8 distinct self-overlap periods";
            pi.period[pi.n_periods] = d;
            pi.period_pat[pi.n_periods] = per_pat[d];
            ++pi.n_periods;
        }
    return "";
}

// = TileGeo<S, true>::SEQ (tps_device.h): LDS words of a fused tile's packed bases
inline int fused_seq_dw(int slide) {
    const int bases = NT * 8 * slide + 13 + 15;
    const int nq = (63 + bases + 63) / 64;
    const int seq = SEQ_LEAD + 4 * nq + 4;
    return seq < 144 ? 144 : seq;                  // (= TileGeo::SEQ)
}

// Waves per workgroup: 4, unless the table is big (8 KB and more per workgroup) and sharing it among 8 waves puts more waves on a
// CU.  LDS is handed out in 1280-byte granules, 128 per CU; at most 8 workgroups per CU; and no more waves per SIMD than the
// kernel family's registers allow (`max_waves_simd`: what tps_kernels.h compiles the family for -- the raw-row kernels 5, the
// sums kernels of self-overlap tables 6 since round 5, the others are LDS-bound before their registers matter).
inline void plan_wpg(ScanArgs& a, int64_t budget_dw, int max_waves_simd = 8) {
    auto waves_per_cu = [&](int w) {
        a.wpg = w;
        const int64_t dw = wg_lds_dwords(a);
        if (dw > budget_dw) return 0;
        const int64_t gran = std::max<int64_t>(1, (dw * 4 + 1279) / 1280);
        const int64_t wgs = std::min<int64_t>(std::min<int64_t>(8, 128 / gran), (4 * max_waves_simd) / w);
        return (int)std::min<int64_t>(32, wgs * w);
    };
    int best = WPG, best_w = waves_per_cu(WPG);
    // only multiples of four: a workgroup's waves go round the four SIMDs, and five waves would put two on one of them
    // (measured at k = 6, 10 000 x 25 kb reads: 4 waves per workgroup 0.256 ms, 5: 0.306, 7: 0.251, 8: 0.222)
    if (lut_dw(a) * 4 >= 8192 || (a.pair16 && (a.pair_n + lut_dw(a)) * 4 >= 8192)) {
        // (ten waves per workgroup -- two workgroups of ten with a 16 KB table each fill the CU's LDS, 5 waves per SIMD -- was
        // measured in round 3: 248 us against 207 us with eight at k = 6: ten waves sit 3 / 3 / 2 / 2 on the four SIMDs)
        const int v = waves_per_cu(WPG_MAX);
        if (v > best_w) { best = WPG_MAX; best_w = v; }
    }
    a.wpg = best;
}


// Slides that have a specialised kernel instantiation (tps_scan_kernel_s<S>).
inline bool has_specialised_slide(int s) { return s == 5 || s == 6 || s == 7 || s == 8; }
// ... and the slides only the default kernels (sums only, no self-overlapping k-mer) are also instantiated for
inline bool has_default_only_slide(int s) { return s == 3 || s == 4 || (s >= 9 && s <= 12); }

// The planner reads NO environment (round 5; it used to consult thirteen A/B switches of four rounds of experiments on every plan).
// What tests and diagnostics still need to steer is explicit: the library sets these through tps_ctx_debug_option
// (include/topsicle_hip.h), the emulation through emu_set_knobs; all zero in normal use.
struct PlanKnobs {
    int spans_per_tile = 0;   // > 0: the generic kernel with this many spans per tile (small tiles in tests)
    int force_generic = 0;    // the generic kernel whatever the parameters (A/B against the fused tiles)
    int force_pair = 0;       // keep the pair table where it costs a resident workgroup (the emulation's slices are bigger than the device's)
    int so_order = 0;         // 2: the raw-row kernels of self-overlap tables try the chain-free tile first for EVERY tile (round 3's order)
    int wpg = 0;              // 4 or 8: waves per workgroup whatever plan_wpg would pick (A/B of the workgroup shapes)
};

// Geometry of one scan: fills variant, lut_n, lw/q/r, span_dw, blk_log2, spans_per_tile, nblk_cap,
// rec_rs, seq_dw, head_dw, tot_dw, blk_dw, lc_cap, jump_magic.  budget_dw = LDS dwords one workgroup (WPG waves +
// the shared table) may use.
inline std::string plan_geometry_core(ScanArgs& a, const tps_params& prm, int k, int P, int64_t max_nwin, int64_t budget_dw, const PlanKnobs& kn);
inline std::string plan_geometry(ScanArgs& a, const tps_params& prm, int k, int P, int64_t max_nwin, int64_t budget_dw,
                                 const PlanKnobs& kn = PlanKnobs()) {
    a.wpg = WPG;
    std::string err = plan_geometry_core(a, prm, k, P, max_nwin, budget_dw, kn);
    if (err.empty() && a.variant && !a.pat.hash_shift) {
        // the table gathers address LDS as (offset & mask) | base (tps_device.h: lut_at / lut16_at): the single table, which sits behind
        // the pair table, must start at a multiple of its own (power-of-two) byte size (nothing checked this when pair_n or
        // the table formats changed)
        const int64_t single_bytes = (int64_t)a.lut_n * (a.lut16 ? 2 : 4);
        if ((single_bytes & (single_bytes - 1)) != 0 || ((int64_t)a.pair_n * 4) % single_bytes != 0)
            err = "LDS plan: the single table (" + std::to_string(single_bytes) + " B) would not be aligned behind a pair table of " + std::to_string((int64_t)a.pair_n * 4) + " B";
    }
    if (err.empty()) {
        const bool raw = (prm.flags & TPS_F_STORE_RAW) != 0;
        plan_wpg(a, budget_dw, !a.variant ? 8 : raw ? 5 : a.pat.so_mask != 0 ? 6 : 8);
        if ((kn.wpg == WPG || kn.wpg == WPG_MAX) && a.variant) {
            const int32_t keep = a.wpg;
            a.wpg = kn.wpg;
            if (wg_lds_dwords(a) > budget_dw) a.wpg = keep;
        }
    }
    return err;
}
inline std::string plan_geometry_core(ScanArgs& a, const tps_params& prm, int k, int P, int64_t max_nwin, int64_t budget_dw, const PlanKnobs& kn) {
    const int spans_pref = kn.spans_per_tile, force_generic = kn.force_generic;
    a.lut_n = a.pat.hash_shift ? 2 * 256 : 1 << (2 * k);     // hashed table: 256 (key, mask) pairs
    a.lw = std::max(0, prm.window - k);            // k-mer start positions in a (W-1)-char window
    a.q = a.lw / prm.slide;
    a.r = a.lw % prm.slide;
    if (max_nwin > 500000) return "too many windows per read (" + std::to_string(max_nwin) + ")";
    {
        // Arithmetic limits of the kernels (all far beyond the reference's defaults: window 100, <= 3301 windows per read):
        //  * raw rows are bytes, and so are their accumulators (window_exact counts OCCURRENCES, overlapping ones included,
        //    before the greedy recount of self-overlapping patterns): a pattern may occur at most 255 times in a window;
        //  * the window sums of a read are added up in 32 bits (prefix sums, candidate sums, Binseg chunk sums);
        //  * the exact change-point tournament compares d^2 * den in 128 bits with d <= n * T, T <= n * max S_w and
        //    den <= n^2 / 4: n^6 * (max S_w)^2 / 4 < 2^128.
        int min_per = k;
        for (int i = 0; i < a.pat.n_periods; ++i) min_per = std::min(min_per, std::max(1, (int)a.pat.period[i]));
        const int64_t occ_max = a.lw > 0 ? (a.lw - 1) / min_per + 1 : 0;
        if ((prm.flags & TPS_F_STORE_RAW) && occ_max > 255)
            return "raw counts are bytes: a pattern can occur " + std::to_string(occ_max) + " times in a window of " +
                   std::to_string(prm.window) + " (max 255); use a smaller --windowSize with --rawcountpattern";
        const int64_t sw_max = (int64_t)P * ((a.lw + k - 1) / k + 1);            // S_w <= P * (non-overlapping maximum + 1)
        if (max_nwin * sw_max >= (1ll << 32)) return "window sums of one read exceed 32 bits (windows per read x window size too large)";
        if ((double)max_nwin * (double)max_nwin * (double)max_nwin * (double)sw_max >= 18446744073709551616.0 /* 2^64 */)
            return "change-point arithmetic would exceed 128 bits (windows per read x window size too large)";
    }
    const int jump = std::max(prm.jump, 1);
    a.lc_cap = (int)(max_nwin / jump + 2);
    a.jump_magic = jump == 1 ? 0u : (uint32_t)(((1ull << 32) + (uint64_t)jump - 1) / (uint64_t)jump);   // 0: divide by 1
    a.head_dw = 4 * ((prm.no_bp + 63 + 63) / 64) + 4;      // whole quads of a step-1 head whatever its start offset, + look-ahead words
    // fused kernels: compile-time slide, 16-bit masks (<= 15 patterns: bit 15 is the self-overlap flag of the fallback tile, FLAG16 -- a table
    // WITHOUT self-overlapping k-mers may use it as a sixteenth pattern, sums only: the 8-letter motifs at the reference's default
    // k = len - 2, e.g. TTTTAGGG at k = 6; round 5), a window spans at least one
    // 8-block chunk and its far end lies within the exchange halo (XLANES - NT lanes)
    int max_period = 0;                            // self-overlap periods of the table (0 = none)
    for (int i = 0; i < a.pat.n_periods; ++i) max_period = std::max(max_period, a.pat.period[i]);
    const bool sw16_ok = (int64_t)P * ((a.lw + k - 1) / k + 1) < 65536;        // the fused kernels keep S_w in 16 bits
    // the table's ONE self-overlap period, if the chain-corrected sums tiles (tile_lc_s<.., CD>) take it: distinct k-mers, k >= 4, 2 d >= k
    const int one_period = (a.pat.dup_mask == 0 && k >= 4 && a.pat.n_periods == 1 && a.pat.period[0] >= 2 && 2 * a.pat.period[0] >= k) ? a.pat.period[0] : 0;
    // sixteen patterns (sums only, distinct k-mers): bit 15 of the 16-bit masks is the self-overlap FLAG of the fallback tile (FLAG16), so
    // a sixteenth pattern fits where that tile cannot run -- a table without self-overlapping k-mers, or one the chain-corrected tiles take
    // on a batch without non-ACGT letters (TTTTAGGG at k = 6: TAGGGT has period 5)
    const bool p16_ok = P == 16 && a.pat.dup_mask == 0 && !(prm.flags & TPS_F_STORE_RAW) &&
                        (a.pat.so_mask == 0 || (!a.val_on && one_period > 0 && one_period <= prm.slide && a.lw <= 255));
    const bool fused = !force_generic && spans_pref <= 0 && k <= TPS_DIRECT_K && (has_specialised_slide(prm.slide) || (has_default_only_slide(prm.slide) && a.pat.so_mask == 0 && !(prm.flags & TPS_F_STORE_RAW))) &&
                       (P <= 15 || p16_ok) && a.q >= 8 && sw16_ok &&
                       a.q / 8 + 2 < (XLANES - NT) && max_period <= std::min(prm.slide, 6) &&
                       2 * a.head_dw <= fused_seq_dw(prm.slide);   // the two step-1 heads fit the tile buffer (TileGeo::SEQ)
    a.lc16 = 0; a.tile_cap = 0; a.tw = 0; a.tw_magic = 0; a.pair_n = 0; a.lc_global = 0; a.lc_stride = 0; a.pp_d = -1; a.so_fast = 0; a.seq_alias = 0; a.lut_fields = 0; a.tile_full = 0;
    a.lut16 = 0; a.xt_alias = 0; a.xt_own = 0; a.pair16 = 0;
    if (fused) {
        // per-pattern tiles (tile_pp_s): one-hot 2-bit fields per pattern need distinct k-mers, raw rows of at most 14 bytes
        // (they are staged through 16-byte LDS rows); a lane's 8 blocks hold at
        // most 14 non-overlapping occurrences of a pattern (nibbles), a block at most 2, a window at most 127 (bytes);
        // self-overlap only with ONE period d (then 2 d >= k: picks alternate along a chain)
        const bool pp_counts = a.pat.dup_mask == 0 && k >= 4 && P <= 14 && (8 * prm.slide + k - 1) / k <= 14 && (prm.slide + k - 1) / k <= 2 &&
                               a.lw / k + 2 <= 127;
        if (pp_counts && a.pat.n_periods == 0) a.pp_d = 0;
        if (pp_counts && a.pat.n_periods == 1 && a.pat.period[0] >= 2 && 2 * a.pat.period[0] >= k) a.pp_d = a.pat.period[0];
        // (sums only: the chain-corrected tiles need the one period, not the per-pattern tiles' nibble and row-length limits)
        if (!(prm.flags & TPS_F_STORE_RAW) && P > 14 && one_period > 0) a.pp_d = one_period;
        a.lut_fields = (a.pp_d >= 0 && (prm.flags & TPS_F_STORE_RAW)) ? 1 : 0;   // the per-pattern tiles will run: table of one-hot fields
        a.so_fast = kn.so_order == 2 ? 2 : 1;
        // the chain-corrected sums tiles (tile_lc_s<.., CD>) carry a window's matches and pairs as two 8-bit fields of one 16-bit
        // count (at most one of either per start position): windows of more than 255 start positions take the flag-and-recount tile
        if (!(prm.flags & TPS_F_STORE_RAW) && a.pp_d > 0 && a.lw > 255) a.so_fast = 0;   // sums only, pp_d > 0: chain-free tiles complete as plain tiles (tile_fused_s<.., CD>)
        // the sums-only kernels of self-overlap tables (_s*so, _s*sol): 16-bit table, XT aliased onto the staged bases unless the
        // fallback tile can run (a batch with non-ACGT letters, a table the chain corrections do not take, windows of more than 255 start positions)
        if (a.pat.so_mask != 0 && !(prm.flags & TPS_F_STORE_RAW)) {
            a.lut16 = 1;
            a.xt_alias = 2;                        // (round 5: lane totals in the pad words, like the default kernels -- no XF / XT in the slice)
            a.xt_own = (a.val_on || !a.so_fast || a.pp_d <= 0) ? 1 : 0;
        }
        // the raw-row kernels (_s*r, _s*sor): no XF / XT in the exchange region (tile_pp_s keeps its lane totals in END's pad words);
        // the fallback tile -- a batch with non-ACGT letters, a table the per-pattern tiles do not take -- gets them back
        if (prm.flags & TPS_F_STORE_RAW) {
            // ... and a self-overlap table of 4^6 k-mers or more goes into LDS as 16-bit field indices (LUT_F16, kernels _s*sorh)
            if (a.lut_fields && a.pat.so_mask != 0 && a.pp_d > 0 && a.lut_n >= 4096) a.lut16 = 1;
            a.xt_alias = 2;
            a.xt_own = (a.val_on || (a.pat.so_mask != 0 ? a.pp_d <= 0 : a.pp_d != 0)) ? 1 : 0;
        }
        // ... and so do the default kernels (no self-overlap, sums only: every tile is a tile_lc_s<.., CD = 0>, lane totals in the pad words)
        if (a.pat.so_mask == 0 && !(prm.flags & TPS_F_STORE_RAW)) { a.xt_alias = 2; a.xt_own = 0; }
        a.variant = prm.slide;
        a.blk_log2 = 3;                            // 8 blocks per lane for every slide
        a.span_dw = 0;                             // lanes start at arbitrary bit offsets (per-lane shift)
        a.spans_per_tile = NT;
        a.nblk_cap = NT * 8;
        a.rec_rs = 0;
        a.tot_dw = 4;                              // (keeps seq2 16-byte aligned behind it: carve_fused)
        // all 64 lanes hold window blocks (FULL tiles): with the packed batch a tile is staged as whole 64-base quads, one
        // per lane, whatever the slide -- the halo-lane variant of the ASCII days has no cheaper staging any more
        a.tile_full = 1;
        a.seq_dw = fused_seq_dw(prm.slide);        // = TileGeo<S, true>::SEQ: compile-time size in the kernel (carve_fused)
        // the kernels that never return to the bases after a tile's first phase (sums only, no self-overlapping k-mer) keep
        // them in the tail of row[] (carve_fused): with no invalid letters in the batch that is the sixth workgroup per CU
        a.seq_alias = 0;
#ifndef TPS_EMU
        if (a.pat.so_mask == 0 && !(prm.flags & TPS_F_STORE_RAW)) a.seq_alias = 1;
#endif
        // windows per tile: every lane's 8 blocks hold window starts, a window spans q + 1 blocks; EVEN, because the fused
        // kernels store S_w as 16-bit values in whole dwords (g_store_sw8): no tile but a read's last ends inside a dword
        a.tw = ((int)NT * 8 - a.q - 1) & ~1;
        a.tw_magic = (uint32_t)(((1ull << 32) + (uint64_t)a.tw - 1) / (uint64_t)a.tw);
        a.tile_cap = (int)((max_nwin + a.tw - 1) / a.tw) + 1;
        // The candidates' left sums live off-chip (L2-resident scratch, written once and read once by the same wave) as
        // absolute 32-bit sums: 4 bytes per candidate instead of the 2 of the tile-relative 16-bit form, but the
        // change-point step then needs no per-candidate tile lookup (mul, mulhi, LDS read, add) and no Tc array in LDS.
        // (The tile-relative 16-bit and the in-LDS layouts of rounds 1 - 2 are gone with their switches.)
        // Round 4, the 40 MB this round trip adds to a config-2 launch's 128 MB at the memory side -- three ways around it, measured
        // (A/B on one box each, us per launch) and dropped:  (i) the block indexed by the HARDWARE WAVE SLOT (s_getreg HW_ID / XCC_ID:
        // 12 MB rewritten in place launch after launch instead of 20 MB per batch): 55.3 -> 54.4, config 4's sample 67.7 -> 66.7 -- but
        // WRITE_SIZE does not move (71.3 -> 71.2 MB; FETCH_SIZE x 2: 57.1 -> 52.4 MB): ~7 MB of streaming traffic pass through an
        // XCD's 4 MB L2 between two reads of a slot, the lines are gone either way; and a wave restored into another slot after a
        // queue preemption would share its block with a newcomer unnoticed -- 1.5 % are not worth a checksum-and-fallback.
        // (ii) ... with the bases as non-temporal loads, so that the scratch lines outlive them in the L2: 54.8 -> 55.5 (54.0 with (i)
        // alone on that box; WRITE_SIZE 67.4 MB).  (iii) the sums in 16 VGPRs per lane -- register (tile, pass), a wave-uniform index:
        // a scalar branch to one v_mov per pass; 57 -> 73 VGPRs, still six waves per SIMD; emulation and all GPU tests green --
        // 58.2 against 54.8 (and 56.6 with the registers compiled in but switched off): the longer live ranges and the 16-way uniform
        // dispatch in the tile loop and the change-point step cost more than the round trip, which rides the Infinity Cache.
        a.lc16 = 0;
        a.lc_global = 1;
        a.lc_stride = 2 * ((a.lc_cap + 1) & ~1);        // in 16-bit units
        // pair table (two positions per lookup) while it is small: k <= 4 -> at most 4 KB per workgroup
        a.pair_n = (a.pat.so_mask == 0 && k <= 4 && !(prm.flags & TPS_F_STORE_RAW)) ? (1 << (2 * (k + 1))) : 0;
        // the raw-row kernel of slide 6 (round 5): a pair table of one-hot FIELDS for the per-pattern tiles (tile_pp_s<.., PAIRF>) when
        // the window's partial block ends between two pairs (k = 4 at the default window: r = 0)
        if (a.pat.so_mask == 0 && k <= 4 && (prm.flags & TPS_F_STORE_RAW) && a.lut_fields && a.pp_d == 0 && prm.slide == 6 && a.r == 0) a.pair_n = 1 << (2 * (k + 1));
        // ... and at k = 5 as 16-bit pattern masks (round 4; the single table likewise: kernels _s*q): 4^6 entries = 8 KB per workgroup,
        // which plan_wpg then shares among 8 waves -- the plant-type 7-mer motifs (CCCTAAA at the reference's default k = 5) get two
        // positions per lookup like the 6-mer motifs at k = 4
        a.pair16 = 0;
        if (a.pat.so_mask == 0 && a.pat.dup_mask == 0 && k == 5 && P <= 15 && has_specialised_slide(prm.slide) && !(prm.flags & TPS_F_STORE_RAW)) {
            a.pair16 = 1;
            a.lut16 = 1;
            a.pair_n = (1 << (2 * (k + 1))) / 2;       // dwords
        }
        a.blk_dw = (int32_t)blk_region_dw(a);
        if (a.pair_n && !kn.force_pair) {
            // ... unless it costs a resident workgroup where one is scarce: LDS is handed out in 1280-byte granules,
            // 128 per CU.  Measured at config 2 (ms per batch): 4 workgroups + pair 0.096 vs 5 + single lookups 0.092;
            // 5 + pair 0.089 vs 6 + single lookups 0.091 -- so the pair table stays if 5 workgroups still fit.
            // (in waves per CU since round 4: the 8 KB table of the k = 5 kernels is planned for 8-wave workgroups, plan_wpg below)
            auto waves_per_cu = [&](int w) {
                const int32_t w0 = a.wpg;
                a.wpg = w;
                const int64_t dw = wg_lds_dwords(a);
                a.wpg = w0;
                if (dw > budget_dw) return 0;
                return (int)(std::min<int64_t>(8, 128 / std::max<int64_t>(1, (dw * 4 + 1279) / 1280)) * w);
            };
            const int with_pair = a.pair16 ? std::max(waves_per_cu(WPG), waves_per_cu(WPG_MAX)) : waves_per_cu(WPG);
            const int32_t keep = a.pair_n;
            const int32_t keep16 = a.pair16;
            a.pair_n = 0;
            if (keep16) { a.lut16 = 0; a.pair16 = 0; }        // (without: the 32-bit single table)
            if (with_pair >= 5 * WPG || waves_per_cu(WPG) <= with_pair) { a.pair_n = keep; if (keep16) { a.pair16 = 1; a.lut16 = 1; } }
        }
        if (wg_lds_dwords(a) <= budget_dw) return "";
        // does not fit (very long maxlengthtelo: the candidate sums of 4 reads outgrow LDS): the generic kernel,
        // whose tile size adapts, takes over
        a.lc16 = 0; a.tile_cap = 0; a.tw = 0; a.tw_magic = 0; a.pair_n = 0; a.lc_global = 0; a.lc_stride = 0; a.pp_d = -1; a.so_fast = 0; a.seq_alias = 0; a.lut_fields = 0;
        a.lut16 = 0; a.xt_alias = 0; a.xt_own = 0; a.pair16 = 0;
    }
    a.variant = 0;
    // the generic kernel's 32-bit candidate sums go off-chip too (stride counted in 16-bit units)
    a.lc_global = 1;
    a.lc_stride = 2 * ((a.lc_cap + 1) & ~1);
    const int g = gcd_i(prm.slide, 16);
    a.span_dw = prm.slide / g;
    const int bps = 16 / g;
    a.blk_log2 = 0;
    while ((1 << a.blk_log2) < bps) ++a.blk_log2;
    const int max_spans = 2 * NT;
    const int min_spans = (a.q + 2 + bps - 1) / bps;        // a tile must hold >= 1 window
    const int64_t need_blk = max_nwin + a.q + 1;            // no more spans than the longest read uses
    const int need_spans = (int)std::min<int64_t>((need_blk + bps - 1) / bps, max_spans);
    int spans = spans_pref > 0 ? std::min(spans_pref, max_spans) : std::min(std::max(need_spans, 1), (int)NT);
    spans = std::max(min_spans, std::min(spans, std::max(need_spans, 1)));
    if (spans > max_spans) return "window/slide combination needs " + std::to_string(spans) + " spans per tile (max " + std::to_string(max_spans) + ")";
    auto fill = [&](int sp) {
        a.spans_per_tile = sp;
        a.nblk_cap = sp * bps;
        a.rec_rs = 0;
        a.tot_dw = std::max(((sp + 2) / 2) * 2, (int)NT);
        a.seq_dw = (std::max(sp * a.span_dw + 8, 2 * a.head_dw) + 3) & ~3;     // whole quads; a tile starts up to 63 positions into its first one
        a.blk_dw = (int32_t)blk_region_dw(a);
        return wg_lds_dwords(a);
    };
    if (spans_pref > 0) {
        if (fill(spans) > budget_dw)
            return "LDS plan does not fit with " + std::to_string(spans) + " spans per tile";
        return "";
    }
    while (fill(spans) > budget_dw) {
        if (spans <= min_spans)
            return "LDS plan does not fit: window=" + std::to_string(prm.window) + " slide=" + std::to_string(prm.slide) +
                   " k=" + std::to_string(k) + " windows/read=" + std::to_string(max_nwin);
        spans = std::max(min_spans, spans / 2);
    }
    return "";
}

// A slide that no fused kernel is compiled for (13 and up for the default tables, 9 and up for raw rows and self-overlap tables) takes
// the generic kernel, 12 - 24 x slower per launch (measured, round 5: 10 000 x 15 kb, slide 10: raw rows 1 870 us against 155 us at
// slide 5; k = 6 sums 2 839 against 122).  But the windows of slide m s0 ARE every m-th window of slide s0 (same start grid, same
// length): such a scan runs the fused kernel of the base slide s0 and keeps every m-th window (tps_stride_kernel: S_w and raw rows
// compacted into the layout of the requested slide, the change point searched on the compacted series).  Returns the base slide
// (the largest one whose plan is fused: fewest windows to drop) or 0; `max_nwin_of(s)` = the longest read's windows at slide s.
template <typename F>
inline int stride_base(const ScanArgs& planned, const tps_params& prm, int k, int P, F max_nwin_of, int64_t budget_dw, const PlanKnobs& kn) {
    if (planned.variant != 0 || kn.force_generic || kn.spans_per_tile > 0 || !(prm.flags & TPS_F_WINDOWS)) return 0;
    for (int s0 = 12; s0 >= 3; --s0) {
        if (prm.slide <= s0 || prm.slide % s0 != 0 || prm.slide / s0 > 6) continue;
        if (!has_specialised_slide(s0) && !has_default_only_slide(s0)) continue;
        ScanArgs t{};
        t.val_on = planned.val_on;
        t.pat = planned.pat;
        tps_params pb = prm;
        pb.slide = s0;
        pb.flags = (prm.flags | TPS_F_STORE_SUMS) & ~(uint32_t)TPS_F_BINSEG;
        if (plan_geometry(t, pb, k, P, max_nwin_of(s0), budget_dw, kn).empty() && t.variant != 0) return s0;
    }
    return 0;
}

}  // namespace tps
