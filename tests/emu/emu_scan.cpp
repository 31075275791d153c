// TEST INFRASTRUCTURE ONLY -- sequential host emulation of the HIP kernel source.
//
// Compiles topsicle_amd/csrc/tps_device.h with -DTPS_EMU: the very same per-read program the
// GPU runs, with TPS_PHASE looping over the 256 thread ids and the few gfx950 intrinsics
// replaced by portable C.  It lets the kernel LOGIC be checked against the oracle in a
// container without a GPU (tests/test_emulation.py).  It is never linked into the product
// library and the product never loads it.
#define TPS_EMU 1
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../topsicle_amd/csrc/tps_device.h"
#include "../../topsicle_amd/csrc/tps_pack.h"
#include "../../topsicle_amd/csrc/tps_plan.h"

static std::string g_err;
static int g_variant_calls[9] = {0};
extern "C" int emu_counter(int i) { return (i >= 0 && i < 8) ? tps::emu_counter(i) : -1; }
extern "C" int emu_variant_calls(int v) { return (v >= 0 && v < 9) ? g_variant_calls[v] : -1; }

extern "C" const char* emu_last_error() { return g_err.c_str(); }

// planner knobs + "val_off" of the next emu_scan / emu_plan* calls (tests/emu_driver.py: KNOBS); test infrastructure reads no environment either
static tps::PlanKnobs g_knobs;
static int g_val_off = 0;
static int g_raw_m = 0;        // ScanArgs::raw_m of the next emu_scan (2: the per-pattern tiles store every 2nd window's row, packed per read)
extern "C" void emu_set_raw_m(int m) { g_raw_m = m; }
extern "C" void emu_set_knobs(int force_pair, int so_order, int val_off) { g_knobs = tps::PlanKnobs{}; g_knobs.force_pair = force_pair; g_knobs.so_order = so_order; g_val_off = val_off; }
static tps::PlanKnobs knobs_with(int spans_pref, int force_generic) { tps::PlanKnobs k = g_knobs; k.spans_per_tile = spans_pref; k.force_generic = force_generic; return k; }

// tps::plan_dispatch_order as the library calls it; returns the number of entries written (0 = file order)
extern "C" int64_t emu_dispatch_order(const int64_t* n_win, const uint8_t* passes, int64_t n, int32_t* out) {
    std::vector<int32_t> order;
    tps::plan_dispatch_order(n_win, passes, n, order);
    for (size_t i = 0; i < order.size(); ++i) out[i] = order[i];
    return (int64_t)order.size();
}

// tps::stride_base as the library calls it: the base slide a scan at prm->slide runs at (0 = the planned kernel itself)
extern "C" int emu_stride_base(const char* pats, int P, int k, const tps_params* prm, int64_t max_len) {
    std::vector<uint32_t> lut;
    tps::ScanArgs a{};
    a.val_on = 0;
    if (!tps::build_patterns(pats, P, k, lut, a.pat).empty()) return -1;
    const int64_t budget = 160 * 1024 / 4;
    if (!tps::plan_geometry(a, *prm, k, P, tps::window_count(max_len, prm->window, prm->slide, prm->trimfirst, prm->maxlen), budget, g_knobs).empty()) return -1;
    return tps::stride_base(a, *prm, k, P, [&](int s0) { return tps::window_count(max_len, prm->window, s0, prm->trimfirst, prm->maxlen); }, budget, g_knobs);
}

extern "C" int64_t emu_window_count(int64_t L, int W, int s, int t, int M) { return tps::window_count(L, W, s, t, M); }

// One scan over a batch, like tps_batch_upload + tps_batch_scan + downloads.
// lds_budget_bytes / spans_pref let tests force small tiles.  base_shift (0..15) misaligns the
// concatenated bases relative to the 16-byte load grid.
extern "C" int emu_scan(const char* pats, int P, int k, const uint8_t* bases, const int64_t* offsets, int64_t n,
                        const uint8_t* tails, const tps_params* prm, int spans_pref, int lds_budget_bytes,
                        int base_shift, int force_generic, tps_read_result* results, int32_t* c_start, int32_t* c_end,
                        int64_t* win_off_out, int32_t* sums, uint8_t* raw) {
    std::vector<uint32_t> lut;
    tps::ScanArgs a{};
    a.val_on = g_val_off ? 0 : 1;  // (the emulation keeps the invalid-mask staging area; knob val_off: the layout of a clean batch, bases without invalid letters only)
    std::string err = tps::build_patterns(pats, P, k, lut, a.pat);
    if (!err.empty()) { g_err = err; return TPS_E_PATTERN; }
    std::vector<int64_t> win_off((size_t)n + 1);
    int64_t acc = 0, mx = 0;
    for (int64_t i = 0; i < n; ++i) {
        win_off[(size_t)i] = acc;
        int64_t nw = tps::window_count(offsets[i + 1] - offsets[i], prm->window, prm->slide, prm->trimfirst, prm->maxlen);
        mx = nw > mx ? nw : mx;
        acc += nw;
    }
    win_off[(size_t)n] = acc;
    if (win_off_out) memcpy(win_off_out, win_off.data(), (size_t)(n + 1) * 8);
    err = tps::plan_geometry(a, *prm, k, P, mx, lds_budget_bytes / 4, knobs_with(spans_pref, force_generic));
    if (!err.empty()) { g_err = err; return TPS_E_CAPACITY; }

    // the packed batch, exactly as the library keeps it in HBM (tps_pack.h).  Words the layout does not own are filled
    // with garbage: the kernel must never let them count.  base_shift moves the batch inside its buffer by whole quads
    // (the device only ever sees 16-byte aligned quads).
    std::vector<tps_read_desc> desc((size_t)(n > 0 ? n : 1));
    const int64_t n_words = tps::pack_layout(offsets, n, desc.data());
    const int64_t lead = 4 * (int64_t)(base_shift & 3);
    std::vector<uint32_t> seq2buf((size_t)(n_words + lead + 8), 0xDEADBEEFu);
    std::vector<uint16_t> invbuf((size_t)(n_words + lead + 8), (uint16_t)0xFFFFu);
    for (int64_t i = 0; i < n; ++i) desc[(size_t)i].word_off += lead;
    tps::pack_range(bases, offsets, 0, n, desc.data(), seq2buf.data(), invbuf.data());
    a.seq2 = seq2buf.data();
    a.inv = invbuf.data();
    a.desc = desc.data();
    a.tails_in = ((prm->flags & TPS_F_TAILS_IN) && !(prm->flags & TPS_F_STEP1)) ? tails : nullptr;
    a.lut = lut.data();
    a.results = results;
    a.c_start = (prm->flags & TPS_F_STEP1) ? c_start : nullptr;
    a.c_end = (prm->flags & TPS_F_STEP1) ? c_end : nullptr;
    a.win_off = win_off.data();
    a.sums = sums;                                 // always present, like the library's device buffer
    // fused kernels: 16-bit sums in the padded device layout (tps_plan.h: sums16_slots), widened below like the library's download
    std::vector<int64_t> win_off16((size_t)n + 1);
    {
        int64_t acc16 = 0;
        for (int64_t i = 0; i < n; ++i) { win_off16[(size_t)i] = acc16; acc16 += tps::sums16_slots(win_off[(size_t)i + 1] - win_off[(size_t)i]); }
        win_off16[(size_t)n] = acc16;
    }
    std::vector<uint16_t> sums16((size_t)win_off16[(size_t)n] + 8, (uint16_t)0xBEEF);
    a.sums16 = sums16.data();
    a.win_off16 = win_off16.data();
    a.raw = (prm->flags & TPS_F_STORE_RAW) ? raw : nullptr;
    // knob raw_m = 2: the rows of the even windows only, in the layout of twice the slide (read i's rows start at sum of ceil(n_win / 2))
    std::vector<int64_t> rwo((size_t)n + 1, 0);
    if (g_raw_m == 2 && a.raw) {
        for (int64_t i = 0; i < n; ++i) rwo[(size_t)i + 1] = rwo[(size_t)i] + (win_off[(size_t)i + 1] - win_off[(size_t)i] + 1) / 2;
        a.raw_m = 2;
        a.raw_win_off = rwo.data();
    }
    a.n_reads = n;
    a.prm = *prm;
    std::vector<uint16_t> lc_scratch(a.lc_global ? (size_t)n * (size_t)a.lc_stride + 8 : 8, (uint16_t)0xBEEF);
    a.lc_scratch = a.lc_global ? lc_scratch.data() : nullptr;
    // the workgroup-shared tables (read-only for the waves): [pair table][single table]
    std::vector<uint32_t> lutbuf((size_t)a.pair_n + lut.size());
    uint32_t* lut1 = lutbuf.data() + a.pair_n;
    if (a.lut16 && a.lut_fields) {                 // raw rows of a big self-overlap table: 16-bit field indices (LUT_F16)
        for (size_t i = 0; i < lut.size(); ++i) ((uint16_t*)lut1)[i] = lut[i] ? (uint16_t)(1u << tps::pp_field(__builtin_ctz(lut[i]))) : (uint16_t)0;
    } else if (a.lut16) {                          // sums-only kernels of self-overlap tables: 16-bit pattern masks
        for (size_t i = 0; i < lut.size(); ++i) ((uint16_t*)lut1)[i] = (uint16_t)lut[i];
    } else
    for (size_t i = 0; i < lut.size(); ++i)
        lut1[i] = !a.variant ? lut[i] : a.lut_fields ? tps::mask_to_fields(lut[i])                 // raw-count kernels on the per-pattern tiles: one-hot fields
                                                     : ((lut[i] << 16) | (uint32_t)__builtin_popcount(lut[i]));   // fused kernels: mask << 16 | count
    if (a.pair16) {                                // k = 5 pair-table kernels (_s*q): 4^(k+1) 16-bit masks, the two positions' masks ORed
        for (int c = 0; c < 2 * a.pair_n; ++c)
            ((uint16_t*)lutbuf.data())[c] = (uint16_t)(lut[(size_t)(c & (int)a.pat.kmask)] | lut[(size_t)((c >> 2) & (int)a.pat.kmask)]);
    } else if (a.lut_fields) {                     // raw-row kernels, k <= 4: the two positions' one-hot fields added up (tile_pp_s<.., PAIRF>)
        for (int c = 0; c < a.pair_n; ++c) lutbuf[(size_t)c] = lut1[c & a.pat.kmask] + lut1[(c >> 2) & a.pat.kmask];
    } else
    for (int c = 0; c < a.pair_n; ++c) {
        const uint32_t e1 = lut1[c & a.pat.kmask], e2 = lut1[(c >> 2) & a.pat.kmask];
        lutbuf[(size_t)c] = ((e1 | e2) & 0xFFFF0000u) | ((e1 + e2) & 0xFFFFu);
    }
    std::vector<uint32_t> ldsbuf((size_t)tps::lds_dwords(a) + 16);
    uint32_t* lds_al = (uint32_t*)(((uintptr_t)ldsbuf.data() + 15) & ~(uintptr_t)15);
    struct { uint32_t* p; size_t n; uint32_t* data() { return p; } uint32_t* begin() { return p; } uint32_t* end() { return p + n; } } lds{lds_al, (size_t)tps::lds_dwords(a)};
    g_variant_calls[a.variant] += (int)n;
    // the reads in the library's dispatch order (tps::plan_dispatch_order; results do not depend on it)
    std::vector<int32_t> order;
    {
        std::vector<int64_t> nwv((size_t)n);
        std::vector<uint8_t> longer((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            nwv[(size_t)i] = (prm->flags & TPS_F_WINDOWS) ? win_off[(size_t)i + 1] - win_off[(size_t)i] : 0;
            longer[(size_t)i] = !(prm->flags & TPS_F_STEP1) || offsets[i + 1] - offsets[i] > prm->min_len;
        }
        tps::plan_dispatch_order(nwv.data(), longer.data(), n, order);
    }
    for (int64_t slot = 0; slot < n; ++slot) {
        const int64_t r = order.empty() ? slot : order[(size_t)slot];
        for (auto& w : lds) w = 0xDEADBEEFu;          // LDS content is undefined at workgroup start
        const bool so = a.pat.so_mask != 0;
        const bool want_raw = a.raw != nullptr;
        // the library's kernel families: plain, pair table, raw rows, self-overlap sums, self-overlap raw
#define EMU_CASE(S)                                                                                              \
        case S:                                                                                                  \
            if (so && want_raw && a.lut16) tps::scan_read<S, true, false, true, tps::tile_full_default(S), 3>(a, r, lds.data(), lut1); \
            else if (so && want_raw) tps::scan_read<S, true, false, true>(a, r, lds.data(), lut1);               \
            else if (so) tps::scan_read<S, true, false, false>(a, r, lds.data(), lut1);                          \
            else if (want_raw && a.pair_n) tps::scan_read<S, false, true, true>(a, r, lds.data(), lut1);         \
            else if (want_raw) tps::scan_read<S, false, false, true>(a, r, lds.data(), lut1);                    \
            else if (a.pair_n && a.pair16) tps::scan_read<S, false, true, false, tps::tile_full_default(S), 4>(a, r, lds.data(), lut1); \
            else if (a.pair_n) tps::scan_read<S, false, true, false>(a, r, lds.data(), lut1);                    \
            else tps::scan_read<S, false, false, false>(a, r, lds.data(), lut1);                                 \
            break;
        switch (a.variant) {
            EMU_CASE(5) EMU_CASE(6) EMU_CASE(7) EMU_CASE(8)
#define EMU_CASE_X(S) case S: if (a.pair_n && !a.pair16) tps::scan_read<S, false, true, false>(a, r, lds.data(), lut1); else tps::scan_read<S, false, false, false>(a, r, lds.data(), lut1); break;
            EMU_CASE_X(3) EMU_CASE_X(4) EMU_CASE_X(9) EMU_CASE_X(10) EMU_CASE_X(11) EMU_CASE_X(12)
#undef EMU_CASE_X
            default: tps::scan_read<0, false>(a, r, lds.data(), lut1); break;
        }
#undef EMU_CASE
    }
    if (a.variant && sums)
        for (int64_t i = 0; i < n; ++i)
            for (int64_t w = 0; w < win_off[(size_t)i + 1] - win_off[(size_t)i]; ++w)
                sums[win_off[(size_t)i] + w] = (int32_t)sums16[(size_t)(win_off16[(size_t)i] + w)];
    return TPS_OK;
}

// tps_batch_kmer_followers through the emulation: picks[n][2][n_fwd][pw], hist[2][n_fwd][4^follow + 1] (may be NULL)
extern "C" int emu_followers(const char* pats, int P, int k, const uint8_t* bases, const int64_t* offsets, int64_t n, int n_fwd, int follow,
                             int lo, int hi, int min_len, uint32_t* picks, unsigned long long* hist) {
    std::vector<uint32_t> lut;
    tps::FollowArgs a{};
    std::string err = tps::build_patterns(pats, P, k, lut, a.pat);
    if (!err.empty()) { g_err = err; return TPS_E_PATTERN; }
    if (n_fwd < 1 || n_fwd > 15 || 2 * n_fwd > P || follow < 0 || follow > 8 || lo < 0 || hi <= lo || hi - lo > tps::FOLLOW_MAX_SPAN) { g_err = "bad arguments"; return TPS_E_ARG; }
    std::vector<tps_read_desc> desc((size_t)(n > 0 ? n : 1));
    const int64_t n_words = tps::pack_layout(offsets, n, desc.data());
    std::vector<uint32_t> seq2buf((size_t)(n_words + 8), 0xDEADBEEFu);
    std::vector<uint16_t> invbuf((size_t)(n_words + 8), (uint16_t)0xFFFFu);
    tps::pack_range(bases, offsets, 0, n, desc.data(), seq2buf.data(), invbuf.data());
    a.seq2 = seq2buf.data(); a.inv = invbuf.data(); a.desc = desc.data(); a.lut = lut.data();
    a.picks = picks; a.hist = hist; a.n_reads = n;
    a.n_fwd = n_fwd; a.follow = follow; a.lo = lo; a.hi = hi; a.min_len = min_len; a.pw = (hi - lo + 31) / 32; a.nbins = (1 << (2 * follow)) + 1;
    std::vector<uint32_t> ldsbuf((size_t)tps::FOLLOW_LDS_DW + 16);
    uint32_t* lds = (uint32_t*)(((uintptr_t)ldsbuf.data() + 15) & ~(uintptr_t)15);
    for (int64_t r = 0; r < n; ++r) {
        for (int i = 0; i < tps::FOLLOW_LDS_DW; ++i) lds[i] = 0xDEADBEEFu;
        tps::followers_read(a, r, lds);
    }
    return TPS_OK;
}

extern "C" int emu_binseg(const int32_t* sums, const int64_t* win_off, int64_t n, int n_patterns, int jump,
                          int min_size, int32_t* bkp, double* gain, uint8_t* tie) {
    tps::BinsegArgs a{sums, win_off, bkp, gain, n, n_patterns, jump, min_size, tie};
    std::vector<uint32_t> miscbuf(tps::BINSEG_SMEM_DW + 8);
    struct { uint32_t* p; size_t n; uint32_t* data() { return p; } uint32_t* begin() { return p; } uint32_t* end() { return p + n; } } misc{(uint32_t*)(((uintptr_t)miscbuf.data() + 15) & ~(uintptr_t)15), (size_t)tps::BINSEG_SMEM_DW};
    for (int64_t r = 0; r < n; ++r) {
        for (auto& w : misc) w = 0xDEADBEEFu;
        tps::binseg_read(a, r, misc.data());
    }
    return TPS_OK;
}

extern "C" int emu_plan(int k, int P, const tps_params* prm, int64_t max_nwin, int spans_pref, int lds_budget_bytes, int force_generic, int32_t* out10) {
    tps::ScanArgs a{};
    a.val_on = 1;                                  // (the emulation always keeps the invalid-mask staging area)
    std::string err = tps::plan_geometry(a, *prm, k, P, max_nwin, lds_budget_bytes / 4, knobs_with(spans_pref, force_generic));
    if (!err.empty()) { g_err = err; return TPS_E_CAPACITY; }
    out10[0] = a.spans_per_tile; out10[1] = a.span_dw; out10[2] = a.blk_log2; out10[3] = a.q; out10[4] = a.r;
    out10[5] = a.lw; out10[6] = a.seq_dw; out10[7] = (int32_t)(tps::wg_lds_dwords(a) * 4); out10[8] = a.variant; out10[9] = a.rec_rs;
    return TPS_OK;
}

// plan for a real pattern table: out = {variant, pp_d, workgroup LDS bytes, pair_n, tile_full, tw}
extern "C" int emu_plan_table(const char* pats, int P, int k, const tps_params* prm, int64_t max_nwin, int32_t* out4) {
    tps::ScanArgs a{};
    a.val_on = g_val_off ? 0 : 1;                  // (the emulation keeps the invalid-mask staging area unless the knob says "clean batch")
    std::vector<uint32_t> lut;
    std::string err = tps::build_patterns(pats, P, k, lut, a.pat);
    if (err.empty()) err = tps::plan_geometry(a, *prm, k, P, max_nwin, 160 * 1024 / 4, knobs_with(0, 0));
    if (!err.empty()) { g_err = err; return TPS_E_CAPACITY; }
    out4[0] = a.variant; out4[1] = a.pp_d; out4[2] = (int32_t)(tps::wg_lds_dwords(a) * 4); out4[3] = a.pair_n; out4[4] = a.tile_full; out4[5] = a.tw;
    return TPS_OK;
}
