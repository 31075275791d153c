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
#include "../../topsicle_amd/csrc/tps_plan.h"

static std::string g_err;

extern "C" const char* emu_last_error() { return g_err.c_str(); }

extern "C" int64_t emu_window_count(int64_t L, int W, int s, int t, int M) { return tps::window_count(L, W, s, t, M); }

// One scan over a batch, like tps_batch_upload + tps_batch_scan + downloads.
// lds_budget_bytes / spans_pref let tests force small tiles.  base_shift (0..15) misaligns the
// concatenated bases relative to the 16-byte load grid.
extern "C" int emu_scan(const char* pats, int P, int k, const uint8_t* bases, const int64_t* offsets, int64_t n,
                        const uint8_t* tails, const tps_params* prm, int spans_pref, int lds_budget_bytes,
                        int base_shift, tps_read_result* results, int32_t* c_start, int32_t* c_end,
                        int64_t* win_off_out, int32_t* sums, uint8_t* raw) {
    std::vector<uint32_t> lut;
    tps::ScanArgs a{};
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
    err = tps::plan_geometry(a, *prm, k, mx, lds_budget_bytes / 4, spans_pref);
    if (!err.empty()) { g_err = err; return TPS_E_CAPACITY; }

    const int64_t total = offsets[n];
    const int64_t PAD = 64;
    std::vector<uint8_t> buf((size_t)(total + 2 * PAD + 64), (uint8_t)'A');
    uint8_t* al = (uint8_t*)(((uintptr_t)buf.data() + 15) & ~(uintptr_t)15);
    uint8_t* data = al + 16 + (base_shift & 15);
    // bytes around the data are deliberately non-ACGT garbage: the kernel must never let them count
    memset(al, '#', (size_t)(16 + (base_shift & 15)));
    if (total) memcpy(data, bases, (size_t)total);
    memset(data + total, '#', 40);

    a.bases = data;
    a.offsets = offsets;
    a.tails_in = ((prm->flags & TPS_F_TAILS_IN) && !(prm->flags & TPS_F_STEP1)) ? tails : nullptr;
    a.lut = lut.data();
    a.results = results;
    a.c_start = (prm->flags & TPS_F_STEP1) ? c_start : nullptr;
    a.c_end = (prm->flags & TPS_F_STEP1) ? c_end : nullptr;
    a.win_off = win_off.data();
    a.sums = (prm->flags & TPS_F_STORE_SUMS) ? sums : nullptr;
    a.raw = (prm->flags & TPS_F_STORE_RAW) ? raw : nullptr;
    a.n_reads = n;
    a.prm = *prm;
    std::vector<uint32_t> lds((size_t)tps::lds_dwords(a) + 8);
    for (int64_t r = 0; r < n; ++r) {
        for (auto& w : lds) w = 0xDEADBEEFu;          // LDS content is undefined at workgroup start
        tps::scan_read(a, r, lds.data());
    }
    return TPS_OK;
}

extern "C" int emu_binseg(const int32_t* sums, const int64_t* win_off, int64_t n, int n_patterns, int jump,
                          int min_size, int32_t* bkp, double* gain) {
    tps::BinsegArgs a{sums, win_off, bkp, gain, n, n_patterns, jump, min_size};
    std::vector<uint32_t> misc(tps::MISC_DW);
    for (int64_t r = 0; r < n; ++r) {
        for (auto& w : misc) w = 0xDEADBEEFu;
        tps::binseg_read(a, r, misc.data());
    }
    return TPS_OK;
}

extern "C" int emu_plan(int k, const tps_params* prm, int64_t max_nwin, int spans_pref, int lds_budget_bytes, int32_t* out8) {
    tps::ScanArgs a{};
    std::string err = tps::plan_geometry(a, *prm, k, max_nwin, lds_budget_bytes / 4, spans_pref);
    if (!err.empty()) { g_err = err; return TPS_E_CAPACITY; }
    out8[0] = a.spans_per_tile; out8[1] = a.span_dw; out8[2] = a.blk_log2; out8[3] = a.q; out8[4] = a.r;
    out8[5] = a.lw; out8[6] = a.seq_dw; out8[7] = (int32_t)(tps::lds_dwords(a) * 4);
    return TPS_OK;
}
