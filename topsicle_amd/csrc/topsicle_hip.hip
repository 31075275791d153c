// topsicle_hip.hip -- libtopsicle_hip.so: kernels + C ABI (include/topsicle_hip.h).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC  (see __graft_entry__.build()).
// gfx950 only; there is no CPU fallback in this library.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "tps_device.h"
#include "tps_pack.h"
#include "tps_plan.h"

// ======================================================================== kernels
// The scan kernels live in tps_kernels.h (one macro body, instantiated per slide / table kind / outputs).  Compiled alone,
// this file holds all of them; in the split build (-DTPS_KGROUP=0) only the plain and pair kernels, the others come
// from tps_kernels.hip objects.
#include "tps_kernels.h"

extern "C" __global__ void __launch_bounds__(tps::NT * tps::WPG) tps_binseg_kernel(tps::BinsegArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t smem[tps::WPG * tps::BINSEG_SMEM_DW];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t r = (int64_t)blockIdx.x * tps::WPG + wave;
    if (r >= a.n_reads) return;
    tps::binseg_read(a, r, smem + wave * tps::BINSEG_SMEM_DW);
}

// every m-th window of a base-slide scan, and the change point of the compacted series (tps_plan.h: stride_base)
extern "C" __global__ void __launch_bounds__(tps::NT * tps::WPG) tps_stride_kernel(tps::StrideArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];      // per wave: the change point's scratch, then the series as 16-bit values
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t r = (int64_t)blockIdx.x * tps::WPG + wave;
    if (r >= a.n_reads) return;
    uint32_t* mine = smem + wave * (tps::BINSEG_SMEM_DW + a.s16_dw);
    tps::stride_read(a, r, mine, a.s16_dw ? (uint16_t*)(mine + tps::BINSEG_SMEM_DW) : nullptr);
}

extern "C" __global__ void __launch_bounds__(tps::NT * tps::WPG) tps_followers_kernel(tps::FollowArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t smem[tps::WPG * tps::FOLLOW_LDS_DW];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t r = (int64_t)blockIdx.x * tps::WPG + wave;
    if (r >= a.n_reads) return;
    tps::followers_read(a, r, smem + wave * tps::FOLLOW_LDS_DW);
}

// ASCII -> packed batch (tps_pack.h), one workgroup per read, one thread per word of 16 bases.  Runs once per
// tps_batch_upload, right behind the copy of the ASCII bytes; the scan kernels only ever see the packed batch.
// bits 1-2 of an ASCII letter: A,C,T,G (either case) -> 0,1,2,3; one v_dot4_u32_u8 packs 4 bases; a v_perm rebuilds the
// lower-case letter each code stands for and any byte that differs from it in more than the case bit is invalid.
extern "C" __global__ void __launch_bounds__(256) tps_pack_kernel(const uint8_t* bases, const int64_t* offsets, tps_read_desc* desc,
                                                                   uint32_t* seq2, uint16_t* inv, int64_t n_reads, uint32_t* any_flag) {
    const int64_t r = blockIdx.x;
    if (r >= n_reads) return;
    const int64_t off = offsets[r];
    const int64_t L = desc[r].len;
    const int64_t w0 = desc[r].word_off;
    const int64_t nw = ((L + 63) / 64) * 4;
    uint32_t any = 0;
    for (int64_t w = threadIdx.x; w < nw; w += 256) {
        const int64_t left = L - 16 * w;               // bases of this word (<= 0: padding up to the quad boundary)
        uint32_t packed = 0, bad = 0;
        if (left > 0) {
            // the read starts at an arbitrary byte: five aligned dwords + byte funnel shifts
            const uintptr_t p = (uintptr_t)(bases + off + 16 * w);
            const uint32_t* q = (const uint32_t*)(p & ~(uintptr_t)3);
            const uint32_t sh = (uint32_t)(p & 3u);
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
            uint32_t v[4] = {__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                             __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh)};
            uint32_t any_bad = 0, b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t y = v[i] & 0x06060606u;
                packed |= (tps::udot4(y, 0x40100401u) >> 1) << (8 * i);
                b[i] = (v[i] ^ tps::perm(0x00670074u, 0x00630061u, y)) & 0xDFDFDFDFu;
                any_bad |= b[i];
            }
            if (any_bad) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((b[i] >> (8 * j)) & 255u) bad |= 1u << (4 * i + j);
            }
            if (left < 16) {
                packed &= (1u << (2 * (int)left)) - 1u;
                bad &= (1u << (int)left) - 1u;
            }
        }
        seq2[w0 + w] = packed;
        inv[w0 + w] = (uint16_t)bad;
        any |= bad;
    }
    if (any) {
        atomicOr(&desc[r].flags, TPS_RD_HAS_INVALID);
        *any_flag = 1u;                            // (mapped host word: the batch has a read with a non-ACGT letter)
    }
}

// ======================================================================== host side
namespace {

thread_local std::string g_err;

// Pinned host buffers handed out by tps_host_alloc, process-wide: the pipeline allocates its staging pool through ONE context and
// every context uploads from it (hipHostMallocPortable: pinned for all devices); tps_batch_upload_packed copies asynchronously
// whenever its sources lie in registered buffers, whoever allocated them.
std::mutex g_pinned_mu;
std::map<uintptr_t, size_t> g_pinned;            // base -> bytes
bool is_pinned(const void* p, size_t bytes) {
    if (!p || !bytes) return true;
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    auto it = g_pinned.upper_bound((uintptr_t)p);
    if (it == g_pinned.begin()) return false;
    --it;
    return (uintptr_t)p + bytes <= it->first + it->second;
}

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(TPS_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr int64_t PAD = 64;          // readable bytes before / after the concatenated bases
constexpr int INTERNAL_SLOT = TPS_MAX_SLOTS;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool borrowed = false;               // p belongs to another context's slot (tps_batch_share): never freed, never grown here
    int ensure(size_t bytes) {
        if (borrowed) { p = nullptr; cap = 0; borrowed = false; }
        if (bytes <= cap) return TPS_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return TPS_OK;
    }
    void release() { if (p && !borrowed) (void)hipFree(p); p = nullptr; cap = 0; borrowed = false; }
    void borrow(const DevBuf& o) { release(); p = o.p; cap = o.cap; borrowed = true; }
};

struct Slot {
    DevBuf seq2, inv, desc, tails, results, c_start, c_end, win_off, win_off16, sums, raw, stamps, lc, order;
    int64_t n_words = 0;                 // words of seq2 / inv in use
    bool inv_valid = true;               // false: the batch came packed without an inv array (no read is flagged)
    bool any_invalid = true;             // some read of the batch is flagged TPS_RD_HAS_INVALID (the kernels then stage the invalid masks)
    std::vector<int64_t> h_offsets;      // host copy of offsets (n+1)
    std::vector<int64_t> h_win_off;      // window layout of the last plan
    std::vector<int64_t> h_win_off16;    // ... of the fused kernels' 16-bit sums on the device (every read padded to a multiple of 8 windows)
    std::vector<uint16_t> h_sums16;      // download scratch
    std::vector<int32_t> h_order;        // dispatch order of the last plan (tps::plan_dispatch_order; empty = file order)
    tps_read_result* h_results = nullptr;   // pinned
    size_t h_results_cap = 0;
    int64_t n = -1;
    bool has_tails = false;
    // cached plan
    bool planned = false;
    tps_params plan_prm{};
    int plan_k = 0, plan_p = 0;
    uint32_t plan_dup = 0, plan_so = 0;   // the plan depends on whether the pattern list holds duplicate / self-overlapping k-mers
    tps::ScanArgs args{};
    size_t lds_bytes = 0;
    const char* kernel_name = "";       // what the last scan of this slot launched
    bool scanned = false;
    uint32_t last_flags = 0;
    // scans at a multiple of a fused kernel's slide (tps::stride_base): the base-slide scan runs in a slot of its own that borrows this
    // slot's batch; `sub_stale` = the batch has changed since it was borrowed
    Slot* sub = nullptr;
    bool sub_stale = true;
    // (a base-slide slot) where the scan may store every 2nd raw row itself: the requesting scan's buffer and window layout
    void* ext_raw = nullptr;
    const int64_t* ext_raw_win_off = nullptr;
    bool rows_inline = false;            // ... and it did (last scan)
    int stride_base = 0;                 // of the cached plan: 0 = the planned kernel runs itself
    std::string info_name;               // kernel_name of a strided scan ("<base kernel> every <m>th window")
};

struct EventPair { hipEvent_t a, b; };

}  // namespace

struct tps_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop{};
    // dev: [4^k masks][pair table, k <= 4][ready-made LDS images of the table for the fused kernels, each a multiple of 4 dwords:
    // mask << 16 | count, one-hot fields, 16-bit masks -- a workgroup copies its image in 16-byte pieces instead of converting it]
    struct Table { DevBuf dev; int P = 0, k = 0; std::string key; tps::PatInfo pat{}; size_t off_e32 = 0, off_fld = 0, off_m16 = 0, off_f16 = 0, off_p16 = 0, off_pfld = 0; };
    std::deque<Table> tables;         // resident pattern tables (deque: pointers to elements stay valid)
    Table* lut_cur = nullptr;
    size_t table_rr = 0;
    DevBuf follow_picks, follow_hist; // outputs of tps_batch_kmer_followers
    DevBuf ascii, ascii_off;          // staging of tps_batch_upload: ASCII bases + offsets, packed on the device right after the copy
    std::set<void*> pinned;           // host buffers handed out by tps_host_alloc
    std::vector<tps_read_desc> h_desc;   // scratch of the ASCII upload path
    tps::PatInfo pat{};
    bool have_pat = false;
    Slot slots[TPS_MAX_SLOTS + 1];
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;      // next free event pair (never rewound: re-recording old events was measured slow)
    size_t ev_base = 0;      // first pair of the current measurement window
    tps::PlanKnobs knobs{};           // tps_ctx_debug_option: tests / diagnostics only; the library reads no environment
    int want_stamps = 0;
    int no_events = 0;
    int no_inline_rows = 0;           // tps_ctx_debug_option "no_inline_rows": strided scans at twice the base slide copy their raw rows like the others (A/B of ScanArgs::raw_m)
    int no_stride = 0;                // tps_ctx_debug_option "no_stride": slides that are a multiple of a fused kernel's keep the generic kernel (A/B of tps::stride_base)
    int file_order = 0;               // tps_ctx_debug_option "file_order": wave slot i takes read i whatever the reads' lengths (A/B of tps::plan_dispatch_order)
    int event_stride = 1;             // time every event_stride-th launch (tps_ctx_debug_option "event_stride"): timing costs ~3.5 us per launch
    uint64_t launch_seq = 0;
    size_t lds_set_v[56] = {0};
    uint32_t* h_flag = nullptr;      // mapped host word the pack kernel raises when a read has a non-ACGT letter
    hipEvent_t share_ev = nullptr;   // tps_batch_share: orders this context's stream behind the lender's upload
    void* raw_stage[2] = {nullptr, nullptr};    // tps_batch_raw_to_fd: two pinned pieces, one being written while the other is filled
    hipEvent_t raw_ev[2] = {nullptr, nullptr};
};

namespace {

int bind(tps_ctx* c) {
    if (!c) return fail(TPS_E_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    return TPS_OK;
}

using tps::window_count;

int plan_lds(tps_ctx* c, Slot& sl, const tps_params& prm, int64_t max_nwin) {
    const size_t lds_max = c->prop.sharedMemPerBlock > 0 ? std::min<size_t>(c->prop.sharedMemPerBlock, 160 * 1024) : 64 * 1024;
    sl.args.pat = c->pat;                          // the plan looks at dup_mask
    std::string err = tps::plan_geometry(sl.args, prm, c->pat.k, c->pat.P, max_nwin, (int64_t)lds_max / 4, c->knobs);
    if (!err.empty()) return fail(TPS_E_CAPACITY, "%s", err.c_str());
    sl.lds_bytes = (size_t)tps::wg_lds_dwords(sl.args) * 4;
    return TPS_OK;
}

int check_params(const tps_params& p) {
    if (p.window < 1 || p.slide < 1 || p.trimfirst < 0 || p.maxlen < 0 || p.no_bp < 0)
        return fail(TPS_E_ARG, "bad window/slide/trimfirst/maxlen/no_bp");
    if ((p.flags & TPS_F_BINSEG) && (p.jump < 1 || p.min_size < 1))
        return fail(TPS_E_ARG, "bad jump/min_size");
    if (p.slide > 4096 || p.window > 65536 || p.jump > 4096) return fail(TPS_E_CAPACITY, "window/slide/jump too large");
    if ((p.flags & TPS_F_WINDOWS) && p.jump < 1) return fail(TPS_E_ARG, "jump must be >= 1");
    return TPS_OK;
}

void reset_slot(Slot& sl, int64_t n, int64_t n_words) {
    sl.sub_stale = true;
    sl.n = n;
    sl.n_words = n_words;
    sl.has_tails = false;
    sl.planned = false;
    sl.scanned = false;
}

int ensure_packed(Slot& sl, int64_t n, int64_t n_words) {
    int rc;
    if ((rc = sl.seq2.ensure((size_t)std::max<int64_t>(n_words, 4) * 4))) return rc;
    if ((rc = sl.inv.ensure((size_t)std::max<int64_t>(n_words, 4) * 2))) return rc;
    if ((rc = sl.desc.ensure((size_t)std::max<int64_t>(n, 1) * sizeof(tps_read_desc)))) return rc;
    return TPS_OK;
}

// ASCII batch -> HBM -> packed batch (tps_pack_kernel).  The ASCII copy only lives in the context's staging buffer.
int do_upload(tps_ctx* c, Slot& sl, const uint8_t* bases, const int64_t* offsets, int64_t n) {
    if (n < 0 || !offsets || (n > 0 && !bases)) return fail(TPS_E_ARG, "bad batch pointers");
    if (offsets[0] != 0) return fail(TPS_E_ARG, "offsets[0] must be 0");
    for (int64_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i]) return fail(TPS_E_ARG, "offsets not monotone at %lld", (long long)i);
        if (offsets[i + 1] - offsets[i] > 0x7FFFFFFFll) return fail(TPS_E_CAPACITY, "read %lld is longer than 2^31 - 1 bases", (long long)i);
    }
    const int64_t total = offsets[n];
    c->h_desc.resize((size_t)std::max<int64_t>(n, 1));
    const int64_t n_words = tps::pack_layout(offsets, n, c->h_desc.data());
    int rc;
    if ((rc = ensure_packed(sl, n, n_words))) return rc;
    if ((rc = c->ascii.ensure((size_t)(total + 2 * PAD + 32)))) return rc;
    if ((rc = c->ascii_off.ensure((size_t)(n + 1) * 8))) return rc;
    uint8_t* d = (uint8_t*)c->ascii.p;
    if (total) HIP_TRY(hipMemcpyAsync(d + PAD, bases, (size_t)total, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ascii_off.p, offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    if (n) HIP_TRY(hipMemcpyAsync(sl.desc.p, c->h_desc.data(), (size_t)n * sizeof(tps_read_desc), hipMemcpyHostToDevice, c->stream));
    if (n) {
        if (!c->h_flag) HIP_TRY(hipHostMalloc((void**)&c->h_flag, 64, hipHostMallocMapped));
        *c->h_flag = 0u;
        hipLaunchKernelGGL(tps_pack_kernel, dim3((unsigned)n), dim3(256), 0, c->stream, (const uint8_t*)d + PAD,
                           (const int64_t*)c->ascii_off.p, (tps_read_desc*)sl.desc.p, (uint32_t*)sl.seq2.p, (uint16_t*)sl.inv.p, n, c->h_flag);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(c->stream));       // the caller's buffers and h_desc are free again
    sl.h_offsets.assign(offsets, offsets + n + 1);
    sl.inv_valid = true;
    sl.any_invalid = n > 0 && *(volatile uint32_t*)c->h_flag != 0u;
    reset_slot(sl, n, n_words);
    return TPS_OK;
}

// Host-packed batch: three plain copies.  Pinned sources (tps_host_alloc) are copied asynchronously.
int do_upload_packed(tps_ctx* c, Slot& sl, const uint32_t* seq2, const uint16_t* inv, const tps_read_desc* desc, int64_t n, int64_t n_words) {
    if (n < 0 || n_words < 0 || (n > 0 && !desc) || (n_words > 0 && !seq2)) return fail(TPS_E_ARG, "bad packed batch pointers");
    sl.h_offsets.resize((size_t)n + 1);
    int64_t acc = 0, need = 0;
    bool flagged = false;
    for (int64_t i = 0; i < n; ++i) {
        const tps_read_desc& d = desc[i];
        if (d.len < 0 || d.word_off < 0 || (d.word_off & 3) || d.word_off + tps::packed_words(d.len) > n_words)
            return fail(TPS_E_ARG, "read %lld: descriptor outside the packed batch (word_off %lld, len %d, %lld words)", (long long)i,
                        (long long)d.word_off, d.len, (long long)n_words);
        sl.h_offsets[(size_t)i] = acc;
        acc += d.len;
        need = std::max(need, d.word_off + tps::packed_words(d.len));
        flagged = flagged || (d.flags & TPS_RD_HAS_INVALID);
    }
    sl.h_offsets[(size_t)n] = acc;
    if (flagged && !inv) return fail(TPS_E_ARG, "a read is flagged TPS_RD_HAS_INVALID but inv is NULL");
    int rc;
    if ((rc = ensure_packed(sl, n, n_words))) return rc;
    if (n_words) HIP_TRY(hipMemcpyAsync(sl.seq2.p, seq2, (size_t)n_words * 4, hipMemcpyHostToDevice, c->stream));
    if (n_words && inv) HIP_TRY(hipMemcpyAsync(sl.inv.p, inv, (size_t)n_words * 2, hipMemcpyHostToDevice, c->stream));
    if (n) HIP_TRY(hipMemcpyAsync(sl.desc.p, desc, (size_t)n * sizeof(tps_read_desc), hipMemcpyHostToDevice, c->stream));
    const bool all_pinned = is_pinned(seq2, (size_t)n_words * 4) && is_pinned(inv, inv ? (size_t)n_words * 2 : 0) && is_pinned(desc, (size_t)n * sizeof(tps_read_desc));
    if (!all_pinned) HIP_TRY(hipStreamSynchronize(c->stream));     // ordinary memory: the copy is over when the call returns
    sl.inv_valid = inv != nullptr;
    sl.any_invalid = flagged;
    reset_slot(sl, n, n_words);
    return TPS_OK;
}

bool same_params(const tps_params& x, const tps_params& y) { return memcmp(&x, &y, sizeof x) == 0; }

int do_scan(tps_ctx* c, Slot& sl, const tps_params& prm, bool inner = false, hipEvent_t ev_start = nullptr);

// The scan of `sl` at prm.slide = m x base: the fused kernel of the base slide in sl.sub (which borrows the batch), then
// tps_stride_kernel.  Outputs land in sl's buffers in the generic kernel's layout (int32 sums, rows at win_off).
int do_scan_strided(tps_ctx* c, Slot& sl, const tps_params& prm, int base, hipEvent_t ev_a, hipEvent_t ev_b) {
    int rc;
    if (!sl.sub) sl.sub = new Slot();
    Slot& sb = *sl.sub;
    if (sl.sub_stale) {
        sb.seq2.borrow(sl.seq2);
        sb.inv.borrow(sl.inv);
        sb.desc.borrow(sl.desc);
        sb.h_offsets = sl.h_offsets;
        sb.inv_valid = sl.inv_valid;
        sb.any_invalid = sl.any_invalid;
        reset_slot(sb, sl.n, sl.n_words);
        sl.sub_stale = false;
    }
    sb.tails.borrow(sl.tails);
    sb.has_tails = sl.has_tails;
    tps_params pb = prm;
    pb.slide = base;
    pb.flags = (prm.flags | TPS_F_STORE_SUMS) & ~(uint32_t)TPS_F_BINSEG;
    sb.ext_raw = (prm.slide == 2 * base && !c->no_inline_rows) ? sl.args.raw : nullptr;
    sb.ext_raw_win_off = (const int64_t*)sl.win_off.p;
    if ((rc = do_scan(c, sb, pb, true, ev_a))) return rc;
    if (!sb.args.variant) return fail(TPS_E_STATE, "strided scan: the base slide %d did not plan a fused kernel", base);
    const int64_t n = sl.n;
    tps::StrideArgs a{};
    a.base_results = (const tps_read_result*)sb.results.p;
    a.base_sums16 = (const uint16_t*)sb.sums.p;
    a.base_win_off16 = (const int64_t*)sb.win_off16.p;
    a.base_raw = ((prm.flags & TPS_F_STORE_RAW) && !sb.rows_inline) ? (const uint8_t*)sb.raw.p : nullptr;
    a.base_win_off = (const int64_t*)sb.win_off.p;
    a.win_off = (const int64_t*)sl.win_off.p;
    a.sums = sl.args.sums;
    a.raw = sb.rows_inline ? nullptr : sl.args.raw;
    a.results = sl.h_results;
    a.n_reads = n;
    a.m = prm.slide / base;
    a.P = c->pat.P;
    a.n_patterns = c->pat.P;
    a.jump = prm.jump;
    a.min_size = prm.min_size;
    a.binseg = (prm.flags & TPS_F_BINSEG) ? 1 : 0;
    int64_t mx = 0;
    for (int64_t i = 0; i < n; ++i) mx = std::max(mx, sl.h_win_off[(size_t)i + 1] - sl.h_win_off[(size_t)i]);
    a.s16_dw = (int32_t)((((mx + 1) / 2) + 3) & ~3ll);
    if (a.s16_dw > 3072) a.s16_dw = 0;                 // (12 KB per wave: five workgroups per CU; longer series are read back from HBM)
    const size_t lds = (size_t)tps::WPG * (size_t)(tps::BINSEG_SMEM_DW + a.s16_dw) * 4;
    void* kargs[] = {(void*)&a};
    HIP_TRY(hipExtLaunchKernel((const void*)tps_stride_kernel, dim3((unsigned)((n + tps::WPG - 1) / tps::WPG)), dim3(tps::NT * tps::WPG), kargs, lds, c->stream,
                               nullptr, ev_b, 0));
    // the step-1 counts are the base scan's
    sl.c_start.borrow(sb.c_start);
    sl.c_end.borrow(sb.c_end);
    sl.info_name = std::string(sb.kernel_name) + (a.m == 2 ? " every 2nd window" : a.m == 3 ? " every 3rd window" : " every " + std::to_string(a.m) + "th window");
    sl.kernel_name = sl.info_name.c_str();
    sl.lds_bytes = sb.lds_bytes;
    sl.args.wpg = sb.args.wpg;
    return TPS_OK;
}

int do_scan(tps_ctx* c, Slot& sl, const tps_params& prm, bool inner, hipEvent_t ev_start) {
    int rc;
    if (!c->have_pat) return fail(TPS_E_PATTERN, "tps_set_patterns has not been called");
    if (sl.n < 0) return fail(TPS_E_STATE, "no batch uploaded in this slot");
    if ((rc = check_params(prm))) return rc;
    if (!(prm.flags & TPS_F_STEP1) && (prm.flags & TPS_F_TAILS_IN) && !sl.has_tails)
        return fail(TPS_E_STATE, "TPS_F_TAILS_IN without tps_batch_set_tails");
    const int64_t n = sl.n;
    const int P = c->pat.P;
    if (!sl.planned || !same_params(prm, sl.plan_prm) || sl.plan_k != c->pat.k || sl.plan_p != P ||
        (sl.plan_dup != 0) != (c->pat.dup_mask != 0) || (sl.plan_so != 0) != (c->pat.so_mask != 0)) {
        sl.h_win_off.resize((size_t)n + 1);
        sl.h_win_off16.resize((size_t)n + 1);
        int64_t acc = 0, acc16 = 0, mx = 0;
        std::vector<int64_t> nwv((size_t)n);
        std::vector<uint8_t> longer((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            sl.h_win_off[(size_t)i] = acc;
            sl.h_win_off16[(size_t)i] = acc16;
            const int64_t len = sl.h_offsets[i + 1] - sl.h_offsets[i];
            int64_t nw = window_count(len, prm.window, prm.slide, prm.trimfirst, prm.maxlen);
            mx = std::max(mx, nw);
            acc += nw;
            acc16 += tps::sums16_slots(nw);
            nwv[(size_t)i] = (prm.flags & TPS_F_WINDOWS) ? nw : 0;
            longer[(size_t)i] = !(prm.flags & TPS_F_STEP1) || len > prm.min_len;      // (scan_read: pass = L > min_len && ...)
        }
        sl.h_win_off[(size_t)n] = acc;
        sl.h_win_off16[(size_t)n] = acc16;
        sl.h_order.clear();
        if (!c->file_order) tps::plan_dispatch_order(nwv.data(), longer.data(), n, sl.h_order);
        if (!sl.h_order.empty()) {
            if ((rc = sl.order.ensure((size_t)n * 4))) return rc;
            HIP_TRY(hipMemcpyAsync(sl.order.p, sl.h_order.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        }
        sl.args = tps::ScanArgs{};
        sl.args.val_on = sl.any_invalid ? 1 : 0;
        if ((rc = plan_lds(c, sl, prm, mx))) return rc;
        sl.stride_base = 0;
        if (!inner && !c->no_stride) {
            const size_t lds_max = c->prop.sharedMemPerBlock > 0 ? std::min<size_t>(c->prop.sharedMemPerBlock, 160 * 1024) : 64 * 1024;
            int64_t max_len = 0;
            for (int64_t i = 0; i < n; ++i) max_len = std::max(max_len, sl.h_offsets[i + 1] - sl.h_offsets[i]);
            sl.stride_base = tps::stride_base(sl.args, prm, c->pat.k, P, [&](int s0) { return window_count(max_len, prm.window, s0, prm.trimfirst, prm.maxlen); },
                                              (int64_t)lds_max / 4, c->knobs);
        }
        if ((rc = sl.win_off.ensure((size_t)(n + 1) * 8))) return rc;
        HIP_TRY(hipMemcpyAsync(sl.win_off.p, sl.h_win_off.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, c->stream));
        if (sl.args.variant) {                          // fused kernels: the padded layout of their 16-bit sums
            if ((rc = sl.win_off16.ensure((size_t)(n + 1) * 8))) return rc;
            HIP_TRY(hipMemcpyAsync(sl.win_off16.p, sl.h_win_off16.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));       // h_win_off may be reused by the caller's next plan
        sl.plan_prm = prm;
        sl.plan_k = c->pat.k;
        sl.plan_p = P;
        sl.plan_dup = c->pat.dup_mask;
        sl.plan_so = c->pat.so_mask;
        sl.planned = true;
    }
    const int64_t total_win = sl.h_win_off[(size_t)n];
    if (sl.h_results_cap < (size_t)n) {
        if (sl.h_results) (void)hipHostFree(sl.h_results);
        sl.h_results = nullptr;
        size_t want = (size_t)n + (size_t)n / 8 + 16;
        HIP_TRY(hipHostMalloc((void**)&sl.h_results, want * sizeof(tps_read_result), hipHostMallocMapped));
        sl.h_results_cap = want;
    }
    tps::ScanArgs& a = sl.args;
    a.seq2 = (const uint32_t*)sl.seq2.p;
    a.inv = (const uint16_t*)sl.inv.p;
    a.desc = (const tps_read_desc*)sl.desc.p;
    a.tails_in = ((prm.flags & TPS_F_TAILS_IN) && !(prm.flags & TPS_F_STEP1)) ? (const uint8_t*)sl.tails.p : nullptr;
    a.lut = (const uint32_t*)c->lut_cur->dev.p;
    a.pair_img = a.lut + (a.pair16 ? c->lut_cur->off_p16 : (a.lut_fields && a.pair_n) ? c->lut_cur->off_pfld : (size_t)a.lut_n);      // (the 32-bit pair table follows the 4^k masks)
    a.lut_img = a.lut + ((a.lut16 && a.lut_fields) ? c->lut_cur->off_f16 : a.lut16 ? c->lut_cur->off_m16 : a.lut_fields ? c->lut_cur->off_fld : c->lut_cur->off_e32);
    a.results = sl.h_results;                      // (written by the kernel straight into mapped pinned host memory)
    if (inner) {                                   // a base-slide scan: its results are read by tps_stride_kernel, not by the host -- they stay in HBM
        if ((rc = sl.results.ensure((size_t)std::max<int64_t>(n, 1) * sizeof(tps_read_result)))) return rc;
        a.results = (tps_read_result*)sl.results.p;
    }
    a.c_start = a.c_end = nullptr;
    if ((prm.flags & TPS_F_STEP1) && !sl.stride_base) {       // (a strided scan borrows the base scan's counts)
        if ((rc = sl.c_start.ensure((size_t)std::max<int64_t>(n * P, 1) * 4))) return rc;
        if ((rc = sl.c_end.ensure((size_t)std::max<int64_t>(n * P, 1) * 4))) return rc;
        a.c_start = (int32_t*)sl.c_start.p;
        a.c_end = (int32_t*)sl.c_end.p;
    }
    a.win_off = (const int64_t*)sl.win_off.p;
    a.order = sl.h_order.empty() ? nullptr : (const int32_t*)sl.order.p;
    a.sums = nullptr;
    a.sums16 = nullptr;
    a.win_off16 = nullptr;
    a.raw = nullptr;
    if (prm.flags & TPS_F_WINDOWS) {
        // S_w always goes to HBM: it is the step's output and the only copy the exact change-point fallback can re-read;
        // TPS_F_STORE_SUMS just makes it downloadable.  The fused kernels write 16-bit values (2 B per window, every read
        // padded to 8 windows), the generic kernel int32; tps_batch_window_sums hands out int32 either way.
        if (a.variant) {
            if ((rc = sl.sums.ensure((size_t)std::max<int64_t>(sl.h_win_off16[(size_t)n], 8) * 2))) return rc;
            a.sums16 = (uint16_t*)sl.sums.p;
            a.win_off16 = (const int64_t*)sl.win_off16.p;
        } else {
            if ((rc = sl.sums.ensure((size_t)std::max<int64_t>(total_win, 1) * 4))) return rc;
            a.sums = (int32_t*)sl.sums.p;
        }
    }
    a.raw_m = 0;
    a.raw_win_off = nullptr;
    sl.rows_inline = false;
    if (prm.flags & TPS_F_STORE_RAW) {
        // a base-slide scan at half the requested slide: the per-pattern tiles store the even windows' rows straight into the requesting
        // scan's layout (rows of whole dwords; a clean batch -- the fallback tile of a batch with non-ACGT letters writes whole rows by
        // address; the self-overlap tiles' repairs follow the same mapping)
        if (inner && sl.ext_raw && a.variant && a.pp_d >= 0 && (c->pat.so_mask == 0 ? a.pp_d == 0 : a.pp_d > 0) && (P & 3) == 0 && !sl.any_invalid) {
            a.raw = (uint8_t*)sl.ext_raw;
            a.raw_m = 2;
            a.raw_win_off = sl.ext_raw_win_off;
            sl.rows_inline = true;
        } else {
            if ((rc = sl.raw.ensure((size_t)std::max<int64_t>(total_win * P, 1)))) return rc;
            a.raw = (uint8_t*)sl.raw.p;
        }
    }
    a.lc_scratch = nullptr;
    if (a.lc_global) {
        if ((rc = sl.lc.ensure((size_t)std::max<int64_t>(n, 1) * (size_t)a.lc_stride * 2))) return rc;
        a.lc_scratch = (uint16_t*)sl.lc.p;
    }
    a.stamps = nullptr;
    if (c->want_stamps) {
        if ((rc = sl.stamps.ensure((size_t)std::max<int64_t>(n, 1) * 16 * 8))) return rc;
        HIP_TRY(hipMemsetAsync(sl.stamps.p, 0, (size_t)std::max<int64_t>(n, 1) * 16 * 8, c->stream));
        a.stamps = (uint64_t*)sl.stamps.p;
    }
    a.n_reads = n;
    a.pat = c->pat;
    a.prm = prm;
    sl.last_flags = prm.flags;
    if (n == 0) { sl.scanned = true; return TPS_OK; }

    const void* kfn;
    int kidx;
    const bool so = a.pat.so_mask != 0;
    const bool pair = a.pair_n != 0;
    const bool want_raw = a.raw != nullptr;
    {
        // kernel family by table and outputs: [slide 5..8] x {plain, pair table, raw rows, self-overlap sums, self-overlap raw}
        struct K { const void* fn; const char* name; };
        static const K plain[4] = {{(const void*)tps_scan_kernel_s5, "tps_scan_kernel_s5"}, {(const void*)tps_scan_kernel_s6, "tps_scan_kernel_s6"},
                                   {(const void*)tps_scan_kernel_s7, "tps_scan_kernel_s7"}, {(const void*)tps_scan_kernel_s8, "tps_scan_kernel_s8"}};
        static const K pairk[4] = {{(const void*)tps_scan_kernel_s5p, "tps_scan_kernel_s5p"}, {(const void*)tps_scan_kernel_s6p, "tps_scan_kernel_s6p"},
                                   {(const void*)tps_scan_kernel_s7p, "tps_scan_kernel_s7p"}, {(const void*)tps_scan_kernel_s8p, "tps_scan_kernel_s8p"}};
        static const K rawk[4] = {{(const void*)tps_scan_kernel_s5r, "tps_scan_kernel_s5r"}, {(const void*)tps_scan_kernel_s6r, "tps_scan_kernel_s6r"},
                                  {(const void*)tps_scan_kernel_s7r, "tps_scan_kernel_s7r"}, {(const void*)tps_scan_kernel_s8r, "tps_scan_kernel_s8r"}};
        static const K sok[4] = {{(const void*)tps_scan_kernel_s5so, "tps_scan_kernel_s5so"}, {(const void*)tps_scan_kernel_s6so, "tps_scan_kernel_s6so"},
                                 {(const void*)tps_scan_kernel_s7so, "tps_scan_kernel_s7so"}, {(const void*)tps_scan_kernel_s8so, "tps_scan_kernel_s8so"}};
        static const K sork[4] = {{(const void*)tps_scan_kernel_s5sor, "tps_scan_kernel_s5sor"}, {(const void*)tps_scan_kernel_s6sor, "tps_scan_kernel_s6sor"},
                                  {(const void*)tps_scan_kernel_s7sor, "tps_scan_kernel_s7sor"}, {(const void*)tps_scan_kernel_s8sor, "tps_scan_kernel_s8sor"}};
        static const K sorhk[4] = {{(const void*)tps_scan_kernel_s5sorh, "tps_scan_kernel_s5sorh"}, {(const void*)tps_scan_kernel_s6sorh, "tps_scan_kernel_s6sorh"},
                                   {(const void*)tps_scan_kernel_s7sorh, "tps_scan_kernel_s7sorh"}, {(const void*)tps_scan_kernel_s8sorh, "tps_scan_kernel_s8sorh"}};
        static const K pairqk[4] = {{(const void*)tps_scan_kernel_s5q, "tps_scan_kernel_s5q"}, {(const void*)tps_scan_kernel_s6q, "tps_scan_kernel_s6q"},
                                    {(const void*)tps_scan_kernel_s7q, "tps_scan_kernel_s7q"}, {(const void*)tps_scan_kernel_s8q, "tps_scan_kernel_s8q"}};
        static const K solk[4] = {{(const void*)tps_scan_kernel_s5sol, "tps_scan_kernel_s5sol"}, {(const void*)tps_scan_kernel_s6sol, "tps_scan_kernel_s6sol"},
                                  {(const void*)tps_scan_kernel_s7sol, "tps_scan_kernel_s7sol"}, {(const void*)tps_scan_kernel_s8sol, "tps_scan_kernel_s8sol"}};
        static const K plainx[6] = {{(const void*)tps_scan_kernel_s3, "tps_scan_kernel_s3"}, {(const void*)tps_scan_kernel_s4, "tps_scan_kernel_s4"}, {(const void*)tps_scan_kernel_s9, "tps_scan_kernel_s9"},
                                    {(const void*)tps_scan_kernel_s10, "tps_scan_kernel_s10"}, {(const void*)tps_scan_kernel_s11, "tps_scan_kernel_s11"},
                                    {(const void*)tps_scan_kernel_s12, "tps_scan_kernel_s12"}};
        static const K pairx[6] = {{(const void*)tps_scan_kernel_s3p, "tps_scan_kernel_s3p"}, {(const void*)tps_scan_kernel_s4p, "tps_scan_kernel_s4p"}, {(const void*)tps_scan_kernel_s9p, "tps_scan_kernel_s9p"},
                                   {(const void*)tps_scan_kernel_s10p, "tps_scan_kernel_s10p"}, {(const void*)tps_scan_kernel_s11p, "tps_scan_kernel_s11p"},
                                   {(const void*)tps_scan_kernel_s12p, "tps_scan_kernel_s12p"}};
        if (tps::has_default_only_slide(a.variant)) {
            // the default kernels' other slides (sums only, no self-overlap: plan_geometry took the fused path for nothing else)
            const int xi = a.variant <= 4 ? a.variant - 3 : a.variant - 7;
            const K& k = (pair && !a.pair16) ? pairx[xi] : plainx[xi];
            kfn = k.fn;
            sl.kernel_name = k.name;
            kidx = 40 + ((pair && !a.pair16) ? 6 : 0) + xi;
        } else if (a.variant >= 5 && a.variant <= 8) {
            // sums only, self-overlap table: periods 2 .. 4 have their own kernels (96 registers, 5 waves per SIMD)
            const int fam = so ? (want_raw ? (a.lut16 ? 6 : 4) : (a.pp_d >= 2 && a.pp_d <= 4) ? 5 : 3) : want_raw ? 2 : (pair && a.pair16) ? 7 : pair ? 1 : 0;
            const K* tab[8] = {plain, pairk, rawk, sok, sork, solk, sorhk, pairqk};
            const K& k = tab[fam][a.variant - 5];
            kfn = k.fn;
            sl.kernel_name = k.name;
            kidx = 1 + fam * 4 + (a.variant - 5);
        } else {
            kfn = (const void*)tps_scan_kernel;
            sl.kernel_name = "tps_scan_kernel";
            kidx = 0;
        }
    }
    if (sl.lds_bytes > c->lds_set_v[kidx]) {
        HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sl.lds_bytes));
        c->lds_set_v[kidx] = sl.lds_bytes;
    }
    if (c->ev_used == c->ev_pool.size()) {
        if (c->ev_pool.size() >= 16384) {
            c->ev_used = c->ev_base = 0;                  // wrap: the measurement window restarts
        } else {
            // events are created in batches, never one per launch (hipEventCreate costs ~60 us)
            const size_t grow = c->ev_pool.empty() ? 512 : c->ev_pool.size();
            for (size_t i = 0; i < grow; ++i) {
                EventPair ep;
                HIP_TRY(hipEventCreate(&ep.a));
                HIP_TRY(hipEventCreate(&ep.b));
                c->ev_pool.push_back(ep);
            }
        }
    }
    const bool timed = !inner && !c->no_events && (c->launch_seq++ % (uint64_t)c->event_stride) == 0;
    EventPair& ep = c->ev_pool[timed ? c->ev_used++ : 0];
    if (sl.stride_base) {
        // (timed from the base kernel's start to the end of the compaction + change-point kernel)
        if ((rc = do_scan_strided(c, sl, prm, sl.stride_base, timed ? ep.a : nullptr, timed ? ep.b : nullptr))) return rc;
        sl.scanned = true;
        return TPS_OK;
    }
    {
        // hipExtLaunchKernel stamps the pair of events from the dispatch packet's own start / end timestamps:
        // no separate barrier packets around the kernel (two hipEventRecord calls cost ~6 us per launch and
        // kept consecutive launches from running back to back)
        const int64_t grid = (n + a.wpg - 1) / a.wpg;
        void* kargs[] = {(void*)&a};
        HIP_TRY(hipExtLaunchKernel(kfn, dim3((unsigned)grid), dim3(tps::NT * a.wpg), kargs, sl.lds_bytes, c->stream,
                                   inner ? ev_start : (timed ? ep.a : nullptr), timed ? ep.b : nullptr, 0));
    }
    sl.scanned = true;
    return TPS_OK;
}

Slot* get_slot(tps_ctx* c, int slot) {
    if (!c || slot < 0 || slot >= TPS_MAX_SLOTS) { fail(TPS_E_ARG, "slot out of range"); return nullptr; }
    return &c->slots[slot];
}

}  // namespace

// ======================================================================== C ABI
extern "C" {

int tps_abi_version(void) { return TPS_ABI_VERSION; }

const char* tps_last_error(void) { return g_err.c_str(); }

int tps_device_count(int* n) {
    if (!n) return fail(TPS_E_ARG, "null pointer");
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess) { *n = 0; return fail(TPS_E_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *n = cnt;
    return TPS_OK;
}

int tps_ctx_create(int device, tps_ctx** out) {
    if (!out) return fail(TPS_E_ARG, "null pointer");
    *out = nullptr;
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0)
        return fail(TPS_E_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= cnt) return fail(TPS_E_NO_DEVICE, "device %d out of range (0..%d)", device, cnt - 1);
    tps_ctx* c = new tps_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&c->prop, device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(TPS_E_HIP, "cannot initialise device %d", device);
    }
    *out = c;
    return TPS_OK;
}

int tps_ctx_destroy(tps_ctx* c) {
    if (!c) return TPS_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& sl : c->slots) {
        if (sl.sub) {
            Slot& sb = *sl.sub;
            sb.seq2.release(); sb.inv.release(); sb.desc.release(); sb.tails.release(); sb.results.release(); sb.win_off16.release();
            sb.c_start.release(); sb.c_end.release(); sb.win_off.release(); sb.sums.release(); sb.raw.release(); sb.stamps.release(); sb.lc.release(); sb.order.release();
            if (sb.h_results) (void)hipHostFree(sb.h_results);
            delete sl.sub;
            sl.sub = nullptr;
        }
        sl.seq2.release(); sl.inv.release(); sl.desc.release(); sl.tails.release(); sl.results.release(); sl.win_off16.release();
        sl.c_start.release(); sl.c_end.release(); sl.win_off.release(); sl.sums.release(); sl.raw.release(); sl.stamps.release(); sl.lc.release(); sl.order.release();
        if (sl.h_results) (void)hipHostFree(sl.h_results);
    }
    for (auto& t : c->tables) t.dev.release();
    c->ascii.release();
    c->ascii_off.release();
    c->follow_picks.release();
    c->follow_hist.release();
    for (void* hp : c->pinned) {
        { std::lock_guard<std::mutex> lk(g_pinned_mu); g_pinned.erase((uintptr_t)hp); }
        (void)hipHostFree(hp);
    }
    if (c->h_flag) (void)hipHostFree(c->h_flag);
    if (c->share_ev) (void)hipEventDestroy(c->share_ev);
    for (int i = 0; i < 2; ++i) {
        if (c->raw_stage[i]) (void)hipHostFree(c->raw_stage[i]);
        if (c->raw_ev[i]) (void)hipEventDestroy(c->raw_ev[i]);
    }
    c->pinned.clear();
    for (auto& ep : c->ev_pool) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    (void)hipStreamDestroy(c->stream);
    delete c;
    return TPS_OK;
}

int tps_device_info(tps_ctx* c, char* buf, int32_t buf_len) {
    if (!c || !buf || buf_len < 1) return fail(TPS_E_ARG, "bad arguments");
    snprintf(buf, (size_t)buf_len, "%s arch=%s CUs=%d LDS/block=%zu clock=%dkHz pci=%04x:%02x:%02x", c->prop.name, c->prop.gcnArchName,
             c->prop.multiProcessorCount, c->prop.sharedMemPerBlock, c->prop.clockRate, c->prop.pciDomainID, c->prop.pciBusID, c->prop.pciDeviceID);
    return TPS_OK;
}

int tps_batch_kernel_info(tps_ctx* c, int32_t slot, char* buf, int32_t buf_len) {
    if (!c || !buf || buf_len < 1) return fail(TPS_E_ARG, "bad arguments");
    Slot* sl = get_slot(c, slot);
    if (!sl) return TPS_E_ARG;
    if (!sl->scanned) return fail(TPS_E_STATE, "slot %d has not been scanned", slot);
    const size_t gran = (sl->lds_bytes + 1279) / 1280;
    const int wgs = gran ? (int)std::min<size_t>(128 / gran, 8) : 8;
    snprintf(buf, (size_t)buf_len, "%s lds=%zu wgs_per_cu=%d waves_per_wg=%d", sl->kernel_name, sl->lds_bytes, wgs, sl->args.wpg);
    return TPS_OK;
}

int tps_set_patterns(tps_ctx* c, const char* pats, int32_t P, int32_t k) {
    int rc;
    if ((rc = bind(c))) return rc;
    if (!pats) return fail(TPS_E_ARG, "null pattern table");
    if (P < 1 || k < 1 || P > TPS_MAX_PATTERNS || k > TPS_MAX_K) return fail(TPS_E_PATTERN, "%d patterns of %d letters not supported", P, k);
    // a few tables stay resident (a multi-k run switches between them once per batch and k: no copy, no sync)
    const std::string key(pats, (size_t)P * (size_t)k);
    for (auto& t : c->tables)
        if (t.P == P && t.k == k && t.key == key) {
            c->lut_cur = &t;
            c->pat = t.pat;
            c->have_pat = true;
            for (auto& sl : c->slots) sl.planned = false;
            return TPS_OK;
        }
    std::vector<uint32_t> lut;
    tps::PatInfo pi{};
    std::string err = tps::build_patterns(pats, P, k, lut, pi);
    if (!err.empty()) return fail(TPS_E_PATTERN, "%s", err.c_str());
    if (k <= 4) {
        // pair table for the pair kernels (k <= 4, see plan_geometry): entry of the (k+1)-mer code c =
        // packed entries (mask << 16 | popcount) of the k-mers at p and p+1, masks ORed, counts added
        const size_t n1 = lut.size(), n2 = n1 * 4;
        const uint32_t km = (uint32_t)n1 - 1u;
        lut.resize(n1 + n2);
        for (size_t cc = 0; cc < n2; ++cc) {
            const uint32_t m1 = lut[cc & km], m2 = lut[(cc >> 2) & km];
            lut[n1 + cc] = ((m1 | m2) << 16) | (uint32_t)(__builtin_popcount(m1) + __builtin_popcount(m2));
        }
    }
    size_t off_e32 = 0, off_fld = 0, off_m16 = 0, off_f16 = 0, off_p16 = 0, off_pfld = 0;
    if (!pi.hash_shift) {
        const size_t n1 = (size_t)1 << (2 * k), n4 = (n1 + 3) & ~(size_t)3, n16 = ((n1 + 1) / 2 + 3) & ~(size_t)3;
        off_e32 = (lut.size() + 3) & ~(size_t)3;       // (16-byte aligned: the kernels copy uint4)
        off_fld = off_e32 + n4;
        off_m16 = off_fld + n4;
        off_f16 = off_m16 + n16;
        lut.resize(off_f16 + n16, 0u);
        for (size_t i = 0; i < n1; ++i) {
            const uint32_t m = lut[i];
            lut[off_e32 + i] = (m << 16) | (uint32_t)__builtin_popcount(m);
            lut[off_fld + i] = tps::mask_to_fields(m);
            ((uint16_t*)&lut[off_m16])[i] = (uint16_t)m;
            // (field index of THE pattern: tables with duplicate k-mers never take the kernels that read this image)
            ((uint16_t*)&lut[off_f16])[i] = m ? (uint16_t)(1u << tps::pp_field(__builtin_ctz(m))) : (uint16_t)0;
        }
        if (k <= 4) {
            // pair table of one-hot FIELDS for the raw-row kernels' per-pattern tiles (tile_pp_s<.., PAIRF>): entry of the (k+1)-mer
            // code c = field of the k-mer at p + field of the one at p + 1
            const size_t n2 = n1 * 4;
            off_pfld = lut.size();                      // (a multiple of 4 dwords)
            lut.resize(off_pfld + n2, 0u);
            for (size_t cc = 0; cc < n2; ++cc) lut[off_pfld + cc] = lut[off_fld + (cc & (n1 - 1))] + lut[off_fld + ((cc >> 2) & (n1 - 1))];
        }
        if (k == 5 && P <= 16) {
            // 16-bit pair table of the _s*q kernels (ScanArgs::pair16): entry of the (k+1)-mer code c = the masks of the k-mers at p
            // and p + 1 ORed (counts are popcounts: the planner takes it for tables without self-overlap only)
            const size_t n2 = n1 * 4;
            off_p16 = lut.size();                       // (a multiple of 4 dwords)
            lut.resize(off_p16 + n2 / 2, 0u);
            for (size_t cc = 0; cc < n2; ++cc)
                ((uint16_t*)&lut[off_p16])[cc] = (uint16_t)(lut[cc & (n1 - 1)] | lut[(cc >> 2) & (n1 - 1)]);
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));          // no launch may still be reading the table that gets recycled
    tps_ctx::Table* slot = nullptr;
    if (c->tables.size() < 6) { c->tables.emplace_back(); slot = &c->tables.back(); }
    else { slot = &c->tables[c->table_rr++ % c->tables.size()]; }
    if ((rc = slot->dev.ensure(lut.size() * 4))) return rc;
    HIP_TRY(hipMemcpyAsync(slot->dev.p, lut.data(), lut.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    slot->P = P; slot->k = k; slot->key = key; slot->pat = pi;
    slot->off_e32 = off_e32; slot->off_fld = off_fld; slot->off_m16 = off_m16; slot->off_f16 = off_f16; slot->off_p16 = off_p16; slot->off_pfld = off_pfld;
    c->lut_cur = slot;
    c->pat = pi;
    c->have_pat = true;
    for (auto& sl : c->slots) sl.planned = false;     // kernel choice and LDS plan depend on the table (periods, duplicates, k)
    return TPS_OK;
}

int tps_batch_upload(tps_ctx* c, int32_t slot, const uint8_t* bases, const int64_t* offsets, int64_t n) {
    int rc;
    if ((rc = bind(c))) return rc;
    Slot* sl = get_slot(c, slot);
    if (!sl) return TPS_E_ARG;
    return do_upload(c, *sl, bases, offsets, n);
}

int tps_batch_upload_packed(tps_ctx* c, int32_t slot, const uint32_t* seq2, const uint16_t* inv, const tps_read_desc* desc,
                            int64_t n, int64_t n_words) {
    int rc;
    if ((rc = bind(c))) return rc;
    Slot* sl = get_slot(c, slot);
    if (!sl) return TPS_E_ARG;
    return do_upload_packed(c, *sl, seq2, inv, desc, n, n_words);
}

int tps_batch_share(tps_ctx* c, int32_t slot, tps_ctx* src, int32_t src_slot) {
    int rc;
    if ((rc = bind(c))) return rc;
    if (!src || src == c) return fail(TPS_E_ARG, "the batch must come from another context");
    if (src->device != c->device) return fail(TPS_E_ARG, "both contexts must be on the same device (%d vs %d)", c->device, src->device);
    Slot* sl = get_slot(c, slot);
    Slot* from = get_slot(src, src_slot);
    if (!sl || !from) return TPS_E_ARG;
    if (from->n < 0) return fail(TPS_E_STATE, "no batch uploaded in the source slot");
    // whatever this context still has in flight on the slot's old buffers must be over before they are let go
    HIP_TRY(hipStreamSynchronize(c->stream));
    sl->seq2.borrow(from->seq2);
    sl->inv.borrow(from->inv);
    sl->desc.borrow(from->desc);
    sl->h_offsets = from->h_offsets;
    sl->inv_valid = from->inv_valid;
    sl->any_invalid = from->any_invalid;
    reset_slot(*sl, from->n, from->n_words);
    // the source's upload may still be running on ITS stream: this context's stream waits for it
    if (!c->share_ev) HIP_TRY(hipEventCreateWithFlags(&c->share_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->share_ev, src->stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->share_ev, 0));
    return TPS_OK;
}

int tps_host_alloc(tps_ctx* c, int64_t bytes, void** out) {
    int rc;
    if ((rc = bind(c))) return rc;
    if (!out || bytes < 0) return fail(TPS_E_ARG, "bad arguments");
    *out = nullptr;
    void* p = nullptr;
    HIP_TRY(hipHostMalloc(&p, (size_t)std::max<int64_t>(bytes, 16), hipHostMallocPortable));
    c->pinned.insert(p);
    { std::lock_guard<std::mutex> lk(g_pinned_mu); g_pinned[(uintptr_t)p] = (size_t)std::max<int64_t>(bytes, 16); }
    *out = p;
    return TPS_OK;
}

int tps_host_free(tps_ctx* c, void* p) {
    int rc;
    if ((rc = bind(c))) return rc;
    if (!p) return TPS_OK;
    if (!c->pinned.erase(p)) return fail(TPS_E_ARG, "pointer was not allocated by tps_host_alloc of this context");
    { std::lock_guard<std::mutex> lk(g_pinned_mu); g_pinned.erase((uintptr_t)p); }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipHostFree(p));
    return TPS_OK;
}

int tps_batch_download_packed(tps_ctx* c, int32_t slot, uint32_t* seq2, uint16_t* inv, tps_read_desc* desc, int64_t n, int64_t n_words,
                              int64_t* n_words_out) {
    int rc;
    if ((rc = bind(c))) return rc;
    Slot* sl = get_slot(c, slot);
    if (!sl) return TPS_E_ARG;
    if (sl->n < 0) return fail(TPS_E_STATE, "no batch uploaded in this slot");
    if (n_words_out) *n_words_out = sl->n_words;
    if ((seq2 || inv) && n_words != sl->n_words) return fail(TPS_E_ARG, "buffers must hold %lld words", (long long)sl->n_words);
    if (desc && n != sl->n) return fail(TPS_E_ARG, "desc must hold %lld reads", (long long)sl->n);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (seq2 && n_words) HIP_TRY(hipMemcpy(seq2, sl->seq2.p, (size_t)n_words * 4, hipMemcpyDeviceToHost));
    if (inv && n_words) {
        if (!sl->inv_valid) memset(inv, 0, (size_t)n_words * 2);
        else HIP_TRY(hipMemcpy(inv, sl->inv.p, (size_t)n_words * 2, hipMemcpyDeviceToHost));
    }
    if (desc && n) HIP_TRY(hipMemcpy(desc, sl->desc.p, (size_t)n * sizeof(tps_read_desc), hipMemcpyDeviceToHost));
    return TPS_OK;
}

int tps_batch_kmer_followers(tps_ctx* c, int32_t slot, int32_t n_fwd, int32_t follow, int32_t lo, int32_t hi, int32_t min_len,
                             uint32_t* picks, int64_t picks_words, int64_t* hist, int64_t hist_len) {
    int rc;
    if ((rc = bind(c))) return rc;
    Slot* sl = get_slot(c, slot);
    if (!sl) return TPS_E_ARG;
    if (!c->have_pat) return fail(TPS_E_PATTERN, "tps_set_patterns has not been called");
    if (sl->n < 0) return fail(TPS_E_STATE, "no batch uploaded in this slot");
    if (n_fwd < 1 || n_fwd > 15 || 2 * n_fwd > c->pat.P) return fail(TPS_E_ARG, "n_fwd must be 1..15 and the table must hold the complements behind the k-mers");
    if (follow < 0 || follow > 8) return fail(TPS_E_CAPACITY, "follow must be 0..8 bases");
    if (lo < 0 || hi <= lo || hi - lo > tps::FOLLOW_MAX_SPAN) return fail(TPS_E_CAPACITY, "the scanned range [lo, hi) must hold 1..%d bases", tps::FOLLOW_MAX_SPAN);
    const int pw = (hi - lo + 31) / 32;
    const int64_t n = sl->n;
    const int64_t want = n * 2 * n_fwd * pw;
    if (!picks || picks_words != want) return fail(TPS_E_ARG, "picks must hold %lld words", (long long)want);
    const int nbins = (1 << (2 * follow)) + 1;
    if (hist && hist_len != 2ll * n_fwd * nbins) return fail(TPS_E_ARG, "hist must hold %lld counters", 2ll * n_fwd * nbins);
    if (n == 0) { if (hist) memset(hist, 0, (size_t)hist_len * 8); return TPS_OK; }
    // scratch: the slot's raw / sums buffers are not touched; picks and counters get buffers of their own in the context
    if ((rc = c->follow_picks.ensure((size_t)want * 4))) return rc;
    if ((rc = c->follow_hist.ensure((size_t)2 * n_fwd * nbins * 8))) return rc;
    HIP_TRY(hipMemsetAsync(c->follow_picks.p, 0, (size_t)want * 4, c->stream));
    HIP_TRY(hipMemsetAsync(c->follow_hist.p, 0, (size_t)2 * n_fwd * nbins * 8, c->stream));
    tps::FollowArgs a{};
    a.seq2 = (const uint32_t*)sl->seq2.p;
    a.inv = (const uint16_t*)sl->inv.p;
    a.desc = (const tps_read_desc*)sl->desc.p;
    a.lut = (const uint32_t*)c->lut_cur->dev.p;
    a.picks = (uint32_t*)c->follow_picks.p;
    a.hist = hist ? (unsigned long long*)c->follow_hist.p : nullptr;
    a.n_reads = n;
    a.pat = c->pat;
    a.n_fwd = n_fwd; a.follow = follow; a.lo = lo; a.hi = hi; a.min_len = min_len; a.pw = pw; a.nbins = nbins;
    hipLaunchKernelGGL(tps_followers_kernel, dim3((unsigned)((n + tps::WPG - 1) / tps::WPG)), dim3(tps::NT * tps::WPG), 0, c->stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(picks, c->follow_picks.p, (size_t)want * 4, hipMemcpyDeviceToHost, c->stream));
    if (hist) HIP_TRY(hipMemcpyAsync(hist, c->follow_hist.p, (size_t)hist_len * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TPS_OK;
}

int tps_batch_set_tails(tps_ctx* c, int32_t slot, const uint8_t* tails) {
    int rc;
    if ((rc = bind(c))) return rc;
    Slot* sl = get_slot(c, slot);
    if (!sl) return TPS_E_ARG;
    if (sl->n < 0) return fail(TPS_E_STATE, "no batch uploaded in this slot");
    if (!tails && sl->n > 0) return fail(TPS_E_ARG, "null tails");
    if ((rc = sl->tails.ensure((size_t)std::max<int64_t>(sl->n, 1)))) return rc;
    if (sl->n) HIP_TRY(hipMemcpyAsync(sl->tails.p, tails, (size_t)sl->n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    sl->has_tails = true;
    return TPS_OK;
}

int tps_batch_scan(tps_ctx* c, int32_t slot, const tps_params* prm) {
    int rc;
    if ((rc = bind(c))) return rc;
    Slot* sl = get_slot(c, slot);
    if (!sl) return TPS_E_ARG;
    if (!prm) return fail(TPS_E_ARG, "null params");
    return do_scan(c, *sl, *prm);
}

int tps_sync(tps_ctx* c) {
    int rc;
    if ((rc = bind(c))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TPS_OK;
}

static int need_scanned(tps_ctx* c, int slot, Slot** out) {
    int rc;
    if ((rc = bind(c))) return rc;
    Slot* sl = (slot == INTERNAL_SLOT) ? &c->slots[INTERNAL_SLOT] : get_slot(c, slot);
    if (!sl) return TPS_E_ARG;
    if (!sl->scanned) return fail(TPS_E_STATE, "slot has not been scanned");
    HIP_TRY(hipStreamSynchronize(c->stream));
    *out = sl;
    return TPS_OK;
}

int tps_batch_results(tps_ctx* c, int32_t slot, tps_read_result* out, int64_t n) {
    Slot* sl;
    int rc;
    if ((rc = need_scanned(c, slot, &sl))) return rc;
    if (n != sl->n || (n > 0 && !out)) return fail(TPS_E_ARG, "result buffer must hold %lld reads", (long long)sl->n);
    if (n) memcpy(out, sl->h_results, (size_t)n * sizeof(tps_read_result));
    return TPS_OK;
}

int tps_batch_window_offsets(tps_ctx* c, int32_t slot, int64_t* win_off, int64_t n1) {
    Slot* sl;
    int rc;
    if ((rc = need_scanned(c, slot, &sl))) return rc;
    if (n1 != sl->n + 1 || !win_off) return fail(TPS_E_ARG, "win_off must hold n+1 entries");
    memcpy(win_off, sl->h_win_off.data(), (size_t)n1 * 8);
    return TPS_OK;
}

int tps_batch_window_sums(tps_ctx* c, int32_t slot, int32_t* sums, int64_t nw) {
    Slot* sl;
    int rc;
    if ((rc = need_scanned(c, slot, &sl))) return rc;
    if (!(sl->last_flags & TPS_F_STORE_SUMS)) return fail(TPS_E_STATE, "last scan did not store window sums");
    if (nw != sl->h_win_off[(size_t)sl->n] || (nw > 0 && !sums)) return fail(TPS_E_ARG, "sums must hold %lld windows", (long long)sl->h_win_off[(size_t)sl->n]);
    if (nw && sl->args.variant) {
        // fused kernels: 16-bit sums in the padded device layout -> the caller's contiguous int32 array
        const int64_t n16 = sl->h_win_off16[(size_t)sl->n];
        sl->h_sums16.resize((size_t)n16);
        HIP_TRY(hipMemcpy(sl->h_sums16.data(), sl->sums.p, (size_t)n16 * 2, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < sl->n; ++i) {
            const uint16_t* src = sl->h_sums16.data() + sl->h_win_off16[(size_t)i];
            int32_t* dst = sums + sl->h_win_off[(size_t)i];
            const int64_t cnt = sl->h_win_off[(size_t)i + 1] - sl->h_win_off[(size_t)i];
            for (int64_t w = 0; w < cnt; ++w) dst[w] = (int32_t)src[w];
        }
    } else if (nw) {
        HIP_TRY(hipMemcpy(sums, sl->sums.p, (size_t)nw * 4, hipMemcpyDeviceToHost));
    }
    return TPS_OK;
}

int tps_batch_read_sums(tps_ctx* c, int32_t slot, int64_t read, int32_t* sums, int64_t nw) {
    Slot* sl;
    int rc;
    if ((rc = need_scanned(c, slot, &sl))) return rc;
    if (!(sl->last_flags & TPS_F_WINDOWS)) return fail(TPS_E_STATE, "last scan did not run the window step");
    if (read < 0 || read >= sl->n) return fail(TPS_E_ARG, "read %lld out of range", (long long)read);
    const int64_t lo = sl->h_win_off[(size_t)read], cnt = sl->h_win_off[(size_t)read + 1] - lo;
    if (nw != cnt || (nw > 0 && !sums)) return fail(TPS_E_ARG, "sums must hold %lld windows", (long long)cnt);
    if (!nw) return TPS_OK;
    if (sl->args.variant) {
        sl->h_sums16.resize((size_t)nw);
        HIP_TRY(hipMemcpy(sl->h_sums16.data(), (const uint16_t*)sl->sums.p + sl->h_win_off16[(size_t)read], (size_t)nw * 2, hipMemcpyDeviceToHost));
        for (int64_t w = 0; w < nw; ++w) sums[w] = (int32_t)sl->h_sums16[(size_t)w];
    } else {
        HIP_TRY(hipMemcpy(sums, (const int32_t*)sl->sums.p + lo, (size_t)nw * 4, hipMemcpyDeviceToHost));
    }
    return TPS_OK;
}

int tps_batch_window_raw(tps_ctx* c, int32_t slot, uint8_t* raw, int64_t nwp) {
    Slot* sl;
    int rc;
    if ((rc = need_scanned(c, slot, &sl))) return rc;
    if (!(sl->last_flags & TPS_F_STORE_RAW)) return fail(TPS_E_STATE, "last scan did not store raw counts");
    int64_t want = sl->h_win_off[(size_t)sl->n] * sl->plan_p;
    if (nwp != want || (nwp > 0 && !raw)) return fail(TPS_E_ARG, "raw must hold %lld bytes", (long long)want);
    if (nwp) HIP_TRY(hipMemcpy(raw, sl->raw.p, (size_t)nwp, hipMemcpyDeviceToHost));
    return TPS_OK;
}

// Raw rows of selected reads -> a file, without a stop in the caller's memory (include/topsicle_hip.h).  The selected reads' rows
// are runs of the slot's raw buffer; runs closer than RAW_GAP are fetched as one stretch (what lies between is copied and skipped:
// cheaper than a copy call per read), stretches are cut into pieces of RAW_STAGE bytes, piece j + 1 is copied into one pinned
// buffer while piece j is written from the other (pwritev of the wanted parts only) and checksummed.
constexpr size_t RAW_STAGE = (size_t)32 << 20;
constexpr int64_t RAW_GAP = 256 << 10;

int tps_batch_raw_to_fd(tps_ctx* c, int32_t slot, const int64_t* reads, int64_t n_sel, int fd, int64_t file_off,
                        tps_crc32_fn crc_fn, uint32_t* crc_out, int64_t* bytes_out) {
    Slot* sl;
    int rc;
    if ((rc = need_scanned(c, slot, &sl))) return rc;
    if (!(sl->last_flags & TPS_F_STORE_RAW)) return fail(TPS_E_STATE, "last scan did not store raw counts");
    if (n_sel < 0 || (n_sel > 0 && !reads) || fd < 0 || file_off < 0) return fail(TPS_E_ARG, "bad arguments");
    if (crc_out) *crc_out = 0;
    if (bytes_out) *bytes_out = 0;
    const int64_t P = sl->plan_p;
    struct Run { int64_t lo, hi; };                    // bytes of the raw buffer
    std::vector<Run> runs;
    int64_t prev = -1;
    for (int64_t j = 0; j < n_sel; ++j) {
        const int64_t i = reads[j];
        if (i <= prev || i >= sl->n) return fail(TPS_E_ARG, "reads[] must be ascending indices of the batch (entry %lld)", (long long)j);
        prev = i;
        const int64_t lo = sl->h_win_off[(size_t)i] * P, hi = sl->h_win_off[(size_t)i + 1] * P;
        if (hi == lo) continue;
        if (!runs.empty() && runs.back().hi == lo) runs.back().hi = hi;
        else runs.push_back({lo, hi});
    }
    if (runs.empty()) return TPS_OK;
    for (int i = 0; i < 2; ++i) {
        if (!c->raw_stage[i]) HIP_TRY(hipHostMalloc(&c->raw_stage[i], RAW_STAGE, hipHostMallocPortable));
        if (!c->raw_ev[i]) HIP_TRY(hipEventCreateWithFlags(&c->raw_ev[i], hipEventDisableTiming));
    }
    // pieces: [first run, last run) + the device span [lo, hi) that covers them (hi - lo <= RAW_STAGE); a run longer than what is
    // left of a piece is split
    struct Piece { size_t r0, r1; int64_t lo, hi; };
    std::vector<Piece> pieces;
    {
        std::vector<Run> cut;                          // runs split at piece boundaries, in order
        size_t i = 0;
        int64_t pos = runs[0].lo;
        while (i < runs.size()) {
            Piece pc{cut.size(), cut.size(), pos, pos};
            while (i < runs.size()) {
                const int64_t start = std::max(pos, runs[i].lo);
                if (pc.r1 > pc.r0 && (start - pc.hi > RAW_GAP || start >= pc.lo + (int64_t)RAW_STAGE)) break;
                if (pc.r1 == pc.r0) pc.lo = start;
                const int64_t end = std::min(runs[i].hi, pc.lo + (int64_t)RAW_STAGE);
                if (end <= start) break;
                cut.push_back({start, end});
                pc.r1 = cut.size();
                pc.hi = end;
                pos = end;
                if (end == runs[i].hi) { ++i; if (i < runs.size()) pos = runs[i].lo; }
                else break;                            // the piece is full
            }
            pieces.push_back(pc);
        }
        runs.swap(cut);
    }
    auto enqueue = [&](size_t j) -> int {
        const Piece& pc = pieces[j];
        HIP_TRY(hipMemcpyAsync(c->raw_stage[j & 1], (const uint8_t*)sl->raw.p + pc.lo, (size_t)(pc.hi - pc.lo), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(c->raw_ev[j & 1], c->stream));
        return TPS_OK;
    };
    if ((rc = enqueue(0))) return rc;
    uint32_t crc = 0;
    int64_t written = 0;
    std::vector<struct iovec> iov;
    for (size_t j = 0; j < pieces.size(); ++j) {
        if (j + 1 < pieces.size() && (rc = enqueue(j + 1))) { (void)hipStreamSynchronize(c->stream); return rc; }
        HIP_TRY(hipEventSynchronize(c->raw_ev[j & 1]));
        const Piece& pc = pieces[j];
        const uint8_t* base = (const uint8_t*)c->raw_stage[j & 1];
        iov.clear();
        for (size_t r = pc.r0; r < pc.r1; ++r) {
            const uint8_t* p = base + (runs[r].lo - pc.lo);
            const size_t len = (size_t)(runs[r].hi - runs[r].lo);
            if (crc_fn) crc = crc_fn(crc, p, (int64_t)len);
            iov.push_back({(void*)p, len});
        }
        size_t first = 0;
        while (first < iov.size()) {
            const int cnt = (int)std::min<size_t>(iov.size() - first, 1024);
            const ssize_t w = pwritev(fd, iov.data() + first, cnt, (off_t)(file_off + written));
            if (w < 0) {
                if (errno == EINTR) continue;
                const int e = errno;
                (void)hipStreamSynchronize(c->stream);
                return fail(TPS_E_ARG, "pwritev: %s", strerror(e));
            }
            written += w;
            size_t left = (size_t)w;
            while (first < iov.size() && left >= iov[first].iov_len) { left -= iov[first].iov_len; ++first; }
            if (left) { iov[first].iov_base = (char*)iov[first].iov_base + left; iov[first].iov_len -= left; }
        }
    }
    if (crc_out) *crc_out = crc;
    if (bytes_out) *bytes_out = written;
    return TPS_OK;
}

int tps_batch_trc_counts(tps_ctx* c, int32_t slot, int32_t* c_start, int32_t* c_end, int64_t n) {
    Slot* sl;
    int rc;
    if ((rc = need_scanned(c, slot, &sl))) return rc;
    if (!(sl->last_flags & TPS_F_STEP1)) return fail(TPS_E_STATE, "last scan did not run step 1");
    if (n != sl->n || (n > 0 && (!c_start || !c_end))) return fail(TPS_E_ARG, "count buffers must hold n*P entries");
    size_t bytes = (size_t)n * (size_t)sl->plan_p * 4;
    if (bytes) {
        HIP_TRY(hipMemcpy(c_start, sl->c_start.p, bytes, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(c_end, sl->c_end.p, bytes, hipMemcpyDeviceToHost));
    }
    return TPS_OK;
}

int64_t tps_window_count(int64_t L, int32_t W, int32_t s, int32_t t, int32_t M) { return window_count(L, W, s, t, M); }

int tps_trc_counts(tps_ctx* c, const uint8_t* bases, const int64_t* offsets, int64_t n, int32_t no_bp,
                   int32_t* c_start, int32_t* c_end) {
    int rc;
    if ((rc = bind(c))) return rc;
    Slot& sl = c->slots[INTERNAL_SLOT];
    if ((rc = do_upload(c, sl, bases, offsets, n))) return rc;
    tps_params p{};
    p.no_bp = no_bp; p.min_len = 0; p.min_count = 0; p.window = 100; p.slide = 6; p.trimfirst = 0; p.maxlen = 0;
    p.jump = 5; p.min_size = 2; p.flags = TPS_F_STEP1;
    if ((rc = do_scan(c, sl, p))) return rc;
    return tps_batch_trc_counts(c, INTERNAL_SLOT, c_start, c_end, n);
}

int tps_window_counts(tps_ctx* c, const uint8_t* bases, const int64_t* offsets, const uint8_t* tails, int64_t n,
                      int32_t W, int32_t s, int32_t t, int32_t M, const int64_t* win_off, int32_t* sums, uint8_t* raw) {
    int rc;
    if ((rc = bind(c))) return rc;
    if (!win_off || (n > 0 && !tails)) return fail(TPS_E_ARG, "null win_off / tails");
    Slot& sl = c->slots[INTERNAL_SLOT];
    if ((rc = do_upload(c, sl, bases, offsets, n))) return rc;
    if ((rc = sl.tails.ensure((size_t)std::max<int64_t>(n, 1)))) return rc;
    if (n) HIP_TRY(hipMemcpy(sl.tails.p, tails, (size_t)n, hipMemcpyHostToDevice));
    sl.has_tails = true;
    tps_params p{};
    p.no_bp = 0; p.window = W; p.slide = s; p.trimfirst = t; p.maxlen = M; p.jump = 5; p.min_size = 2;
    p.flags = TPS_F_WINDOWS | TPS_F_TAILS_IN | TPS_F_STORE_SUMS | (raw ? TPS_F_STORE_RAW : 0u);
    if ((rc = do_scan(c, sl, p))) return rc;
    for (int64_t i = 0; i <= n; ++i)
        if (win_off[i] != sl.h_win_off[(size_t)i])
            return fail(TPS_E_ARG, "win_off[%lld]=%lld does not match the window layout (%lld)", (long long)i,
                        (long long)win_off[i], (long long)sl.h_win_off[(size_t)i]);
    int64_t nw = sl.h_win_off[(size_t)n];
    if ((rc = tps_batch_window_sums(c, INTERNAL_SLOT, sums, nw))) return rc;
    if (raw && (rc = tps_batch_window_raw(c, INTERNAL_SLOT, raw, nw * c->pat.P))) return rc;
    return TPS_OK;
}

int tps_binseg_l2_ties(tps_ctx* c, const int32_t* sums, const int64_t* win_off, int64_t n, int32_t n_patterns,
                       int32_t jump, int32_t min_size, int32_t* bkp, double* gain, uint8_t* tie);
int tps_binseg_l2(tps_ctx* c, const int32_t* sums, const int64_t* win_off, int64_t n, int32_t n_patterns,
                  int32_t jump, int32_t min_size, int32_t* bkp, double* gain) {
    return tps_binseg_l2_ties(c, sums, win_off, n, n_patterns, jump, min_size, bkp, gain, nullptr);
}

int tps_binseg_l2_ties(tps_ctx* c, const int32_t* sums, const int64_t* win_off, int64_t n, int32_t n_patterns,
                       int32_t jump, int32_t min_size, int32_t* bkp, double* gain, uint8_t* tie) {
    int rc;
    if ((rc = bind(c))) return rc;
    if (n < 0 || !win_off || !bkp || jump < 1 || min_size < 1 || n_patterns < 1) return fail(TPS_E_ARG, "bad arguments");
    if (n == 0) return TPS_OK;
    const int64_t nw = win_off[n];
    if (nw > 0 && !sums) return fail(TPS_E_ARG, "null sums");
    for (int64_t i = 0; i < n; ++i) {
        // the kernel adds a read's sums up in 32 bits and compares d^2 * den in 128 bits (d <= len * total, den <= len^2 / 4)
        const int64_t lo = win_off[i], len = win_off[i + 1] - lo;
        if (len < 0 || lo < 0 || lo + len > nw) return fail(TPS_E_ARG, "win_off not monotone at %lld", (long long)i);
        if (len > 500000) return fail(TPS_E_CAPACITY, "read %lld has %lld windows (max 500000)", (long long)i, (long long)len);
        uint64_t tot = 0;
        for (int64_t w = 0; w < len; ++w) {
            if (sums[lo + w] < 0) return fail(TPS_E_ARG, "negative window sum at read %lld", (long long)i);
            tot += (uint64_t)sums[lo + w];
        }
        if (tot >> 32 || (double)len * (double)len * (double)tot >= 18446744073709551616.0)
            return fail(TPS_E_CAPACITY, "window sums of read %lld exceed the change-point arithmetic (total %llu over %lld windows)",
                        (long long)i, (unsigned long long)tot, (long long)len);
    }
    Slot& sl = c->slots[INTERNAL_SLOT];
    if ((rc = sl.sums.ensure((size_t)std::max<int64_t>(nw, 1) * 4))) return rc;
    if ((rc = sl.win_off.ensure((size_t)(n + 1) * 8))) return rc;
    if ((rc = sl.results.ensure((size_t)n * 17))) return rc;          // gain (8n), bkp (4n), tie (n): gain first for alignment
    sl.planned = false;                                                // the slot's plan buffers were overwritten
    sl.scanned = false;
    if (nw) HIP_TRY(hipMemcpyAsync(sl.sums.p, sums, (size_t)nw * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(sl.win_off.p, win_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    tps::BinsegArgs a{};
    a.sums = (const int32_t*)sl.sums.p;
    a.win_off = (const int64_t*)sl.win_off.p;
    a.gain = (double*)sl.results.p;
    a.bkp = (int32_t*)((char*)sl.results.p + (size_t)n * 8);
    a.tie = (uint8_t*)sl.results.p + (size_t)n * 12;
    a.n_reads = n;
    a.n_patterns = n_patterns;
    a.jump = jump;
    a.min_size = min_size;
    hipLaunchKernelGGL(tps_binseg_kernel, dim3((unsigned)((n + tps::WPG - 1) / tps::WPG)), dim3(tps::NT * tps::WPG), 0, c->stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(bkp, a.bkp, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (gain) HIP_TRY(hipMemcpyAsync(gain, a.gain, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    if (tie) HIP_TRY(hipMemcpyAsync(tie, a.tie, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TPS_OK;
}

int tps_kernel_time_ms(tps_ctx* c, int32_t* n_launches, double* total_ms, double* mean_ms) {
    int rc;
    if ((rc = bind(c))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    double tot = 0.0;
    for (size_t i = c->ev_base; i < c->ev_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b));
        tot += ms;
    }
    const size_t cnt = c->ev_used - c->ev_base;
    if (n_launches) *n_launches = (int32_t)cnt;
    if (total_ms) *total_ms = tot;
    if (mean_ms) *mean_ms = cnt ? tot / (double)cnt : 0.0;
    return TPS_OK;
}

int tps_ctx_debug_option(tps_ctx* c, const char* key, int64_t value) {
    if (!c || !key) return fail(TPS_E_ARG, "null argument");
    const std::string k(key);
    if (k == "event_stride") c->event_stride = (int)std::max<int64_t>(1, value);
    else if (k == "no_events") c->no_events = value != 0;
    else if (k == "force_generic") c->knobs.force_generic = value != 0;
    else if (k == "spans_per_tile") c->knobs.spans_per_tile = (int)std::max<int64_t>(0, value);
    else if (k == "force_pair") c->knobs.force_pair = value != 0;
    else if (k == "so_order") c->knobs.so_order = (int)value;
    else if (k == "wpg") c->knobs.wpg = (int)value;
    else if (k == "stamps") c->want_stamps = value != 0;
    else if (k == "file_order") c->file_order = value != 0;
    else if (k == "no_stride") c->no_stride = value != 0;
    else if (k == "no_inline_rows") c->no_inline_rows = value != 0;
    else return fail(TPS_E_ARG, "unknown debug option '%s' (event_stride, no_events, force_generic, spans_per_tile, force_pair, so_order, wpg, stamps, file_order, no_stride, no_inline_rows)", key);
    for (auto& sl : c->slots) sl.planned = false;
    return TPS_OK;
}

/* per-read phase clock stamps of the last scan (only a -DTPS_STAMPS build of the kernels writes them; option "stamps") */
int tps_debug_stamps_get(tps_ctx* c, int32_t slot, uint64_t* out, int64_t n) {
    Slot* sl;
    int rc;
    if ((rc = need_scanned(c, slot, &sl))) return rc;
    if (!sl->stamps.p || n != sl->n) return fail(TPS_E_STATE, "no stamps recorded");
    HIP_TRY(hipMemcpy(out, sl->stamps.p, (size_t)n * 16 * 8, hipMemcpyDeviceToHost));
    return TPS_OK;
}

int tps_kernel_time_reset(tps_ctx* c) {
    int rc;
    if ((rc = bind(c))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->ev_base = c->ev_used;
    c->launch_seq = 0;                             // the next launch is a timed one
    return TPS_OK;
}

}  // extern "C"
