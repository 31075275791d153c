// tps_io.cpp -- libtopsicle_io.so: native streaming FASTA / FASTQ (.gz) reader for the host side.
//
// Replaces the reference's Biopython parse (Bio.SeqIO.parse over gzip.open, allsteps.py:127-149,
// main.py:83-86) on the way INTO the GPU path: records are decoded straight into the caller's batch
// buffers in the layout tps_batch_upload() takes (concatenated ASCII bases + n+1 offsets), so no
// per-read Python object is created for the ~99 % of reads that are never written back out.
// Record ids follow Biopython: the first whitespace-delimited token of the header line.
//
//
// Two decoders behind one interface:
//   * plain (uncompressed) 4-line FASTQ: the file is mmap'ed; newline positions of a window are indexed
//     by a team of threads (memchr), records are validated ('@', '+', equal sequence / quality length)
//     and their sequence, quality and header bytes are copied into the batch buffers by the same team.
//     Anything unexpected (blank or padded lines inside a record, length mismatch; in ASCII mode also wrapped lines) hands the
//     rest of the file to
//   * the streaming decoder on zlib (gz or plain, FASTA or FASTQ, wrapped lines), single-threaded.
//
// Packed mode (tps_reader_next_packed): plain FASTQ records are 2-bit packed straight from the mmap'ed file into the
// caller's (pinned) upload buffers in the layout of tps_pack.h -- no ASCII copy of the bases, the qualities are never
// touched; the caller gets each record's spans in the file so that the few reads that pass the TRC filter can be written
// back out from the mapping.  tps_pack_reads packs an ASCII batch (the .gz / BGZF / FASTA paths) with the same team.
//
// Plain C ABI (ctypes-loadable).  Build: g++ -O2 -shared -fPIC tps_io.cpp -lz -lpthread
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sys/uio.h>
#include <zlib.h>

#include "tps_pack.h"
#include "tps_gzpar.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <pthread.h>
#include <ctime>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;

struct Reader {
    gzFile gz = nullptr;
    std::vector<char> buf;
    size_t pos = 0, len = 0;
    bool eof = false;
    int format = 0;                     // 1 = fasta, 2 = fastq
    // one parsed record kept back when it did not fit into the caller's batch
    bool have_pending = false;
    std::string p_head, p_seq, p_qual;
    std::string next_head;              // FASTA: header line already consumed while reading the previous record
    bool have_next_head = false;
    // byte ranges (tps_reader_open_range): offset of buf[0] in the (plain) file, where the last parsed record began, and where the
    // FASTA header consumed ahead began
    uint64_t base = 0, rec_start = 0, next_head_start = 0;

    bool fill() {
        if (eof) return false;
        if (pos < len) memmove(buf.data(), buf.data() + pos, len - pos);
        len -= pos;
        base += pos;
        pos = 0;
        int got = gzread(gz, buf.data() + len, (unsigned)(buf.size() - len));
        if (got < 0) { g_err = "gzread failed"; eof = true; return false; }
        if (got == 0) { eof = true; return len > 0; }
        len += (size_t)got;
        return true;
    }
    // next line without its line end; false at end of input.  Line ends as Python's text mode -- what the reference reads its
    // input through (allsteps.py:127-149) -- sees them: "\n", "\r\n", and a lone "\r" (cr_lines: the file's first line ended in
    // one -- classic Mac OS text; without that a stray CR inside a line stays what it was before round 3: data)
    bool cr_lines = false;
    bool getline(std::string& out) {
        out.clear();
        for (;;) {
            char* nl = (char*)memchr(buf.data() + pos, '\n', len - pos);
            if (cr_lines) {
                char* cr = (char*)memchr(buf.data() + pos, '\r', (nl ? (size_t)(nl - buf.data()) : len) - pos);
                if (cr) {
                    if (cr + 1 == buf.data() + len && !eof) {         // is a "\n" behind it?  it may be in the next buffer
                        out.append(buf.data() + pos, (size_t)(cr - (buf.data() + pos)));
                        pos = (size_t)(cr - buf.data());
                        if (fill() && len - pos > 1) continue;        // (the CR is at buf[pos] again, with what follows it)
                        pos = len;                                    // end of input behind the CR
                        break;
                    }
                    out.append(buf.data() + pos, (size_t)(cr - (buf.data() + pos)));
                    pos = (size_t)(cr - buf.data()) + ((cr + 1 < buf.data() + len && cr[1] == '\n') ? 2 : 1);
                    break;
                }
            }
            if (nl) {
                out.append(buf.data() + pos, (size_t)(nl - (buf.data() + pos)));
                pos = (size_t)(nl - buf.data()) + 1;
                break;
            }
            out.append(buf.data() + pos, len - pos);
            pos = len;
            if (!fill()) {
                if (out.empty()) return false;
                break;
            }
        }
        while (!out.empty() && (out.back() == '\r' || out.back() == '\n')) out.pop_back();
        return true;
    }
    static void strip(std::string& s) {
        size_t a = 0, b = s.size();
        while (a < b && (s[a] == ' ' || s[a] == '\t')) ++a;
        while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t')) --b;
        if (a || b != s.size()) s = s.substr(a, b - a);
    }
    // parse one record into p_head / p_seq / p_qual; false at end of input
    bool parse_one() {
        std::string line;
        if (format == 2) {
            do {
                rec_start = base + pos;
                if (!getline(line)) return false;
            } while (line.empty());
            if (line[0] != '@') { g_err = "FASTQ record does not start with '@'"; return false; }
            p_head.assign(line, 1, std::string::npos);
            p_seq.clear();
            p_qual.clear();
            while (getline(line)) {
                if (!line.empty() && line[0] == '+') break;
                strip(line);
                p_seq += line;
            }
            while (p_qual.size() < p_seq.size()) {
                if (!getline(line)) break;
                p_qual += line;
            }
            if (p_qual.size() != p_seq.size()) {       // mis-framed from here on: an error, like the Python parser and Biopython
                g_err = "FASTQ record '" + p_head.substr(0, p_head.find_first_of(" \t")) + "': " + std::to_string(p_qual.size()) +
                        " quality characters for " + std::to_string(p_seq.size()) + " bases";
                return false;
            }
            return true;
        }
        // FASTA
        if (!have_next_head) {
            for (;;) {
                next_head_start = base + pos;
                if (!getline(line)) return false;
                if (!line.empty() && line[0] == '>') break;
            }
            next_head.assign(line, 1, std::string::npos);
        }
        p_head = next_head;
        rec_start = next_head_start;
        have_next_head = false;
        p_seq.clear();
        p_qual.clear();
        for (;;) {
            const uint64_t line_start = base + pos;
            if (!getline(line)) break;
            if (!line.empty() && line[0] == '>') {
                next_head.assign(line, 1, std::string::npos);
                next_head_start = line_start;
                have_next_head = true;
                break;
            }
            strip(line);
            p_seq += line;
        }
        return true;
    }
};


// ---------------------------------------------------------------- mmap + thread-team decoder
// Options of the reader (tps_io_set_option: tests and diagnostics; include/topsicle_io.h).  The library reads NO environment
// variables (round 5): topsicle_amd.seqio applies $TOPSICLE_IO_DEBUG = "key=value,..." when it loads the library.
struct IoOptions {
    std::atomic<int> threads{0};               // 0 = by the host's CPUs and the cgroup quota
    std::atomic<int> timing{0};                // phase times of the fast reader on stderr
    std::atomic<long long> bgzf_group{0};      // bytes of inflated text per refill of a compressed input's window (0 = 128 MiB)
    std::atomic<long long> pack_min_span{-1};  // text below this many bytes is decoded by one thread (-1 = 4 MiB)
    std::atomic<int> no_pargz{0};              // ordinary gzip through zlib's one stream instead of the thread team's inflater
    std::atomic<long long> pargz_min{-1};      // smallest .gz file the team inflates (-1 = 1 MiB)
};
IoOptions g_opt;

int io_threads() {
    if (const int t = g_opt.threads.load()) return std::min(t, 64);
    unsigned hc = std::thread::hardware_concurrency();
    unsigned n = std::min(hc ? hc : 1u, 32u);
    // containers: the cgroup CPU quota, not the number of logical CPUs, is what the team can use
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        long long period = 0;
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long long quota = atoll(q);
            if (quota > 0) n = (unsigned)std::min<long long>(n, std::max<long long>(1, (quota + period / 2) / period));
        }
        fclose(f);
    }
    return (int)std::max(1u, n);
}
// option "timing": phase times of the fast reader on stderr (diagnostics)
inline bool io_timing() { return g_opt.timing.load() != 0; }
inline double now_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
// The thread team: f(thread index, thread count) on `nthreads` threads, the caller being thread 0.  The workers are
// created once and parked on a condition variable between calls (a team call per batch used to create and join its
// threads: on a busy or quota-limited host that cost more than the decoding).  A forked child starts with fresh pools.
struct Pool {
    std::mutex run_mu;                         // (one call at a time per pool: a leased pool has one user)
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::function<void(int, int)> job;
    int nt = 0, remaining = 0, spawned = 0;
    uint64_t gen = 0;
    void worker(int w) {
        uint64_t seen = 0;
        for (;;) {
            std::function<void(int, int)> fn;
            int n;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return gen != seen; });
                seen = gen;
                if (w >= nt) continue;
                fn = job;
                n = nt;
            }
            fn(w, n);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--remaining == 0) cv_done.notify_one();
            }
        }
    }
    void run(int nthreads, const std::function<void(int, int)>& f) {
        std::lock_guard<std::mutex> run_lk(run_mu);
        {
            std::lock_guard<std::mutex> lk(mu);
            while (spawned < nthreads - 1) {
                ++spawned;
                std::thread(&Pool::worker, this, spawned).detach();
            }
            job = f;
            nt = nthreads;
            remaining = nthreads - 1;
            ++gen;
        }
        cv_work.notify_all();
        f(0, nthreads);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return remaining == 0; });
    }
};
// Pools are LEASED per team call (round 5): readers of different shards of one file -- or of different files -- run their teams at the
// same time, each on a pool of its own; a pool goes back to the idle list when its call is over, so there are never more pools than
// calls that overlapped (round 4 had ONE pool behind a mutex: the teams of several readers took turns).
std::mutex g_pools_mu;
std::vector<Pool*>* g_idle_pools = nullptr;    // never destroyed: the pools' threads are detached and end with the process
std::once_flag g_pool_once;
struct PoolLease {
    Pool* p;
    PoolLease() {
        std::call_once(g_pool_once, [] {
            g_idle_pools = new std::vector<Pool*>();
            pthread_atfork(nullptr, nullptr, [] { g_idle_pools = new std::vector<Pool*>(); new (&g_pools_mu) std::mutex(); });   // (threads do not survive fork)
        });
        std::lock_guard<std::mutex> lk(g_pools_mu);
        if (g_idle_pools->empty()) p = new Pool();
        else { p = g_idle_pools->back(); g_idle_pools->pop_back(); }
    }
    ~PoolLease() { std::lock_guard<std::mutex> lk(g_pools_mu); g_idle_pools->push_back(p); }
};
template <typename F>
void team(int nthreads, F f) {                 // f(thread index, thread count); runs inline for one thread
    if (nthreads <= 1) { f(0, 1); return; }
    PoolLease lease;
    lease.p->run(nthreads, std::function<void(int, int)>(f));
}

// A compressed input whose text arrives group by group: read_group() appends the next stretch of text to `out` (about `want`
// bytes), inflated with the thread team.
struct Bgzf;
struct TextSource {
    bool failed = false;
    virtual ~TextSource() {}
    virtual Bgzf* as_bgzf() { return nullptr; }
    virtual bool read_group(gzpar::TextBuf& out, size_t want) = 0;
    virtual bool eof() const = 0;
};

// BGZF (bgzip): a gzip file made of independent <= 64 KiB members whose size is in the header's "BC" extra field:
// blocks inflate independently.  The compressed file is mmap'ed.
struct Bgzf : TextSource {
    int fd = -1;
    const uint8_t* data = nullptr;
    size_t size = 0, cpos = 0;
    int threads = 1;
    // byte ranges (tps_reader_open_range on a BGZF file): the blocks that start in [cpos at open, own_end) are this reader's; the text
    // behind them (the next reader's blocks) is only inflated to complete the last record.  own_text = bytes of text the own blocks
    // hold, known once the reader has come to own_end (a group of blocks never straddles it).
    size_t own_end = (size_t)-1;
    uint64_t text_total = 0, own_text = ~0ull;
    Bgzf* as_bgzf() override { return this; }
    // first offset >= from where a BGZF block starts that is followed by another block (or the end of the file): compressed data can
    // hold the magic bytes by chance, two chained headers with consistent sizes it does not
    static size_t find_block(const uint8_t* d, size_t n, size_t from) {
        for (size_t p = from; p + 18 <= n; ++p) {
            if (d[p] != 0x1f || d[p + 1] != 0x8b) continue;
            size_t bs = 0, hd = 0;
            if (!block_at(d + p, n - p, bs, hd)) continue;
            size_t bs2 = 0, hd2 = 0;
            if (p + bs == n || block_at(d + p + bs, n - p - bs, bs2, hd2)) return p;
        }
        return n;
    }
    ~Bgzf() override {
        if (data) munmap((void*)data, size);
        if (fd >= 0) close(fd);
    }
    // a BGZF member at p?  -> total block size and header length
    static bool block_at(const uint8_t* p, size_t avail, size_t& bsize, size_t& hdr) {
        if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return false;
        const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
        if (avail < 12 + xlen) return false;
        for (size_t i = 12; i + 4 <= 12 + xlen;) {
            const size_t sl = (size_t)p[i + 2] | ((size_t)p[i + 3] << 8);
            if (p[i] == 'B' && p[i + 1] == 'C' && sl == 2 && i + 6 <= 12 + xlen) {
                bsize = ((size_t)p[i + 4] | ((size_t)p[i + 5] << 8)) + 1;
                hdr = 12 + xlen;
                return bsize >= hdr + 8 && bsize <= avail;
            }
            i += 4 + sl;
        }
        return false;
    }
    bool eof() const override { return cpos >= size; }
    bool read_group(gzpar::TextBuf& out, size_t want) override {
        struct Blk { size_t in, in_len, isize, ooff; uint32_t crc; };
        std::vector<Blk> blks;
        size_t total = 0;
        while (cpos < size && total < want) {
            if (cpos >= own_end && own_text == ~0ull) {         // the next reader's first block: the own text ends here, and so does this group
                own_text = text_total + total;
                if (total) break;
            }
            size_t bsize = 0, hdr = 0;
            if (!block_at(data + cpos, size - cpos, bsize, hdr)) { g_err = "not a BGZF block (file truncated or mixed gzip members)"; failed = true; return false; }
            const uint8_t* tail = data + cpos + bsize - 8;
            const uint32_t crc = (uint32_t)tail[0] | ((uint32_t)tail[1] << 8) | ((uint32_t)tail[2] << 16) | ((uint32_t)tail[3] << 24);
            const size_t isize = (size_t)tail[4] | ((size_t)tail[5] << 8) | ((size_t)tail[6] << 16) | ((size_t)tail[7] << 24);
            blks.push_back(Blk{cpos + hdr, bsize - hdr - 8, isize, total, crc});
            total += isize;
            cpos += bsize;
        }
        const double t_bg0 = io_timing() ? now_s() : 0.0;
        const size_t base = out.size();
        out.resize(base + total);
        std::vector<int> bad((size_t)std::max(1, threads), 0);
        const int T = total < (4u << 20) ? 1 : threads;
        struct Rep { double t0; size_t n, bytes; int T; ~Rep() { if (io_timing()) fprintf(stderr, "[tps_io] bgzf group: %zu blocks, %zu MB of text, %d threads, %.2f ms\n", n, bytes >> 20, T, 1e3 * (now_s() - t0)); } } rep{t_bg0, blks.size(), total, T};
        // every block is a raw deflate stream of its own: the in-tree inflater (tps_gzpar.h: one-word table entries, 8-byte refills,
        // word copies -- 1.7 x zlib's rate per thread) into a per-thread scratch block, checked against the block's ISIZE and its
        // CRC-32 (carry-less multiplication), then copied to its place (zlib's inflate, the A/B partner of round 3, is gone)
        team(T, [&](int t, int nt) {
            const size_t a = blks.size() * (size_t)t / (size_t)nt, b = blks.size() * (size_t)(t + 1) / (size_t)nt;
            gzpar::ByteBuf scratch;
            scratch.reserve((size_t)80 << 10);
            for (size_t i = a; i < b; ++i) {
                const Blk& k = blks[i];
                if (k.isize == 0) continue;                  // the empty end-of-file block
                scratch.clear();
                uint64_t end_bit = 0;
                // (the input ends where the block's deflate data ends: the inflater cannot read into the trailer or beyond)
                const int rc = gzpar::inflate_blocks<false>(data, k.in + k.in_len, (uint64_t)k.in * 8u, ~0ull >> 1, nullptr, 0, scratch, end_bit,
                                                            k.isize + 1024);
                if (rc != 1 || scratch.size() != k.isize) { bad[(size_t)t] = 1; break; }
                char* dst = out.data() + base + k.ooff;
                memcpy(dst, scratch.data(), k.isize);
                if ((uint32_t)gzpar::crc32_fast(crc32(0L, Z_NULL, 0), (const uint8_t*)dst, k.isize) != k.crc) { bad[(size_t)t] = 1; break; }
            }
        });
        for (int x : bad)
            if (x) { g_err = "BGZF block failed to inflate (corrupt file)"; failed = true; return false; }
        text_total += total;
        return true;
    }
};

// Ordinary gzip (one deflate stream per member): speculative parallel inflate, tps_gzpar.h.  The compressed file is mmap'ed.
struct GzSource : TextSource {
    int fd = -1;
    const uint8_t* data = nullptr;
    size_t size = 0;
    gzpar::ParGz z;
    ~GzSource() override {
        if (data) munmap((void*)data, size);
        if (fd >= 0) close(fd);
    }
    bool eof() const override { return z.eof(); }
    bool read_group(gzpar::TextBuf& out, size_t want) override {
        // (a round keeps 2 bytes per byte of text in flight: smaller groups than BGZF's)
        if (!z.read(out, std::min<size_t>(want, (size_t)96 << 20))) { g_err = "gzip: " + z.err; failed = true; return false; }
        return true;
    }
};

// The inflated text of a compressed input, one buffer per group of blocks.  Reference-counted: the reader holds the window it is
// decoding, and every packed batch handed to the caller holds the window its records' spans point into (header, sequence and
// quality text of the passing records are written out from there, long after the reader has moved on) -- tps_reader_text_hold /
// tps_text_release.  Released buffers are parked (a few hundred MB are not mapped, faulted in and unmapped per group).
struct TextHold {
    gzpar::TextBuf buf;                        // (resize() does not zero: the thread team writes the text)
    std::atomic<int> refs{1};
    static std::mutex& mu() { static std::mutex m; return m; }
    static std::vector<TextHold*>& parked() { static std::vector<TextHold*> v; return v; }
    static TextHold* get() {
        {
            std::lock_guard<std::mutex> lk(mu());
            if (!parked().empty()) { TextHold* h = parked().back(); parked().pop_back(); h->refs = 1; h->buf.clear(); return h; }
        }
        return new TextHold();
    }
    void ref() { refs.fetch_add(1); }
    void unref() {
        if (refs.fetch_sub(1) != 1) return;
        {
            std::lock_guard<std::mutex> lk(mu());
            if (parked().size() < 3) { parked().push_back(this); return; }
        }
        delete this;
    }
};

struct Fast {
    int fd = -1;
    TextSource* src = nullptr;                 // compressed input (BGZF, gzip): `data` is the text of `hold`, a new one per group
    TextHold* hold = nullptr;
    size_t fill_target = 0;                    // text a refilled window should hold (set by the packed reader from its caller's buffers)
    uint64_t base_off = 0;                     // uncompressed offset of data[0] (BGZF)
    const char* data = nullptr;
    size_t size = 0;
    size_t pos = 0;                            // start of the first unconsumed record
    size_t limit = (size_t)-1;                 // byte ranges (tps_reader_open_range): records that START at or behind this offset are not this reader's
    // ... of a BGZF file: a record belongs to the reader whose text holds the line end in front of it, i.e. it starts at a text
    // offset in [1, own_text] of the reader's own blocks (the first reader also owns offset 0); first_abs = where the first one began
    bool bgzf_range = false, bgzf_first = false;
    int64_t first_abs = -2;
    size_t eff_limit() const {
        if (!bgzf_range) return limit;
        if (limit == 0) return 0;                                           // (an empty range)
        Bgzf* z = src ? src->as_bgzf() : nullptr;
        if (!z || z->own_text == ~0ull) return (size_t)-1;
        return z->own_text + 1 > base_off ? (size_t)(z->own_text + 1 - base_off) : 0;
    }
    std::vector<uint64_t> nl;                  // newline offsets of the indexed window, ascending
    size_t nl_i = 0;                           // first unconsumed entry of nl
    size_t win_hi = 0;                         // end of the indexed window
    bool whole = false;                        // the window reaches the end of the file
    int threads = 1;
    // h0 / hl: header text (without '@' / '>'), s0: first base, sl: bases, q0: first quality character (FASTQ) or the end of the
    // record's sequence text (FASTA); wrapped: the sequence is spread over several lines (FASTA)
    struct Rec { uint64_t h0, hl, s0, sl, q0; bool wrapped; };
    std::vector<Rec> recs;

    ~Fast() {
        give_spare();
        if (src) {
            delete src;
            if (hold) hold->unref();             // (parked for the next file once the last batch that points into it is gone)
        } else if (data) {
            // Unmapping a few hundred MB that 16 threads have touched takes milliseconds (page-table teardown, TLB shootdowns,
            // and the GPU driver's MMU notifier when the process holds a device context: 7 ms for 300 MB on the GPU box, more
            // than decoding the file) -- off the caller's path: a detached thread does it.
            void* p = (void*)data;
            const size_t n = size;
            try {
                std::thread([p, n] { munmap(p, n); }).detach();
            } catch (...) {
                munmap(p, n);
            }
        }
        if (fd >= 0) close(fd);
    }
    // more text into the CURRENT window while nobody but the reader points into it (appending may move the buffer)
    void top_up(size_t target) {
        if (!src || !hold || hold->refs.load() != 1) return;
        size_t group = (size_t)128 << 20;
        if (const long long g = g_opt.bgzf_group.load()) group = (size_t)g;
        while (!src->eof() && !src->failed && size - pos < target) {
            src->read_group(hold->buf, group);
            data = hold->buf.data();
            size = hold->buf.size();
        }
    }
    // ASCII consumers (next) want the line index of every window; the packed decoder (next_packed) never looks at it -- a window
    // of inflated text is then only refilled (round 4: indexing 128 MB of lines per window for nothing).  `lazy`: a
    // compressed source's FIRST window is inflated by the first call that wants records, not by tps_reader_open -- an ASCII consumer
    // of compressed FASTA, which this decoder does not serve, no longer inflates a window it throws away.
    bool want_lines = true, lines_valid = false, lazy = false;
    void first_window() {
        if (!lazy) return;
        lazy = false;
        index_window();                                  // first group of blocks; leading blank lines skipped like the streaming decoder does
        if (bgzf_range && !bgzf_first) {
            // a later range of a BGZF file: its text begins somewhere inside a record -- the first line start that opens one
            pos = limit == 0 ? size : first_record_from(1, size);
            first_abs = pos < size ? (int64_t)(base_off + pos) : -2;
        } else if (bgzf_range) {
            first_abs = 0;
        }
        while (pos < size && (data[pos] == '\n' || data[pos] == '\r' || data[pos] == ' ')) ++pos;
        if (pos && want_lines) index_lines();
    }
    void index_window() {
        refill();
        if (want_lines) { index_lines(); return; }
        nl.clear(); nl_i = 0; lines_valid = false; whole = false; win_hi = size;
    }
    void refill() {
        if (src) {
            // a NEW buffer for the next group of blocks (batches already handed out keep pointing into the old one): the
            // unconsumed tail (a partial record) is copied to its front, the group is inflated behind it
            const size_t keep = size - pos;
            TextHold* nh = TextHold::get();
            nh->buf.resize(keep);
            if (keep) memcpy(nh->buf.data(), data + pos, keep);
            if (hold) hold->unref();
            hold = nh;
            base_off += pos;
            pos = 0;
            size_t group = (size_t)128 << 20;          // text per refill (tests shrink it to exercise the carry-over)
            if (const long long g = g_opt.bgzf_group.load()) group = (size_t)g;
            if (!src->eof()) src->read_group(hold->buf, group);
            // (packed batches: a window should hold a whole batch's worth of text -- nobody else points into this new buffer
            // yet, so further groups are simply appended to it)
            while (!src->eof() && !src->failed && hold->buf.size() < fill_target) src->read_group(hold->buf, group);
            data = hold->buf.data();
            size = hold->buf.size();
        }
    }
    void index_lines() {
        const double t_ix = io_timing() ? now_s() : 0.0;
        const size_t lo = pos, span = src ? size - lo : std::min<size_t>(size - lo, (size_t)512 << 20);
        win_hi = lo + span;
        whole = src ? src->eof() : win_hi == size;
        const int T = span < (8u << 20) ? 1 : threads;
        std::vector<std::vector<uint64_t>> part((size_t)T);
        team(T, [&](int t, int nt) {
            const size_t a = lo + span * (size_t)t / (size_t)nt, b = lo + span * (size_t)(t + 1) / (size_t)nt;
            auto& v = part[(size_t)t];
            v.reserve((b - a) / 2048 + 16);
            const char* p = data + a;
            const char* e = data + b;
            while (p < e) {
                const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
                if (!q) break;
                v.push_back((uint64_t)(q - data));
                p = q + 1;
            }
        });
        nl.clear();
        for (auto& v : part) nl.insert(nl.end(), v.begin(), v.end());
        if (whole && size && data[size - 1] != '\n') nl.push_back(size);      // last line without a newline
        nl_i = 0;
        lines_valid = true;
        if (io_timing()) fprintf(stderr, "[tps_io] index %.2f ms: %zu bytes, %zu lines, %d threads\n", 1e3 * (now_s() - t_ix), span, nl.size(), T);
    }
    // >= 0: records decoded; -3: not plain 4-line FASTQ here -> caller switches to the streaming decoder at `pos`
    int64_t next(uint8_t* bases, int64_t bases_cap, int64_t* offsets, int64_t max_records, char* heads, int64_t heads_cap,
                 int64_t* head_off, uint8_t* quals) {
        if (fasta) return -3;                          // (ASCII batches of FASTA: the streaming decoder, from this byte on)
        want_lines = true;
        first_window();
        if (src && src->failed) return -1;
        if (!lines_valid) index_lines();               // (the packed decoder had the window before: same text, now with its lines)
        recs.clear();
        int64_t nb = 0, nh = 0;
        offsets[0] = 0;
        head_off[0] = 0;
        size_t p = pos;
        while ((int64_t)recs.size() < max_records) {
            if (nl_i + 4 > nl.size()) {                  // fewer than four indexed lines left
                if (!whole) {
                    if (!recs.empty()) break;            // hand over what we have; the next call re-indexes
                    if (p >= size && (!src || src->eof())) break;
                    pos = p;
                    index_window();                      // window starts at the next record
                    p = pos;                             // (BGZF: the buffer was compacted)
                    if (src && src->failed) return -1;
                    if (nl.size() >= 4) continue;
                    if (!whole) return -3;               // one record larger than the window: streaming decoder
                }
                if (only_blank(p)) break;                // clean end of file
                if (recs.empty()) return -3;             // a partial / odd tail: let the streaming decoder judge it
                break;
            }
            const uint64_t e0 = nl[nl_i], e1 = nl[nl_i + 1], e2 = nl[nl_i + 2], e3 = nl[nl_i + 3];
            auto trim = [&](uint64_t a, uint64_t e) { return (e > a && data[e - 1] == '\r') ? e - 1 : e; };
            const uint64_t h0 = p, h1 = trim(p, e0), s0 = e0 + 1, s1 = trim(s0, e1), q0 = e2 + 1, q1 = trim(q0, e3);
            const bool ok = h1 > h0 && data[h0] == '@' && e1 + 1 < size && data[e1 + 1] == '+' && (s1 - s0) == (q1 - q0) &&
                            (s1 == s0 || (data[s0] != ' ' && data[s0] != '\t' && data[s1 - 1] != ' ' && data[s1 - 1] != '\t'));
            if (!ok) {
                if (recs.empty()) return -3;
                break;
            }
            const int64_t sl = (int64_t)(s1 - s0), hl = (int64_t)(h1 - h0 - 1);
            if (nb + sl > bases_cap || nh + hl > heads_cap) {
                if (recs.empty()) return -2;
                break;
            }
            recs.push_back(Rec{h0 + 1, (uint64_t)hl, s0, (uint64_t)sl, q0, false});
            nb += sl;
            nh += hl;
            offsets[recs.size()] = nb;
            head_off[recs.size()] = nh;
            nl_i += 4;
            p = std::min<size_t>((size_t)e3 + 1, size);
        }
        return (int64_t)finish(bases, offsets, heads, head_off, quals, p);
    }
    // Packed variant of next() for mmap'ed plain FASTQ: records are packed (tps_pack.h) into seq2 / inv / desc, no ASCII
    // copy.  spans gets 4 entries per record: header offset and length, sequence offset, quality offset (text offsets).
    // -3: not plain 4-line FASTQ here; -2: a single record does not fit.
    // One 4-line FASTQ record at text offset s: 0 = well formed (r filled, next = start of the following record), 1 = not a
    // plain 4-line record here (or the text ends inside it).  The quality line is located by the sequence length and only
    // the byte behind it is looked at -- its own bytes are never read.
    bool fasta = false;                        // two-line FASTA records ('>' header, the whole sequence on one line) instead of FASTQ
    bool final_window() const { return !src || src->eof(); }      // the text in hand reaches the end of the input
    // One FASTA record: the header line, then sequence lines up to the next line that begins with '>' (or the end of the input).
    // 0 = well formed (r filled, next = start of the following record), 1 = not this decoder's business (a line with leading /
    // trailing blanks, which the streaming decoder strips; text that ends inside the record while the source has more).  A
    // sequence on ONE line is packed from the text itself; a wrapped one (60 / 80 columns, what most FASTA files look like) is
    // joined line by line into `dewrap` (the calling thread's scratch) and packed from there -- round 4: wrapped FASTA used to go
    // to the one-thread streaming decoder (6.6e8 against 2.4e9 bases/s for the same reads on one line each).
    int parse_fasta_at(size_t s, Rec& r, size_t& next, std::vector<uint8_t>* dewrap = nullptr) const {
        if (s >= size || data[s] != '>') return 1;
        const char* e0p = (const char*)memchr(data + s, '\n', size - s);
        if (!e0p) return 1;
        const size_t e0 = (size_t)(e0p - data);
        const size_t h1 = (e0 > s && data[e0 - 1] == '\r') ? e0 - 1 : e0;
        if (h1 <= s) return 1;
        const size_t s0 = e0 + 1;
        size_t p = s0, first_a = 0, first_b = 0;       // the first non-empty sequence line
        uint64_t sl = 0;
        int nlines = 0;
        for (;;) {
            if (p >= size) { if (!final_window()) return 1; next = size; break; }       // (the record may go on in the next group of blocks)
            if (data[p] == '>') { next = p; break; }
            const char* ep = (const char*)memchr(data + p, '\n', size - p);
            if (!ep && !final_window()) return 1;
            const size_t e = ep ? (size_t)(ep - data) : size;
            const size_t l1 = (e > p && data[e - 1] == '\r') ? e - 1 : e;
            if (l1 > p) {
                if (data[p] == ' ' || data[p] == '\t' || data[l1 - 1] == ' ' || data[l1 - 1] == '\t') return 1;
                if (nlines == 0) { first_a = p; first_b = l1; }
                else if (dewrap) {
                    if (nlines == 1) dewrap->assign((const uint8_t*)data + first_a, (const uint8_t*)data + first_b);
                    dewrap->insert(dewrap->end(), (const uint8_t*)data + p, (const uint8_t*)data + l1);
                }
                sl += l1 - p;
                ++nlines;
            }
            p = ep ? e + 1 : size;
        }
        if (sl > 0x7FFFFFFFull) return 1;
        r = Rec{s + 1, (uint64_t)(h1 - s - 1), nlines ? first_a : s0, sl, next, nlines > 1};
        return 0;
    }
    // A FASTQ record whose sequence (and quality) is spread over several lines (round 4; it used to end the packed decoding of the
    // file): header at s (ends at h1), sequence lines from s0 up to the first line that begins with '+', then quality lines until
    // they hold as many characters as the sequence -- which has to happen exactly at a line end ('@' and '+' may begin a quality
    // line: lengths decide, as in Biopython's FastqGeneralIterator).  The sequence is joined into `dewrap` like a wrapped FASTA
    // record's; r.q0 is the first quality character (consumers join the quality lines by count).  1 = not this decoder's business:
    // blank or padded lines inside the record, a sequence line that begins with '@' (a mis-framed candidate), text that ends inside
    // the record.
    int parse_fastq_multiline_at(size_t s, size_t h1, size_t s0, Rec& r, size_t& next, std::vector<uint8_t>* dewrap) const {
        size_t p = s0, first_a = 0, first_b = 0;
        uint64_t sl = 0;
        int nlines = 0;
        for (;;) {                                     // sequence lines
            if (p >= size) return 1;
            if (data[p] == '+') break;
            if (data[p] == '@') return 1;
            const char* ep = (const char*)memchr(data + p, '\n', size - p);
            if (!ep) return 1;
            const size_t e = (size_t)(ep - data);
            const size_t l1 = (e > p && data[e - 1] == '\r') ? e - 1 : e;
            if (l1 == p) return 1;                     // a blank line inside a record: the streaming decoder judges that
            if (data[p] == ' ' || data[p] == '\t' || data[l1 - 1] == ' ' || data[l1 - 1] == '\t') return 1;
            if (nlines == 0) { first_a = p; first_b = l1; }
            else if (dewrap) {
                if (nlines == 1) dewrap->assign((const uint8_t*)data + first_a, (const uint8_t*)data + first_b);
                dewrap->insert(dewrap->end(), (const uint8_t*)data + p, (const uint8_t*)data + l1);
            }
            sl += l1 - p;
            ++nlines;
            p = e + 1;
        }
        if (nlines == 0 || sl > 0x7FFFFFFFull) return 1;         // (an empty sequence has its '+' right behind the header: the 4-line path)
        const char* epl = (const char*)memchr(data + p, '\n', size - p);
        if (!epl) return 1;
        size_t q = (size_t)(epl - data) + 1;
        const size_t q0 = q;
        uint64_t ql = 0;
        while (ql < sl) {                              // quality lines
            if (q >= size) return 1;
            const char* ep = (const char*)memchr(data + q, '\n', size - q);
            if (!ep && !final_window()) return 1;
            const size_t e = ep ? (size_t)(ep - data) : size;
            const size_t l1 = (e > q && data[e - 1] == '\r') ? e - 1 : e;
            if (l1 == q) return 1;
            ql += l1 - q;
            if (!ep) { if (ql != sl) return 1; q = size; break; }
            q = e + 1;
        }
        if (ql != sl) return 1;
        if (q < size && data[q] != '@' && !only_blank(q)) return 1;      // what follows has to be a record (or the end of the input)
        next = q;
        r = Rec{s + 1, (uint64_t)(h1 - s - 1), first_a, sl, q0, nlines > 1};
        return 0;
    }
    int parse_at(size_t s, Rec& r, size_t& next, std::vector<uint8_t>* dewrap = nullptr) const {
        if (fasta) return parse_fasta_at(s, r, next, dewrap);
        if (s >= size || data[s] != '@') return 1;
        const char* e0p = (const char*)memchr(data + s, '\n', size - s);
        if (!e0p) return 1;
        const size_t e0 = (size_t)(e0p - data);
        const size_t h1 = (e0 > s && data[e0 - 1] == '\r') ? e0 - 1 : e0;
        if (h1 <= s) return 1;
        const size_t s0 = e0 + 1;
        if (s0 >= size) return 1;
        const char* e1p = (const char*)memchr(data + s0, '\n', size - s0);
        if (!e1p) return 1;
        const size_t e1 = (size_t)(e1p - data);
        const size_t s1 = (e1 > s0 && data[e1 - 1] == '\r') ? e1 - 1 : e1;
        if (e1 + 1 >= size) return 1;
        if (data[e1 + 1] != '+') return parse_fastq_multiline_at(s, h1, s0, r, next, dewrap);
        const char* e2p = (const char*)memchr(data + e1 + 1, '\n', size - (e1 + 1));
        if (!e2p) return 1;
        const size_t q0 = (size_t)(e2p - data) + 1, sl = s1 - s0;
        size_t q1 = q0 + sl;                                    // where the quality line has to end
        if (q1 > size) return 1;
        if (q1 < size) {
            if (data[q1] == '\r' && q1 + 1 < size && data[q1 + 1] == '\n') next = q1 + 2;
            else if (data[q1] == '\n') next = q1 + 1;
            else return 1;                                      // more (or fewer) quality characters than bases
        } else {
            // the quality line ends exactly where the text in hand ends: the last line of the input without a newline -- or, in a
            // window of inflated text, a record whose line end is the first byte of the NEXT window (accepting it here
            // left that newline at the head of the next window, the packed decoder gave up there and the rest of the file went
            // through the one-thread streaming decoder): the record waits for the next window
            if (!final_window()) return 1;
            next = size;
        }
        // A quality line SHORTER than the sequence puts q1 inside the next record; that is only mistaken for a line end if that
        // record's header happens to end exactly there -- and then what follows is not a record start.  Scanning every quality
        // line for a stray newline cost a third of the decoding time, so the bytes are only looked at in that case: the byte
        // behind the record is not '@' (such a record was accepted and written out with an embedded newline).
        if (sl && data[q1 - 1] == '\n') return 1;
        if (next < size && data[next] != '@' && memchr(data + q0, '\n', sl)) return 1;
        if (sl > 0x7FFFFFFFull) return 1;
        if (sl && (data[s0] == ' ' || data[s0] == '\t' || data[s1 - 1] == ' ' || data[s1 - 1] == '\t')) return 1;
        r = Rec{s + 1, (uint64_t)(h1 - s - 1), s0, (uint64_t)sl, q0, false};
        return 0;
    }
    // What one thread of the team found in its stretch of the text: the records that START there, already packed into
    // position-independent staging (every read begins on a quad boundary, so a run of reads is copied as one block).
    struct Chunk {
        std::vector<Rec> recs;
        std::vector<uint32_t> seq2;
        std::vector<uint16_t> inv;
        std::vector<uint8_t> bad;                  // per record: has an invalid base
        std::vector<int64_t> woff;                 // per record: word offset inside the staging
        std::vector<uint8_t> dewrap;               // a wrapped FASTA sequence joined into one stretch (the record being packed)
        std::vector<uint8_t> ends;                 // heads mode: the first + last heads_bp bases of the record being packed
        size_t first = 0, end = 0;                 // start of the first record, start of the record after the last
        bool odd = false;                          // stopped at something that is not a plain 4-line record (at `end`)
        int64_t base_rec = 0, base_word = 0, base_head = 0, take = 0;
    };
    std::vector<Chunk> chunks;
    // the staging of a finished reader is kept for the next one (a few tens of MB that would otherwise be allocated, faulted
    // in and freed again for every input file)
    static std::mutex& spare_mu() { static std::mutex m; return m; }
    static std::vector<std::vector<Chunk>>& spare() { static std::vector<std::vector<Chunk>> v; return v; }
    void take_spare() {
        std::lock_guard<std::mutex> lk(spare_mu());
        if (!spare().empty()) { chunks.swap(spare().back()); spare().pop_back(); }
    }
    void give_spare() {
        if (chunks.empty()) return;
        std::lock_guard<std::mutex> lk(spare_mu());
        if (spare().size() < 2) { spare().emplace_back(); spare().back().swap(chunks); }
    }
    // heads_bp > 0 ("heads mode", round 4): a read longer than 2 heads_bp is packed as its first heads_bp + its last heads_bp bases
    // only -- all step 1 looks at (allsteps.py:176-177); desc[i].len is that pseudo-read's length and full_len[i] the read's own.
    // The caller uploads ~2 kb per read instead of the whole read, runs step 1, and packs the scanned part of the FEW reads that
    // pass from their spans afterwards (tps_pack_spans).
    int64_t next_packed(uint32_t* seq2, uint16_t* inv, int64_t words_cap, tps_read_desc* desc, int64_t max_records, char* heads,
                        int64_t heads_cap, int64_t* head_off, int64_t* spans, int32_t heads_bp = 0, int32_t* full_len = nullptr) {
        // The text from the first unconsumed record on is cut into one stretch per thread; a thread finds the first record
        // that starts in its stretch (a line that begins with '@', is followed by a sequence line, a '+' line and a quality
        // line of the sequence's length -- a quality line that happens to begin with '@' fails that test), then decodes
        // record after record: two memchr for the header and the sequence, the quality line skipped by length, the bases
        // packed (AVX2) into the thread's staging while they are still in cache.  The stretches are then joined in order
        // (each has to begin exactly where the previous one ended), capped at what the caller's buffers hold, and copied out.
        // Text read: headers + sequence lines once.  Not read: the quality lines (half of the file).
        head_off[0] = 0;
        nl.clear(); nl_i = 0; whole = false;       // (the line index of next() is not used here and is stale afterwards)
        const double t_a = io_timing() ? now_s() : 0.0;
        const size_t p = pos;
        const size_t lim = eff_limit();
        if (p >= size || p >= lim || only_blank(p)) return 0;
        // text that yields at most words_cap words if it were nothing but sequence + quality lines
        size_t span = std::min<size_t>(std::min(size, lim) - p, (size_t)std::max<int64_t>(words_cap, 1024) * (fasta ? 17 : 32));      // (FASTA: no quality lines; a record that starts before `limit` is decoded to its end)
        const size_t min_span = g_opt.pack_min_span.load() >= 0 ? (size_t)g_opt.pack_min_span.load() : (size_t)4 << 20;   // (tests: team on small files)
        const int T = span < min_span ? 1 : threads;
        if (chunks.empty()) take_spare();
        if ((int)chunks.size() < T) chunks.resize((size_t)T);
        team(T, [&](int t, int nt) {
            Chunk& c = chunks[(size_t)t];
            c.recs.clear(); c.seq2.clear(); c.inv.clear(); c.bad.clear(); c.woff.clear();
            c.odd = false; c.take = 0;
            const size_t a = p + span * (size_t)t / (size_t)nt, b = p + span * (size_t)(t + 1) / (size_t)nt;
            size_t s = a;
            Rec r;
            size_t nx = 0;
            if (t > 0) {
                // first line start >= a that opens a well-formed record (a itself counts if the previous byte ends a line)
                s = b;
                const char* q = (const char*)memchr(data + a - 1, '\n', b - (a - 1));
                while (q) {
                    const size_t cand = (size_t)(q - data) + 1;
                    if (cand >= b) break;
                    if (data[cand] == (fasta ? '>' : '@') && parse_at(cand, r, nx) == 0) { s = cand; break; }
                    q = (const char*)memchr(data + cand, '\n', b - cand);
                }
            }
            c.first = c.end = s;
            c.seq2.reserve((b - a) / 28 + 1024);
            if (inv) c.inv.reserve((b - a) / 28 + 1024);
            while (s < b) {
                if (parse_at(s, r, nx, &c.dewrap) != 0) { c.odd = true; break; }
                const uint8_t* bases = r.wrapped ? c.dewrap.data() : (const uint8_t*)data + r.s0;
                int64_t pl = (int64_t)r.sl;                                 // bases packed for this record
                if (heads_bp > 0 && pl > 2 * (int64_t)heads_bp) {
                    c.ends.resize((size_t)(2 * heads_bp));
                    memcpy(c.ends.data(), bases, (size_t)heads_bp);
                    memcpy(c.ends.data() + heads_bp, bases + r.sl - heads_bp, (size_t)heads_bp);
                    bases = c.ends.data();
                    pl = 2 * (int64_t)heads_bp;
                }
                const int64_t w = tps::packed_words(pl), at = (int64_t)c.seq2.size();
                c.seq2.resize((size_t)(at + w));
                if (inv) c.inv.resize((size_t)(at + w));
                const bool bad = tps::pack_one(bases, pl, c.seq2.data() + at, inv ? c.inv.data() + at : nullptr);
                c.recs.push_back(r);
                c.bad.push_back(bad ? 1 : 0);
                c.woff.push_back(at);
                s = nx;
                c.end = s;
            }
        });
        const double t_b = io_timing() ? now_s() : 0.0;
        // join: stretch t has to begin where the records before it ended (a stretch in which no record starts is empty)
        int64_t n = 0, nw = 0, nh = 0;
        size_t cur = p;
        bool stop = false, too_big = false;
        for (int t = 0; t < T && !stop; ++t) {
            Chunk& c = chunks[(size_t)t];
            c.base_rec = n; c.base_word = nw; c.base_head = nh;
            if (c.recs.empty()) {
                if (c.odd && c.first == cur) stop = true;     // the very next record is not a plain one
                continue;
            }
            if (c.first != cur) break;                        // mis-framed stretch: the next call starts at `cur`, a known record start
            for (size_t i = 0; i < c.recs.size(); ++i) {
                const int64_t sl_full = (int64_t)c.recs[i].sl;
                const int64_t pl = (heads_bp > 0 && sl_full > 2 * (int64_t)heads_bp) ? 2 * (int64_t)heads_bp : sl_full;
                const int64_t w = tps::packed_words(pl), hl = (int64_t)c.recs[i].hl;
                if (n >= max_records || nw + w > words_cap || nh + hl > heads_cap) {
                    too_big = n == 0 && (w > words_cap || hl > heads_cap);
                    stop = true;
                    break;
                }
                desc[n].word_off = nw;
                desc[n].len = (int32_t)pl;
                if (full_len) full_len[n] = (int32_t)sl_full;
                desc[n].flags = c.bad[i] ? TPS_RD_HAS_INVALID : 0;
                nw += w; nh += hl; ++n;
                head_off[n] = nh;
                ++c.take;
                cur = i + 1 < c.recs.size() ? (size_t)(c.recs[i + 1].h0 - 1) : c.end;
            }
            if (c.odd) stop = true;
        }
        if (n == 0) {
            if (too_big) return -2;
            return -3;                                        // not plain 4-line FASTQ at `pos`: the streaming decoder judges it
        }
        team(T, [&](int t, int) {
            const Chunk& c = chunks[(size_t)t];
            if (!c.take) return;
            const int64_t words = (c.take < (int64_t)c.recs.size()) ? c.woff[(size_t)c.take] : (int64_t)c.seq2.size();
            memcpy(seq2 + c.base_word, c.seq2.data(), (size_t)words * sizeof(uint32_t));
            if (inv) memcpy(inv + c.base_word, c.inv.data(), (size_t)words * sizeof(uint16_t));
            for (int64_t i = 0; i < c.take; ++i) {
                const Rec& r = c.recs[(size_t)i];
                const int64_t g = c.base_rec + i;
                memcpy(heads + head_off[g], data + r.h0, (size_t)r.hl);
                if (spans) {
                    spans[4 * g] = (int64_t)r.h0; spans[4 * g + 1] = (int64_t)r.hl;
                    spans[4 * g + 2] = (int64_t)r.s0; spans[4 * g + 3] = (int64_t)r.q0;
                }
            }
        });
        if (io_timing()) fprintf(stderr, "[tps_io] batch of %lld records: decode+pack %.2f ms, join+copy %.2f ms (%lld words, %d threads, %zu bytes of text)\n",
                                 (long long)n, 1e3 * (t_b - t_a), 1e3 * (now_s() - t_b), (long long)nw, T, cur - p);
        pos = cur;
        release_behind();
        return n;
    }
    // Plain files: the pages of the mapping well behind the read position leave the process's resident set (MADV_DONTNEED on a
    // read-only file mapping only drops the page-table entries -- the page cache keeps the data, a late reader faults it back in):
    // a 3 GB input used to sit in RSS for the whole run (round 4's configs[4] shard: 4.8 GB resident with the plain file).
    size_t released = 0;
    void release_behind() {
        if (src || !data) return;
        const size_t lag = (size_t)256 << 20, step = (size_t)64 << 20;
        if (pos < lag + step || pos - lag < released + step) return;
        const size_t hi = (pos - lag) & ~(size_t)4095;
        if (hi > released) { madvise((void*)(data + released), hi - released, MADV_DONTNEED); released = hi; }
    }
    // Byte ranges: the first line start >= lo that opens a record -- '>' for FASTA; for FASTQ an '@' line that parse_at accepts (a
    // quality line may begin with '@': the four-line / multi-line framing test rejects it) -- or `hi` if no record starts in [lo, hi).
    size_t first_record_from(size_t lo, size_t hi) const {
        if (lo == 0) return 0;
        if (hi > size) hi = size;
        if (lo >= hi) return hi;
        Rec r;
        size_t nx = 0;
        const char* q = (const char*)memchr(data + lo - 1, '\n', hi - (lo - 1));
        while (q) {
            const size_t cand = (size_t)(q - data) + 1;
            if (cand >= hi) break;
            if (fasta ? data[cand] == '>' : (data[cand] == '@' && parse_at(cand, r, nx) == 0)) return cand;
            q = (const char*)memchr(data + cand, '\n', hi - cand);
        }
        return hi;
    }
    bool only_blank(size_t from) const {
        for (size_t i = from; i < size; ++i)
            if (data[i] != '\n' && data[i] != '\r' && data[i] != ' ') return false;
        return true;
    }
    size_t finish(uint8_t* bases, const int64_t* offsets, char* heads, const int64_t* head_off, uint8_t* quals, size_t new_pos) {
        const size_t n = recs.size();
        const int T = (int64_t)offsets[n] < (4 << 20) ? 1 : threads;
        team(T, [&](int t, int nt) {
            const size_t a = n * (size_t)t / (size_t)nt, b = n * (size_t)(t + 1) / (size_t)nt;
            for (size_t i = a; i < b; ++i) {
                const Rec& r = recs[i];
                memcpy(bases + offsets[i], data + r.s0, (size_t)r.sl);
                if (quals) memcpy(quals + offsets[i], data + r.q0, (size_t)r.sl);
                memcpy(heads + head_off[i], data + r.h0, (size_t)r.hl);
            }
        });
        pos = new_pos;
        return n;
    }
};

struct Handle {
    Reader* slow = nullptr;
    Fast* fast = nullptr;
    std::string path;
    int format = 0;
    // byte ranges (tps_reader_open_range): the reader owns the records that start in [first, limit); first = the first record start
    // at or behind the range's lower bound
    int64_t limit = -1, first = 0, stream_stopped = -1;
    ~Handle() {
        if (slow) { if (slow->gz) gzclose(slow->gz); delete slow; }
        delete fast;
    }
};

}  // namespace

extern "C" {

const char* tps_io_last_error(void) { return g_err.c_str(); }

// Tests and diagnostics (process-wide; readers opened afterwards see the new value): "threads" (0 = by the host's CPUs and cgroup
// quota), "timing", "bgzf_group" (bytes; 0 = 128 MiB), "pack_min_span" (bytes; -1 = 4 MiB), "no_pargz", "pargz_min" (bytes; -1 = 1 MiB).
int tps_io_set_option(const char* key, int64_t value) {
    if (!key) { g_err = "null argument"; return -1; }
    const std::string k(key);
    if (k == "threads") g_opt.threads = (int)std::max<int64_t>(0, std::min<int64_t>(value, 64));
    else if (k == "timing") g_opt.timing = value != 0;
    else if (k == "bgzf_group") g_opt.bgzf_group = std::max<int64_t>(0, value);
    else if (k == "pack_min_span") g_opt.pack_min_span = value;
    else if (k == "no_pargz") g_opt.no_pargz = value != 0;
    else if (k == "pargz_min") g_opt.pargz_min = value;
    else { g_err = "unknown option '" + k + "' (threads, timing, bgzf_group, pack_min_span, no_pargz, pargz_min)"; return -1; }
    return 0;
}

// Opens a FASTA/FASTQ file (plain or .gz).  The format comes from the first byte, like check_file_type
// (allsteps.py:36-50).  Plain FASTQ files are mmap'ed for the thread-team decoder.  Returns 0 or -1.
static Reader* open_stream(const char* path, int64_t seek_to, int format, size_t buf_bytes = (size_t)4 << 20) {
    gzFile gz = gzopen(path, "rb");
    if (!gz) { g_err = std::string("cannot open ") + path; return nullptr; }
    gzbuffer(gz, buf_bytes < (1u << 20) ? (unsigned)buf_bytes : 1u << 20);
    if (seek_to > 0 && gzseek(gz, (z_off_t)seek_to, SEEK_SET) < 0) { g_err = "seek failed"; gzclose(gz); return nullptr; }
    Reader* r = new Reader();
    r->gz = gz;
    r->buf.resize(buf_bytes);
    r->format = format;
    return r;
}

int tps_reader_open(const char* path, void** out) {
    if (!path || !out) { g_err = "null argument"; return -1; }
    *out = nullptr;
    Handle* h = new Handle();
    h->path = path;
    // plain file?  (gzip magic 1f 8b otherwise)
    bool plain = false;
    {
        FILE* f = fopen(path, "rb");
        if (!f) { g_err = std::string("cannot open ") + path; delete h; return -1; }
        unsigned char m[2] = {0, 0};
        size_t got = fread(m, 1, 2, f);
        fclose(f);
        plain = !(got == 2 && m[0] == 0x1f && m[1] == 0x8b);
    }
    // (a plain file only needs its first bytes here -- the thread-team decoder works on the mapping; the streaming
    // decoder's buffer grows to its working size when it is really used)
    Reader* r = open_stream(path, 0, 0, plain ? (size_t)64 << 10 : (size_t)4 << 20);
    if (!r) { delete h; return -1; }
    h->slow = r;
    g_err.clear();
    if (!r->fill() && r->len == 0) {
        if (!g_err.empty()) { delete h; return -1; }                             // unreadable (corrupt gzip), not empty
        h->format = 0; *out = h; return 0;                                       // empty file: no records
    }
    // the format comes from the first character of the FIRST line, like check_file_type (allsteps.py:36-50): a file that
    // starts with a blank line is "format cannot be identified" there, so it is here
    size_t i = 0;
    char c = i < r->len ? r->buf[i] : 0;
    r->format = h->format = c == '>' ? 1 : c == '@' ? 2 : 0;
    if (!h->format) {
        g_err = "format cannot be identified (first character is neither '>' nor '@')";
        delete h;
        return -1;
    }
    // Text whose first line ends in a lone CR (classic Mac OS line ends): only the streaming decoder reads that as the reference
    // does; the thread-team decoders below split at "\n" and are left out
    {
        const char* p0 = r->buf.data();
        for (size_t j = i; j < r->len; ++j) {
            if (p0[j] == '\n') break;
            if (p0[j] == '\r') { r->cr_lines = j + 1 < r->len && p0[j + 1] != '\n'; break; }
        }
    }
    const bool cr_lines = r->cr_lines;
    if (!plain && !cr_lines) {
        // bgzip'ed FASTQ: blocks inflate in parallel, then the same thread-team record decoder runs over the text
        Bgzf* z = new Bgzf();
        z->fd = open(path, O_RDONLY);
        struct stat st;
        if (z->fd >= 0 && fstat(z->fd, &st) == 0 && st.st_size >= 28) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, z->fd, 0);
            if (m != MAP_FAILED) {
                z->data = (const uint8_t*)m;
                z->size = (size_t)st.st_size;
                size_t bs = 0, hd = 0;
                if (Bgzf::block_at(z->data, z->size, bs, hd)) {
                    madvise(m, z->size, MADV_SEQUENTIAL);
                    z->threads = io_threads();
                    Fast* f = new Fast();
                    f->fasta = h->format == 1;
                    f->src = z;
                    f->threads = z->threads;
                    f->pos = 0;
                    f->lazy = true;                      // (the first group of blocks is inflated by the first call that wants records)
                    h->fast = f;
                    z = nullptr;
                }
            }
        }
        delete z;
    }
    if (!plain && !cr_lines && !h->fast && !g_opt.no_pargz.load()) {
        // ordinary gzip'ed FASTQ: the deflate stream is inflated by the thread team (speculative block starts, tps_gzpar.h),
        // then the same thread-team record decoder runs over the text.  Small files stay with zlib's stream.
        GzSource* z = new GzSource();
        z->fd = open(path, O_RDONLY);
        struct stat st;
        const long long pm = g_opt.pargz_min.load();                // (tests: the team's inflater on small files too)
        if (z->fd >= 0 && fstat(z->fd, &st) == 0 && st.st_size >= (pm >= 0 ? (off_t)pm : ((off_t)1 << 20)) && st.st_size >= 64) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, z->fd, 0);
            if (m != MAP_FAILED) {
                z->data = (const uint8_t*)m;
                z->size = (size_t)st.st_size;
                madvise(m, z->size, MADV_SEQUENTIAL);
                z->z.data = z->data;
                z->z.size = z->size;
                z->z.threads = io_threads();
                z->z.timing = io_timing();
                z->z.team = [](int n, const std::function<void(int, int)>& f) { team(n, f); };
                Fast* f = new Fast();
                    f->fasta = h->format == 1;
                f->src = z;
                f->threads = z->z.threads;
                f->pos = 0;
                f->lazy = true;
                h->fast = f;
                z = nullptr;
            }
        }
        delete z;
    }
    if (plain && !cr_lines) {
        Fast* f = new Fast();
                    f->fasta = h->format == 1;
        f->fd = open(path, O_RDONLY);
        struct stat st;
        if (f->fd >= 0 && fstat(f->fd, &st) == 0 && st.st_size > 0) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, f->fd, 0);
            if (m != MAP_FAILED) {
                madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                f->data = (const char*)m;
                f->size = (size_t)st.st_size;
                f->pos = i;                        // leading blank lines skipped like the streaming decoder does
                f->threads = io_threads();
                h->fast = f;
                f = nullptr;
            }
        }
        delete f;
    }
    *out = h;
    return 0;
}

// One reader per BYTE RANGE of a plain (uncompressed) FASTA / FASTQ file: it yields the records that START in [lo, hi) -- the first
// one is found like a thread of the team finds the first record of its stretch, the last one is decoded to its end beyond hi -- so
// readers over adjacent ranges partition the file's records, each with a team of its own (`threads`; 0 = the default), and one big
// file feeds several GPUs (the reference's advice for "> 20 GB and / or > 1 million reads" is to split the file by hand:
// README.md:267-268).  tps_reader_range_info tells where the reader's first record began and where it stopped, for the caller's
// seam check (reader i must have stopped where reader i + 1 began).  Compressed input: -1 (BGZF and gzip go through one reader).
int tps_reader_open_range(const char* path, int64_t lo, int64_t hi, int32_t threads, void** out) {
    if (lo < 0 || hi < lo) { g_err = "bad byte range"; return -1; }
    if (tps_reader_open(path, out) != 0) return -1;
    Handle* h = (Handle*)*out;
    if (!h->format) return 0;                                                // an empty file
    if (h->fast && h->fast->src && h->fast->src->as_bgzf()) {
        // BGZF: the range is one of COMPRESSED bytes; the reader owns the blocks that start in it
        Fast* f = h->fast;
        Bgzf* z = f->src->as_bgzf();
        if (threads > 0) { f->threads = std::min(threads, 64); z->threads = f->threads; }
        const size_t lo_c = std::min<size_t>((size_t)lo, z->size), hi_c2 = std::min<size_t>((size_t)hi, z->size);
        z->cpos = lo_c == 0 ? 0 : Bgzf::find_block(z->data, z->size, lo_c);
        z->own_end = hi_c2 >= z->size ? z->size : Bgzf::find_block(z->data, z->size, hi_c2);
        f->bgzf_range = true;
        f->bgzf_first = lo_c == 0;
        if (z->cpos >= z->own_end) f->limit = 0;                             // no block starts in this range
        h->limit = (int64_t)hi_c2;
        h->first = -2;
        return 0;
    }
    if (!h->fast || h->fast->src) { g_err = "byte ranges need a plain or BGZF-compressed FASTA / FASTQ file (an ordinary .gz is one stream)"; delete h; *out = nullptr; return -1; }
    Fast* f = h->fast;
    if (threads > 0) f->threads = std::min(threads, 64);
    const size_t hi_c = std::min<size_t>((size_t)hi, f->size);
    if (lo > 0) f->pos = f->first_record_from(std::min<size_t>((size_t)lo, f->size), hi_c);
    f->limit = hi_c;
    h->limit = (int64_t)hi_c;
    h->first = f->pos < hi_c ? (int64_t)f->pos : -2;                        // -2: no record starts in this range
    return 0;
}
int tps_reader_range_info(void* hv, int64_t* first, int64_t* stopped) {
    Handle* h = (Handle*)hv;
    if (!h || !first || !stopped) { g_err = "null argument"; return -1; }
    *first = h->first;
    int64_t at = 0;
    if (h->fast && h->fast->bgzf_range) {
        // BGZF: `first` counts from the start of the reader's own text, `stopped` from its END (= the start of the next reader's text)
        Fast* f = h->fast;
        if (f->lazy && f->limit != 0) f->first_window();
        *first = f->first_abs;
        Bgzf* z = f->src->as_bgzf();
        size_t p = f->pos;
        while (p < f->size && (f->data[p] == '\n' || f->data[p] == '\r' || f->data[p] == ' ')) ++p;
        *stopped = (f->limit == 0 || z->own_text == ~0ull) ? -2 : (int64_t)(f->base_off + p) - (int64_t)z->own_text;
        return 0;
    }
    if (h->fast) {
        size_t p = h->fast->pos;
        while (p < h->fast->size && (h->fast->data[p] == '\n' || h->fast->data[p] == '\r' || h->fast->data[p] == ' ')) ++p;     // (blank lines between records are nobody's)
        at = (int64_t)p;
    } else if (h->stream_stopped >= 0) {
        at = h->stream_stopped;
    } else if (h->slow) {
        at = (int64_t)(h->slow->have_pending ? h->slow->rec_start : h->slow->base + h->slow->pos);
    }
    *stopped = at;
    return 0;
}

int tps_reader_format(void* h) { return h ? ((Handle*)h)->format : 0; }

void tps_reader_close(void* h) { delete (Handle*)h; }

// Decodes up to max_records records into the caller's buffers, stopping before a record that would
// overflow bases_cap / heads_cap.  offsets / head_off get n+1 entries (offsets[0] = 0).  quals may be
// NULL; otherwise it receives the quality strings in the same layout as bases (FASTQ only; records
// whose quality length differs from the sequence length are padded with '!' / truncated).
// Returns the number of records (0 at end of input), -1 on a parse error, -2 if a single record
// does not fit into empty buffers.
int64_t tps_reader_next(void* hv, uint8_t* bases, int64_t bases_cap, int64_t* offsets, int64_t max_records, char* heads,
                        int64_t heads_cap, int64_t* head_off, uint8_t* quals) {
    Handle* h = (Handle*)hv;
    if (!h || !bases || !offsets || !heads || !head_off) { g_err = "null argument"; return -1; }
    offsets[0] = 0;
    head_off[0] = 0;
    if (!h->format) return 0;
    if (h->fast && h->fast->bgzf_range) {
        g_err = "records the thread-team decoder does not take inside a BGZF byte range: read this file with one reader";
        return -1;
    }
    if (h->fast) {
        // (a byte-range reader's ASCII batches come from the streaming decoder, which knows where every record begins)
        const int64_t n = h->limit >= 0 ? -3 : h->fast->next(bases, bases_cap, offsets, max_records, heads, heads_cap, head_off, quals);
        if (n != -3) return n;
        // not plain 4-line FASTQ from here on: the streaming decoder takes over at the same byte
        const int64_t at = (int64_t)(h->fast->base_off + h->fast->pos);
        delete h->fast;
        h->fast = nullptr;
        if (h->slow) { if (h->slow->gz) gzclose(h->slow->gz); delete h->slow; }
        h->slow = open_stream(h->path.c_str(), at, h->format);
        if (!h->slow) return -1;
        h->slow->base = (uint64_t)at;
    }
    Reader* r = h->slow;
    if (r->buf.size() < ((size_t)4 << 20)) r->buf.resize((size_t)4 << 20);      // (opened with the small sniffing buffer)
    int64_t n = 0, nb = 0, nh = 0;
    g_err.clear();
    while (n < max_records) {
        if (!r->have_pending) {
            if (!r->parse_one()) {
                if (!g_err.empty()) return -1;
                break;
            }
            if (h->limit >= 0 && r->rec_start >= (uint64_t)h->limit) {               // the next shard's record
                h->stream_stopped = (int64_t)r->rec_start;
                r->eof = true; r->len = r->pos = 0; r->have_next_head = false;
                break;
            }
            r->have_pending = true;
        }
        const int64_t sl = (int64_t)r->p_seq.size(), hl = (int64_t)r->p_head.size();
        if (nb + sl > bases_cap || nh + hl > heads_cap) {
            if (n == 0) { g_err = "record larger than the batch buffers"; return -2; }
            break;                       // keep it for the next call
        }
        memcpy(bases + nb, r->p_seq.data(), (size_t)sl);
        if (quals) {
            const int64_t ql = (int64_t)r->p_qual.size();
            memcpy(quals + nb, r->p_qual.data(), (size_t)(ql < sl ? ql : sl));
            if (ql < sl) memset(quals + nb + ql, '!', (size_t)(sl - ql));
        }
        memcpy(heads + nh, r->p_head.data(), (size_t)hl);
        nb += sl;
        nh += hl;
        ++n;
        offsets[n] = nb;
        head_off[n] = nh;
        r->have_pending = false;
    }
    return n;
}

// Packed mode for plain (uncompressed, mmap'ed) 4-line FASTQ: see Fast::next_packed.  Returns the number of records (0 at
// the end of the input), -1 on an error, -2 if one record does not fit into empty buffers, -4 if this input cannot be read
// in packed mode (compressed, FASTA, wrapped or odd records from here on): the caller continues with tps_reader_next at
// the same record and packs with tps_pack_reads.  *n_words receives the words used in seq2 / inv.
static int64_t reader_next_packed(void* hv, uint32_t* seq2, uint16_t* inv, int64_t words_cap, tps_read_desc* desc, int64_t max_records,
                                  char* heads, int64_t heads_cap, int64_t* head_off, int64_t* spans, int64_t* n_words, int32_t heads_bp,
                                  int32_t* full_len) {
    Handle* h = (Handle*)hv;
    if (!h || !seq2 || !desc || !heads || !head_off || !n_words) { g_err = "null argument"; return -1; }
    *n_words = 0;
    head_off[0] = 0;
    if (!h->format) return 0;
    if (!h->fast) return -4;
    Fast* f = h->fast;
    f->want_lines = false;
    f->first_window();
    if (f->src && f->src->failed) return -1;
    if (f->src) {
        f->top_up((size_t)std::max<int64_t>(words_cap, 1024) * 32);      // (a batch's worth of text in the window, if it can still grow)
        if (f->src->failed) return -1;
    }
    int64_t n = f->next_packed(seq2, inv, words_cap, desc, max_records, heads, heads_cap, head_off, spans, heads_bp, full_len);
    // compressed input: the window ends in an incomplete record (or is used up) while the source has more -- the next group
    // of blocks is inflated into a new window behind the unconsumed tail.  (An unconsumed stretch longer than any record that
    // yields nothing is not an incomplete record: the streaming decoder judges it.)
    while (f->src && (n == 0 || n == -3) && !f->src->eof() && !f->src->failed && f->size - f->pos < ((size_t)64 << 20) && f->pos < f->eff_limit()) {
        f->fill_target = (size_t)std::max<int64_t>(words_cap, 1024) * 32 + (f->size - f->pos);
        f->index_window();
        if (f->src->failed) return -1;
        n = f->next_packed(seq2, inv, words_cap, desc, max_records, heads, heads_cap, head_off, spans, heads_bp, full_len);
    }
    if (n == -3) return -4;                 // (the position is unchanged: tps_reader_next re-reads this record its own way)
    if (n > 0) *n_words = desc[n - 1].word_off + tps::packed_words(desc[n - 1].len);
    return n;
}

int64_t tps_reader_next_packed(void* hv, uint32_t* seq2, uint16_t* inv, int64_t words_cap, tps_read_desc* desc, int64_t max_records,
                               char* heads, int64_t heads_cap, int64_t* head_off, int64_t* spans, int64_t* n_words) {
    return reader_next_packed(hv, seq2, inv, words_cap, desc, max_records, heads, heads_cap, head_off, spans, n_words, 0, nullptr);
}
// Heads mode (see Fast::next_packed): reads longer than 2 heads_bp are packed as their first + last heads_bp bases; full_len[i]
// receives every read's own length.
int64_t tps_reader_next_heads(void* hv, int32_t heads_bp, uint32_t* seq2, uint16_t* inv, int64_t words_cap, tps_read_desc* desc,
                              int64_t max_records, char* heads, int64_t heads_cap, int64_t* head_off, int64_t* spans, int32_t* full_len,
                              int64_t* n_words) {
    if (heads_bp < 1 || !full_len) { g_err = "bad heads_bp / full_len"; return -1; }
    return reader_next_packed(hv, seq2, inv, words_cap, desc, max_records, heads, heads_cap, head_off, spans, n_words, heads_bp, full_len);
}

// Second pass of the heads mode: packs, for the n reads idx[0 .. n) of a batch whose spans point into `text`, the part of the
// read a scan of tail tails[j] (0 = forward, 1 = reverse) touches -- its first / last min(length, maxlen) bases, as a read of its
// own (allsteps.py:263-271: s = seq[t:min(L, M)], or the same of the reversed read).  fasta: spans[4 i + 3] is the end of the
// record's sequence text (a wrapped sequence is joined first).  seq2 / inv / desc as tps_pack_reads fills them; returns the words
// used, -2 if words_cap is too small, -1 on a bad span.
int64_t tps_pack_spans(const char* text, int64_t text_len, int32_t fasta, const int64_t* spans, const int32_t* full_len, const int64_t* idx,
                       const uint8_t* tails, int64_t n, int32_t maxlen, uint32_t* seq2, uint16_t* inv, tps_read_desc* desc, int64_t words_cap) {
    if (!text || !spans || !full_len || (n > 0 && (!idx || !tails)) || !seq2 || !desc || maxlen < 0) { g_err = "null argument"; return -1; }
    int64_t nw = 0;
    for (int64_t j = 0; j < n; ++j) {
        const int64_t L = full_len[idx[j]], m = std::min<int64_t>(L, maxlen);
        desc[j].word_off = nw;
        desc[j].len = (int32_t)m;
        desc[j].flags = 0;
        nw += tps::packed_words(m);
    }
    if (nw > words_cap) { g_err = "the passing reads do not fit the buffers"; return -2; }
    std::atomic<int> bad{0};
    const int T = nw * 16 < (4 << 20) ? 1 : io_threads();
    team(T, [&](int t, int nt) {
        std::vector<uint8_t> joined;
        const int64_t a = n * (int64_t)t / nt, b = n * (int64_t)(t + 1) / nt;
        for (int64_t j = a; j < b; ++j) {
            const int64_t i = idx[j], L = full_len[i], m = desc[j].len, s0 = spans[4 * i + 2];
            const uint8_t* src = (const uint8_t*)text + s0;
            if (s0 < 0 || s0 > text_len) { bad = 1; return; }
            if (fasta) {
                const int64_t se = spans[4 * i + 3];
                if (se < s0 || se > text_len) { bad = 1; return; }
                const size_t look = (size_t)std::min<int64_t>(L, se - s0);      // a sequence on one line has its L bases in front of the first line end
                if (se - s0 > L && (memchr(text + s0, '\n', look) || memchr(text + s0, '\r', look))) {
                    // wrapped: join the lines (line ends dropped, a trailing CR with them: what the reader packed)
                    joined.clear();
                    const char* p = text + s0;
                    const char* e = text + se;
                    while (p < e) {
                        const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
                        const char* le = q ? q : e;
                        const char* l1 = (le > p && le[-1] == '\r') ? le - 1 : le;
                        joined.insert(joined.end(), (const uint8_t*)p, (const uint8_t*)l1);
                        p = q ? q + 1 : e;
                    }
                    if ((int64_t)joined.size() != L) { bad = 1; return; }
                    src = joined.data();
                } else if (s0 + L > text_len) { bad = 1; return; }
            } else if (s0 + L > text_len) { bad = 1; return; }
            else if (memchr(text + s0, '\n', (size_t)L) || memchr(text + s0, '\r', (size_t)L)) {
                // a multi-line FASTQ record: the first L bytes from s0 on that are not line ends
                joined.clear();
                int64_t p = s0;
                while ((int64_t)joined.size() < L) {
                    if (p >= text_len) { bad = 1; return; }
                    const char* q = (const char*)memchr(text + p, '\n', (size_t)(text_len - p));
                    const int64_t le = q ? (int64_t)(q - text) : text_len;
                    const int64_t l1 = (le > p && text[le - 1] == '\r') ? le - 1 : le;
                    const int64_t take = std::min<int64_t>(L - (int64_t)joined.size(), l1 - p);
                    joined.insert(joined.end(), (const uint8_t*)text + p, (const uint8_t*)text + p + take);
                    p = le + 1;
                }
                src = joined.data();
            }
            if (tps::pack_one(src + (tails[j] & 1 ? L - m : 0), m, seq2 + desc[j].word_off, inv ? inv + desc[j].word_off : nullptr))
                desc[j].flags |= TPS_RD_HAS_INVALID;
        }
    });
    if (bad.load()) { g_err = "record span outside the text"; return -1; }
    return nw;
}

// The text the spans of the LAST packed batch point into, for a compressed input (for a plain file the caller maps the file
// itself and *hold stays NULL): *text / *len = the window, *hold = a reference the caller gives back with tps_text_release when
// it is done with the batch's records.
int tps_reader_text_hold(void* hv, const char** text, int64_t* len, void** hold) {
    Handle* h = (Handle*)hv;
    if (!h || !text || !len || !hold) { g_err = "null argument"; return -1; }
    *text = nullptr; *len = 0; *hold = nullptr;
    if (!h->fast || !h->fast->src || !h->fast->hold) return 0;
    h->fast->hold->ref();
    *text = h->fast->data;
    *len = (int64_t)h->fast->size;
    *hold = h->fast->hold;
    return 0;
}
void tps_text_release(void* hold) { if (hold) ((TextHold*)hold)->unref(); }

// Packs an ASCII batch (concatenated bases + n+1 offsets) into the layout of tps_pack.h with the reader's thread team.
// seq2 / inv must hold tps_packed_words_total(offsets, n) entries; inv may be NULL.  Returns the words written.
int64_t tps_packed_words_total(const int64_t* offsets, int64_t n) {
    int64_t w = 0;
    for (int64_t i = 0; i < n; ++i) w += tps::packed_words(offsets[i + 1] - offsets[i]);
    return w;
}
int64_t tps_pack_reads(const uint8_t* bases, const int64_t* offsets, int64_t n, uint32_t* seq2, uint16_t* inv, tps_read_desc* desc,
                       int32_t nthreads) {
    if (n < 0 || !offsets || !seq2 || !desc || (n > 0 && !bases)) { g_err = "null argument"; return -1; }
    for (int64_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 0x7FFFFFFFll) { g_err = "bad offsets"; return -1; }
    const int64_t nw = tps::pack_layout(offsets, n, desc);
    int T = nthreads > 0 ? nthreads : io_threads();
    if (offsets[n] < (4 << 20)) T = 1;
    team(T, [&](int t, int nt) {
        const int64_t wa = nw * (int64_t)t / nt, wb = nw * (int64_t)(t + 1) / nt;
        int64_t a = std::lower_bound(desc, desc + n, wa, [](const tps_read_desc& d, int64_t v) { return d.word_off < v; }) - desc;
        int64_t b = t + 1 == nt ? n : std::lower_bound(desc, desc + n, wb, [](const tps_read_desc& d, int64_t v) { return d.word_off < v; }) - desc;
        tps::pack_range(bases, offsets, a, b, desc, seq2, inv);
    });
    return nw;
}

// Test / diagnostics hook: the whole text of a gzip file through the parallel inflater (tps_gzpar.h).  Returns the number of
// bytes (written to out up to cap), -1 on a corrupt file; stats[0..2] = chunks tried, speculative chunks accepted, chunks redone
// from the known position.  want = bytes of text per round (0 = default), threads = 0 = the team's size.
int64_t tps_gz_inflate(const char* path, uint8_t* out, int64_t cap, int32_t threads, int64_t want, int64_t* stats) {
    int fd = open(path, O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) { g_err = "cannot open file"; if (fd >= 0) close(fd); return -1; }
    if (st.st_size == 0) { close(fd); return 0; }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { close(fd); g_err = "mmap failed"; return -1; }
    gzpar::ParGz z;
    z.data = (const uint8_t*)m;
    z.size = (size_t)st.st_size;
    z.threads = threads > 0 ? threads : io_threads();
    z.timing = io_timing();
    z.team = [](int n, const std::function<void(int, int)>& f) { team(n, f); };
    gzpar::TextBuf buf;
    int64_t total = 0;
    bool ok = true;
    while (ok && !z.eof()) {
        buf.clear();
        ok = z.read(buf, want > 0 ? (size_t)want : (size_t)64 << 20);
        if (!ok) break;
        const int64_t n = (int64_t)buf.size();
        if (out && n > 0 && total < cap) memcpy(out + total, buf.data(), (size_t)std::min<int64_t>(n, cap - total));
        total += n;
    }
    if (stats) { stats[0] = (int64_t)z.n_chunks; stats[1] = (int64_t)z.n_spec_ok; stats[2] = (int64_t)z.n_serial; }
    munmap(m, (size_t)st.st_size);
    close(fd);
    if (!ok) { g_err = "gzip: " + z.err; return -1; }
    return total;
}

// zlib's crc32(crc, p, n) by carry-less multiplication (tps_gzpar.h; tests compare it with zlib's).  Also the checksum callback of
// libtopsicle_hip.so's tps_batch_raw_to_fd: the raw-count archive's member CRC is computed block by block while the rows are
// written, and the blocks' values are joined with tps_crc32_combine (= zlib's crc32_combine) instead of re-reading the file.
uint32_t tps_crc32(uint32_t crc, const uint8_t* p, int64_t n) { return (uint32_t)gzpar::crc32_fast((uLong)crc, p, (size_t)(n > 0 ? n : 0)); }
uint32_t tps_crc32_combine(uint32_t crc1, uint32_t crc2, int64_t len2) { return (uint32_t)crc32_combine((uLong)crc1, (uLong)crc2, (z_off_t)len2); }

// Writes the records idx[0 .. n) of a packed batch that was read from the mmap'ed plain FASTQ `text` to `fd`, in the layout
// Biopython's SeqIO.write gives (main.py:83-86): "@" header "\n" sequence "\n+\n" quality "\n".  Nothing is copied in user
// space: the iovecs point into the mapping (spans: 4 entries per record -- header offset, header length, sequence offset,
// quality offset; lens: bases per record), a record whose text already has that layout is one iovec, and neighbouring
// records that are neighbours in the file merge into one.  Replaces the per-record Python loop of the round-2 writer
// (0.1 s per 300 MB, the largest part of the CLI's per-read time).  Returns the bytes written or -1.
static int64_t write_fastq_spans(int fd, int64_t file_off, const char* text, int64_t text_len, const int64_t* spans, const int32_t* lens, const int64_t* idx, int64_t n);
int64_t tps_write_fastq_spans(int fd, const char* text, int64_t text_len, const int64_t* spans, const int32_t* lens, const int64_t* idx, int64_t n) {
    return write_fastq_spans(fd, -1, text, text_len, spans, lens, idx, n);
}
// The same at an explicit file offset (pwritev: the descriptor's position is not used), so that several threads write different
// batches of one filtered file at the same time; a record always takes header + 2 x bases + 6 bytes there (tps_fastq_spans_bytes),
// whatever the input's line layout was.  BASELINE configs[3]'s shard rewrites 7.5 GB of passing records: one writev thread at
// 4.9 - 6.5 GB/s was what the run waited for (round 5).
int64_t tps_write_fastq_spans_at(int fd, int64_t file_off, const char* text, int64_t text_len, const int64_t* spans, const int32_t* lens,
                                 const int64_t* idx, int64_t n) {
    if (file_off < 0) { g_err = "negative file offset"; return -1; }
    return write_fastq_spans(fd, file_off, text, text_len, spans, lens, idx, n);
}
int64_t tps_fastq_spans_bytes(const int64_t* spans, const int32_t* lens, const int64_t* idx, int64_t n) {
    if (!spans || !lens || (n > 0 && !idx)) { g_err = "null argument"; return -1; }
    int64_t total = 0;
    for (int64_t j = 0; j < n; ++j) total += spans[4 * idx[j] + 1] + 2 * (int64_t)lens[idx[j]] + 6;     // "@" head "\n" seq "\n+\n" qual "\n"
    return total;
}
static int64_t write_fastq_spans(int fd, int64_t file_off, const char* text, int64_t text_len, const int64_t* spans, const int32_t* lens, const int64_t* idx, int64_t n) {
    if (fd < 0 || !text || !spans || !lens || (n > 0 && !idx)) { g_err = "null argument"; return -1; }
    static const char at = '@', nl = '\n', plus[3] = {'\n', '+', '\n'};
    std::vector<struct iovec> iov;
    iov.reserve(1024);
    int64_t total = 0;
    auto flush = [&]() -> bool {
        size_t first = 0;
        while (first < iov.size()) {
            const int cnt = (int)std::min<size_t>(iov.size() - first, 1024);
            ssize_t w = file_off < 0 ? writev(fd, iov.data() + first, cnt) : pwritev(fd, iov.data() + first, cnt, (off_t)(file_off + total));
            if (w < 0) { if (errno == EINTR) continue; g_err = std::string("writev: ") + strerror(errno); return false; }
            total += w;
            size_t left = (size_t)w;                               // partial writes: advance inside the vector
            while (first < iov.size() && left >= iov[first].iov_len) { left -= iov[first].iov_len; ++first; }
            if (left) { iov[first].iov_base = (char*)iov[first].iov_base + left; iov[first].iov_len -= left; }
        }
        iov.clear();
        return true;
    };
    auto push = [&](const char* p, size_t len) {
        if (!len) return;
        if (!iov.empty() && (const char*)iov.back().iov_base + iov.back().iov_len == p) { iov.back().iov_len += len; return; }
        iov.push_back({(void*)p, len});
    };
    for (int64_t j = 0; j < n; ++j) {
        const int64_t i = idx[j];
        const int64_t h0 = spans[4 * i], hl = spans[4 * i + 1], s0 = spans[4 * i + 2], q0 = spans[4 * i + 3], sl = lens[i];
        if (h0 < 1 || hl < 0 || s0 < 0 || q0 < 0 || sl < 0 || h0 + hl > text_len || s0 + sl > text_len || q0 + sl > text_len) { g_err = "record span outside the text"; return -1; }
        // the text itself is "@head\nseq\n+\nqual\n": one piece
        const bool verbatim = s0 == h0 + hl + 1 && q0 == s0 + sl + 3 && q0 + sl < text_len && text[h0 - 1] == '@' && text[h0 + hl] == '\n' &&
                              memcmp(text + s0 + sl, plus, 3) == 0 && text[q0 + sl] == '\n';
        if (verbatim) {
            push(text + h0 - 1, (size_t)(q0 + sl + 1 - (h0 - 1)));
        } else {
            // (a multi-line record: its sl bases / quality characters are the first sl bytes that are not line ends)
            auto push_joined = [&](int64_t from) -> bool {
                if (!memchr(text + from, '\n', (size_t)sl) && !memchr(text + from, '\r', (size_t)sl)) { push(text + from, (size_t)sl); return true; }
                int64_t p = from, need = sl;
                while (need > 0) {
                    if (p >= text_len) return false;
                    const char* q = (const char*)memchr(text + p, '\n', (size_t)(text_len - p));
                    const int64_t le = q ? (int64_t)(q - text) : text_len;
                    const int64_t l1 = (le > p && text[le - 1] == '\r') ? le - 1 : le;
                    const int64_t take = std::min<int64_t>(need, l1 - p);
                    push(text + p, (size_t)take);
                    need -= take;
                    p = le + 1;
                }
                return true;
            };
            push(&at, 1); push(text + h0, (size_t)hl); push(&nl, 1);
            if (!push_joined(s0)) { g_err = "record span outside the text"; return -1; }
            push(plus, 3);
            if (!push_joined(q0)) { g_err = "record span outside the text"; return -1; }
            push(&nl, 1);
        }
        if (iov.size() > 1000 && !flush()) return -1;
    }
    if (!flush()) return -1;
    return total;
}

}  // extern "C"
