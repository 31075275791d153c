// tps_gzpar.h -- parallel decompression of ORDINARY gzip (one long deflate stream per member), for libtopsicle_io.so.
//
// The reference reads .fastq.gz through Python's gzip (Topsicle/allsteps.py:127-149): one inflate stream, one core.  A
// deflate stream has no index, but it can still be inflated by many threads (the idea of pugz / rapidgzip, written here from
// the format, RFC 1951 / 1952):
//   1. cut the compressed bytes into chunks; chunk 0 starts at a known block boundary with a known 32 KiB window;
//   2. every other chunk SEARCHES a deflate block start behind its cut (a dynamic-Huffman header whose code-length code and
//      literal / distance codes are complete prefix codes) and inflates from there with an UNKNOWN window: output symbols are
//      16 bit, a back-reference that reaches into the 32 KiB before the chunk becomes a marker "byte k of the window";
//   3. chunks are stitched in order: chunk j is accepted iff it started exactly where chunk j - 1 stopped; its markers are
//      resolved from the last 32 KiB of everything before it (a chunk that was not found, or does not fit, is inflated again
//      from the known position -- slow, but only for that chunk);
//   4. CRC32 and ISIZE of the member are checked against the trailer (crc32_combine over the chunks).
// Everything that is not plain deflate (headers, trailers, CRC) uses zlib; the inflater below only exists because zlib's
// cannot run without its window.  No dependency besides zlib.
#pragma once
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <utility>
#include <string>
#include <vector>

namespace gzpar {

// std::vector that does not zero what resize() adds: the buffers below are written right after they grow, by the thread
// team -- a zeroing resize would touch (and page-fault) every byte once more, on one thread
template <class T>
struct NoInit : std::allocator<T> {
    template <class U> struct rebind { using other = NoInit<U>; };
    template <class U> void construct(U* p) noexcept { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
typedef std::vector<char, NoInit<char>> TextBuf;
typedef std::vector<uint16_t, NoInit<uint16_t>> SymBuf;
typedef std::vector<uint8_t, NoInit<uint8_t>> ByteBuf;

// CRC-32 (the gzip polynomial, reflected) by carry-less multiplication: four 128-bit lanes folded over 64 bytes per step, then
// 128 -> 64 -> 32 bits by Barrett reduction -- the scheme of Gopal et al., "Fast CRC Computation for Generic Polynomials Using
// PCLMULQDQ Instruction" (Intel, 2009), with that paper's constants for this polynomial (x^(4*128+32) mod P etc.).  zlib 1.2.11's
// table-driven crc32 runs at ~1 GB/s per core: behind a 16-thread inflate it was two thirds of a round's resolve phase.
// Checked against zlib's crc32 in tests/test_gzpar.py; zlib's is used for the tails and when the CPU has no PCLMULQDQ.
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) static inline uint32_t crc32_clmul_body(const uint8_t* buf, size_t len, uint32_t crc) {
    // len >= 64 and a multiple of 16; crc is the running register (already inverted)
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x1 = _mm_loadu_si128((const __m128i*)(buf + 0)), x2 = _mm_loadu_si128((const __m128i*)(buf + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i*)(buf + 32)), x4 = _mm_loadu_si128((const __m128i*)(buf + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    buf += 64;
    len -= 64;
    while (len >= 64) {
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        const __m128i a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
        x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
        x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128((const __m128i*)(buf + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128((const __m128i*)(buf + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128((const __m128i*)(buf + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128((const __m128i*)(buf + 48)));
        buf += 64;
        len -= 64;
    }
    // (a macro: a lambda would not inherit this function's target attribute)
#define GZ_FOLD(acc, next) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(acc, k3k4, 0x11), _mm_clmulepi64_si128(acc, k3k4, 0x00)), next)
    x1 = GZ_FOLD(x1, x2);
    x1 = GZ_FOLD(x1, x3);
    x1 = GZ_FOLD(x1, x4);
    while (len >= 16) {
        const __m128i nx = _mm_loadu_si128((const __m128i*)buf);
        x1 = GZ_FOLD(x1, nx);
        buf += 16;
        len -= 16;
    }
#undef GZ_FOLD
    // 128 -> 64 bits
    const __m128i m32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, m32);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, k5, 0x00), t);
    // Barrett reduction to 32 bits
    t = _mm_and_si128(x1, m32);
    t = _mm_clmulepi64_si128(t, poly, 0x10);
    t = _mm_and_si128(t, m32);
    t = _mm_clmulepi64_si128(t, poly, 0x00);
    x1 = _mm_xor_si128(x1, t);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif
// crc32(crc, p, n) as zlib defines it
inline uLong crc32_fast(uLong crc, const uint8_t* p, size_t n) {
#if defined(__x86_64__)
    static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (have && n >= 256) {
        const size_t body = n & ~(size_t)15;
        crc = (uLong)(~crc32_clmul_body(p, body, ~(uint32_t)crc) & 0xFFFFFFFFu);
        p += body;
        n -= body;
    }
#endif
    while (n) {                                       // (zlib's crc32 takes 32-bit lengths)
        const uInt m = (uInt)std::min<size_t>(n, (size_t)1 << 30);
        crc = crc32(crc, p, m);
        p += m;
        n -= m;
    }
    return crc;
}

constexpr int WSIZE = 32768;
constexpr uint16_t MARK = 0x8000;              // symbol >= MARK: byte (symbol - MARK) of the 32 KiB window before the chunk

struct Bits {
    const uint8_t* base;
    const uint8_t* p;
    const uint8_t* end;
    uint64_t buf = 0;
    int cnt = 0;
    bool over = false;                         // ran past the end of the input
    Bits(const uint8_t* b, size_t nbytes, uint64_t bitpos) : base(b), p(b + (bitpos >> 3)), end(b + nbytes) {
        refill();
        const int skip = (int)(bitpos & 7);
        buf >>= skip;
        cnt -= skip;
        if (cnt < 0) { cnt = 0; over = true; }
    }
    inline void refill() {
        if (p + 8 <= end) {                        // eight bytes at once: as many whole bytes as fit above the cnt bits in the buffer
            uint64_t w;
            memcpy(&w, p, 8);
            buf |= w << cnt;
            const int adv = (63 - cnt) >> 3;
            p += adv;
            cnt += adv * 8;
            return;
        }
        while (cnt <= 56 && p < end) { buf |= (uint64_t)*p++ << cnt; cnt += 8; }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1ull)); }
    inline void drop(int n) {
        if (n > cnt) { over = true; buf = 0; cnt = 0; return; }
        buf >>= n;
        cnt -= n;
    }
    inline uint32_t get(int n) {
        if (cnt < n) refill();
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    uint64_t bitpos() const { return (uint64_t)(p - base) * 8u - (uint64_t)cnt; }
    void align_byte() { drop(cnt & 7); }
};

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

// Entries of the FAST tables the symbol loop decodes from (Huff::fast, same indexing as Huff::tab): everything a symbol
// needs in one word -- bits 0-4 codeword length, bits 8-11 number of extra bits, bits 16-31 the literal / the length base /
// the distance base; bit 15 literal; bit 14 anything else (bit 13 pointer to a secondary table: bits 8-11 its index bits,
// bits 16-31 its offset; bit 12 end of block; neither: not a code)
enum : uint32_t { FE_LIT = 0x8000u, FE_EXC = 0x4000u, FE_SUB = 0x2000u, FE_EOB = 0x1000u };

// Canonical Huffman decoding table: primary table of 2^PB entries, secondary tables for longer codes.
// entry: bits 0-15 symbol (or secondary table offset), bits 16-19 code length (or secondary index bits), bit 31 = secondary
struct Huff {
    std::vector<uint32_t> tab, fast;
    int pb = 0;
    bool complete = false, empty = true;
    static uint32_t rev(uint32_t c, int n) {
        uint32_t r = 0;
        for (int i = 0; i < n; ++i) { r = (r << 1) | (c & 1u); c >>= 1; }
        return r;
    }
    // false: over-subscribed (not a prefix code).  `complete` tells whether every bit string decodes.
    bool build(const uint8_t* len, int n, int primary_bits) {
        int count[16] = {0};
        for (int i = 0; i < n; ++i) ++count[len[i]];
        empty = count[0] == n;
        int left = 1, maxlen = 0;
        for (int l = 1; l <= 15; ++l) {
            left <<= 1;
            left -= count[l];
            if (left < 0) return false;
            if (count[l]) maxlen = l;
        }
        complete = left == 0;
        pb = std::min(primary_bits, std::max(maxlen, 1));
        uint32_t next[16];
        uint32_t code = 0;
        count[0] = 0;
        for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
        tab.assign((size_t)1 << pb, 0u);                          // 0 = invalid
        // secondary tables: per primary index the longest code that shares it
        std::vector<uint8_t> sub_bits;
        if (maxlen > pb) {
            sub_bits.assign((size_t)1 << pb, 0);
            uint32_t nx[16];
            memcpy(nx, next, sizeof nx);
            for (int i = 0; i < n; ++i) {
                const int l = len[i];
                if (l > pb) {
                    const uint32_t r = rev(nx[l], l) & ((1u << pb) - 1u);
                    sub_bits[r] = (uint8_t)std::max<int>(sub_bits[r], l - pb);
                }
                if (l) ++nx[l];
            }
            for (size_t r = 0; r < sub_bits.size(); ++r)
                if (sub_bits[r]) {
                    tab[r] = 0x80000000u | ((uint32_t)sub_bits[r] << 16) | (uint32_t)tab.size();
                    if (tab.size() > 0xFFFFu) return false;
                    tab.resize(tab.size() + ((size_t)1 << sub_bits[r]), 0u);
                }
        }
        for (int i = 0; i < n; ++i) {
            const int l = len[i];
            if (!l) continue;
            const uint32_t r = rev(next[l]++, l);
            if (l <= pb) {
                const uint32_t e = ((uint32_t)l << 16) | (uint32_t)i;
                for (uint32_t k = r; k < (1u << pb); k += 1u << l) tab[k] = e;
            } else {
                const uint32_t prim = tab[r & ((1u << pb) - 1u)];
                const int sb = (int)((prim >> 16) & 15u);
                const uint32_t off = prim & 0xFFFFu, hi = r >> pb;
                const uint32_t e = ((uint32_t)l << 16) | (uint32_t)i;
                for (uint32_t k = hi; k < (1u << sb); k += 1u << (l - pb)) tab[off + k] = e;
            }
        }
        return true;
    }
    void make_fast(bool is_dist) {
        fast.resize(tab.size());
        for (size_t i = 0; i < tab.size(); ++i) {
            const uint32_t e = tab[i];
            uint32_t f = FE_EXC;                                   // not a code
            if (e & 0x80000000u) {
                f = FE_EXC | FE_SUB | (((e >> 16) & 15u) << 8) | ((e & 0xFFFFu) << 16);
            } else if (const uint32_t l = (e >> 16) & 15u) {
                const uint32_t sym = e & 0xFFFFu;
                if (is_dist) {
                    if (sym < 30) f = l | ((uint32_t)DIST_EXTRA[sym] << 8) | ((uint32_t)DIST_BASE[sym] << 16);
                } else if (sym < 256) {
                    f = FE_LIT | l | (sym << 16);
                } else if (sym == 256) {
                    f = FE_EXC | FE_EOB | l;
                } else if (sym - 257 < 29) {
                    f = l | ((uint32_t)LEN_EXTRA[sym - 257] << 8) | ((uint32_t)LEN_BASE[sym - 257] << 16);
                }
            }
            fast[i] = f;
        }
    }
    // decoded symbol, or -1 (invalid code)
    inline int decode(Bits& b) const {
        if (b.cnt < 15) b.refill();
        return decode_nofill(b);
    }
    // ... when the caller has made sure the buffer holds 15 bits (or the input has ended)
    inline int decode_nofill(Bits& b) const {
        uint32_t e = tab[b.peek(pb)];
        if (e & 0x80000000u) {
            const int sb = (int)((e >> 16) & 15u);
            e = tab[(e & 0xFFFFu) + ((b.peek(pb + sb)) >> pb)];
        }
        const int l = (int)((e >> 16) & 15u);
        if (!l) return -1;
        b.drop(l);
        return (int)(e & 0xFFFFu);
    }
};

static const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Header of a dynamic block (the 3 header bits already consumed).  `strict`: what a block-start SEARCH accepts -- complete
// codes only; a real stream may also hold the incomplete one-code distance tree zlib itself writes.
inline bool read_dynamic(Bits& b, Huff& lit, Huff& dist, bool strict) {
    const int hlit = (int)b.get(5) + 257, hdist = (int)b.get(5) + 1, hclen = (int)b.get(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; ++i) cl[CL_ORDER[i]] = (uint8_t)b.get(3);
    if (b.over) return false;
    Huff clh;
    if (!clh.build(cl, 19, 7) || clh.empty) return false;
    if (!clh.complete && strict) return false;
    uint8_t len[286 + 30] = {0};
    int i = 0;
    while (i < hlit + hdist) {
        const int s = clh.decode(b);
        if (s < 0 || b.over) return false;
        if (s < 16) { len[i++] = (uint8_t)s; continue; }
        int rep, val = 0;
        if (s == 16) {
            if (i == 0) return false;
            val = len[i - 1];
            rep = 3 + (int)b.get(2);
        } else if (s == 17) {
            rep = 3 + (int)b.get(3);
        } else {
            rep = 11 + (int)b.get(7);
        }
        if (i + rep > hlit + hdist) return false;
        while (rep--) len[i++] = (uint8_t)val;
    }
    if (len[256] == 0) return false;                              // no end-of-block code
    if (!lit.build(len, hlit, 11) || !dist.build(len + hlit, hdist, 9)) return false;
    if (strict) {
        if (!lit.complete) return false;
        int nd = 0;
        for (int k = 0; k < hdist; ++k) nd += len[hlit + k] != 0;
        if (!dist.complete && nd > 1) return false;
    }
    return !b.over;
}

inline void fixed_codes(Huff& lit, Huff& dist) {
    uint8_t l[288], d[30];
    for (int i = 0; i < 144; ++i) l[i] = 8;
    for (int i = 144; i < 256; ++i) l[i] = 9;
    for (int i = 256; i < 280; ++i) l[i] = 7;
    for (int i = 280; i < 288; ++i) l[i] = 8;
    for (int i = 0; i < 30; ++i) d[i] = 5;
    lit.build(l, 288, 11);
    dist.build(d, 30, 9);
}

// Inflates whole blocks from bit position `start` until a block ends at or behind `stop_bit` (or the final block ends).
// SPEC = false: `win` holds the known history (up to 32 KiB) and bytes come out; SPEC = true: the history is unknown and
// 16-bit symbols come out (references into it become markers).  Returns 0 = stopped at a block boundary (`end_bit`),
// 1 = the member's final block ended (`end_bit` behind it), -1 = invalid stream / out of input.
template <bool SPEC, typename V>
int inflate_blocks(const uint8_t* data, size_t nbytes, uint64_t start, uint64_t stop_bit, const uint8_t* win, size_t win_len,
                   V& out, uint64_t& end_bit, size_t max_out) {
    typedef typename V::value_type T;
    Bits b(data, nbytes, start);
    Huff lit, dist;
    // `out` is used as a buffer larger than what is in use (n elements): the symbol loop writes through a raw pointer with one
    // capacity test per symbol; every exit trims it to n
    size_t n = out.size();
    out.resize(std::max<size_t>(std::min<size_t>(out.capacity(), max_out + ((size_t)4 << 20)), n + (1u << 20)));
    T* o = out.data();
    size_t cap = out.size();
    // false: the chunk would outgrow max_out -- a speculative start that decodes garbage (or a deflate bomb) must not be able to
    // ask for memory without end: growth is checked HERE, not only where a block ends
    auto need = [&](size_t extra) -> bool {
        if (n + extra > cap) {
            if (n + extra > max_out + ((size_t)2 << 20)) return false;
            out.resize(std::min(std::max(cap * 2, n + extra + (1u << 20)), max_out + ((size_t)4 << 20)));
            o = out.data();
            cap = out.size();
        }
        return true;
    };
    struct Trim { V& v; size_t& n; ~Trim() { v.resize(n); } } trim{out, n};
    for (;;) {
        if (b.cnt < 3) b.refill();
        if (b.cnt < 3) return -1;
        const uint32_t hdr = b.get(3);
        const bool fin = hdr & 1u;
        const uint32_t type = hdr >> 1;
        if (type == 3) return -1;
        if (type == 0) {
            b.align_byte();
            if (b.cnt < 32) b.refill();
            if (b.cnt < 32) return -1;
            const uint32_t ln = b.get(16), nln = b.get(16);
            if ((ln ^ 0xFFFFu) != nln) return -1;
            if (!need((size_t)ln + 8)) return -1;
            // the rest of the bit buffer is whole bytes; then straight from the input
            size_t left = ln;
            while (left && b.cnt >= 8) { o[n++] = (T)b.get(8); --left; }
            if (left) {
                if ((size_t)(b.end - b.p) < left) return -1;
                for (size_t i = 0; i < left; ++i) o[n + i] = (T)b.p[i];
                n += left;
                b.p += left;
                b.buf = 0;                                        // (cnt is 0 here: nothing of the old position may stay in the buffer)
                b.cnt = 0;
            }
        } else {
            if (type == 1) fixed_codes(lit, dist);
            else if (!read_dynamic(b, lit, dist, false)) return -1;
            lit.make_fast(false);
            dist.make_fast(true);
            // FAST loop, while 16 input bytes and 1 KB of output space are at hand: one 8-byte refill covers up to three
            // literals or a whole length / distance pair, every symbol is one table word, matches are copied in 8-byte words
            bool eob = false;
            {
                const uint32_t* LT = lit.fast.data();
                const uint32_t* DT = dist.fast.data();
                const int lpb = lit.pb, dpb = dist.pb;
                const uint64_t lmask = (1ull << lpb) - 1ull, dmask = (1ull << dpb) - 1ull;
                uint64_t buf = b.buf;
                int cnt = b.cnt;
                const uint8_t* ip = b.p;
                // (up to three 8-byte refills behind one test of the input position; inputs below 32 bytes: slow loop only)
                const uint8_t* const ifast = (b.end - b.base) >= 32 ? b.end - 32 : b.base - 0;
                const bool fast_ok = (b.end - b.base) >= 32;
                constexpr int64_t EW = 8 / (int64_t)sizeof(T);        // elements per 8-byte word
                int bad = 0;
#define GZ_REFILL() do { uint64_t w_; memcpy(&w_, ip, 8); buf |= w_ << cnt; ip += (63 - cnt) >> 3; cnt |= 56; } while (0)
#define GZ_TAKE(nb) do { buf >>= (nb); cnt -= (int)(nb); } while (0)
                // the entry of the NEXT symbol is looked up as soon as the bits are there -- before the match in hand is copied --
                // so the table load's latency runs beside the copy instead of in front of every symbol
                uint32_t e = 0;
                if (fast_ok && ip <= ifast) { GZ_REFILL(); e = LT[buf & lmask]; }
                while (fast_ok && ip <= ifast && !b.over) {
                    if (n + 1024 > cap && !need(1024)) { bad = 1; break; }
                    if (e & FE_LIT) {
                        GZ_TAKE(e & 31u); o[n++] = (T)(e >> 16);
                        e = LT[buf & lmask];
                        if (e & FE_LIT) {
                            GZ_TAKE(e & 31u); o[n++] = (T)(e >> 16);
                            e = LT[buf & lmask];
                            if (e & FE_LIT) { GZ_TAKE(e & 31u); o[n++] = (T)(e >> 16); GZ_REFILL(); e = LT[buf & lmask]; continue; }
                        }
                        GZ_REFILL();                               // (the entry in hand stays valid: the low bits did not change)
                    }
                    if (e & FE_EXC) {
                        if (e & FE_SUB) {
                            e = LT[(e >> 16) + (uint32_t)((buf >> lpb) & ((1ull << ((e >> 8) & 15u)) - 1ull))];
                            if (e & FE_LIT) { GZ_TAKE(e & 31u); o[n++] = (T)(e >> 16); GZ_REFILL(); e = LT[buf & lmask]; continue; }
                        }
                        if (e & FE_EXC) {
                            if (e & FE_EOB) { GZ_TAKE(e & 31u); eob = true; break; }
                            bad = 1;
                            break;
                        }
                    }
                    const uint32_t lx = (e >> 8) & 15u, ll = e & 31u;
                    const int ml = (int)((e >> 16) + (uint32_t)((buf >> ll) & ((1ull << lx) - 1ull)));
                    GZ_TAKE(ll + lx);                              // <= 20 bits: >= 36 are left, a distance takes <= 28
                    uint32_t f = DT[buf & dmask];
                    if (f & FE_EXC) {
                        if (!(f & FE_SUB)) { bad = 1; break; }
                        f = DT[(f >> 16) + (uint32_t)((buf >> dpb) & ((1ull << ((f >> 8) & 15u)) - 1ull))];
                        if (f & FE_EXC) { bad = 1; break; }
                    }
                    const uint32_t dx = (f >> 8) & 15u, dl = f & 31u;
                    const int64_t d = (int64_t)((f >> 16) + (uint32_t)((buf >> dl) & ((1ull << dx) - 1ull)));
                    GZ_TAKE(dl + dx);
                    GZ_REFILL();
                    e = LT[buf & lmask];                           // (the next symbol's entry: in flight during the copy)
                    const int64_t at = (int64_t)n;
                    if (d <= at) {
                        T* dst = o + at;
                        const T* src = dst - d;
                        if (d >= EW) {                             // whole words; up to 7 bytes past the match, overwritten by what follows
                            T* const dend = dst + ml;
                            do { memcpy(dst, src, 8); dst += EW; src += EW; } while (dst < dend);
                        } else {
                            // short period (runs: a quality line of one letter is a chain of 258-byte matches at distance 1):
                            // element by element up to a multiple of the period that is at least a word, whole words from there
                            const int64_t dd = d * ((EW + d - 1) / d);
                            int i = 0;
                            for (; i < ml && i < dd; ++i) dst[i] = src[i];
                            for (; i < ml; i += (int)EW) memcpy(dst + i, dst + i - dd, 8);
                        }
                    } else {
                        if (SPEC ? (d - at > WSIZE) : (d - at > (int64_t)win_len)) { bad = 1; break; }
                        for (int i = 0; i < ml; ++i) {
                            const int64_t src = at + i - d;
                            if (src >= 0) o[at + i] = o[src];
                            else if (SPEC) o[at + i] = (T)(MARK + (uint16_t)(WSIZE + src));
                            else o[at + i] = (T)win[(int64_t)win_len + src];
                        }
                    }
                    n += (size_t)ml;
                }
#undef GZ_REFILL
#undef GZ_TAKE
                if (bad) return -1;
                // hand the position back: whole bytes only above `cnt` bits may stay in the buffer
                if (cnt < 0) return -1;
                b.buf = cnt < 64 ? buf & ((1ull << cnt) - 1ull) : buf;
                b.cnt = cnt;
                b.p = ip;
            }
            for (; !eob;) {
                if (!need(300)) return -1;
                if (b.cnt < 48) b.refill();                       // one refill per symbol: 15 + 5 + 15 + 13 bits at most
                int s = lit.decode_nofill(b);
                if (s < 256) {
                    if (s < 0) return -1;
                    o[n++] = (T)s;
                    continue;
                }
                if (s == 256) break;
                s -= 257;
                if (s >= 29) return -1;
                const int ml = LEN_BASE[s] + (int)b.peek(LEN_EXTRA[s]);
                b.drop(LEN_EXTRA[s]);
                const int ds = dist.decode_nofill(b);
                if (ds < 0 || ds >= 30) return -1;
                const int64_t d = DIST_BASE[ds] + (int64_t)b.peek(DIST_EXTRA[ds]);
                b.drop(DIST_EXTRA[ds]);
                if (b.over) return -1;
                const int64_t at = (int64_t)n;
                if (d <= at) {
                    const T* src = o + (at - d);
                    T* dst = o + at;
                    if (d >= ml) memcpy(dst, src, (size_t)ml * sizeof(T));
                    else for (int i = 0; i < ml; ++i) dst[i] = src[i];           // overlapping: element by element, as deflate defines it
                } else {
                    // reaches into the history before this chunk
                    if (SPEC) {
                        if (d - at > WSIZE) return -1;
                    } else if (d - at > (int64_t)win_len) {
                        return -1;
                    }
                    for (int i = 0; i < ml; ++i) {
                        const int64_t src = at + i - d;
                        if (src >= 0) o[at + i] = o[src];
                        else if (SPEC) o[at + i] = (T)(MARK + (uint16_t)(WSIZE + src));          // byte WSIZE + src of the window
                        else o[at + i] = (T)win[(int64_t)win_len + src];
                    }
                }
                n += (size_t)ml;
            }
            if (n > max_out) return -1;
        }
        if (b.over) return -1;
        end_bit = b.bitpos();
        if (fin) return 1;
        if (end_bit >= stop_bit) return 0;
    }
}

// First bit position >= from (and < limit) at which a plausible non-final dynamic block starts, or ~0.
inline uint64_t find_block(const uint8_t* data, size_t nbytes, uint64_t from, uint64_t limit) {
    Huff lit, dist;
    for (uint64_t pos = from; pos < limit; ++pos) {
        // BFINAL = 0, BTYPE = 2: the three bits are 0, 0, 1 (LSB first) -- checked on the raw bytes before any set-up
        const size_t byte = (size_t)(pos >> 3);
        if (byte + 8 > nbytes) break;
        uint32_t w;
        memcpy(&w, data + byte, 4);
        w >>= (pos & 7);
        if ((w & 7u) != 4u) continue;
        if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;      // HLIT, HDIST
        // the code-length code must be a complete prefix code (what read_dynamic(strict) asks first): its Kraft sum straight from
        // the header's 3-bit lengths -- 99 % of the candidates end here, without a table being built
        {
            const uint32_t hclen = ((w >> 13) & 15u) + 4u;
            uint64_t q = 0;
            const uint64_t at = pos + 17;                                   // first 3-bit length
            if ((at >> 3) + 9 <= nbytes) {
                memcpy(&q, data + (at >> 3), 8);
                q >>= (at & 7);                                              // >= 57 bits: 19 lengths
                uint32_t kraft = 0;
                for (uint32_t j = 0; j < hclen; ++j) {
                    const uint32_t l = (uint32_t)(q >> (3 * j)) & 7u;
                    if (l) kraft += 128u >> l;
                }
                if (kraft != 128u) continue;
            }
        }
        Bits b(data, nbytes, pos + 3);
        if (read_dynamic(b, lit, dist, true)) return pos;
    }
    return ~0ull;
}

struct Member {                                  // where a gzip member's deflate data starts
    size_t data_off = 0;
};
// parses a gzip member header at `off`; false if there is none
inline bool parse_header(const uint8_t* d, size_t n, size_t off, size_t& data_off) {
    if (off + 18 > n || d[off] != 0x1f || d[off + 1] != 0x8b || d[off + 2] != 8) return false;
    const uint8_t flg = d[off + 3];
    size_t p = off + 10;
    if (flg & 4) {
        if (p + 2 > n) return false;
        p += 2 + ((size_t)d[p] | ((size_t)d[p + 1] << 8));
    }
    if (flg & 8) { while (p < n && d[p]) ++p; ++p; }
    if (flg & 16) { while (p < n && d[p]) ++p; ++p; }
    if (flg & 2) p += 2;
    if (p >= n) return false;
    data_off = p;
    return true;
}

// Streaming parallel inflater over a memory-mapped gzip file.  read(out, want) appends roughly `want` bytes of text (whole
// deflate blocks) and returns false on a corrupt stream (`err` says why); eof() once every member has been read.
struct ParGz {
    const uint8_t* data = nullptr;
    size_t size = 0;
    int threads = 1;
    std::function<void(int, const std::function<void(int, int)>&)> team;      // team(n, f(thread, n))
    std::string err;
    // position
    bool in_member = false, done = false;
    uint64_t bit = 0;                            // next block boundary (bit offset in the file)
    std::vector<uint8_t> win;                    // last <= 32 KiB of the member's text so far
    uLong crc = 0;
    uint64_t isize = 0;
    double ratio = 3.5;                          // text bytes per compressed byte, measured as we go
    bool measured = false;
    size_t next_member = 0;
    // statistics (tests, diagnostics)
    uint64_t n_chunks = 0, n_spec_ok = 0, n_serial = 0, n_gap = 0;
    bool timing = false;                         // phase times on stderr (tps_io_set_option "timing")
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

    bool eof() const { return done; }

    bool start_member() {
        size_t doff = 0;
        if (next_member >= size) { done = true; return true; }
        if (!parse_header(data, size, next_member, doff)) {
            // trailing garbage / padding behind the last member: gzip tools ignore zeros
            bool zeros = true;
            for (size_t i = next_member; i < size; ++i) zeros = zeros && data[i] == 0;
            if (zeros) { done = true; return true; }
            err = "not a gzip member header";
            return false;
        }
        bit = (uint64_t)doff * 8u;
        win.clear();
        crc = crc32(0L, Z_NULL, 0);
        isize = 0;
        in_member = true;
        return true;
    }
    bool finish_member(uint64_t end_bit) {
        size_t p = (size_t)((end_bit + 7) >> 3);
        if (p + 8 > size) { err = "gzip trailer missing"; return false; }
        const uint32_t c = (uint32_t)data[p] | ((uint32_t)data[p + 1] << 8) | ((uint32_t)data[p + 2] << 16) | ((uint32_t)data[p + 3] << 24);
        const uint32_t n = (uint32_t)data[p + 4] | ((uint32_t)data[p + 5] << 8) | ((uint32_t)data[p + 6] << 16) | ((uint32_t)data[p + 7] << 24);
        if (c != (uint32_t)crc || n != (uint32_t)isize) { err = "gzip CRC / length mismatch"; return false; }
        next_member = p + 8;
        in_member = false;
        return true;
    }
    void push_window(const uint8_t* p, size_t n) {
        if (n >= (size_t)WSIZE) { win.assign(p + n - WSIZE, p + n); return; }
        if (win.size() + n > (size_t)WSIZE) win.erase(win.begin(), win.begin() + (ptrdiff_t)(win.size() + n - WSIZE));
        win.insert(win.end(), p, p + n);
    }

    struct Chunk {
        uint64_t cut = 0, start = ~0ull, end = 0;
        int rc = -1;
        SymBuf sym;                              // speculative output
        ByteBuf bytes;                           // chunk 0 / serial redo: plain bytes (a gap in front of `sym` when both are set)
        std::vector<uint8_t> last;               // its resolved last <= 32 KiB
        uLong crc = 0;
        size_t out_off = 0, out_len = 0;
    };

    std::vector<Chunk> ch;                       // kept from round to round: their buffers are grown (and page-faulted) once
    // ... and from file to file: a few hundred MB of symbol buffers cost milliseconds to map, fault in and unmap again (more
    // when the process holds a GPU context: the driver's MMU notifier sees every unmap), so the last file's set is parked here
    static std::mutex& spare_mu() { static std::mutex m; return m; }
    static std::vector<std::vector<Chunk>>& spare() { static std::vector<std::vector<Chunk>> v; return v; }
    bool took_spare = false;
    void take_spare() {
        took_spare = true;
        std::lock_guard<std::mutex> lk(spare_mu());
        if (!spare().empty()) { ch.swap(spare().back()); spare().pop_back(); }
    }
    ~ParGz() {
        if (ch.empty()) return;
        std::lock_guard<std::mutex> lk(spare_mu());
        if (spare().size() < 2) { spare().emplace_back(); spare().back().swap(ch); }
    }

    bool read(TextBuf& out, size_t want) {
        if (done) return true;
        if (!in_member && !start_member()) return false;
        if (done) return true;
        const int T = std::max(1, threads);
        if (!took_spare) take_spare();
        // compressed bytes per chunk: the round should yield about `want` bytes of text
        size_t cs = (size_t)((double)want / ratio / (double)T);
        cs = std::min<size_t>(std::max<size_t>(cs, (size_t)256 << 10), (size_t)8 << 20);
        const size_t first = (size_t)(bit >> 3);
        // the end of the file is shared out evenly: no last round of two long chunks with the other threads idle
        if (first < size && size - first < cs * (size_t)T) cs = std::max<size_t>((size - first + (size_t)T - 1) / (size_t)T, (size_t)256 << 10);
        size_t nc = 0;
        for (int j = 0; j < T; ++j) {
            const size_t cut = first + (size_t)j * cs;
            if (j && cut + 64 >= size) break;
            if (ch.size() <= nc) ch.emplace_back();
            Chunk& c = ch[nc++];
            c.cut = (uint64_t)cut * 8u;
            c.start = ~0ull; c.end = 0; c.rc = -1;
            c.sym.clear(); c.bytes.clear(); c.last.clear();
            c.crc = 0; c.out_off = c.out_len = 0;
        }
        const uint64_t file_bits = (uint64_t)size * 8u;
        const size_t max_out = cs * 1100 + (1u << 20);            // deflate cannot expand more than 1032 x
        // ... but a SPECULATIVE chunk (16-bit symbols, possibly a false start decoding garbage) is not allowed that much memory:
        // beyond a few times what the round expects of it, it counts as not found and is inflated again from the known position
        const size_t spec_max = std::min(max_out, std::max<size_t>((size_t)32 << 20, 8 * (want / (size_t)T + 1)));
        std::vector<double> t_chunk(timing ? nc : 0, 0.0), t_find(timing ? nc : 0, 0.0);
        auto work = [&](int t, int nt) {
            for (size_t j = (size_t)t; j < nc; j += (size_t)nt) {
                Chunk& c = ch[j];
                const double tc0 = timing ? now() : 0.0;
                struct Fin { std::vector<double>& v; size_t j; double t0; bool on; ~Fin() { if (on) v[j] = now() - t0; } } fin{t_chunk, j, tc0, timing};
                const uint64_t stop = j + 1 < nc ? ch[j + 1].cut : std::min<uint64_t>(file_bits, c.cut + (uint64_t)cs * 8u);
                if (j == 0) {
                    c.start = bit;
                    c.bytes.reserve((size_t)((double)cs * ratio * 1.3) + 4096);
                    c.rc = inflate_blocks<false>(data, size, bit, stop, win.data(), win.size(), c.bytes, c.end, max_out);
                } else {
                    c.start = find_block(data, size, c.cut, stop);
                    if (timing) t_find[j] = now() - tc0;
                    if (c.start == ~0ull) { c.rc = -1; continue; }
                    c.sym.reserve((size_t)((double)cs * ratio * 1.3) + 4096);
                    c.rc = inflate_blocks<true>(data, size, c.start, stop, nullptr, 0, c.sym, c.end, spec_max);
                }
            }
        };
        const double t0 = timing ? now() : 0.0;
        if (team && nc > 1) team((int)std::min<size_t>(nc, (size_t)T), work);
        else work(0, 1);
        const double t1 = timing ? now() : 0.0;
        n_chunks += nc;
        if (ch[0].rc < 0) { err = "invalid deflate data"; return false; }
        // stitch: accept chunk j iff it begins where its predecessor ended; windows are propagated through the (resolved)
        // last 32 KiB of every chunk
        std::vector<uint8_t> cur_win = win;
        auto append_window = [](std::vector<uint8_t>& w, const uint8_t* p, size_t n) {
            if (n >= (size_t)WSIZE) { w.assign(p + n - WSIZE, p + n); return; }
            if (w.size() + n > (size_t)WSIZE) w.erase(w.begin(), w.begin() + (ptrdiff_t)(w.size() + n - WSIZE));
            w.insert(w.end(), p, p + n);
        };
        // `mid[j]`: the history in front of chunk j's SPECULATIVE part (= history before the chunk + its gap bytes)
        std::vector<std::vector<uint8_t>> mid(nc);
        auto take_last = [&](Chunk& c, const std::vector<uint8_t>& before, std::vector<uint8_t>& mid_w) {
            // last <= 32 KiB of (before + the chunk's plain bytes + the chunk's resolved symbols)
            mid_w = before;
            if (!c.bytes.empty()) append_window(mid_w, c.bytes.data(), c.bytes.size());
            c.last = mid_w;
            if (c.sym.empty()) return;
            const size_t n = c.sym.size(), k = std::min<size_t>(n, (size_t)WSIZE);
            std::vector<uint8_t> tail(k);
            // markers index the 32 KiB window that ends where the speculative part begins; the known history may be shorter
            // than that at the start of a member -- a marker below its start would be a reference before the stream
            const int64_t shift = (int64_t)WSIZE - (int64_t)mid_w.size();
            for (size_t i = 0; i < k; ++i) {
                const uint16_t s = c.sym[n - k + i];
                const int64_t idx = (int64_t)(s - MARK) - shift;
                tail[i] = s < MARK ? (uint8_t)s : (uint8_t)(idx >= 0 ? mid_w[(size_t)idx] : 0);
            }
            append_window(c.last, tail.data(), k);
        };
        size_t used = 0;                        // chunks accepted
        std::vector<std::vector<uint8_t>> before(nc);
        uint64_t pos = 0;
        bool fin = false;
        for (size_t j = 0; j < nc; ++j) {
            Chunk& c = ch[j];
            if (j > 0 && c.rc >= 0 && c.start != ~0ull && c.start > pos) {
                // the speculative part begins behind the known position (its first block boundaries were not dynamic blocks):
                // inflate the gap from the known position; if that lands exactly on the chunk's start the chunk is kept
                uint64_t gend = 0;
                const int grc = inflate_blocks<false>(data, size, pos, c.start, cur_win.data(), cur_win.size(), c.bytes, gend, max_out);
                ++n_gap;
                if (grc == 0 && gend == c.start) {
                    pos = c.start;
                } else {
                    c.bytes.clear();
                }
            }
            if (j > 0) {
                if (c.rc < 0 || c.start != pos) {
                    // speculative start missing or wrong: inflate this chunk again from the known position (only if it still
                    // lies ahead: a predecessor may have run past this chunk's end already)
                    const uint64_t stop = j + 1 < nc ? ch[j + 1].cut : std::min<uint64_t>(file_bits, c.cut + (uint64_t)cs * 8u);
                    if (pos >= stop) { c.out_len = 0; c.sym.clear(); c.bytes.clear(); c.last = cur_win; c.end = pos; c.rc = 0; before[j] = cur_win; ++used; continue; }
                    c.sym.clear();
                    c.bytes.clear();
                    c.rc = inflate_blocks<false>(data, size, pos, stop, cur_win.data(), cur_win.size(), c.bytes, c.end, max_out);
                    if (c.rc < 0) { err = "invalid deflate data"; return false; }
                    ++n_serial;
                } else {
                    ++n_spec_ok;
                }
            }
            before[j] = cur_win;
            take_last(c, cur_win, mid[j]);
            cur_win = c.last;
            pos = c.end;
            ++used;
            if (c.rc == 1) { fin = true; break; }
        }
        // resolve + copy out + CRC, in parallel
        size_t total = 0;
        for (size_t j = 0; j < used; ++j) {
            ch[j].out_len = ch[j].bytes.size() + ch[j].sym.size();
            ch[j].out_off = total;
            total += ch[j].out_len;
        }
        const size_t base = out.size();
        // grow the caller's buffer ONCE, to what a round of `want` bytes can come to: regrowing it round after round while the
        // ratio estimate settles cost 5-10 ms of munmap / mmap per round on the serial path (a third of a 300 MB file's time)
        if (out.capacity() < base + total) out.reserve(std::max(base + total + total / 4, base + want + want / 2));
        out.resize(base + total);
        auto resolve = [&](int t, int nt) {
            for (size_t j = (size_t)t; j < used; j += (size_t)nt) {
                Chunk& c = ch[j];
                uint8_t* dst = (uint8_t*)out.data() + base + c.out_off;
                if (!c.bytes.empty()) memcpy(dst, c.bytes.data(), c.bytes.size());
                if (!c.sym.empty()) {
                    uint8_t* d2 = dst + c.bytes.size();
                    const std::vector<uint8_t>& w = mid[j];
                    const int64_t shift = (int64_t)WSIZE - (int64_t)w.size();
                    const size_t ns = c.sym.size();
                    const uint16_t* sy = c.sym.data();
                    size_t i = 0;
#if defined(__SSE2__)
                    // markers only occur while a chunk still reaches into the 32 KiB before it: almost every group of 16 symbols
                    // is 16 plain bytes -- one pack instruction
                    for (; i + 16 <= ns; i += 16) {
                        const __m128i a = _mm_loadu_si128((const __m128i*)(sy + i)), b2 = _mm_loadu_si128((const __m128i*)(sy + i + 8));
                        if (_mm_movemask_epi8(_mm_or_si128(a, b2)) & 0xAAAA) {
                            for (size_t j = i; j < i + 16; ++j) {
                                const uint16_t s = sy[j];
                                const int64_t k = (int64_t)(s - MARK) - shift;
                                d2[j] = s < MARK ? (uint8_t)s : (k >= 0 ? w[(size_t)k] : 0);
                            }
                        } else {
                            _mm_storeu_si128((__m128i*)(d2 + i), _mm_packus_epi16(a, b2));
                        }
                    }
#endif
                    for (; i < ns; ++i) {
                        const uint16_t s = sy[i];
                        if (s < MARK) d2[i] = (uint8_t)s;
                        else {
                            const int64_t k = (int64_t)(s - MARK) - shift;
                            d2[i] = k >= 0 ? w[(size_t)k] : 0;
                        }
                    }
                }
                c.crc = crc32_fast(crc32(0L, Z_NULL, 0), dst, c.out_len);
            }
        };
        const double t2 = timing ? now() : 0.0;
        if (team && used > 1) team((int)std::min<size_t>(used, (size_t)T), resolve);
        else resolve(0, 1);
        if (timing && nc > 1) {
            double mn = 1e9, mx = 0, sum = 0, fsum = 0;
            for (size_t j = 0; j < nc; ++j) { mn = std::min(mn, t_chunk[j]); mx = std::max(mx, t_chunk[j]); sum += t_chunk[j]; fsum += t_find[j]; }
            fprintf(stderr, "[gzpar] chunks: decode min %.1f / mean %.1f / max %.1f ms, block search mean %.2f ms\n", 1e3 * mn, 1e3 * sum / (double)nc, 1e3 * mx, 1e3 * fsum / (double)nc);
        }
        if (timing) fprintf(stderr, "[gzpar] round: %zu chunks of %zu KB, inflate %.1f ms, stitch %.1f ms, resolve+crc %.1f ms, %zu MB text, %llu gaps so far\n", nc, cs >> 10,
                            1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (now() - t2), total >> 20, (unsigned long long)n_gap);
        // a marker below the start of the known history would be a reference before the member's first byte
        for (size_t j = 0; j < used; ++j) {
            crc = crc32_combine(crc, ch[j].crc, (z_off_t)ch[j].out_len);
            isize += ch[j].out_len;
        }
        const size_t consumed = (size_t)((pos + 7) >> 3) - first;
        if (consumed > 0 && total > 0) {
            const double seen = (double)total / (double)consumed;
            ratio = std::min(50.0, std::max(1.0, measured ? 0.5 * ratio + 0.5 * seen : seen));    // (the first round replaces the guess)
            measured = true;
        }
        win = cur_win;
        bit = pos;
        if (fin) {
            if (!finish_member(pos)) return false;
            if (next_member >= size) done = true;
        }
        return true;
    }
};

}  // namespace gzpar
