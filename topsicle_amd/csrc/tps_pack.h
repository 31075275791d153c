// tps_pack.h -- the packed batch format the scan kernels read, and its host-side packer.
//
// A batch of reads lives in HBM as
//   seq2   uint32 words, 16 bases per word, base j of a word in bits [2j, 2j+1]; code = (ASCII >> 1) & 3, i.e.
//          A,a -> 0   C,c -> 1   T,t -> 2   G,g -> 3   (anything else gets the code its bits give and is flagged in inv)
//   inv    uint16 per word, bit j set = base j of the word is NOT one of acgtACGT (it can never match a k-mer, exactly like
//          the reference's literal regex on .upper(): allsteps.py:176-177, 267-271)
//   desc   one tps_read_desc per read: word offset of its first base (a multiple of 4: reads start on 16-byte boundaries,
//          so the kernels load whole aligned 64-base quads that belong to ONE read), length, flags (bit 0 = the read holds
//          at least one invalid base; without it the kernels never touch inv)
// Words past a read's last base up to the next quad boundary are zero in seq2 and inv.
//
// This is 3 bits per base on the host side of PCIe instead of 8 (2.67x less upload traffic) and the kernels' staging
// becomes a plain copy HBM -> LDS.  The same layout is produced by the device pack kernel behind tps_batch_upload (ASCII
// in) and by tps::pack_reads below (used by the native reader libtopsicle_io.so and by the test emulation).
// Plain C++, no HIP.
#pragma once
#include <stdint.h>
#include <string.h>

#include "../../include/topsicle_hip.h"

namespace tps {

constexpr int QUAD_BASES = 64;                    // bases per 16-byte quad (4 words)

// words a read of L bases occupies (whole quads)
inline int64_t packed_words(int64_t L) { return ((L + QUAD_BASES - 1) / QUAD_BASES) * 4; }

// ASCII -> (code, invalid) tables
struct PackLut {
    uint8_t code[256];
    uint8_t bad[256];
    PackLut() {
        for (int c = 0; c < 256; ++c) {
            code[c] = (uint8_t)((c >> 1) & 3);
            bad[c] = 1;
        }
        for (const char* p = "ACGTacgt"; *p; ++p) bad[(unsigned char)*p] = 0;
    }
};
inline const PackLut& pack_lut() { static const PackLut l; return l; }

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define TPS_PACK_AVX2 1
}  // namespace tps
#include <immintrin.h>
namespace tps {
// 32 bases per iteration: codes = (byte >> 1) & 3; two multiply-add steps fold four codes into one byte; validity =
// (byte | 0x20) is one of a, c, g, t.  Returns the words done (a multiple of 2); *any_bad accumulates the invalid bits.
__attribute__((target("avx2"))) inline int64_t pack_avx2(const uint8_t* s, int64_t full_words, uint32_t* seq2, uint16_t* inv,
                                                          uint32_t* any_bad) {
    const __m256i m3 = _mm256_set1_epi8(3), w14 = _mm256_set1_epi16(0x0401), w116 = _mm256_set1_epi32(0x00100001);
    const __m256i lower = _mm256_set1_epi8(0x20), ca = _mm256_set1_epi8('a'), cc = _mm256_set1_epi8('c'),
                  cg = _mm256_set1_epi8('g'), ct = _mm256_set1_epi8('t');
    // after the two madds every 32-bit lane holds one packed byte (4 bases) in its low byte: gather them
    const __m256i gather = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                            0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    uint32_t any = 0;
    int64_t w = 0;
    for (; w + 2 <= full_words; w += 2) {
        const __m256i v = _mm256_loadu_si256((const __m256i*)(s + 16 * w));
        const __m256i code = _mm256_and_si256(_mm256_srli_epi16(v, 1), m3);
        const __m256i p2 = _mm256_maddubs_epi16(code, w14);             // c0 + 4 c1 per 16-bit lane
        const __m256i p4 = _mm256_madd_epi16(p2, w116);                 // + 16 (c2 + 4 c3) per 32-bit lane
        const __m256i g = _mm256_shuffle_epi8(p4, gather);              // dword 0 of each 128-bit half = 16 bases
        seq2[w] = (uint32_t)_mm256_extract_epi32(g, 0);
        seq2[w + 1] = (uint32_t)_mm256_extract_epi32(g, 4);
        const __m256i lc = _mm256_or_si256(v, lower);
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(lc, ca), _mm256_cmpeq_epi8(lc, cc)),
                                           _mm256_or_si256(_mm256_cmpeq_epi8(lc, cg), _mm256_cmpeq_epi8(lc, ct)));
        const uint32_t bad = ~(uint32_t)_mm256_movemask_epi8(ok);
        if (inv) { inv[w] = (uint16_t)bad; inv[w + 1] = (uint16_t)(bad >> 16); }
        any |= bad;
    }
    *any_bad |= any;
    return w;
}
inline bool have_avx2() { static const bool v = __builtin_cpu_supports("avx2"); return v; }
#endif

// Packs one read into seq2 / inv (both hold packed_words(L) entries); returns true if the read has an invalid base.
inline bool pack_one(const uint8_t* s, int64_t L, uint32_t* seq2, uint16_t* inv) {
    const PackLut& t = pack_lut();
    const int64_t nw = packed_words(L);
    const int64_t full = L >> 4;
    uint32_t any = 0;
    int64_t w0 = 0;
#ifdef TPS_PACK_AVX2
    if (have_avx2()) w0 = pack_avx2(s, full, seq2, inv, &any);
#endif
    for (int64_t w = w0; w < full; ++w) {
        const uint8_t* p = s + 16 * w;
        uint32_t v = 0, b = 0;
        for (int j = 0; j < 16; ++j) {
            v |= (uint32_t)t.code[p[j]] << (2 * j);
            b |= (uint32_t)t.bad[p[j]] << j;
        }
        seq2[w] = v;
        if (inv) inv[w] = (uint16_t)b;
        any |= b;
    }
    int64_t w = full;
    if (L & 15) {
        const uint8_t* p = s + 16 * w;
        uint32_t v = 0, b = 0;
        for (int j = 0; j < (int)(L & 15); ++j) {
            v |= (uint32_t)t.code[p[j]] << (2 * j);
            b |= (uint32_t)t.bad[p[j]] << j;
        }
        seq2[w] = v;
        if (inv) inv[w] = (uint16_t)b;
        any |= b;
        ++w;
    }
    for (; w < nw; ++w) {
        seq2[w] = 0;
        if (inv) inv[w] = 0;
    }
    return any != 0;
}

// Layout of a batch: fills desc[i].word_off / len (flags = 0) from the base offsets and returns the total words.
inline int64_t pack_layout(const int64_t* offsets, int64_t n, tps_read_desc* desc) {
    int64_t w = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t L = offsets[i + 1] - offsets[i];
        desc[i].word_off = w;
        desc[i].len = (int32_t)L;
        desc[i].flags = 0;
        w += packed_words(L);
    }
    return w;
}

// Packs reads [lo, hi) of a batch whose layout is already in desc (threads of a team each take a range of reads).
inline void pack_range(const uint8_t* bases, const int64_t* offsets, int64_t lo, int64_t hi, tps_read_desc* desc,
                       uint32_t* seq2, uint16_t* inv) {
    for (int64_t i = lo; i < hi; ++i)
        if (pack_one(bases + offsets[i], desc[i].len, seq2 + desc[i].word_off, inv ? inv + desc[i].word_off : nullptr))
            desc[i].flags |= TPS_RD_HAS_INVALID;
}

}  // namespace tps
