/*
 * topsicle_hip.h -- C ABI of libtopsicle_hip.so: the MI355X (gfx950) implementation of
 * Topsicle's per-read hot path.
 *
 * The reference (jaeyoungchoilab/Topsicle) has NO plugin / FFI seam: the path is a chain of
 * plain Python calls inside Topsicle/allsteps.py, driven by Topsicle/main.py:52-154.  This
 * header is therefore the boundary a maintainer would bind (ctypes stub in INTEGRATION.md);
 * each entry point names the reference code it replaces (file:line relative to the reference
 * root).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no exceptions cross the boundary.
 *   - every function returns 0 on success or a negative TPS_E_* code; tps_last_error() gives
 *     the message of the last failure on the calling thread.
 *   - the caller allocates and owns every host buffer; the library keeps no host pointer
 *     after a call returns -- with ONE exception: uploads from buffers obtained with tps_host_alloc
 *     (pinned memory) are asynchronous on the context's stream, and those buffers must not be
 *     rewritten or freed before the next tps_sync (or blocking download) on that context.
 *     Device memory is owned by the context.
 *   - one context per (host thread, device); a context is not thread-safe, distinct contexts
 *     are independent.  There is NO CPU fallback: creating a context without a usable GPU
 *     fails with TPS_E_NO_DEVICE.
 *   - reads are passed as ONE concatenated ASCII byte string plus n+1 offsets (any case,
 *     any IUPAC letter; only A/C/G/T in either case can match, exactly like the reference's
 *     .upper() + literal regex) -- or already 2-bit packed (tps_batch_upload_packed).  Either way a batch is
 *     RESIDENT in HBM in the packed format below; the scan kernels read nothing else.
 *
 * Packed batch format (what the reference's Bio.SeqIO records, allsteps.py:127-149, become on this path)
 *   seq2   uint32 words, 16 bases per word, base j of a word in bits [2j, 2j+1]; code = (ASCII >> 1) & 3:
 *          A,a = 0   C,c = 1   T,t = 2   G,g = 3
 *   inv    uint16 per word: bit j set = base j of the word is not one of acgtACGT (it never matches)
 *   desc   one tps_read_desc per read; a read starts on a 16-byte boundary (word_off is a multiple of 4) and the words
 *          up to the next such boundary after its last base are zero in seq2 and inv
 *   3 bits per base instead of 8 across PCIe; libtopsicle_io.so's reader emits this format directly.
 */
#ifndef TOPSICLE_HIP_H
#define TOPSICLE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPS_ABI_VERSION 4

/* error codes */
#define TPS_OK            0
#define TPS_E_NO_DEVICE  -1   /* no HIP device / device index out of range              */
#define TPS_E_HIP        -2   /* a HIP runtime call failed (message has the HIP error)  */
#define TPS_E_ARG        -3   /* invalid argument                                        */
#define TPS_E_PATTERN    -4   /* pattern table not set / unsupported (non-ACGT, k>15, P>31) */
#define TPS_E_CAPACITY   -5   /* parameter combination does not fit the kernel's LDS plan */
#define TPS_E_STATE      -6   /* call order (e.g. scan before upload)                    */

/* limits of this build */
#define TPS_MAX_K         15  /* k-mer length: 4^k-entry LDS table up to TPS_DIRECT_K, a perfect-hash table above */
#define TPS_DIRECT_K      7
#define TPS_MAX_PATTERNS 31   /* patterns in one table (bit 31 of the mask is a flag)       */

/* flags for tps_params.flags */
#define TPS_F_STEP1      1u   /* run the TRC step (a3) and choose tail / pass per read       */
#define TPS_F_WINDOWS    2u   /* run the sliding-window count step (a5) on passing reads     */
#define TPS_F_BINSEG     4u   /* run the single-split change-point step (a6) in the same launch */
#define TPS_F_STORE_SUMS 8u   /* make S_w downloadable (tps_batch_window_sums hands out int32)  */
#define TPS_F_STORE_RAW 16u   /* keep c'_p per window and pattern (u8) -- rawCountPattern (a7) */
#define TPS_F_TAILS_IN  32u   /* without STEP1: tails[] and pass come from the caller         */

typedef struct tps_ctx tps_ctx;

/* One read of a packed batch (16 bytes). */
typedef struct tps_read_desc {
    int64_t  word_off;    /* index of the read's first word in seq2 / inv; a multiple of 4            */
    int32_t  len;         /* bases                                                                     */
    uint32_t flags;       /* TPS_RD_*                                                                  */
} tps_read_desc;
#define TPS_RD_HAS_INVALID 1u   /* the read holds at least one base that is not acgtACGT (inv is consulted) */

/* Parameters of one scan.  Names follow the reference CLI (Topsicle/main.py:319-334). */
typedef struct tps_params {
    int32_t no_bp;        /* step-1 tail length; reference hard-codes 1000 (main.py:57)        */
    int32_t min_len;      /* --minSeqLength: keep reads with L >  min_len (allsteps.py:175)    */
    int32_t min_count;    /* keep reads whose best count > min_count; the host derives it from
                             --cutoff so that count/(no_bp/len(pattern)) > cutoff in float64  */
    int32_t window;       /* --windowSize W (window text is W-1 chars, allsteps.py:221-224)    */
    int32_t slide;        /* --slide s                                                         */
    int32_t trimfirst;    /* --trimfirst t                                                     */
    int32_t maxlen;       /* --maxlengthtelo M                                                 */
    int32_t jump;         /* ruptures Binseg jump (5)                                          */
    int32_t min_size;     /* ruptures Binseg min_size (2)                                      */
    uint32_t flags;       /* TPS_F_*                                                           */
} tps_params;

/* Per-read result of a scan (48 bytes; ABI 3 added `flags`). */
typedef struct tps_read_result {
    int32_t best_start;      /* max_p count in the first no_bp bases                (a3)      */
    int32_t best_start_idx;  /* first pattern index reaching it (allsteps.py:190)             */
    int32_t best_end;        /* max_p count in the reversed last no_bp bases                  */
    int32_t best_end_idx;    /* first pattern index reaching it (allsteps.py:191)             */
    int32_t tail;            /* 0 = forward, 1 = reverse (allsteps.py:193-198)                */
    int32_t pass;            /* 1 if L > min_len and best count > min_count                   */
    int32_t n_win;           /* windows of the chosen tail (0 if not scanned)                 */
    int32_t bkp;             /* best split index, -1 if none admissible / not run   (a6)      */
    double  gain;            /* l2 gain of that split on y = S/P (float64, informational)     */
    uint32_t flags;          /* TPS_RES_*                                                     */
    uint32_t reserved;
} tps_read_result;
/* The change-point was decided by the exact integer tournament: two or more candidates lay within float64 rounding noise of
 * the best gain (a constant signal; exact rational ties).  ruptures compares float64 gains there (allsteps.py:310-311), so
 * its answer is a matter of rounding noise, not of the data: a caller that wants THAT answer downloads the read's S_w
 * (tps_batch_read_sums) and repeats ruptures' float64 arithmetic on it -- a handful of reads at most; topsicle_amd does
 * (hiplib.HipScanner.resolve_ties).  bkp itself holds the exact rule's answer: ties -> the larger index. */
#define TPS_RES_TIE 1u

/* ---- context ------------------------------------------------------------------------- */
int  tps_abi_version(void);
int  tps_device_count(int* n);
int  tps_ctx_create(int device, tps_ctx** out);
int  tps_ctx_destroy(tps_ctx* ctx);
const char* tps_last_error(void);

/* Replaces patterns_to_search's consumer side (allsteps.py:167-168, 249-250, 378-379): the
 * compiled regex list.  `pats` holds P strings of k ASCII letters back to back, in reference
 * order (sorted k-mers then their complements, allsteps.py:104-120); order defines pattern
 * indices (first-max tie-break, raw count column order). */
int  tps_set_patterns(tps_ctx* ctx, const char* pats, int32_t n_patterns, int32_t k);

/* ---- resident batch (throughput path) ------------------------------------------------ */
/* Copy a batch of reads into HBM.  `slot` (0..TPS_MAX_SLOTS-1) names one resident batch so
 * several can be kept and scanned in turn.  Replaces the per-call file parse of
 * unzip_file/SeqIO.parse (allsteps.py:127-149, 174, 257) with one upload per batch. */
#define TPS_MAX_SLOTS 16
int  tps_batch_upload(tps_ctx* ctx, int32_t slot, const uint8_t* bases, const int64_t* offsets,
                      int64_t n_reads);
/* The same for a batch the host has already packed (format above): seq2[n_words], inv[n_words] (NULL = no read has an
 * invalid base), desc[n_reads].  Three bits per base cross PCIe instead of eight.  Replaces allsteps.py:127-149 +
 * main.py:68-86 (the parse that feeds every later step) together with libtopsicle_io.so's tps_reader_next_packed.
 * Buffers obtained from tps_host_alloc are pinned: the copies then run asynchronously on the context's stream and the
 * call returns at once -- the caller must keep those buffers unchanged until tps_sync (or any result download of the
 * slot) has returned; ordinary memory is copied before the call returns. */
int  tps_batch_upload_packed(tps_ctx* ctx, int32_t slot, const uint32_t* seq2, const uint16_t* inv,
                             const tps_read_desc* desc, int64_t n_reads, int64_t n_words);
/* Make `slot` of `ctx` refer to the resident packed batch of `src_slot` of ANOTHER context on the same device (no copy; the
 * batch stays owned by `src`, which must neither upload into that slot nor be destroyed while the borrower may still scan it;
 * `ctx`'s stream waits for `src`'s pending upload).  Several pattern tables over one batch -- `--telophrase 4 5 6`, the
 * reference's outer loop over k (Topsicle/main.py:206-235) -- are then scanned by one context per table at the same time:
 * every context keeps its own table, outputs and stream, the launches overlap on the GPU. */
int  tps_batch_share(tps_ctx* ctx, int32_t slot, tps_ctx* src, int32_t src_slot);
/* Pinned host memory for upload staging buffers (double buffering: fill one while the other is in flight). */
int  tps_host_alloc(tps_ctx* ctx, int64_t bytes, void** out);
int  tps_host_free(tps_ctx* ctx, void* p);
/* Download the resident packed batch of `slot` (tests: the device pack kernel vs the host packer; diagnostics).
 * seq2 / inv hold n_words entries, desc n_reads; any of the three may be NULL.  n_words out via *n_words_out. */
int  tps_batch_download_packed(tps_ctx* ctx, int32_t slot, uint32_t* seq2, uint16_t* inv, tps_read_desc* desc,
                               int64_t n_reads, int64_t n_words, int64_t* n_words_out);
/* Optional per-read tails (0/1) and pass flags for scans without TPS_F_STEP1. */
int  tps_batch_set_tails(tps_ctx* ctx, int32_t slot, const uint8_t* tails);

/* One pass of the hot path over a resident batch: patternTRC_count (allsteps.py:152-204) +
 * bound_detect's window loop (allsteps.py:257-297) + process_mean/Binseg
 * (allsteps.py:300-333), one 64-lane wave per read, one launch.  Asynchronous on the
 * context's stream; results land in context-owned pinned memory. */
int  tps_batch_scan(tps_ctx* ctx, int32_t slot, const tps_params* prm);
/* Wait for everything enqueued on the context's stream. */
int  tps_sync(tps_ctx* ctx);
/* Copy the per-read results of the last scan of `slot` (after tps_sync). */
int  tps_batch_results(tps_ctx* ctx, int32_t slot, tps_read_result* out, int64_t n_reads);
/* Window layout of the last scan: win_off[n+1] (prefix of n_win over ALL reads of the batch,
 * scanned or not, computed from lengths and params only). */
int  tps_batch_window_offsets(tps_ctx* ctx, int32_t slot, int64_t* win_off, int64_t n_plus_1);
/* Download S_w (needs TPS_F_STORE_SUMS) / c'_p (needs TPS_F_STORE_RAW) of the last scan. */
int  tps_batch_window_sums(tps_ctx* ctx, int32_t slot, int32_t* sums, int64_t n_windows);
int  tps_batch_window_raw(tps_ctx* ctx, int32_t slot, uint8_t* raw, int64_t n_windows_times_p);
/* c'_p of SELECTED reads of the last scan (needs TPS_F_STORE_RAW) straight to a file (ABI 4): the rows of reads[0 .. n_sel)
 * -- ascending indices into the batch -- back to back, u8[n_win x P] each, at byte `file_off` of `fd` (pwrite: the descriptor's
 * own position is not used, several contexts may write into one file at once).  The rows never stop in the caller's memory:
 * device -> two pinned pieces of the context -> pwritev, the next piece copied while the current one is written.  crc_fn (may be
 * NULL; e.g. libtopsicle_io.so's tps_crc32) is run over the bytes in file order, starting from 0: *crc_out (may be NULL) is the
 * CRC-32 of exactly what was written, *bytes_out its length -- a caller that lays the blocks of many batches into one zip / npy
 * member joins the per-block values with crc32_combine.  Replaces, for the columnar raw-count output, the per-read
 * DataFrame.to_csv of the reference (Topsicle/main.py:146-150; Topsicle/allsteps.py:398-411 builds the rows). */
typedef uint32_t (*tps_crc32_fn)(uint32_t crc, const uint8_t* p, int64_t n);
int  tps_batch_raw_to_fd(tps_ctx* ctx, int32_t slot, const int64_t* reads, int64_t n_sel, int fd, int64_t file_off,
                         tps_crc32_fn crc_fn, uint32_t* crc_out, int64_t* bytes_out);
/* S_w of ONE read of the last scan (any scan with TPS_F_WINDOWS; n_windows = the read's n_win). */
int  tps_batch_read_sums(tps_ctx* ctx, int32_t slot, int64_t read, int32_t* sums, int64_t n_windows);
/* Step-1 per-pattern counts of the last scan: c_start[n*P], c_end[n*P] (int32). */
int  tps_batch_trc_counts(tps_ctx* ctx, int32_t slot, int32_t* c_start, int32_t* c_end, int64_t n_reads);

/* ---- one-shot host-pointer calls (parity tests, per-read API) -------------------------- */
/* patternTRC_count's counting loop (allsteps.py:176-184): integer counts only; the float64
 * TRC, arg-max and cutoff stay in the host language so they are bit-identical. */
int  tps_trc_counts(tps_ctx* ctx, const uint8_t* bases, const int64_t* offsets, int64_t n_reads,
                    int32_t no_bp, int32_t* c_start, int32_t* c_end);
/* bound_detect / rawCountPattern window loops (allsteps.py:275-291, 398-411) for the given
 * tail per read.  win_off[n+1] must be the prefix of tps_window_count() over the reads.
 * sums: int32[win_off[n]]; raw: u8[win_off[n]*P] or NULL. */
int  tps_window_counts(tps_ctx* ctx, const uint8_t* bases, const int64_t* offsets,
                       const uint8_t* tails, int64_t n_reads, int32_t window, int32_t slide,
                       int32_t trimfirst, int32_t maxlen, const int64_t* win_off,
                       int32_t* sums, uint8_t* raw);
/* rpt.Binseg(model="l2").fit(y).predict(n_bkps=1) (allsteps.py:310-311) on integer window
 * sums: bkp[n] (-1 = no admissible split), gain[n] (may be NULL). */
int  tps_binseg_l2(tps_ctx* ctx, const int32_t* sums, const int64_t* win_off, int64_t n_reads,
                   int32_t n_patterns, int32_t jump, int32_t min_size, int32_t* bkp, double* gain);
/* The same, also reporting per read whether the exact tournament decided (tie[n]: 1 = TPS_RES_TIE, see above; may be NULL). */
int  tps_binseg_l2_ties(tps_ctx* ctx, const int32_t* sums, const int64_t* win_off, int64_t n_reads,
                        int32_t n_patterns, int32_t jump, int32_t min_size, int32_t* bkp, double* gain, uint8_t* tie);

/* Number of windows seq_cut_windows yields for a read of length L (allsteps.py:219, 263-271). */
int64_t tps_window_count(int64_t read_len, int32_t window, int32_t slide, int32_t trimfirst, int32_t maxlen);

/* ---- exploratory counts (overview plots) ------------------------------------------------- */
/* patterns_vs_match_heatmap's counting loop (descriptive_plot.py:259-291): for every read of the resident batch longer
 * than min_len, in bases [lo, hi) of the read (strand 0) and of its reverse complement (strand 1), the leftmost
 * non-overlapping matches of  kmer(.{follow})  for each of the first n_fwd patterns of the table -- the k-mers of the
 * doubled motif; the table must hold their complements behind them, as patterns_to_search builds it (allsteps.py:104-120).
 *   picks  uint32[n_reads][2][n_fwd][pw], pw = ceil((hi - lo) / 32): bit j of word w = a match starts at base lo + 32 w + j
 *          of that strand's string (the reference's DataFrame rows: Pattern = the k-mer, Match = the `follow` letters
 *          behind it, which the caller reads from its own copy of the read)
 *   hist   int64[2][n_fwd][4^follow + 1] or NULL: the crosstab summed over the batch; the bin of a match is the 2-bit
 *          code of the following bases (first base in the low bits, A C T G = 0 1 2 3, as the reference's upper-cased,
 *          complemented strand-1 string spells them); the last bin counts matches followed by a non-ACGT letter.
 * follow <= 8, hi - lo <= 4096, n_fwd <= 15. */
int  tps_batch_kmer_followers(tps_ctx* ctx, int32_t slot, int32_t n_fwd, int32_t follow, int32_t lo, int32_t hi,
                              int32_t min_len, uint32_t* picks, int64_t picks_words, int64_t* hist, int64_t hist_len);

/* ---- measurement ----------------------------------------------------------------------- */
/* hipEvent timings of the scan kernel launches since the last reset: number of launches,
 * total and mean milliseconds (events are recorded on the stream the kernel runs on). */
int  tps_kernel_time_ms(tps_ctx* ctx, int32_t* n_launches, double* total_ms, double* mean_ms);
int  tps_kernel_time_reset(tps_ctx* ctx);
/* Diagnostics and tests only (ABI 4).  The library reads NO environment variables: what experiments and tests need to steer is
 * set per context through this call -- "event_stride" (time every n-th launch; default 1), "no_events", "force_generic" (the
 * generic kernel instead of the fused tiles), "spans_per_tile", "force_pair", "so_order", "wpg" (csrc/tps_plan.h: PlanKnobs), "stamps"
 * (per-read phase clocks; only a -DTPS_STAMPS build writes them), "file_order" (wave slot i takes read i: switches
 * tps::plan_dispatch_order off, which otherwise starts the longest reads of a batch first), "no_stride" (a slide that is a multiple of a
 * fused kernel's slide keeps the generic kernel instead of running that kernel and keeping every m-th window: tps::stride_base), "no_inline_rows" (such a scan at twice the base
 * slide copies its raw rows like the other multiples instead of letting the tiles store every second row: ScanArgs::raw_m).  topsicle_amd.hiplib applies $TOPSICLE_HIP_DEBUG
 * ("key=value,key=value") to every context it creates: the ONE documented variable of the Python host. */
int  tps_ctx_debug_option(tps_ctx* ctx, const char* key, int64_t value);
/* The stamps of the last scan of `slot`: 16 uint64 per read. */
int  tps_debug_stamps_get(tps_ctx* ctx, int32_t slot, uint64_t* out, int64_t n_reads);
/* Name of the device and a few properties, as a NUL-terminated string. */
int  tps_device_info(tps_ctx* ctx, char* buf, int32_t buf_len);
/* What the last tps_batch_scan of `slot` launched: "<kernel name> lds=<bytes per workgroup> wgs_per_cu=<n> waves_per_wg=<n>"
 * (the kernel family is chosen per scan from slide, table and LDS plan; profiles and bench.py name it).  A scan whose slide is a
 * multiple of a fused kernel's slide reports "<that kernel> every 2nd window" (3rd, 4th ...): it ran that kernel at the base slide
 * and kept every m-th window (tps_stride_kernel; outputs in the layout of the requested slide). */
int  tps_batch_kernel_info(tps_ctx* ctx, int32_t slot, char* buf, int32_t buf_len);

#ifdef __cplusplus
}
#endif
#endif /* TOPSICLE_HIP_H */
