/*
 * oracle.c -- TEST INFRASTRUCTURE ONLY: plain-C restatement of Topsicle's hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * oracle/_build/liboracle.so; the product (topsicle_amd) never does.  It restates the
 * reference the way the reference computes it -- build the (upper-cased, possibly reversed)
 * tail string, cut W-1 character windows, count leftmost non-overlapping literal matches per
 * pattern with a string search, floor to 1, average -- so it is independent of the HIP kernel's
 * formulation (2-bit packing, lookup table, block masks).  Citations are file:line relative to
 * the reference root.
 *
 * Pinned by tests/test_oracle_c.py against the Python oracle, the reference-generated goldens
 * (tests/golden) and numpy's float64 variance for the change-point part.
 * Change-point arithmetic (a6) restates the un-vendored third-party ruptures==1.1.9 Binseg /
 * CostL2 (requirements.txt:7): beyond the 17 demo boundaries its parity is UNPINNED.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---------------------------------------------------------------- match counting
 * len(list(re.finditer(pat, text))) for a literal pattern (allsteps.py:182-183, 281, 288, 402, 408):
 * leftmost match, then continue right after it. */
static int nonoverlap_count(const char* text, int n, const char* pat, int k) {
    int cnt = 0, i = 0;
    if (k <= 0) return n + 1;
    while (i + k <= n) {
        if (text[i] == pat[0] && memcmp(text + i, pat, (size_t)k) == 0) {
            ++cnt;
            i += k;
        } else {
            ++i;
        }
    }
    return cnt;
}

static char up(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }

int orc_nonoverlap_count(const char* text, int n, const char* pat, int k) { return nonoverlap_count(text, n, pat, k); }

/* ---------------------------------------------------------------- a3: step-1 counts
 * seq[:no_bp].upper() and seq[-no_bp:][::-1].upper() (allsteps.py:176-177). */
void orc_trc_counts(const char* seq, int64_t L, const char* pats, int P, int k, int no_bp, int32_t* c_start, int32_t* c_end) {
    int n = (int)(L < no_bp ? L : no_bp);
    char* head = (char*)malloc((size_t)n + 1);
    char* tail = (char*)malloc((size_t)n + 1);
    for (int i = 0; i < n; ++i) {
        head[i] = up(seq[i]);
        tail[i] = up(seq[L - 1 - i]);
    }
    for (int p = 0; p < P; ++p) {
        c_start[p] = nonoverlap_count(head, n, pats + (size_t)p * k, k);
        c_end[p] = nonoverlap_count(tail, n, pats + (size_t)p * k, k);
    }
    free(head);
    free(tail);
}

/* first maximum (allsteps.py:190-191), forward only if strictly larger (193), strict cutoff (194,197).
 * returns 1 if kept; *tail 0 forward / 1 reverse; *best_idx pattern index; *trc float64 value. */
int orc_trc_call(const int32_t* c_start, const int32_t* c_end, int P, int no_bp, int motif_len, double cutoff,
                 int* tail, int* best_idx, double* trc) {
    double ratio = (double)no_bp / (double)motif_len;
    int is = 0, ie = 0;
    for (int p = 1; p < P; ++p) {
        if ((double)c_start[p] / ratio > (double)c_start[is] / ratio) is = p;
        if ((double)c_end[p] / ratio > (double)c_end[ie] / ratio) ie = p;
    }
    double fs = (double)c_start[is] / ratio, fe = (double)c_end[ie] / ratio;
    if (fs > fe) { *tail = 0; *best_idx = is; *trc = fs; return fs > cutoff; }
    *tail = 1; *best_idx = ie; *trc = fe;
    return fe > cutoff;
}

/* ---------------------------------------------------------------- a4: window count */
int64_t orc_window_count(int64_t L, int W, int s, int t, int M) {
    int64_t m = L < M ? L : M, ns = m - t;
    if (W < 1 || s < 1 || ns < W) return 0;
    return (ns - W) / s + 1;     /* len(range(0, ns - W + 1, s)) (allsteps.py:219) */
}

/* ---------------------------------------------------------------- a5 / a7: per-window counts
 * tail string (allsteps.py:263-271), windows of W-1 characters (221-224), `matches or 1` (281, 288).
 * sums[n_win] (may be NULL), raw[n_win*P] (may be NULL).  Returns n_win. */
int64_t orc_window_counts(const char* seq, int64_t L, int tail, const char* pats, int P, int k, int W, int s, int t, int M,
                          int32_t* sums, uint8_t* raw) {
    int64_t m = L < M ? L : M, ns = m - t;
    int64_t nwin = orc_window_count(L, W, s, t, M);
    if (nwin <= 0) return 0;
    char* str = (char*)malloc((size_t)ns + 1);
    for (int64_t i = 0; i < ns; ++i) str[i] = up(tail ? seq[L - 1 - t - i] : seq[t + i]);
    for (int64_t w = 0; w < nwin; ++w) {
        const char* text = str + w * s;
        int32_t sum = 0;
        for (int p = 0; p < P; ++p) {
            int c = nonoverlap_count(text, W - 1, pats + (size_t)p * k, k);
            if (c == 0) c = 1;
            if (raw) raw[w * P + p] = (uint8_t)c;
            sum += c;
        }
        if (sums) sums[w] = sum;
    }
    free(str);
    return nwin;
}

/* ---------------------------------------------------------------- a6: Binseg(model="l2"), n_bkps=1
 * numpy float64 semantics: ndarray.var = mean(abs(x - x.mean())**2), sums are numpy's pairwise
 * summation (unrolled by 8, blocks of 128). */
static double pairwise_sum(const double* a, int64_t n) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
    }
}

/* CostL2.error(a, b) = signal[a:b].var(axis=0).sum() * (b - a) */
static double cost_l2(const double* y, int64_t a, int64_t b, double* scratch) {
    int64_t n = b - a;
    double mean = pairwise_sum(y + a, n) / (double)n;
    for (int64_t i = 0; i < n; ++i) {
        double d = y[a + i] - mean;
        scratch[i] = d * d;
    }
    double var = pairwise_sum(scratch, n) / (double)n;
    return var * (double)n;
}

int orc_binseg_admissible(int n, int jump, int min_size) {
    if (n / jump < 1) return 0;
    int need = ((min_size + jump - 1) / jump) * jump + min_size;
    return need <= n;
}

/* y[n] float64 in; returns bkp or -1; *gain receives the winning gain.
 * max over (gain, bkp) tuples: larger gain, ties -> larger bkp (Binseg._single_bkp). */
int orc_binseg_l2_y(const double* y, int n, int jump, int min_size, double* gain) {
    if (gain) *gain = 0.0;
    if (!orc_binseg_admissible(n, jump, min_size)) return -1;
    double* scratch = (double*)malloc(sizeof(double) * (size_t)n);
    double whole = cost_l2(y, 0, n, scratch);
    int best = -1;
    double bg = 0.0;
    for (int b = 0; b < n; b += jump) {
        if (b >= min_size && n - b >= min_size) {
            double g = whole - cost_l2(y, 0, b, scratch) - cost_l2(y, b, n, scratch);
            if (best < 0 || g > bg || (g == bg && b > best)) { best = b; bg = g; }
        }
    }
    free(scratch);
    if (gain) *gain = bg;
    return best;
}

int orc_binseg_l2(const int32_t* sums, int n, int P, int jump, int min_size, double* gain) {
    double* y = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) y[i] = (double)sums[i] / (double)P;     /* sum(counts) / len(counts), allsteps.py:284 */
    int b = orc_binseg_l2_y(y, n, jump, min_size, gain);
    free(y);
    return b;
}

/* ---------------------------------------------------------------- whole-read pipeline, as process_file does it
 * (main.py:57, 125-133): step 1; if kept, windows of BOTH tails are counted (allsteps.py:279-291) when
 * both_tails != 0, the unchosen one is dropped (294-297), Binseg on the chosen one.
 * out: [0]=pass [1]=tail [2]=best_idx [3]=best_count [4]=n_win [5]=bkp [6]=boundary_bp */
/* Position-weighted 64-bit checksum of a vector (wrap-around arithmetic): sum over i of (v[i] + 1) * CK_MUL^i.  The GPU tests
 * compute the same over the kernels' window sums / raw rows of EVERY read (tests/ckutil.py), so the at-scale comparisons are
 * total instead of sampled. */
#define CK_MUL 0x9E3779B97F4A7C15ull
static uint64_t ck_i32(const int32_t* v, int64_t n) {
    uint64_t h = 0, pw = 1;
    for (int64_t i = 0; i < n; ++i) { h += ((uint64_t)(uint32_t)v[i] + 1ull) * pw; pw *= CK_MUL; }
    return h;
}
static uint64_t ck_u8(const uint8_t* v, int64_t n) {
    uint64_t h = 0, pw = 1;
    for (int64_t i = 0; i < n; ++i) { h += ((uint64_t)v[i] + 1ull) * pw; pw *= CK_MUL; }
    return h;
}

void orc_read_pipeline_ck(const char* seq, int64_t L, const char* pats, int P, int k, int motif_len, int no_bp, int min_len,
                          double cutoff, int W, int s, int t, int M, int both_tails, int32_t* out, uint64_t* ck, int want_raw) {
    int32_t cs[64], ce[64];
    int tail = 0, idx = 0;
    double trc = 0.0;
    memset(out, 0, sizeof(int32_t) * 7);
    out[5] = -1;
    if (ck) { ck[0] = 0; ck[1] = 0; }
    if (!(L > min_len)) return;
    orc_trc_counts(seq, L, pats, P, k, no_bp, cs, ce);
    int keep = orc_trc_call(cs, ce, P, no_bp, motif_len, cutoff, &tail, &idx, &trc);
    out[1] = tail; out[2] = idx; out[3] = tail ? ce[idx] : cs[idx];
    if (!keep) return;
    out[0] = 1;
    int64_t nwin = orc_window_count(L, W, s, t, M);
    out[4] = (int32_t)nwin;
    if (nwin <= 0) return;
    int32_t* sums = (int32_t*)malloc(sizeof(int32_t) * (size_t)nwin);
    uint8_t* raw = (ck && want_raw) ? (uint8_t*)malloc((size_t)nwin * (size_t)P) : NULL;
    if (both_tails) orc_window_counts(seq, L, 1 - tail, pats, P, k, W, s, t, M, sums, NULL);
    orc_window_counts(seq, L, tail, pats, P, k, W, s, t, M, sums, raw);
    if (ck) {
        ck[0] = ck_i32(sums, nwin);
        if (raw) ck[1] = ck_u8(raw, nwin * (int64_t)P);
    }
    free(raw);
    double g;
    int bkp = orc_binseg_l2(sums, (int)nwin, P, 5, 2, &g);
    out[5] = bkp;
    if (bkp >= 0) {
        int64_t m = L < M ? L : M;
        int64_t point = (int64_t)bkp * s + t;
        out[6] = (point <= m && point != 0) ? (int32_t)point : 0;
    }
    free(sums);
}

void orc_read_pipeline(const char* seq, int64_t L, const char* pats, int P, int k, int motif_len, int no_bp, int min_len,
                       double cutoff, int W, int s, int t, int M, int both_tails, int32_t* out) {
    orc_read_pipeline_ck(seq, L, pats, P, k, motif_len, no_bp, min_len, cutoff, W, s, t, M, both_tails, out, NULL, 0);
}

/* ---------------------------------------------------------------- threaded batch driver (cpu_baseline) */
typedef struct {
    const char* bases; const int64_t* offsets; int64_t n; const char* pats; int P, k, motif_len, no_bp, min_len;
    double cutoff; int W, s, t, M, both_tails; int32_t* out; volatile int64_t* next; double deadline; volatile int64_t* done;
    uint64_t* ck; int want_raw;
} job_t;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* worker(void* arg) {
    job_t* j = (job_t*)arg;
    for (;;) {
        if (j->deadline > 0 && now_s() > j->deadline) break;
        int64_t i = __sync_fetch_and_add(j->next, 1);
        if (i >= j->n) break;
        orc_read_pipeline_ck(j->bases + j->offsets[i], j->offsets[i + 1] - j->offsets[i], j->pats, j->P, j->k, j->motif_len,
                             j->no_bp, j->min_len, j->cutoff, j->W, j->s, j->t, j->M, j->both_tails, j->out + 7 * i,
                             j->ck ? j->ck + 2 * i : NULL, j->want_raw);
        __sync_fetch_and_add(j->done, 1);
    }
    return NULL;
}

/* Runs the per-read pipeline over reads [0,n) with `threads` workers, in read order of a shared
 * counter; stops early after budget_s seconds (0 = no limit).  Returns the number of reads
 * completed (a prefix of the batch, up to thread skew); *elapsed receives wall seconds. */
/* orc_batch_ck: the same, plus per read ck[2 i] = checksum of its window sums, ck[2 i + 1] = of its raw rows (want_raw), 0 for
 * a read that does not pass */
int64_t orc_batch_ck(const char* bases, const int64_t* offsets, int64_t n, const char* pats, int P, int k, int motif_len,
                     int no_bp, int min_len, double cutoff, int W, int s, int t, int M, int both_tails, int threads,
                     double budget_s, int32_t* out, double* elapsed, uint64_t* ck, int want_raw) {
    volatile int64_t next = 0, done = 0;
    double t0 = now_s();
    job_t j = {bases, offsets, n, pats, P, k, motif_len, no_bp, min_len, cutoff, W, s, t, M, both_tails, out, &next,
               budget_s > 0 ? t0 + budget_s : 0.0, &done, ck, want_raw};
    if (threads < 1) threads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int i = 0; i < threads; ++i) pthread_create(&th[i], NULL, worker, &j);
    for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
    free(th);
    if (elapsed) *elapsed = now_s() - t0;
    return done;
}
int64_t orc_batch(const char* bases, const int64_t* offsets, int64_t n, const char* pats, int P, int k, int motif_len,
                  int no_bp, int min_len, double cutoff, int W, int s, int t, int M, int both_tails, int threads,
                  double budget_s, int32_t* out, double* elapsed) {
    return orc_batch_ck(bases, offsets, n, pats, P, k, motif_len, no_bp, min_len, cutoff, W, s, t, M, both_tails, threads, budget_s,
                        out, elapsed, NULL, 0);
}
