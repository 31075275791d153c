/*
 * topsicle_io.h -- C ABI of libtopsicle_io.so: the host half of the ingest boundary (no HIP, no GPU): FASTA / FASTQ (plain,
 * gzip, BGZF) decoded by a thread team straight into the packed batch format of include/topsicle_hip.h, the second-pass packer
 * of the two-pass route, the writer of passing records and the CRC-32 of the raw-count archive.
 *
 * What it replaces in the reference (jaeyoungchoilab/Topsicle, file:line relative to the reference root): Bio.SeqIO behind
 * check_file_type / unzip_file (Topsicle/allsteps.py:36-50, 127-149) -- called once per read and step upstream
 * (allsteps.py:174, 257, 381) -- and the filtered-file rewrite with SeqIO.write (Topsicle/main.py:68-86).
 *
 * Conventions: extern "C", plain pointers and sizes; the caller owns every buffer; functions that return int64_t give a count or a
 * negative code, the others 0 / -1; tps_io_last_error() holds the message of the calling thread's last failure.  A reader handle
 * is used by one thread at a time.  The library reads no environment variables (tps_io_set_option).
 */
#ifndef TOPSICLE_IO_H
#define TOPSICLE_IO_H

#include <stdint.h>

#include "topsicle_hip.h" /* tps_read_desc */

#ifdef __cplusplus
extern "C" {
#endif

const char* tps_io_last_error(void);

/* ---- reader ---------------------------------------------------------------------------------------------------------------- */
/* Opens a FASTA / FASTQ file, plain or gzip'ed (ordinary gzip and BGZF are told apart by their headers).  The format comes from
 * the first character of the first line, like check_file_type (allsteps.py:36-50).  0, or -1 (unreadable, unknown format). */
int  tps_reader_open(const char* path, void** out);
/* One reader per BYTE RANGE of a plain (uncompressed) file: the records that START in [lo, hi) -- the first one found like a team
 * thread finds the first record of its stretch, the last one decoded to its end beyond hi.  Readers over adjacent ranges partition
 * the file's records; each runs a team of `threads` threads (0 = default) at the same time as the others: ONE big file feeds every
 * GPU of the node (the reference's answer to "> 20 GB and / or > 1 million reads" is splitting the file by hand, README.md:267-268).
 * tps_reader_range_info: where this reader's first record began and where it stopped -- reader i must have stopped where reader
 * i + 1 began (the caller's seam check; FASTQ framing is a heuristic only at a range's first record).  Compressed input: -1. */
int  tps_reader_open_range(const char* path, int64_t lo, int64_t hi, int32_t threads, void** out);
int  tps_reader_range_info(void* reader, int64_t* first_record, int64_t* stopped_at);
/* 0 = empty file, 1 = FASTA, 2 = FASTQ. */
int  tps_reader_format(void* reader);
void tps_reader_close(void* reader);
/* ASCII batches -- what SeqIO.parse yields (allsteps.py:127-149), for inputs the packed decoder declines and for the per-read API:
 * up to max_records records, bases back to back (bases_cap bytes) with n + 1 offsets, the header lines without their first
 * character back to back in heads (heads_cap bytes) with n + 1 head_off, quals (may be NULL; FASTQ only) laid out like bases.
 * Returns the number of records, 0 at the end, -1 on a parse error, -2 if one record does not fit the empty buffers. */
int64_t tps_reader_next(void* reader, uint8_t* bases, int64_t bases_cap, int64_t* offsets, int64_t max_records, char* heads,
                        int64_t heads_cap, int64_t* head_off, uint8_t* quals);
/* Packed batches: the records' bases leave as seq2 / inv / desc (include/topsicle_hip.h "Packed batch format"; words_cap words),
 * ready for tps_batch_upload_packed; spans (may be NULL) receives 4 entries per record -- header offset, header length, sequence
 * offset, quality offset (FASTA: end of the sequence text) -- into the input's text (the mmap'ed file, or the window handed out by
 * tps_reader_text_hold), for writing passing records back out without a copy.  Returns like tps_reader_next, and -4 when this
 * input cannot be decoded in packed mode from here on (records with blank or padded lines inside, lone-CR line ends): the caller
 * goes on with tps_reader_next at the same record.  *n_words = words used. */
int64_t tps_reader_next_packed(void* reader, uint32_t* seq2, uint16_t* inv, int64_t words_cap, tps_read_desc* desc, int64_t max_records,
                               char* heads, int64_t heads_cap, int64_t* head_off, int64_t* spans, int64_t* n_words);
/* The same in HEADS mode (first pass of the two-pass route): a read longer than 2 heads_bp arrives as its first + last heads_bp
 * bases -- all patternTRC_count looks at (allsteps.py:176-177) -- with its own length in full_len[i]. */
int64_t tps_reader_next_heads(void* reader, int32_t heads_bp, uint32_t* seq2, uint16_t* inv, int64_t words_cap, tps_read_desc* desc,
                              int64_t max_records, char* heads, int64_t heads_cap, int64_t* head_off, int64_t* spans, int32_t* full_len,
                              int64_t* n_words);
/* Compressed input: the window of inflated text the LAST packed batch's spans point into (*text, *len) and a reference on it
 * (*hold; NULL for a plain file, whose text is the file itself), to be given back with tps_text_release when the batch's records
 * have been written. */
int  tps_reader_text_hold(void* reader, const char** text, int64_t* len, void** hold);
void tps_text_release(void* hold);

/* ---- packers --------------------------------------------------------------------------------------------------------------- */
/* Second pass of the two-pass route: for the n reads idx[0 .. n) of a heads-mode batch, the part step 2 scans -- the first
 * (tails[j] = 0) or last (1) min(length, maxlen) bases (allsteps.py:263-271) -- packed from the batch's text as reads of their
 * own.  Returns the words used, -2 if words_cap is too small, -1 on a bad span. */
int64_t tps_pack_spans(const char* text, int64_t text_len, int32_t fasta, const int64_t* spans, const int32_t* full_len, const int64_t* idx,
                       const uint8_t* tails, int64_t n, int32_t maxlen, uint32_t* seq2, uint16_t* inv, tps_read_desc* desc, int64_t words_cap);
/* An ASCII batch (bases + n + 1 offsets) into the packed format with the thread team; seq2 / inv hold
 * tps_packed_words_total(offsets, n) words (inv may be NULL).  Returns the words written or -1. */
int64_t tps_packed_words_total(const int64_t* offsets, int64_t n);
int64_t tps_pack_reads(const uint8_t* bases, const int64_t* offsets, int64_t n, uint32_t* seq2, uint16_t* inv, tps_read_desc* desc,
                       int32_t nthreads);

/* ---- writers --------------------------------------------------------------------------------------------------------------- */
/* The records idx[0 .. n) of a packed FASTQ batch to `fd` in SeqIO.write's layout (main.py:83-86): writev straight from the
 * text the spans point into.  Returns the bytes written or -1. */
int64_t tps_write_fastq_spans(int fd, const char* text, int64_t text_len, const int64_t* spans, const int32_t* lens, const int64_t* idx, int64_t n);
/* The same at an explicit file offset (pwritev; several threads write different batches of one file at once) and the bytes a set
 * of records takes there: header + 2 x bases + 6 each, whatever the input's line layout. */
int64_t tps_write_fastq_spans_at(int fd, int64_t file_off, const char* text, int64_t text_len, const int64_t* spans, const int32_t* lens,
                                 const int64_t* idx, int64_t n);
int64_t tps_fastq_spans_bytes(const int64_t* spans, const int32_t* lens, const int64_t* idx, int64_t n);
/* zlib's crc32(crc, p, n) by carry-less multiplication (the checksum callback of tps_batch_raw_to_fd) and crc32_combine. */
uint32_t tps_crc32(uint32_t crc, const uint8_t* p, int64_t n);
uint32_t tps_crc32_combine(uint32_t crc1, uint32_t crc2, int64_t len2);

/* ---- tests and diagnostics -------------------------------------------------------------------------------------------------- */
/* Process-wide options; readers opened afterwards see them: "threads" (0 = by the host's CPUs and cgroup quota), "timing" (phase
 * times on stderr), "bgzf_group" (bytes of inflated text per refill; 0 = 128 MiB), "pack_min_span" (text below this many bytes
 * is decoded by one thread; -1 = 4 MiB), "no_pargz" (ordinary gzip through zlib's one stream), "pargz_min" (smallest .gz the
 * team inflates; -1 = 1 MiB).  topsicle_amd.seqio applies $TOPSICLE_IO_DEBUG = "key=value,..." when it loads the library. */
int  tps_io_set_option(const char* key, int64_t value);
/* The whole text of a gzip file through the parallel inflater: bytes (written to out up to cap) or -1; stats[0..2] = chunks
 * tried, speculative chunks accepted, chunks redone serially. */
int64_t tps_gz_inflate(const char* path, uint8_t* out, int64_t cap, int32_t threads, int64_t want, int64_t* stats);

#ifdef __cplusplus
}
#endif
#endif /* TOPSICLE_IO_H */
