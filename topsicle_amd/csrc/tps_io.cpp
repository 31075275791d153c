// tps_io.cpp -- libtopsicle_io.so: native streaming FASTA / FASTQ (.gz) reader for the host side.
//
// Replaces the reference's Biopython parse (Bio.SeqIO.parse over gzip.open, allsteps.py:127-149,
// main.py:83-86) on the way INTO the GPU path: records are decoded straight into the caller's batch
// buffers in the layout tps_batch_upload() takes (concatenated ASCII bases + n+1 offsets), so no
// per-read Python object is created for the ~99 % of reads that are never written back out.
// Record ids follow Biopython: the first whitespace-delimited token of the header line.
//
// Plain C ABI (ctypes-loadable), zlib only.  Build: g++ -O2 -shared -fPIC tps_io.cpp -lz
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
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

    bool fill() {
        if (eof) return false;
        if (pos < len) memmove(buf.data(), buf.data() + pos, len - pos);
        len -= pos;
        pos = 0;
        int got = gzread(gz, buf.data() + len, (unsigned)(buf.size() - len));
        if (got < 0) { g_err = "gzread failed"; eof = true; return false; }
        if (got == 0) { eof = true; return len > 0; }
        len += (size_t)got;
        return true;
    }
    // next line without the trailing \r\n; false at end of input
    bool getline(std::string& out) {
        out.clear();
        for (;;) {
            char* nl = (char*)memchr(buf.data() + pos, '\n', len - pos);
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
            return true;
        }
        // FASTA
        if (!have_next_head) {
            for (;;) {
                if (!getline(line)) return false;
                if (!line.empty() && line[0] == '>') break;
            }
            next_head.assign(line, 1, std::string::npos);
        }
        p_head = next_head;
        have_next_head = false;
        p_seq.clear();
        p_qual.clear();
        while (getline(line)) {
            if (!line.empty() && line[0] == '>') {
                next_head.assign(line, 1, std::string::npos);
                have_next_head = true;
                break;
            }
            strip(line);
            p_seq += line;
        }
        return true;
    }
};

}  // namespace

extern "C" {

const char* tps_io_last_error(void) { return g_err.c_str(); }

// Opens a FASTA/FASTQ file (plain or .gz -- zlib reads both transparently).  The format comes from
// the first byte, like check_file_type (allsteps.py:36-50).  Returns 0 or -1.
int tps_reader_open(const char* path, void** out) {
    if (!path || !out) { g_err = "null argument"; return -1; }
    *out = nullptr;
    gzFile gz = gzopen(path, "rb");
    if (!gz) { g_err = std::string("cannot open ") + path; return -1; }
    gzbuffer(gz, 1 << 20);
    Reader* r = new Reader();
    r->gz = gz;
    r->buf.resize(4 << 20);
    if (!r->fill() && r->len == 0) { r->format = 0; *out = r; return 0; }      // empty file: no records
    size_t i = 0;
    while (i < r->len && (r->buf[i] == '\n' || r->buf[i] == '\r' || r->buf[i] == ' ')) ++i;
    char c = i < r->len ? r->buf[i] : 0;
    r->format = c == '>' ? 1 : c == '@' ? 2 : 0;
    if (!r->format) {
        g_err = "format cannot be identified (first character is neither '>' nor '@')";
        gzclose(gz);
        delete r;
        return -1;
    }
    *out = r;
    return 0;
}

int tps_reader_format(void* h) { return h ? ((Reader*)h)->format : 0; }

void tps_reader_close(void* h) {
    if (!h) return;
    Reader* r = (Reader*)h;
    if (r->gz) gzclose(r->gz);
    delete r;
}

// Decodes up to max_records records into the caller's buffers, stopping before a record that would
// overflow bases_cap / heads_cap.  offsets / head_off get n+1 entries (offsets[0] = 0).  quals may be
// NULL; otherwise it receives the quality strings in the same layout as bases (FASTQ only; records
// whose quality length differs from the sequence length are padded with '!' / truncated).
// Returns the number of records (0 at end of input), -1 on a parse error, -2 if a single record
// does not fit into empty buffers.
int64_t tps_reader_next(void* h, uint8_t* bases, int64_t bases_cap, int64_t* offsets, int64_t max_records, char* heads,
                        int64_t heads_cap, int64_t* head_off, uint8_t* quals) {
    Reader* r = (Reader*)h;
    if (!r || !bases || !offsets || !heads || !head_off) { g_err = "null argument"; return -1; }
    int64_t n = 0, nb = 0, nh = 0;
    offsets[0] = 0;
    head_off[0] = 0;
    if (!r->format) return 0;
    g_err.clear();
    while (n < max_records) {
        if (!r->have_pending) {
            if (!r->parse_one()) {
                if (!g_err.empty()) return -1;
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

}  // extern "C"
