// mapcaller_amd/csrc/mcx_files.cpp — files in, SAM text out, around mcx_map_batch (host only).
//
// Replaces the reading and writing halves of the reference's ReadMapping() loop: GetNextChunk /
// gzGetNextChunk (src/GetData.cpp:85-140) and Generate{Paired,Single}SamStream + fprintf
// (src/SamReport.cpp:324-488, src/ReadMapping.cpp:536-546).  The reference does both per 200-read
// chunk under locks; at GPU mapping rates they are the wall, so here
//   * each input file has its own parser thread that splits big blocks into lines (plain files and
//     .gz through zlib) and delivers flat arrays — no per-read allocation;
//   * batches of up to max_batch_reads flow through a three-stage pipeline (parse | map on the GPU |
//     format + write) so that the stages overlap;
//   * SAM lines of a batch are formatted by a pool of host threads into per-slice buffers and
//     written in input order.
// Text semantics follow the reference byte for byte: header trimming (GetData.cpp:3-20), the last
// byte of a FASTQ sequence line dropped (:48-53), multi-line FASTA for plain files (:56-77), the
// 1024-byte line buffer, the '@'/'>' check and single-line FASTA of the .gz reader (:101-128), an
// odd tail chunk of interleaved input mapped as single reads (ReadMapping.cpp:442).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <zlib.h>

#include "../../include/mcx.h"
#include "mcx_internal.h"

using namespace mcx;

namespace {

// ---- input -------------------------------------------------------------------------------------------
struct Entries { // reads of one file for one batch, flat
    std::vector<char> names; std::vector<uint32_t> name_off;
    std::vector<uint8_t> seq; std::vector<uint32_t> seq_off;
    std::vector<char> qual; // FASTQ: parallel to seq (NUL-padded where the quality line was shorter)
    bool last = false;      // the file ended (or delivered an empty read) after these
    std::string error;
    uint32_t n() const { return (uint32_t)name_off.size() - 1; }
    void clear() { names.clear(); name_off.assign(1, 0); seq.clear(); seq_off.assign(1, 0); qual.clear(); last = false; error.clear(); }
};

template <typename T> class Queue { // bounded hand-over between two stages
public:
    explicit Queue(size_t cap) : cap_(cap) {}
    void push(T v) { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return q_.size() < cap_; }); q_.push_back(std::move(v)); cv_.notify_all(); }
    T pop() { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return !q_.empty(); }); T v = std::move(q_.front()); q_.pop_front(); cv_.notify_all(); return v; }
private:
    std::mutex m_; std::condition_variable cv_; std::deque<T> q_; size_t cap_;
};

class Parser {
public:
    bool open(const std::string &path, std::string &err)
    {
        gz_mode_ = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0; // ReadMapping.cpp:709
        gz_ = gzopen(path.c_str(), "rb");
        if (!gz_) { err = "cannot open " + path; return false; }
        gzbuffer(gz_, 1 << 20);
        buf_.resize(1 << 24);
        // reading (and inflating) runs ahead of the line splitter on a thread of its own
        for (int k = 0; k < 4; k++) { std::unique_ptr<Block> b(new Block); b->d.resize(kBlockBytes); free_.push(std::move(b)); }
        feeder_ = std::thread([this] {
            for (;;) {
                std::unique_ptr<Block> b = free_.pop();
                int got = stop_.load() ? 0 : gzread(gz_, b->d.data(), (unsigned)kBlockBytes);
                b->n = got > 0 ? (size_t)got : 0;
                const bool end = b->n == 0;
                ready_.push(std::move(b));
                if (end) break;
            }
        });
        fill();
        fastq_ = end_ > 0 && buf_[0] == '@'; // CheckReadFormat, GetData.cpp:22-31
        return true;
    }
    ~Parser()
    {
        if (feeder_.joinable()) {
            stop_.store(true);
            while (!eof_) { std::unique_ptr<Block> b = ready_.pop(); if (b->n == 0) eof_ = true; else free_.push(std::move(b)); }
            feeder_.join();
        }
        if (gz_) gzclose(gz_);
    }
    bool fastq() const { return fastq_; }

    // appends up to `want` reads; false once the input is exhausted (Entries::last set)
    bool take(Entries &e, uint32_t want, int max_len)
    {
        for (uint32_t i = 0; i < want; i++) {
            if (!entry(e, max_len)) { e.last = true; return false; }
        }
        return true;
    }

private:
    enum : size_t { kBlockBytes = 8u << 20 };
    struct Block { std::vector<char> d; size_t n = 0; };
    gzFile gz_ = nullptr;
    bool gz_mode_ = false, fastq_ = true, eof_ = false;
    std::vector<char> buf_;
    size_t pos_ = 0, end_ = 0;
    Queue<std::unique_ptr<Block>> ready_{4}, free_{4};
    std::thread feeder_;
    std::atomic<bool> stop_{false};

    void fill() // one more block of input behind what is left of the buffer
    {
        if (eof_) return;
        if (pos_ > 0) { memmove(buf_.data(), buf_.data() + pos_, end_ - pos_); end_ -= pos_; pos_ = 0; }
        std::unique_ptr<Block> b = ready_.pop();
        if (b->n == 0) { eof_ = true; return; }
        if (end_ + b->n > buf_.size()) buf_.resize(std::max(buf_.size() * 2, end_ + b->n));
        memcpy(buf_.data() + end_, b->d.data(), b->n);
        end_ += b->n;
        free_.push(std::move(b));
    }

    // next line including its '\n' (getline); the .gz reader's gzgets(buffer, 1024) cuts at 1023 bytes
    bool line(const char *&p, size_t &len, bool consume = true)
    {
        for (;;) {
            const size_t avail = end_ - pos_;
            const size_t lim = gz_mode_ ? std::min<size_t>(avail, 1023) : avail;
            const char *nl = lim ? (const char *)memchr(buf_.data() + pos_, '\n', lim) : nullptr;
            if (nl) { p = buf_.data() + pos_; len = (size_t)(nl - p) + 1; break; }
            if (gz_mode_ && avail >= 1023) { p = buf_.data() + pos_; len = 1023; break; }
            if (eof_) { if (avail == 0) return false; p = buf_.data() + pos_; len = avail; break; }
            fill();
        }
        if (consume) pos_ += len;
        return true;
    }

    // IdentifyHeaderBegPos / IdentifyHeaderEndPos, GetData.cpp:3-20
    static void header_of(const char *l, int len, int &p1, int &p2)
    {
        const int lim = len > 100 ? 100 : len;
        p1 = len - 1; p2 = lim - 1;
        for (int i = 1; i < len; i++) if (l[i] != '>' && l[i] != '@') { p1 = i; break; }
        for (int i = 1; i < lim; i++) if (l[i] == ' ' || l[i] == '/' || !isprint((unsigned char)l[i])) { p2 = i; break; }
    }

    bool entry(Entries &e, int max_len)
    {
        const char *p; size_t len;
        if (!line(p, len)) return false;
        if (gz_mode_) { // gzGetNextEntry :101-128 (strlen semantics: a line is a C string)
            len = strnlen(p, len);
            if (len == 0 || (p[0] != '@' && p[0] != '>')) return false;
        }
        int p1, p2;
        header_of(p, (int)len, p1, p2);
        const size_t name_at = e.names.size();
        if (p2 > p1) e.names.insert(e.names.end(), p + p1, p + p2);
        const size_t seq_at = e.seq.size();
        size_t rlen = 0;
        if (fastq_ || gz_mode_) {
            if (!line(p, len)) { e.names.resize(name_at); return false; }
            if (gz_mode_) len = strnlen(p, len);
            rlen = len ? len - 1 : 0; // the last byte of the line is dropped (GetData.cpp:48-53, :113)
            e.seq.insert(e.seq.end(), (const uint8_t *)p, (const uint8_t *)p + rlen);
            if (fastq_) {
                const char *q; size_t ql;
                line(q, ql);
                if (!line(q, ql)) ql = 0;
                if (gz_mode_) ql = strnlen(q, ql);
                const size_t take = std::min(ql, rlen);
                e.qual.insert(e.qual.end(), q, q + take);
                e.qual.insert(e.qual.end(), rlen - take, '\0'); // strncpy pads with NUL
            }
        } else { // plain FASTA: every line up to the next header (GetData.cpp:56-77)
            while (line(p, len, false)) {
                if (p[0] == '>') break;
                pos_ += len;
                e.seq.insert(e.seq.end(), (const uint8_t *)p, (const uint8_t *)p + len - 1);
            }
            rlen = e.seq.size() - seq_at;
        }
        if (rlen == 0) { e.names.resize(name_at); e.seq.resize(seq_at); return false; } // `.rlen == 0` ends the input (GetData.cpp:91)
        if ((int)rlen > max_len) { e.error = "read " + std::string(e.names.data() + name_at, e.names.size() - name_at) + " is longer than max_read_len"; return false; }
        e.name_off.push_back((uint32_t)e.names.size());
        e.seq_off.push_back((uint32_t)e.seq.size());
        return true;
    }
};

// ---- a batch on its way through the stages -------------------------------------------------------------
struct Batch {
    Entries in[2];
    uint32_t n = 0;          // reads
    uint64_t number = 0;     // position of the batch in the input stream
    bool two_files = false, fastq = true, last = false;
    // interleaved reads as mcx_map_batch wants them, and its results: pinned host memory, allocated once per batch object
    uint8_t *bases = nullptr; uint32_t *off = nullptr; AlnRec *recs = nullptr; uint32_t *cig = nullptr;
    size_t cap_reads = 0, cap_bases = 0;
    std::vector<uint8_t> is_mate2;                          // mapped as the second read of a pair
    std::vector<uint32_t> cig_ext;                          // operations past a row of cig (mcx_cigar_ext), AlnRec::pad[0] = offset
    bool reserve(size_t reads, size_t n_bases)
    {
        if (reads > cap_reads) {
            mcx_pinned_free(off); mcx_pinned_free(recs); mcx_pinned_free(cig);
            cap_reads = reads;
            off = (uint32_t *)mcx_pinned_alloc((reads + 1) * sizeof(uint32_t));
            recs = (AlnRec *)mcx_pinned_alloc(reads * sizeof(AlnRec));
            cig = (uint32_t *)mcx_pinned_alloc(reads * MCX_CIGAR_STRIDE * sizeof(uint32_t));
        }
        if (n_bases > cap_bases) { mcx_pinned_free(bases); cap_bases = n_bases + n_bases / 8; bases = (uint8_t *)mcx_pinned_alloc(cap_bases); }
        return off && recs && cig && bases;
    }
    ~Batch() { mcx_pinned_free(bases); mcx_pinned_free(off); mcx_pinned_free(recs); mcx_pinned_free(cig); }
    std::string error;
    // read r of the batch -> (file, index in that file's entries)
    void locate(uint32_t r, int &f, uint32_t &i) const { if (two_files) { f = (int)(r & 1); i = r >> 1; } else { f = 0; i = r; } }
};

template <typename F> void parallel_for(uint32_t n, int threads, F f) // f(begin, end, slice)
{
    const int t = (int)std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)threads, (n + 4095) / 4096));
    if (t == 1) { f(0u, n, 0); return; }
    std::vector<std::thread> pool;
    for (int k = 0; k < t; k++) pool.emplace_back([=] { f((uint32_t)((uint64_t)n * k / t), (uint32_t)((uint64_t)n * (k + 1) / t), k); });
    for (auto &th : pool) th.join();
}

// ---- SAM text (GeneratePairedSamStream / GenerateSingleSamStream, SamReport.cpp:324-488) --------------------
inline char comp_char(char c) // GetComplementaryBase, tools.cpp:3-18
{
    switch (c) {
    case 'A': case 'a': return 'T';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    case 'T': case 't': return 'A';
    default: return 'N';
    }
}

struct Text { // writer over a buffer sized beforehand from an upper bound
    std::vector<char> b;
    char *w = nullptr;
    void start(size_t bound) { if (b.size() < bound) b.resize(bound); w = b.data(); }
    size_t size() const { return (size_t)(w - b.data()); }
    void put(const char *p, size_t n) { memcpy(w, p, n); w += n; }
    void put(char c) { *w++ = c; }
    void lit(const char *s) { put(s, strlen(s)); }
    void num(long long v)
    {
        char t[24]; int n = 0;
        unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
        do { t[n++] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) t[n++] = '-';
        while (n) *w++ = t[--n];
    }
};

// bytes one SAM line can take at most
inline size_t sam_bound(const HostIndex &ix, size_t name_len, size_t rlen, int chr, int n_cigar)
{
    return name_len + 2 * rlen + (chr >= 0 ? ix.chr_name[chr].size() : 1) + 11 * (size_t)(n_cigar > 0 ? n_cigar : 0) + 160;
}

void sam_record(const HostIndex &ix, const Batch &bt, uint32_t r, Text &o)
{
    static const char opc[8] = {'M', 'I', 'D', 'N', 'S', 'H', 'P', '='};
    int f; uint32_t i;
    bt.locate(r, f, i);
    const Entries &e = bt.in[f];
    const AlnRec &rec = bt.recs[r];
    const uint32_t *cigar = bt.cig + (size_t)r * MCX_CIGAR_STRIDE;
    const char *seq = (const char *)e.seq.data() + e.seq_off[i];
    const int rlen = (int)(e.seq_off[i + 1] - e.seq_off[i]);
    const char *qual = bt.fastq ? e.qual.data() + e.seq_off[i] : nullptr;
    o.put(e.names.data() + e.name_off[i], e.name_off[i + 1] - e.name_off[i]);
    const bool mapped = rec.chr >= 0;
    // The reference reverse-complements mate 2 in place before mapping (ReadMapping.cpp:451) and prints
    // that string for forward-strand hits and unmapped reads, its reverse complement otherwise.
    const bool flipped = bt.is_mate2[r] != 0;
    const bool again = mapped && rec.fwd == 0; // a second reverse complement for the output
    o.put('\t'); o.num(rec.flag); o.put('\t');
    if (!mapped) o.lit("*\t0\t0\t*\t*\t0\t0\t");
    else {
        const std::string &cn = ix.chr_name[rec.chr];
        o.put(cn.data(), cn.size()); o.put('\t'); o.num(rec.pos); o.put('\t'); o.num(rec.mapq); o.put('\t');
        for (int k = 0; k < rec.n_cigar; k++) {
            const uint32_t w = k < MCX_CIGAR_STRIDE ? cigar[k] : bt.cig_ext[(size_t)rec.pad[0] + (size_t)(k - MCX_CIGAR_STRIDE)];
            o.num(w >> 4); o.put(opc[w & 7]);
        }
        if (rec.has_mate) { o.lit("\t=\t"); o.num(rec.mate_pos); o.put('\t'); o.num(rec.tlen); o.put('\t'); }
        else o.lit("\t*\t0\t0\t");
    }
    if (!flipped && !again) o.put(seq, (size_t)rlen);
    else if (flipped != again) { char *w = o.w; for (int k = rlen - 1; k >= 0; k--) *w++ = comp_char(seq[k]); o.w = w; }
    else { char *w = o.w; for (int k = 0; k < rlen; k++) *w++ = comp_char(comp_char(seq[k])); o.w = w; } // complemented twice: upper case, N for anything else
    o.put('\t');
    if (!qual) o.put('*');
    else if (flipped == again) o.put(qual, strnlen(qual, (size_t)rlen)); // printed with %s: stops at a NUL pad
    else { char *w = o.w; for (int k = rlen - 1; k >= 0 && qual[k] != '\0'; k--) *w++ = qual[k]; o.w = w; }
    if (!mapped) o.lit("\tAS:i:0\tXS:i:0\n");
    else { o.lit("\tNM:i:"); o.num(rec.nm); o.lit("\tAS:i:"); o.num(rec.as); o.lit("\tXS:i:"); o.num(rec.xs); o.put('\n'); }
}

} // namespace

extern "C" void mcx_file_opts_default(mcx_file_opts *o) { memset(o, 0, sizeof *o); }

extern "C" int mcx_map_files_ex(mcx_ctx *c, const char *fq1, const char *fq2, const mcx_file_opts *fo, const char *sam_path, mcx_stats *stats)
{
    if (!c || !fq1) return mcx_set_error(MCX_ERR_ARG, "mcx_map_files: null argument");
    mcx_file_opts opt;
    mcx_file_opts_default(&opt);
    if (fo) opt = *fo;
    const mcx_index *idx = mcx_ctx_index(c);
    const HostIndex &hix = idx->host;
    const int max_len = mcx_ctx_max_read_len(c);
    const bool two = fq2 && fq2[0];
    const bool paired = two || opt.interleaved_pairs;
    int threads = opt.host_threads > 0 ? opt.host_threads : (int)std::min<unsigned>(32, std::max<unsigned>(1, std::thread::hardware_concurrency() / 2));
    std::string err;
    Parser ps[2];
    if (!ps[0].open(fq1, err)) return mcx_set_error(MCX_ERR_IO, err);
    if (two && !ps[1].open(fq2, err)) return mcx_set_error(MCX_ERR_IO, err);
    if (two && ps[0].fastq() != ps[1].fastq()) return mcx_set_error(MCX_ERR_IO, std::string(fq1) + " and " + fq2 + " are with different format");
    FILE *sam = nullptr;
    if (sam_path && sam_path[0]) {
        sam = strcmp(sam_path, "-") == 0 ? stdout : fopen(sam_path, opt.append_sam ? "a" : "w");
        if (!sam) return mcx_set_error(MCX_ERR_IO, std::string("cannot write ") + sam_path);
        setvbuf(sam, nullptr, _IOFBF, 1 << 22);
        if (!opt.append_sam && !opt.no_sam_header) { std::string hdr; sam_header(hix, hdr); fputs(hdr.c_str(), sam); }
    }
    FILE *sam_index = nullptr; // "batch number, bytes" per batch written: lets the parts of a sharded run be merged in input order
    if (sam && opt.sam_index_path && opt.sam_index_path[0]) {
        sam_index = fopen(opt.sam_index_path, "w");
        if (!sam_index) return mcx_set_error(MCX_ERR_IO, std::string("cannot write ") + opt.sam_index_path);
    }
    const uint64_t batch_reads = std::max<uint64_t>(kReadChunkSize, mcx_ctx_max_reads(c) / kReadChunkSize * kReadChunkSize);
    int64_t local_avg[4];
    mcx_avg_init(local_avg);
    int64_t *avg = opt.avg_state ? opt.avg_state : local_avg;

    // busy seconds per stage (MCX_TIMING=1 prints them)
    double t_parse = 0, t_pack = 0, t_map = 0, t_format = 0, t_write = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    typedef std::unique_ptr<Batch> BatchPtr;
    Queue<BatchPtr> parsed(2), mapped(2), spare(4); // batch objects circulate: their buffers are allocated (and faulted in) once
    for (int k = 0; k < 4; k++) spare.push(BatchPtr(new Batch));

    // stage 1: parse.  One thread per file fills its half of the batch.
    std::atomic<bool> abort(false);
    const uint64_t shard_count = opt.shard_count > 1 ? (uint64_t)opt.shard_count : 1, shard_rank = shard_count > 1 ? (uint64_t)opt.shard_rank : 0;
    std::thread reader([&] {
        bool done = false;
        uint64_t number = 0;
        while (!done) {
            BatchPtr b = spare.pop();
            const auto t0 = now();
            b->two_files = two; b->fastq = ps[0].fastq(); b->n = 0; b->last = false; b->error.clear(); b->number = number;
            const uint32_t per_file = (uint32_t)(two ? batch_reads / 2 : batch_reads);
            b->in[0].clear(); b->in[1].clear();
            if (two) {
                std::thread t2([&] { ps[1].take(b->in[1], per_file, max_len); });
                ps[0].take(b->in[0], per_file, max_len);
                t2.join();
                // the reference stops at the first empty read of file 1 and takes whatever file 2 holds (GetData.cpp:91-93)
                if (b->in[1].n() < b->in[0].n()) b->error = std::string(fq2) + " holds fewer reads than " + fq1;
                b->n = 2 * b->in[0].n();
                done = b->in[0].last;
            } else {
                ps[0].take(b->in[0], per_file, max_len);
                b->n = b->in[0].n();
                done = b->in[0].last;
            }
            for (int f = 0; f < 2; f++) if (!b->in[f].error.empty()) b->error = b->in[f].error;
            if (!b->error.empty() || abort.load()) done = true;
            // batches are dealt to the shards in turn; another shard's batch is parsed (the stream has to be
            // walked) and dropped — unless it carries the end of the input or an error, which every shard must see
            const bool mine = number % shard_count == shard_rank;
            number++;
            t_parse += secs(t0, now());
            if (!mine && !done) { spare.push(std::move(b)); continue; }
            if (!mine && b->error.empty()) b->n = 0;
            b->last = done;
            parsed.push(std::move(b));
        }
    });

    // stage 3: format + write
    int write_rc = 0;
    std::thread writer([&] {
        std::vector<Text> slices;
        for (;;) {
            BatchPtr b = mapped.pop();
            if (sam && b->n && write_rc == 0) {
                const auto t0 = now();
                slices.resize((size_t)threads);
                for (auto &s : slices) s.w = nullptr;
                parallel_for(b->n, threads, [&](uint32_t lo, uint32_t hi, int k) {
                    Text &t = slices[(size_t)k];
                    size_t bound = 0;
                    for (uint32_t r = lo; r < hi; r++) {
                        int f; uint32_t i;
                        b->locate(r, f, i);
                        const Entries &e = b->in[f];
                        bound += sam_bound(hix, e.name_off[i + 1] - e.name_off[i], e.seq_off[i + 1] - e.seq_off[i], b->recs[r].chr, b->recs[r].n_cigar);
                    }
                    t.start(bound);
                    for (uint32_t r = lo; r < hi; r++) sam_record(hix, *b, r, t);
                });
                const auto t1 = now();
                size_t bytes = 0;
                for (const Text &t : slices)
                    if (t.w && t.size()) { bytes += t.size(); if (fwrite(t.b.data(), 1, t.size(), sam) != t.size()) write_rc = MCX_ERR_IO; }
                if (sam_index) fprintf(sam_index, "%llu %zu\n", (unsigned long long)b->number, bytes);
                t_format += secs(t0, t1); t_write += secs(t1, now());
            }
            const bool stop = b->last;
            spare.push(std::move(b));
            if (stop) break;
        }
    });

    // stage 2 (this thread): interleave, map on the GPU
    int rc = 0;
    for (;;) {
        BatchPtr b = parsed.pop();
        const bool last = b->last;
        if (rc == 0 && !b->error.empty()) rc = mcx_set_error(b->error.find("max_read_len") != std::string::npos ? MCX_ERR_UNSUPPORTED : MCX_ERR_IO, b->error);
        if (rc) b->n = 0;
        const uint32_t n = b->n;
        if (rc == 0 && n) {
            const auto t0 = now();
            size_t total = 0;
            for (int f = 0; f < 2; f++) total += b->in[f].seq.size();
            if (!b->reserve(std::max<size_t>(n, batch_reads), std::max<size_t>(total + 64, batch_reads * 160))) { rc = mcx_set_error(MCX_ERR_DEVICE, "cannot allocate pinned host memory"); b->n = 0; }
            t_pack += secs(t0, now());
        }
        if (rc == 0 && n) {
            const auto t0 = now();
            b->off[0] = 0;
            if (two) for (uint32_t i = 0; i < n / 2; i++) { // mates alternate
                b->off[2 * i + 1] = b->off[2 * i] + (b->in[0].seq_off[i + 1] - b->in[0].seq_off[i]);
                b->off[2 * i + 2] = b->off[2 * i + 1] + (b->in[1].seq_off[i + 1] - b->in[1].seq_off[i]);
            } else memcpy(b->off, b->in[0].seq_off.data(), ((size_t)n + 1) * sizeof(uint32_t));
            if (two) parallel_for(n, threads, [&](uint32_t lo, uint32_t hi, int) {
                for (uint32_t r = lo; r < hi; r++) { int f; uint32_t i; b->locate(r, f, i); memcpy(b->bases + b->off[r], b->in[f].seq.data() + b->in[f].seq_off[i], b->off[r + 1] - b->off[r]); }
            });
            else memcpy(b->bases, b->in[0].seq.data(), b->off[n]);
            memset(b->bases + b->off[n], 0, 64);
            b->is_mate2.assign(n, 0);
            const auto t1 = now();
            t_pack += secs(t0, t1);
            uint32_t n_pairs_reads = paired ? n : 0;
            if (paired && (n & 1)) n_pairs_reads = n / kReadChunkSize * kReadChunkSize;
            if (n_pairs_reads) {
                rc = mcx_map_batch(c, b->bases, b->off, n_pairs_reads, 1, avg, (mcx_aln *)b->recs, b->cig, stats);
                for (uint32_t r = 1; r < n_pairs_reads; r += 2) b->is_mate2[r] = 1;
            }
            b->cig_ext.clear();
            auto take_ext = [&](uint32_t first, uint32_t last) { // long CIGARs of the reads just mapped
                const uint32_t *w = nullptr; uint64_t nw = 0;
                int e = mcx_cigar_ext(c, 0, &w, &nw);
                if (e || nw == 0) return e;
                const size_t base = b->cig_ext.size();
                b->cig_ext.insert(b->cig_ext.end(), w, w + nw);
                if (base) for (uint32_t r = first; r < last; r++) if (b->recs[r].n_cigar > MCX_CIGAR_STRIDE) b->recs[r].pad[0] += (int32_t)base;
                return 0;
            };
            if (rc == 0 && n_pairs_reads) rc = take_ext(0, n_pairs_reads);
            if (rc == 0 && n_pairs_reads < n) {
                std::vector<uint32_t> off2(b->off + n_pairs_reads, b->off + n + 1);
                const uint32_t base = off2[0];
                for (auto &x : off2) x -= base;
                rc = mcx_map_batch(c, b->bases + base, off2.data(), n - n_pairs_reads, 0, avg, (mcx_aln *)b->recs + n_pairs_reads,
                                   b->cig + (size_t)n_pairs_reads * MCX_CIGAR_STRIDE, stats);
                if (rc == 0) rc = take_ext(n_pairs_reads, n);
            }
            t_map += secs(t1, now());
            if (rc) b->n = 0;
        }
        if (rc) abort.store(true);
        b->last = last; // after an error the remaining batches pass through empty until the parser's last one
        mapped.push(std::move(b));
        if (last) break;
    }
    writer.join();
    reader.join();
    if (getenv("MCX_TIMING"))
        fprintf(stderr, "[mcx_map_files] busy seconds: parse %.2f | pack %.2f map %.2f | format %.2f write %.2f  (%d host threads)\n", t_parse, t_pack, t_map, t_format,
                t_write, threads);
    if (sam_index) fclose(sam_index);
    if (sam && sam != stdout) { if (fclose(sam) != 0 && write_rc == 0) write_rc = MCX_ERR_IO; }
    else if (sam) fflush(sam);
    if (rc == 0 && write_rc) rc = mcx_set_error(MCX_ERR_IO, std::string("cannot write ") + (sam_path ? sam_path : ""));
    return rc;
}

extern "C" int mcx_map_files(mcx_ctx *c, const char *fq1, const char *fq2, const char *sam_path, mcx_stats *stats)
{
    return mcx_map_files_ex(c, fq1, fq2, nullptr, sam_path, stats);
}
