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
    uint32_t n_pair_reads = 0;                              // reads [0, n_pair_reads) were mapped as pairs, the rest one by one: two CIGAR pools
    bool reserve(size_t reads, size_t n_bases)
    {
        if (reads > cap_reads) {
            mcx_pinned_free(off); mcx_pinned_free(recs); mcx_pinned_free(cig);
            cap_reads = reads;
            off = (uint32_t *)mcx_pinned_alloc((reads + 1) * sizeof(uint32_t));
            recs = (AlnRec *)mcx_pinned_alloc(reads * sizeof(AlnRec));
            cig = (uint32_t *)mcx_pinned_alloc((MCX_CIGAR_POOL_WORDS(reads) + MCX_CIGAR_SLACK) * sizeof(uint32_t)); // (two pools: the pairs', the single reads')
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
    // the batch's CIGAR pool (the single-read part of a batch has one of its own behind the pairs'), AlnRec::pad[0] = offset
    const uint32_t *cigar = bt.cig + (r < bt.n_pair_reads ? 0 : MCX_CIGAR_POOL_WORDS(bt.n_pair_reads)) + (size_t)rec.pad[0];
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
            const uint32_t w = cigar[k];
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

// ---- exchange between the host threads of one process (mapcaller-mi355x -gpus N) ------------------------
namespace {
struct Rendezvous {
    std::mutex m; std::condition_variable cv;
    int size = 0, arrived = 0, left = 0;
    uint64_t gen = 0, gen_out = 0;
    std::vector<const void *> ptr;
};
struct LocalPeer { Rendezvous *rv; int rank; };

int local_allgather(void *user, const void *send, void *recv, uint64_t bytes)
{
    LocalPeer *p = (LocalPeer *)user;
    Rendezvous &rv = *p->rv;
    {
        std::unique_lock<std::mutex> l(rv.m);
        rv.ptr[(size_t)p->rank] = send;
        const uint64_t g = rv.gen;
        if (++rv.arrived == rv.size) { rv.arrived = 0; rv.gen++; rv.cv.notify_all(); }
        else rv.cv.wait(l, [&] { return rv.gen != g; });
    }
    for (int r = 0; r < rv.size; r++) memcpy((uint8_t *)recv + (size_t)r * bytes, rv.ptr[(size_t)r], bytes);
    { // nobody's send buffer may change before everyone has copied it
        std::unique_lock<std::mutex> l(rv.m);
        const uint64_t g = rv.gen_out;
        if (++rv.left == rv.size) { rv.left = 0; rv.gen_out++; rv.cv.notify_all(); }
        else rv.cv.wait(l, [&] { return rv.gen_out != g; });
    }
    return 0;
}
} // namespace

extern "C" int mcx_exchange_local(int32_t size, mcx_exchange *out)
{
    if (size < 1 || !out) return mcx_set_error(MCX_ERR_ARG, "mcx_exchange_local: bad argument");
    Rendezvous *rv = new Rendezvous();
    rv->size = size; rv->ptr.assign((size_t)size, nullptr);
    LocalPeer *peers = new LocalPeer[(size_t)size];
    for (int r = 0; r < size; r++) {
        peers[r].rv = rv; peers[r].rank = r;
        out[r].user = &peers[r]; out[r].rank = r; out[r].size = size; out[r].allgather = local_allgather;
    }
    return 0;
}

extern "C" void mcx_exchange_local_free(mcx_exchange *first)
{
    if (!first || !first->user) return;
    LocalPeer *peers = (LocalPeer *)first->user; // (rank 0's entry is the head of the array)
    delete peers[0].rv;
    delete[] peers;
    first->user = nullptr;
}

// The parts of a sharded run back into input order (batch k was written by shard k % parts): <sam>.part<r> with
// <sam>.part<r>.idx ("batch bytes" per line); part 0 starts with the header.  The parts are removed.
extern "C" int mcx_sam_merge(const char *sam_path, int32_t parts)
{
    if (!sam_path || parts < 1) return mcx_set_error(MCX_ERR_ARG, "mcx_sam_merge: bad argument");
    const std::string base(sam_path);
    struct Part { FILE *f = nullptr; std::vector<std::pair<uint64_t, uint64_t>> idx; size_t next = 0; };
    std::vector<Part> ps((size_t)parts);
    auto close_all = [&] { for (Part &q : ps) if (q.f) fclose(q.f); };
    for (int r = 0; r < parts; r++) {
        const std::string pn = base + ".part" + std::to_string(r);
        FILE *ix = fopen((pn + ".idx").c_str(), "r");
        ps[(size_t)r].f = fopen(pn.c_str(), "rb");
        if (!ix || !ps[(size_t)r].f) { if (ix) fclose(ix); close_all(); return mcx_set_error(MCX_ERR_IO, "cannot read " + pn); }
        unsigned long long k, b;
        while (fscanf(ix, "%llu %llu", &k, &b) == 2) ps[(size_t)r].idx.push_back(std::make_pair((uint64_t)k, (uint64_t)b));
        fclose(ix);
    }
    FILE *out = fopen(sam_path, "wb");
    if (!out) { close_all(); return mcx_set_error(MCX_ERR_IO, "cannot write " + base); }
    std::vector<char> buf(1 << 22);
    auto copy = [&](FILE *f, uint64_t n) {
        while (n) {
            const size_t want = (size_t)std::min<uint64_t>(n, buf.size()), got = fread(buf.data(), 1, want, f);
            if (got == 0 || fwrite(buf.data(), 1, got, out) != got) return false;
            n -= got;
        }
        return true;
    };
    bool ok = true;
    { // the header: what part 0 holds before its first batch
        fseek(ps[0].f, 0, SEEK_END);
        uint64_t total = (uint64_t)ftell(ps[0].f), body = 0;
        fseek(ps[0].f, 0, SEEK_SET);
        for (auto &e : ps[0].idx) body += e.second;
        ok = total >= body && copy(ps[0].f, total - body);
    }
    for (uint64_t k = 0; ok; k++) { // batches in input order
        Part &q = ps[(size_t)(k % (uint64_t)parts)];
        if (q.next >= q.idx.size()) {
            // batches without output (an empty tail) are not listed: done when no part has anything left
            bool any = false;
            for (Part &o : ps) if (o.next < o.idx.size()) any = true;
            if (!any) break;
            continue;
        }
        if (q.idx[q.next].first != k) continue;
        ok = copy(q.f, q.idx[q.next].second);
        q.next++;
    }
    close_all();
    if (fclose(out) != 0) ok = false;
    if (!ok) return mcx_set_error(MCX_ERR_IO, "cannot merge the parts of " + base);
    for (int r = 0; r < parts; r++) { const std::string pn = base + ".part" + std::to_string(r); remove(pn.c_str()); remove((pn + ".idx").c_str()); }
    return 0;
}

// ---- one round of a run spread over several shards ------------------------------------------------------
// Round j holds batches j*N .. j*N+N-1, one per shard.  The shards exchange (a) what each has in the round,
// (b) per-chunk pair sums until the ONE insert-size trajectory of the input stream (ReadMapping.cpp:462,
// :538-539) has been walked over all of them and no shard had to re-run a pair, (c) with -vcf, the duplicate-check
// keys, so that the cap admits reads in input order across shards (AlignmentProfile.cpp:76-77).  Every shard makes
// the same sequence of exchange calls whatever it holds; a failing shard keeps taking part until the round's
// next message has told the others.
namespace {
struct Shards {
    const mcx_exchange *x;
    uint32_t slot_stride;   // reads a batch holds at most
    uint32_t cap_chunks;
    std::vector<uint8_t> recv;
    std::vector<uint32_t> msg;
    std::vector<uint64_t> all_keys, pad_keys;
    struct Head { int32_t rc; uint32_t n_pair, n_single, last; };

    int gather(const void *send, size_t bytes)
    {
        recv.resize(bytes * (size_t)x->size);
        return x->allgather(x->user, send, recv.data(), bytes) ? mcx_set_error(MCX_ERR_DEVICE, "the exchange between the shards failed") : 0;
    }
    // any shard's failure ends the run on all of them
    int agree(int my_rc)
    {
        int32_t v = my_rc;
        if (int e = gather(&v, sizeof v)) return e;
        if (my_rc) return my_rc;
        for (int r = 0; r < x->size; r++) { int32_t o; memcpy(&o, recv.data() + (size_t)r * sizeof o, sizeof o); if (o) return mcx_set_error(o, "shard " + std::to_string(r) + " failed"); }
        return 0;
    }

    // closes a part of the round: -vcf bookkeeping with the keys of every shard, or the plain end
    int finish_part(mcx_ctx *c, bool mine, bool profile, mcx_stats *stats, int rc)
    {
        if (!profile) { if (rc == 0 && mine) rc = mcx_batch_end(c, stats); return agree(rc); }
        const uint64_t *keys = nullptr; uint64_t nk = 0;
        if (rc == 0 && mine) rc = mcx_batch_end_keys(c, stats, &keys, &nk);
        if (rc) nk = 0;
        struct { int32_t rc; uint32_t pad; uint64_t n; } h = {rc, 0, nk}, o;
        if (int e = gather(&h, sizeof h)) return e;
        uint64_t most = 0, total = 0;
        std::vector<uint64_t> cnt((size_t)x->size);
        int bad = rc;
        for (int r = 0; r < x->size; r++) { memcpy(&o, recv.data() + (size_t)r * sizeof o, sizeof o); cnt[(size_t)r] = o.n; most = std::max(most, o.n); total += o.n; if (!bad && o.rc) bad = mcx_set_error(o.rc, "shard " + std::to_string(r) + " failed"); }
        if (bad) return bad;
        if (most == 0) { if (mine) rc = mcx_batch_accumulate(c, nullptr, 0, slot_stride, (uint32_t)x->rank); return agree(rc); }
        pad_keys.assign((size_t)most, ~0ull);
        for (uint64_t i = 0; i < nk; i++) pad_keys[(size_t)i] = keys[i] + (uint64_t)x->rank * slot_stride; // the read's number within the round
        if (int e = gather(pad_keys.data(), (size_t)most * sizeof(uint64_t))) return e;
        all_keys.clear(); all_keys.reserve((size_t)total);
        for (int r = 0; r < x->size; r++) {
            const uint64_t *p = (const uint64_t *)(recv.data() + (size_t)r * (size_t)most * sizeof(uint64_t));
            all_keys.insert(all_keys.end(), p, p + cnt[(size_t)r]);
        }
        rc = mcx_batch_accumulate(c, all_keys.data(), all_keys.size(), slot_stride, mine ? (uint32_t)x->rank : 0xFFFFFFFFu);
        return agree(rc);
    }

    // The paired part of a round.  n = this shard's reads (0: none); avg = the run's state {avgDist, pairs, distance, reads}.
    int pairs(mcx_ctx *c, const uint8_t *bases, const uint32_t *off, uint32_t n, int64_t read_base, int64_t avg[4], bool profile,
              mcx_aln *aln, uint32_t *cig, mcx_stats *stats)
    {
        int rc = 0;
        const uint8_t *d_bases = nullptr; const uint32_t *d_off = nullptr; mcx_aln *d_aln = nullptr; uint32_t *d_cig = nullptr;
        if (n) {
            rc = mcx_stage_in(c, bases, off, n, &d_bases, &d_off, &d_aln, &d_cig);
            if (rc == 0) rc = mcx_batch_begin(c, d_bases, d_off, n, 1, (int32_t)((uint32_t)avg[0] * 1.5), read_base, d_aln, d_cig, stats);
        }
        const size_t words = 4 + 2 * (size_t)cap_chunks;
        msg.assign(words, 0);
        std::vector<int32_t> est(cap_chunks);
        uint32_t n_redo = 0xFFFFFFFFu; // "not replayed yet"
        int64_t st[3] = {avg[0], avg[1], avg[2]};
        for (int iter = 0;; iter++) {
            uint32_t nc = 0;
            const uint32_t *ok = nullptr, *ds = nullptr;
            if (rc == 0 && n) rc = mcx_batch_sums(c, &nc, &ok, &ds, nullptr);
            if (rc == 0 && nc > cap_chunks) rc = mcx_set_error(MCX_ERR_ARG, "a batch holds more chunks than the shards agreed on");
            msg[0] = (uint32_t)rc; msg[1] = rc ? 0 : nc; msg[2] = n_redo; msg[3] = 0;
            if (rc == 0 && nc) { memcpy(&msg[4], ok, nc * 4); memcpy(&msg[4 + cap_chunks], ds, nc * 4); }
            if (int e = gather(msg.data(), words * 4)) return e;
            bool settled = iter > 0;
            for (int r = 0; r < x->size; r++) {
                const uint32_t *m = (const uint32_t *)(recv.data() + (size_t)r * words * 4);
                if (m[0]) return rc ? rc : mcx_set_error((int32_t)m[0], "shard " + std::to_string(r) + " failed");
                if (m[1] && m[2]) settled = false;
            }
            // the trajectory over the round's batches in input order; this shard keeps the estimates of its own chunks
            st[0] = avg[0]; st[1] = avg[1]; st[2] = avg[2];
            for (int r = 0; r < x->size; r++) {
                const uint32_t *m = (const uint32_t *)(recv.data() + (size_t)r * words * 4);
                mcx_avg_walk(st, m + 4, m + 4 + cap_chunks, m[1], r == x->rank ? est.data() : nullptr);
            }
            if (settled) break;
            if (iter == 255) return mcx_set_error(MCX_ERR_CAPACITY, "avgDist replay did not converge");
            n_redo = 0;
            if (n) rc = mcx_batch_replay(c, est.data(), &n_redo, stats);
        }
        avg[0] = st[0]; avg[1] = st[1]; avg[2] = st[2];
        rc = finish_part(c, n != 0, profile, stats, 0);
        if (rc == 0 && n) rc = mcx_stage_out(c, n, aln, cig);
        return rc;
    }

    // reads mapped one by one (single-end libraries, the odd tail of an interleaved file): no trajectory
    int singles(mcx_ctx *c, const uint8_t *bases, const uint32_t *off, uint32_t n, int64_t read_base, bool profile, mcx_aln *aln, uint32_t *cig,
                mcx_stats *stats)
    {
        int rc = 0;
        const uint8_t *d_bases = nullptr; const uint32_t *d_off = nullptr; mcx_aln *d_aln = nullptr; uint32_t *d_cig = nullptr;
        if (n) {
            rc = mcx_stage_in(c, bases, off, n, &d_bases, &d_off, &d_aln, &d_cig);
            if (rc == 0) rc = mcx_batch_begin(c, d_bases, d_off, n, 0, 0, read_base, d_aln, d_cig, stats);
        }
        rc = finish_part(c, n != 0 && rc == 0, profile, stats, rc);
        if (rc == 0 && n) rc = mcx_stage_out(c, n, aln, cig);
        return rc;
    }
};
} // namespace

extern "C" int mcx_map_files_ex(mcx_ctx *c, const char *fq1, const char *fq2, const mcx_file_opts *fo, const char *sam_path, mcx_stats *stats)
{
    if (!c || !fq1) return mcx_set_error(MCX_ERR_ARG, "mcx_map_files: null argument");
    mcx_file_opts opt;
    mcx_file_opts_default(&opt);
    if (fo) opt = *fo;
    if (opt.shard_count > 1 && (!opt.exchange || !opt.exchange->allgather || opt.exchange->size != opt.shard_count || opt.exchange->rank != opt.shard_rank))
        return mcx_set_error(MCX_ERR_ARG, "mcx_map_files_ex: a sharded run needs mcx_file_opts.exchange with the shard's rank and count");
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
    const bool sharded = shard_count > 1;
    const bool profile = mcx_ctx_has_profile(c);
    Shards sh;
    sh.x = opt.exchange; sh.slot_stride = (uint32_t)batch_reads; sh.cap_chunks = (uint32_t)(batch_reads / kReadChunkSize + 2);
    uint64_t rounds_done = 0;
    bool dead = false; // a shard failed and every shard knows
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
        uint32_t n_pairs_reads = 0;
        auto t1 = now();
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
            t1 = now();
            t_pack += secs(t0, t1);
            n_pairs_reads = paired ? n : 0;
            if (paired && (n & 1)) n_pairs_reads = n / kReadChunkSize * kReadChunkSize;
            for (uint32_t r = 1; r < n_pairs_reads; r += 2) b->is_mate2[r] = 1;
            b->n_pair_reads = n_pairs_reads;
        }
        // the single-read part (a single-end library, or the odd tail of an interleaved file) as a batch of its own
        std::vector<uint32_t> off2;
        uint32_t base2 = 0;
        if (rc == 0 && n_pairs_reads < n) {
            off2.assign(b->off + n_pairs_reads, b->off + n + 1);
            base2 = off2[0];
            for (auto &x : off2) x -= base2;
        }
        if (!sharded) {
            if (rc == 0 && n_pairs_reads) {
                rc = mcx_map_batch(c, b->bases, b->off, n_pairs_reads, 1, avg, (mcx_aln *)b->recs, b->cig, stats);
            }
            if (rc == 0 && n_pairs_reads < n) {
                rc = mcx_map_batch(c, b->bases + base2, off2.data(), n - n_pairs_reads, 0, avg, (mcx_aln *)b->recs + n_pairs_reads,
                                   b->cig + MCX_CIGAR_POOL_WORDS(n_pairs_reads), stats);
            }
        } else if (!dead && b->number / shard_count >= rounds_done) {
            rounds_done = b->number / shard_count + 1;
            Shards::Head h = {rc, rc ? 0u : n_pairs_reads, rc ? 0u : n - n_pairs_reads, last ? 1u : 0u};
            int e = sh.gather(&h, sizeof h);
            bool any_pair = false, any_single = false;
            uint64_t before = 0, round_total = 0;
            if (e) rc = e;
            else for (int r = 0; r < sh.x->size; r++) {
                Shards::Head o; memcpy(&o, sh.recv.data() + (size_t)r * sizeof o, sizeof o);
                if (o.rc && rc == 0) rc = mcx_set_error(o.rc, "shard " + std::to_string(r) + " failed");
                any_pair |= o.n_pair != 0; any_single |= o.n_single != 0;
                if (r < sh.x->rank) before += (uint64_t)o.n_pair + o.n_single;
                round_total += (uint64_t)o.n_pair + o.n_single;
            }
            const int64_t round_base = avg[3];
            if (rc == 0 && any_pair) {
                rc = sh.pairs(c, b->bases, b->off, n_pairs_reads, round_base + (int64_t)before, avg, profile, (mcx_aln *)b->recs, b->cig, stats);
            }
            if (rc == 0 && any_single) {
                const uint32_t ns = n - n_pairs_reads;
                rc = sh.singles(c, ns ? b->bases + base2 : nullptr, ns ? off2.data() : nullptr, ns, round_base + (int64_t)before + n_pairs_reads, profile,
                                (mcx_aln *)b->recs + n_pairs_reads, b->cig + MCX_CIGAR_POOL_WORDS(n_pairs_reads), stats);
            }
            avg[3] = round_base + (int64_t)round_total;
            if (rc) dead = true;
        }
        if (n) t_map += secs(t1, now());
        if (rc) { b->n = 0; abort.store(true); }
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
