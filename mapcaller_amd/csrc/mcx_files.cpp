// mapcaller_amd/csrc/mcx_files.cpp — files in, SAM text out, around the batch API (host only).
//
// Replaces the reading and writing halves of the reference's ReadMapping() loop: GetNextChunk /
// gzGetNextChunk (src/GetData.cpp:85-140) and Generate{Paired,Single}SamStream + fprintf
// (src/SamReport.cpp:324-488, src/ReadMapping.cpp:536-546).  The reference does both per 200-read
// chunk under locks; at GPU mapping rates they are the wall, so here
//   * a plain FASTQ file is mapped into memory and indexed by line count (a record is four lines from
//     the start of the file, whatever the lines hold — GetData.cpp:45-55), in parallel; a batch is then
//     parsed by a pool of host threads, each from the first byte of its own stretch of records, with
//     the reads' names, bases and qualities left where they lie in the file (the SAM formatter reads
//     them there) and the bases packed to 2 bits on the way (a quarter of the bytes cross PCIe); a
//     shard of a several-GPU run touches only the bytes of its own batches;
//   * .gz files and FASTA go through one sequential reader per file (zlib / multi-line records);
//   * batches flow through parse | copy in, map, copy out (three device slots: the copies of one
//     batch under the kernels of its neighbours) | format + write;
//   * SAM lines are formatted by a second pool into per-slice buffers and written with positioned
//     writes, every slice at its final place in the file — by every shard into the one output file:
//     the shards tell each other their batches' sizes with the rounds' other messages.
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
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include "mcx_pgz.h"
#include <immintrin.h>

#include "../../include/mcx.h"
#include "mcx_internal.h"
#include "mcx_cpus.h"

using namespace mcx;

namespace {

typedef std::chrono::steady_clock::time_point Tick;
inline Tick now() { return std::chrono::steady_clock::now(); }
inline double secs(Tick a, Tick b) { return std::chrono::duration<double>(b - a).count(); }

// ---- a pool of host threads that lives as long as the run ---------------------------------------------------
class Pool {
public:
    explicit Pool(int n) : n_(std::max(1, n))
    {
        for (int k = 1; k < n_; k++) th_.emplace_back([this] { work(); });
    }
    ~Pool()
    {
        { std::unique_lock<std::mutex> l(m_); stop_ = true; gen_++; cv_.notify_all(); }
        for (auto &t : th_) t.join();
    }
    int size() const { return n_; }
    // f(k) for k in [0, parts), the calling thread taking its share; returns when all are done.  One run() at a time per pool.
    void run(int parts, const std::function<void(int)> &f)
    {
        if (parts <= 0) return;
        if (parts == 1 || n_ == 1) { for (int k = 0; k < parts; k++) f(k); return; }
        {
            std::unique_lock<std::mutex> l(m_);
            f_ = &f; parts_ = parts; next_.store(0); left_ = parts; gen_++;
            cv_.notify_all();
        }
        drain();
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [&] { return left_ == 0; });
        f_ = nullptr;
    }
private:
    void drain()
    {
        for (;;) {
            const int k = next_.fetch_add(1);
            if (k >= parts_) break;
            (*f_)(k);
            std::unique_lock<std::mutex> l(m_);
            if (--left_ == 0) done_.notify_all();
        }
    }
    void work()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                if (!f_ || left_ == 0) continue;
            }
            drain();
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex m_; std::condition_variable cv_, done_;
    const std::function<void(int)> *f_ = nullptr;
    int parts_ = 0, left_ = 0;
    std::atomic<int> next_{0};
    uint64_t gen_ = 0;
    bool stop_ = false;
};

template <typename T> class Queue { // bounded hand-over between two stages
public:
    explicit Queue(size_t cap) : cap_(cap) {}
    void push(T v) { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return q_.size() < cap_; }); q_.push_back(std::move(v)); cv_.notify_all(); }
    T pop() { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return !q_.empty(); }); T v = std::move(q_.front()); q_.pop_front(); cv_.notify_all(); return v; }
    bool try_pop(T &v) { std::unique_lock<std::mutex> l(m_); if (q_.empty()) return false; v = std::move(q_.front()); q_.pop_front(); cv_.notify_all(); return true; }
private:
    std::mutex m_; std::condition_variable cv_; std::deque<T> q_; size_t cap_;
};

// ---- input -------------------------------------------------------------------------------------------
// One read as the formatter needs it: where its name, bases and qualities lie (offsets from View::base — the mapped file,
// or the batch's own copy for .gz / FASTA input).
struct Rec {
    uint64_t name, seq, qual;
    uint32_t rlen, q_take;  // q_take: bytes of the quality line that count (min(line, rlen), GetData.cpp:51-52; printed up to a NUL)
    uint32_t name_len;
};
// the records of a View: plain memory that is not cleared when it is handed out (a batch object's 40 MB of them would be written twice)
class RecBuf {
public:
    RecBuf() {}
    RecBuf(const RecBuf &) = delete;
    RecBuf &operator=(const RecBuf &) = delete;
    ~RecBuf() { free(p_); }
    size_t size() const { return n_; }
    void clear() { n_ = 0; }
    bool reserve(size_t n) { if (n > cap_) { Rec *q = (Rec *)realloc(p_, n * sizeof(Rec)); if (!q) return false; p_ = q; cap_ = n; } return true; }
    bool resize(size_t n) { if (!reserve(n)) return false; n_ = n; return true; } // (new entries are the caller's to write)
    bool push_back(const Rec &r) { if (n_ == cap_ && !reserve(cap_ ? cap_ * 2 : 4096)) return false; p_[n_++] = r; return true; } // false: out of memory — the caller says so (a record dropped in silence would shift mate 1 against mate 2)
    Rec &operator[](size_t i) { return p_[i]; }
    const Rec &operator[](size_t i) const { return p_[i]; }
    Rec *data() { return p_; }
    const Rec *begin() const { return p_; }
    const Rec *end() const { return p_ + n_; }
private:
    Rec *p_ = nullptr; size_t n_ = 0, cap_ = 0;
};
struct View { // the reads of one file for one batch
    const char *base = nullptr;
    RecBuf recs;
    std::vector<char> own;  // .gz / FASTA: the batch's copy of names, bases, qualities
    bool last = false;      // the file ended (or delivered an empty read) after these
    std::string error;
    uint32_t n() const { return (uint32_t)recs.size(); }
    void clear() { recs.clear(); own.clear(); last = false; error.clear(); base = nullptr; }
};

// IdentifyHeaderBegPos / IdentifyHeaderEndPos, GetData.cpp:3-20
inline void header_of(const char *l, int len, int &p1, int &p2)
{
    const int lim = len > 100 ? 100 : len;
    p1 = len - 1; p2 = lim - 1;
    for (int i = 1; i < len; i++) if (l[i] != '>' && l[i] != '@') { p1 = i; break; }
    for (int i = 1; i < lim; i++) { const unsigned char c = (unsigned char)l[i]; if (c <= ' ' || c == '/' || c >= 0x7f) { p2 = i; break; } } // (' ', '/', or not printable: isprint in the C locale is 0x20..0x7e)
}

// 2-bit row for mcx_stream_submit_packed: sixteen bases to a word, the first on top; bytes that are not ACGT are listed
inline uint32_t pack_word(const uint8_t *seq, uint32_t i, uint32_t n, uint32_t read, std::vector<uint64_t> &odd) // bases [i, i + n), n <= 16
{
    static const struct Lut { uint8_t v[256]; Lut() { memset(v, 4, sizeof v); v['A'] = 0; v['C'] = 1; v['G'] = 2; v['T'] = 3; } } lut;
    uint32_t w = 0, bad = 0;
    for (uint32_t j = 0; j < n; j++) { const uint32_t c = lut.v[seq[i + j]]; bad |= c; w |= (c & 3u) << (30 - 2 * j); }
    if (bad & 4u) {
        w = 0;
        for (uint32_t j = 0; j < n; j++) {
            const uint32_t c = lut.v[seq[i + j]];
            if (c > 3) odd.push_back(((uint64_t)read << 32) | ((uint64_t)(i + j) << 8) | seq[i + j]);
            else w |= c << (30 - 2 * j);
        }
    }
    return w;
}
inline void pack_row_plain(const uint8_t *seq, uint32_t rlen, uint32_t read, uint32_t *row, uint32_t row_words, std::vector<uint64_t> &odd)
{
    uint32_t k = 0;
    for (uint32_t i = 0; i < rlen; i += 16, k++) row[k] = pack_word(seq, i, rlen - i < 16 ? rlen - i : 16, read, odd);
    for (; k < row_words; k++) row[k] = 0;
}
// the same sixteen bases at a time: of A C G T, ((c >> 1) ^ (c >> 2)) & 3 is the code; the two-bit fields gathered by pext
__attribute__((target("sse2,bmi2"))) inline void pack_row_bmi2(const uint8_t *seq, uint32_t rlen, uint32_t read, uint32_t *row, uint32_t row_words, std::vector<uint64_t> &odd)
{
    const __m128i cA = _mm_set1_epi8('A'), cC = _mm_set1_epi8('C'), cG = _mm_set1_epi8('G'), cT = _mm_set1_epi8('T'), three = _mm_set1_epi8(3);
    uint32_t k = 0, i = 0;
    for (; i + 16 <= rlen; i += 16, k++) {
        const __m128i c = _mm_loadu_si128((const __m128i *)(seq + i));
        const __m128i known = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(c, cA), _mm_cmpeq_epi8(c, cC)), _mm_or_si128(_mm_cmpeq_epi8(c, cG), _mm_cmpeq_epi8(c, cT)));
        if (_mm_movemask_epi8(known) != 0xFFFF) { row[k] = pack_word(seq, i, 16, read, odd); continue; }
        const __m128i code = _mm_and_si128(_mm_xor_si128(_mm_srli_epi16(c, 1), _mm_srli_epi16(c, 2)), three); // (what the 16-bit shifts carry across bytes lands above bit 1)
        const uint64_t lo = (uint64_t)_mm_cvtsi128_si64(code), hi = (uint64_t)_mm_cvtsi128_si64(_mm_unpackhi_epi64(code, code));
        row[k] = ((uint32_t)_pext_u64(__builtin_bswap64(lo), 0x0303030303030303ull) << 16) | (uint32_t)_pext_u64(__builtin_bswap64(hi), 0x0303030303030303ull);
    }
    if (i < rlen) { row[k++] = pack_word(seq, i, rlen - i, read, odd); }
    for (; k < row_words; k++) row[k] = 0;
}
inline void pack_row(const uint8_t *seq, uint32_t rlen, uint32_t read, uint32_t *row, uint32_t row_words, std::vector<uint64_t> &odd)
{
    static const bool wide = __builtin_cpu_supports("bmi2") && !getenv("MCX_PLAIN_PACK");
    if (wide) pack_row_bmi2(seq, rlen, read, row, row_words, odd); else pack_row_plain(seq, rlen, read, row, row_words, odd);
}

// the next '\n' in [p, e), or nullptr: FASTQ lines are a few bytes to a few hundred, so sixteen bytes at a time from the first byte on (memchr's set-up costs more than the search)
inline const char *find_nl(const char *p, const char *e)
{
    const __m128i nl = _mm_set1_epi8('\n');
    while (p + 16 <= e) {
        const int m = _mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)p), nl));
        if (m) return p + __builtin_ctz((unsigned)m);
        p += 16;
    }
    for (; p < e; p++) if (*p == '\n') return p;
    return nullptr;
}

// A plain FASTQ file in memory, with the line count ahead of every 64 KB of it: record r begins at line 4 r.
class MappedFastq {
public:
    ~MappedFastq() { if (map_ && size_) munmap((void *)map_, size_); if (fd_ >= 0) close(fd_); }
    bool open(const std::string &path, std::string &err)
    {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) { err = "cannot open " + path; return false; }
        struct stat st;
        if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode)) { err = "cannot map " + path; return false; }
        size_ = (size_t)st.st_size;
        if (size_) {
            map_ = (const char *)mmap(nullptr, size_, PROT_READ, MAP_SHARED, fd_, 0);
            if (map_ == MAP_FAILED) { map_ = nullptr; err = "cannot map " + path; return false; }
            (void)madvise((void *)map_, size_, MADV_WILLNEED);
        }
        n_blocks_ = (size_ + kBlock - 1) / kBlock;
        cnt_.assign(n_blocks_ + 1, 0);
        return true;
    }
    const char *data() const { return map_; }
    size_t bytes() const { return size_; }
    size_t n_blocks() const { return n_blocks_; }
    // newlines of blocks [b0, b1) (the shards of a run count a share each and tell one another)
    void count(size_t b0, size_t b1, Pool &pool)
    {
        const size_t n = b1 > b0 ? b1 - b0 : 0;
        const int parts = (int)std::min<size_t>(n, (size_t)pool.size() * 4);
        pool.run(parts, [&](int k) {
            for (size_t b = b0 + n * (size_t)k / (size_t)parts; b < b0 + n * (size_t)(k + 1) / (size_t)parts; b++) {
                const char *p = map_ + b * kBlock, *e = map_ + std::min(size_, (b + 1) * kBlock);
                uint32_t c = 0;
                while (p < e) { const char *q = (const char *)memchr(p, '\n', (size_t)(e - p)); if (!q) break; c++; p = q + 1; }
                cnt_[b] = c;
            }
        });
    }
    uint32_t *counts() { return cnt_.data(); }
    void finish() // prefix sums; lines of the file (an unterminated last line counts, like getline's)
    {
        pre_.assign(n_blocks_ + 1, 0);
        for (size_t b = 0; b < n_blocks_; b++) pre_[b + 1] = pre_[b] + cnt_[b];
        lines_ = pre_[n_blocks_] + ((size_ && map_[size_ - 1] != '\n') ? 1 : 0);
    }
    uint64_t lines() const { return lines_; }
    // byte at which line L begins (the file's size when it has no such line)
    size_t line_start(uint64_t L) const
    {
        if (L == 0) return 0;
        size_t lo = 0, hi = n_blocks_; // the block that holds the L-th newline
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (pre_[mid + 1] < L) lo = mid + 1; else hi = mid; }
        if (lo >= n_blocks_) return size_;
        uint64_t need = L - pre_[lo];
        const char *p = map_ + lo * kBlock, *e = map_ + std::min(size_, (lo + 1) * kBlock);
        while (need) { const char *q = (const char *)memchr(p, '\n', (size_t)(e - p)); if (!q) return size_; p = q + 1; need--; }
        return (size_t)(p - map_);
    }
    // Records [r0, r1) into out[0 ..), their number in n_out; stops like GetNextEntry at a missing sequence line or an empty read (false then).
    bool parse(uint64_t r0, uint64_t r1, int max_len, Rec *out, size_t &n_out, std::string &err) const
    {
        n_out = 0;
        size_t p = line_start(4 * r0);
        auto line = [&](const char *&l, size_t &len) { // the next line with its '\n' (getline); false at the end of the file
            if (p >= size_) return false;
            l = map_ + p;
            const char *e = find_nl(l, map_ + size_);
            len = e ? (size_t)(e - l) + 1 : size_ - p;
            p += len;
            return true;
        };
        for (uint64_t r = r0; r < r1; r++) {
            const char *l; size_t len;
            if (!line(l, len)) return false;
            int p1, p2;
            header_of(l, (int)len, p1, p2);
            Rec rec; memset(&rec, 0, sizeof rec);
            rec.name = (uint64_t)(l - map_) + (uint64_t)p1; rec.name_len = p2 > p1 ? (uint32_t)(p2 - p1) : 0;
            if (!line(l, len)) return false; // no sequence line
            rec.seq = (uint64_t)(l - map_); rec.rlen = len ? (uint32_t)(len - 1) : 0; // the last byte of the line is dropped (GetData.cpp:48-53)
            const char *q; size_t ql;
            (void)line(q, ql);              // the '+' line
            if (!line(q, ql)) { ql = 0; q = map_; }
            rec.qual = (uint64_t)(q - map_); rec.q_take = (uint32_t)std::min<size_t>(ql, rec.rlen);
            if (rec.rlen == 0) return false; // `.rlen == 0` ends the input (GetData.cpp:91)
            if ((int)rec.rlen > max_len) { err = "read " + std::string(map_ + rec.name, rec.name_len) + " is longer than max_read_len"; return false; }
            out[n_out++] = rec;
        }
        return true;
    }
private:
    enum : size_t { kBlock = 64u << 10 };
    int fd_ = -1;
    const char *map_ = nullptr;
    size_t size_ = 0, n_blocks_ = 0;
    std::vector<uint32_t> cnt_;
    std::vector<uint64_t> pre_;
    uint64_t lines_ = 0;
};

// The sequential reader: .gz through zlib, FASTA (multi-line records).
class Parser {
public:
    bool open(const std::string &path, std::string &err)
    {
        gz_mode_ = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0; // ReadMapping.cpp:709
        for (int k = 0; k < 4; k++) { std::unique_ptr<Block> b(new Block); b->d.resize(kHead + kBlockBytes); free_.push(std::move(b)); } // (one with the feeder, one with the splitter, two on their way)
        if (gz_mode_ && map_bgzf(path)) {
            // BGZF (bgzip, samtools): a gzip file made of independent members of at most 64 KB, each saying how long it is — the
            // members of a stretch are inflated side by side by a few threads
            feeder_ = std::thread([this] { feed_bgzf(); });
        } else if (gz_mode_ && !getenv("MCX_GZ_SERIAL") && map_gz(path)) {
            // an ordinary gzip stream (what real FASTQ comes as): no entry points, so block starts are searched for and the stretches between them
            // inflated side by side against windows that are filled in afterwards (mcx_pgz.h) — zlib's one thread gives 0.5 GB/s of text per file
            feeder_ = std::thread([this] { feed_pgz(); });
        } else {
            gz_ = gzopen(path.c_str(), "rb");
            if (!gz_) { err = "cannot open " + path; return false; }
            gzbuffer(gz_, 1 << 20);
            // reading (and inflating) runs ahead of the line splitter on a thread of its own
            feeder_ = std::thread([this] {
                for (;;) {
                    std::unique_ptr<Block> b = free_.pop();
                    int got = stop_.load() ? 0 : gzread(gz_, b->text(), (unsigned)kBlockBytes);
                    b->n = got > 0 ? (size_t)got : 0;
                    b->look_for_nul();
                    const bool end = b->n == 0;
                    ready_.push(std::move(b));
                    if (end) break;
                }
            });
        }
        fill();
        fastq_ = end_ > pos_ && *pos_ == '@'; // CheckReadFormat, GetData.cpp:22-31
        return true;
    }
    ~Parser()
    {
        if (feeder_.joinable()) {
            stop_.store(true);
            if (cur_) { free_.push(std::move(cur_)); pos_ = end_ = nullptr; } // (the feeder may be waiting for a block to fill)
            while (!eof_) { std::unique_ptr<Block> b = ready_.pop(); if (b->n == 0) eof_ = true; else free_.push(std::move(b)); }
            feeder_.join();
        }
        if (gz_) gzclose(gz_);
        if (map_) munmap((void *)map_, map_size_);
    }
    bool fastq() const { return fastq_; }

    // appends up to `want` reads (copied into v.own); false once the input is exhausted (View::last set)
    bool take(View &v, uint32_t want, int max_len)
    {
        bool more = true;
        if (v.own.capacity() < (size_t)want * 64) v.own.reserve((size_t)want * (size_t)(own_per_rec_ + 16)); // (what the last batch's records took: no growth by doubling, no copies)
        const size_t own0 = v.own.size();
        uint32_t got = 0;
        for (uint32_t i = 0; i < want && more; i++) { if (!entry(v, max_len)) { v.last = true; more = false; } else got++; }
        if (got) own_per_rec_ = (v.own.size() - own0) / got + 1;
        v.base = v.own.data();
        return more;
    }

private:
    // A block of text on its way from the feeder to the line splitter: kBlockBytes of it behind kHead bytes of room, into which the splitter moves what the
    // block before left unfinished (a line's beginning, a record's first lines) — the lines are then cut where the feeder put them.  (Until round 6 every block
    // was copied once more, into the splitter's own buffer, and searched for a NUL there: both on the one thread per file that the .gz rate hangs on.)
    enum : size_t { kBlockBytes = 8u << 20, kHead = 64u << 10 };
    struct Block {
        std::vector<char> d; size_t n = 0; bool nul = false;
        char *text() { return d.data() + kHead; }
        void look_for_nul() { nul = n && memchr(text(), 0, n) != nullptr; } // (gzgets' lines are C strings: a NUL cuts one short — looked for per block, by the feeder)
    };
    gzFile gz_ = nullptr;
    bool gz_mode_ = false, fastq_ = true, eof_ = false, has_nul_ = false;
    size_t own_per_rec_ = 340; // bytes of names, bases and qualities a record of the last batch took
    std::unique_ptr<Block> cur_;             // the block the splitter is in
    const char *pos_ = nullptr, *end_ = nullptr; // what is left of it (with the carried-over bytes in front)
    std::vector<char> long_;                 // a line or record tail longer than kHead (a FASTA line of megabytes): block and tail put together here
    Queue<std::unique_ptr<Block>> ready_{4}, free_{4};
    std::thread feeder_;
    std::atomic<bool> stop_{false};
    const uint8_t *map_ = nullptr; // a BGZF file, mapped
    size_t map_size_ = 0;

    // a BGZF member at p (n bytes left in the file): its whole size and the length of its extra field; 0 if it is not one
    static size_t bgzf_member(const uint8_t *p, size_t n, size_t &xlen)
    {
        if (n < 28 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
        xlen = (size_t)p[10] | ((size_t)p[11] << 8);
        if (12 + xlen + 8 > n) return 0;
        for (size_t o = 12; o + 4 <= 12 + xlen;) { // the subfields of the extra field: SI1 SI2 SLEN(2) data
            const size_t slen = (size_t)p[o + 2] | ((size_t)p[o + 3] << 8);
            if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) {
                const size_t size = ((size_t)p[o + 4] | ((size_t)p[o + 5] << 8)) + 1;
                return (size >= 12 + xlen + 8 && size <= n) ? size : 0;
            }
            o += 4 + slen;
        }
        return 0;
    }
    bool map_bgzf(const std::string &path)
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 28;
        if (ok) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            ok = m != MAP_FAILED;
            if (ok) {
                size_t xlen = 0;
                if (bgzf_member((const uint8_t *)m, (size_t)st.st_size, xlen)) { map_ = (const uint8_t *)m; map_size_ = (size_t)st.st_size; }
                else { munmap(m, (size_t)st.st_size); ok = false; }
            }
        }
        close(fd);
        return ok;
    }
    bool map_gz(const std::string &path)
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 18;
        if (ok) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            ok = m != MAP_FAILED;
            if (ok) {
                if (pgz::gzip_header((const uint8_t *)m, (size_t)st.st_size)) { map_ = (const uint8_t *)m; map_size_ = (size_t)st.st_size; }
                else { munmap(m, (size_t)st.st_size); ok = false; } // (not a gzip file after all: zlib's reader passes such bytes through, as the reference's does)
            }
        }
        close(fd);
        return ok;
    }
    void feed_pgz()
    {
        Pool pool((int)std::max(2u, std::min(12u, mcx_usable_cpus() * 3 / 8)));
        pgz::Reader rd;
        pgz::Text text[2];
        bool have = rd.open(map_, map_size_, pool.size(), (size_t)2 << 20, [&](int n, const std::function<void(int)> &f) { pool.run(n, f); }) && rd.next(text[0]);
        for (int cur = 0; have && !stop_.load(); cur ^= 1) {
            // the next round is inflated while this round's text goes into the blocks (the pool stood still meanwhile: a tenth of the reader's time)
            bool more = false;
            std::thread ahead([&] { more = rd.next(text[cur ^ 1]); });
            const pgz::Text &t = text[cur];
            for (size_t o = 0; o < t.size() && !stop_.load();) {
                std::unique_ptr<Block> b = free_.pop();
                const size_t m = std::min<size_t>(t.size() - o, kBlockBytes);
                memcpy(b->text(), t.data() + o, m);
                b->n = m; o += m;
                b->look_for_nul();
                ready_.push(std::move(b));
            }
            ahead.join();
            have = more;
        }
        std::unique_ptr<Block> b = free_.pop(); // the end of the input (a damaged stream ends it where it stops making sense, as gzread's error does)
        b->n = 0;
        ready_.push(std::move(b));
    }
    void feed_bgzf()
    {
        Pool pool((int)std::max(2u, std::min(8u, mcx_usable_cpus() / 2)));
        struct Task { const uint8_t *src; uint32_t clen, isize, crc; size_t dst; };
        std::vector<Task> tasks;
        size_t o = 0;
        std::atomic<int> bad(0);
        bool last = false; // what follows is not a BGZF member: the input ends there, as it does where gzread gives up
        while (o < map_size_ && !stop_.load() && !bad.load() && !last) {
            std::unique_ptr<Block> b = free_.pop();
            tasks.clear();
            size_t total = 0;
            while (o < map_size_) { // as many members as a block of the pipe holds
                size_t xlen = 0;
                const uint8_t *p = map_ + o;
                const size_t size = bgzf_member(p, map_size_ - o, xlen);
                if (!size) { last = true; break; }
                const uint32_t isize = (uint32_t)p[size - 4] | ((uint32_t)p[size - 3] << 8) | ((uint32_t)p[size - 2] << 16) | ((uint32_t)p[size - 1] << 24);
                const uint32_t crc = (uint32_t)p[size - 8] | ((uint32_t)p[size - 7] << 8) | ((uint32_t)p[size - 6] << 16) | ((uint32_t)p[size - 5] << 24);
                if (isize > 65536) { last = true; break; }
                if (total + isize > kBlockBytes) break;
                Task t; t.src = p + 12 + xlen; t.clen = (uint32_t)(size - 12 - xlen - 8); t.isize = isize; t.crc = crc; t.dst = total;
                tasks.push_back(t);
                total += isize; o += size;
            }
            char *out = b->text();
            pool.run((int)tasks.size(), [&](int k) {
                const Task &t = tasks[(size_t)k];
                if (t.isize == 0) return; // (the empty member that ends a BGZF file)
                z_stream zs; memset(&zs, 0, sizeof zs);
                if (inflateInit2(&zs, -15) != Z_OK) { bad.store(1); return; }
                zs.next_in = const_cast<Bytef *>(t.src); zs.avail_in = t.clen;
                zs.next_out = (Bytef *)(out + t.dst); zs.avail_out = t.isize;
                const int rc = inflate(&zs, Z_FINISH);
                const bool ok = rc == Z_STREAM_END && zs.total_out == t.isize;
                inflateEnd(&zs);
                if (!ok || crc32(crc32(0L, Z_NULL, 0), (const Bytef *)(out + t.dst), t.isize) != t.crc) bad.store(1);
            });
            if (bad.load()) total = 0; // (a damaged stretch is not handed on)
            if (total == 0 && !bad.load() && !last && o < map_size_) { free_.push(std::move(b)); continue; } // (empty members in the middle of a file)
            b->n = total;
            b->look_for_nul();
            const bool end = total == 0;
            ready_.push(std::move(b));
            if (end) return;
        }
        std::unique_ptr<Block> b = free_.pop(); // the end of the input
        b->n = 0;
        ready_.push(std::move(b));
    }

    void fill() // one more block of input behind what is left of this one
    {
        if (eof_) return;
        std::unique_ptr<Block> b = ready_.pop();
        if (b->n == 0) { eof_ = true; free_.push(std::move(b)); return; } // (what is left stays where it is: pos_ .. end_)
        const size_t left = (size_t)(end_ - pos_);
        if (b->nul) has_nul_ = true;
        if (left <= kHead) {
            if (left) memcpy(b->text() - left, pos_, left);
            pos_ = b->text() - left; end_ = b->text() + b->n;
            if (cur_) free_.push(std::move(cur_));
            cur_ = std::move(b);
            long_.clear();
        } else { // (rare: more left over than a block has room for in front)
            std::vector<char> both(left + b->n);
            memcpy(both.data(), pos_, left);
            memcpy(both.data() + left, b->text(), b->n);
            long_.swap(both);
            pos_ = long_.data(); end_ = long_.data() + long_.size();
            if (cur_) free_.push(std::move(cur_));
            free_.push(std::move(b));
        }
    }

    // next line including its '\n' (getline); the .gz reader's gzgets(buffer, 1024) cuts at 1023 bytes
    bool line(const char *&p, size_t &len, bool consume = true)
    {
        for (;;) {
            const size_t avail = (size_t)(end_ - pos_);
            const size_t lim = gz_mode_ ? std::min<size_t>(avail, 1023) : avail;
            const char *nl = lim ? find_nl(pos_, pos_ + lim) : nullptr; // (lines of tens to hundreds of bytes: memchr's set-up costs more than the search)
            if (nl) { p = pos_; len = (size_t)(nl - p) + 1; break; }
            if (gz_mode_ && avail >= 1023) { p = pos_; len = 1023; break; }
            if (eof_) { if (avail == 0) return false; p = pos_; len = avail; break; }
            fill();
        }
        if (consume) pos_ += len;
        return true;
    }

    bool entry(View &v, int max_len)
    {
        const char *p; size_t len;
        if (!line(p, len)) return false;
        if (gz_mode_) { // gzGetNextEntry :101-128 (strlen semantics: a line is a C string)
            if (has_nul_) len = strnlen(p, len);
            if (len == 0 || (p[0] != '@' && p[0] != '>')) return false;
        }
        int p1, p2;
        header_of(p, (int)len, p1, p2);
        std::vector<char> &o = v.own;
        const size_t name_at = o.size();
        if (p2 > p1) o.insert(o.end(), p + p1, p + p2);
        Rec rec; memset(&rec, 0, sizeof rec);
        rec.name = name_at; rec.name_len = (uint32_t)(o.size() - name_at);
        const size_t seq_at = o.size();
        size_t rlen = 0;
        if (fastq_ || gz_mode_) {
            if (!line(p, len)) { o.resize(name_at); return false; }
            if (gz_mode_ && has_nul_) len = strnlen(p, len);
            rlen = len ? len - 1 : 0; // the last byte of the line is dropped (GetData.cpp:48-53, :113)
            o.insert(o.end(), p, p + rlen);
            if (fastq_) {
                const char *q; size_t ql;
                line(q, ql);
                if (!line(q, ql)) ql = 0;
                if (gz_mode_ && has_nul_) ql = strnlen(q, ql);
                const size_t take = std::min(ql, rlen);
                rec.qual = o.size(); rec.q_take = (uint32_t)take;
                o.insert(o.end(), q, q + take);
            }
        } else { // plain FASTA: every line up to the next header (GetData.cpp:56-77)
            while (line(p, len, false)) {
                if (p[0] == '>') break;
                pos_ += len;
                o.insert(o.end(), p, p + len - 1);
            }
            rlen = o.size() - seq_at;
        }
        rec.seq = seq_at; rec.rlen = (uint32_t)rlen;
        if (rlen == 0) { o.resize(name_at); return false; } // `.rlen == 0` ends the input (GetData.cpp:91)
        if ((int)rlen > max_len) { v.error = "read " + std::string(o.data() + name_at, rec.name_len) + " is longer than max_read_len"; return false; }
        if (!v.recs.push_back(rec)) { v.error = "out of memory for the batch's read records"; o.resize(name_at); return false; }
        return true;
    }
};

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
    size_t size() const { return w ? (size_t)(w - b.data()) : 0; }
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

// ---- a batch on its way through the stages -------------------------------------------------------------
struct Batch {
    View in[2];
    uint32_t n = 0;          // reads
    uint64_t number = 0;     // position of the batch in the input stream
    bool two_files = false, fastq = true, last = false;
    std::string error;
    // what crosses the device boundary, in page-locked memory (allocated once per batch object): 2-bit rows, lengths and the
    // bytes that are not ACGT on the way in; records and CIGAR words on the way out
    uint32_t *rows = nullptr, *lens = nullptr; uint64_t *odd = nullptr; mcx_aln32 *recs = nullptr; uint32_t *cig = nullptr; // (the records as they cross PCIe: 32 bytes each)
    size_t cap_reads = 0, cap_rows = 0, cap_odd = 0;
    uint32_t row_words = 0, n_odd[2] = {0, 0};
    std::vector<uint8_t> is_mate2;   // mapped as the second read of a pair
    uint32_t n_pair_reads = 0;       // reads [0, n_pair_reads) are mapped as pairs, the rest one by one: two parts, two CIGAR pools
    std::vector<Text> slices;        // the batch's SAM text
    uint64_t sam_bytes = 0;
    bool reserve(size_t reads, size_t words_per_read)
    {
        if (reads > cap_reads) {
            mcx_pinned_free(lens); mcx_pinned_free(recs); mcx_pinned_free(cig);
            cap_reads = reads;
            lens = (uint32_t *)mcx_pinned_alloc((reads + 1) * sizeof(uint32_t));
            recs = (mcx_aln32 *)mcx_pinned_alloc(reads * sizeof(mcx_aln32));
            cig = (uint32_t *)mcx_pinned_alloc((MCX_CIGAR_POOL_WORDS(reads) + MCX_CIGAR_SLACK) * sizeof(uint32_t)); // (two pools: the pairs', the single reads')
        }
        if (reads * words_per_read > cap_rows) { mcx_pinned_free(rows); cap_rows = reads * words_per_read; rows = (uint32_t *)mcx_pinned_alloc(cap_rows * sizeof(uint32_t)); }
        return lens && recs && cig && rows;
    }
    bool reserve_odd(size_t n)
    {
        if (n > cap_odd) { mcx_pinned_free(odd); cap_odd = n + n / 2 + 1024; odd = (uint64_t *)mcx_pinned_alloc(cap_odd * sizeof(uint64_t)); }
        return odd != nullptr;
    }
    ~Batch() { mcx_pinned_free(rows); mcx_pinned_free(lens); mcx_pinned_free(odd); mcx_pinned_free(recs); mcx_pinned_free(cig); }
    // read r of the batch -> (file, index in that file's records)
    const Rec &rec(uint32_t r, const char *&base) const
    {
        const int f = two_files ? (int)(r & 1) : 0;
        base = in[f].base;
        return in[f].recs[two_files ? r >> 1 : r];
    }
    int n_parts() const { return n == 0 ? 0 : (n_pair_reads ? 1 : 0) + (n_pair_reads < n ? 1 : 0); }
};

// bytes one SAM line can take at most
inline size_t sam_bound(const HostIndex &ix, size_t name_len, size_t rlen, int chr, int n_cigar)
{
    return name_len + 2 * rlen + (chr >= 0 ? ix.chr_name[chr].size() : 1) + 11 * (size_t)(n_cigar > 0 ? n_cigar : 0) + 160;
}

void sam_record(const HostIndex &ix, const Batch &bt, uint32_t r, Text &o)
{
    static const char opc[8] = {'M', 'I', 'D', 'N', 'S', 'H', 'P', '='};
    const char *base;
    const Rec &e = bt.rec(r, base);
    mcx_aln rec;
    mcx_aln_unpack(&bt.recs[r], &rec);
    // the batch's CIGAR pool (the single-read part of a batch has one of its own behind the pairs'), cigar_off = the read's place in it
    const uint32_t *cigar = bt.cig + (r < bt.n_pair_reads ? 0 : MCX_CIGAR_POOL_WORDS(bt.n_pair_reads)) + (size_t)(uint32_t)rec.cigar_off;
    const char *seq = base + e.seq;
    const int rlen = (int)e.rlen;
    const char *qual = bt.fastq ? base + e.qual : nullptr;
    o.put(base + e.name, e.name_len);
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
    else {
        // the quality string as the reference holds it: q_take bytes of the line, NUL from there to the read's length (strncpy);
        // printed with %s — and its reversed copy, when the line was short, begins with that NUL
        const size_t ql = strnlen(qual, (size_t)e.q_take);
        if (flipped == again) o.put(qual, ql);
        else if (e.q_take == e.rlen) { char *w = o.w; for (int k = rlen - 1; k >= 0 && qual[k] != '\0'; k--) *w++ = qual[k]; o.w = w; }
    }
    if (!mapped) o.lit("\tAS:i:0\tXS:i:0\n");
    else { o.lit("\tNM:i:"); o.num(rec.nm); o.lit("\tAS:i:"); o.num(rec.as); o.lit("\tXS:i:"); o.num(rec.xs); o.put('\n'); }
}

} // namespace

extern "C" void mcx_file_opts_default(mcx_file_opts *o) { memset(o, 0, sizeof *o); }

extern "C" uint32_t mcx_pack_row(const uint8_t *seq, uint32_t rlen, uint32_t read, uint32_t *row, uint32_t row_words, uint64_t *odd, uint32_t odd_cap, uint32_t *n_odd)
{
    std::vector<uint64_t> o;
    pack_row(seq, rlen, read, row, row_words, o);
    for (uint64_t v : o) if (odd && n_odd && *n_odd < odd_cap) odd[(*n_odd)++] = v;
    return (uint32_t)o.size();
}

extern "C" uint32_t mcx_host_cpus(void) { return mcx_usable_cpus(); }

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

// ---- one round of a run spread over several shards ------------------------------------------------------
// Round j holds batches j*N .. j*N+N-1, one per shard.  The shards exchange (a) what each has in the round,
// (b) per-chunk pair sums until the ONE insert-size trajectory of the input stream (ReadMapping.cpp:462,
// :538-539) has been walked over all of them and no shard had to re-run a pair, (c) with -vcf, the duplicate-check
// keys, so that the cap admits reads in input order across shards (AlignmentProfile.cpp:76-77), (d) the bytes of
// SAM text their batches of an earlier round came to, so that every shard writes at its final place.  Every shard
// makes the same sequence of exchange calls whatever it holds; a failing shard keeps taking part until the
// round's next message has told the others.
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

    // The paired part of a round.  n = this shard's reads (0: none), in HBM already; avg = the run's state {avgDist, pairs, distance, reads}.
    int pairs(mcx_ctx *c, const uint8_t *d_bases, const uint32_t *d_off, uint32_t n, int64_t read_base, int64_t avg[4], bool profile,
              mcx_aln *d_aln, uint32_t *d_cig, mcx_stats *stats)
    {
        int rc = 0;
        if (n) rc = mcx_batch_begin(c, d_bases, d_off, n, 1, (int32_t)((uint32_t)avg[0] * 1.5), read_base, d_aln, d_cig, stats);
        // What the shards tell each other per exchange: {status, chunks, pairs re-run, -, proper pairs, their summed distance} — totals, not the
        // chunks' sums: a shard walks its own chunks from the round's state plus the totals of the shards before it in input order (below).
        struct Msg { uint32_t rc, n_chunks, n_redo, pad; int64_t pairs, dist; } mine, o;
        uint32_t n_redo = 0xFFFFFFFFu; // "not replayed yet"
        int64_t st[3] = {avg[0], avg[1], avg[2]};
        std::vector<int32_t> est;
        for (int iter = 0;; iter++) {
            uint32_t nc = 0;
            int64_t tot[2] = {0, 0};
            const uint32_t *ok = nullptr, *ds = nullptr;
            if (rc == 0 && n) rc = mcx_batch_sums(c, &nc, &ok, &ds, nullptr);
            if (rc == 0 && n) rc = mcx_batch_totals(c, tot);
            mine.rc = (uint32_t)rc; mine.n_chunks = rc ? 0 : nc; mine.n_redo = n_redo; mine.pad = 0; mine.pairs = tot[0]; mine.dist = tot[1];
            if (int e = gather(&mine, sizeof mine)) return e;
            bool settled = iter > 0;
            int64_t before[3] = {avg[0], avg[1], avg[2]}, all_pairs = 0, all_dist = 0, all_chunks = 0;
            bool first = true; // no shard before this one holds a chunk
            for (int r = 0; r < x->size; r++) {
                memcpy(&o, recv.data() + (size_t)r * sizeof o, sizeof o);
                if (o.rc) return rc ? rc : mcx_set_error((int32_t)o.rc, "shard " + std::to_string(r) + " failed");
                if (o.n_chunks && o.n_redo) settled = false;
                if (r < x->rank) { before[1] += o.pairs; before[2] += o.dist; if (o.n_chunks) first = false; }
                all_pairs += o.pairs; all_dist += o.dist; all_chunks += o.n_chunks;
            }
            st[0] = avg[0]; st[1] = avg[1]; st[2] = avg[2];
            mcx_avg_advance(st, all_pairs, all_dist, all_chunks);
            if (settled) break;
            if (iter == 255) return mcx_set_error(MCX_ERR_CAPACITY, "avgDist replay did not converge");
            n_redo = 0;
            if (n) {
                // this shard's chunks walked HERE, from the round's state plus the totals of the shards before it (ReadMapping.cpp:462, :538-539): the
                // estimate a chunk is paired with is the state before it, re-estimated once a thousand proper pairs have been seen — not at the round's
                // very first chunk, whose estimate is the state the round began with.  The device checks every pair against the list and re-runs the
                // ones whose estimate moved (mcx_batch_replay).  (Round 5 left the walk to the device here — mcx_batch_check, closed form — with nothing
                // on the host to hold it against; the one-shard path has always compared the two.)
                est.resize(nc);
                uint32_t cur = (uint32_t)before[0];
                int64_t tp = before[1], td = before[2];
                for (uint32_t k = 0; k < nc; k++) {
                    if ((k > 0 || !first) && tp > 1000) cur = (uint32_t)(int)(1. * td / tp + .5);
                    est[k] = (int32_t)(cur * 1.5);
                    tp += ok[k]; td += ds[k];
                }
                rc = mcx_batch_replay(c, est.data(), &n_redo, stats);
            }
        }
        avg[0] = st[0]; avg[1] = st[1]; avg[2] = st[2];
        return finish_part(c, n != 0, profile, stats, 0);
    }

    // reads mapped one by one (single-end libraries, the odd tail of an interleaved file): no trajectory
    int singles(mcx_ctx *c, const uint8_t *d_bases, const uint32_t *d_off, uint32_t n, int64_t read_base, bool profile, mcx_aln *d_aln, uint32_t *d_cig,
                mcx_stats *stats)
    {
        int rc = 0;
        if (n) rc = mcx_batch_begin(c, d_bases, d_off, n, 0, 0, read_base, d_aln, d_cig, stats);
        return finish_part(c, n != 0 && rc == 0, profile, stats, rc);
    }
};

// what the formatter and the thread that talks to the other shards tell each other: sizes one way, places the other
struct Places {
    std::mutex m; std::condition_variable cv;
    std::map<uint64_t, uint64_t> size, place; // batch number -> bytes of its text; -> where it goes
    bool failed = false;
    void put_size(uint64_t k, uint64_t v) { std::unique_lock<std::mutex> l(m); size[k] = v; cv.notify_all(); }
    bool wait_size(uint64_t k, uint64_t &v) // false: the run has failed, there is no such size
    {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return failed || size.count(k); });
        if (!size.count(k)) return false;
        v = size[k]; size.erase(k);
        return true;
    }
    void put_place(uint64_t k, uint64_t v) { std::unique_lock<std::mutex> l(m); place[k] = v; cv.notify_all(); }
    // the place of batch k, or of nothing at all when the run has failed (false)
    bool wait_place(uint64_t k, uint64_t &v, bool block)
    {
        std::unique_lock<std::mutex> l(m);
        if (block) cv.wait(l, [&] { return failed || place.count(k); });
        auto it = place.find(k);
        if (it == place.end()) return false;
        v = it->second; place.erase(it);
        return true;
    }
    void fail() { std::unique_lock<std::mutex> l(m); failed = true; cv.notify_all(); }
    bool has_failed() { std::unique_lock<std::mutex> l(m); return failed; }
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
    const uint64_t shard_count = opt.shard_count > 1 ? (uint64_t)opt.shard_count : 1, shard_rank = shard_count > 1 ? (uint64_t)opt.shard_rank : 0;
    const bool sharded = shard_count > 1;
    // (two pools of this size — parse + pack, format — beside the mapper's and the writer's threads: three quarters of the CPUs the process may use each;
    //  the pools take turns more than they overlap.  On the bench box's 16-CPU share: 6 / 8 / 12 / 16 / 24 / 64 threads a pool -> 10.6 / 10.9 / 13.2 / 12.2 / 10.9 / 10.2 M reads/s to SAM.
    //  The shards of a run share the host: each takes its part.)
    int threads = opt.host_threads > 0 ? opt.host_threads : (int)std::min<unsigned>(64, std::max<unsigned>(1, mcx_usable_cpus() * 3 / 4 / (unsigned)shard_count));
    std::string err;
    Shards sh;
    const uint64_t batch_reads = std::max<uint64_t>(kReadChunkSize, mcx_ctx_max_reads(c) / kReadChunkSize * kReadChunkSize);
    sh.x = opt.exchange; sh.slot_stride = (uint32_t)batch_reads; sh.cap_chunks = (uint32_t)(batch_reads / kReadChunkSize + 2);
    // (from here on a sharded run's shards leave together: whatever fails on one is told to the others)
    int rc = 0;

    // ---- the input: mapped and indexed (plain FASTQ), or a sequential reader per file ---------------------------
    auto plain_fastq = [](const char *path) {
        const std::string p(path);
        if (p.size() > 3 && p.compare(p.size() - 3, 3, ".gz") == 0) return false;
        if (getenv("MCX_SERIAL_PARSER")) return false; // (tests: the sequential reader on plain files)
        struct stat st;
        if (stat(path, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0) return false;
        FILE *f = fopen(path, "rb");
        if (!f) return false;
        const int ch = fgetc(f);
        fclose(f);
        return ch == '@'; // CheckReadFormat, GetData.cpp:22-31
    };
    const bool mapped_input = plain_fastq(fq1) && (!two || plain_fastq(fq2));
    const Tick t_begin = now();
    double w_open = 0, w_first_parsed = 0, w_reader = 0, w_mapped = 0, w_first_mapped = 0; // wall seconds since t_begin (MCX_TIMING)
    Pool pool(threads);        // parse + pack
    Pool fpool(threads);       // format + write (threads of its own: the two stages overlap)
    MappedFastq mf[2];
    Parser ps[2];
    bool fastq = true;
    if (mapped_input) {
        for (int f = 0; f < (two ? 2 : 1) && rc == 0; f++) if (!mf[f].open(f ? fq2 : fq1, err)) rc = mcx_set_error(MCX_ERR_IO, err);
        if (sharded && (rc = sh.agree(rc))) return rc;
        if (rc) return rc;
        for (int f = 0; f < (two ? 2 : 1); f++) {
            // the line counts: every shard counts a share of the blocks, then they tell one another
            const size_t nb = mf[f].n_blocks();
            const size_t b0 = nb * shard_rank / shard_count, b1 = nb * (shard_rank + 1) / shard_count;
            mf[f].count(b0, b1, pool);
            if (sharded) {
                const size_t most = (nb + shard_count - 1) / shard_count + 1;
                std::vector<uint32_t> mine(most, 0);
                memcpy(mine.data(), mf[f].counts() + b0, (b1 - b0) * sizeof(uint32_t));
                if (int e = sh.gather(mine.data(), most * sizeof(uint32_t))) return e;
                for (uint64_t r = 0; r < shard_count; r++) {
                    const size_t r0 = nb * r / shard_count, r1 = nb * (r + 1) / shard_count;
                    memcpy(mf[f].counts() + r0, sh.recv.data() + (size_t)r * most * sizeof(uint32_t), (r1 - r0) * sizeof(uint32_t));
                }
            }
            mf[f].finish();
        }
    } else {
        if (!ps[0].open(fq1, err)) rc = mcx_set_error(MCX_ERR_IO, err);
        if (rc == 0 && two && !ps[1].open(fq2, err)) rc = mcx_set_error(MCX_ERR_IO, err);
        if (rc == 0 && two && ps[0].fastq() != ps[1].fastq()) rc = mcx_set_error(MCX_ERR_IO, std::string(fq1) + " and " + fq2 + " are with different format");
        if (sharded && (rc = sh.agree(rc))) return rc;
        if (rc) return rc;
        fastq = ps[0].fastq();
    }

    // ---- the output: one file, every batch's text at its final place ------------------------------------------------
    int sam_fd = -1;
    bool sam_stream = false; // stdout: written front to back
    uint64_t sam_base = 0;   // where the first batch's text goes
    if (sam_path && sam_path[0]) {
        if (strcmp(sam_path, "-") == 0) {
            if (sharded) rc = mcx_set_error(MCX_ERR_ARG, "a sharded run cannot write its SAM to stdout");
            sam_fd = 1; sam_stream = true;
        } else {
            // shard 0 creates (or empties) the file; the others open it once that has happened
            if (shard_rank == 0) {
                sam_fd = ::open(sam_path, O_RDWR | O_CREAT | (opt.append_sam ? 0 : O_TRUNC), 0644);
                if (sam_fd < 0) rc = mcx_set_error(MCX_ERR_IO, std::string("cannot write ") + sam_path);
            }
            if (sharded && (rc = sh.agree(rc))) { if (sam_fd >= 0) close(sam_fd); return rc; }
            if (shard_rank != 0) {
                sam_fd = ::open(sam_path, O_RDWR, 0644);
                if (sam_fd < 0) rc = mcx_set_error(MCX_ERR_IO, std::string("cannot write ") + sam_path);
            }
            struct stat st;
            if (rc == 0 && opt.append_sam && fstat(sam_fd, &st) == 0) sam_base = (uint64_t)st.st_size;
        }
        if (rc == 0 && !opt.append_sam) {
            std::string hdr;
            sam_header(hix, hdr);
            if (shard_rank == 0) {
                const ssize_t w = sam_stream ? write(sam_fd, hdr.data(), hdr.size()) : pwrite(sam_fd, hdr.data(), hdr.size(), 0);
                if (w != (ssize_t)hdr.size()) rc = mcx_set_error(MCX_ERR_IO, std::string("cannot write ") + sam_path);
            }
            sam_base = hdr.size();
        }
        if (sharded) rc = sh.agree(rc);
        if (rc) { if (sam_fd >= 0 && !sam_stream) close(sam_fd); return rc; }
    }

    int64_t local_avg[4];
    mcx_avg_init(local_avg);
    int64_t *avg = opt.avg_state ? opt.avg_state : local_avg;
    w_open = secs(t_begin, now());

    // busy seconds per stage (MCX_TIMING=1 prints them)
    double t_parse = 0, t_map = 0, t_format = 0, t_write = 0, t_p_lines = 0, t_p_pack = 0, t_p_wait = 0, t_p_push = 0, t_m_take = 0, t_m_collect = 0, t_f_push = 0, t_m_in = 0, t_m_dev = 0, t_m_out = 0, t_m_submit = 0;
    std::vector<double> each_dev; // (MCX_TIMING) mcx_map_batch_dev, batch by batch
    typedef std::unique_ptr<Batch> BatchPtr;
    // batch objects circulate: their buffers (page-locked: slow to get) are allocated once and stay with the context from call to call
    // (one shard holds at most eleven at a time: two per queue between the stages, three on the device, one each in the reader's, the
    //  formatter's and the writer's hands; shards wait two rounds for the places of their text: sixteen.  At -batch 2 M reads an object
    //  pins ~0.5 GB of host memory — 6 GB a shard —, which is the host-memory bill of a run: see INTEGRATION.md)
    const int n_objects = shard_count > 1 ? 16 : 12;
    struct Kept { std::vector<BatchPtr> objects; };
    void **slot = mcx_ctx_files_slot(c, [](void *p) { delete (Kept *)p; });
    if (!*slot) *slot = new Kept();
    Kept *kept = (Kept *)*slot;
    Queue<BatchPtr> parsed(2), mapped(2), formatted(2), spare((size_t)n_objects);
    for (int k = 0; k < n_objects; k++) {
        if (!kept->objects.empty()) { spare.push(std::move(kept->objects.back())); kept->objects.pop_back(); }
        else spare.push(BatchPtr(new Batch));
    }
    std::atomic<bool> abort(false);

    // ---- stage 1: parse + pack ---------------------------------------------------------------------------------------
    std::thread reader([&] {
        bool done = false;
        uint64_t number = 0;
        const uint32_t per_file = (uint32_t)(two ? batch_reads / 2 : batch_reads);
        uint64_t total_recs[2] = {0, 0};
        if (mapped_input) for (int f = 0; f < (two ? 2 : 1); f++) total_recs[f] = (mf[f].lines() + 2) / 4; // a record needs its header and sequence lines
        while (!done) {
            const bool mine = number % shard_count == shard_rank;
            if (mapped_input && !mine && (number + 1) * (uint64_t)per_file < total_recs[0]) { number++; continue; } // another shard's batch: not a byte of it is touched
            const Tick tw = now();
            BatchPtr b = spare.pop();
            const Tick t0 = now();
            t_p_wait += secs(tw, t0);
            b->two_files = two; b->fastq = fastq; b->n = 0; b->last = false; b->error.clear(); b->number = number;
            b->in[0].clear(); b->in[1].clear(); b->n_odd[0] = b->n_odd[1] = 0; b->n_pair_reads = 0;
            if (mapped_input) {
                // the records of both files in one pass of the pool, every share written where it belongs
                const uint64_t r0 = number * per_file;
                const int nf = two ? 2 : 1;
                uint64_t cnt[2] = {0, 0}, r1[2] = {0, 0};
                int parts[2] = {0, 0};
                for (int f = 0; f < nf; f++) {
                    View &v = b->in[f];
                    v.base = mf[f].data();
                    r1[f] = std::min<uint64_t>(r0 + per_file, total_recs[f]);
                    cnt[f] = r1[f] > r0 ? r1[f] - r0 : 0;
                    if (!mine) { v.last = true; continue; } // (the input ends inside another shard's batch: an empty batch carries the news)
                    parts[f] = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)pool.size(), cnt[f] / 2048));
                    v.recs.resize((size_t)cnt[f]);
                    if (v.recs.size() != cnt[f]) { v.error = "out of memory"; parts[f] = 0; }
                }
                const int all = parts[0] + parts[1];
                std::vector<size_t> got((size_t)std::max(all, 1), 0);
                std::vector<std::string> perr((size_t)std::max(all, 1));
                std::vector<uint8_t> ok((size_t)std::max(all, 1), 1);
                if (all) pool.run(all, [&](int t) {
                    const int f = t < parts[0] ? 0 : 1, k = f ? t - parts[0] : t;
                    const uint64_t a = cnt[f] * (uint64_t)k / (uint64_t)parts[f], z = cnt[f] * (uint64_t)(k + 1) / (uint64_t)parts[f];
                    ok[(size_t)t] = mf[f].parse(r0 + a, r0 + z, max_len, b->in[f].recs.data() + a, got[(size_t)t], perr[(size_t)t]) ? 1 : 0;
                });
                for (int f = 0; f < nf; f++) {
                    if (!parts[f]) continue;
                    View &v = b->in[f];
                    bool stopped = false;
                    for (int k = 0; k < parts[f] && !stopped; k++) {
                        const size_t t = (size_t)(f ? parts[0] + k : k);
                        if (!ok[t]) { // the records end inside this share: those before the stop count, nothing behind them
                            stopped = true;
                            if (!perr[t].empty()) v.error = perr[t];
                            v.recs.resize((size_t)(cnt[f] * (uint64_t)k / (uint64_t)parts[f]) + got[t]);
                        }
                    }
                    v.last = stopped || cnt[f] < per_file || r1[f] >= total_recs[f];
                }
            } else {
                if (two) {
                    std::thread t2([&] { ps[1].take(b->in[1], per_file, max_len); });
                    ps[0].take(b->in[0], per_file, max_len);
                    t2.join();
                } else ps[0].take(b->in[0], per_file, max_len);
            }
            t_p_lines += secs(t0, now());
            if (two) {
                // the reference stops at the first empty read of file 1 and takes whatever file 2 holds (GetData.cpp:91-93)
                if (mine || !mapped_input) {
                    if (b->in[1].n() < b->in[0].n()) b->error = std::string(fq2) + " holds fewer reads than " + fq1;
                    if (b->in[1].n() > b->in[0].n()) b->in[1].recs.resize(b->in[0].n());
                    b->n = 2 * b->in[0].n();
                }
            } else b->n = b->in[0].n();
            done = b->in[0].last;
            for (int f = 0; f < 2; f++) if (!b->in[f].error.empty()) b->error = b->in[f].error;
            if (!b->error.empty() || abort.load()) done = true;
            // batches are dealt to the shards in turn; another shard's batch is dropped — unless it carries the end of the input
            // or an error, which every shard must see (the sequential reader has to walk the stream to get past it)
            number++;
            if (!mine && !done) { spare.push(std::move(b)); continue; }
            if (!mine && b->error.empty()) b->n = 0;
            // 2-bit rows of this shard's reads, the bytes that are not ACGT beside them
            const Tick tp = now();
            if (b->n && b->error.empty()) {
                const uint32_t n = b->n;
                uint32_t npr = paired ? n : 0; // reads mapped as pairs; the odd tail of an interleaved file is mapped read by read
                if (paired && (n & 1)) npr = n / kReadChunkSize * kReadChunkSize;
                b->n_pair_reads = npr;
                uint32_t longest = 0;
                {
                    const int slices = (int)std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)pool.size(), n / 4096));
                    std::vector<uint32_t> most((size_t)slices, 0);
                    pool.run(slices, [&](int k) {
                        uint32_t m = 0;
                        for (uint32_t r = (uint32_t)((uint64_t)n * k / slices); r < (uint32_t)((uint64_t)n * (k + 1) / slices); r++) { const char *base; m = std::max(m, b->rec(r, base).rlen); }
                        most[(size_t)k] = m;
                    });
                    for (uint32_t m : most) longest = std::max(longest, m);
                }
                b->row_words = (longest + 15) / 16;
                if (!b->reserve(std::max<size_t>(n, batch_reads), std::max<size_t>(b->row_words, ((size_t)max_len + 15) / 16))) b->error = "cannot allocate pinned host memory";
                else {
                    std::vector<uint64_t> all_odd[2];
                    for (int part = 0; part < 2; part++) { // (a part's reads are numbered from 0: it is a batch of its own on the device)
                        const uint32_t first = part ? npr : 0, cnt = part ? n - npr : npr;
                        if (!cnt) continue;
                        const int slices = (int)std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)pool.size(), cnt / 4096));
                        std::vector<std::vector<uint64_t>> odd((size_t)slices);
                        pool.run(slices, [&](int k) {
                            const uint32_t lo = first + (uint32_t)((uint64_t)cnt * k / slices), hi = first + (uint32_t)((uint64_t)cnt * (k + 1) / slices);
                            for (uint32_t r = lo; r < hi; r++) {
                                const char *base;
                                const Rec &e = b->rec(r, base);
                                b->lens[r] = e.rlen;
                                pack_row((const uint8_t *)base + e.seq, e.rlen, r - first, b->rows + (size_t)r * b->row_words, b->row_words, odd[(size_t)k]);
                            }
                        });
                        for (auto &o : odd) all_odd[part].insert(all_odd[part].end(), o.begin(), o.end());
                    }
                    const size_t total = all_odd[0].size() + all_odd[1].size();
                    if (total && !b->reserve_odd(total)) b->error = "cannot allocate pinned host memory";
                    else if (total) {
                        memcpy(b->odd, all_odd[0].data(), all_odd[0].size() * 8);
                        memcpy(b->odd + all_odd[0].size(), all_odd[1].data(), all_odd[1].size() * 8);
                    }
                    b->n_odd[0] = (uint32_t)all_odd[0].size(); b->n_odd[1] = (uint32_t)all_odd[1].size();
                }
                b->is_mate2.assign(n, 0);
                for (uint32_t r = 1; r < npr; r += 2) b->is_mate2[r] = 1;
            }
            if (!b->error.empty()) done = true;
            b->last = done;
            t_p_pack += secs(tp, now());
            t_parse += secs(t0, now());
            if (w_first_parsed == 0) w_first_parsed = secs(t_begin, now());
            const Tick tq = now();
            parsed.push(std::move(b));
            t_p_push += secs(tq, now());
        }
        w_reader = secs(t_begin, now());
    });

    // ---- stage 3: format + write -------------------------------------------------------------------------------------
    std::atomic<int> write_rc(0);
    Places places;
    std::thread writer([&] {
        std::deque<BatchPtr> waiting;   // (sharded) formatted, their place in the file not known yet
        uint64_t next_place = sam_base; // one shard: the batches follow one another
        bool stop = false;
        auto write_out = [&](BatchPtr &b, uint64_t at) {
            const Tick t1 = now();
            if (sam_stream) { for (Text &t : b->slices) if (t.size() && write(sam_fd, t.b.data(), t.size()) != (ssize_t)t.size()) write_rc = MCX_ERR_IO; }
            else if (b->sam_bytes) {
                std::vector<uint64_t> off(b->slices.size() + 1, at);
                for (size_t k = 0; k < b->slices.size(); k++) off[k + 1] = off[k] + b->slices[k].size();
                // Positioned writes, every slice at its final place, from this one thread: writers of one growing file queue up
                // behind its lock and get in each other's way (tools/ubench_filewrite.cpp on the bench box's tmpfs, a gigabyte of source text:
                // one thread 5.7 GB/s, two to sixteen 3.4-5.1; into pages that exist already 8.8 GB/s — but laying them out ahead of the
                // writer with fallocate(KEEP_SIZE) from a thread of its own, 128 MB at a time, made runs slower as often as faster, 1.17 / 1.59 s
                // against 1.31 / 1.28 for 6 GB of text: the two take the file's lock in turns; memcpy into a mapping of a sparse file 3.5-4.5 GB/s).
                // MCX_SAM_MMAP=1 keeps the mapping path for file systems where it pays; the file then grows under a lock of its
                // own and never shrinks: the other shards write further on.
                bool done_w = false;
                if (getenv("MCX_SAM_MMAP")) {
                    const uint64_t end = at + b->sam_bytes, page = (uint64_t)sysconf(_SC_PAGESIZE), a0 = at & ~(page - 1);
                    struct stat st;
                    bool ok = flock(sam_fd, LOCK_EX) == 0;
                    if (ok) { ok = fstat(sam_fd, &st) == 0 && ((uint64_t)st.st_size >= end || ftruncate(sam_fd, (off_t)end) == 0); (void)flock(sam_fd, LOCK_UN); }
                    char *m = ok ? (char *)mmap(nullptr, (size_t)(end - a0), PROT_READ | PROT_WRITE, MAP_SHARED, sam_fd, (off_t)a0) : (char *)MAP_FAILED;
                    if (m != (char *)MAP_FAILED) {
                        for (size_t k = 0; k < b->slices.size(); k++) { const Text &t = b->slices[k]; if (t.size()) memcpy(m + (off[k] - a0), t.b.data(), t.size()); } // (the pool belongs to the formatter)
                        (void)munmap(m, (size_t)(end - a0));
                        done_w = true;
                    }
                }
                if (!done_w) { // positioned writes, by this thread alone (see above)
                    for (size_t k = 0; k < b->slices.size() && write_rc == 0; k++) {
                        const Text &t = b->slices[k];
                        size_t done_b = 0;
                        while (done_b < t.size()) {
                            const ssize_t w = pwrite(sam_fd, t.b.data() + done_b, t.size() - done_b, (off_t)(off[k] + done_b));
                            if (w <= 0) { write_rc = MCX_ERR_IO; break; }
                            done_b += (size_t)w;
                        }
                    }
                }
            }
            t_write += secs(t1, now());
        };
        auto flush_waiting = [&](bool block) { // batches whose place has arrived go out, oldest first
            while (!waiting.empty()) {
                uint64_t at = 0;
                if (!places.wait_place(waiting.front()->number, at, block)) {
                    if (!places.has_failed()) return; // not yet
                    waiting.front()->sam_bytes = 0;   // the run has failed: nothing more is written
                }
                write_out(waiting.front(), at);
                spare.push(std::move(waiting.front()));
                waiting.pop_front();
            }
        };
        while (!stop) {
            BatchPtr b;
            if (waiting.empty()) b = formatted.pop();
            else if (!formatted.try_pop(b)) { flush_waiting(false); std::this_thread::sleep_for(std::chrono::microseconds(100)); continue; }
            if (b->last) stop = true;
            if (!sharded) { // its place is behind the batch before it
                if (sam_fd >= 0) write_out(b, next_place);
                next_place += b->sam_bytes;
                spare.push(std::move(b));
            } else {
                // (a batch that carries nothing of this shard's — the news of the input's end — has no round of its own to be placed in)
                if (b->number % shard_count == shard_rank) waiting.push_back(std::move(b));
                else spare.push(std::move(b));
                flush_waiting(false);
            }
        }
        flush_waiting(true); // the places of the last rounds' text arrive with the closing exchanges
    });
    // (the text of batch i + 1 is made while batch i's is written)
    std::thread formatter([&] {
        bool stop = false;
        while (!stop) {
            BatchPtr b = mapped.pop();
            if (b->last) stop = true;
            b->sam_bytes = 0;
            if (sam_fd >= 0 && b->n && write_rc == 0) {
                const Tick t0 = now();
                const int parts = (int)std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)fpool.size(), b->n / 2048));
                b->slices.resize((size_t)parts);
                fpool.run(parts, [&](int k) {
                    const uint32_t lo = (uint32_t)((uint64_t)b->n * k / parts), hi = (uint32_t)((uint64_t)b->n * (k + 1) / parts);
                    Text &t = b->slices[(size_t)k];
                    size_t bound = 0;
                    for (uint32_t r = lo; r < hi; r++) {
                        const char *base;
                        const Rec &e = b->rec(r, base);
                        bound += sam_bound(hix, e.name_len, e.rlen, b->recs[r].chr == 0xFFFFu ? -1 : (int)b->recs[r].chr, b->recs[r].n_cigar);
                    }
                    t.start(bound);
                    for (uint32_t r = lo; r < hi; r++) sam_record(hix, *b, r, t);
                });
                for (const Text &t : b->slices) b->sam_bytes += t.size();
                t_format += secs(t0, now());
            } else for (Text &t : b->slices) t.w = nullptr; // (no text of this batch; the buffers stay with the object)
            if (sharded && b->number % shard_count == shard_rank) places.put_size(b->number, b->sam_bytes); // (the other shards wait for the sizes of a round)
            const Tick tq = now();
            formatted.push(std::move(b));
            t_f_push += secs(tq, now());
        }
    });

    // ---- stage 2 (this thread): copy in | map | copy out, three parts of batches on the device at a time ------------------
    const bool profile = mcx_ctx_has_profile(c);
    int in_flight = 0;             // parts submitted and not collected yet
    std::deque<BatchPtr> leaving;  // mapped, their last part on its way out (oldest first), with the number of parts each still waits for
    std::deque<int> leaving_parts;
    bool input_done = false, dead = false, ended = false; // dead: a shard failed and every shard knows; ended: the input ended in an earlier batch
    uint64_t rounds_done = 0;
    uint64_t place_next = sam_base;
    std::deque<std::pair<uint64_t, bool>> rounds_unplaced; // (sharded) rounds mapped whose text sizes have not been exchanged; did this shard hold a batch of the round?
    auto place_round = [&](uint64_t round, bool own) -> int { // the shards' batches of a round find their places in the file
        uint64_t mine = 0;
        if (own && !places.wait_size(round * shard_count + shard_rank, mine)) mine = 0;
        if (int e = sh.gather(&mine, sizeof mine)) return e;
        uint64_t at = place_next;
        for (uint64_t r = 0; r < shard_count; r++) {
            uint64_t v; memcpy(&v, sh.recv.data() + (size_t)r * sizeof v, sizeof v);
            if (r == shard_rank && own) places.put_place(round * shard_count + shard_rank, at);
            at += v;
        }
        place_next = at;
        return 0;
    };
    auto collect_oldest = [&]() -> int { // the oldest mapped part has arrived in host memory
        const Tick tq = now();
        const int e = mcx_stream_collect(c, nullptr, nullptr);
        t_m_collect += secs(tq, now());
        in_flight--;
        if (!leaving.empty() && --leaving_parts.front() == 0) { mapped.push(std::move(leaving.front())); leaving.pop_front(); leaving_parts.pop_front(); }
        return e;
    };
    auto submit = [&](Batch *p) -> int {
        const uint32_t n = p->n, npr = p->n_pair_reads;
        int e = 0;
        if (npr) { e = mcx_stream_submit_packed(c, p->rows, p->row_words, p->lens, npr, p->odd, p->n_odd[0]); if (e) return e; in_flight++; }
        if (npr < n) {
            e = mcx_stream_submit_packed(c, p->rows + (size_t)npr * p->row_words, p->row_words, p->lens + npr, n - npr, p->odd ? p->odd + p->n_odd[0] : nullptr, p->n_odd[1]);
            if (e) return e;
            in_flight++;
        }
        return 0;
    };
    // the batch being mapped and the one behind it (already on its way in)
    BatchPtr cur, nxt;
    bool cur_in = false, nxt_in = false;
    auto take_next = [&](bool block) { // a parsed batch, its copy to the device started when there is room
        if (nxt || input_done) return;
        BatchPtr b;
        const Tick tq = now();
        if (block) b = parsed.pop(); else if (!parsed.try_pop(b)) return;
        t_m_take += secs(tq, now());
        if (b->last) input_done = true;
        if (rc == 0 && !b->error.empty()) rc = mcx_set_error(b->error.find("max_read_len") != std::string::npos ? MCX_ERR_UNSUPPORTED : MCX_ERR_IO, b->error);
        if (rc || ended) b->n = 0;
        nxt = std::move(b); nxt_in = false;
    };
    auto try_submit_next = [&]() {
        if (!nxt || nxt_in || rc) return;
        if (nxt->n == 0) { nxt_in = true; return; }
        if (in_flight + nxt->n_parts() > 3) return;
        const Tick tq = now();
        const int e = submit(nxt.get());
        t_m_submit += secs(tq, now());
        if (e) rc = e; else nxt_in = true;
    };
    for (;;) {
        if (!cur) {
            if (!nxt) { if (input_done) break; take_next(true); }
            while (nxt && !nxt_in && rc == 0) { try_submit_next(); if (!nxt_in && rc == 0) { const int e = collect_oldest(); if (e) rc = e; } }
            cur = std::move(nxt); cur_in = nxt_in; nxt_in = false;
            if (rc) { cur->n = 0; abort.store(true); }
        }
        // the batch behind it: parsed already?  then its copy in runs under this one's kernels
        take_next(false);
        try_submit_next();
        Batch *p = cur.get();
        const Tick t1 = now();
        // what of the batch is on the device (copied in, or on its way); after a failure it leaves the device unmapped
        const uint32_t n_pr = cur_in ? p->n_pair_reads : 0u, n_sg = cur_in ? p->n - p->n_pair_reads : 0u;
        int parts_out = 0;
        const uint8_t *d_bases = nullptr; const uint32_t *d_off = nullptr; mcx_aln *d_aln = nullptr; uint32_t *d_cig = nullptr; uint32_t n_dev = 0;
        auto part_in = [&]() { const Tick tq = now(); const int e = mcx_stream_next(c, &d_bases, &d_off, &n_dev, &d_aln, &d_cig); t_m_in += secs(tq, now()); if (e && rc == 0) rc = e; return e == 0; };
        auto part_out = [&](bool second) {
            const Tick tq = now();
            const int e = second ? mcx_stream_mapped32(c, p->recs + p->n_pair_reads, p->cig + MCX_CIGAR_POOL_WORDS(p->n_pair_reads)) : mcx_stream_mapped32(c, p->recs, p->cig);
            t_m_out += secs(tq, now());
            if (e && rc == 0) rc = e;
            if (e == 0) parts_out++;
        };
        if (!sharded) {
            if (n_pr && part_in()) { const Tick tq = now(); if (rc == 0) rc = mcx_map_batch_dev(c, d_bases, d_off, n_pr, 1, avg, d_aln, d_cig, stats); t_m_dev += secs(tq, now()); each_dev.push_back(secs(tq, now())); part_out(false); }
            if (n_sg && part_in()) { if (rc == 0) rc = mcx_map_batch_dev(c, d_bases, d_off, n_sg, 0, avg, d_aln, d_cig, stats); part_out(true); }
        } else if (!dead && !ended && p->number / shard_count >= rounds_done) {
            rounds_done = p->number / shard_count + 1;
            Shards::Head h = {rc, rc ? 0u : n_pr, rc ? 0u : n_sg, p->last ? 1u : 0u};
            int e = sh.gather(&h, sizeof h);
            bool any_pair = false, any_single = false;
            uint64_t before = 0, round_total = 0;
            uint32_t my_pr = rc ? 0u : n_pr, my_sg = rc ? 0u : n_sg;
            if (e) rc = e;
            else {
                // the input ends with the first batch of the round that says so: the batches behind it do not exist
                int cut = sh.x->size;
                for (int r = 0; r < sh.x->size; r++) { Shards::Head o; memcpy(&o, sh.recv.data() + (size_t)r * sizeof o, sizeof o); if (o.last && r < cut) cut = r; }
                for (int r = 0; r < sh.x->size; r++) {
                    Shards::Head o; memcpy(&o, sh.recv.data() + (size_t)r * sizeof o, sizeof o);
                    if (o.rc && rc == 0) rc = mcx_set_error(o.rc, "shard " + std::to_string(r) + " failed");
                    if (r > cut) { o.n_pair = o.n_single = 0; if (r == sh.x->rank) my_pr = my_sg = 0; }
                    any_pair |= o.n_pair != 0; any_single |= o.n_single != 0;
                    if (r < sh.x->rank) before += (uint64_t)o.n_pair + o.n_single;
                    round_total += (uint64_t)o.n_pair + o.n_single;
                }
                if (cut < sh.x->size) { ended = true; abort.store(true); }
            }
            if (rc) my_pr = my_sg = 0;
            const int64_t round_base = avg[3];
            // (a part that was copied in but does not count any more — the input ended in a batch before this one — still leaves the device)
            const bool in1 = n_pr && part_in();
            if (rc == 0 && any_pair) rc = sh.pairs(c, d_bases, d_off, in1 ? my_pr : 0u, round_base + (int64_t)before, avg, profile, d_aln, d_cig, stats);
            if (in1) part_out(false);
            const bool in2 = n_sg && part_in();
            if (rc == 0 && any_single) rc = sh.singles(c, d_bases, d_off, in2 ? my_sg : 0u, round_base + (int64_t)before + my_pr, profile, d_aln, d_cig, stats);
            if (in2) part_out(true);
            if (my_pr == 0 && my_sg == 0) p->n = 0; // (nothing of this batch counts)
            avg[3] = round_base + (int64_t)round_total;
            if (rc) dead = true;
            // the text of an earlier round finds its place (two rounds later every formatter has long been through it)
            rounds_unplaced.push_back(std::make_pair(p->number / shard_count, p->number % shard_count == shard_rank));
            while (rounds_unplaced.size() > 2 && !dead) { const int e3 = place_round(rounds_unplaced.front().first, rounds_unplaced.front().second); rounds_unplaced.pop_front(); if (e3) { rc = e3; dead = true; } }
        } else {
            // (sharded, after the end of the input or a failure: what is on the device leaves it unmapped)
            if (n_pr && part_in()) part_out(false);
            if (n_sg && part_in()) part_out(true);
            p->n = 0;
        }
        if (p->n) t_map += secs(t1, now());
        if (w_first_mapped == 0) w_first_mapped = secs(t_begin, now());
        if (rc) { p->n = 0; abort.store(true); places.fail(); }
        if (parts_out == 0) { // nothing on its way out: the batch goes on as it is, behind the ones that are
            while (!leaving.empty()) { const int e = collect_oldest(); if (e && rc == 0) rc = e; }
            mapped.push(std::move(cur));
        } else { leaving.push_back(std::move(cur)); leaving_parts.push_back(parts_out); }
        cur.reset(); cur_in = false;
        // at most one batch's parts on their way out behind the one mapped next
        while (leaving.size() > 1) { const int e = collect_oldest(); if (e && rc == 0) rc = e; }
    }
    while (!leaving.empty()) { const int e = collect_oldest(); if (e && rc == 0) rc = e; }
    // (sharded) the places of the last rounds' text
    while (sharded && !rounds_unplaced.empty()) {
        if (!dead) { const int e = place_round(rounds_unplaced.front().first, rounds_unplaced.front().second); if (e) { if (rc == 0) rc = e; dead = true; places.fail(); } }
        rounds_unplaced.pop_front();
    }
    if (rc) places.fail();
    w_mapped = secs(t_begin, now());
    formatter.join();
    writer.join();
    reader.join();
    { BatchPtr b; while (spare.try_pop(b)) kept->objects.push_back(std::move(b)); while (parsed.try_pop(b)) kept->objects.push_back(std::move(b)); while (mapped.try_pop(b)) kept->objects.push_back(std::move(b)); }
    if (getenv("MCX_TIMING")) {
        std::string e;
        for (size_t k = 0; k < each_dev.size() && k < 24; k++) e += " " + std::to_string((int)(each_dev[k] * 1e4) / 10.0).substr(0, 5);
        fprintf(stderr, "[mcx_map_files] mcx_map_batch_dev, ms per batch:%s\n", e.c_str());
    }
    if (getenv("MCX_TIMING"))
        fprintf(stderr, "[mcx_map_files] busy seconds: parse + pack %.3f (lines %.3f, rows %.3f; waited for a free batch %.3f) | map %.3f | format %.3f write %.3f  (%d + %d host threads, %s input)\n",
                t_parse, t_p_lines, t_p_pack, t_p_wait, t_map, t_format, t_write, threads, threads, mapped_input ? "mapped" : "sequential"),
        fprintf(stderr, "[mcx_map_files] waits: reader for room behind it %.3f | mapper for a parsed batch %.3f, for copies out %.3f | formatter for the writer %.3f || mapper's calls: submit %.3f, next %.3f, map_batch_dev %.3f, mapped %.3f\n", t_p_push, t_m_take, t_m_collect, t_f_push, t_m_submit, t_m_in, t_m_dev, t_m_out),
        fprintf(stderr, "[mcx_map_files] wall seconds: input opened and indexed %.3f | first batch parsed %.3f, mapped %.3f | last batch parsed %.3f, mapped %.3f | all written %.3f\n",
                w_open, w_first_parsed, w_first_mapped, w_reader, w_mapped, secs(t_begin, now()));
    if (sam_fd >= 0 && !sam_stream) { if (close(sam_fd) != 0 && write_rc.load() == 0) write_rc.store(MCX_ERR_IO); }
    if (rc == 0 && write_rc.load()) rc = mcx_set_error(MCX_ERR_IO, std::string("cannot write ") + (sam_path ? sam_path : ""));
    return rc;
}

extern "C" int mcx_map_files(mcx_ctx *c, const char *fq1, const char *fq2, const char *sam_path, mcx_stats *stats)
{
    return mcx_map_files_ex(c, fq1, fq2, nullptr, sam_path, stats);
}

// The file front end's parallel inflater on a file by itself (tests, scripts/gz_rate.py): the text of `path` into out[0 .. cap) as far as it fits; returns the
// text's whole length, -1 when the file cannot be mapped or is no gzip file, -2 when the stream is damaged (*n_out: what was delivered before that).
extern "C" int64_t mcx_gz_inflate(const char *path, int threads, uint64_t stretch_bytes, uint8_t *out, uint64_t cap, uint64_t *n_out)
{
    if (n_out) *n_out = 0;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 18) { close(fd); return -1; }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return -1;
    int64_t total = 0;
    {
        Pool pool(std::max(1, threads));
        mcx::pgz::Reader rd;
        mcx::pgz::Text text;
        if (!rd.open((const uint8_t *)m, (size_t)st.st_size, pool.size(), stretch_bytes ? (size_t)stretch_bytes : (size_t)2 << 20, [&](int n, const std::function<void(int)> &f) { pool.run(n, f); })) total = -1;
        while (total >= 0 && rd.next(text)) {
            if (out && (uint64_t)total < cap) memcpy(out + total, text.data(), (size_t)std::min<uint64_t>(text.size(), cap - (uint64_t)total));
            total += (int64_t)text.size();
        }
        if (n_out) *n_out = (uint64_t)std::max<int64_t>(total, 0);
        if (rd.failed()) total = total < 0 ? -1 : -2;
    }
    munmap(m, (size_t)st.st_size);
    return total;
}
