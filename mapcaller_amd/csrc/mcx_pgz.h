// mapcaller_amd/csrc/mcx_pgz.h — a gzip stream inflated by several threads (host code; the file front end's reader for plain .gz input).
//
// The reference reads .gz FASTQ through zlib's gzgets (src/GetData.cpp:101-146, gzGetNextChunk): one thread, about 0.5 GB/s of text per file — fifteen
// times under what the plain-file front end parses and a hundred and fifty under what the device maps.  A deflate stream has no entry points: a block
// starts at an arbitrary BIT, and what it copies from reaches 32 KB back into text that has not been inflated yet.  Both are dealt with the way pugz does
// (Kerbiriou & Chikhi, "Parallel decompression of gzip-compressed files and random access to DNA sequences", 2019) — restated here from the description of
// the method and RFC 1951 / 1952; no code of theirs, none of zlib's inflate:
//
//   1. the compressed bytes of a round are cut into as many stretches as there are threads.  Every thread but the first looks for a block start in its
//      stretch: bit by bit, a candidate must be a dynamic-Huffman block (BTYPE 2) whose header describes a COMPLETE code-length code, COMPLETE literal /
//      length and distance codes with an end-of-block symbol, and whose first symbols decode to text (printable ASCII, tab, CR, LF).  A false start
//      that survives this decodes rubbish and is found out in step 3;
//   2. every thread inflates from its start to a later thread's start with a window it does not know: its output is 16-bit symbols — a value below 256 is
//      a byte, 256 + p stands for "the byte p of the 32 KB before my start" — and copies move such symbols like any others;
//   3. a stretch must END exactly where a later one begins (a block boundary of the true stream is a true start; a false one is never met, its
//      output is dropped and the stretch before it simply runs on to the next true one); then the placeholders are resolved stretch by stretch: the last
//      32 KB of each in sequence (a short serial chain), everything else in parallel, with the member's CRC-32 taken on the way (crc32_combine).
//
// The first stretch of a round starts where the round before ended — an exact block boundary, window known — so nothing is guessed about the stream's
// beginning, and a file in which no start is ever found (not text, stored blocks only, blocks longer than a stretch) is inflated by one thread, correctly.
// Concatenated members (RFC 1952 2.2) are followed; a damaged stream ends the input where zlib's reader would have given up (no more text is delivered).
#ifndef MCX_PGZ_H
#define MCX_PGZ_H
#include <zlib.h>
#include <immintrin.h>
#include <wmmintrin.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <new>
#include <vector>

namespace mcx {
namespace pgz {

constexpr int kWin = 32768;
constexpr int kPBits = 10, kSubBits = 5; // primary table index bits; a longer code's remaining bits (15 - 10)

struct Bits { // LSB-first bit reader over [base, end); reading past the end yields zeros and sets `over`
    const uint8_t *base, *p, *end;
    uint64_t buf = 0;
    int cnt = 0;
    bool over = false;
    void start(const uint8_t *b, const uint8_t *e, uint64_t bit)
    {
        base = b; end = e; p = b + (bit >> 3); buf = 0; cnt = 0; over = false;
        refill();
        const int skip = (int)(bit & 7);
        buf >>= skip; cnt -= skip;
    }
    inline void refill()
    {
        if (p + 8 <= end) { // eight bytes at once: as many whole bytes as fit are kept, the rest is fetched again next time
            uint64_t w; memcpy(&w, p, 8);
            buf |= w << cnt;
            p += (63 - cnt) >> 3;
            cnt |= 56;
            return;
        }
        while (cnt <= 56) {
            if (p < end) buf |= (uint64_t)*p << cnt;
            else if (p >= end + 8) over = true; // (eight bytes of slack: a decoder looks ahead of what it consumes)
            p++; cnt += 8;
        }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    inline void drop(int n) { buf >>= n; cnt -= n; }
    inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
    uint64_t pos() const { return (uint64_t)(p - base) * 8 - (uint64_t)cnt; }
    void align() { const int k = cnt & 7; drop(k); } // to the next byte boundary (cnt is congruent to the bits left of the current byte)
};

struct Ent { uint16_t val; uint8_t bits, sub; }; // sub != 0: val = offset of a subtable of 2^sub entries; else val = symbol, bits = its code length

struct Huff {
    std::vector<Ent> tab;
    // lens[0..n): code lengths (0 = unused).  Returns 0 = complete code, 1 = a single code of length 1 (allowed for distances), 2 = no code at all, -1 = invalid.
    int build(const uint8_t *lens, int n)
    {
        int count[16] = {0};
        for (int i = 0; i < n; i++) count[lens[i]]++;
        if (count[0] == n) return 2;
        long left = 1;
        for (int l = 1; l <= 15; l++) { left = left * 2 - count[l]; if (left < 0) return -1; }
        int kind = 0;
        if (left > 0) { if (n - count[0] == 1 && count[1] == 1) kind = 1; else return -1; }
        uint16_t next[16];
        { uint16_t c = 0; for (int l = 1; l <= 15; l++) { c = (uint16_t)((c + count[l - 1]) << 1); next[l] = c; } next[0] = 0; }
        tab.assign((size_t)1 << kPBits, Ent{0, 0, 0});
        for (int s = 0; s < n; s++) {
            const int l = lens[s];
            if (!l) continue;
            uint32_t code = next[l]++, rev = 0;
            for (int k = 0; k < l; k++) rev |= ((code >> k) & 1u) << (l - 1 - k);
            if (l <= kPBits) {
                for (uint32_t i = rev; i < (1u << kPBits); i += 1u << l) tab[i] = Ent{(uint16_t)s, (uint8_t)l, 0};
            } else {
                const uint32_t low = rev & ((1u << kPBits) - 1);
                if (!tab[low].sub) {
                    if (tab.size() + (1u << kSubBits) > 65535) return -1;
                    tab[low] = Ent{(uint16_t)tab.size(), (uint8_t)kPBits, (uint8_t)kSubBits};
                    tab.resize(tab.size() + (1u << kSubBits), Ent{0, 0, 0});
                }
                const uint32_t hi = rev >> kPBits, step = 1u << (l - kPBits);
                for (uint32_t i = hi; i < (1u << kSubBits); i += step) tab[tab[low].val + i] = Ent{(uint16_t)s, (uint8_t)(l - kPBits), 0};
            }
        }
        return kind;
    }
    // the next symbol (bits consumed); -1: the bits are no code (possible only with an incomplete code)
    inline int decode(Bits &b) const
    {
        Ent e = tab[b.peek(kPBits)];
        if (e.sub) { b.drop(kPBits); e = tab[e.val + b.peek(e.sub)]; }
        if (!e.bits) return -1;
        b.drop(e.bits);
        return e.val;
    }
};

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct Codes { Huff lit, dist; bool dist_none = false; };

// the header of a dynamic block (RFC 1951 3.2.7), the bits behind BFINAL / BTYPE; strict: a search may only accept what a compressor writes
static inline bool read_dynamic(Bits &b, Codes &c)
{
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    b.refill();
    const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; i++) { b.refill(); cl[order[i]] = (uint8_t)b.take(3); }
    Huff clh;
    if (clh.build(cl, 19) != 0) return false;
    uint8_t lens[286 + 30];
    int n = 0;
    while (n < hlit + hdist) {
        b.refill();
        const int s = clh.decode(b);
        if (s < 0) return false;
        if (s < 16) lens[n++] = (uint8_t)s;
        else {
            int rep, v = 0;
            if (s == 16) { if (n == 0) return false; v = lens[n - 1]; rep = 3 + (int)b.take(2); }
            else if (s == 17) rep = 3 + (int)b.take(3);
            else rep = 11 + (int)b.take(7);
            if (n + rep > hlit + hdist) return false;
            while (rep--) lens[n++] = (uint8_t)v;
        }
    }
    if (b.over || lens[256] == 0) return false;
    { const int lk = c.lit.build(lens, hlit); if (lk != 0 && lk != 1) return false; }
    const int dk = c.dist.build(lens + hlit, hdist);
    if (dk < 0) return false;
    c.dist_none = dk == 2;
    return true;
}

static inline void fixed_codes(Codes &c)
{
    uint8_t l[288];
    for (int i = 0; i < 144; i++) l[i] = 8;
    for (int i = 144; i < 256; i++) l[i] = 9;
    for (int i = 256; i < 280; i++) l[i] = 7;
    for (int i = 280; i < 288; i++) l[i] = 8;
    c.lit.build(l, 288);
    uint8_t d[32];
    for (int i = 0; i < 32; i++) d[i] = 5;
    c.dist.build(d, 32); // (32 codes of 5 bits; 30 and 31 never occur in a valid stream and are refused where they are decoded)
    c.dist_none = false;
}

// One stretch's output: kWin placeholder symbols (or the real window) in front, then what was inflated.  (Plain storage that is not cleared when it
// grows: a vector's value-initialisation wrote every symbol's place twice.)
struct Out {
    struct Store {
        uint16_t *p = nullptr; size_t cap = 0;
        Store() {}
        Store(const Store &) = delete;
        Store &operator=(const Store &) = delete;
        Store(Store &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
        ~Store() { free(p); }
        uint16_t *data() const { return p; }
        size_t size() const { return cap; }
        void grow(size_t n) { if (n > cap) { uint16_t *q = (uint16_t *)realloc(p, n * sizeof(uint16_t)); if (!q) throw std::bad_alloc(); p = q; cap = n; } }
    } s;
    size_t n = 0; // symbols behind the window prefix
    Out() {}
    Out(Out &&) noexcept = default;
    void begin(const uint8_t *window, size_t have) // window: the `have` <= kWin bytes before the start (null: unknown)
    {
        s.grow((size_t)kWin + ((size_t)1 << 20));
        if (window) { for (size_t i = 0; i < (size_t)kWin - have; i++) s.p[i] = 0; for (size_t i = 0; i < have; i++) s.p[(size_t)kWin - have + i] = window[i]; }
        else for (int i = 0; i < kWin; i++) s.p[(size_t)i] = (uint16_t)(256 + i);
        n = 0;
    }
    inline void room(size_t more) { if ((size_t)kWin + n + more > s.cap) s.grow(std::max(s.cap * 2, (size_t)kWin + n + more)); }
};

enum Stop { kBoundary = 0, kFinal = 1, kError = 2 };

// Inflates blocks from bit `from` until a block ends at a bit for which stop_at(bit) is true (kBoundary), the final block ends (kFinal), or the stream
// makes no sense (kError).  text_only: literals outside text end the attempt at once (the search's probe).  max_out: give up beyond that many symbols.
template <class StopAt>
static inline Stop inflate_blocks(const uint8_t *base, const uint8_t *end, uint64_t from, Out &out, uint64_t &end_bit, StopAt stop_at, bool text_only, size_t max_out)
{
    Bits b;
    b.start(base, end, from);
    Codes dyn, fix;
    bool have_fix = false;
    for (;;) {
        b.refill();
        const uint32_t bfinal = b.take(1), btype = b.take(2);
        if (btype == 3) return kError;
        if (btype == 0) {
            b.align();
            b.refill();
            const uint32_t len = b.take(16), nlen = b.take(16);
            if ((len ^ nlen) != 0xFFFFu) return kError;
            out.room(len);
            uint16_t *o = out.s.data() + kWin + out.n;
            for (uint32_t i = 0; i < len; i++) { b.refill(); o[i] = (uint16_t)b.take(8); }
            out.n += len;
            if (b.over) return kError;
        } else {
            Codes *c = &dyn;
            if (btype == 1) { if (!have_fix) { fixed_codes(fix); have_fix = true; } c = &fix; }
            else if (!read_dynamic(b, dyn)) return kError;
            const Ent *lt = c->lit.tab.data(), *dt = c->dist.tab.data();
            const bool no_dist = c->dist_none;
            for (;;) {
                out.room(320);
                uint16_t *o = out.s.data() + kWin;
                size_t n = out.n;
                const size_t stop = out.s.size() - (size_t)kWin - 300;
                bool eob = false;
                while (n < stop) {
                    b.refill(); // >= 56 bits: a literal / length code (15) + length extra (5) + distance code (15) + distance extra (13) = 48
                    Ent e = lt[b.peek(kPBits)];
                    if (e.sub) { b.drop(kPBits); e = lt[e.val + b.peek(e.sub)]; }
                    if (!e.bits) return kError;
                    b.drop(e.bits);
                    int s = e.val;
                    if (s < 256) {
                        if (text_only && !(s >= 32 && s < 127) && s != '\n' && s != '\r' && s != '\t') return kError;
                        o[n++] = (uint16_t)s;
                        // (two more symbols fit what the refill left when they are literals: 56 - 15 - 15 >= 15)
                        e = lt[b.peek(kPBits)];
                        if (e.sub || !e.bits || e.val >= 256) continue;
                        if (text_only && !(e.val >= 32 && e.val < 127) && e.val != '\n' && e.val != '\r' && e.val != '\t') return kError;
                        b.drop(e.bits); o[n++] = e.val;
                        e = lt[b.peek(kPBits)];
                        if (e.sub || !e.bits || e.val >= 256) continue;
                        if (text_only && !(e.val >= 32 && e.val < 127) && e.val != '\n' && e.val != '\r' && e.val != '\t') return kError;
                        b.drop(e.bits); o[n++] = e.val;
                        continue;
                    }
                    if (s == 256) { eob = true; break; }
                    s -= 257;
                    if (s >= 29 || no_dist) return kError;
                    const int len = kLenBase[s] + (int)b.take(kLenExtra[s]);
                    Ent d = dt[b.peek(kPBits)];
                    if (d.sub) { b.drop(kPBits); d = dt[d.val + b.peek(d.sub)]; }
                    if (!d.bits || d.val >= 30) return kError;
                    b.drop(d.bits);
                    const int dist = kDistBase[d.val] + (int)b.take(kDistExtra[d.val]);
                    const uint16_t *src = o + n - dist; // (dist <= kWin: inside the prefix at worst)
                    uint16_t *dst = o + n;
                    if (dist >= 8) { // eight symbols at a time (a piece never overlaps its own source; up to seven symbols too many land in the slack)
                        int k = 0;
                        do { memcpy(dst + k, src + k, 16); k += 8; } while (k < len);
                    } else if (dist == 1) { // a run of one symbol (quality strings are full of them)
                        const uint16_t v = src[0];
                        uint64_t w = v; w |= w << 16; w |= w << 32;
                        int k = 0;
                        do { memcpy(dst + k, &w, 8); memcpy(dst + k + 4, &w, 8); k += 8; } while (k < len);
                    } else for (int k = 0; k < len; k++) dst[k] = src[k];
                    n += (size_t)len;
                }
                out.n = n;
                if (b.over) return kError;
                if (eob) break;
                if (out.n > max_out) return kError;
            }
        }
        end_bit = b.pos();
        if (bfinal) return kFinal;
        if (stop_at(end_bit)) return kBoundary;
        if (out.n > max_out) return kError;
    }
}

// a block start at or behind bit `from`, before bit `limit`: the first candidate that passes (0: none)
static inline uint64_t find_start(const uint8_t *base, const uint8_t *end, uint64_t from, uint64_t limit, Out &probe)
{
    Codes c;
    for (uint64_t bit = from; bit < limit; bit++) {
        const uint8_t *p = base + (bit >> 3);
        if (p + 24 >= end) return 0;
        uint64_t w; memcpy(&w, p, 8);
        w >>= (bit & 7);
        if ((w & 7u) != 4u) continue; // BFINAL 0, BTYPE 2 (LSB first: bit 0 = BFINAL, bits 1-2 = 10b read LSB first -> value 2)
        {   // the cheap part of read_dynamic first, on the word at hand: HLIT <= 29, HDIST <= 29, and the code-length code's lengths (3 bits each, HCLEN + 4
            // of them from bit 17 on) must fill the code space exactly — of random bits about one candidate in a hundred gets past this
            const uint32_t hlit = (uint32_t)(w >> 3) & 31u, hdist = (uint32_t)(w >> 8) & 31u, hclen = ((uint32_t)(w >> 13) & 15u) + 4u;
            if (hlit > 29u || hdist > 29u) continue;
            uint64_t x = w >> 17; // 57 - (bit & 7) >= 50 bits left: sixteen lengths; the last three come from the next bytes
            if (hclen > 13) { uint64_t y; memcpy(&y, p + 8, 8); x |= y << (47 - (bit & 7)); } // (bits 64.. of the stream at this offset)
            uint32_t space = 0;
            for (uint32_t i = 0; i < hclen; i++) { const uint32_t l = (uint32_t)(x >> (3 * i)) & 7u; if (l) space += 128u >> l; }
            if (space != 128u) continue;
        }
        Bits b;
        b.start(base, end, bit + 3);
        if (!read_dynamic(b, c)) continue;
        // the block itself, to its end, as text; then one more block header must follow sensibly (inflate_blocks checks the next header as it goes)
        probe.begin(nullptr, 0);
        uint64_t eb = 0;
        int blocks = 0;
        const Stop st = inflate_blocks(base, end, bit, probe, eb, [&](uint64_t) { return ++blocks >= 2; }, true, (size_t)4 << 20);
        if (st == kError) continue;
        if (probe.n < 64 && st != kFinal) continue; // (a block of a few bytes in the middle of a stream: not something a compressor writes)
        return bit;
    }
    return 0;
}

// A gzip member's header at p (RFC 1952): its length, 0 if it is not one
static inline size_t gzip_header(const uint8_t *p, size_t n)
{
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return 0;
    size_t o = 10;
    const int flg = p[3];
    if (flg & 4) { if (o + 2 > n) return 0; o += 2 + ((size_t)p[o] | ((size_t)p[o + 1] << 8)); }
    if (flg & 8) { while (o < n && p[o]) o++; o++; }
    if (flg & 16) { while (o < n && p[o]) o++; o++; }
    if (flg & 2) o += 2;
    return o < n ? o : 0;
}

// a round's text: plain storage, not cleared when it grows
struct Text {
    char *p = nullptr; size_t n = 0, cap = 0;
    Text() {}
    Text(const Text &) = delete;
    Text &operator=(const Text &) = delete;
    ~Text() { free(p); }
    const char *data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void clear() { n = 0; }
    void resize(size_t m) { if (m > cap) { char *q = (char *)realloc(p, m); if (!q) throw std::bad_alloc(); p = q; cap = m; } n = m; }
};

// symbols -> bytes against the 32 KB window w: sixteen at a time where none of them is a placeholder (most of a stretch, all of a round's first one)
static inline void resolve(const uint16_t *src, size_t n, const uint8_t *w, uint8_t *dst)
{
    size_t i = 0;
    const __m128i lim = _mm_set1_epi16((short)0xFF00);
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128((const __m128i *)(src + i)), b = _mm_loadu_si128((const __m128i *)(src + i + 8));
        if (_mm_movemask_epi8(_mm_cmpeq_epi16(_mm_and_si128(_mm_or_si128(a, b), lim), _mm_setzero_si128())) == 0xFFFF) {
            _mm_storeu_si128((__m128i *)(dst + i), _mm_packus_epi16(a, b));
            continue;
        }
        for (size_t k = i; k < i + 16; k++) { const uint16_t s = src[k]; dst[k] = s < 256 ? (uint8_t)s : w[s - 256]; }
    }
    for (; i < n; i++) { const uint16_t s = src[i]; dst[i] = s < 256 ? (uint8_t)s : w[s - 256]; }
}

// CRC-32 (the gzip polynomial, reflected) by carry-less multiplication: four 128-bit lanes folded by x^512, then down to one, then Barrett's reduction —
// Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction" (Intel, 2009), with that paper's constants for this polynomial.
// zlib's table-driven crc32 runs at 1 GB/s a thread and was a quarter of a thread's time in a round; this runs at 16.  Held to zlib's on random buffers,
// lengths and starting values by tests/test_gz_inflate.py (through the reader: every member's CRC is checked against its trailer).
__attribute__((target("sse4.2,pclmul")))
static inline uint32_t crc32_fold(const uint8_t *buf, size_t len, uint32_t crc) // len >= 64 and a multiple of 16; crc: the raw register (zlib's value inverted)
{
    alignas(16) static const uint64_t k1k2[] = {0x0154442bd4ull, 0x01c6e41596ull}; // x^(512+32) mod P, x^(512-32) mod P
    alignas(16) static const uint64_t k3k4[] = {0x01751997d0ull, 0x00ccaa009eull}; // x^(128+32), x^(128-32)
    alignas(16) static const uint64_t k5k0[] = {0x0163cd6124ull, 0x0000000000ull}; // x^64
    alignas(16) static const uint64_t poly[] = {0x01db710641ull, 0x01f7011641ull}; // P, floor(x^64 / P)
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i *)k1k2);
    buf += 64; len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00); x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11); x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(x1, x5); x2 = _mm_xor_si128(x2, x6); x3 = _mm_xor_si128(x3, x7); x4 = _mm_xor_si128(x4, x8);
        x1 = _mm_xor_si128(x1, y5); x2 = _mm_xor_si128(x2, y6); x3 = _mm_xor_si128(x3, y7); x4 = _mm_xor_si128(x4, y8);
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128((const __m128i *)k3k4);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(x1, x2); x1 = _mm_xor_si128(x1, x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(x1, x3); x1 = _mm_xor_si128(x1, x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(x1, x4); x1 = _mm_xor_si128(x1, x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(x1, x2); x1 = _mm_xor_si128(x1, x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i *)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i *)poly);
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
// zlib's crc32(crc, p, n), by folding where the CPU can
static inline uint32_t fast_crc32(uint32_t crc, const uint8_t *p, size_t n)
{
    static const bool fold = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.2");
    if (fold && n >= 64) { const size_t m = n & ~(size_t)15; crc = ~crc32_fold(p, m, ~crc); p += m; n -= m; }
    for (; n;) { const size_t m = n < ((size_t)1 << 30) ? n : ((size_t)1 << 30); crc = (uint32_t)crc32(crc, p, (uInt)m); p += m; n -= m; }
    return crc;
}

// runs body(0 .. n-1) on the caller's pool (any callable that runs the jobs and returns when all are done)
using ParallelFor = std::function<void(int, const std::function<void(int)> &)>;

// The reader: next() delivers the stream's text in order, a round at a time.
class Reader {
public:
    // data: the whole .gz file (mapped); threads: stretches per round; stretch_bytes: compressed bytes per stretch
    bool open(const uint8_t *data, size_t size, int threads, size_t stretch_bytes, ParallelFor pf)
    {
        d_ = data; size_ = size; threads_ = std::max(1, threads); stretch_ = std::max<size_t>(stretch_bytes, (size_t)1 << 16); pf_ = std::move(pf);
        return next_member(0);
    }
    bool failed() const { return failed_; }
    // the next piece of text into `text` (replaced); false at the end of the input (or where the stream is damaged: failed())
    bool next(Text &text)
    {
        text.clear();
        while (text.empty()) {
            if (done_) return false;
            if (!round(text)) { done_ = true; return !text.empty(); }
        }
        return true;
    }

private:
    const uint8_t *d_ = nullptr; size_t size_ = 0; int threads_ = 1; size_t stretch_ = 0; ParallelFor pf_;
    size_t def_ = 0;          // where the current member's deflate data begins (byte offset)
    uint64_t bit_ = 0;        // the next block's bit, relative to def_
    std::vector<uint8_t> win_; // the last <= kWin bytes of text of this member
    uint32_t crc_ = 0; uint64_t total_ = 0; // of this member so far
    bool done_ = false, failed_ = false;
    std::vector<Out> outs_, probes_;

    bool next_member(size_t at)
    {
        if (at >= size_) { done_ = true; return at == size_ && at > 0; }
        const size_t h = gzip_header(d_ + at, size_ - at);
        if (!h) { done_ = true; failed_ = at == 0; return false; } // (trailing garbage behind a complete member ends the input quietly, as gzread has it)
        def_ = at + h; bit_ = 0; win_.clear(); crc_ = (uint32_t)crc32(0L, Z_NULL, 0); total_ = 0;
        return true;
    }

    bool round(Text &text)
    {
        const uint8_t *base = d_ + def_, *end = d_ + size_;
        const size_t avail = size_ - def_;
        const size_t byte0 = (size_t)(bit_ >> 3);
        int n = threads_;
        while (n > 1 && byte0 + (size_t)n * stretch_ > avail + stretch_) n--; // (no stretch begins beyond the file's end)
        std::vector<uint64_t> start((size_t)n + 1, 0);
        start[0] = bit_;
        const uint64_t round_end = (uint64_t)(byte0 + (size_t)n * stretch_) * 8; // the last stretch runs to the first boundary at or beyond it
        if ((int)outs_.size() < n) { outs_.resize((size_t)n); probes_.resize((size_t)n); }
#ifdef MCX_PGZ_TIMING
        double t_ph[6]; t_ph[0] = pgz_now();
#endif
        // 1. the starts
        if (n > 1) pf_(n - 1, [&](int k) {
            const int i = k + 1;
            const uint64_t lo = (uint64_t)(byte0 + (size_t)i * stretch_) * 8, hi = (uint64_t)(byte0 + (size_t)(i + 1) * stretch_) * 8;
            start[(size_t)i] = find_start(base, end, lo, hi, probes_[(size_t)i]); // (the probe's storage stays with the reader: fresh pages per search were most of its time)
        });
#ifdef MCX_PGZ_TIMING
        t_ph[1] = pgz_now();
#endif
        // 2. every stretch with a start inflates until it ends on a later stretch's start (or, the last: beyond the round's end)
        std::vector<Stop> how((size_t)n, kError);
        std::vector<uint64_t> stop_bit((size_t)n, 0);
        pf_(n, [&](int i) {
            if (i > 0 && !start[(size_t)i]) return;
            Out &o = outs_[(size_t)i];
            if (i == 0) o.begin(win_.data(), win_.size()); else o.begin(nullptr, 0);
            auto stop_at = [&](uint64_t b) {
                if (b >= round_end) return true;
                for (int j = i + 1; j < n; j++) if (start[(size_t)j] == b) return true;
                return false;
            };
            // (a stretch that began on a false start runs on rubbish: it may not outgrow what a stretch of text can be)
            how[(size_t)i] = inflate_blocks(base, end, start[(size_t)i], o, stop_bit[(size_t)i], stop_at, false, i == 0 ? ~(size_t)0 : stretch_ * (size_t)n * 64);
        });
#ifdef MCX_PGZ_TIMING
        t_ph[2] = pgz_now();
#endif
        // 3. the chain: stretch 0, then whichever stretch begins where the last one ended
        std::vector<int> chain;
        int cur = 0;
        bool final_block = false;
        uint64_t at = bit_;
        for (;;) {
            if (how[(size_t)cur] == kError) {
                if (cur == 0) { failed_ = true; return false; }
                break; // (cannot be: a stretch that was reached began on a true boundary)
            }
            chain.push_back(cur);
            at = stop_bit[(size_t)cur];
            if (how[(size_t)cur] == kFinal) { final_block = true; break; }
            int nxt = -1;
            for (int j = cur + 1; j < n; j++) if (start[(size_t)j] == at) { nxt = j; break; }
            if (nxt < 0) break; // ended beyond the round's end: the next round starts there
            cur = nxt;
        }
        // 4. placeholders -> bytes.  The windows in sequence (each stretch's last kWin symbols against the window before it), then all text in parallel.
        size_t total = 0;
        std::vector<size_t> off(chain.size() + 1, 0);
        for (size_t k = 0; k < chain.size(); k++) { off[k] = total; total += outs_[(size_t)chain[k]].n; }
        off[chain.size()] = total;
        std::vector<std::vector<uint8_t>> wins(chain.size() + 1);
        wins[0].assign((size_t)kWin, 0);
        if (!win_.empty()) memcpy(wins[0].data() + (kWin - win_.size()), win_.data(), win_.size());
        for (size_t k = 0; k < chain.size(); k++) {
            const Out &o = outs_[(size_t)chain[k]];
            const std::vector<uint8_t> &w = wins[k];
            std::vector<uint8_t> &nw = wins[k + 1];
            nw.resize((size_t)kWin);
            // the kWin symbols that end at the stretch's end (reaching into the prefix when it is shorter)
            const uint16_t *src = o.s.data() + o.n; // = prefix start + n: the last kWin symbols of prefix + output
            resolve(src, (size_t)kWin, w.data(), nw.data());
        }
#ifdef MCX_PGZ_TIMING
        t_ph[3] = pgz_now();
#endif
        text.resize(total);
        std::vector<uint32_t> crcs(chain.size(), 0);
        pf_((int)chain.size(), [&](int k) {
            const Out &o = outs_[(size_t)chain[(size_t)k]];
            const uint16_t *src = o.s.data() + kWin;
            const uint8_t *w = wins[(size_t)k].data();
            uint8_t *dst = (uint8_t *)text.data() + off[(size_t)k];
            resolve(src, o.n, w, dst);
            crcs[(size_t)k] = fast_crc32((uint32_t)crc32(0L, Z_NULL, 0), dst, o.n);
        });
#ifdef MCX_PGZ_TIMING
        t_ph[4] = pgz_now();
        for (int k = 0; k < 4; k++) pgz_phase[k] += t_ph[k + 1] - t_ph[k];
        pgz_phase[4] += (double)chain.size(); pgz_phase[5] += 1; pgz_phase[6] += n;
#endif
        for (size_t k = 0; k < chain.size(); k++) crc_ = (uint32_t)crc32_combine(crc_, crcs[k], (z_off_t)outs_[(size_t)chain[k]].n);
        total_ += total;
        // the member's window for the next round
        {
            const std::vector<uint8_t> &lw = wins[chain.size()];
            const size_t have = (size_t)std::min<uint64_t>(total_, (uint64_t)kWin);
            win_.assign(lw.end() - (ptrdiff_t)have, lw.end());
        }
        bit_ = at;
        if (final_block) { // trailer: CRC-32 and ISIZE (RFC 1952 2.3.1), then perhaps another member
            const size_t tb = def_ + (size_t)((at + 7) >> 3);
            if (tb + 8 > size_) { failed_ = true; return false; }
            const uint32_t want_crc = (uint32_t)d_[tb] | ((uint32_t)d_[tb + 1] << 8) | ((uint32_t)d_[tb + 2] << 16) | ((uint32_t)d_[tb + 3] << 24);
            const uint32_t want_len = (uint32_t)d_[tb + 4] | ((uint32_t)d_[tb + 5] << 8) | ((uint32_t)d_[tb + 6] << 16) | ((uint32_t)d_[tb + 7] << 24);
            if (want_crc != crc_ || want_len != (uint32_t)total_) { failed_ = true; text.clear(); return false; }
            if (!next_member(tb + 8)) return false; // (the text of this round is still delivered: next() looks at it before at done_)
        }
        return true;
    }
};

} // namespace pgz
} // namespace mcx
#endif
