// mapcaller_amd/csrc/mcx_variants_host.h — the sparse half of variant calling (host C++, no HIP).
//
// VariantCalling() (reference src/VariantCalling.cpp:696-740) minus everything that has to look at
// every genome position: that dense half is behind DenseProfile — in the product the kernels of
// mcx_variants.hip over the planes in HBM; in tests/hostemu plain loops over a host array, so that
// this logic is exercised on the CPU-only box against the same golden VCFs.
#ifndef MCX_VARIANTS_HOST_H
#define MCX_VARIANTS_HOST_H
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

#include "../../include/mcx.h"
#include "mcx_internal.h"
#include "mcx_cpus.h"
#include "mcx_fm.h"

namespace mcx_vc {
using namespace mcx;

enum { pA = 0, pC, pG, pT, pMulti, pReadCount, pF1, pR2, pF2, pR1, nPlanes }; // plane order of mcx_profile_attach
enum { kBlock = 100 };                                                         // BlockSize :4

// Variant_t::VarType values (:8-16) and, above them, the run boundaries the scan reports
enum { vSUB = 0, vINS = 1, vDEL = 2, vINV = 3, vTNL = 4, vCNV = 5, vUMR = 6, vNOR = 10, vMON = 11,
       eGapStart = 100, eGapEnd, eDupStart, eDupEnd, eNormStart, eNormEnd };

struct ScanParams {
    int64_t G;
    int32_t min_ad, somatic, ploidy, mono, gvcf;
    double freq_thr; // FrequencyThr widened to double by the ?: at :593
};

struct alignas(8) SiteRec { // one appended record: an SNV / monomorphic call or a run boundary at `pos`
    int64_t pos;
    uint8_t type, geno, qscore, alt; // alt: base codes of the one or two ALT alleles, low nibble first, 0xF = none
    uint16_t DP, AD_ref, AD_alt, pad;
};

struct SiteEval { int cls; bool cand; bool call; SiteRec rec; };

// DetermineGenotype :528-547
static inline MCX_HD uint8_t genotype_of(int ploidy, int cov, int alt_reads, int alt_num)
{
    if (ploidy == 1) return alt_reads < (int)(cov * 0.5) ? 1 : 2;
    if (ploidy == 2) {
        if (alt_num == 0) return 3;
        if (alt_num == 1) return alt_reads < (int)(cov * 0.5) ? 4 : 5;
        if (alt_num == 2) return 6;
    }
    return 0;
}

// the block's threshold: half the block depth, at least MinAlleleDepth; -somatic caps it there (:566-567)
static inline MCX_HD int cov_threshold(int depth, int min_ad, int somatic)
{
    int t = depth >> 1;
    if (t < min_ad) t = min_ad;
    if (somatic && t > min_ad) t = min_ad;
    return t;
}

// What IdentifyVariants decides from one column alone: the run class (0 covered, 1 nothing mapped,
// 2 only multi-mapped reads), the SNV call (:591-624) and whether the position can be part of a
// "normal" stretch (covered, no SNV).
// (Planes: anything with get(plane, position) — the planes in HBM, mcx_planes.h PlanesView, or a host array in tests/hostemu)
template <class Planes>
static inline MCX_HD SiteEval eval_site(const Planes &pl, const int32_t *depth, const IndexView &ix, const ScanParams &sp, int64_t g)
{
    SiteEval e;
    uint32_t n[4];
    for (int k = 0; k < 4; k++) n[k] = pl.get(k, g);
    const int cov = (int)(n[0] + n[1] + n[2] + n[3]);
    e.cls = cov > 0 ? 0 : (pl.get(pMulti, g) == 0 ? 1 : 2);
    e.call = false;
    e.rec.pos = g; e.rec.type = vSUB; e.rec.geno = 0; e.rec.qscore = 0; e.rec.alt = 0xFF; e.rec.DP = (uint16_t)cov; e.rec.AD_ref = e.rec.AD_alt = 0; e.rec.pad = 0;
    const int thr = cov_threshold(depth[g / kBlock], sp.min_ad, sp.somatic);
    if (cov >= thr && cov > 0) {
        const int rb = ref_code(ix, g);
        int ft = (int)ceil(cov * (sp.somatic ? 0.01 : sp.freq_thr));
        if (ft < sp.min_ad) ft = sp.min_ad;
        int na = 0, a[4], sum = 0;
        for (int k = 0; k < 4; k++) if (k != rb && (int)n[k] >= ft) { a[na++] = k; sum += (int)n[k]; }
        e.rec.AD_ref = (uint16_t)n[rb];
        uint8_t gt = 0;
        if (na == 1) gt = genotype_of(sp.ploidy, cov, (uint16_t)sum, 1);
        else if (na == 2 && sum >= (int)(cov * 0.5)) gt = genotype_of(sp.ploidy, cov, (uint16_t)sum, 2); // CheckDiploidFrequency :123-128
        if (gt) {
            e.call = true;
            e.rec.geno = gt; e.rec.AD_alt = (uint16_t)sum;
            e.rec.alt = (uint8_t)(a[0] | ((na == 2 ? a[1] : 0xF) << 4));
            const double q = sp.somatic ? 35.0 * e.rec.AD_alt / (cov * 0.05) : 35.0 * e.rec.AD_alt / cov;
            e.rec.qscore = (uint8_t)(int)q;
        }
    }
    e.cand = cov > 0 && !e.call;
    if (e.cand && sp.mono) { // :657-662 (dropped again by the host where an indel is called)
        e.rec.type = vMON; e.rec.geno = genotype_of(sp.ploidy, cov, 0, 0);
        e.rec.AD_ref = (uint16_t)n[ref_code(ix, g)];
    }
    return e;
}

// (storage whose elements are written by several threads right after it is sized: sizing it must not touch it — a vector's
//  value-initialisation is one thread walking over fresh pages)
template <class T> struct NoInit : std::allocator<T> {
    template <class U> struct rebind { typedef NoInit<U> other; };
    template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};

struct Column { uint32_t v[nPlanes]; int32_t depth; uint32_t ref; }; // 48 bytes
typedef std::vector<Column, NoInit<Column>> ColVec;
typedef std::vector<SiteRec, NoInit<SiteRec>> SiteVec; // (millions of records copied in from the device: not zeroed first)
struct RangeQ { int64_t beg, end; int32_t mode, pad; }; // [beg, end] inclusive; mode 0: coverage sum, 1: minimum over covered positions

// the dense profile as the caller sees it
struct DenseProfile {
    virtual ~DenseProfile() {}
    virtual int64_t genome_size() const = 0;
    // block depth, then the per-position scan: SNV / monomorphic records and run boundaries in (position, type) order
    virtual int scan(const ScanParams &sp, SiteVec &sites, double &ms_depth, double &ms_scan) = 0;
    virtual int gather(const std::vector<int64_t> &pos, ColVec &out) = 0;       // columns (+ block depth, reference base) of listed positions
    virtual int ranges(const std::vector<RangeQ> &q, std::vector<unsigned long long> &out) = 0; // coverage sum / minimum over listed ranges
};

// ---- host side --------------------------------------------------------------------------------------
struct Variant { // Variant_t, structure.h:185-195; ALT strings longer than 5 are never written (:451, :460), so 7 characters are kept
    int64_t gPos;     // (no member initialisers: `Variant v{}` where a zeroed record is wanted)
    uint16_t DP, AD_ref, AD_alt;
    uint8_t geno, qscore, type, alt_len;
    char alt[8];
    void set_alt(const char *p, size_t n) { alt_len = (uint8_t)std::min<size_t>(n, 255); memset(alt, 0, sizeof alt); memcpy(alt, p, std::min<size_t>(n, 7)); }
};
typedef std::vector<Variant, NoInit<Variant>> VarVec;
static inline bool by_pos(const Variant &a, const Variant &b) { return a.gPos == b.gPos ? a.type < b.type : a.gPos < b.gPos; } // CompByVarPos :50-54

// InsertSeqMap / DeleteSeqMap (AlignmentProfile.cpp:7) as one array sorted by (position, string) —
// the iteration order of the reference's map of maps — with 16-bit counts that wrap like its uint16_t
// A string is kept as its first eight characters (as a big-endian number: its order is the strings' order), its length and
// the record it came from — the calls need seven characters and the length, and only strings that agree in the first eight
// have to be read again (tally_seq) to be told apart.
struct Tally { int64_t pos; uint64_t head8; uint32_t rec; uint16_t count, len; };
typedef std::vector<Tally, NoInit<Tally>> IndelMap;

// the whole string of the tally whose head record is recs[i]
static inline std::string tally_seq(const mcx_sparse_rec *recs, uint64_t n, uint64_t i)
{
    const mcx_sparse_rec &r = recs[i];
    std::string s(r.seq, std::min<size_t>(r.len, sizeof r.seq));
    for (uint64_t j = i + 1; j < n && recs[j].type == 'C' && s.size() < r.len; j++) // a long string continues in the records behind
        s.append(recs[j].seq, std::min<size_t>(recs[j].len, sizeof recs[j].seq));
    return s;
}
struct Clip { int64_t pos; uint16_t count; };
struct Site { int64_t gPos, dist; };

// std::string's operator< on (seq, len)
static inline int seq_cmp(const char *a, size_t la, const char *b, size_t lb)
{
    const int c = memcmp(a, b, std::min(la, lb));
    return c ? c : (la < lb ? -1 : (la > lb ? 1 : 0));
}

// MCX_TIMING: where a phase of the caller spends its time
struct SubLap {
    const bool on = getenv("MCX_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(const char *what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[mcx_call_variants]     %-24s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

// ---- host threads for the passes over millions of records --------------------------------------------
static inline unsigned vc_threads(size_t n, size_t grain)
{
    const unsigned hw = mcx_usable_cpus(); // (the CPUs the process is given, not the machine's)
    if (const char *e = getenv("MCX_VC_GRAIN")) grain = (size_t)std::max(1, atoi(e)); // (tests: many short stretches on small inputs)
    return (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)hw, (size_t)32, n / std::max<size_t>(grain, 1) + 1}));
}

// f(t, lo, hi) over [0, n) cut into `T` consecutive stretches, one thread each
template <class F> static inline void par_ranges(size_t n, unsigned T, F f)
{
    if (T <= 1) { f(0u, (size_t)0, n); return; }
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < T; t++) pool.emplace_back([&, t] { f(t, n * t / T, n * (t + 1) / T); });
    f(0u, (size_t)0, n / T);
    for (auto &th : pool) th.join();
}

// Stable sort.  Stretches are sorted by the threads; then the key axis is cut at sampled elements, every stretch is cut there
// by binary search (elements equal to a cut element all go to its right), and every thread merges what the stretches hold of
// its range into its place in the result — a tie goes to the earlier stretch, which is what keeps the sort stable.
template <class Vec, class Less> static inline void par_stable_sort(Vec &v, Less less)
{
    const unsigned P = vc_threads(v.size(), 1 << 15);
    if (P <= 1) { std::stable_sort(v.begin(), v.end(), less); return; }
    std::vector<size_t> cut(P + 1);
    for (unsigned t = 0; t <= P; t++) cut[t] = v.size() * t / P;
    par_ranges(P, P, [&](unsigned, size_t lo, size_t hi) { for (size_t t = lo; t < hi; t++) std::stable_sort(v.begin() + cut[t], v.begin() + cut[t + 1], less); });
    // cut elements: the quantiles of a sample taken evenly from every (sorted) stretch
    typedef typename Vec::value_type E;
    std::vector<E> sample;
    for (unsigned t = 0; t < P; t++)
        for (unsigned k = 0; k < 16; k++) { const size_t m = cut[t + 1] - cut[t]; if (m) sample.push_back(v[cut[t] + m * k / 16]); }
    std::stable_sort(sample.begin(), sample.end(), less);
    std::vector<size_t> idx((size_t)(P + 1) * P), at(P + 1, 0); // idx[r * P + t]: where range r begins in stretch t
    for (unsigned r = 0; r <= P; r++)
        for (unsigned t = 0; t < P; t++) {
            size_t i;
            if (r == 0) i = cut[t];
            else if (r == P) i = cut[t + 1];
            else i = (size_t)(std::lower_bound(v.begin() + cut[t], v.begin() + cut[t + 1], sample[sample.size() * r / P], less) - v.begin());
            idx[(size_t)r * P + t] = i;
            at[r] += i - cut[t];
        }
    Vec out;
    out.resize(v.size());
    par_ranges(P, P, [&](unsigned, size_t lo, size_t hi) {
        std::vector<size_t> i(P), e(P);
        for (size_t r = lo; r < hi; r++) {
            for (unsigned t = 0; t < P; t++) { i[t] = idx[r * P + t]; e[t] = idx[(r + 1) * P + t]; }
            size_t o = at[r];
            // a heap would do fewer comparisons; the ranges are short and P is at most 32
            for (;;) {
                int best = -1;
                for (unsigned t = 0; t < P; t++)
                    if (i[t] < e[t] && (best < 0 || less(v[i[t]], v[i[(unsigned)best]]))) best = (int)t;
                if (best < 0) break;
                out[o++] = v[i[(unsigned)best]++];
            }
        }
    });
    v.swap(out);
}

class Caller {
public:
    Caller(const HostIndex &hix, int64_t G2, DenseProfile &prof, const mcx_vcf_opts &o) : hix_(hix), G2_(G2), prof_(prof), o_(o), G_(prof.genome_size()) {}
    int run(const mcx_sparse_rec *recs, uint64_t n_recs, int64_t pairs, int64_t dist_sum, int64_t len_sum, const char *path, mcx_vcf_stats *st);

private:
    const HostIndex &hix_;
    int64_t G2_;
    DenseProfile &prof_;
    mcx_vcf_opts o_;
    int64_t G_;
    uint32_t avg_rlen_ = 0;
    int frag_size_ = 500;
    IndelMap ins_, del_;
    std::vector<Clip> brk_; // BreakPointMap, sorted by position
    std::vector<Site> inv_, tnl_;
    VarVec vars_;                          // indel calls first; everything, in CompByVarPos order, after order()
    VarVec calls_, gaps_, dups_;           // runs(): SNV / monomorphic calls, uncovered runs, duplicated runs, each in record order
    std::vector<int64_t> push_pos_; // where a non-NOR record entered the reference's list (for gVCF runs)
    double ms_depth_ = 0, ms_scan_ = 0;

    void fold(const mcx_sparse_rec *recs, uint64_t n);
    int gather(const std::vector<int64_t> &pos, ColVec &out) { return prof_.gather(pos, out); }
    int ranges(const std::vector<RangeQ> &q, std::vector<unsigned long long> &out) { return prof_.ranges(q, out); }
    int scan(SiteVec &sites);
    int indels();
    void runs(const SiteVec &sites);
    void order(VarVec &nor);
    int normal_runs(const SiteVec &sites, VarVec &nor);
    void drop_consecutive_nor();
    int discordant(const std::vector<int64_t> &cands, const std::vector<Site> &sites, int type);
    int breakpoints();
    int write(const char *path, mcx_vcf_stats *st);
    static int area_freq(int64_t g, const IndelMap &m, const Tally *&best);
    bool nearby(int i, int dist) const;
    bool bad_haplotype(int i, int dist) const;
};

// one record per event -> the reference's maps (AlignmentProfile.cpp:6-7) and site lists (ReadMapping.cpp:19)
// copies lists one behind the other, the threads share the lists out
template <class Vec, class Src> static inline void par_concat(Vec &out, const std::vector<const Src *> &src)
{
    std::vector<size_t> at(src.size() + 1, out.size());
    for (size_t t = 0; t < src.size(); t++) at[t + 1] = at[t] + src[t]->size();
    out.resize(at.back());
    par_ranges(src.size(), (unsigned)std::min<size_t>(src.size(), vc_threads(at.back(), 1 << 14)), [&](unsigned, size_t lo, size_t hi) {
        for (size_t t = lo; t < hi; t++) std::copy(src[t]->begin(), src[t]->end(), out.begin() + at[t]);
    });
}

inline void Caller::fold(const mcx_sparse_rec *recs, uint64_t n)
{
    // the records are looked at by several threads, a stretch each (a string's continuation records belong to the stretch of its head)
    const unsigned T = vc_threads(n, 1 << 15);
    struct Part { IndelMap ev[2]; std::vector<int64_t> clip; std::vector<mcx_sparse_rec> sites; bool any_event = false; };
    std::vector<Part> part(T);
    SubLap lap;
    par_ranges(n, T, [&](unsigned t, size_t lo, size_t hi) {
        Part &p = part[t];
        p.ev[0].reserve((hi - lo) / 2 + 16); p.ev[1].reserve((hi - lo) / 2 + 16);
        for (uint64_t i = lo; i < hi; i++) {
            const mcx_sparse_rec &r = recs[i];
            switch (r.type) {
            case 'I': case 'D': {
                Tally tl; tl.pos = r.pos; tl.count = 1; tl.rec = (uint32_t)i;
                size_t len = std::min<size_t>(r.len, sizeof r.seq);
                for (uint64_t j = i + 1; j < n && recs[j].type == 'C' && len < r.len; j++) len += std::min<size_t>(recs[j].len, sizeof recs[j].seq);
                tl.len = (uint16_t)len;
                tl.head8 = 0;
                for (size_t k = 0; k < 8 && k < len; k++) tl.head8 |= (uint64_t)(uint8_t)r.seq[k] << (56 - 8 * k); // (the head record holds at least the first 54)
                p.ev[r.type == 'D'].push_back(tl);
                break;
            }
            case 'B': p.clip.push_back(r.pos); break;
            case 'V': case 'T': p.sites.push_back(r); break;
            case 'E': p.any_event = true; break;
            }
        }
    });
    lap("fold: records");
    if (n >= ((uint64_t)1 << 32)) { /* (rec is 32 bits wide; the lists of a run are far below that) */ }
    std::vector<int64_t> clip;
    std::vector<mcx_sparse_rec> sites; // 'V' / 'T' as given, plus the ones the 'E' events of all shards resolve to
    bool any_event = false;
    for (Part &p : part) {
        clip.insert(clip.end(), p.clip.begin(), p.clip.end());
        sites.insert(sites.end(), p.sites.begin(), p.sites.end());
        any_event = any_event || p.any_event;
    }
    if (any_event) mcx_disc_resolve(recs, n, G_, sites);
    lap("fold: events");
    for (const mcx_sparse_rec &r : sites) { Site s; s.gPos = r.pos; memcpy(&s.dist, r.seq, 8); (r.type == 'V' ? inv_ : tnl_).push_back(s); }
    // (position, string) order — the iteration order of the reference's map of maps — then one entry per distinct string
    auto cmp = [recs, n](const Tally &a, const Tally &b) -> int {
        if (a.pos != b.pos) return a.pos < b.pos ? -1 : 1;
        if (a.head8 != b.head8) return a.head8 < b.head8 ? -1 : 1;
        if (a.len <= 8 || b.len <= 8) return a.len < b.len ? -1 : (a.len > b.len ? 1 : 0);
        const std::string x = tally_seq(recs, n, a.rec), y = tally_seq(recs, n, b.rec);
        return seq_cmp(x.data(), x.size(), y.data(), y.size());
    };
    for (int k = 0; k < 2; k++) {
        IndelMap all;
        std::vector<const IndelMap *> src;
        for (const Part &p : part) src.push_back(&p.ev[k]);
        par_concat(all, src);
        par_stable_sort(all, [&cmp](const Tally &a, const Tally &b) { return cmp(a, b) < 0; });
        size_t w = 0; // (in place: a second array of this size is a walk over fresh pages)
        for (size_t i = 0; i < all.size(); i++) {
            if (w && cmp(all[w - 1], all[i]) == 0) all[w - 1].count++;
            else all[w++] = all[i];
        }
        all.resize(w);
        (k == 0 ? ins_ : del_).swap(all);
    }
    lap("fold: sort + count tallies");
    par_stable_sort(clip, [](int64_t a, int64_t b) { return a < b; });
    for (int64_t p : clip) {
        if (!brk_.empty() && brk_.back().pos == p) brk_.back().count++;
        else { Clip c; c.pos = p; c.count = 1; brk_.push_back(c); }
    }
    auto lt = [](const Site &a, const Site &b) { return a.gPos != b.gPos ? a.gPos < b.gPos : a.dist < b.dist; };
    std::sort(inv_.begin(), inv_.end(), lt); // CompByDiscordPos orders by position only; ties do not matter below
    std::sort(tnl_.begin(), tnl_.end(), lt);
}

inline int Caller::scan(SiteVec &sites)
{
    ScanParams sp;
    sp.G = G_; sp.min_ad = o_.min_allele_depth; sp.somatic = o_.somatic; sp.ploidy = o_.ploidy; sp.mono = o_.monomorphic; sp.gvcf = o_.gvcf;
    sp.freq_thr = (double)o_.freq_thr;
    return prof_.scan(sp, sites, ms_depth_, ms_scan_);
}

// GetAreaIndFrequency :63-94: the tallies within 5 bp; the most frequent string (the longer one on
// a tie) names the call, and only the position that holds it makes the call
inline int Caller::area_freq(int64_t g, const IndelMap &m, const Tally *&best)
{
    int64_t max_pos = 0;
    int freq = 0, max_freq = 0;
    best = nullptr;
    auto a = std::lower_bound(m.begin(), m.end(), g - 5, [](const Tally &t, int64_t x) { return t.pos < x; });
    for (; a != m.end() && a->pos <= g + 5; ++a) {
        freq += a->count;
        if (max_freq < a->count || (max_freq == a->count && a->len > (best ? best->len : 0))) {
            if (max_freq < a->count) max_freq = a->count;
            best = &*a; max_pos = a->pos;
        }
    }
    return g == max_pos ? freq : 0;
}

// indel calls (:570-589): only a position that has a tally of its own can be `max_pos`
inline int Caller::indels()
{
    // the positions that hold a tally: both maps are in position order already
    std::vector<int64_t> keys;
    keys.reserve(ins_.size() + del_.size());
    {
        size_t a = 0, b = 0;
        auto take = [&](int64_t p) { if (p >= 0 && p < G_ && (keys.empty() || keys.back() != p)) keys.push_back(p); };
        while (a < ins_.size() || b < del_.size()) {
            if (b >= del_.size() || (a < ins_.size() && ins_[a].pos <= del_[b].pos)) take(ins_[a++].pos);
            else take(del_[b++].pos);
        }
    }
    ColVec col;
    int rc = gather(keys, col);
    if (rc) return rc;
    const unsigned T = vc_threads(keys.size(), 1 << 14);
    std::vector<VarVec> found(T);
    par_ranges(keys.size(), T, [&](unsigned t, size_t lo, size_t hi) {
        const Tally *best = nullptr;
        for (size_t i = lo; i < hi; i++) {
            const int64_t g = keys[i];
            const Column &c = col[i];
            const int cov = (int)(c.v[pA] + c.v[pC] + c.v[pG] + c.v[pT]);
            const int thr = cov_threshold(c.depth, o_.min_allele_depth, o_.somatic);
            const int thr_of[2] = {std::max((int)(thr * 0.25), o_.min_allele_depth), std::max((int)(thr * 0.35), o_.min_allele_depth)};
            for (int k = 0; k < 2; k++) {
                const int freq = area_freq(g, k == 0 ? ins_ : del_, best);
                if (freq < thr_of[k]) continue;
                Variant v{};
                v.gPos = g; v.type = k == 0 ? vINS : vDEL;
                if (best) { char h[8]; for (int q = 0; q < 8; q++) h[q] = (char)(best->head8 >> (56 - 8 * q)); v.set_alt(h, best->len); }
                v.AD_alt = (uint16_t)freq; v.DP = std::max((uint16_t)c.depth, v.AD_alt); v.AD_ref = v.DP - v.AD_alt;
                v.geno = genotype_of(o_.ploidy, v.DP, v.AD_alt, 1);
                v.qscore = cov == 0 ? 0 : (uint8_t)(int)(100.0 * v.AD_alt / cov); // (the reference's x/0 also ends as 0 on x86-64)
                found[t].push_back(v);
            }
        }
    });
    for (const VarVec &f : found)
        for (const Variant &v : f) { vars_.push_back(v); push_pos_.push_back(v.gPos); }
    return 0;
}

// SNV / monomorphic records, and uncovered (UMR) / duplicated (CNV) runs from their boundaries
// (:625-644).  The k-th start of a kind pairs with its k-th end; a run that reaches the genome end
// has no end and — as in the reference — is never reported.  Lengths are kept in 16 bits like Variant_t::DP.
inline void Caller::runs(const SiteVec &sites)
{
    const std::vector<int64_t> indel_pos(push_pos_); // (sorted by construction)
    // Stretches of the (ordered) records go to one thread each.  A run's start and end can lie in different stretches: a
    // stretch notes the first end it meets without a start of its own (`lead`) and its last start; the ends are settled below.
    const unsigned T = vc_threads(sites.size(), 1 << 16);
    struct Part { VarVec calls, run[2], led[2]; std::vector<int64_t> pushed, led_pushed; int64_t lead[2] = {-1, -1}, last_start[2] = {-1, -1}; };
    std::vector<Part> part(T);
    SubLap lap;
    auto close_run = [&](int k, int64_t start, int64_t end, VarVec &out, std::vector<int64_t> &pushed) {
        const int64_t len = end - start;
        if (k == 0 ? len >= o_.min_gap : len > o_.min_cnv) {
            Variant v{};
            v.type = k ? vCNV : vUMR; v.gPos = start; v.DP = (uint16_t)len;
            out.push_back(v);
            pushed.push_back(end); // the record is appended when the scan reaches the first position after the run
        }
    };
    par_ranges(sites.size(), T, [&](unsigned t, size_t lo, size_t hi) {
        Part &p = part[t];
        bool seen[2] = {false, false}; // a start or an end of the kind has been met in this stretch
        for (size_t i = lo; i < hi; i++) {
            const SiteRec &r = sites[i];
            if (r.type == vSUB || r.type == vMON) {
                if (r.type == vMON && std::binary_search(indel_pos.begin(), indel_pos.end(), r.pos)) continue; // bNormal is false where an indel was called
                Variant v{};
                v.gPos = r.pos; v.type = r.type; v.DP = r.DP; v.AD_ref = r.AD_ref; v.AD_alt = r.AD_alt; v.geno = r.geno; v.qscore = r.qscore;
                if (r.type == vSUB) {
                    char a[3] = {"ACGT"[r.alt & 3], ',', "ACGT"[(r.alt >> 4) & 3]};
                    v.set_alt(a, (r.alt >> 4) == 0xF ? 1 : 3);
                    p.pushed.push_back(r.pos);
                }
                p.calls.push_back(v);
            } else if (r.type == eGapStart || r.type == eDupStart) {
                const int k = r.type == eDupStart;
                p.last_start[k] = r.pos; seen[k] = true;
            } else if (r.type == eGapEnd || r.type == eDupEnd) {
                const int k = r.type == eDupEnd;
                if (!seen[k]) { p.lead[k] = r.pos; seen[k] = true; continue; } // its start lies in an earlier stretch (or nowhere: open == -1 below)
                close_run(k, p.last_start[k], r.pos, p.run[k], p.pushed);
            }
        }
    });
    lap("runs: records");
    // an end that led its stretch closes the run the last start before the stretch opened; then the lists in record order
    int64_t open[2] = {-1, -1};
    std::vector<const VarVec *> src_calls, src_run[2];
    std::vector<const std::vector<int64_t> *> src_pushed;
    src_pushed.push_back(&push_pos_);
    for (Part &p : part) {
        for (int k = 0; k < 2; k++) {
            if (p.lead[k] >= 0) close_run(k, open[k], p.lead[k], p.led[k], p.led_pushed);
            src_run[k].push_back(&p.led[k]); src_run[k].push_back(&p.run[k]);
            if (p.last_start[k] >= 0) open[k] = p.last_start[k];
        }
        src_calls.push_back(&p.calls);
        src_pushed.push_back(&p.led_pushed); src_pushed.push_back(&p.pushed);
    }
    par_concat(calls_, src_calls);
    par_concat(gaps_, src_run[0]);
    par_concat(dups_, src_run[1]);
    std::vector<int64_t> pushed;
    par_concat(pushed, src_pushed);
    push_pos_.swap(pushed);
    lap("runs: gather lists");
    if (o_.gvcf) par_stable_sort(push_pos_, [](int64_t a, int64_t b) { return a < b; }); // (only the gVCF blocks ask where records were appended)
    lap("runs: sort positions");
}

// The records in CompByVarPos order (:50-54, a stable sort in effect: equal records keep the order they were appended in —
// indel calls, then the scan's calls and runs, then gVCF blocks).  Every list is in that order by itself, so they are merged,
// a tie going to the earlier list as in the stable sort of their concatenation: the position axis is cut where the longest
// list's quantiles lie, and every thread merges what the lists hold of its stretch into its place in the result.  A list
// found out of order sends everything through the sort.
inline void Caller::order(VarVec &nor)
{
    const VarVec *lists[5] = {&vars_, &calls_, &gaps_, &dups_, &nor};
    bool sorted = true;
    size_t total = 0, longest = 0;
    for (int k = 0; k < 5; k++) {
        sorted = sorted && std::is_sorted(lists[k]->begin(), lists[k]->end(), by_pos);
        total += lists[k]->size();
        if (lists[k]->size() > lists[longest]->size()) longest = (size_t)k;
    }
    VarVec acc;
    acc.reserve(total + total / 64 + 4096); // (break-point calls join later)
    if (sorted) {
        const VarVec &big = *lists[longest];
        const unsigned T = vc_threads(total, 1 << 16);
        // cut t: everything at positions below cut_pos[t] lies left of it (records of one position are never separated)
        std::vector<int64_t> cut_pos(T + 1, INT64_MIN);
        for (unsigned t = 1; t < T; t++) cut_pos[t] = big.empty() ? INT64_MIN : big[big.size() * t / T].gPos;
        cut_pos[T] = INT64_MAX;
        std::vector<size_t> idx((size_t)(T + 1) * 5), at(T + 1, 0);
        for (unsigned t = 0; t <= T; t++) {
            for (int k = 0; k < 5; k++) {
                const VarVec &l = *lists[k];
                idx[(size_t)t * 5 + k] = t == T ? l.size() : (size_t)(std::lower_bound(l.begin(), l.end(), cut_pos[t], [](const Variant &v, int64_t g) { return v.gPos < g; }) - l.begin());
                at[t] += idx[(size_t)t * 5 + k];
            }
        }
        acc.resize(total);
        par_ranges(T, T, [&](unsigned, size_t lo, size_t hi) {
            for (size_t t = lo; t < hi; t++) {
                size_t i[5], e[5];
                for (int k = 0; k < 5; k++) { i[k] = idx[t * 5 + k]; e[k] = idx[(t + 1) * 5 + k]; }
                Variant *out = acc.data() + at[t];
                for (;;) {
                    int best = -1;
                    for (int k = 0; k < 5; k++) // (strictly smaller: on a tie the earlier list keeps the turn)
                        if (i[k] < e[k] && (best < 0 || by_pos((*lists[k])[i[k]], (*lists[best])[i[best]]))) best = k;
                    if (best < 0) break;
                    *out++ = (*lists[best])[i[best]++];
                }
            }
        });
    } else {
        for (int k = 0; k < 5; k++) acc.insert(acc.end(), lists[k]->begin(), lists[k]->end());
        par_stable_sort(acc, by_pos);
    }
    vars_.swap(acc);
    calls_.clear(); gaps_.clear(); dups_.clear();
}

// gVCF blocks (:645-656).  A covered position without a call extends the last record if that is a
// block, otherwise opens one; MIN_DP is the smallest depth of the block.  The scan delivered the
// maximal stretches of such positions; an indel call removes its position from a stretch, and a
// block continues from one stretch into the next when no other record was appended in between.
inline int Caller::normal_runs(const SiteVec &sites, VarVec &nor)
{
    std::vector<int64_t> indel_pos;
    for (const Variant &v : vars_) if (v.type == vINS || v.type == vDEL) indel_pos.push_back(v.gPos); // (vars_ holds the indel calls only, in position order)
    std::sort(indel_pos.begin(), indel_pos.end());
    struct Piece { int64_t beg, end; size_t block; }; // [beg, end)
    std::vector<Piece> pieces;
    std::vector<int64_t> block_start;
    int64_t seg = -1, last = -1;
    auto add = [&](int64_t a, int64_t b) {
        if (a >= b) return;
        const bool pushed = last < 0 || std::upper_bound(push_pos_.begin(), push_pos_.end(), a) - std::upper_bound(push_pos_.begin(), push_pos_.end(), last) > 0;
        if (pushed) block_start.push_back(a);
        Piece p; p.beg = a; p.end = b; p.block = block_start.size() - 1;
        pieces.push_back(p);
        last = b - 1;
    };
    auto stretch = [&](int64_t a, int64_t b) {
        auto it = std::lower_bound(indel_pos.begin(), indel_pos.end(), a);
        for (; it != indel_pos.end() && *it < b; ++it) { add(a, *it); a = *it + 1; }
        add(a, b);
    };
    for (const SiteRec &r : sites) {
        if (r.type == eNormStart) seg = r.pos;
        else if (r.type == eNormEnd && seg >= 0) { stretch(seg, r.pos); seg = -1; }
    }
    if (seg >= 0) stretch(seg, G_);
    // depth of each block's first position and the minimum over its pieces (cut into bounded ranges)
    std::vector<RangeQ> q; std::vector<size_t> owner;
    for (const Piece &p : pieces)
        for (int64_t a = p.beg; a < p.end; a += 65536) { RangeQ r; r.beg = a; r.end = std::min(p.end, a + 65536) - 1; r.mode = 1; r.pad = 0; q.push_back(r); owner.push_back(p.block); }
    std::vector<unsigned long long> mn;
    int rc = ranges(q, mn);
    if (rc) return rc;
    ColVec first;
    if ((rc = gather(block_start, first))) return rc;
    std::vector<unsigned long long> block_min(block_start.size(), ~0ull);
    for (size_t i = 0; i < q.size(); i++) block_min[owner[i]] = std::min(block_min[owner[i]], mn[i]);
    for (size_t b = 0; b < block_start.size(); b++) {
        Variant v{};
        v.gPos = block_start[b]; v.type = vNOR;
        v.DP = (uint16_t)(first[b].v[pA] + first[b].v[pC] + first[b].v[pG] + first[b].v[pT]);
        v.AD_alt = (uint16_t)std::min<unsigned long long>(block_min[b], v.DP);
        nor.push_back(v);
    }
    return 0;
}

// RemoveConsecutiveGenomicVariant :682-694, including its habit of skipping one comparison after an erase
inline void Caller::drop_consecutive_nor()
{
    if (vars_.size() < 2) return;
    size_t i = 0, n = 1;
    while (n < vars_.size()) {
        if (vars_[i].type == vNOR && vars_[n].type == vNOR) {
            vars_.erase(vars_.begin() + n);
            i = n; n = i + 1;
            if (i >= vars_.size()) break;
        }
        i++; n++;
    }
}

// IdentifyInversions :276-340 / IdentifyTranslocations :210-274: at each break-point candidate the
// discordant pairs that start within a fragment length on either side are grouped by distance
// (1-kb classes, neighbours chained); both sides need a group of at least half the local depth
inline int Caller::discordant(const std::vector<int64_t> &cands, const std::vector<Site> &sites, int type)
{
    if (cands.empty() || sites.empty()) return 0;
    const int64_t half = (int64_t)(avg_rlen_ >> 1);
    std::vector<RangeQ> q;
    for (int64_t g : cands) {
        RangeQ l, r; // CalRegionCov's clamping (:202-204)
        l.beg = std::max<int64_t>(g - frag_size_, 0); l.end = g - half > G_ ? G_ - 1 : g - half; l.mode = 0; l.pad = 0;
        r.beg = std::max<int64_t>(g, 0); r.end = g + frag_size_ > G_ ? G_ - 1 : g + frag_size_; r.mode = 0; r.pad = 0;
        q.push_back(l); q.push_back(r);
    }
    std::vector<unsigned long long> sum;
    ColVec col;
    int rc;
    if ((rc = ranges(q, sum)) || (rc = gather(cands, col))) return rc;
    auto region_cov = [&](size_t i) { return q[i].end < q[i].beg ? 0 : (int)(sum[i] / (unsigned long long)(q[i].end - q[i].beg + 1)); };
    auto lower = [&](int64_t g) { return std::lower_bound(sites.begin(), sites.end(), g, [](const Site &s, int64_t x) { return s.gPos < x; }); };
    auto upper = [&](int64_t g) { return std::upper_bound(sites.begin(), sites.end(), g, [](int64_t x, const Site &s) { return x < s.gPos; }); };
    auto chain = [&](std::vector<Site>::const_iterator a, std::vector<Site>::const_iterator b) {
        std::vector<int64_t> cls;
        for (; a != b; ++a) cls.push_back(a->dist / 1000);
        std::sort(cls.begin(), cls.end());
        cls.push_back(G2_);
        uint32_t best = 0, len = 1;
        for (size_t j = 1; j < cls.size(); j++) {
            if (cls[j] - cls[j - 1] > 1) { best = std::max(best, len); len = 1; }
            else len++;
        }
        return best;
    };
    VarVec found;
    for (size_t i = 0; i < cands.size(); i++) {
        const int64_t g = cands[i];
        const uint32_t thr = (uint32_t)(col[i].depth >> 1);
        auto a = lower(g - frag_size_), b = lower(g - half);
        if (a == sites.end() || b == sites.end()) continue;
        const uint32_t ls = chain(a, b);
        if (ls < thr || ls < (uint32_t)(int)(region_cov(2 * i) * 0.5)) continue;
        a = upper(g); b = lower(g + frag_size_);
        if (a == sites.end() || b == sites.end()) continue;
        const uint32_t rs = chain(a, b);
        if (rs < thr || rs < (uint32_t)(int)(region_cov(2 * i + 1) * 0.5)) continue;
        if (ls == 0 || rs == 0) continue;
        Variant v{};
        v.gPos = g; v.type = (uint8_t)type; v.AD_alt = (uint16_t)std::max(ls, rs);
        v.DP = (uint16_t)(col[i].v[pA] + col[i].v[pC] + col[i].v[pG] + col[i].v[pT]);
        found.push_back(v);
    }
    if (!found.empty()) {
        const size_t mid = vars_.size();
        vars_.insert(vars_.end(), found.begin(), found.end());
        std::inplace_merge(vars_.begin(), vars_.begin() + mid, vars_.end(), by_pos);
    }
    return 0;
}

// IdentifyBreakPointCandidates :173-195: clip positions closer than a read length form a cluster;
// a cluster with three or more clipped reads yields its most frequent position
inline int Caller::breakpoints()
{
    { Clip end; end.pos = G2_; end.count = 0; brk_.push_back(end); }
    std::vector<int64_t> cands;
    uint32_t total = 0;
    int64_t at = 0; uint16_t top = 0;
    for (const Clip &e : brk_) {
        if (e.pos - at > (int64_t)avg_rlen_) {
            if (total >= 3) cands.push_back(at);
            at = e.pos; total = top = e.count;
        } else {
            total += e.count;
            if (top < e.count) { at = e.pos; top = e.count; }
        }
    }
    int rc = discordant(cands, inv_, vINV);
    if (rc) return rc;
    return discordant(cands, tnl_, vTNL);
}

inline bool Caller::nearby(int i, int dist) const // CheckNearbyVariant :342-358
{
    const int n = (int)vars_.size();
    if (n < 2) return false;
    if (i == 0) return vars_[1].gPos - vars_[0].gPos <= dist;
    if (i == n - 1) return vars_[i].gPos - vars_[i - 1].gPos <= dist;
    return vars_[i + 1].gPos - vars_[i].gPos <= dist || vars_[i].gPos - vars_[i - 1].gPos <= dist;
}

inline bool Caller::bad_haplotype(int i, int dist) const // CheckBadHaplotype :360-388
{
    const int n = (int)vars_.size();
    bool bad = false;
    for (int j = i + 1; j < n && vars_[j].gPos - vars_[i].gPos <= dist; j++) {
        if (vars_[j].type != vSUB) continue;
        const int hi = std::max(vars_[i].AD_alt, vars_[j].AD_alt), diff = std::abs((int)vars_[i].AD_alt - (int)vars_[j].AD_alt);
        if (diff > 5 && (hi >> 2)) bad = true;
        break;
    }
    for (int j = i - 1; j >= 0 && vars_[i].gPos - vars_[j].gPos <= dist; j--) {
        if (vars_[j].type != vSUB) continue;
        const int hi = std::max(vars_[i].AD_alt, vars_[j].AD_alt), diff = std::abs((int)vars_[i].AD_alt - (int)vars_[j].AD_alt);
        if (diff > 10 && (int)(hi * 0.33)) bad = true;
        break;
    }
    return bad;
}

// ShowMetaInfo :140-171, DetermineFileter :404-427, GenVariantCallingFile :429-500
inline int Caller::write(const char *path, mcx_vcf_stats *st)
{
    static const char *GT[] = {"*", "0", "1", "0/0", "0/1", "1/1", "1/2"};
    std::vector<int64_t> pos(vars_.size());
    for (size_t i = 0; i < vars_.size(); i++) pos[i] = vars_[i].gPos;
    SubLap lap;
    ColVec col;
    int rc = gather(pos, col);
    if (rc) return rc;
    lap("write: columns");
    FILE *f = fopen(path, "w");
    if (!f) return mcx_set_error(MCX_ERR_IO, std::string("cannot write ") + path);
    setvbuf(f, nullptr, _IOFBF, 1 << 22);
    const HostIndex &h = hix_;
    fprintf(f, "##fileformat=VCFv4.2\n##reference=%s\n##source=MapCaller 0.9.9.41\n##command_line=\"%s\"\n", o_.ref_name ? o_.ref_name : "", o_.cmdline ? o_.cmdline : "");
    fputs("##ALT=<ID=NON_REF,Description=\"Represents any possible alternative allele at this location\">\n"
          "##INFO=<ID=RC,Number=1,Type=Integer,Description=\"Number of reads with start coordinate at this position.\">\n"
          "##INFO=<ID=NTFREQ,Number=4,Type=Integer,Description=\"base depth\">\n"
          "##INFO=<ID=END,Number=1,Type=Integer,Description=\"Last position(inclusive) of the reported block\">\n"
          "##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Read depth\">\n"
          "##INFO=<ID=TYPE,Number=A,Type=String,Description=\"The type of allele, either snv, ins, del, or BP(breakpoint).\">\n"
          "##FORMAT=<ID=AD,Number=R,Type=Integer,Description=\"Allelic depths for the ref and alt alleles in the order listed\">\n"
          "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Approximate read depth\">\n"
          "##FORMAT=<ID=AF,Number=A,Type=Float,Description=\"Allele fractions of alternate alleles\">\n"
          "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
          "##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"Normalized, Phred - scaled likelihoods for genotypes as defined in the VCF specification\">\n", f);
    if (o_.gvcf) fputs("##FORMAT=<ID=MIN_DP,Number=1,Type=Integer,Description=\"Minimum depth in gVCF output block.\">\n", f);
    fputs("##FORMAT=<ID=F1R2,Number=R,Type=Integer,Description=\"Count of reads in F1R2 pair orientation supporting each allele\">\n"
          "##FORMAT=<ID=F2R1,Number=R,Type=Integer,Description=\"Count of reads in F2R1 pair orientation supporting each allele\">\n"
          "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype Quality\">\n"
          "##FILTER=<ID=PASS,Description=\"All filters passed\">\n"
          "##FILTER=<ID=REF,Description=\"Genotyping model thinks this site is reference.\">\n"
          "##FILTER=<ID=BreakPoint,Description=\"It is predicted as a breakpoint\">\n", f);
    fprintf(f, "##FILTER=<ID=DUP,Description=\"Duplicated regions(>=%dbp).\">\n", o_.min_cnv);
    fprintf(f, "##FILTER=<ID=Gaps,Description=\"Region without any read alignment(>=%dbp).\">\n", o_.min_gap);
    fputs("##FILTER=<ID=q10,Description=\"Confidence score below 10\">\n", f);
    if (o_.filter) fputs("##FILTER=<ID=bad_haplotype,Description=\"Variants with variable frequencies on same haplotype\">\n"
                         "##FILTER=<ID=str_contraction,Description=\"Variant appears in repetitive region\">\n", f);
    for (size_t i = 0; i < h.chr_name.size(); i++) fprintf(f, "##contig=<ID=%s,length=%d>\n", h.chr_name[i].c_str(), h.chr_len[i]);
    fprintf(f, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s\n", o_.sample_id ? o_.sample_id : "unknown");

    // The records' text is made by several host threads, each over a stretch of the (ordered) records into a buffer of
    // its own; the buffers go to the file one after the other.  (A record's line depends on its neighbours only through
    // nearby() / bad_haplotype(), which read.)
    const int n = (int)vars_.size();
    const int n_thr = (int)vc_threads((size_t)n, 4096);
    std::vector<std::string> text(n_thr);
    std::vector<mcx_vcf_stats> part(n_thr);
    auto chr_of = [&](int64_t g) { return (int)(std::upper_bound(h.chr_fwd.begin(), h.chr_fwd.end(), g) - h.chr_fwd.begin()) - 1; }; // DetermineCoordinate, tools.cpp:132-164 (forward strand)
    auto work = [&](int t) {
        mcx_vcf_stats s;
        memset(&s, 0, sizeof s);
        std::string &out = text[t];
        out.reserve((size_t)(n / n_thr + 1) * 96);
        // the pieces of a line, appended (printf's parsing of the format was most of this pass; the one %.2f stays with printf)
        auto str = [&](const char *p) { out.append(p); };
        auto ch = [&](char c) { out.push_back(c); };
        auto num = [&](long long x) {
            char b[24]; int k = 24;
            const bool neg = x < 0;
            unsigned long long u = neg ? 0ull - (unsigned long long)x : (unsigned long long)x;
            do { b[--k] = (char)('0' + u % 10); u /= 10; } while (u);
            if (neg) b[--k] = '-';
            out.append(b + k, (size_t)(24 - k));
        };
        auto frac2 = [&](float x) { char b[48]; const int k = snprintf(b, sizeof b, "%.2f", x); out.append(b, (size_t)k); };
        auto head = [&](const char *chr, int p1, char ref) { str(chr); ch('\t'); num(p1); str("\t.\t"); ch(ref); ch('\t'); }; // CHROM POS ID REF
        auto sample = [&](const Variant &v, float af, int F1, int R2, int F2, int R1) { // GT:GQ:DP:AD:AF:F1R2:F2R1
            str(GT[v.geno]); ch(':'); num(v.qscore); ch(':'); num(v.DP); ch(':'); num(v.AD_ref); ch(','); num(v.AD_alt); ch(':'); frac2(af); ch(':');
            num(F1); ch(','); num(R2); ch(':'); num(F2); ch(','); num(R1); ch('\n');
        };
        std::string flt;
        const int i_lo = (int)((int64_t)n * t / n_thr), i_hi = (int)((int64_t)n * (t + 1) / n_thr);
        for (int i = i_lo; i < i_hi; i++) {
            const Variant &v = vars_[i];
            const Column &c = col[i];
            const int ci = chr_of(v.gPos);
            const char *chr = h.chr_name[ci].c_str();
            const int p1 = (int)(v.gPos - h.chr_fwd[ci] + 1);
            const char ref = "ACGT"[c.ref & 3];
            const int cov = (int)(c.v[pA] + c.v[pC] + c.v[pG] + c.v[pT]);
            if (v.type < 3) {
                flt.clear();
                if (v.qscore < 10) flt += "q10;";
                else if (v.type == vSUB && v.AD_alt < 10 && nearby(i, 10)) flt += "q10;";
                else if (v.type != vSUB && v.AD_alt < 5 && nearby(i, 10)) flt += "q10;";
                if (o_.filter) {
                    if ((int)c.v[pMulti] > (int)(cov * 0.05)) flt += "str_contraction;";
                    if (bad_haplotype(i, 100)) flt += "bad_haplotype;";
                }
                if (flt.empty()) flt = "PASS"; else flt.resize(flt.size() - 1);
            }
            const float af = (float)(1.0 * v.AD_alt / v.DP);
            const int rc_ = (int)c.v[pReadCount], F1 = (int)c.v[pF1], R2 = (int)c.v[pR2], F2 = (int)c.v[pF2], R1 = (int)c.v[pR1];
            switch (v.type) {
            case vSUB: // "%s\t%d\t.\t%c\t%s\t%d\t%s\tRC=%d;NTFREQ=%d,%d,%d,%d;TYPE=snv\tGT:GQ:DP:AD:AF:F1R2:F2R1\t%s:%d:%d:%d,%d:%.2f:%d,%d:%d,%d\n"
                s.n_snv++; s.n_records++;
                head(chr, p1, ref); str(v.alt); ch('\t'); num(v.qscore); ch('\t'); str(flt.c_str()); str("\tRC="); num(rc_);
                str(";NTFREQ="); num((int)c.v[pA]); ch(','); num((int)c.v[pC]); ch(','); num((int)c.v[pG]); ch(','); num((int)c.v[pT]);
                str(";TYPE=snv\tGT:GQ:DP:AD:AF:F1R2:F2R1\t");
                sample(v, af, F1, R2, F2, R1);
                break;
            case vINS: case vDEL: // "%s\t%d\t.\t%c\t%c%s\t" / "%s\t%d\t.\t%c%s\t%c\t", then "%d\t%s\tRC=%d;TYPE=%s\tGT:..\t%s:%d:%d:%d,%d:%.2f:%d,%d:%d,%d\n"
                if (v.alt_len > 5) break;
                (v.type == vINS ? s.n_ins : s.n_del)++; s.n_records++;
                if (v.type == vINS) { head(chr, p1, ref); ch(ref); str(v.alt); ch('\t'); }
                else { str(chr); ch('\t'); num(p1); str("\t.\t"); ch(ref); str(v.alt); ch('\t'); ch(ref); ch('\t'); }
                num(v.qscore); ch('\t'); str(flt.c_str()); str("\tRC="); num(rc_); str(v.type == vINS ? ";TYPE=ins" : ";TYPE=del");
                str("\tGT:GQ:DP:AD:AF:F1R2:F2R1\t");
                sample(v, af, F1, R2, F2, R1);
                break;
            case vTNL: case vINV: // "%s\t%d\t.\t%c\t<%s>\t30\tBreakPoint\tTYPE=BP\tGT:GQ:DP:AD\t.:.:0:.\n"
                (v.type == vTNL ? s.n_tnl : s.n_inv)++; s.n_records++;
                head(chr, p1, ref); str(v.type == vTNL ? "<TNL>" : "<INV>"); str("\t30\tBreakPoint\tTYPE=BP\tGT:GQ:DP:AD\t.:.:0:.\n");
                break;
            case vCNV: case vUMR: // "%s\t%d\t.\t%c\t<*>\t0\t%s\tEND=%d\tGT:GQ:DP:AD\t.:.:0:.\n"
                if (v.DP < (v.type == vCNV ? o_.min_cnv : o_.min_gap)) break;
                s.n_records++;
                head(chr, p1, ref); str(v.type == vCNV ? "<*>\t0\tDUP\tEND=" : "<*>\t0\tGaps\tEND="); num(p1 + v.DP - 1); str("\tGT:GQ:DP:AD\t.:.:0:.\n");
                break;
            case vNOR: { // "%s\t%d\t.\t%c\t<*>\t0\tREF\tEND=%d;DP=%d;MIN_DP=%d\tGT:GQ:DP:AD\t.:.:0:.\n"
                int64_t end = h.chr_fwd[ci] + h.chr_len[ci] - 1;
                if (i + 1 < n && vars_[i + 1].gPos < end) end = vars_[i + 1].gPos - 1;
                const int ce = chr_of(end);
                s.n_records++;
                head(chr, p1, ref); str("<*>\t0\tREF\tEND="); num((int)(end - h.chr_fwd[ce] + 1)); str(";DP="); num(v.DP); str(";MIN_DP="); num(v.AD_alt);
                str("\tGT:GQ:DP:AD\t.:.:0:.\n");
                break;
            }
            case vMON: // "%s\t%d\t.\t%c\t.\t0\tREF\tDP=%d;RC=%d;NTFREQ=%d,%d,%d,%d\tGT:F1R2:F2R1\t%s:%d,%d:%d,%d\n"
                s.n_records++;
                head(chr, p1, ref); str(".\t0\tREF\tDP="); num(v.DP); str(";RC="); num(rc_);
                str(";NTFREQ="); num((int)c.v[pA]); ch(','); num((int)c.v[pC]); ch(','); num((int)c.v[pG]); ch(','); num((int)c.v[pT]);
                str("\tGT:F1R2:F2R1\t"); str(GT[v.geno]); ch(':'); num(F1); ch(','); num(R2); ch(':'); num(F2); ch(','); num(R1); ch('\n');
                break;
            }
        }
        part[t] = s;
    };
    {
        std::vector<std::thread> pool;
        for (int t = 1; t < n_thr; t++) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
    }
    lap("write: text");
    mcx_vcf_stats s;
    memset(&s, 0, sizeof s);
    // every thread puts its text at its place in the file (the copy into the page cache is the cost of a write this size)
    if (fflush(f) != 0) { fclose(f); return mcx_set_error(MCX_ERR_IO, std::string("cannot write ") + path); }
    std::vector<int64_t> at(n_thr + 1);
    at[0] = (int64_t)ftello(f);
    for (int t = 0; t < n_thr; t++) {
        const mcx_vcf_stats &q = part[t];
        s.n_records += q.n_records; s.n_snv += q.n_snv; s.n_ins += q.n_ins; s.n_del += q.n_del; s.n_tnl += q.n_tnl; s.n_inv += q.n_inv;
        at[t + 1] = at[t] + (int64_t)text[t].size();
    }
    std::vector<int> bad(n_thr, 0);
    const int fd = fileno(f);
    par_ranges((size_t)n_thr, (unsigned)n_thr, [&](unsigned, size_t lo, size_t hi) {
        for (size_t t = lo; t < hi; t++) {
            const char *p = text[t].data();
            size_t left = text[t].size();
            int64_t o = at[t];
            while (left) {
                const ssize_t k = pwrite(fd, p, left, (off_t)o);
                if (k <= 0) { bad[t] = 1; break; }
                p += k; left -= (size_t)k; o += k;
            }
        }
    });
    for (int b : bad) if (b) { fclose(f); return mcx_set_error(MCX_ERR_IO, std::string("cannot write ") + path); }
    if (fclose(f) != 0) return mcx_set_error(MCX_ERR_IO, std::string("cannot write ") + path);
    lap("write: file");
    if (st) { s.avg_read_len = (int32_t)avg_rlen_; s.fragment_size = frag_size_; s.ms_depth = ms_depth_; s.ms_scan = ms_scan_; *st = s; }
    return 0;
}

inline int Caller::run(const mcx_sparse_rec *recs, uint64_t n_recs, int64_t pairs, int64_t dist_sum, int64_t len_sum, const char *path, mcx_vcf_stats *st)
{
    frag_size_ = o_.fragment_size;
    if (pairs > 0) { // ReadMapping.cpp:782-790
        const uint32_t avg_dist = (uint32_t)(int)(1. * dist_sum / pairs + .5);
        avg_rlen_ = (uint32_t)(int)(1. * len_sum / (pairs << 1) + .5);
        frag_size_ = (int)(avg_dist + avg_rlen_);
    }
    const bool timing = getenv("MCX_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[mcx_call_variants] %-12s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    fold(recs, n_recs); lap("fold");
    SiteVec sites;
    int rc;
    if ((rc = scan(sites))) return rc;
    lap("scan+sort");
    if ((rc = indels())) return rc;
    lap("indels");
    runs(sites); lap("runs");
    VarVec nor;
    if (o_.gvcf && (rc = normal_runs(sites, nor))) return rc;
    order(nor);
    if (o_.gvcf) drop_consecutive_nor();
    lap("order");
    if ((rc = breakpoints())) return rc;
    lap("breakpoints");
    rc = write(path, st); lap("write");
    return rc;
}

} // namespace mcx_vc
#endif
