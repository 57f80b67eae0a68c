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

#include "../../include/mcx.h"
#include "mcx_internal.h"
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
static inline MCX_HD SiteEval eval_site(const uint32_t *pl, const int32_t *depth, const IndexView &ix, const ScanParams &sp, int64_t g)
{
    SiteEval e;
    uint32_t n[4];
    for (int k = 0; k < 4; k++) n[k] = pl[(uint64_t)k * sp.G + g];
    const int cov = (int)(n[0] + n[1] + n[2] + n[3]);
    e.cls = cov > 0 ? 0 : (pl[(uint64_t)pMulti * sp.G + g] == 0 ? 1 : 2);
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

struct Column { uint32_t v[nPlanes]; int32_t depth; uint32_t ref; }; // 48 bytes
struct RangeQ { int64_t beg, end; int32_t mode, pad; }; // [beg, end] inclusive; mode 0: coverage sum, 1: minimum over covered positions

// the dense profile as the caller sees it
struct DenseProfile {
    virtual ~DenseProfile() {}
    virtual int64_t genome_size() const = 0;
    // block depth, then the per-position scan: SNV / monomorphic records and run boundaries in (position, type) order
    virtual int scan(const ScanParams &sp, std::vector<SiteRec> &sites, double &ms_depth, double &ms_scan) = 0;
    virtual int gather(const std::vector<int64_t> &pos, std::vector<Column> &out) = 0;       // columns (+ block depth, reference base) of listed positions
    virtual int ranges(const std::vector<RangeQ> &q, std::vector<unsigned long long> &out) = 0; // coverage sum / minimum over listed ranges
};

// ---- host side --------------------------------------------------------------------------------------
struct Variant { // Variant_t, structure.h:185-195; ALT strings longer than 5 are never written (:451, :460), so 7 characters are kept
    int64_t gPos = 0;
    uint16_t DP = 0, AD_ref = 0, AD_alt = 0;
    uint8_t geno = 0, qscore = 0, type = 0, alt_len = 0;
    char alt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void set_alt(const char *p, size_t n) { alt_len = (uint8_t)std::min<size_t>(n, 255); memset(alt, 0, sizeof alt); memcpy(alt, p, std::min<size_t>(n, 7)); }
};
static inline bool by_pos(const Variant &a, const Variant &b) { return a.gPos == b.gPos ? a.type < b.type : a.gPos < b.gPos; } // CompByVarPos :50-54

// InsertSeqMap / DeleteSeqMap (AlignmentProfile.cpp:7) as one array sorted by (position, string) —
// the iteration order of the reference's map of maps — with 16-bit counts that wrap like its uint16_t
struct Tally { int64_t pos; uint16_t count; std::string seq; };
typedef std::vector<Tally> IndelMap;
struct Clip { int64_t pos; uint16_t count; };
struct Site { int64_t gPos, dist; };

// std::string's operator< on (seq, len)
static inline int seq_cmp(const char *a, size_t la, const char *b, size_t lb)
{
    const int c = memcmp(a, b, std::min(la, lb));
    return c ? c : (la < lb ? -1 : (la > lb ? 1 : 0));
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
    std::vector<Variant> vars_;
    std::vector<int64_t> push_pos_; // where a non-NOR record entered the reference's list (for gVCF runs)
    double ms_depth_ = 0, ms_scan_ = 0;

    void fold(const mcx_sparse_rec *recs, uint64_t n);
    int gather(const std::vector<int64_t> &pos, std::vector<Column> &out) { return prof_.gather(pos, out); }
    int ranges(const std::vector<RangeQ> &q, std::vector<unsigned long long> &out) { return prof_.ranges(q, out); }
    int scan(std::vector<SiteRec> &sites);
    int indels();
    void runs(const std::vector<SiteRec> &sites);
    int normal_runs(const std::vector<SiteRec> &sites);
    void drop_consecutive_nor();
    int discordant(const std::vector<int64_t> &cands, const std::vector<Site> &sites, int type);
    int breakpoints();
    int write(const char *path, mcx_vcf_stats *st);
    static int area_freq(int64_t g, const IndelMap &m, const Tally *&best);
    bool nearby(int i, int dist) const;
    bool bad_haplotype(int i, int dist) const;
};

// one record per event -> the reference's maps (AlignmentProfile.cpp:6-7) and site lists (ReadMapping.cpp:19)
inline void Caller::fold(const mcx_sparse_rec *recs, uint64_t n)
{
    std::vector<Tally> ev[2];
    std::vector<int64_t> clip;
    std::vector<mcx_sparse_rec> sites; // 'V' / 'T' as given, plus the ones the 'E' events of all shards resolve to
    bool any_event = false;
    for (uint64_t i = 0; i < n; i++) {
        const mcx_sparse_rec &r = recs[i];
        switch (r.type) {
        case 'I': case 'D': {
            Tally t; t.pos = r.pos; t.count = 1;
            t.seq.assign(r.seq, std::min<size_t>(r.len, sizeof r.seq));
            for (uint64_t j = i + 1; j < n && recs[j].type == 'C' && t.seq.size() < r.len; j++) // a long string continues in the records behind
                t.seq.append(recs[j].seq, std::min<size_t>(recs[j].len, sizeof recs[j].seq));
            ev[r.type == 'D'].push_back(std::move(t));
            break;
        }
        case 'B': clip.push_back(r.pos); break;
        case 'V': case 'T': sites.push_back(r); break;
        case 'E': any_event = true; break;
        }
    }
    if (any_event) mcx_disc_resolve(recs, n, G_, sites);
    for (const mcx_sparse_rec &r : sites) { Site s; s.gPos = r.pos; memcpy(&s.dist, r.seq, 8); (r.type == 'V' ? inv_ : tnl_).push_back(s); }
    for (int k = 0; k < 2; k++) {
        std::sort(ev[k].begin(), ev[k].end(), [](const Tally &a, const Tally &b) { return a.pos != b.pos ? a.pos < b.pos : a.seq < b.seq; });
        IndelMap &m = k == 0 ? ins_ : del_;
        for (Tally &t : ev[k]) {
            if (!m.empty() && m.back().pos == t.pos && m.back().seq == t.seq) { m.back().count++; continue; }
            m.push_back(std::move(t));
        }
    }
    std::sort(clip.begin(), clip.end());
    for (int64_t p : clip) {
        if (!brk_.empty() && brk_.back().pos == p) brk_.back().count++;
        else { Clip c; c.pos = p; c.count = 1; brk_.push_back(c); }
    }
    auto lt = [](const Site &a, const Site &b) { return a.gPos != b.gPos ? a.gPos < b.gPos : a.dist < b.dist; };
    std::sort(inv_.begin(), inv_.end(), lt); // CompByDiscordPos orders by position only; ties do not matter below
    std::sort(tnl_.begin(), tnl_.end(), lt);
}

inline int Caller::scan(std::vector<SiteRec> &sites)
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
        if (max_freq < a->count || (max_freq == a->count && a->seq.size() > (best ? best->seq.size() : 0))) {
            if (max_freq < a->count) max_freq = a->count;
            best = &*a; max_pos = a->pos;
        }
    }
    return g == max_pos ? freq : 0;
}

// indel calls (:570-589): only a position that has a tally of its own can be `max_pos`
inline int Caller::indels()
{
    std::vector<int64_t> keys;
    for (const Tally &t : ins_) if (t.pos >= 0 && t.pos < G_) keys.push_back(t.pos);
    for (const Tally &t : del_) if (t.pos >= 0 && t.pos < G_) keys.push_back(t.pos);
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    std::vector<Column> col;
    int rc = gather(keys, col);
    if (rc) return rc;
    const Tally *best = nullptr;
    for (size_t i = 0; i < keys.size(); i++) {
        const int64_t g = keys[i];
        const Column &c = col[i];
        const int cov = (int)(c.v[pA] + c.v[pC] + c.v[pG] + c.v[pT]);
        const int thr = cov_threshold(c.depth, o_.min_allele_depth, o_.somatic);
        const int thr_of[2] = {std::max((int)(thr * 0.25), o_.min_allele_depth), std::max((int)(thr * 0.35), o_.min_allele_depth)};
        for (int k = 0; k < 2; k++) {
            const int freq = area_freq(g, k == 0 ? ins_ : del_, best);
            if (freq < thr_of[k]) continue;
            Variant v;
            v.gPos = g; v.type = k == 0 ? vINS : vDEL;
            if (best) v.set_alt(best->seq.data(), best->seq.size());
            v.AD_alt = (uint16_t)freq; v.DP = std::max((uint16_t)c.depth, v.AD_alt); v.AD_ref = v.DP - v.AD_alt;
            v.geno = genotype_of(o_.ploidy, v.DP, v.AD_alt, 1);
            v.qscore = cov == 0 ? 0 : (uint8_t)(int)(100.0 * v.AD_alt / cov); // (the reference's x/0 also ends as 0 on x86-64)
            vars_.push_back(v);
            push_pos_.push_back(g);
        }
    }
    return 0;
}

// SNV / monomorphic records, and uncovered (UMR) / duplicated (CNV) runs from their boundaries
// (:625-644).  The k-th start of a kind pairs with its k-th end; a run that reaches the genome end
// has no end and — as in the reference — is never reported.  Lengths are kept in 16 bits like Variant_t::DP.
inline void Caller::runs(const std::vector<SiteRec> &sites)
{
    std::vector<int64_t> indel_pos(push_pos_); // (sorted by construction)
    int64_t open[2] = {-1, -1};
    for (const SiteRec &r : sites) {
        if (r.type == vSUB || r.type == vMON) {
            if (r.type == vMON && std::binary_search(indel_pos.begin(), indel_pos.end(), r.pos)) continue; // bNormal is false where an indel was called
            Variant v;
            v.gPos = r.pos; v.type = r.type; v.DP = r.DP; v.AD_ref = r.AD_ref; v.AD_alt = r.AD_alt; v.geno = r.geno; v.qscore = r.qscore;
            if (r.type == vSUB) {
                char a[3] = {"ACGT"[r.alt & 3], ',', "ACGT"[(r.alt >> 4) & 3]};
                v.set_alt(a, (r.alt >> 4) == 0xF ? 1 : 3);
                push_pos_.push_back(r.pos);
            }
            vars_.push_back(v);
        } else if (r.type == eGapStart) open[0] = r.pos;
        else if (r.type == eDupStart) open[1] = r.pos;
        else if (r.type == eGapEnd || r.type == eDupEnd) {
            const int k = r.type == eDupEnd;
            const int64_t len = r.pos - open[k];
            if (k == 0 ? len >= o_.min_gap : len > o_.min_cnv) {
                Variant v;
                v.type = k ? vCNV : vUMR; v.gPos = open[k]; v.DP = (uint16_t)len;
                vars_.push_back(v);
                push_pos_.push_back(r.pos); // the record is appended when the scan reaches the first position after the run
            }
        }
    }
    std::sort(push_pos_.begin(), push_pos_.end());
}

// gVCF blocks (:645-656).  A covered position without a call extends the last record if that is a
// block, otherwise opens one; MIN_DP is the smallest depth of the block.  The scan delivered the
// maximal stretches of such positions; an indel call removes its position from a stretch, and a
// block continues from one stretch into the next when no other record was appended in between.
inline int Caller::normal_runs(const std::vector<SiteRec> &sites)
{
    std::vector<int64_t> indel_pos;
    for (const Variant &v : vars_) if (v.type == vINS || v.type == vDEL) indel_pos.push_back(v.gPos);
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
    std::vector<Column> first;
    if ((rc = gather(block_start, first))) return rc;
    std::vector<unsigned long long> block_min(block_start.size(), ~0ull);
    for (size_t i = 0; i < q.size(); i++) block_min[owner[i]] = std::min(block_min[owner[i]], mn[i]);
    for (size_t b = 0; b < block_start.size(); b++) {
        Variant v;
        v.gPos = block_start[b]; v.type = vNOR;
        v.DP = (uint16_t)(first[b].v[pA] + first[b].v[pC] + first[b].v[pG] + first[b].v[pT]);
        v.AD_alt = (uint16_t)std::min<unsigned long long>(block_min[b], v.DP);
        vars_.push_back(v);
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
    std::vector<Column> col;
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
    std::vector<Variant> found;
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
        Variant v;
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
    std::vector<Column> col;
    int rc = gather(pos, col);
    if (rc) return rc;
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
    const int n_thr = (int)std::max(1u, std::min(16u, std::min(std::thread::hardware_concurrency(), (unsigned)(n / 4096 + 1))));
    std::vector<std::string> text(n_thr);
    std::vector<mcx_vcf_stats> part(n_thr);
    auto chr_of = [&](int64_t g) { return (int)(std::upper_bound(h.chr_fwd.begin(), h.chr_fwd.end(), g) - h.chr_fwd.begin()) - 1; }; // DetermineCoordinate, tools.cpp:132-164 (forward strand)
    auto work = [&](int t) {
        mcx_vcf_stats s;
        memset(&s, 0, sizeof s);
        std::string &out = text[t];
        out.reserve((size_t)(n / n_thr + 1) * 96);
        std::vector<char> line(1024);
        auto put = [&](const char *fmt, auto... a) { // one formatted piece, appended
            int k = snprintf(line.data(), line.size(), fmt, a...);
            if (k >= (int)line.size()) { line.resize((size_t)k + 1); k = snprintf(line.data(), line.size(), fmt, a...); }
            out.append(line.data(), (size_t)k);
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
            case vSUB:
                s.n_snv++; s.n_records++;
                put("%s\t%d\t.\t%c\t%s\t%d\t%s\tRC=%d;NTFREQ=%d,%d,%d,%d;TYPE=snv\tGT:GQ:DP:AD:AF:F1R2:F2R1\t%s:%d:%d:%d,%d:%.2f:%d,%d:%d,%d\n", chr, p1, ref, v.alt,
                        v.qscore, flt.c_str(), rc_, (int)c.v[pA], (int)c.v[pC], (int)c.v[pG], (int)c.v[pT], GT[v.geno], v.qscore, v.DP, v.AD_ref, v.AD_alt, af, F1, R2, F2, R1);
                break;
            case vINS: case vDEL:
                if (v.alt_len > 5) break;
                (v.type == vINS ? s.n_ins : s.n_del)++; s.n_records++;
                if (v.type == vINS) put("%s\t%d\t.\t%c\t%c%s\t", chr, p1, ref, ref, v.alt);
                else put("%s\t%d\t.\t%c%s\t%c\t", chr, p1, ref, v.alt, ref);
                put("%d\t%s\tRC=%d;TYPE=%s\tGT:GQ:DP:AD:AF:F1R2:F2R1\t%s:%d:%d:%d,%d:%.2f:%d,%d:%d,%d\n", v.qscore, flt.c_str(), rc_, v.type == vINS ? "ins" : "del",
                        GT[v.geno], v.qscore, v.DP, v.AD_ref, v.AD_alt, af, F1, R2, F2, R1);
                break;
            case vTNL: case vINV:
                (v.type == vTNL ? s.n_tnl : s.n_inv)++; s.n_records++;
                put("%s\t%d\t.\t%c\t<%s>\t30\tBreakPoint\tTYPE=BP\tGT:GQ:DP:AD\t.:.:0:.\n", chr, p1, ref, v.type == vTNL ? "TNL" : "INV");
                break;
            case vCNV: case vUMR:
                if (v.DP < (v.type == vCNV ? o_.min_cnv : o_.min_gap)) break;
                s.n_records++;
                put("%s\t%d\t.\t%c\t<*>\t0\t%s\tEND=%d\tGT:GQ:DP:AD\t.:.:0:.\n", chr, p1, ref, v.type == vCNV ? "DUP" : "Gaps", p1 + v.DP - 1);
                break;
            case vNOR: {
                int64_t end = h.chr_fwd[ci] + h.chr_len[ci] - 1;
                if (i + 1 < n && vars_[i + 1].gPos < end) end = vars_[i + 1].gPos - 1;
                const int ce = chr_of(end);
                s.n_records++;
                put("%s\t%d\t.\t%c\t<*>\t0\tREF\tEND=%d;DP=%d;MIN_DP=%d\tGT:GQ:DP:AD\t.:.:0:.\n", chr, p1, ref, (int)(end - h.chr_fwd[ce] + 1), v.DP, v.AD_alt);
                break;
            }
            case vMON:
                s.n_records++;
                put("%s\t%d\t.\t%c\t.\t0\tREF\tDP=%d;RC=%d;NTFREQ=%d,%d,%d,%d\tGT:F1R2:F2R1\t%s:%d,%d:%d,%d\n", chr, p1, ref, v.DP, rc_, (int)c.v[pA], (int)c.v[pC],
                        (int)c.v[pG], (int)c.v[pT], GT[v.geno], F1, R2, F2, R1);
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
    mcx_vcf_stats s;
    memset(&s, 0, sizeof s);
    for (int t = 0; t < n_thr; t++) {
        const mcx_vcf_stats &q = part[t];
        s.n_records += q.n_records; s.n_snv += q.n_snv; s.n_ins += q.n_ins; s.n_del += q.n_del; s.n_tnl += q.n_tnl; s.n_inv += q.n_inv;
        if (!text[t].empty() && fwrite(text[t].data(), 1, text[t].size(), f) != text[t].size()) { fclose(f); return mcx_set_error(MCX_ERR_IO, std::string("cannot write ") + path); }
    }
    if (fclose(f) != 0) return mcx_set_error(MCX_ERR_IO, std::string("cannot write ") + path);
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
    std::vector<SiteRec> sites;
    int rc;
    if ((rc = scan(sites))) return rc;
    lap("scan+sort");
    if ((rc = indels())) return rc;
    lap("indels");
    runs(sites); lap("runs");
    if (o_.gvcf && (rc = normal_runs(sites))) return rc;
    std::stable_sort(vars_.begin(), vars_.end(), by_pos);
    if (o_.gvcf) drop_consecutive_nor();
    lap("order");
    if ((rc = breakpoints())) return rc;
    lap("breakpoints");
    rc = write(path, st); lap("write");
    return rc;
}

} // namespace mcx_vc
#endif
