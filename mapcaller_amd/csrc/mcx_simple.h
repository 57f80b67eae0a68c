// mapcaller_amd/csrc/mcx_simple.h — the straight-line case of the per-pair path, seeds to records, without a pair record.
//
// The general path (mcx_glue.h) carries every pair through clustering, fragment construction and the finish stage in a
// fixed-capacity record in HBM: three kernels, each touching ~57 scattered 16-byte pieces of an 8 KB record per pair.  Most
// pairs need none of that machinery: a few seeds per read that form ONE cluster, one candidate per read, the two paired
// within the estimated distance, every gap between seeds a stretch of equal length on read and genome with at most a
// mismatch or two (or a pure insertion / deletion), every gate passed.  For such a pair the reference's functions
//
//   IdentifySimplePairs' tail + SimplePairClustering     ReadMapping.cpp:141-152, :194-226
//   CheckPairedAlignmentDistance, MaskUnPairedAlnCan     ReadMapping.cpp:244-322
//   ProduceReadAlignment, ProcessNormalPair and gates    ReadAlignment.cpp:155-191, :306-430
//   GenCoordinatePair + counting                         ReadMapping.cpp:361-394, :479-531
//   flags, MAPQ, coordinates, CIGAR, TLEN                SamReport.cpp:26-316, :377-488; tools.cpp:132
//
// collapse to one pass over the seeds in read order.  simple_read() makes that pass for one read — in registers: at most
// kSimpleHits seeds — and simple_pair() joins two reads into records.  Whatever is not straight-line (several clusters, a
// seed out of order, a gap that needs the DP, a gate that fails, an unpaired pair, reads with N) makes them return false
// BEFORE anything is written, and the pair takes the general path from its untouched seeds.  The decisions are the general
// path's own, restated for the case where they all go one way; the parity tests run both.
#ifndef MCX_SIMPLE_H
#define MCX_SIMPLE_H
#include "mcx_glue.h"

namespace mcx {

constexpr int kSimpleHits = 4;                  // seeds per read the straight-line path takes (one 64-byte line of hits)
constexpr int kSimpleRuns = 2 * kSimpleHits;    // CIGAR operations such a read can have (seeds and gaps alternate; equal neighbours merge)

struct SimpleRead {
    int64_t pd0;        // PosDiff of the candidate: of its first seed in (PosDiff, rPos) order
    int64_t g_first;    // gPos of the first fragment in ALIGNMENT order (GenCoordinatePair; fwd: in read order, else the last one's)
    int64_t g_coord;    // the position GetAlnCoordinate reports: the first fragment (alignment order) that holds genome bases
    int32_t score;      // matched bases after the gates (AlnSummary score; NM = rlen - score)
    int32_t fwd;        // orientation
    int32_t n_cig;      // CIGAR operations, in alignment order at cig[0 .. n_cig)
};

// One read: true when it is straight-line; then `out` and cig[k * cig_stride] (k < out.n_cig) are filled.
// hits: the read's seeds as k_seed left them (text positions), n of them (1..kSimpleHits); codes: its 2-bit words (no N).
static inline MCX_HD bool simple_read(const IndexView &ix, const Params &pm, int rlen, const uint32_t *codes, const Hit *hits, int n,
                                      SimpleRead &out, uint32_t *cig, int cig_stride)
{
    if (n < 1 || n > kSimpleHits || !codes) return false;
    // ---- the seeds with PosDiff > 0 (IdentifySimplePairs' tail); the straight-line case needs all of them to stay
    int64_t g[kSimpleHits], pd[kSimpleHits];
    int r[kSimpleHits], len[kSimpleHits];
    MCX_UNROLL
    for (int i = 0; i < kSimpleHits; i++) {
        if (i < n) { const Hit h = hits[i]; g[i] = h.gPos; r[i] = h.rPos; len[i] = h.len; pd[i] = h.gPos - h.rPos; if (pd[i] <= 0) return false; }
        else { g[i] = 0; r[i] = 0x7fffffff; len[i] = 0; pd[i] = (int64_t)1 << 62; } // (sorts behind the real ones)
    }
    // ---- one cluster (SimplePairClustering): in (PosDiff, rPos) order neighbours lie within MaxPosDiff of each other and
    //      every seed starts before the chromosome end behind the first one
    {
        int64_t sp[kSimpleHits], sg0 = 0;
        MCX_UNROLL
        for (int i = 0; i < kSimpleHits; i++) sp[i] = pd[i];
        // the first seed of that order: smallest (PosDiff, rPos)
        {
            int64_t bp = sp[0]; int br = r[0]; sg0 = g[0];
            MCX_UNROLL
            for (int i = 1; i < kSimpleHits; i++) if (i < n && (pd[i] < bp || (pd[i] == bp && r[i] < br))) { bp = pd[i]; br = r[i]; sg0 = g[i]; }
            out.pd0 = bp;
        }
        // PosDiffs in ascending order (a 4-element network on the values alone)
        auto cx2 = [](int64_t &a, int64_t &b) { if (a > b) { const int64_t t = a; a = b; b = t; } };
        cx2(sp[0], sp[1]); cx2(sp[2], sp[3]); cx2(sp[0], sp[2]); cx2(sp[1], sp[3]); cx2(sp[1], sp[2]);
        MCX_UNROLL
        for (int i = 1; i < kSimpleHits; i++) if (i < n && sp[i] - sp[i - 1] > pm.max_pos_diff) return false;
        const int64_t g_end = boundary_of(ix, sg0);
        int score = 0;
        MCX_UNROLL
        for (int i = 0; i < kSimpleHits; i++) if (i < n) { if (g[i] > g_end) return false; score += len[i]; }
        if (score <= (rlen >> 2)) return false;       // no candidate at all
        if (score >= rlen && n > 1) return false;     // the tandem-repeat branch picks a run of equal PosDiff
    }
    // ---- the seeds in read order (ProduceReadAlignment sorts by (rPos, gPos), ReadAlignment.cpp:317)
    {
        auto cx4 = [&](int a, int b) {
            if (r[a] > r[b]) {
                const int64_t tg = g[a]; g[a] = g[b]; g[b] = tg;
                const int tr = r[a]; r[a] = r[b]; r[b] = tr;
                const int tl = len[a]; len[a] = len[b]; len[b] = tl;
            }
        };
        cx4(0, 1); cx4(2, 3); cx4(0, 2); cx4(1, 3); cx4(1, 2);
    }
    // ---- one pass over seeds and gaps: fragments (IdentifyNormalPairs), their kinds (ProcessNormalPair), the gates and scores of
    //      ProduceReadAlignment's tail, the CIGAR runs — all in read order
    const int max_mm = (int)(rlen * pm.max_mm_rate);
    const int min_score = (int)(rlen * (1 - pm.max_mm_rate));
    int score = 0, mism = 0;
    int n_run = 0, run_len = 0, run_op = -1;     // the run being built; closed runs at cig[.]
    uint32_t runs[kSimpleRuns];
    MCX_UNROLL
    for (int k = 0; k < kSimpleRuns; k++) runs[k] = 0;
    auto add = [&](int l, int op) {
        if (op != run_op) {
            if (run_len > 0) {
                const uint32_t w = ((uint32_t)run_len << 4) | (uint32_t)run_op;
                MCX_UNROLL
                for (int k = 0; k < kSimpleRuns; k++) if (k == n_run) runs[k] = w;
                n_run++;
            }
            run_op = op; run_len = 0;
        }
        run_len += l;
    };
    // a gap fragment of equal lengths: mismatches, the DP decision (ReadAlignment.cpp:184), the gate of its place
    // (head / tail: dropped when it is long enough and bad, :343-372; in between: the whole candidate dies, :373-381)
    auto plain_gap = [&](int rp, int64_t gp, int l) -> bool {
        Frag x; x.rPos = rp; x.gPos = gp; x.rLen = l; x.gLen = l; x.ops_off = 0; x.ops_len = 0; x.kind = kPlain; x.meta = 0;
        ReadRef rd; rd.ascii = nullptr; rd.rlen = rlen; rd.flipped = 0; rd.codes = codes;
        const int mm = frag_mismatches(ix, x, rd);
        if (mm > 1 && mm >= (int)(l * 0.2)) return false;                  // a DP problem
        if (l >= kMinAlnBlockSize && mm >= 3 && mm >= (int)(l * 0.3)) return false; // (one kind of column: switches = 1) the quality gate would fire
        score += l - mm; mism += mm;
        add(l, 0);
        return true;
    };
    int pr = 0;               // read / genome position behind the previous seed
    int64_t pg = 0, g_head = 0, g_tail = 0; // gPos of the first fragment; end (exclusive) of the last
    int prev_r = -1;
    int64_t prev_g = -1;
    MCX_UNROLL
    for (int i = 0; i < kSimpleHits; i++) {
        if (i >= n) break;
        const int r0 = r[i];
        const int64_t g0 = g[i];
        if (i == 0) {
            g_head = g0 - r0;
            if (r0 > 0 && !plain_gap(0, g0 - r0, r0)) return false;
        } else {
            const int rg = r0 - pr;
            const int64_t gg = g0 - pg;
            if (r0 <= prev_r || g0 <= prev_g || rg < 0 || gg < 0) return false; // not in order / overlapping: the general path sorts and trims
            if (rg > 0 && gg > 0) {
                if ((int64_t)rg != gg) return false;                            // a DP problem
                if (!plain_gap(pr, pg, rg)) return false;
            } else if (rg > 0) add(rg, 1);                                       // read bases against '-'
            else if (gg > 0) { if (gg >= 4096) return false; add((int)gg, 2); } // '-' against genome bases (Frag::gLen is 12 bits)
        }
        score += len[i];
        add(len[i], 0);
        prev_r = r0; prev_g = g0; pr = r0 + len[i]; pg = g0 + len[i];
    }
    if (pr < rlen) { if (!plain_gap(pr, pg, rlen - pr)) return false; pg += rlen - pr; }
    g_tail = pg;
    // CheckAlignmentValidity (tools.cpp:119-130): inside [0, 2G) and on one chromosome
    if (g_head < 0 || g_tail > ix.G2) return false;
    {
        const int e1 = end_slot(ix, g_head), e2 = end_slot(ix, g_tail - 1);
        if (e1 < 0 || e2 < 0 || ix.end_pos[e1] != ix.end_pos[e2]) return false;
    }
    if (score == 0 || (score < min_score && mism > max_mm)) return false;          // the candidate would be dropped (:392-396)
    // close the last run; the operations in alignment order (a reverse-strand candidate's fragments are read backwards, :412-416)
    {
        const uint32_t w = ((uint32_t)run_len << 4) | (uint32_t)run_op;
        MCX_UNROLL
        for (int k = 0; k < kSimpleRuns; k++) if (k == n_run) runs[k] = w;
        n_run++;
    }
    const int fwd = g_head < ix.G ? 1 : 0;
    MCX_UNROLL
    for (int k = 0; k < kSimpleRuns; k++) if (k < n_run) cig[(fwd ? k : n_run - 1 - k) * cig_stride] = runs[k];
    out.score = score; out.fwd = fwd; out.n_cig = n_run;
    // first fragment in alignment order: forward = the head (gap or seed) at g_head; reverse = the tail, which ends at g_tail.
    // GetAlnCoordinate takes gPos (forward) or gPos + gLen - 1 (reverse) of it; GenCoordinatePair its gPos.
    if (fwd) { out.g_first = g_head; out.g_coord = g_head; }
    else {
        // the last fragment in read order: the tail gap if the last seed ends before the read does, else the last seed
        int64_t last_g = 0;
        MCX_UNROLL
        for (int i = 0; i < kSimpleHits; i++) if (i == n - 1) last_g = (r[i] + len[i] < rlen) ? g[i] + len[i] : g[i];
        out.g_first = last_g; out.g_coord = g_tail - 1;
    }
    return true;
}

// CheckPairedAlignmentDistance with one candidate each: read 2's PosDiff not before read 1's and closer than the estimate
// (anything else: unpaired, mate rescue decides)
static inline MCX_HD bool simple_pair_ok(const SimpleRead &a, const SimpleRead &b, int est)
{
    return b.pd0 >= a.pd0 && b.pd0 - a.pd0 < (int64_t)est;
}

// The pair (or the single read: paired == 0, b unused).  est: EstiDistance.  Fills the records and the pair's outcome; returns
// false — nothing written — when the pair is not straight-line (not paired within the estimate: mate rescue decides).
// cig_off: where the reads' CIGAR words lie in the batch's pool (for the records).
static inline MCX_HD bool simple_pair(const Ctx &cx, int paired, const SimpleRead &a, const SimpleRead &b, int rlen_a, int rlen_b, int est,
                                      AlnRec *rec2, const uint32_t cig_off[2], PairOut &po)
{
    const IndexView &ix = cx.ix;
    po.flags = 0; po.est = est; po.est_lo = 0; po.est_hi = 0x7fffffff; po.pair_dist = 0; po.pair_ok = 0; po.mapped = (int16_t)(paired ? 2 : 1); po.pad[0] = po.pad[1] = 0;
    int64_t d = 0;
    if (paired) {
        if (!simple_pair_ok(a, b, est)) return false;
        d = b.pd0 - a.pd0;
        po.est_lo = (int)(d + 1); // the pairing holds for every estimate above the distance (pair_by_distance's interval)
        // GenCoordinatePair + the counting of ReadMapping.cpp:479-531
        const int64_t g1 = a.g_first, g2 = b.g_first;
        const int64_t dist = g2 > g1 ? g2 - g1 : g1 - g2;
        const bool inv = (g1 < ix.G && g2 >= ix.G) || (g1 >= ix.G && g2 < ix.G);
        if (dist != 0 && !inv && dist <= kMinTranslocationSize) { po.pair_ok = 1; po.pair_dist = (int)dist; }
    }
    const SimpleRead *rd2[2] = {&a, &b};
    const int rl[2] = {rlen_a, rlen_b};
    Coord k[2];
    MCX_UNROLL
    for (int s = 0; s < 2; s++) { if (s == 1 && !paired) break; k[s] = to_coord(ix, rd2[s]->g_coord); }
    MCX_UNROLL
    for (int s = 0; s < 2; s++) {
        if (s == 1 && !paired) break;
        const SimpleRead &me = *rd2[s];
        AlnRec o;
        o.pos = k[s].pos; o.chr = k[s].chr; o.mapq = 60; o.fwd = me.fwd; // (EvaluateMAPQ: no second-best score -> 60)
        o.nm = rl[s] - me.score; o.as = me.score; o.xs = 0; o.n_cigar = me.n_cig; o.pad[0] = (int32_t)cig_off[s]; o.pad[1] = 0;
        if (paired) {
            // SetPairedAlignmentFlag: a proper pair, the mate's candidate alive
            int fl = s == 0 ? 0x41 : 0x81;
            if (s == 0) fl |= me.fwd ? 0x20 : 0x10; else fl |= me.fwd ? 0x10 : 0x20;
            o.flag = fl | 0x2;
            // TLEN is defined from read 1's side and negated for read 2 (SamReport.cpp:428, :475)
            const int dist = (int)(k[1].pos - k[0].pos + (a.fwd ? rl[1] : 0 - rl[0]));
            o.tlen = s == 0 ? dist : 0 - dist;
            o.mate_pos = k[1 - s].pos; o.has_mate = 1;
        } else { o.flag = me.fwd ? 0 : 0x10; o.tlen = 0; o.mate_pos = 0; o.has_mate = 0; }
        rec2[s] = o;
    }
    return true;
}

} // namespace mcx
#endif
