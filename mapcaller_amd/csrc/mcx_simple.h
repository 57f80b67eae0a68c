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
#include "mcx_dp_lane.h"

namespace mcx {

// (the test harness counts which exit a read takes: MCX_SIMPLE_FAIL(exit number))
#ifndef MCX_SIMPLE_FAIL
#define MCX_SIMPLE_FAIL(code) return 0
#endif
#ifndef MCX_SIMPLE_NOTE
#define MCX_SIMPLE_NOTE(k, l) ((void)0)
#endif

constexpr int kSimpleHits = 4;                  // seeds per read the straight-line path takes (one 64-byte line of hits)
constexpr int kSimpleRuns = 12;                 // CIGAR operations the path keeps per read (seeds and gaps alternate, equal neighbours merge; a read that needs more leaves)
constexpr int kSimpleDp = 16;                   // a gap of up to kSimpleDp x kSimpleDp bases — between two seeds or at a read end — is a DP problem the path takes on itself
constexpr int kSimpleJobs = 4;                  // such problems per pair

// The path meets its DP problems in three passes, so that the problems are solved one per lane by lanes that all have one
// (a lane that aligned its own gap would hold up the 63 beside it that have none):
//   collect  k_simple       a gap that ProcessNormalPair hands to the gapped extension (ReadAlignment.cpp:184-187) is written down as a
//                           SimpleJob and the pass goes on — everything that does not hang on the alignment is still decided here;
//   solve    k_simple_dp    one problem per lane (mcx_dp_lane.h, one strip of 16 columns): the column string and its counts;
//   replay   k_simple_rest  the same pass over the pairs that wrote problems down, taking the results in the order they were met.
struct SimpleJob { uint32_t read; uint16_t rp; uint8_t rl, gl; int64_t gp; };   // read: pair * 2 + mate
// what the walk of such a problem leaves: the column string (2 bits a column, the LAST column in the low bits) and the counts the
// gates and scores read (frag_columns: 'M' columns, mismatches among them, runs)
struct SimpleCols {
    uint64_t w = 0; int len = 0, n = 0, mis = 0, switches = 0, cur = -1;
    MCX_HD bool wants_bases() const { return true; }
    MCX_HD void col(int kind, int differ) { w |= (uint64_t)kind << (2 * len); len++; if (kind == 0) { n++; mis += differ; } if (kind != cur) { cur = kind; switches++; } }
};
struct SimpleRes { uint64_t w; uint8_t len, n, mis, switches; uint8_t pad[4]; };
enum : int { kDpNone = 0, kDpCollect = 1, kDpReplay = 2 };
// a problem as the collecting pass leaves it (three words) -> its descriptor; first_read: the pair's first read
static inline MCX_HD SimpleJob simple_job_unpack(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t first_read, int nr)
{
    SimpleJob j;
    j.rp = (uint16_t)(w0 & 0xFFFFu); j.rl = (uint8_t)((w0 >> 16) & 31u); j.gl = (uint8_t)((w0 >> 21) & 31u);
    j.read = nr == 2 ? (first_read & ~1u) | ((w0 >> 26) & 1u) : first_read;
    j.gp = (int64_t)((uint64_t)w1 | ((uint64_t)w2 << 32));
    return j;
}
struct SimpleDpIo {
    int mode;               // kDpNone: a DP gap makes the read leave
    uint32_t *jobs;         // collect: the pair's problems as they are met, three words each (simple_job_pack), word w of problem k at jobs[(3 * k + w) * job_stride]
    int job_stride;
    const SimpleRes *res;   // replay: their results
    int n;                  // problems met so far in the pair
    uint32_t read;          // pair * 2 + mate of the read at hand
};

// the lane's words for such a problem: the query (codes, N flags) and the traceback words of one strip of 16 columns
template <bool NW> static inline MCX_HD LaneLayout simple_dp_layout()
{
    LaneLayout l; l.off_q = 0; l.off_edge = 0; l.off_dir = 2; l.rows = kSimpleDp; l.words = 2 + kSimpleDp * LaneDir<16, NW>::words; return l;
}

// solve: one problem by its lane (codes: the 2-bit words of the job's read; mem: simple_dp_layout().words words)
template <bool NW>
static inline MCX_HD SimpleRes simple_dp_job(const IndexView &ix, const SimpleJob &j, const uint32_t *codes, const LaneMem &mem)
{
    const bool rev = j.gp >= ix.G;
    const LaneLayout l = simple_dp_layout<NW>();
    const int rl_ = j.rl, gl_ = j.gl;
    mem.put(l.off_q, lane_query16(codes, j.rp, rl_, rev, 0));
    mem.put(l.off_q + 1, 0u);
    auto tgt16 = [&](int b0) -> uint32_t { return lane_target16(ix, j.gp, gl_, rev, b0); };
    SimpleCols sc;
    if (NW) { (void)lane_sweep_nw<16>(mem, l, rl_, gl_, tgt16); lane_walk_nw<16>(mem, l, rl_, gl_, tgt16, sc); }
    else { lane_sweep_ksw2<16>(mem, l, rl_, gl_, tgt16); lane_walk_ksw2<16>(mem, l, rl_, gl_, tgt16, sc); }
    SimpleRes r; r.w = sc.w; r.len = (uint8_t)sc.len; r.n = (uint8_t)sc.n; r.mis = (uint8_t)sc.mis; r.switches = (uint8_t)sc.switches;
    r.pad[0] = r.pad[1] = r.pad[2] = r.pad[3] = 0;
    return r;
}

// What the -vcf bookkeeping reads of a straight-line read: the record write_detail (mcx_glue.h) leaves for a read with exactly one surviving
// candidate — its fragments in read order as extend_read leaves them (a trimmed or dropped end included), the column strings of its DP
// fragments ('M' 'I' 'D', in the order the DP kernels leave them: against the read on the reverse strand).  simple_read hands them to a
// sink as it meets them; a read that leaves the path half way has written fragments nobody counts (the general path writes the record).
// The record keeps a candidate's fragments in ALIGNMENT order — a reverse-strand candidate's are read backwards (frag_index) —, the pass
// meets them in read order: the sink fills the record's first kSimpleFrags places from the front (forward) or from the back, and the record's
// header says where the first fragment lies (DetailHdr::frag0).  The strand is known with the first seed: the validity check at the
// end of the pass keeps an alignment on one chromosome, hence on one strand.
constexpr int kSimpleFrags = 2 * kSimpleHits + 1;
struct SimpleNoDetail {
    static constexpr bool on = false;
    static constexpr int nf = 0, no = 0, fwd = 1;
    MCX_HD void frag(int, int, int64_t, int, int, int, int, int) {}
    MCX_HD int cols(uint64_t, int, int, int) { return 0; }
    MCX_HD void begin(bool) {}
    MCX_HD int frag0() const { return 0; }
};
struct SimpleDetail {
    static constexpr bool on = true;
    Frag *frags; uint8_t *ops; int nf, no, fwd;
    MCX_HD void begin(bool forward) { nf = 0; no = 0; fwd = forward ? 1 : 0; }
    MCX_HD int frag0() const { return fwd ? 0 : kSimpleFrags - nf; }
    MCX_HD void frag(int kind, int rp, int64_t gp, int rl, int gl, int ops_off, int ops_len, int meta)
    {
        Frag x; x.gPos = gp; x.rPos = rp; x.gLen = gl; x.rLen = rl; x.ops_off = ops_off; x.ops_len = ops_len; x.kind = (uint64_t)kind; x.meta = (uint64_t)meta;
        frags[fwd ? nf : kSimpleFrags - 1 - nf] = x;
        nf++;
    }
    // the columns [a, b) of a SimpleRes string (column j at bits 2 (len - 1 - j)): where they went
    MCX_HD int cols(uint64_t w, int len, int a, int b)
    {
        const int at = no;
        for (int j = a; j < b; j++) ops[no++] = (uint8_t)"MID"[(w >> (2 * (len - 1 - j))) & 3u];
        return at;
    }
};

struct SimpleRead {
    int64_t pd0;        // PosDiff of the candidate: of its first seed in (PosDiff, rPos) order
    int64_t g_first;    // gPos of the first fragment in ALIGNMENT order (GenCoordinatePair; fwd: in read order, else the last one's)
    int64_t g_coord;    // the position GetAlnCoordinate reports: the first fragment (alignment order) that holds genome bases
    int32_t score;      // matched bases after the gates (AlnSummary score; NM = rlen - score)
    int32_t fwd;        // orientation
    int32_t n_cig;      // CIGAR operations, in alignment order at cig[0 .. n_cig)
    int32_t n_frags, n_ops, frag0; // (with a detail sink) what it holds, and where its first fragment lies
};

// One read.  kSimpleYes: it is straight-line; `out` and cig[k * cig_stride] (k < out.n_cig) are filled.  kSimpleLater (collect only): it
// is, but for DP problems now written down (out.pd0 is set: the pairing test does not wait).  kSimpleNo: the general path.
// hits: the read's seeds as k_seed left them (text positions), n of them (1..kSimpleHits); codes: its 2-bit words (no N).
enum : int { kSimpleNo = 0, kSimpleYes = 1, kSimpleLater = 2 };
// (CigT: 32-bit words, or 16-bit ones — a run is at most 4095 long: reads of up to 1000 bases, deletions below 4096)
template <bool NW, class CigT, class Det>
static inline MCX_HD int simple_read(const IndexView &ix, const Params &pm, int rlen, const uint32_t *codes, const Hit *hits, int n,
                                     SimpleRead &out, CigT *cig, int cig_stride, SimpleDpIo &io, Det &det)
{
    if (n < 1 || n > kSimpleHits || !codes) MCX_SIMPLE_FAIL(1);
    // ---- the seeds with PosDiff > 0 (IdentifySimplePairs' tail); the straight-line case needs all of them to stay
    int64_t g[kSimpleHits], pd[kSimpleHits];
    int r[kSimpleHits], len[kSimpleHits];
    MCX_UNROLL
    for (int i = 0; i < kSimpleHits; i++) {
        if (i < n) { const Hit h = hits[i]; g[i] = h.gPos; r[i] = h.rPos; len[i] = h.len; pd[i] = h.gPos - h.rPos; if (pd[i] <= 0) MCX_SIMPLE_FAIL(2); }
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
        for (int i = 1; i < kSimpleHits; i++) if (i < n && sp[i] - sp[i - 1] > pm.max_pos_diff) MCX_SIMPLE_FAIL(3);
        const int64_t g_end = boundary_of(ix, sg0);
        int score = 0;
        MCX_UNROLL
        for (int i = 0; i < kSimpleHits; i++) if (i < n) { if (g[i] > g_end) MCX_SIMPLE_FAIL(4); score += len[i]; }
        if (score <= (rlen >> 2)) MCX_SIMPLE_FAIL(5);       // no candidate at all
        if (score >= rlen && n > 1) MCX_SIMPLE_FAIL(6);     // the tandem-repeat branch picks a run of equal PosDiff
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
                n_run++; // (past kSimpleRuns the read leaves: checked where the last run is closed)
            }
            run_op = op; run_len = 0;
        }
        run_len += l;
    };
    // A gap that ProcessNormalPair hands to the gapped extension (ReadAlignment.cpp:184-187), small enough for the path: written down
    // (collect) or its result taken (replay) — the product's own sweep and walk (mcx_dp_lane.h), both strings reversed on the reverse strand
    // as the DP kernels take them.  place 0: between two seeds — the middle fragment's gate (:373-381), its matches, its columns as CIGAR
    // runs in read order.  place 1 / 2: the read's first / last fragment — RemoveHeadingGaps / RemoveTailingGaps (:264-304) take the gap
    // columns off its outer end (the read bases among them end up soft-clipped, the genome bases move the fragment's start or end), then
    // the end's gate (:343-372: a long enough bad end is dropped), its matches, its columns.
    bool later = false, end_dropped = false;
    int64_t head_shift = 0, tail_shift = 0; // genome bases taken off the first fragment's start / the last one's end
    auto dp_gap = [&](int rp, int64_t gp, int rl_, int gl_, int place) -> bool {
        if (io.mode == kDpNone || rl_ > kSimpleDp || gl_ > kSimpleDp) { if (rl_ == gl_) MCX_SIMPLE_NOTE(2, rl_); MCX_SIMPLE_FAIL(18); }
        if (io.n >= kSimpleJobs) MCX_SIMPLE_FAIL(18);
        if (io.mode == kDpCollect) {
            uint32_t *w = io.jobs + (size_t)(3 * io.n++) * io.job_stride;
            w[0] = (uint32_t)rp | ((uint32_t)rl_ << 16) | ((uint32_t)gl_ << 21) | ((io.read & 1u) << 26);
            w[io.job_stride] = (uint32_t)gp; w[2 * io.job_stride] = (uint32_t)((uint64_t)gp >> 32);
            later = true;
            return true;
        }
        const SimpleRes sc = io.res[io.n++];
        const bool rev = gp >= ix.G;
        int a = 0, b = sc.len, sw = sc.switches, clip = 0; // the string's columns [a, b) stay
        int rs = 0, gs = 0;                                // read / genome bases among the gap columns that come off an outer end
        if (place) {
            const bool lead = (place == 1) != rev; // the read's outer end is the string's start (head, forward; tail, reverse) or its end
            int runs = 0, cur = 0;
            for (int j = 0; j < sc.len; j++) {
                const int k = (int)((sc.w >> (2 * (lead ? sc.len - 1 - j : j))) & 3u);
                if (k == 0) break;
                if (k != cur) { cur = k; runs++; }
                if (k == 1) rs++; else gs++;
                if (lead) a++; else b--;
            }
            if (a >= b) MCX_SIMPLE_FAIL(19); // nothing but gap columns
            sw -= runs; clip = rs;
            if (b - a >= kMinAlnBlockSize && (sw >= 4 || (sc.mis >= 3 && sc.mis >= (int)(sc.n * 0.3)))) {
                // the end is dropped (:349-356, :364-371): an empty fragment where the neighbouring seed begins / ends, its bases soft-clipped;
                // both ends dropped: the candidate dies
                if (end_dropped) MCX_SIMPLE_FAIL(19);
                end_dropped = true;
                if (place == 1) head_shift = gl_; else tail_shift = gl_;
                add(rl_, 4);
                // (extend_read: the emptied end sits where its neighbour begins — the first seed — or ends — the last one)
                if (place == 1) det.frag(kEmpty, rp + rl_, gp + gl_, 0, 0, 0, 0, 0); else det.frag(kEmpty, rp, gp, 0, 0, 0, 0, 0);
                return true;
            }
            if (place == 1) head_shift = gs; else tail_shift = gs;
        } else if (rl_ >= kMinAlnBlockSize && gl_ >= kMinAlnBlockSize && (sw >= 4 || (sc.mis >= 3 && sc.mis >= (int)(sc.n * 0.3)))) MCX_SIMPLE_FAIL(19); // the candidate would die
        score += sc.n - sc.mis; mism += sc.mis;
        if (Det::on) { // strip_end_gaps: the head moves with what it lost, the tail only shrinks
            const int at = det.cols(sc.w, sc.len, a, b);
            det.frag(kDp, place == 1 ? rp + rs : rp, place == 1 ? gp + gs : gp, rl_ - rs, gl_ - gs, at, b - a, 0);
        }
        // column j of the string at bits 2 (len - 1 - j); on the reverse strand the string runs against the read
        if (place == 1 && clip > 0) add(clip, 4);
        for (int j = a; j < b; j++) add(1, (int)((sc.w >> (2 * (rev ? j - a + (sc.len - b) : sc.len - 1 - j))) & 3u));
        if (place == 2 && clip > 0) add(clip, 4);
        return true;
    };
    // a gap fragment of equal lengths: mismatches, the DP decision (ReadAlignment.cpp:184), the gate of its place
    // (head / tail: dropped when it is long enough and bad, :343-372; in between: the whole candidate dies, :373-381)
    auto plain_gap = [&](int rp, int64_t gp, int l, int place) -> bool {
        Frag x; x.rPos = rp; x.gPos = gp; x.rLen = l; x.gLen = l; x.ops_off = 0; x.ops_len = 0; x.kind = kPlain; x.meta = 0;
        ReadRef rd; rd.ascii = nullptr; rd.rlen = rlen; rd.flipped = 0; rd.codes = codes;
        const int mm = frag_mismatches(ix, x, rd);
        if (mm > 1 && mm >= (int)(l * 0.2)) { if (place) MCX_SIMPLE_NOTE(0, l); return dp_gap(rp, gp, l, l, place); } // a DP problem
        if (l >= kMinAlnBlockSize && mm >= 3 && mm >= (int)(l * 0.3)) MCX_SIMPLE_FAIL(8); // (one kind of column: switches = 1) the quality gate would fire
        score += l - mm; mism += mm;
        add(l, 0);
        det.frag(kPlain, rp, gp, l, l, 0, l, mm + 1);
        return true;
    };
    int pr = 0;               // read / genome position behind the previous seed
    int64_t pg = 0, g_head = 0, g_tail = 0; // gPos of the first fragment; end (exclusive) of the last
    int prev_r = -1;
    int64_t prev_g = -1;
    MCX_UNROLL
    for (int i = 0; i < kSimpleHits; i++) {
        if (i >= n) continue;
        const int r0 = r[i];
        const int64_t g0 = g[i];
        if (i == 0) {
            g_head = g0 - r0;
            det.begin(g_head < ix.G);
            if (r0 > 0 && !plain_gap(0, g0 - r0, r0, 1)) MCX_SIMPLE_FAIL(9);
        } else {
            const int rg = r0 - pr;
            const int64_t gg = g0 - pg;
            if (r0 <= prev_r || g0 <= prev_g || rg < 0 || gg < 0) MCX_SIMPLE_FAIL(10); // not in order / overlapping: the general path sorts and trims
            if (rg > 0 && gg > 0) {
                if ((int64_t)rg != gg) { if (gg > kSimpleDp || !dp_gap(pr, pg, rg, (int)gg, 0)) { MCX_SIMPLE_NOTE(1, (int)(gg > rg ? gg : rg)); MCX_SIMPLE_FAIL(11); } } // a DP problem
                else if (!plain_gap(pr, pg, rg, 0)) MCX_SIMPLE_FAIL(12);
            } else if (rg > 0) { add(rg, 1); det.frag(kIns, pr, pg, rg, 0, 0, rg, 0); }         // read bases against '-'
            else if (gg > 0) { if (gg >= 4096) MCX_SIMPLE_FAIL(13); add((int)gg, 2); det.frag(kDel, pr, pg, 0, (int)gg, 0, (int)gg, 0); } // '-' against genome bases (Frag::gLen is 12 bits)
        }
        score += len[i];
        add(len[i], 0);
        det.frag(kSimple, r0, g0, len[i], len[i], 0, 0, 0);
        prev_r = r0; prev_g = g0; pr = r0 + len[i]; pg = g0 + len[i];
    }
    if (pr < rlen) { if (!plain_gap(pr, pg, rlen - pr, 2)) MCX_SIMPLE_FAIL(14); pg += rlen - pr; }
    g_tail = pg;
    // CheckAlignmentValidity (tools.cpp:119-130): inside [0, 2G) and on one chromosome
    if (g_head < 0 || g_tail > ix.G2) MCX_SIMPLE_FAIL(15);
    {
        const int e1 = end_slot(ix, g_head), e2 = end_slot(ix, g_tail - 1);
        if (e1 < 0 || e2 < 0 || ix.end_pos[e1] != ix.end_pos[e2]) MCX_SIMPLE_FAIL(16);
    }
    if (later) return kSimpleLater; // (scores, runs and the ends' shifts hang on the alignments)
    if (score == 0 || (score < min_score && mism > max_mm)) MCX_SIMPLE_FAIL(17);          // the candidate would be dropped (:392-396)
    // close the last run; the operations in alignment order (a reverse-strand candidate's fragments are read backwards, :412-416)
    {
        const uint32_t w = ((uint32_t)run_len << 4) | (uint32_t)run_op;
        MCX_UNROLL
        for (int k = 0; k < kSimpleRuns; k++) if (k == n_run) runs[k] = w;
        n_run++;
    }
    if (n_run > kSimpleRuns) MCX_SIMPLE_FAIL(20); // more operations than the path keeps
    const int fwd = g_head + head_shift < ix.G ? 1 : 0;
    if (Det::on && fwd != det.fwd) MCX_SIMPLE_FAIL(21); // (cannot happen: one chromosome, one strand)
    MCX_UNROLL
    for (int k = 0; k < kSimpleRuns; k++) if (k < n_run) cig[(fwd ? k : n_run - 1 - k) * cig_stride] = (CigT)runs[k];
    out.score = score; out.fwd = fwd; out.n_cig = n_run;
    if (Det::on) { out.n_frags = det.nf; out.n_ops = det.no; out.frag0 = det.frag0(); }
    // first fragment in alignment order: forward = the head (gap or seed) at g_head, behind the genome bases its outer end lost; reverse =
    // the tail, which ends at g_tail less what its outer end lost.  GetAlnCoordinate takes gPos (forward) or gPos + gLen - 1 (reverse)
    // of it; GenCoordinatePair its gPos.
    if (fwd) { out.g_first = g_head + head_shift; out.g_coord = g_head + head_shift; }
    else {
        // the last fragment in read order: the tail gap if the last seed ends before the read does, else the last seed
        int64_t last_g = 0;
        MCX_UNROLL
        for (int i = 0; i < kSimpleHits; i++) if (i == n - 1) last_g = (r[i] + len[i] < rlen) ? g[i] + len[i] : g[i];
        out.g_first = last_g; out.g_coord = g_tail - tail_shift - 1;
    }
    return kSimpleYes;
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

// the header of read s's detail record (write_detail's for a read with one surviving candidate; in read 1's, pair_stats' account of which
// branch of ReadMapping.cpp:486-521 the pair takes when -vcf is on)
static inline MCX_HD DetailHdr simple_detail_hdr(const IndexView &ix, int paired, int s, const SimpleRead &a, const SimpleRead &b)
{
    const SimpleRead &me = s == 0 ? a : b;
    DetailHdr d;
    d.type = 1; d.n_frags = me.n_frags; d.fwd = me.fwd; d.n_ops = me.n_ops; d.disc_kind = 0; d.frag0 = me.frag0; d.disc_g1 = d.disc_g2 = d.disc_dist = 0;
    if (s == 0 && paired) {
        const int64_t g1 = a.g_first, g2 = b.g_first, dist = g2 > g1 ? g2 - g1 : g1 - g2, G = ix.G;
        d.disc_g1 = g1; d.disc_g2 = g2; d.disc_dist = dist;
        if (dist != 0) {
            if (g1 < G && g2 >= G) d.disc_kind = 1;
            else if (g1 >= G && g2 < G) d.disc_kind = 2;
            else if (dist > kMinTranslocationSize) d.disc_kind = (g1 < G && g2 < G) ? 3 : 4;
        }
    }
    return d;
}

} // namespace mcx
#endif
