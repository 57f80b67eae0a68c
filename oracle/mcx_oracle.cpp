// oracle/mcx_oracle.cpp — TEST INFRASTRUCTURE ONLY.
//
// A plain single-threaded-by-default CPU restatement of MapCaller's seed-and-extend read
// alignment path (reference v0.9.9.41 under /root/reference/src).  It exists to CHECK the HIP
// path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the
// product (mapcaller_amd/) never links, loads or calls anything in this directory.
//
// Parity status: PINNED.  The restatement is validated against the real reference compiled from
// its own sources into oracle/_ref (oracle/Makefile `make ref`): function level through
// oracle/ref_shim.cpp (BWT_Search, bwt_sa, nw_alignment, ksw2_alignment, ksw_extz2_sse) and end
// to end through `_ref/MapCaller -t 1 -sam` (tests/test_oracle_vs_ref.py, tests/golden/*).
//
// Each function names the reference lines it restates.  Data structures are our own
// (value-typed candidates holding index ranges, integer-scaled nw scores, a scalar per-cell
// version of the SSE ksw2 recurrence); results are required to be bit-identical.
#include "mcx_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;

// ---------------------------------------------------------------------------------------------
// constants of the path
// ---------------------------------------------------------------------------------------------
const int kMinSeed = 16;        // MinSeedLength, structure.h:23
const int kChunk = 200;         // ReadChunkSize, structure.h:24
const int kOccThr = 50;         // OCC_Thr, bwt_search.cpp:3
const int kKmer = 8;            // KmerSize, structure.h:20
const u32 kKmerMask = 0x3FFF;   // KmerPower, structure.h:21
const int kMinBlock = 5;        // MinAlnBlcokSize, ReadAlignment.cpp:2

struct Params {
    int max_pos_diff = 30;       // MaxPosDiff, main.cpp:179
    float max_mm_rate = 0.05f;   // MaxMisMatchRate, main.cpp:186
    bool use_nw = true;          // NW_ALG, main.cpp:166
    bool unique = true;          // bUnique, main.cpp:164
};

static uint8_t g_nt4[256];
static bool g_nt4_ready = false;
// nst_nt4_table, BWT_Index/bntseq.c:40-57: A/a C/c G/g T/t -> 0..3, everything else 4
static void init_nt4()
{
    if (g_nt4_ready) return;
    memset(g_nt4, 4, sizeof(g_nt4));
    g_nt4[(int)'A'] = g_nt4[(int)'a'] = 0;
    g_nt4[(int)'C'] = g_nt4[(int)'c'] = 1;
    g_nt4[(int)'G'] = g_nt4[(int)'g'] = 2;
    g_nt4[(int)'T'] = g_nt4[(int)'t'] = 3;
    g_nt4_ready = true;
}

// GetComplementaryBase, tools.cpp:3-18 (anything that is not ACGT/acgt becomes 'N')
static inline char comp_base(char c)
{
    switch (c) {
    case 'A': case 'a': return 'T';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    case 'T': case 't': return 'A';
    default: return 'N';
    }
}

static void revcomp_inplace(std::string &s) // SelfComplementarySeq, tools.cpp:31-43
{
    std::reverse(s.begin(), s.end());
    for (size_t i = 0; i < s.size(); i++) s[i] = comp_base(s[i]);
}

// ---------------------------------------------------------------------------------------------
// index  (bwt_t structure.h:32-42; files written by BWT_Index/bwtindex.c:53-75, bwt.c:174-196)
// ---------------------------------------------------------------------------------------------
struct Chrom {
    std::string name;
    int len;
    i64 fwd_off; // FowardLocation
    i64 rev_off; // ReverseLocation
};

} // namespace

struct mcxo_index {
    u64 primary = 0, L2[5] = {0, 0, 0, 0, 0}, seq_len = 0;
    std::vector<u32> bwt; // 64-byte blocks: 4 x u64 occ + 8 x u32 (128 bases)
    std::vector<u64> sa;  // sa[0] = (u64)-1
    int sa_intv = 32;
    i64 G = 0, G2 = 0;
    std::vector<Chrom> chr;
    std::vector<std::pair<i64, int>> ends; // PosChrIdMap (bwt_index.cpp:253-254) as a sorted array
    std::string ref;                       // RefSequence: forward + reverse complement, ASCII
};

namespace {

typedef mcxo_index Index;

static bool slurp(const std::string &path, std::vector<uint8_t> &out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n);
    size_t got = n ? fread(out.data(), 1, n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

// bwt_restore_bwt :105-124, bwt_restore_sa :16-36, bns_restore_core :38-92,
// RestoreReferenceInfo :232-258, IdvLoadReferenceSequences :196-215 (all bwt_index.cpp)
static Index *load_index(const std::string &prefix)
{
    init_nt4();
    Index *ix = new Index();
    std::vector<uint8_t> raw;
    if (!slurp(prefix + ".bwt", raw) || raw.size() < 40) { delete ix; return nullptr; }
    memcpy(&ix->primary, raw.data(), 8);
    memcpy(&ix->L2[1], raw.data() + 8, 32);
    ix->seq_len = ix->L2[4];
    ix->bwt.resize((raw.size() - 40) / 4);
    memcpy(ix->bwt.data(), raw.data() + 40, ix->bwt.size() * 4);

    if (!slurp(prefix + ".sa", raw) || raw.size() < 56) { delete ix; return nullptr; }
    u64 intv;
    memcpy(&intv, raw.data() + 40, 8);
    ix->sa_intv = (int)intv;
    u64 n_sa = (ix->seq_len + ix->sa_intv) / ix->sa_intv;
    ix->sa.assign(n_sa, 0);
    ix->sa[0] = (u64)-1;
    size_t avail = (raw.size() - 56) / 8;
    memcpy(ix->sa.data() + 1, raw.data() + 56, std::min<size_t>(avail, n_sa - 1) * 8);

    // .ann: "l_pac n_seqs seed" then per sequence "gi name [anno]\n offset len n_ambs"
    FILE *f = fopen((prefix + ".ann").c_str(), "r");
    if (!f) { delete ix; return nullptr; }
    long long l_pac; int n_seqs; unsigned seed;
    if (fscanf(f, "%lld%d%u", &l_pac, &n_seqs, &seed) != 3) { fclose(f); delete ix; return nullptr; }
    ix->G = l_pac; ix->G2 = 2 * ix->G;
    i64 total = 0;
    for (int i = 0; i < n_seqs; i++) {
        unsigned gi; char name[1024]; long long off; int len, n_ambs;
        if (fscanf(f, "%u%1023s", &gi, name) != 2) break;
        int c;
        while ((c = fgetc(f)) != '\n' && c != EOF) {}
        if (fscanf(f, "%lld%d%d", &off, &len, &n_ambs) != 3) break;
        Chrom ch; ch.name = name; ch.len = len; ch.fwd_off = total; total += len; ch.rev_off = ix->G2 - total;
        ix->chr.push_back(ch);
    }
    fclose(f);
    for (size_t i = 0; i < ix->chr.size(); i++) {
        ix->ends.push_back(std::make_pair(ix->chr[i].fwd_off + ix->chr[i].len - 1, (int)i));
        ix->ends.push_back(std::make_pair(ix->chr[i].rev_off + ix->chr[i].len - 1, (int)i));
    }
    std::sort(ix->ends.begin(), ix->ends.end());

    if (!slurp(prefix + ".pac", raw)) { delete ix; return nullptr; }
    ix->ref.assign((size_t)ix->G2, 'N');
    static const char fw[4] = {'A', 'C', 'G', 'T'}, rc[4] = {'T', 'G', 'C', 'A'};
    for (i64 p = 0; p < ix->G; p++) {
        int b = (raw[p >> 2] >> ((~p & 3) << 1)) & 3;
        ix->ref[p] = fw[b];
        ix->ref[ix->G2 - 1 - p] = rc[b];
    }
    return ix;
}

// first chromosome end >= gPos (PosChrIdMap.lower_bound); returns index into ends or -1
static inline int end_slot(const Index &ix, i64 gPos)
{
    auto it = std::lower_bound(ix.ends.begin(), ix.ends.end(), std::make_pair(gPos, -1));
    return it == ix.ends.end() ? -1 : (int)(it - ix.ends.begin());
}

// GetAlignmentBoundary, tools.cpp:112-117.  For gPos beyond the last end the reference
// dereferences map::end(); the value is never used on that path (sentinel seed only).
static inline i64 boundary_of(const Index &ix, i64 gPos)
{
    int s = end_slot(ix, gPos);
    return s < 0 ? -1 : ix.ends[s].first;
}

// ---------------------------------------------------------------------------------------------
// FM-index primitives
// ---------------------------------------------------------------------------------------------
// count the four symbols among the first n (0..32) 2-bit symbols, MSB first, of w
static inline void count4(u64 w, int n, u64 cnt[4])
{
    if (n <= 0) return;
    const u64 even = 0x5555555555555555ull;
    u64 keep = n >= 32 ? ~0ull : ~((1ull << (64 - 2 * n)) - 1);
    u64 hi = (w >> 1) & even, lo = w & even, m = keep & even;
    cnt[0] += __builtin_popcountll(~hi & ~lo & m);
    cnt[1] += __builtin_popcountll(~hi & lo & m);
    cnt[2] += __builtin_popcountll(hi & ~lo & m);
    cnt[3] += __builtin_popcountll(hi & lo & m);
}

// bwt_occ4, bwt_search.cpp:49-66: occurrences of each base in BWT[0..k]
static void occ4(const Index &ix, u64 k, u64 cnt[4])
{
    if (k == (u64)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
    k -= (k >= ix.primary);
    const u32 *blk = ix.bwt.data() + ((k >> 7) << 4);
    memcpy(cnt, blk, 32);
    int n = (int)(k & 127) + 1;
    for (int w = 0; w < 4 && n > 0; w++, n -= 32) {
        u64 word = ((u64)blk[8 + 2 * w] << 32) | blk[9 + 2 * w];
        count4(word, n, cnt);
    }
}

// bwt_occ, bwt_search.cpp:25-47
static u64 occ1(const Index &ix, u64 k, int c)
{
    if (k == ix.seq_len) return ix.L2[c + 1] - ix.L2[c];
    if (k == (u64)-1) return 0;
    u64 cnt[4];
    occ4(ix, k, cnt);
    return cnt[c];
}

// bwt_invPsi, bwt_search.cpp:101-107
static inline u64 lf_step(const Index &ix, u64 k)
{
    if (k == ix.primary) return 0;
    u64 x = k - (k > ix.primary);
    u32 word = ix.bwt[((x >> 7) << 4) + 8 + ((x & 127) >> 4)];
    int c = (word >> ((~x & 15) << 1)) & 3;
    return ix.L2[c] + occ1(ix, k, c);
}

// bwt_sa, bwt_search.cpp:109-119
static u64 sa_lookup(const Index &ix, u64 k, i64 *lf_steps = nullptr)
{
    u64 steps = 0, mask = (u64)ix.sa_intv - 1;
    while (k & mask) { ++steps; k = lf_step(ix, k); }
    if (lf_steps) *lf_steps += (i64)steps;
    return steps + ix.sa[k / ix.sa_intv];
}

struct SeedResult { int len; int freq; u64 x0, x2; };

// BWT_Search, bwt_search.cpp:121-164 (forward extension of a bi-interval, BWA-MEM style)
static SeedResult fm_search(const Index &ix, const uint8_t *seq, int start, int stop)
{
    int c0 = seq[start];
    u64 x0 = ix.L2[c0] + 1, x1 = ix.L2[3 - c0] + 1, x2 = ix.L2[c0 + 1] - ix.L2[c0];
    int pos;
    for (pos = start + 1; pos < stop; pos++) {
        if (seq[pos] > 3) break;
        u64 tk[4], tl[4];
        // bwt_2occ4 (:68-99) is two bwt_occ4 calls sharing one block when it can
        occ4(ix, x1 - 1, tk);
        occ4(ix, x1 - 1 + x2, tl);
        u64 n1[4], n2[4], n0[4];
        for (int b = 0; b < 4; b++) { n1[b] = ix.L2[b] + 1 + tk[b]; n2[b] = tl[b] - tk[b]; }
        n0[3] = x0 + ((x1 <= ix.primary && x1 + x2 - 1 >= ix.primary) ? 1 : 0);
        n0[2] = n0[3] + n2[3];
        n0[1] = n0[2] + n2[2];
        n0[0] = n0[1] + n2[1];
        int b = 3 - seq[pos];
        if (n2[b] == 0) break;
        x0 = n0[b]; x1 = n1[b]; x2 = n2[b];
    }
    SeedResult r;
    r.len = pos - start; r.x0 = x0; r.x2 = x2;
    if (r.len < kMinSeed) r.freq = 0;
    else r.freq = ((int)x2 <= kOccThr) ? (int)x2 : 0;
    return r;
}

// ---------------------------------------------------------------------------------------------
// gapped extension kernels
// ---------------------------------------------------------------------------------------------
// ksw_extz2_sse (ksw2_alignment.cpp:70-248) with the arguments ksw2_alignment passes
// (m=5, q=2, e=1, w=-1 -> full band), restated one cell at a time.  Lane arithmetic of the SSE
// version is int8 with the mixed signed/unsigned ops noted below; the 16-lane padding cells the
// SSE code also computes never feed a cell inside the matrix, so they are not computed here.
// Returns the reversed op string of ksw_backtrack (:25-68); *score = ez.score.
static std::string ksw2_ops(const uint8_t *query, int qlen, const uint8_t *target, int tlen, int *score)
{
    const int8_t Q = 2, E = 1;
    const int QE = Q + E;
    const int8_t QE2 = (int8_t)(QE * 2);
    const int8_t SC_MCH = 1, SC_MIS = -1; // mat[0], mat[1] (:9)
    const uint8_t MAX_SC = (uint8_t)(SC_MCH + QE * 2);
    const int NEG_INF = -0x40000000;
    std::string ops;
    *score = NEG_INF;
    if (qlen <= 0 || tlen <= 0) return ops;

    std::vector<int8_t> u(tlen, 0), v(tlen, 0), x(tlen, 0), y(tlen, 0);
    std::vector<int32_t> H(tlen, NEG_INF);
    std::vector<uint8_t> dir((size_t)(qlen + tlen - 1) * tlen, 0);

    for (int r = 0; r < qlen + tlen - 1; r++) {
        int st = std::max(0, r - qlen + 1), en = std::min(tlen - 1, r);
        int8_t x_left, v_left; // (r-1, t-1) values (:159-164)
        if (st > 0) { x_left = x[st - 1]; v_left = v[st - 1]; }
        else { x_left = 0; v_left = r ? Q : 0; }
        if (en == r) { y[r] = 0; u[r] = r ? Q : 0; } // first row of the matrix (:165)
        uint8_t *drow = &dir[(size_t)r * tlen];
        for (int t = st; t <= en; t++) {
            uint8_t tb = target[t], qb = query[r - t];
            int8_t sc = (tb == 4 || qb == 4) ? 0 : (tb == qb ? SC_MCH : SC_MIS); // :167-176
            int8_t z = (int8_t)(sc + QE2);
            int8_t a = (int8_t)(x_left + v_left);
            int8_t ut = u[t];
            int8_t b = (int8_t)(y[t] + ut);
            int8_t vt1 = v_left;
            x_left = x[t]; v_left = v[t];
            uint8_t d = a > z ? 1 : 0;                     // signed compare (:187)
            z = z > a ? z : a;                             // signed max (:188)
            if (b > z) d = 2;                              // signed compare (:189-190)
            uint8_t zu = std::max((uint8_t)z, (uint8_t)b); // unsigned max (:89)
            zu = std::min(zu, MAX_SC);                     // unsigned min (:90)
            z = (int8_t)zu;
            u[t] = (int8_t)(z - vt1);
            v[t] = (int8_t)(z - ut);
            z = (int8_t)(z - Q);
            a = (int8_t)(a - z);
            b = (int8_t)(b - z);
            if (a > 0) { x[t] = a; d |= 0x08; } else x[t] = 0;
            if (b > 0) { y[t] = b; d |= 0x10; } else y[t] = 0;
            drow[t] = d;
        }
        // H[] bookkeeping (:200-239); u8/v8 are read as unsigned bytes there
        if (r > 0) {
            H[en] = en > 0 ? H[en - 1] + (uint8_t)u[en] - QE : H[en] + (uint8_t)v[en] - QE;
            for (int t = st; t < en; t++) H[t] += (int32_t)(uint8_t)v[t] - QE;
        } else H[0] = (uint8_t)v[0] - QE - QE;
        if (r == qlen + tlen - 2 && en == tlen - 1) *score = H[tlen - 1];
    }
    // ksw_backtrack (:25-68); with the full band force_state never fires
    int i = tlen - 1, j = qlen - 1, state = 0;
    while (i >= 0 && j >= 0) {
        uint32_t d = dir[(size_t)(i + j) * tlen + i];
        if (state == 0) state = d & 7;
        else if (!((d >> (state + 2)) & 1)) state = 0;
        if (state == 0) state = d & 7;
        if (state == 0) { ops.push_back('M'); --i; --j; }
        else if (state == 1 || state == 3) { ops.push_back('D'); --i; }
        else { ops.push_back('I'); --j; }
    }
    if (i >= 0) ops.append(i + 1, 'D');
    if (j >= 0) ops.append(j + 1, 'I');
    return ops;
}

// ksw2_alignment, ksw2_alignment.cpp:250-272
static void ksw2_align(std::string &s1, std::string &s2, i64 *cells = nullptr)
{
    init_nt4();
    int m = (int)s1.size(), n = (int)s2.size(), score;
    std::vector<uint8_t> q(m), t(n);
    for (int i = 0; i < m; i++) q[i] = g_nt4[(uint8_t)s1[i]];
    for (int i = 0; i < n; i++) t[i] = g_nt4[(uint8_t)s2[i]];
    std::string ops = ksw2_ops(q.data(), m, t.data(), n, &score);
    if (cells) *cells += (i64)m * n;
    int p = 0;
    for (int i = (int)ops.size() - 1; i >= 0; i--, p++) {
        if (ops[i] == 'D') s1.insert(s1.begin() + p, '-');
        else if (ops[i] == 'I') s2.insert(s2.begin() + p, '-');
    }
}

// nw_alignment, nw_alignment.cpp:18-83.  The reference works in float with scores that are all
// multiples of 0.5 and |x| < 2^17, hence exact; here everything is doubled and kept in int32.
// Traceback compares s with r, then with t, for equality (gap preferred on ties, :59-74).
static void nw_align(std::string &s1, std::string &s2, i64 *cells = nullptr)
{
    init_nt4();
    const int NEG = -131072, EXT = -1, NEW = -3;
    int m = (int)s1.size() + 1, n = (int)s2.size() + 1;
    if (cells) *cells += (i64)(m - 1) * (n - 1);
    std::vector<int> R((size_t)m * n), T((size_t)m * n), S((size_t)m * n);
#define AT(M, i, j) M[(size_t)(i) * n + (j)]
    AT(R, 0, 0) = AT(T, 0, 0) = AT(S, 0, 0) = 0;
    for (int i = 1; i < m; i++) { AT(R, i, 0) = NEG; AT(S, i, 0) = AT(T, i, 0) = -2 - i; }
    for (int j = 1; j < n; j++) { AT(T, 0, j) = NEG; AT(S, 0, j) = AT(R, 0, j) = -2 - j; }
    for (int i = 1; i < m; i++) {
        uint8_t ci = g_nt4[(uint8_t)s1[i - 1]];
        for (int j = 1; j < n; j++) {
            int r = std::max(AT(R, i, j - 1) + EXT, AT(S, i, j - 1) + NEW);
            int t = std::max(AT(T, i - 1, j) + EXT, AT(S, i - 1, j) + NEW);
            int d = AT(S, i - 1, j - 1) + (ci == g_nt4[(uint8_t)s2[j - 1]] ? 2 : -2);
            AT(R, i, j) = r; AT(T, i, j) = t;
            AT(S, i, j) = std::max(d, std::max(r, t));
        }
    }
    int i = m - 1, j = n - 1;
    while (i > 0 || j > 0) {
        if (AT(S, i, j) == AT(R, i, j)) { s1.insert(s1.begin() + i, '-'); j--; }
        else if (AT(S, i, j) == AT(T, i, j)) { s2.insert(s2.begin() + j, '-'); i--; }
        else { i--; j--; }
    }
#undef AT
}

// ---------------------------------------------------------------------------------------------
// per-read state
// ---------------------------------------------------------------------------------------------
struct Frag {            // FragPair_t, structure.h:113-123
    bool simple;
    int rPos;
    i64 gPos;
    int rLen, gLen;
    i64 posDiff;
    std::string a1, a2;  // read / genome fragment alignment
};

struct Cand {            // AlnCan_t, structure.h:125-133
    int score = 0;
    int flag = 0;
    bool fwd = true;     // orientation
    int mate = -1;       // PairedAlnCanIdx
    std::vector<Frag> frags;
};

struct Read {            // ReadItem_t, structure.h:142-150
    int rlen = 0;
    std::string name, seq, qual;
    int best = -1, score = 0, sub = 0; // AlnSummary
    std::vector<Cand> cands;
};

struct Counters { i64 ext_steps = 0, sa_hits = 0, lf_steps = 0, dp_calls = 0, dp_cells = 0; };

static bool by_posdiff(const Frag &a, const Frag &b) // CompByPosDiff, ReadMapping.cpp:43-47
{
    return a.posDiff == b.posDiff ? a.rPos < b.rPos : a.posDiff < b.posDiff;
}

static bool by_readpos(const Frag &a, const Frag &b) // CompByReadPos, ReadAlignment.cpp:23-27
{
    return a.rPos == b.rPos ? a.gPos < b.gPos : a.rPos < b.rPos;
}

// IdentifySimplePairs, ReadMapping.cpp:125-158
static std::vector<Frag> find_seeds(const Index &ix, const std::string &seq, Counters &ct)
{
    int rlen = (int)seq.size();
    std::vector<uint8_t> enc(rlen); // EnCodeReadSeq, ReadMapping.cpp:404-407
    for (int i = 0; i < rlen; i++) enc[i] = g_nt4[(uint8_t)seq[i]];
    std::vector<Frag> out;
    int pos = 0, stop = rlen - kMinSeed;
    while (pos < stop) {
        if (enc[pos] > 3) { pos++; continue; }
        SeedResult s = fm_search(ix, enc.data(), pos, rlen);
        ct.ext_steps += s.len;
        for (int i = 0; i < s.freq; i++) {
            u64 loc = sa_lookup(ix, s.x0 + i, &ct.lf_steps);
            ct.sa_hits++;
            Frag f; f.simple = true; f.rPos = pos; f.rLen = f.gLen = s.len;
            f.gPos = (i64)loc; f.posDiff = f.gPos - pos;
            if (f.posDiff > 0) out.push_back(f);
        }
        pos += s.len + 1;
    }
    std::sort(out.begin(), out.end(), by_posdiff);
    Frag end; end.simple = true; end.rPos = 0; end.rLen = end.gLen = 0; end.gPos = end.posDiff = ix.G2;
    out.push_back(end); // terminal fragment pair
    return out;
}

// IdentifyClosestFragmentPairs, ReadMapping.cpp:160-192: best run of equal PosDiff in [b,e)
static Cand best_equal_run(const std::vector<Frag> &v, int b, int e)
{
    Cand c;
    int run_b = b, s = v[b].rLen, best_b = b, best_e = b;
    int j;
    for (j = b + 1; j < e; j++) {
        if (v[j].posDiff != v[run_b].posDiff) {
            if (s > c.score) { c.score = s; best_b = run_b; best_e = j; }
            run_b = j; s = v[j].rLen;
        } else s += v[j].rLen;
    }
    if (s > c.score) { c.score = s; best_b = run_b; best_e = j; }
    c.frags.assign(v.begin() + best_b, v.begin() + best_e);
    return c;
}

// SimplePairClustering, ReadMapping.cpp:194-226
static std::vector<Cand> cluster_seeds(const Index &ix, const Params &pm, int rlen, const std::vector<Frag> &v)
{
    std::vector<Cand> out;
    int num = (int)v.size(), head = 0, score = v[0].rLen, thr = rlen >> 2;
    i64 g_end = boundary_of(ix, v[0].gPos);
    for (int i = 0, j = 1; j < num; i++, j++) {
        i64 d = v[j].posDiff - v[i].posDiff;
        if (v[j].gPos > g_end || (d < 0 ? -d : d) > pm.max_pos_diff) {
            if (score > thr) {
                if (thr < (score >> 1)) thr = score >> 1;
                if (score >= rlen) out.push_back(best_equal_run(v, head, j)); // tandem repeats
                else {
                    Cand c; c.score = score; c.frags.assign(v.begin() + head, v.begin() + j);
                    out.push_back(c);
                }
            }
            head = j; g_end = boundary_of(ix, v[j].gPos); score = v[j].rLen;
        } else score += v[j].rLen;
    }
    return out;
}

// RemoveRedundantAlnCan, ReadMapping.cpp:228-242
static void keep_top_scores(std::vector<Cand> &cs)
{
    if (cs.size() <= 1) return;
    int best = 0;
    for (auto &c : cs) best = std::max(best, c.score);
    for (auto &c : cs) if (c.score < best) c.score = 0;
}

// CheckPairedAlignmentDistance, ReadMapping.cpp:244-303
static int pair_by_distance(i64 est, std::vector<Cand> &c1, std::vector<Cand> &c2)
{
    struct P { int a, b, s; };
    std::vector<P> picks;
    int n1 = (int)c1.size(), n2 = (int)c2.size(), paired = 0;
    i64 top = 0;
    if (n1 * n2 > 100) { keep_top_scores(c1); keep_top_scores(c2); }
    for (int i = 0; i < n1; i++) {
        if (c1[i].score == 0) continue;
        int pick = -1, pscore = 0;
        for (int j = 0; j < n2; j++) {
            if (c2[j].score == 0 || c2[j].frags[0].posDiff < c1[i].frags[0].posDiff) continue;
            if (c2[j].frags[0].posDiff - c1[i].frags[0].posDiff < est && c2[j].score > pscore) {
                pick = j; pscore = c2[j].score;
            }
        }
        if (pick < 0) continue;
        int s = c1[i].score + c2[pick].score;
        if (s > top) { top = s; picks.push_back({i, pick, s}); }
        else if (s == top) picks.push_back({i, pick, s});
    }
    if (top > 0)
        for (auto &p : picks)
            if (p.s == top) { paired++; c1[p.a].mate = p.b; c2[p.b].mate = p.a; }
    return paired;
}

// MaskUnPairedAlnCan, ReadMapping.cpp:305-322
static void mask_unpaired(std::vector<Cand> &c1, std::vector<Cand> &c2)
{
    int top = 0;
    for (auto &c : c1) if (c.mate != -1) top = std::max(top, c.score + c2[c.mate].score);
    for (auto &c : c1) if (c.mate == -1 || c.score + c2[c.mate].score < top) c.score = 0;
    for (auto &c : c2) if (c.mate == -1 || c.score + c1[c.mate].score < top) c.score = 0;
}

// ---- mate rescue (AlignmentRescue.cpp + KmerAnalysis.cpp) -----------------------------------
struct Kmer { u32 wid, pos; };                 // KmerItem_t
struct KmerPair { int diff; u32 rPos, gPos; }; // KmerPair_t

// CreateKmerVecFromReadSeq, KmerAnalysis.cpp:57-103 (only a literal 'N' breaks a k-mer)
static std::vector<Kmer> kmers_of(const char *s, int len)
{
    std::vector<Kmer> v;
    u32 tail = 0, count = 0, head;
    while (count < (u32)kKmer && tail < (u32)len) { if (s[tail++] != 'N') count++; else count = 0; }
    if (count != (u32)kKmer) return v;
    auto full_id = [&](u32 at) { u32 id = 0; for (u32 i = at; i < at + kKmer; i++) id = (id << 2) + g_nt4[(uint8_t)s[i]]; return id; };
    Kmer k; k.pos = head = tail - kKmer; k.wid = full_id(head);
    v.push_back(k);
    for (head += 1; tail < (u32)len; head++, tail++) {
        if (s[tail] != 'N') {
            k.pos = head; k.wid = ((k.wid & kKmerMask) << 2) + g_nt4[(uint8_t)s[tail]];
            v.push_back(k);
        } else {
            count = 0; tail++;
            while (count < (u32)kKmer && tail < (u32)len) { if (s[tail++] != 'N') count++; else count = 0; }
            if (count != (u32)kKmer) break;
            k.pos = head = tail - kKmer; k.wid = full_id(head);
            v.push_back(k);
        }
    }
    std::sort(v.begin(), v.end(), [](const Kmer &a, const Kmer &b) { return a.wid < b.wid; });
    return v;
}

// IdentifyCommonKmers, KmerAnalysis.cpp:105-131.  vec2 is ordered by wid only (std::sort on
// that key alone), so equal-wid entries may come in any order; the result is re-sorted by a
// total order (PosDiff, rPos) and (rPos, PosDiff) determines gPos, so the outcome is fixed.
static std::vector<KmerPair> common_kmers(u32 max_shift, const std::vector<Kmer> &a, const std::vector<Kmer> &b)
{
    std::vector<KmerPair> out;
    for (const Kmer &k : a) {
        auto it = std::lower_bound(b.begin(), b.end(), k, [](const Kmer &x, const Kmer &y) { return x.wid < y.wid; });
        for (; it != b.end() && it->wid == k.wid; ++it) {
            if ((it->pos >= k.pos && it->pos - k.pos < max_shift) || (it->pos < k.pos && k.pos - it->pos < max_shift)) {
                KmerPair p; p.rPos = k.pos; p.gPos = it->pos; p.diff = (int)(p.gPos - p.rPos);
                out.push_back(p);
            }
        }
    }
    std::sort(out.begin(), out.end(), [](const KmerPair &x, const KmerPair &y) {
        return x.diff == y.diff ? x.rPos < y.rPos : x.diff < y.diff;
    });
    return out;
}

// GenerateSimplePairsFromCommonKmers, KmerAnalysis.cpp:133-163
static std::vector<Frag> seeds_from_kmers(int thr, i64 base, const std::vector<KmerPair> &kp)
{
    std::vector<Frag> out;
    int num = (int)kp.size();
    for (int i = 0; i < num;) {
        int j, next = (int)kp[i].rPos + 1;
        for (j = i + 1; j < num; j++) {
            if (kp[j].rPos != (u32)next || kp[j].diff != kp[i].diff) break;
            next++;
        }
        int l = kKmer + (j - 1 - i);
        if (l >= thr) {
            Frag f; f.simple = true; f.rPos = (int)kp[i].rPos; f.gPos = (i64)kp[i].gPos + base;
            f.posDiff = (i64)kp[i].diff + base; f.rLen = f.gLen = l;
            out.push_back(f);
        }
        i = j;
    }
    return out;
}

// IdentifyBestAlnCan, AlignmentRescue.cpp:3-26
static Cand best_run_cand(const std::vector<Frag> &v)
{
    Cand c;
    int num = (int)v.size();
    for (int i = 0; i < num;) {
        int s = v[i].rLen, j;
        for (j = i + 1; j < num && v[j].posDiff == v[i].posDiff; j++) s += v[j].rLen;
        if (s > c.score) { c.score = s; c.frags.assign(v.begin() + i, v.begin() + j); }
        i = j;
    }
    return c;
}

// AlignmentRescue, AlignmentRescue.cpp:28-111.  Windows that leave [0, 2G) make the reference
// read outside RefSequence (undefined); they are skipped here and in the product.
static int rescue_mate(const Index &ix, u32 est, Read &r1, Read &r2)
{
    int s1 = 0, s2 = 0, paired = 0;
    for (auto &c : r1.cands) s1 = std::max(s1, c.score);
    for (auto &c : r2.cands) s2 = std::max(s2, c.score);
    int mode;
    if (s1 < (r1.rlen >> 2) && s2 < (r2.rlen >> 2)) return 0;
    else if (s1 - s2 > (r2.rlen >> 2)) mode = 1;
    else if (s2 - s1 > (r1.rlen >> 2)) mode = 2;
    else mode = 3;
    int n1 = (int)r1.cands.size(), n2 = (int)r2.cands.size();
    auto same_chr = [&](i64 a, i64 b) {
        int sa = end_slot(ix, a), sb = end_slot(ix, b);
        return sa >= 0 && sb >= 0 && ix.ends[sa].second == ix.ends[sb].second;
    };
    if (mode == 1 || mode == 3) { // place read2 next to read1's candidates
        std::vector<Kmer> kq = kmers_of(r2.seq.c_str(), r2.rlen);
        int thr = s1 >> 1;
        size_t lim = r1.cands.size();
        for (size_t ci = 0; ci < lim; ci++) {
            Cand &c = r1.cands[ci];
            if (c.score < thr || c.mate != -1) continue;
            i64 left = c.frags[0].posDiff, right = c.frags[0].posDiff + est + r2.rlen;
            if (right > ix.G2) right = ix.G2;
            if (left < 0 || right >= ix.G2 || !same_chr(left, right)) continue;
            int slen = (int)(right - left);
            if (slen < r2.rlen) continue;
            std::vector<Kmer> kg = kmers_of(ix.ref.c_str() + left, slen);
            std::vector<KmerPair> kp = common_kmers((u32)slen, kq, kg);
            std::vector<Frag> sp = seeds_from_kmers(10, left, kp);
            if (sp.empty()) continue;
            Cand nc = best_run_cand(sp);
            if (nc.score > s2) {
                paired++;
                r1.cands[ci].mate = n2++;
                nc.mate = (int)ci;
                r2.cands.push_back(nc);
            }
        }
    }
    if (mode == 2 || mode == 3) { // place read1 next to read2's candidates
        std::vector<Kmer> kq = kmers_of(r1.seq.c_str(), r1.rlen);
        int thr = s2 >> 1;
        size_t lim = r2.cands.size();
        for (size_t ci = 0; ci < lim; ci++) {
            Cand &c = r2.cands[ci];
            if (c.score < thr || c.mate != -1) continue;
            i64 left = c.frags[0].posDiff - (i64)est, right = c.frags[0].posDiff + r1.rlen;
            if (right > ix.G2) right = ix.G2;
            if (left < 0 || right >= ix.G2 || !same_chr(left, right)) continue;
            int slen = (int)(right - left);
            if (slen < r1.rlen) continue;
            std::vector<Kmer> kg = kmers_of(ix.ref.c_str() + left, slen);
            std::vector<KmerPair> kp = common_kmers((u32)slen, kq, kg);
            std::vector<Frag> sp = seeds_from_kmers(10, left, kp);
            if (sp.empty()) continue;
            Cand nc = best_run_cand(sp);
            if (nc.score > s1) {
                paired++;
                r2.cands[ci].mate = n1++;
                nc.mate = (int)ci;
                r1.cands.push_back(nc);
            }
        }
    }
    return paired;
}

// ---- ProduceReadAlignment and helpers (ReadAlignment.cpp) -----------------------------------
// RemoveOverlaps :38-65 followed by RemoveNullFragPairs :29-36
static void trim_overlaps(std::vector<Frag> &v)
{
    bool any = false;
    int num = (int)v.size();
    for (int i = 0, j = 1; j < num; i++, j++) {
        if (v[i].rPos == v[j].rPos) { any = true; v[i].rLen = v[i].gLen = 0; }
        else if (v[i].gPos >= v[j].gPos || v[i].gPos + v[i].gLen > v[j].gPos) {
            any = true;
            int ov = (int)(v[i].gPos + v[i].gLen - v[j].gPos);
            if ((v[i].rLen -= ov) < 0) v[i].rLen = 0;
            if ((v[i].gLen -= ov) < 0) v[i].gLen = 0;
        }
    }
    if (any) v.erase(std::remove_if(v.begin(), v.end(), [](const Frag &f) { return f.rLen == 0; }), v.end());
}

// IdentifyNormalPairs :67-108
static void add_gap_frags(int rlen, std::vector<Frag> &v)
{
    int num = (int)v.size();
    Frag g; g.simple = false;
    for (int i = 0, j = 1; j < num; i++, j++) {
        int rg = v[j].rPos - (v[i].rPos + v[i].rLen); if (rg < 0) rg = 0;
        int gg = (int)(v[j].gPos - (v[i].gPos + v[i].gLen)); if (gg < 0) gg = 0;
        if (rg > 0 || gg > 0) {
            g.rPos = v[i].rPos + v[i].rLen; g.gPos = v[i].gPos + v[i].gLen;
            g.posDiff = g.gPos - g.rPos; g.rLen = rg; g.gLen = gg;
            v.push_back(g);
        }
    }
    if ((int)v.size() > num) std::inplace_merge(v.begin(), v.begin() + num, v.end(), by_readpos);
    if (v[0].rPos > 0) {
        g.rPos = 0; g.gPos = g.posDiff = v[0].posDiff; g.rLen = g.gLen = v[0].rPos;
        v.insert(v.begin(), g);
    }
    num = (int)v.size();
    if (num > 0 && v[num - 1].rPos + v[num - 1].rLen < rlen) {
        g.rPos = v[num - 1].rPos + v[num - 1].rLen; g.gPos = v[num - 1].gPos + v[num - 1].gLen;
        g.posDiff = v[num - 1].posDiff; g.rLen = g.gLen = rlen - g.rPos;
        v.push_back(g);
    }
}

// CheckAlignmentValidity, tools.cpp:119-130
static bool on_one_chromosome(const Index &ix, const std::vector<Frag> &v)
{
    const Frag &a = v.front(), &b = v.back();
    if (a.gPos < 0 || b.gPos + b.gLen > ix.G2) return false;
    int s1 = end_slot(ix, a.gPos), s2 = end_slot(ix, b.gPos + b.gLen - 1);
    return s1 >= 0 && s2 >= 0 && ix.ends[s1].first == ix.ends[s2].first;
}

// ProcessNormalPair :155-191 (+ CalFragPairMismatches :133-142)
static void align_gap_frag(const Index &ix, const Params &pm, const std::string &seq, Frag &f, Counters &ct)
{
    if (f.rLen > 0) f.a1.assign(seq, f.rPos, f.rLen); else f.a1.assign(f.gLen, '-');
    if (f.gLen > 0) f.a2.assign(ix.ref, (size_t)f.gPos, f.gLen); else f.a2.assign(f.rLen, '-');
    if (f.gPos >= ix.G) {
        if (f.rLen > 0) revcomp_inplace(f.a1);
        if (f.gLen > 0) revcomp_inplace(f.a2);
    }
    if (f.rLen > 0 && f.gLen > 0) {
        bool dp = f.rLen != f.gLen;
        if (!dp) {
            int mm = 0;
            for (int i = 0; i < f.rLen; i++) if (f.a1[i] != f.a2[i]) mm++;
            dp = mm > 1 && mm >= (int)(f.rLen * 0.2);
        }
        if (dp) {
            ct.dp_calls++;
            if (pm.use_nw) nw_align(f.a1, f.a2, &ct.dp_cells); else ksw2_align(f.a1, f.a2, &ct.dp_cells);
        }
    }
}

// RemoveHeadingGaps :264-283
static void strip_leading_gaps(bool move_pos, Frag &f)
{
    int len = (int)f.a1.size(), j, rs = 0, gs = 0;
    for (j = 0; j < len; j++) {
        if (f.a1[j] == '-') gs++; else if (f.a2[j] == '-') rs++; else break;
    }
    if (j > 0) {
        f.a1.erase(0, j); f.a2.erase(0, j);
        f.rLen -= rs; f.gLen -= gs;
        if (move_pos) { f.rPos += rs; f.gPos += gs; }
    }
}

// RemoveTailingGaps :285-304
static void strip_trailing_gaps(bool move_pos, Frag &f)
{
    int len = (int)f.a1.size(), j, rs = 0, gs = 0;
    for (j = len - 1; j >= 0; j--) {
        if (f.a1[j] == '-') gs++; else if (f.a2[j] == '-') rs++; else break;
    }
    if (++j < len) {
        f.a1.resize(j); f.a2.resize(j);
        f.rLen -= rs; f.gLen -= gs;
        if (move_pos) { f.rPos += rs; f.gPos += gs; }
    }
}

// CheckLocalAlignmentQuality :193-232
static bool local_quality_ok(const Frag &f)
{
    int kind = -1, switches = 0, n = 0, mis = 0, len = (int)f.a1.size();
    for (int i = 0; i < len; i++) {
        int k;
        if (f.a1[i] == '-') k = 0;
        else if (f.a2[i] == '-') k = 1;
        else { k = 2; n++; if (f.a1[i] != f.a2[i]) mis++; }
        if (k != kind) { kind = k; switches++; }
    }
    return !(switches >= 4 || (mis >= 3 && mis >= (int)(n * 0.3)));
}

// EvaluateAlignmentScore :234-245
static int alignment_score(const std::vector<Frag> &v)
{
    int s = 0;
    for (const Frag &f : v) {
        if (f.simple) s += f.rLen;
        else for (size_t i = 0; i < f.a1.size(); i++) if (f.a1[i] == f.a2[i]) s++;
    }
    return s;
}

// FindMisMatchNumber :247-262
static int mismatch_count(const std::vector<Frag> &v)
{
    int mm = 0;
    for (const Frag &f : v)
        if (!f.simple)
            for (size_t i = 0; i < f.a1.size(); i++)
                if (f.a1[i] != f.a2[i] && f.a1[i] != '-' && f.a2[i] != '-') mm++;
    return mm;
}

// ProduceReadAlignment :306-430
static bool extend_read(const Index &ix, const Params &pm, Read &rd, Counters &ct)
{
    int max_mm = (int)(rd.rlen * pm.max_mm_rate);
    for (size_t ci = 0; ci < rd.cands.size(); ci++) {
        Cand &c = rd.cands[ci];
        if (c.score == 0) continue;
        std::vector<Frag> &v = c.frags;
        std::sort(v.begin(), v.end(), by_readpos);
        trim_overlaps(v);
        add_gap_frags(rd.rlen, v);
        if (!on_one_chromosome(ix, v)) { c.score = 0; continue; }
        bool head_ok = true, tail_ok = true;
        int num = (int)v.size(), last = num - 1;
        for (int i = 0; i < num; i++) {
            if (v[i].simple) continue;
            align_gap_frag(ix, pm, rd.seq, v[i], ct);
            if (i == 0) {
                if (v[i].gPos < ix.G) strip_leading_gaps(true, v[i]); else strip_trailing_gaps(true, v[i]);
                if ((int)v[i].a1.size() >= kMinBlock && !local_quality_ok(v[i])) {
                    head_ok = false;
                    v[i].rLen = v[i].gLen = 0; v[i].a1.clear(); v[i].a2.clear();
                    v[i].rPos = v[i + 1].rPos; v[i].gPos = v[i + 1].gPos;
                }
            } else if (i == last) {
                if (v[i].gPos < ix.G) strip_trailing_gaps(false, v[i]); else strip_leading_gaps(false, v[i]);
                if ((int)v[i].a1.size() >= kMinBlock && !local_quality_ok(v[i])) {
                    tail_ok = false;
                    v[i].rLen = v[i].gLen = 0;
                    v[i].rPos = v[i - 1].rPos + v[i - 1].rLen; v[i].gPos = v[i - 1].gPos + v[i - 1].gLen;
                    v[i].a1.clear(); v[i].a2.clear();
                }
            } else if (v[i].rLen >= kMinBlock && v[i].gLen >= kMinBlock && !local_quality_ok(v[i])) {
                c.score = 0;
                break;
            }
        }
        if (c.score == 0) continue;
        if (!head_ok && !tail_ok) { c.score = 0; continue; }
        c.score = alignment_score(v);
        if (c.score == 0) continue;
        if (c.score < (int)(rd.rlen * (1 - pm.max_mm_rate)) && mismatch_count(v) > max_mm) { c.score = 0; continue; }
        c.fwd = v[0].gPos < ix.G;
        if (!c.fwd) std::reverse(v.begin(), v.end());
        if (c.score > rd.score) { rd.score = c.score; rd.best = (int)ci; }
        else if (c.score > rd.sub) rd.sub = c.score;
    }
    for (Cand &c : rd.cands) if (c.score < rd.score) c.score = 0;
    return rd.score > 0;
}

// ---- SAM (SamReport.cpp) ---------------------------------------------------------------------
struct Coord { i64 pos; int chr; };

// DetermineCoordinate, tools.cpp:132-164
static Coord to_coord(const Index &ix, i64 g)
{
    Coord c;
    bool one = ix.chr.size() == 1;
    if (g < ix.G) {
        if (one) { c.chr = 0; c.pos = g + 1; }
        else { int s = end_slot(ix, g); c.chr = ix.ends[s].second; c.pos = g + 1 - ix.chr[c.chr].fwd_off; }
    } else {
        if (one) { c.chr = 0; c.pos = ix.G2 - g; }
        else { int s = end_slot(ix, g); c.chr = ix.ends[s].second; c.pos = ix.ends[s].first - g + 1; }
    }
    return c;
}

// GetAlnCoordinate, SamReport.cpp:121-149
static Coord aln_coord(const Index &ix, const Cand &c)
{
    Coord k; k.pos = 0; k.chr = 0;
    for (const Frag &f : c.frags)
        if (f.gLen > 0) return to_coord(ix, c.fwd ? f.gPos : f.gPos + f.gLen - 1);
    return k;
}

// EvaluateMAPQ, SamReport.cpp:86-101
static int mapq_of(const Read &r)
{
    if (r.score == 0 || r.score == r.sub) return 0;
    if (r.sub == 0 || r.score - r.sub > 5) return 60;
    int q = (int)(30 * (1 - (float)(r.score - r.sub) / r.score) * log(r.score) + 0.4999);
    return q > 60 ? 60 : q;
}

// GenerateCIGARstring, SamReport.cpp:172-316
static std::string cigar_of(int rlen, const Cand &c)
{
    std::string out;
    char buf[32];
    const std::vector<Frag> &v = c.frags;
    int run = 0;
    char st = ' ';
    auto flush_to = [&](char ns) {
        if (st != ns) {
            if (run > 0) { snprintf(buf, sizeof buf, "%d%c", run, st); out += buf; }
            st = ns; run = 0;
        }
    };
    if (!v[0].simple) {
        int clip = c.fwd ? v[0].rPos : rlen - (v[0].rPos + v[0].rLen);
        if (clip > 0) { snprintf(buf, sizeof buf, "%dS", clip); out += buf; }
    }
    int num = (int)v.size();
    for (int i = 0; i < num; i++) {
        const Frag &f = v[i];
        if (f.simple) { flush_to('M'); run += f.rLen; }
        else if (!f.a1.empty()) {
            for (size_t j = 0; j < f.a1.size(); j++) {
                flush_to(f.a1[j] == '-' ? 'D' : (f.a2[j] == '-' ? 'I' : 'M'));
                run++;
            }
        } else if (f.rLen > 0) { flush_to('I'); run += f.rLen; }
        else if (f.gLen > 0) { flush_to('D'); run += f.gLen; }
    }
    if (run > 0) { snprintf(buf, sizeof buf, "%d%c", run, st); out += buf; }
    int i = num - 1;
    if (i > 0 && !v[i].simple) {
        int clip = c.fwd ? rlen - (v[i].rPos + v[i].rLen) : v[i].rPos;
        if (clip > 0) { snprintf(buf, sizeof buf, "%dS", clip); out += buf; }
    }
    return out;
}

// SetSingledAlignmentFlag, SamReport.cpp:7-24
static void set_single_flags(const Params &pm, Read &r)
{
    if (r.score > r.sub || !pm.unique) r.cands[r.best].flag = r.cands[r.best].fwd ? 0 : 0x10;
    else if (r.score > 0) { for (Cand &c : r.cands) if (c.score > 0) c.flag = c.fwd ? 0 : 0x10; }
}

// SetPairedAlignmentFlag, SamReport.cpp:26-84
static void set_paired_flags(Read &r1, Read &r2)
{
    auto one = [](Read &me, Read &other, int first_bit, bool me_is_first, bool unique_branch, Cand &c) {
        c.flag = first_bit;
        if (me_is_first) c.flag |= c.fwd ? 0x20 : 0x10; else c.flag |= c.fwd ? 0x10 : 0x20;
        if (c.mate != -1 && other.cands[c.mate].score > 0) c.flag |= 0x2;
        else {
            if (unique_branch) { if (me_is_first) c.flag |= c.fwd ? 0x10 : 0x20; else c.flag |= c.fwd ? 0x20 : 0x10; }
            c.flag |= 0x8;
        }
        (void)me;
    };
    if (r1.score > r1.sub) one(r1, r2, 0x41, true, true, r1.cands[r1.best]);
    else if (r1.score > 0) { for (Cand &c : r1.cands) if (c.score > 0) one(r1, r2, 0x41, true, false, c); }
    if (r2.score > r2.sub) one(r2, r1, 0x81, false, true, r2.cands[r2.best]);
    else if (r2.score > 0) { for (Cand &c : r2.cands) if (c.score > 0) one(r2, r1, 0x81, false, false, c); }
}

static std::string revcomp_copy(const std::string &s) // GetComplementarySeq, tools.cpp:20-29
{
    std::string r(s);
    revcomp_inplace(r);
    return r;
}

// GenerateSingleSamStream, SamReport.cpp:324-375.  For reverse-strand FASTQ records the
// reference leaves rqual[0] uninitialised (GetReverseQualityStr :318-322); we emit the plain
// reversal, tests compare fields 1-10 + tags in single-end mode.
static void sam_single(const Index &ix, const Params &pm, bool fastq, Read &r, std::vector<std::string> &out)
{
    char num[64];
    if (r.score == 0) {
        out.push_back(r.name + "\t4\t*\t0\t0\t*\t*\t0\t0\t" + r.seq + "\t" + (fastq ? r.qual : "*") + "\tAS:i:0\tXS:i:0");
        return;
    }
    set_single_flags(pm, r);
    int mq = mapq_of(r);
    std::string rseq, rqual;
    for (size_t i = r.best; i < r.cands.size(); i++) {
        Cand &c = r.cands[i];
        if (c.score != r.score) continue;
        if (!c.fwd && rseq.empty()) { rseq = revcomp_copy(r.seq); rqual.assign(r.qual.rbegin(), r.qual.rend()); }
        Coord k = aln_coord(ix, c);
        std::string line = r.name;
        snprintf(num, sizeof num, "\t%d\t", c.flag); line += num;
        line += ix.chr[k.chr].name;
        snprintf(num, sizeof num, "\t%lld\t%d\t", (long long)k.pos, mq); line += num;
        line += cigar_of(r.rlen, c);
        line += "\t*\t0\t0\t";
        line += c.fwd ? r.seq : rseq;
        line += "\t";
        line += fastq ? (c.fwd ? r.qual : rqual) : "*";
        snprintf(num, sizeof num, "\tNM:i:%d\tAS:i:%d\tXS:i:%d", r.rlen - c.score, r.score, r.sub); line += num;
        out.push_back(line);
        if (pm.unique) break;
    }
}

// GeneratePairedSamStream, SamReport.cpp:377-488.  `rq` persists from mate 1 to mate 2 exactly
// as the reference's `rqual` string does.
static void sam_paired(const Index &ix, const Params &pm, bool fastq, Read &r1, Read &r2, std::vector<std::string> &out)
{
    char num[96];
    std::string rq;
    set_paired_flags(r1, r2);
    auto unmapped = [&](Read &me, Read &other, int bit) {
        int fl = 0x1 | 0x4 | bit;
        if (other.score == 0) fl |= 0x8;
        else if (!other.cands.empty()) fl |= 0x30; // both 0x10 and 0x20 get set (:401-402, :449-450)
        snprintf(num, sizeof num, "\t%d\t*\t0\t0\t*\t*\t0\t0\t", fl);
        out.push_back(me.name + num + me.seq + "\t" + (fastq ? me.qual : "*") + "\tAS:i:0\tXS:i:0");
    };
    auto mapped = [&](Read &me, Read &other, bool me_is_first) {
        int mq = mapq_of(me);
        std::string rseq;
        for (size_t i = me.best; i < me.cands.size(); i++) {
            Cand &c = me.cands[i];
            if (c.score != me.score) continue;
            if (!c.fwd && rseq.empty()) { rseq = revcomp_copy(me.seq); if (fastq) { rq = me.qual; std::reverse(rq.begin(), rq.end()); } }
            Coord km = aln_coord(ix, c);
            std::string line = me.name;
            snprintf(num, sizeof num, "\t%d\t", c.flag); line += num;
            line += ix.chr[km.chr].name;
            snprintf(num, sizeof num, "\t%lld\t%d\t", (long long)km.pos, mq); line += num;
            line += cigar_of(me.rlen, c);
            int j = c.mate;
            if (j != -1 && other.score > 0 && other.cands[j].score == other.score) {
                Coord ko = aln_coord(ix, other.cands[j]);
                // dist is defined from read1's point of view (:428, :475)
                const Cand &c1 = me_is_first ? c : other.cands[j];
                i64 p1 = me_is_first ? km.pos : ko.pos, p2 = me_is_first ? ko.pos : km.pos;
                const Read &ra = me_is_first ? me : other, &rb = me_is_first ? other : me;
                int dist = (int)(p2 - p1 + (c1.fwd ? rb.rlen : 0 - ra.rlen));
                if (!me_is_first) dist = 0 - dist;
                snprintf(num, sizeof num, "\t=\t%lld\t%d\t", (long long)ko.pos, dist); line += num;
            } else line += "\t*\t0\t0\t";
            line += c.fwd ? me.seq : rseq;
            line += "\t";
            line += fastq ? (c.fwd ? me.qual : rq) : "*";
            snprintf(num, sizeof num, "\tNM:i:%d\tAS:i:%d\tXS:i:%d", me.rlen - c.score, me.score, me.sub); line += num;
            out.push_back(line);
            if (pm.unique) break;
        }
    };
    if (r1.score == 0) unmapped(r1, r2, 0x40); else mapped(r1, r2, true);
    if (r2.score == 0) unmapped(r2, r1, 0x80); else mapped(r2, r1, false);
}

// ---- pair bookkeeping (ReadMapping.cpp:324-402, :479-534) --------------------------------------
struct PairDist { i64 dist, g1, g2; };

// GenCoordinatePair :361-394 with GetPairedAlnCanDist :342-359
static PairDist pair_distance(const std::vector<Cand> &c1, const std::vector<Cand> &c2)
{
    PairDist p; p.dist = 0; p.g1 = p.g2 = 0;
    for (const Cand &c : c1) {
        if (c.score > 0 && c.mate != -1 && c2[c.mate].score > 0) {
            p.g1 = c.frags[0].gPos; p.g2 = c2[c.mate].frags[0].gPos;
            p.dist = p.g2 > p.g1 ? p.g2 - p.g1 : p.g1 - p.g2;
            break;
        }
    }
    if (p.dist != 0) return p;
    std::vector<i64> a, b;
    for (const Cand &c : c1) if (c.score > 0) a.push_back(c.frags[0].gPos);
    for (const Cand &c : c2) if (c.score > 0) b.push_back(c.frags[0].gPos);
    if (a.size() == 1 && b.size() == 1) { p.g1 = a[0]; p.g2 = b[0]; p.dist = p.g2 > p.g1 ? p.g2 - p.g1 : p.g1 - p.g2; }
    else if (a.empty() && !b.empty()) { p.g1 = -1; p.dist = p.g2 = b[0]; }
    else if (!a.empty() && b.empty()) { p.dist = p.g1 = a[0]; p.g2 = -1; }
    else p.dist = 0;
    return p;
}

// ---- alignment profile (AlignmentProfile.cpp) --------------------------------------------------
struct Discord { i64 gPos, dist; };

struct Profile {
    // MappingRecord_t (structure.h:152-163): A,C,G,T,multi_hit are 12-bit fields that saturate
    // at MaxAlleleCount (4095); readCount is a 4-bit field; F1,R2,F2,R1 are plain uint16_t
    std::vector<uint16_t> cnt; // 10 per position: A C G T multi_hit readCount F1 R2 F2 R1
    std::map<i64, std::map<std::string, uint16_t>> ins, del;
    std::map<i64, uint16_t> brk;
    std::vector<Discord> inv, tnl;
    int max_dup = 5, max_clip = 5; // iMaxDuplicate, MaxClipSize (main.cpp:175,:181)
    void init(i64 G) { cnt.assign((size_t)G * 10, 0); }
    uint16_t &at(i64 g, int k) { return cnt[(size_t)g * 10 + k]; }
    void base(i64 g, int k) { uint16_t &v = at(g, k); if (v < 4095) v++; }
};

// one aligned column of a forward-strand character: 'A','C','G','T' count, anything else does not
static inline void count_char(Profile &pf, i64 g, char c)
{
    switch (c) { case 'A': pf.base(g, 0); break; case 'C': pf.base(g, 1); break; case 'G': pf.base(g, 2); break; case 'T': pf.base(g, 3); break; }
}

// walks a gapped fragment's strings from genome position g (AlignmentProfile.cpp:130-166 / :205-241)
static void profile_gapped(Profile &pf, const Frag &f, i64 g)
{
    const int len = (int)f.a1.size();
    for (int j = 0; j < len;) {
        if (f.a2[j] == '-') {
            int e = 1; while (j + e < len && f.a2[j + e] == '-') e++;
            pf.ins[g - 1][f.a1.substr(j, e)]++;
            j += e;
        } else if (f.a1[j] == '-') {
            int e = 1; while (j + e < len && f.a1[j + e] == '-') e++;
            pf.del[g - 1][f.a2.substr(j, e)]++;
            j += e; g += e;
        } else { count_char(pf, g, f.a1[j]); j++; g++; }
    }
}

// UpdateProfile, AlignmentProfile.cpp:41-242
static void update_profile(const Index &ix, Profile &pf, bool first_read, const Read &rd)
{
    for (const Cand &c : rd.cands) {
        if (c.score == 0) continue;
        const std::vector<Frag> &v = c.frags;
        const Frag &a = v.front(), &b = v.back();
        if (a.rLen == 0 && a.gLen == 0) {
            if (a.rPos > 20) pf.brk[a.gPos < ix.G ? a.gPos : ix.G2 - 1 - a.gPos]++; // MinBreakPointSize :4
            if (a.rPos > pf.max_clip) continue;
        }
        if (b.rLen == 0 && b.gLen == 0) {
            if (rd.rlen - b.rPos > 20) pf.brk[b.gPos < ix.G ? b.gPos : ix.G2 - 1 - b.gPos]++;
            if (rd.rlen - b.rPos > pf.max_clip) continue;
        }
        i64 g = c.fwd ? a.gPos : ix.G2 - (a.gPos + a.gLen);
        if (pf.at(g, 5) < pf.max_dup) pf.at(g, 5)++; else continue;
        const int strand = first_read ? (c.fwd ? 6 : 9) : (c.fwd ? 7 : 8); // F1 / R1 / R2 / F2
        for (int i = 0; i < rd.rlen && g + i < ix.G; i++) pf.at(g + i, strand)++; // (the reference runs past the array end at the genome end)
        for (const Frag &f : v) {
            if (c.fwd) {
                i64 gp = f.gPos;
                if (f.simple) { for (int j = 0; j < f.rLen; j++) count_char(pf, gp + j, rd.seq[f.rPos + j]); }
                else if (f.gLen == 0) pf.ins[gp - 1][f.a1]++;
                else if (f.rLen == 0) pf.del[gp - 1][f.a2]++;
                else profile_gapped(pf, f, gp);
            } else {
                if (f.simple) {
                    i64 gp = ix.G2 - 1 - f.gPos;
                    for (int j = 0; j < f.rLen; j++, gp--) {
                        switch (rd.seq[f.rPos + j]) { case 'A': pf.base(gp, 3); break; case 'C': pf.base(gp, 2); break; case 'G': pf.base(gp, 1); break; case 'T': pf.base(gp, 0); break; }
                    }
                } else if (f.gLen == 0) pf.ins[ix.G2 - f.gPos - 1][f.a1]++;
                else if (f.rLen == 0) pf.del[ix.G2 - f.gPos - f.gLen - 1][f.a2]++;
                else profile_gapped(pf, f, ix.G2 - (f.gPos + f.gLen));
            }
        }
    }
}

// UpdateMultiHitCount, AlignmentProfile.cpp:244-271
static void update_multi_hit(const Index &ix, Profile &pf, const Read &rd)
{
    for (const Cand &c : rd.cands) {
        if (c.score <= 0) continue;
        const Frag &a = c.frags.front(), &b = c.frags.back();
        i64 g0 = c.fwd ? a.gPos : ix.G2 - (a.gPos + a.gLen);
        i64 g1 = c.fwd ? b.gPos + b.gLen : ix.G2 - b.gPos;
        for (i64 g = g0; g < g1; g++) pf.base(g, 4);
    }
}

// ---- input (GetData.cpp) ---------------------------------------------------------------------
// IdentifyHeaderBegPos :3-10 / IdentifyHeaderEndPos :12-20
static std::string trim_header(const std::string &line)
{
    int len = (int)line.size(), p1 = len - 1, p2;
    for (int i = 1; i < len; i++) if (line[i] != '>' && line[i] != '@') { p1 = i; break; }
    int lim = len > 100 ? 100 : len;
    p2 = lim - 1;
    for (int i = 1; i < lim; i++) if (line[i] == ' ' || line[i] == '/' || !isprint((unsigned char)line[i])) { p2 = i; break; }
    return p2 > p1 ? line.substr(p1, p2 - p1) : std::string();
}

struct Reader {
    FILE *f = nullptr;
    bool fastq = true;
    char *buf = nullptr;
    size_t cap = 0;
    bool open(const char *path)
    {
        f = fopen(path, "r");
        if (!f) return false;
        int c = fgetc(f);
        fastq = (c == '@'); // CheckReadFormat :22-31
        ungetc(c, f);
        return true;
    }
    bool line(std::string &s)
    {
        ssize_t n = getline(&buf, &cap, f);
        if (n < 0) return false;
        s.assign(buf, n);
        return true;
    }
    // GetNextEntry :32-83 (plain files; FASTQ 4-line records, FASTA possibly multi-line)
    bool next(Read &r)
    {
        std::string l;
        r = Read();
        if (!line(l)) return false;
        r.name = trim_header(l);
        if (fastq) {
            if (!line(l)) return false;
            r.seq = l; r.rlen = (int)l.size() - 1; // the last byte (newline) is dropped
            std::string plus, q;
            line(plus); line(q);
            q.resize(l.size(), '\0');
            r.seq.resize(r.rlen); r.qual = q.substr(0, r.rlen);
        } else {
            for (;;) {
                long at = ftell(f);
                if (!line(l)) break;
                if (l[0] == '>') { fseek(f, at, SEEK_SET); break; }
                l.resize(l.size() - 1);
                r.seq += l;
            }
            r.rlen = (int)r.seq.size();
        }
        return r.rlen > 0;
    }
    void close() { if (f) fclose(f); f = nullptr; free(buf); buf = nullptr; }
};

// ReverseOrientation, tools.cpp:45-55
static void flip_read(Read &r)
{
    revcomp_inplace(r.seq);
    std::reverse(r.qual.begin(), r.qual.end());
}

// ---- the chunk loop (ReadMapping, ReadMapping.cpp:416-646) -----------------------------------
struct Shared {
    const Index *ix;
    Params pm;
    Reader in1, in2;
    bool paired = false, fastq = true; // paired: bPairEnd (two files, or -p)
    bool two_files = false;            // bSepLibrary
    FILE *sam = nullptr;
    std::mutex in_lock, out_lock;
    u32 avg_dist = 1000; // ReadMapping.cpp:20
    i64 n_reads = 0, n_mapped = 0, n_paired = 0, dist_sum = 0, len_sum = 0;
    Counters ct;
    Profile *pf = nullptr; // -vcf bookkeeping (ReadMapping.cpp:562-573); single thread only
};

static void seed_and_cluster(const Index &ix, const Params &pm, Read &r, Counters &ct)
{
    std::vector<Frag> seeds = find_seeds(ix, r.seq, ct);
    r.cands = cluster_seeds(ix, pm, r.rlen, seeds);
    r.best = -1; r.score = r.sub = 0;
    for (Cand &c : r.cands) c.mate = -1; // ResetPairedIdx :69-72
}

static void worker(Shared *sh)
{
    const Index &ix = *sh->ix;
    std::vector<Read> chunk;
    std::vector<std::string> lines;
    Counters ct;
    Discord last_disc = {0, 0}; // the reference's DiscordPair variable lives across pairs
    for (;;) {
        chunk.clear();
        {
            std::lock_guard<std::mutex> g(sh->in_lock); // GetNextChunk, GetData.cpp:85-99
            Read a, b;
            while ((int)chunk.size() < kChunk) {
                if (!sh->in1.next(a)) break;
                chunk.push_back(a);
                if (sh->two_files) { sh->in2.next(b); chunk.push_back(b); }
                else { if (!sh->in1.next(b)) break; chunk.push_back(b); }
            }
        }
        int n = (int)chunk.size();
        if (n == 0) break;
        int mapped = 0, pairs = 0;
        i64 dsum = 0, lsum = 0;
        lines.clear();
        if (sh->paired && n % 2 == 0) {
            for (int i = 0; i < n; i += 2) {
                Read &r1 = chunk[i], &r2 = chunk[i + 1];
                seed_and_cluster(ix, sh->pm, r1, ct);
                flip_read(r2);
                seed_and_cluster(ix, sh->pm, r2, ct);
                u32 avg = sh->avg_dist; // read without the lock, as the reference does (:462)
                int np = pair_by_distance((int)(avg * 1.5), r1.cands, r2.cands);
                if (np == 0) np = rescue_mate(ix, (u32)(int)(avg * 1.5), r1, r2);
                if (np == 0) { keep_top_scores(r1.cands); keep_top_scores(r2.cands); }
                else mask_unpaired(r1.cands, r2.cands);
                if (extend_read(ix, sh->pm, r1, ct)) mapped++;
                if (extend_read(ix, sh->pm, r2, ct)) mapped++;
                PairDist pd = pair_distance(r1.cands, r2.cands);
                if (pd.dist != 0 && pd.g1 != -1 && pd.g2 != -1) {
                    bool inv = (pd.g1 < ix.G && pd.g2 >= ix.G) || (pd.g1 >= ix.G && pd.g2 < ix.G);
                    if (!inv && pd.dist <= 1000) { pairs++; dsum += pd.dist; lsum += r1.rlen + r2.rlen; } // MinTranslocationSize :9, :527-531
                    if (sh->pf) { // discordant-site lists, ReadMapping.cpp:486-521 (the second branch pushes unconditionally)
                        Profile &pf = *sh->pf;
                        if (pd.g1 < ix.G && pd.g2 >= ix.G) {
                            i64 d = ix.G2 - pd.g1 - pd.g2; if (d < 0) d = -d;
                            last_disc.dist = d; // (.dist always; .gPos only inside the range, :492-496)
                            if (d > 1000 && d < 10000000) { last_disc.gPos = pd.g1; pf.inv.push_back(last_disc); }
                        } else if (pd.g1 >= ix.G && pd.g2 < ix.G) {
                            i64 d = ix.G2 - pd.g1 - pd.g2; if (d < 0) d = -d;
                            last_disc.dist = d;
                            if (d > 1000 && d < 10000000) last_disc.gPos = pd.g2;
                            pf.inv.push_back(last_disc);
                        } else if (pd.dist > 1000) {
                            last_disc.dist = pd.dist;
                            if (pd.g1 < ix.G && pd.g2 < ix.G) { pf.tnl.push_back({pd.g1, pd.dist}); pf.tnl.push_back({pd.g2, pd.dist}); last_disc.gPos = pd.g2; }
                            else if (pd.g1 >= ix.G && pd.g2 >= ix.G) { pf.tnl.push_back({ix.G2 - pd.g1, pd.dist}); pf.tnl.push_back({ix.G2 - pd.g2, pd.dist}); last_disc.gPos = ix.G2 - pd.g2; }
                        }
                    }
                }
            }
            if (sh->sam) for (int i = 0; i < n; i += 2) sam_paired(ix, sh->pm, sh->fastq, chunk[i], chunk[i + 1], lines);
        } else {
            for (int i = 0; i < n; i++) {
                seed_and_cluster(ix, sh->pm, chunk[i], ct);
                keep_top_scores(chunk[i].cands);
                if (extend_read(ix, sh->pm, chunk[i], ct)) mapped++;
            }
            if (sh->sam) for (int i = 0; i < n; i++) sam_single(ix, sh->pm, sh->fastq, chunk[i], lines);
        }
        std::lock_guard<std::mutex> g(sh->out_lock);
        sh->n_reads += n; sh->n_mapped += mapped; sh->n_paired += pairs; sh->dist_sum += dsum; sh->len_sum += lsum;
        if (sh->n_paired > 1000) sh->avg_dist = (u32)(int)(1. * sh->dist_sum / sh->n_paired + .5); // :539
        if (sh->sam) for (auto &l : lines) { fputs(l.c_str(), sh->sam); fputc('\n', sh->sam); }
        if (sh->pf) { // ReadMapping.cpp:562-573 / :610-620
            const bool pe = sh->paired && n % 2 == 0;
            for (int i = 0; i < n; i++) {
                if (chunk[i].score == 0) continue;
                int live = 0;
                for (const Cand &c : chunk[i].cands) if (c.score > 0) live++;
                if (live == 1) update_profile(ix, *sh->pf, pe ? (i % 2 == 0) : true, chunk[i]);
                else update_multi_hit(ix, *sh->pf, chunk[i]);
            }
        }
    }
    std::lock_guard<std::mutex> g(sh->out_lock);
    sh->ct.ext_steps += ct.ext_steps; sh->ct.sa_hits += ct.sa_hits; sh->ct.lf_steps += ct.lf_steps;
    sh->ct.dp_calls += ct.dp_calls; sh->ct.dp_cells += ct.dp_cells;
}


// ---- variant calling (VariantCalling.cpp) -------------------------------------------------------
struct VcfOpts {
    int ploidy = 2, min_ad = 5, min_cnv = 50, min_gap = 50, frag_size = 500; // main.cpp:157-187
    bool filter = false, gvcf = false, mono = false, somatic = false;
    float freq_thr = 0.2f;
    std::string sample = "unknown", ref_name, cmdline;
};

struct Variant { // Variant_t, structure.h:185-195; one object is reused by the scan, stale fields included
    uint16_t DP = 0; i64 gPos = 0; std::string alt; uint16_t AD_ref = 0, AD_alt = 0; uint8_t geno = 0, qscore = 0, type = 0;
};
enum { vSUB = 0, vINS = 1, vDEL = 2, vINV = 3, vTNL = 4, vCNV = 5, vUMR = 6, vNOR = 10, vMON = 11 };

struct Caller {
    const Index &ix; Profile &pf; const VcfOpts &o;
    u32 avg_rlen; int frag_size;
    std::vector<int> depth; // BlockDepthArr
    std::vector<Variant> vars;
    Caller(const Index &i, Profile &p, const VcfOpts &oo, u32 rl, int fs) : ix(i), pf(p), o(oo), avg_rlen(rl), frag_size(fs) {}

    int cov(i64 g) { return g >= 0 && g < ix.G ? pf.at(g, 0) + pf.at(g, 1) + pf.at(g, 2) + pf.at(g, 3) : 0; } // GetProfileColumnSize, tools.cpp:166-169
    static uint8_t as_u8(double v) { return (uint8_t)(int)v; } // `uint8_t q = (int)(expr)` on x86-64
    static bool by_pos(const Variant &a, const Variant &b) { return a.gPos == b.gPos ? a.type < b.type : a.gPos < b.gPos; } // CompByVarPos :50-54

    // GetAreaIndFrequency :63-94
    static int area_freq(i64 gPos, std::map<i64, std::map<std::string, uint16_t>> &m, std::string &str)
    {
        i64 max_pos = 0; int freq = 0, max_freq = 0;
        str.clear();
        for (auto a = m.lower_bound(gPos - 5), b = m.upper_bound(gPos + 5); a != b; ++a) {
            if (std::llabs(a->first - gPos) > 5) continue;
            for (auto &e : a->second) {
                freq += e.second;
                if (max_freq < e.second) { str = e.first; max_freq = e.second; max_pos = a->first; }
                else if (max_freq == e.second && e.first.length() > str.length()) { str = e.first; max_pos = a->first; }
            }
        }
        return gPos == max_pos ? freq : 0;
    }

    // DetermineGenotype :528-547
    uint8_t genotype(int cv, int alt_reads, int alt_num)
    {
        if (o.ploidy == 1) return alt_reads < (int)(cv * 0.5) ? 1 : 2;
        if (o.ploidy == 2) {
            if (alt_num == 0) return 3;
            if (alt_num == 1) return alt_reads < (int)(cv * 0.5) ? 4 : 5;
            if (alt_num == 2) return 6;
        }
        return 0;
    }

    uint16_t ref_count(int base, i64 g) { return base < 4 ? pf.at(g, base) : 0; } // GetRefCount :516-526

    void block_depth() // CalBlockReadDepth :105-121, VariantCalling() :708-711
    {
        i64 nb = ix.G / 100; if (nb * 100 < ix.G) nb++;
        depth.assign((size_t)nb, 0);
        for (i64 b = 0; b < nb; b++) {
            i64 e = std::min<i64>(b * 100 + 100, ix.G); int sum = 0;
            for (i64 g = b * 100; g < e; g++) sum += cov(g);
            if (sum > 0) depth[b] = sum / 100;
        }
    }

    void scan() // IdentifyVariants :549-680 (one thread: VariantCalling() sets iThreadNum = 1, :717)
    {
        Variant v; std::string ins_str, del_str;
        std::vector<std::pair<char, int>> vec;
        int gap = 0, dup = 0;
        for (i64 g = 0; g < ix.G; g++) {
            const int cv = cov(g);
            bool normal = true;
            const int rb = g_nt4[(uint8_t)ix.ref[g]];
            int cov_thr = depth[g / 100] >> 1; if (cov_thr < o.min_ad) cov_thr = o.min_ad;
            if (o.somatic && cov_thr > o.min_ad) cov_thr = o.min_ad;
            int ins_thr = (int)(cov_thr * 0.25); if (ins_thr < o.min_ad) ins_thr = o.min_ad;
            int del_thr = (int)(cov_thr * 0.35); if (del_thr < o.min_ad) del_thr = o.min_ad;
            const int ins_freq = area_freq(g, pf.ins, ins_str), del_freq = area_freq(g, pf.del, del_str);
            if (ins_freq >= ins_thr) {
                v.gPos = g; v.type = vINS; v.DP = (uint16_t)depth[g / 100]; v.AD_alt = (uint16_t)ins_freq;
                if (v.DP < v.AD_alt) v.DP = v.AD_alt;
                v.alt = ins_str; v.AD_ref = v.DP - v.AD_alt; v.geno = genotype(v.DP, v.AD_alt, 1);
                v.qscore = cv == 0 ? 0 : as_u8(100.0 * v.AD_alt / cv); // x/0 -> inf -> INT_MIN -> low byte 0
                normal = false; vars.push_back(v);
            }
            if (del_freq >= del_thr) {
                v.gPos = g; v.type = vDEL; v.DP = (uint16_t)depth[g / 100]; v.AD_alt = (uint16_t)del_freq;
                if (v.DP < v.AD_alt) v.DP = v.AD_alt;
                v.alt = del_str; v.AD_ref = v.DP - v.AD_alt; v.geno = genotype(v.DP, v.AD_alt, 1);
                v.qscore = cv == 0 ? 0 : as_u8(100.0 * v.AD_alt / cv);
                normal = false; vars.push_back(v);
            }
            if (cv >= cov_thr) {
                vec.clear();
                int freq_thr = (int)ceil(cv * (o.somatic ? 0.01 : (double)o.freq_thr)); // float FrequencyThr widened by ?: (:593)
                if (freq_thr < o.min_ad) freq_thr = o.min_ad;
                for (int k = 0; k < 4; k++) if (rb != k && (int)pf.at(g, k) >= freq_thr) vec.push_back(std::make_pair("ACGT"[k], (int)pf.at(g, k)));
                v.AD_ref = ref_count(rb, g);
                if (vec.size() == 1) {
                    v.gPos = g; v.type = vSUB; v.DP = (uint16_t)cv; v.AD_alt = (uint16_t)vec[0].second;
                    if ((v.geno = genotype(cv, v.AD_alt, 1)) != 0) {
                        v.alt = std::string(1, vec[0].first);
                        v.qscore = o.somatic ? as_u8(35.0 * v.AD_alt / (cv * 0.05)) : as_u8(35.0 * v.AD_alt / cv);
                        normal = false; vars.push_back(v);
                    }
                } else if (vec.size() == 2 && vec[0].second + vec[1].second >= (int)(cv * 0.5)) { // CheckDiploidFrequency :123-128
                    v.gPos = g; v.type = vSUB; v.DP = (uint16_t)cv; v.AD_alt = (uint16_t)(vec[0].second + vec[1].second);
                    if ((v.geno = genotype(cv, v.AD_alt, 2)) != 0) {
                        v.alt = std::string(1, vec[0].first) + "," + std::string(1, vec[1].first);
                        v.qscore = o.somatic ? as_u8(35.0 * v.AD_alt / (cv * 0.05)) : as_u8(35.0 * v.AD_alt / cv);
                        normal = false; vars.push_back(v);
                    }
                }
            }
            const int multi = pf.at(g, 4);
            if (cv == 0 && multi == 0) { normal = false; gap++; }
            else if (gap > 0) {
                if (gap >= o.min_gap) { v.type = vUMR; v.gPos = g - gap; v.DP = (uint16_t)gap; vars.push_back(v); }
                gap = 0;
            }
            if (cv == 0 && multi > 0) { normal = false; dup++; }
            else if (dup > 0) {
                if (dup > o.min_cnv) { v.type = vCNV; v.gPos = g - dup; v.DP = (uint16_t)dup; vars.push_back(v); }
                dup = 0;
            }
            if (o.gvcf && normal && cv > 0) {
                if (vars.empty() || vars.back().type != vNOR) { v.qscore = 0; v.gPos = g; v.type = vNOR; v.DP = v.AD_alt = (uint16_t)cv; v.alt.clear(); vars.push_back(v); }
                else if (vars.back().AD_alt > cv) vars.back().AD_alt = (uint16_t)cv;
            }
            if (o.mono && normal && cv > 0) {
                v.qscore = 0; v.gPos = g; v.type = vMON; v.DP = (uint16_t)cv; v.geno = genotype(cv, 0, 0); v.alt.clear();
                v.AD_ref = ref_count(rb, g);
                vars.push_back(v);
            }
        }
        std::stable_sort(vars.begin(), vars.end(), by_pos); // no two entries share (gPos, type), so any sort agrees with :672
    }

    void drop_consecutive_nor() // RemoveConsecutiveGenomicVariant :682-694 (its iterator walk skips one comparison after an erase)
    {
        if (vars.size() < 2) return; // the reference would read past the end here
        size_t i = 0, n = 1;
        while (n < vars.size()) {
            if (vars[i].type == vNOR && vars[n].type == vNOR) {
                vars.erase(vars.begin() + n); i = n; n = i + 1;
                if (i >= vars.size()) break; // (undefined in the reference: erased the last element)
            }
            i++; n++;
        }
    }

    int region_cov(i64 b, i64 e) // CalRegionCov :197-208
    {
        if (b < 0) b = 0;
        if (e > ix.G) e = ix.G - 1;
        if (e < b) return 0;
        i64 c = 0;
        for (i64 g = b; g <= e; g++) c += cov(g); // (e == G reads one record past the array in the reference; counted as 0 here)
        return (int)(c / (e - b + 1));
    }

    std::vector<i64> bp_cands; // IdentifyBreakPointCandidates :173-195
    void breakpoints()
    {
        pf.brk.insert(std::make_pair(ix.G2, (uint16_t)0));
        u32 total = 0; std::pair<i64, uint16_t> p(0, 0);
        for (auto &e : pf.brk) {
            if (e.first - p.first > (i64)avg_rlen) {
                if (total >= 3) bp_cands.push_back(p.first);
                p.first = e.first; total = p.second = e.second;
            } else {
                total += e.second;
                if (p.second < e.second) { p.first = e.first; p.second = e.second; }
            }
        }
    }

    // longest run of distance classes (dist / 1000) that differ by at most one between neighbours
    static u32 run_score(std::vector<i64> &vec, i64 sentinel)
    {
        std::sort(vec.begin(), vec.end()); vec.push_back(sentinel);
        u32 best = 0, score = 1;
        for (size_t j = 1; j < vec.size(); j++) {
            if (vec[j] - vec[j - 1] > 1) { if (score > best) best = score; score = 1; }
            else score++;
        }
        return best;
    }

    // IdentifyInversions :276-340 / IdentifyTranslocations :210-274 (the same procedure on the two site lists)
    void discordant(const std::vector<Discord> &sites, int type)
    {
        auto lower = [&](i64 g) { return std::lower_bound(sites.begin(), sites.end(), g, [](const Discord &d, i64 x) { return d.gPos < x; }); };
        auto upper = [&](i64 g) { return std::upper_bound(sites.begin(), sites.end(), g, [](i64 x, const Discord &d) { return x < d.gPos; }); };
        std::vector<Variant> found;
        std::vector<i64> vec;
        for (i64 g : bp_cands) {
            const u32 lcov = (u32)region_cov(g - frag_size, g - (avg_rlen >> 1));
            const u32 cov_thr = (u32)(depth[(int)(g / 100)] >> 1);
            auto i1 = lower(g - frag_size), i2 = lower(g - (i64)(avg_rlen >> 1));
            if (i1 == sites.end() || i2 == sites.end()) continue;
            vec.clear(); for (; i1 != i2; ++i1) vec.push_back(i1->dist / 1000);
            const u32 ls = run_score(vec, ix.G2);
            if (ls < cov_thr || ls < (u32)(int)(lcov * 0.5)) continue;
            const u32 rcov = (u32)region_cov(g, g + frag_size);
            i1 = upper(g); i2 = lower(g + frag_size);
            if (i1 == sites.end() || i2 == sites.end()) continue;
            vec.clear(); for (; i1 != i2; ++i1) vec.push_back(i1->dist / 1000);
            const u32 rs = run_score(vec, ix.G2);
            if (rs < cov_thr || rs < (u32)(int)(rcov * 0.5)) continue;
            if (ls > 0 && rs > 0) { Variant v; v.gPos = g; v.type = (uint8_t)type; v.DP = (uint16_t)cov(g); v.AD_alt = (uint16_t)std::max(ls, rs); found.push_back(v); }
        }
        if (!found.empty()) { // inplace_merge :273 / :339
            size_t mid = vars.size();
            vars.insert(vars.end(), found.begin(), found.end());
            std::inplace_merge(vars.begin(), vars.begin() + mid, vars.end(), by_pos);
        }
    }

    bool nearby(int i, int dist) // CheckNearbyVariant :342-358
    {
        const int n = (int)vars.size();
        if (n < 2) return false; // (the reference reads VariantVec[1] of a one-element vector)
        if (i == 0) return vars[1].gPos - vars[0].gPos <= dist;
        if (i == n - 1) return vars[i].gPos - vars[i - 1].gPos <= dist;
        return vars[i + 1].gPos - vars[i].gPos <= dist || vars[i].gPos - vars[i - 1].gPos <= dist;
    }

    bool bad_haplotype(int i, int dist) // CheckBadHaplotype :360-388
    {
        bool r = false; const int n = (int)vars.size();
        for (int j = i + 1; j < n; j++) {
            if (vars[j].gPos - vars[i].gPos > dist) break;
            if (vars[j].type == 0) {
                int diff = std::abs((int)vars[i].AD_alt - (int)vars[j].AD_alt);
                if (diff > 5 && (vars[i].AD_alt > vars[j].AD_alt ? vars[i].AD_alt >> 2 : vars[j].AD_alt >> 2)) r = true;
                break;
            }
        }
        for (int j = i - 1; j >= 0; j--) {
            if (vars[i].gPos - vars[j].gPos > dist) break;
            if (vars[j].type == 0) {
                int diff = std::abs((int)vars[i].AD_alt - (int)vars[j].AD_alt);
                if (diff > 10 && (vars[i].AD_alt > vars[j].AD_alt ? (int)(vars[i].AD_alt * 0.33) : (int)(vars[j].AD_alt * 0.33))) r = true;
                break;
            }
        }
        return r;
    }

    std::string filter_of(int i) // DetermineFileter :404-427
    {
        const Variant &v = vars[i]; std::string f;
        if (v.qscore < 10) f += "q10;";
        else if (v.type == vSUB && v.AD_alt < 10 && nearby(i, 10)) f += "q10;";
        else if ((v.type == vINS || v.type == vDEL) && v.AD_alt < 5 && nearby(i, 10)) f += "q10;";
        if (o.filter) {
            if ((int)pf.at(v.gPos, 4) > (int)(cov(v.gPos) * 0.05)) f += "str_contraction;";
            if (bad_haplotype(i, 100)) f += "bad_haplotype;";
        }
        if (f.empty()) return "PASS";
        f.resize(f.size() - 1);
        return f;
    }

    bool write(const char *path) // ShowMetaInfo :140-171, GenVariantCallingFile :429-500
    {
        static const char *GT[] = {"*", "0", "1", "0/0", "0/1", "1/1", "1/2"};
        FILE *f = fopen(path, "w");
        if (!f) return false;
        fprintf(f, "##fileformat=VCFv4.2\n##reference=%s\n##source=MapCaller 0.9.9.41\n##command_line=\"%s\"\n", o.ref_name.c_str(), o.cmdline.c_str());
        fprintf(f, "##ALT=<ID=NON_REF,Description=\"Represents any possible alternative allele at this location\">\n");
        fprintf(f, "##INFO=<ID=RC,Number=1,Type=Integer,Description=\"Number of reads with start coordinate at this position.\">\n");
        fprintf(f, "##INFO=<ID=NTFREQ,Number=4,Type=Integer,Description=\"base depth\">\n");
        fprintf(f, "##INFO=<ID=END,Number=1,Type=Integer,Description=\"Last position(inclusive) of the reported block\">\n");
        fprintf(f, "##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Read depth\">\n");
        fprintf(f, "##INFO=<ID=TYPE,Number=A,Type=String,Description=\"The type of allele, either snv, ins, del, or BP(breakpoint).\">\n");
        fprintf(f, "##FORMAT=<ID=AD,Number=R,Type=Integer,Description=\"Allelic depths for the ref and alt alleles in the order listed\">\n");
        fprintf(f, "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Approximate read depth\">\n");
        fprintf(f, "##FORMAT=<ID=AF,Number=A,Type=Float,Description=\"Allele fractions of alternate alleles\">\n");
        fprintf(f, "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n");
        fprintf(f, "##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"Normalized, Phred - scaled likelihoods for genotypes as defined in the VCF specification\">\n");
        if (o.gvcf) fprintf(f, "##FORMAT=<ID=MIN_DP,Number=1,Type=Integer,Description=\"Minimum depth in gVCF output block.\">\n");
        fprintf(f, "##FORMAT=<ID=F1R2,Number=R,Type=Integer,Description=\"Count of reads in F1R2 pair orientation supporting each allele\">\n");
        fprintf(f, "##FORMAT=<ID=F2R1,Number=R,Type=Integer,Description=\"Count of reads in F2R1 pair orientation supporting each allele\">\n");
        fprintf(f, "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype Quality\">\n");
        fprintf(f, "##FILTER=<ID=PASS,Description=\"All filters passed\">\n");
        fprintf(f, "##FILTER=<ID=REF,Description=\"Genotyping model thinks this site is reference.\">\n");
        fprintf(f, "##FILTER=<ID=BreakPoint,Description=\"It is predicted as a breakpoint\">\n");
        fprintf(f, "##FILTER=<ID=DUP,Description=\"Duplicated regions(>=%dbp).\">\n", o.min_cnv);
        fprintf(f, "##FILTER=<ID=Gaps,Description=\"Region without any read alignment(>=%dbp).\">\n", o.min_gap);
        fprintf(f, "##FILTER=<ID=q10,Description=\"Confidence score below 10\">\n");
        if (o.filter) fprintf(f, "##FILTER=<ID=bad_haplotype,Description=\"Variants with variable frequencies on same haplotype\">\n");
        if (o.filter) fprintf(f, "##FILTER=<ID=str_contraction,Description=\"Variant appears in repetitive region\">\n");
        for (const Chrom &c : ix.chr) fprintf(f, "##contig=<ID=%s,length=%d>\n", c.name.c_str(), c.len);
        fprintf(f, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s\n", o.sample.c_str());
        const int n = (int)vars.size();
        for (int i = 0; i < n; i++) {
            const Variant &v = vars[i];
            const i64 g = v.gPos;
            const Coord co = to_coord(ix, g);
            const char *chr = ix.chr[co.chr].name.c_str();
            const char rc = ix.ref[g];
            const std::string flt = v.type < 3 ? filter_of(i) : ".";
            const float af = (float)(1.0 * v.AD_alt / v.DP);
            const int F1 = pf.at(g, 6), R2 = pf.at(g, 7), F2 = pf.at(g, 8), R1 = pf.at(g, 9), RC = pf.at(g, 5);
            if (v.type == vSUB)
                fprintf(f, "%s\t%d\t.\t%c\t%s\t%d\t%s\tRC=%d;NTFREQ=%d,%d,%d,%d;TYPE=snv\tGT:GQ:DP:AD:AF:F1R2:F2R1\t%s:%d:%d:%d,%d:%.2f:%d,%d:%d,%d\n", chr, (int)co.pos, rc, v.alt.c_str(), v.qscore, flt.c_str(),
                        RC, (int)pf.at(g, 0), (int)pf.at(g, 1), (int)pf.at(g, 2), (int)pf.at(g, 3), GT[v.geno], v.qscore, v.DP, v.AD_ref, v.AD_alt, af, F1, R2, F2, R1);
            else if (v.type == vINS) {
                if (v.alt.length() > 5) continue;
                fprintf(f, "%s\t%d\t.\t%c\t%c%s\t%d\t%s\tRC=%d;TYPE=ins\tGT:GQ:DP:AD:AF:F1R2:F2R1\t%s:%d:%d:%d,%d:%.2f:%d,%d:%d,%d\n", chr, (int)co.pos, rc, rc, v.alt.c_str(), v.qscore, flt.c_str(),
                        RC, GT[v.geno], v.qscore, v.DP, v.AD_ref, v.AD_alt, af, F1, R2, F2, R1);
            } else if (v.type == vDEL) {
                if (v.alt.length() > 5) continue;
                fprintf(f, "%s\t%d\t.\t%c%s\t%c\t%d\t%s\tRC=%d;TYPE=del\tGT:GQ:DP:AD:AF:F1R2:F2R1\t%s:%d:%d:%d,%d:%.2f:%d,%d:%d,%d\n", chr, (int)co.pos, rc, v.alt.c_str(), rc, v.qscore, flt.c_str(),
                        RC, GT[v.geno], v.qscore, v.DP, v.AD_ref, v.AD_alt, af, F1, R2, F2, R1);
            } else if (v.type == vTNL) fprintf(f, "%s\t%d\t.\t%c\t<TNL>\t30\tBreakPoint\tTYPE=BP\tGT:GQ:DP:AD\t.:.:0:.\n", chr, (int)co.pos, rc);
            else if (v.type == vINV) fprintf(f, "%s\t%d\t.\t%c\t<INV>\t30\tBreakPoint\tTYPE=BP\tGT:GQ:DP:AD\t.:.:0:.\n", chr, (int)co.pos, rc);
            else if (v.type == vCNV) { if (v.DP >= o.min_cnv) fprintf(f, "%s\t%d\t.\t%c\t<*>\t0\tDUP\tEND=%d\tGT:GQ:DP:AD\t.:.:0:.\n", chr, (int)co.pos, rc, (int)(co.pos + v.DP - 1)); }
            else if (v.type == vUMR) { if (v.DP >= o.min_gap) fprintf(f, "%s\t%d\t.\t%c\t<*>\t0\tGaps\tEND=%d\tGT:GQ:DP:AD\t.:.:0:.\n", chr, (int)co.pos, rc, (int)(co.pos + v.DP - 1)); }
            else if (v.type == vNOR) {
                i64 ge = ix.chr[co.chr].fwd_off + ix.chr[co.chr].len - 1;
                if (i + 1 < n && vars[i + 1].gPos < ge) ge = vars[i + 1].gPos - 1;
                fprintf(f, "%s\t%d\t.\t%c\t<*>\t0\tREF\tEND=%d;DP=%d;MIN_DP=%d\tGT:GQ:DP:AD\t.:.:0:.\n", chr, (int)co.pos, rc, (int)to_coord(ix, ge).pos, v.DP, v.AD_alt);
            } else if (v.type == vMON)
                fprintf(f, "%s\t%d\t.\t%c\t.\t0\tREF\tDP=%d;RC=%d;NTFREQ=%d,%d,%d,%d\tGT:F1R2:F2R1\t%s:%d,%d:%d,%d\n", chr, (int)co.pos, rc, v.DP, RC,
                        (int)pf.at(g, 0), (int)pf.at(g, 1), (int)pf.at(g, 2), (int)pf.at(g, 3), GT[v.geno], F1, R2, F2, R1);
        }
        fclose(f);
        return true;
    }

    bool run(const char *path) // VariantCalling() :696-740
    {
        block_depth();
        scan();
        if (o.gvcf) drop_consecutive_nor();
        breakpoints();
        if (!bp_cands.empty() && !pf.inv.empty()) discordant(pf.inv, vINV);
        if (!bp_cands.empty() && !pf.tnl.empty()) discordant(pf.tnl, vTNL);
        return write(path);
    }
};

} // namespace

// ---------------------------------------------------------------------------------------------
// C surface
// ---------------------------------------------------------------------------------------------
extern "C" {

mcxo_index *mcxo_index_load(const char *prefix) { return load_index(prefix); }
void mcxo_index_free(mcxo_index *ix) { delete ix; }
int64_t mcxo_genome_size(const mcxo_index *ix) { return ix->G; }

int mcxo_bwt_search(const mcxo_index *ix, const uint8_t *seq, int start, int stop, int *len, int *freq, uint64_t *loc)
{
    SeedResult s = fm_search(*ix, seq, start, stop);
    *len = s.len; *freq = s.freq;
    for (int i = 0; i < s.freq; i++) loc[i] = sa_lookup(*ix, s.x0 + i);
    return 0;
}

int mcxo_bwt_search_iv(const mcxo_index *ix, const uint8_t *seq, int start, int stop, int *len, uint64_t *x0, uint64_t *x2)
{
    SeedResult s = fm_search(*ix, seq, start, stop);
    *len = s.len; *x0 = s.x0; *x2 = s.x2;
    return 0;
}

uint64_t mcxo_bwt_sa(const mcxo_index *ix, uint64_t k) { return sa_lookup(*ix, k); }
void mcxo_occ4(const mcxo_index *ix, uint64_t k, uint64_t cnt[4]) { occ4(*ix, k, cnt); }

static int put_aln(const std::string &a, const std::string &b, char *o1, char *o2, int cap)
{
    if ((int)a.size() >= cap || (int)b.size() >= cap) return -1;
    memcpy(o1, a.c_str(), a.size() + 1);
    memcpy(o2, b.c_str(), b.size() + 1);
    return a.size() == b.size() ? (int)a.size() : -2;
}

int mcxo_ksw2(const char *s1, int m, const char *s2, int n, char *o1, char *o2, int cap)
{
    std::string a(s1, m), b(s2, n);
    ksw2_align(a, b);
    return put_aln(a, b, o1, o2, cap);
}

int mcxo_nw(const char *s1, int m, const char *s2, int n, char *o1, char *o2, int cap)
{
    std::string a(s1, m), b(s2, n);
    nw_align(a, b);
    return put_aln(a, b, o1, o2, cap);
}

int mcxo_ksw2_extz(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int *score, char *ops, int cap)
{
    std::string c = ksw2_ops(q, qlen, t, tlen, score);
    if ((int)c.size() >= cap) return -1;
    memcpy(ops, c.c_str(), c.size() + 1);
    return (int)c.size();
}

static thread_local bool g_interleaved = false;
static int64_t map_files_impl(const mcxo_index *ix, const char *fq1, const char *fq2, int alg, const char *sam_path,
                              int threads, int64_t *stats, Profile *pf, int64_t *pair_stats = nullptr);

int64_t mcxo_map_files(const mcxo_index *ix, const char *fq1, const char *fq2, int alg, const char *sam_path,
                       int threads, int64_t *stats)
{
    return map_files_impl(ix, fq1, fq2, alg, sam_path, threads, stats, nullptr);
}

// run totals VariantCalling() needs from Mapping(): out = {iTotalPairedNum, TotalPairedDistance, ReadLengthSum}
int64_t mcxo_pair_totals(const mcxo_index *ix, const char *fq1, const char *fq2, int alg, int64_t out[3])
{
    return map_files_impl(ix, fq1, fq2, alg, nullptr, 1, nullptr, nullptr, out);
}

// MapCaller -p: one file, mates alternate
int64_t mcxo_map_files_interleaved(const mcxo_index *ix, const char *fq, int alg, const char *sam_path, int64_t *stats)
{
    g_interleaved = true;
    const int64_t n = map_files_impl(ix, fq, nullptr, alg, sam_path, 1, stats, nullptr);
    g_interleaved = false;
    return n;
}

// As `MapCaller ... -vcf` would leave them after Mapping(): writes <out>.prof (10 x u16 per
// position: A C G T multi_hit readCount F1 R2 F2 R1) and <out>.maps (insert / delete / breakpoint
// maps and the inversion / translocation site lists), the format of oracle/_ref/mcref_tool 'P'.
int64_t mcxo_map_files_profile(const mcxo_index *ix, const char *fq1, const char *fq2, int alg, const char *out_prefix)
{
    Profile pf;
    pf.init(ix->G);
    int64_t n = map_files_impl(ix, fq1, fq2, alg, nullptr, 1, nullptr, &pf);
    if (n < 0) return n;
    std::string p = std::string(out_prefix) + ".prof";
    FILE *f = fopen(p.c_str(), "wb");
    if (!f) return -2;
    fwrite(pf.cnt.data(), 2, pf.cnt.size(), f);
    fclose(f);
    p = std::string(out_prefix) + ".maps";
    f = fopen(p.c_str(), "w");
    if (!f) return -2;
    for (auto &a : pf.ins) for (auto &b : a.second) fprintf(f, "I %lld %s %d\n", (long long)a.first, b.first.c_str(), (int)b.second);
    for (auto &a : pf.del) for (auto &b : a.second) fprintf(f, "D %lld %s %d\n", (long long)a.first, b.first.c_str(), (int)b.second);
    for (auto &a : pf.brk) fprintf(f, "B %lld %d\n", (long long)a.first, (int)a.second);
    // the reference sorts each worker's lists by position before merging (ReadMapping.cpp:627-643)
    auto by_pos = [](const Discord &x, const Discord &y) { return x.gPos < y.gPos; };
    std::stable_sort(pf.inv.begin(), pf.inv.end(), by_pos);
    std::stable_sort(pf.tnl.begin(), pf.tnl.end(), by_pos);
    for (auto &d : pf.inv) fprintf(f, "V %lld %lld\n", (long long)d.gPos, (long long)d.dist);
    for (auto &d : pf.tnl) fprintf(f, "T %lld %lld\n", (long long)d.gPos, (long long)d.dist);
    fclose(f);
    return n;
}

void mcxo_vcf_defaults(mcxo_vcf_opts *o)
{
    memset(o, 0, sizeof *o);
    o->ploidy = 2; o->min_allele_depth = 5; o->min_cnv = 50; o->min_gap = 50; o->fragment_size = 500;
    o->max_dup = 5; o->max_clip = 5; o->freq_thr = 0.2f; o->sample_id = "unknown";
}

// MapCaller -i <prefix> -f fq1 [-f2 fq2] -alg .. -vcf <vcf> -t 1: Mapping() then VariantCalling()
int64_t mcxo_map_files_vcf(const mcxo_index *ix, const char *fq1, const char *fq2, int alg, const char *vcf_path, const mcxo_vcf_opts *vo)
{
    VcfOpts o;
    if (vo) {
        o.ploidy = vo->ploidy; o.min_ad = vo->min_allele_depth; o.min_cnv = vo->min_cnv; o.min_gap = vo->min_gap; o.frag_size = vo->fragment_size;
        o.filter = vo->filter != 0; o.gvcf = vo->gvcf != 0 && !vo->monomorphic; o.mono = vo->monomorphic != 0; o.somatic = vo->somatic != 0; // main.cpp:322
        o.freq_thr = vo->freq_thr;
        if (vo->sample_id) o.sample = vo->sample_id;
        if (vo->ref_name) o.ref_name = vo->ref_name;
        if (vo->cmdline) o.cmdline = vo->cmdline;
    }
    Profile pf;
    pf.init(ix->G);
    if (vo) { pf.max_dup = (vo->max_dup <= 0 || vo->max_dup > 15) ? 15 : vo->max_dup; pf.max_clip = vo->max_clip; } // main.cpp:240-244, :323
    int64_t ps[3] = {0, 0, 0};
    int64_t n = map_files_impl(ix, fq1, fq2, alg, nullptr, 1, nullptr, &pf, ps);
    if (n < 0) return n;
    auto by_pos = [](const Discord &x, const Discord &y) { return x.gPos < y.gPos; };
    std::stable_sort(pf.inv.begin(), pf.inv.end(), by_pos);
    std::stable_sort(pf.tnl.begin(), pf.tnl.end(), by_pos);
    u32 avg_rlen = 0; int frag = o.frag_size;
    if (n > 0 && ps[0] > 0) { // ReadMapping.cpp:782-790
        const u32 avg_dist = (u32)(int)(1. * ps[1] / ps[0] + .5);
        avg_rlen = (u32)(int)(1. * ps[2] / (ps[0] << 1) + .5);
        frag = (int)(avg_dist + avg_rlen);
    }
    Caller c(*ix, pf, o, avg_rlen, frag);
    if (!c.run(vcf_path)) return -2;
    return n;
}

static int64_t map_files_impl(const mcxo_index *ix, const char *fq1, const char *fq2, int alg, const char *sam_path,
                              int threads, int64_t *stats, Profile *pf, int64_t *pair_stats)
{
    init_nt4();
    Shared sh;
    sh.ix = ix;
    sh.pf = pf;
    sh.pm.use_nw = (alg == 0);
    if (!sh.in1.open(fq1)) return -1;
    sh.fastq = sh.in1.fastq;
    if (fq2 && fq2[0]) { if (!sh.in2.open(fq2)) return -1; sh.paired = sh.two_files = true; }
    if (g_interleaved) sh.paired = true; // -p (main.cpp:300)
    if (sam_path && sam_path[0]) {
        sh.sam = fopen(sam_path, "w");
        if (!sh.sam) return -2;
        // OutputSamHeaders, ReadMapping.cpp:101-123
        fprintf(sh.sam, "@PG\tID:MapCaller\tPN:MapCaller\tVN:0.9.9.41\n");
        for (const Chrom &c : ix->chr) fprintf(sh.sam, "@SQ\tSN:%s\tLN:%d\n", c.name.c_str(), c.len);
    }
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(worker, &sh);
    worker(&sh);
    for (auto &t : pool) t.join();
    if (sh.sam) fclose(sh.sam);
    sh.in1.close(); sh.in2.close();
    if (stats) {
        stats[0] = sh.n_reads; stats[1] = sh.n_mapped; stats[2] = sh.n_paired;
        stats[3] = sh.ct.ext_steps; stats[4] = sh.ct.sa_hits; stats[5] = sh.ct.lf_steps;
        stats[6] = sh.ct.dp_calls; stats[7] = sh.ct.dp_cells;
    }
    if (pair_stats) { pair_stats[0] = sh.n_paired; pair_stats[1] = sh.dist_sum; pair_stats[2] = sh.len_sum; }
    return sh.n_reads;
}

}
