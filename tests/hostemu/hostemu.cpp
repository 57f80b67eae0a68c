// tests/hostemu/hostemu.cpp — TEST HARNESS ONLY (never part of libmcx.so or any product path).
//
// Runs the device headers of the product (mcx_fm.h, mcx_glue.h: the code every HIP lane
// executes) on the host, one pair after the other, with the same stage order, tiering and
// avgDist replay as mcx_pipeline.hip.  It lets the CPU-only test suite (-m "not gpu") check the
// seeding and per-pair logic against the reference's golden SAM without a GPU.  The wavefront DP
// kernels cannot run here (wave shuffles, LDS); their place is taken by the oracle's DP, which is
// legitimate in a test.  The GPU tests exercise the real kernels through the C ABI.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../mapcaller_amd/csrc/mcx_glue.h"
#include "../../mapcaller_amd/csrc/mcx_dp_lane.h"
#include "../../mapcaller_amd/csrc/mcx_dp_lane2.h"
static long g_detail_checked = 0, g_detail_bad = 0; // MCX_EMU_DETAIL_CHECK: straight-line reads whose detail record was made both ways; those that differ
static long g_simple_why[32]; // which exit of simple_read reads took (MCX_EMU_SIMPLE_WHY=1 prints the tally)
#define MCX_SIMPLE_FAIL(code) do { g_simple_why[code]++; return false; } while (0)
static long g_simple_len[3][40]; // lengths at the exits that a larger in-lane DP would take: [0] end gaps (exit 7), [1] genome side of unequal gaps (11), [2] equal gaps (18)
#define MCX_SIMPLE_NOTE(k, l) (g_simple_len[k][(l) < 39 ? (l) : 39]++)
#include "../../mapcaller_amd/csrc/mcx_simple.h"
#include "../../mapcaller_amd/csrc/mcx_host.h"
#include "simple_io.h"
#include "../../oracle/mcx_oracle.h"

using namespace mcx;

namespace {

struct Emu {
    HostIndex hix;
    IndexView view;
    Params pm;
    std::vector<uint8_t> mapq;
    std::vector<uint32_t> ktab; // the seeding walk's K-mer jump table, here with K = 7
    std::vector<uint64_t> sa_full, c2; // MCX_EMU_RANK2=1: every suffix-array entry and the pair records (two bases per step of the walk)
    std::vector<PairSlot> rank2;
    int mapq_rows = 0;
    Caps caps[2];
    Layout lay[2];
    uint32_t cig_used = 0; // words taken in the batch's CIGAR pool
};

static void set_view(Emu &e)
{
    IndexView &v = e.view;
    HostIndex &h = e.hix;
    for (uint64_t i = 0; i < (h.seq_len + 127) / 128; i++) fm_derive_block(h.bwt.data() + (i << 4)); // the blocks as the kernels see them
    v.bwt = h.bwt.data(); v.sa = h.sa.data(); v.sa_full = nullptr; v.pac = h.pac.data();
    v.end_pos = h.end_pos.data(); v.end_chr = h.end_chr.data(); v.chr_fwd = h.chr_fwd.data();
    v.primary = h.primary; for (int i = 0; i < 5; i++) v.L2[i] = h.L2[i];
    v.seq_len = h.seq_len; v.G = h.G; v.G2 = 2 * h.G;
    v.n_ends = (int)h.end_pos.size(); v.n_chr = (int)h.chr_len.size(); v.sa_intv = h.sa_intv;
    v.ktab = nullptr; v.ktab_k = 0; v.rank = nullptr; v.rank_chunks = 0;
    v.rank2 = nullptr; v.rank2_c2 = nullptr; v.rank2_lone = ~0ull; v.rank2_t0 = 0;
    const int K = 7;
    e.ktab.assign((size_t)4 << (2 * K), 0);
    for (uint32_t i = 0; i < (1u << (2 * K)); i++) {
        uint64_t x0, x1, x2;
        ktab_entry(v, i, K, x0, x1, x2);
        const U4 p = ktab_pack(x0, x1, x2);
        uint32_t *w = e.ktab.data() + (size_t)i * 4;
        w[0] = p.x; w[1] = p.y; w[2] = p.z; w[3] = p.w;
    }
    v.ktab = e.ktab.data(); v.ktab_k = K;
    if (getenv("MCX_EMU_RANK2")) {
        e.sa_full.assign(h.seq_len + 1, 0);
        for (uint64_t r = 0; r <= h.seq_len; r++) { int lf = 0; e.sa_full[r] = r == 0 ? ~0ull : fm_sa(v, r, lf); }
        v.sa_full = e.sa_full.data();
        const uint64_t n_chunks = (h.seq_len + 31) / 32;
        e.rank2.assign(n_chunks * 16 + 8, PairSlot{0, 0});
        uint64_t n_lt[16] = {0};
        for (uint64_t c = 0; c < n_chunks; c++) {
            uint8_t code[32];
            for (int t = 0; t < 32; t++) {
                bool lone;
                code[t] = (uint8_t)fm_pair_code(v, c * 32 + (uint64_t)t, lone);
                if (lone) v.rank2_lone = c * 32 + (uint64_t)t;
            }
            fm_pair_record(code, n_lt, e.rank2.data() + c * 16);
        }
        e.c2.resize(16);
        for (int s2 = 0; s2 < 16; s2++) e.c2[(size_t)s2] = fm_pair_first(v, s2);
        v.rank2 = e.rank2.data(); v.rank2_c2 = e.c2.data(); v.rank2_t0 = ref_code(v, 0);
    }
}

struct Batch {
    std::vector<HostRead> reads;
    std::vector<uint8_t> bases;
    std::vector<uint32_t> off;
    bool paired = false;
};

static void make_reads(const Batch &b, uint32_t pair, ReadRef rd[2])
{
    const int nr = b.paired ? 2 : 1;
    for (int s = 0; s < nr; s++) {
        uint32_t r = pair * nr + s;
        rd[s].ascii = b.bases.data() + b.off[r];
        rd[s].rlen = (int)(b.off[r + 1] - b.off[r]);
        rd[s].flipped = (b.paired && s == 1) ? 1 : 0;
    }
}

// one tier over a selection of pairs; returns the batch ids of the pairs that overflowed
static std::vector<uint32_t> run_tier(Emu &e, int tier, const Batch &b, const std::vector<uint32_t> &ids,
                                      const std::vector<int32_t> &est, std::vector<AlnRec> &recs, std::vector<uint32_t> &cig,
                                      std::vector<PairOut> &pout, int64_t *stats)
{
    const int nr = b.paired ? 2 : 1;
    const uint32_t n = (uint32_t)ids.size();
    std::vector<uint8_t> state((size_t)e.lay[tier].stride * n);
    Ctx cx;
    cx.ix = e.view; cx.pm = e.pm; cx.pm.paired = b.paired; cx.caps = e.caps[tier]; cx.lay = e.lay[tier];
    cx.detail = nullptr; cx.dlay = make_detail_layout(256);
    cx.state = state.data(); cx.mapq_tab = e.mapq.data(); cx.mapq_rows = e.mapq_rows;
    cx.cig_pool = cig.data(); cx.cig_pool_n = &e.cig_used; cx.cig_pool_cap = (uint32_t)cig.size();
    cx.packed = nullptr; cx.wpad = 0; cx.read_ext = nullptr; cx.seed_pool = nullptr;
    std::vector<DpJob> jobs;
    std::vector<uint32_t> kq(4096), kg(e.caps[tier].kmer_cap + 16);
    std::vector<uint32_t> ov;
    // k_seed + k_sa
    for (uint32_t l = 0; l < n; l++) {
        PairState st = pair_state(cx.state, cx.lay, cx.caps, l);
        for (int s = 0; s < nr; s++) {
            uint32_t r = ids[l] * nr + s;
            int64_t ext = 0, blocks = 0;
            ReadRef one;
            one.ascii = b.bases.data() + b.off[r]; one.rlen = (int)(b.off[r + 1] - b.off[r]); one.flipped = (b.paired && s == 1) ? 1 : 0;
            std::vector<uint32_t> pkbuf(packed_words(one.rlen) + 4);
            PackedRead pk; pk.w = pkbuf.data(); pk.stride = 1; pk.n_code = 0;
            int nh = seed_read(cx.ix, one, pk, st.hits[s], cx.caps.hit_cap, ext, blocks);
            st.hdr->n_hits[s] = nh;
            if (stats) { stats[3] += ext; stats[8] += blocks; }
            int keep = nh <= cx.caps.hit_cap ? nh : 0;
            if (cx.ix.sa_full) keep = 0; // (every hit already is a text position)
            for (int i = 0; i < keep; i++) {
                Hit &hh = st.hits[s][i];
                if (hh.len & kHitResolved) { hh.len &= ~kHitResolved; if (stats) stats[4]++; continue; }
                int lf = 0;
                hh.gPos = (int64_t)fm_sa(cx.ix, (uint64_t)hh.gPos, lf);
                if (stats) { stats[4]++; stats[5] += lf; }
            }
        }
    }
    // k_simple: the straight-line pairs go from their seeds to their records at once (mcx_simple.h); the others take the stages below.
    // MCX_EMU_NO_SIMPLE=1: every pair takes the general path (the A/B of the tests).
    std::vector<uint8_t> done(n, 0);
    // MCX_EMU_DETAIL_CHECK=1: the -vcf bookkeeping's per-read detail record made both ways — by the straight-line path (mcx_simple.h
    // SimpleDetail, what k_simple<.., DETAIL> writes) and by the general path's write_detail / pair_stats — for every pair the straight-line
    // path takes; such a pair then goes through the general path as well, and the two records are compared below
    const bool detail_check = tier == 0 && getenv("MCX_EMU_DETAIL_CHECK") != nullptr;
    std::vector<uint8_t> det_a, det_b, took(n, 0);
    if (detail_check) { det_a.assign((size_t)cx.dlay.stride * n * nr, 0); det_b.assign((size_t)cx.dlay.stride * n * nr, 0); }
    if (tier == 0 && !getenv("MCX_EMU_NO_SIMPLE")) {
        // collect (k_simple) -> solve (k_simple_dp) -> replay (k_simple_rest); MCX_EMU_SIMPLE_NO_DP: a pair with a DP problem takes the general path
        const bool with_dp = !getenv("MCX_EMU_SIMPLE_NO_DP");
        std::vector<uint32_t> later;                 // the pairs that wait
        std::vector<SimpleJob> jobs;                 // kSimpleJobs places per waiting pair
        std::vector<SimpleRes> res;
        std::vector<std::vector<uint32_t>> packed((size_t)n * 2);
        auto pass = [&](uint32_t l, int mode, size_t slot) {
            PairState st = pair_state(cx.state, cx.lay, cx.caps, l);
            SimpleRead sr[2];
            uint32_t cg[2][kSimpleRuns];
            int rl[2] = {0, 0};
            bool ok = true, wait = false;
            uint32_t mine[3 * kSimpleJobs];
            SimpleDpIo io; io.mode = mode; io.jobs = mine; io.job_stride = 1; io.res = mode == kDpReplay ? res.data() + slot * kSimpleJobs : nullptr; io.n = 0; io.read = 0;
            SimpleNoDetail nod;
            SimpleDetail det;
            for (int s = 0; s < nr && ok; s++) {
                const uint32_t r = ids[l] * nr + s;
                ReadRef one;
                one.ascii = b.bases.data() + b.off[r]; one.rlen = (int)(b.off[r + 1] - b.off[r]); one.flipped = (b.paired && s == 1) ? 1 : 0;
                rl[s] = one.rlen;
                std::vector<uint32_t> &pk = packed[(size_t)l * 2 + s];
                bool has_n = false;
                if (pk.empty()) {
                    pk.assign(packed_words(one.rlen) + 4, 0u);
                    for (int i = 0; i < one.rlen; i++) { const int c = read_code(one, i); if (c > 3) has_n = true; else pk[i >> 4] |= (uint32_t)c << (30 - 2 * (i & 15)); }
                    if (has_n) pk.assign(1, 0xFFFFFFFFu);
                } else has_n = pk.size() == 1;
                const int nh = st.hdr->n_hits[s];
                if (mode != kDpReplay) { if (has_n) g_simple_why[20]++; else if (nh < 1) g_simple_why[21]++; else if (nh > kSimpleHits) g_simple_why[22]++; }
                io.read = (uint32_t)(l * 2 + s); // (here: the place of the read's words in `packed`)
                ok = !has_n && nh >= 1 && nh <= kSimpleHits;
                if (ok) {
                    int how;
                    if (detail_check) {
                        uint8_t *rec = det_a.data() + ((size_t)l * nr + s) * cx.dlay.stride;
                        det.frags = (Frag *)(rec + sizeof(DetailHdr)); det.ops = rec + cx.dlay.off_ops;
                        how = cx.pm.use_nw ? simple_read<true, uint32_t>(cx.ix, cx.pm, one.rlen, pk.data(), st.hits[s], nh, sr[s], cg[s], 1, io, det)
                                           : simple_read<false, uint32_t>(cx.ix, cx.pm, one.rlen, pk.data(), st.hits[s], nh, sr[s], cg[s], 1, io, det);
                    } else
                        how = cx.pm.use_nw ? simple_read<true, uint32_t>(cx.ix, cx.pm, one.rlen, pk.data(), st.hits[s], nh, sr[s], cg[s], 1, io, nod)
                                           : simple_read<false, uint32_t>(cx.ix, cx.pm, one.rlen, pk.data(), st.hits[s], nh, sr[s], cg[s], 1, io, nod);
                    ok = how != kSimpleNo;
                    wait = wait || how == kSimpleLater;
                }
            }
            if (!ok) return;
            if (nr == 2 && !simple_pair_ok(sr[0], sr[1], est[l])) { g_simple_why[23]++; return; }
            if (wait) {
                later.push_back(l);
                for (int k = 0; k < kSimpleJobs; k++) jobs.push_back(k < io.n ? simple_job_unpack(mine[3 * k], mine[3 * k + 1], mine[3 * k + 2], l * 2, 2) : SimpleJob{0, 0, 0, 0, 0}); // (read: the place of its words in `packed`)
                return;
            }
            const uint32_t want = (uint32_t)sr[0].n_cig + (nr == 2 ? (uint32_t)sr[1].n_cig : 0u);
            if (e.cig_used + want > cx.cig_pool_cap) return;
            const uint32_t off[2] = {e.cig_used, e.cig_used + (uint32_t)sr[0].n_cig};
            AlnRec rec2[2];
            PairOut po;
            if (!simple_pair(cx, nr == 2, sr[0], sr[nr - 1], rl[0], rl[nr - 1], est[l], rec2, off, po)) { g_simple_why[23]++; return; }
            for (int s = 0; s < nr; s++) {
                for (int k = 0; k < sr[s].n_cig; k++) cig[off[s] + k] = cg[s][k];
                recs[(size_t)ids[l] * nr + s] = rec2[s];
            }
            e.cig_used += want;
            pout[ids[l]] = po;
            if (detail_check) { // (the pair takes the general path too)
                for (int s = 0; s < nr; s++) *(DetailHdr *)(det_a.data() + ((size_t)l * nr + s) * cx.dlay.stride) = simple_detail_hdr(cx.ix, nr == 2, s, sr[0], sr[nr - 1]);
                took[l] = 1;
            } else done[l] = 1;
            if (stats) stats[11]++;
        };
        for (uint32_t l = 0; l < n; l++) pass(l, with_dp ? kDpCollect : kDpNone, 0);
        res.resize(jobs.size());
        for (size_t k = 0; k < jobs.size(); k++) {
            const SimpleJob &j = jobs[k];
            if (j.rl == 0) continue;
            uint32_t words[2 + 2 * kSimpleDp];
            LaneMem mem; mem.base = words; mem.stride = 1; mem.lane = 0;
            res[k] = cx.pm.use_nw ? simple_dp_job<true>(cx.ix, j, packed[j.read].data(), mem) : simple_dp_job<false>(cx.ix, j, packed[j.read].data(), mem);
        }
        const std::vector<uint32_t> waiting = later;
        for (size_t i = 0; i < waiting.size(); i++) pass(waiting[i], kDpReplay, i);
        if (stats && getenv("MCX_EMU_SIMPLE_WHY")) fprintf(stderr, "[simple] %zu pairs waited for %zu problems\n", waiting.size(), (size_t)std::count_if(jobs.begin(), jobs.end(), [](const SimpleJob &j) { return j.rl != 0; }));
    }
    if (getenv("MCX_EMU_SIMPLE_WHY") && tier == 0 && n > 100) {
        fprintf(stderr, "[simple] %u pairs; reads leaving simple_read by exit:", n);
        for (int k = 1; k < 24; k++) if (g_simple_why[k]) fprintf(stderr, " %d:%ld", k, g_simple_why[k]);
        fprintf(stderr, " (20 N, 21 no seed, 22 more than %d seeds, 23 not paired within the estimate)\n", kSimpleHits);
        for (int k = 0; k < 3; k++) { fprintf(stderr, "[simple] lengths at exit %s:", k == 0 ? "7" : k == 1 ? "11" : "18"); for (int l = 0; l < 40; l++) if (g_simple_len[k][l]) fprintf(stderr, " %d:%ld", l, g_simple_len[k][l]); fprintf(stderr, "\n"); }
    }
    // k_cluster, k_rescue, k_build
    for (uint32_t l = 0; l < n; l++) {
        if (done[l]) continue;
        ReadRef rd[2];
        make_reads(b, ids[l], rd);
        PairState st = pair_state(cx.state, cx.lay, cx.caps, l);
        st.hdr->flags = 0;
        stage_cluster_pair(cx, l, rd, est[l]);
    }
    for (uint32_t l = 0; l < n; l++) {
        if (done[l]) continue;
        ReadRef rd[2];
        make_reads(b, ids[l], rd);
        RescueSerial ev; ev.kq = kq.data(); ev.kg = kg.data();
        stage_rescue(cx, l, rd, ev);
    }
    for (uint32_t l = 0; l < n; l++) {
        if (done[l]) continue;
        ReadRef rd[2];
        make_reads(b, ids[l], rd);
        const int nj = stage_build(cx, l, rd);
        for (int k = 0; k < nj; k++) {
            DpJob j = pair_job(cx, l, k);
            if (dp_class(j.rLen, j.gLen) < 0) { fprintf(stderr, "hostemu: unsupported DP size\n"); continue; }
            jobs.push_back(j);
        }
    }
    const uint32_t n_jobs = (uint32_t)jobs.size();
    if (getenv("MCX_EMU_DEBUG")) {
        for (uint32_t l = 0; l < n; l++) {
            PairState st = pair_state(cx.state, cx.lay, cx.caps, l);
            fprintf(stderr, "pair %u flags %x n_paired %d est %d\n", ids[l], st.hdr->flags, st.hdr->n_paired, st.hdr->est);
            for (int s = 0; s < nr; s++) {
                fprintf(stderr, " read %d: hits %d cands %d\n", s, st.hdr->n_hits[s], st.hdr->n_cands[s]);
                for (int i = 0; i < st.hdr->n_hits[s]; i++) fprintf(stderr, "   hit r=%d len=%d g=%lld pd=%lld\n", st.hits[s][i].rPos, st.hits[s][i].len, (long long)st.hits[s][i].gPos, (long long)hit_pd(st.hits[s][i]));
                for (int i = 0; i < st.hdr->n_cands[s]; i++) {
                    const Cand &c = st.cands[s][i];
                    fprintf(stderr, "   cand %d score %d mate %d first %d count %d nfrags %d\n", i, c.score, c.mate, c.first, c.count, c.n_frags);
                    for (int k = 0; k < c.n_frags; k++) { const Frag &f = st.frags[c.frag_off + k]; fprintf(stderr, "      frag kind %d r=%d rl=%d g=%lld gl=%d\n", f.kind, f.rPos, f.rLen, (long long)f.gPos, f.gLen); }
                }
            }
        }
    }
    // DP jobs: the product's one-problem-per-lane DP (mcx_dp_lane.h: the code every lane of k_dp_lane runs), one problem after the
    // other with "the lane's words" a plain array; MCX_EMU_ORACLE_DP=1: the oracle's scalar DP instead (what the wavefront kernels of
    // mcx_dp.h, which need wave shuffles and LDS, are checked against on the GPU)
    const bool oracle_dp = getenv("MCX_EMU_ORACLE_DP") != nullptr;
    cx.dp_summary = oracle_dp ? 0 : 1;
    std::vector<uint32_t> words;
    // the lists' kernels take two problems per lane (mcx_dp_lane2.h: k_dp_lane2): neighbours of the list share a lane here too — whatever their shapes —
    // unless MCX_EMU_DP_X1 asks for the one-problem-per-lane form (k_dp_lane, the A/B)
    const bool dp_x2 = !oracle_dp && !getenv("MCX_EMU_DP_X1");
    if (const char *dump = getenv("MCX_EMU_DUMP_JOBS")) { // (the shapes of the batch's problems, for sizing the lists' kernels: scripts/dp_shapes.py)
        if (FILE *f = fopen(dump, "a")) { for (uint32_t j = 0; j < n_jobs; j++) fprintf(f, "%d %d\n", (int)jobs[j].rLen, (int)jobs[j].gLen); fclose(f); }
    }
    for (uint32_t j = 0; dp_x2 && j < n_jobs; j += 2) {
        const bool have_b = j + 1 < n_jobs;
        const DpJob &ja = jobs[j], &jb = jobs[have_b ? j + 1 : j];
        ReadRef rr[2];
        std::vector<uint32_t> pkbuf[2];
        for (int h = 0; h < (have_b ? 2 : 1); h++) {
            const DpJob &job = h ? jb : ja;
            const uint32_t read = ids[job.pair] * nr + job.slot;
            ReadRef &jr = rr[h];
            jr.ascii = b.bases.data() + b.off[read]; jr.rlen = (int)(b.off[read + 1] - b.off[read]); jr.flipped = (b.paired && job.slot == 1) ? 1 : 0;
            if (stats) { stats[6]++; stats[7] += (int64_t)job.rLen * job.gLen; }
            pkbuf[h].assign(packed_words(jr.rlen) + 4, 0u);
            bool has_n = false;
            for (int i = 0; i < jr.rlen; i++) { const int c = read_code(jr, i); if (c > 3) has_n = true; else pkbuf[h][i >> 4] |= (uint32_t)c << (30 - 2 * (i & 15)); }
            if (!has_n && !getenv("MCX_EMU_NO_CODES")) jr.codes = pkbuf[h].data();
        }
        const bool tiny = ja.rLen <= 8 && ja.gLen <= 8 && (!have_b || (jb.rLen <= 8 && jb.gLen <= 8));
        const int K = tiny ? 8 : 16, rows = std::max(ja.rLen, have_b ? jb.rLen : 0), strips = (std::max(ja.gLen, have_b ? jb.gLen : 0) + K - 1) / K;
        const LaneLayout2 l = tiny ? (cx.pm.use_nw ? lane_layout2<8, true>(rows, strips) : lane_layout2<8, false>(rows, strips))
                                   : (cx.pm.use_nw ? lane_layout2<16, true>(rows, strips) : lane_layout2<16, false>(rows, strips));
        words.assign(l.words + 4, 0xDEADBEEFu);
        LaneMem mem; mem.base = words.data(); mem.stride = 1; mem.lane = 0;
        int sc[2];
        if (tiny) { if (cx.pm.use_nw) lane_dp_job2<8, true>(cx, mem, l, ja, rr[0], have_b, jb, rr[have_b ? 1 : 0], sc); else lane_dp_job2<8, false>(cx, mem, l, ja, rr[0], have_b, jb, rr[have_b ? 1 : 0], sc); }
        else { if (cx.pm.use_nw) lane_dp_job2<16, true>(cx, mem, l, ja, rr[0], have_b, jb, rr[have_b ? 1 : 0], sc); else lane_dp_job2<16, false>(cx, mem, l, ja, rr[0], have_b, jb, rr[have_b ? 1 : 0], sc); }
    }
    for (uint32_t j = 0; !dp_x2 && j < n_jobs; j++) {
        const DpJob &job = jobs[j];
        PairState st = pair_state(cx.state, cx.lay, cx.caps, job.pair);
        const uint32_t read = ids[job.pair] * nr + job.slot;
        ReadRef jr;
        jr.ascii = b.bases.data() + b.off[read]; jr.rlen = (int)(b.off[read + 1] - b.off[read]); jr.flipped = (b.paired && job.slot == 1) ? 1 : 0;
        if (stats) { stats[6]++; stats[7] += (int64_t)job.rLen * job.gLen; }
        if (!oracle_dp) {
            // the strip width the pipeline's lists would use: 8 for the tiny list, 16 otherwise (both exercised)
            const bool tiny = job.rLen <= 8 && job.gLen <= 8;
            const int K = tiny ? 8 : 16, strips = (job.gLen + K - 1) / K;
            const LaneLayout l = tiny ? (cx.pm.use_nw ? lane_layout<8, true>(job.rLen, strips) : lane_layout<8, false>(job.rLen, strips))
                                      : (cx.pm.use_nw ? lane_layout<16, true>(job.rLen, strips) : lane_layout<16, false>(job.rLen, strips));
            words.assign(l.words + 4, 0xDEADBEEFu);
            LaneMem mem; mem.base = words.data(); mem.stride = 1; mem.lane = 0;
            // (a read without N also goes through its 2-bit words, as on the device)
            std::vector<uint32_t> pkbuf(packed_words(jr.rlen) + 4, 0u);
            bool has_n = false;
            for (int i = 0; i < jr.rlen; i++) { const int c = read_code(jr, i); if (c > 3) has_n = true; else pkbuf[i >> 4] |= (uint32_t)c << (30 - 2 * (i & 15)); }
            if (!has_n && !getenv("MCX_EMU_NO_CODES")) jr.codes = pkbuf.data();
            if (tiny) { if (cx.pm.use_nw) lane_dp_job<8, true>(cx, mem, l, job, jr); else lane_dp_job<8, false>(cx, mem, l, job, jr); }
            else { if (cx.pm.use_nw) lane_dp_job<16, true>(cx, mem, l, job, jr); else lane_dp_job<16, false>(cx, mem, l, job, jr); }
            continue;
        }
        std::string q(job.rLen, 'N'), t(job.gLen, 'N');
        for (int i = 0; i < job.rLen; i++) q[i] = "ACGTN"[read_code(jr, job.rev ? job.rPos + job.rLen - 1 - i : job.rPos + i)];
        for (int i = 0; i < job.gLen; i++) t[i] = "ACGTN"[ref_code(cx.ix, job.rev ? job.gPos + job.gLen - 1 - i : job.gPos + i)];
        std::vector<char> o1(job.rLen + job.gLen + 2), o2(job.rLen + job.gLen + 2);
        int L = cx.pm.use_nw ? mcxo_nw(q.c_str(), job.rLen, t.c_str(), job.gLen, o1.data(), o2.data(), (int)o1.size())
                             : mcxo_ksw2(q.c_str(), job.rLen, t.c_str(), job.gLen, o1.data(), o2.data(), (int)o1.size());
        if (L < 0) { fprintf(stderr, "hostemu: DP failed\n"); continue; }
        // like the device kernels, right-align the column string in its reserved area
        int w = job.rLen + job.gLen - L;
        for (int i = 0; i < L; i++) st.ops[job.ops_off + w + i] = o1[i] == '-' ? 'D' : (o2[i] == '-' ? 'I' : 'M');
        st.frags[job.frag].ops_off = job.ops_off + w;
        st.frags[job.frag].ops_len = L;
    }
    // k_finish
    for (uint32_t l = 0; l < n; l++) {
        if (done[l]) continue;
        ReadRef rd[2];
        make_reads(b, ids[l], rd);
        const uint32_t pair = ids[l];
        AlnRec *r0 = recs.data() + (int64_t)pair * nr - (int64_t)l * nr; // stage_finish indexes the records by l * nr + s
        stage_finish(cx, l, rd, r0, detail_check ? det_b.data() : nullptr);
        PairState st = pair_state(cx.state, cx.lay, cx.caps, l);
        const PairHdr &h = *st.hdr;
        PairOut o;
        o.flags = h.flags; o.est = h.est; o.est_lo = h.est_lo; o.est_hi = h.est_hi;
        o.pair_dist = h.pair_dist; o.pair_ok = (int16_t)h.pair_ok; o.mapped = (int16_t)h.mapped;
        pout[pair] = o;
        if (h.flags & kOvAny) ov.push_back(pair);
    }
    if (detail_check) {
        for (uint32_t l = 0; l < n; l++) {
            if (!took[l]) continue;
            for (int s = 0; s < nr; s++) {
                const uint8_t *ra = det_a.data() + ((size_t)l * nr + s) * cx.dlay.stride, *rb = det_b.data() + ((size_t)l * nr + s) * cx.dlay.stride;
                const DetailHdr &a = *(const DetailHdr *)ra, &b2 = *(const DetailHdr *)rb;
                bool same = a.type == b2.type && a.n_frags == b2.n_frags && a.fwd == b2.fwd && a.n_ops == b2.n_ops;
                if (s == 0) same = same && a.disc_kind == b2.disc_kind && a.disc_g1 == b2.disc_g1 && a.disc_g2 == b2.disc_g2 && a.disc_dist == b2.disc_dist;
                const Frag *fa = (const Frag *)(ra + sizeof(DetailHdr)) + a.frag0, *fb = (const Frag *)(rb + sizeof(DetailHdr)) + b2.frag0;
                for (int i = 0; same && i < a.n_frags; i++) {
                    const Frag &x = fa[i], &y = fb[i];
                    same = x.kind == y.kind && x.gPos == y.gPos && x.rPos == y.rPos && x.gLen == y.gLen && x.rLen == y.rLen;
                    // (the columns the bookkeeping walks: ops_len of a DP fragment; the others are walked by their lengths)
                    if (same && x.kind == kDp) same = x.ops_len == y.ops_len && memcmp(ra + cx.dlay.off_ops + x.ops_off, rb + cx.dlay.off_ops + y.ops_off, (size_t)x.ops_len) == 0;
                }
                g_detail_checked++;
                if (!same) {
                    g_detail_bad++;
                    if (getenv("MCX_EMU_DEBUG")) {
                        fprintf(stderr, "[detail] pair %u read %d: type %d/%d frags %d/%d fwd %d/%d ops %d/%d disc %d/%d\n", ids[l], s, a.type, b2.type, a.n_frags, b2.n_frags, a.fwd, b2.fwd, a.n_ops, b2.n_ops, a.disc_kind, b2.disc_kind);
                        for (int i = 0; i < std::max(a.n_frags, b2.n_frags) && i < 16; i++)
                            fprintf(stderr, "   %d: kind %d/%d r %d/%d rl %d/%d g %lld/%lld gl %d/%d cols %d/%d\n", i, (int)fa[i].kind, (int)fb[i].kind, (int)fa[i].rPos, (int)fb[i].rPos, (int)fa[i].rLen, (int)fb[i].rLen,
                                    (long long)fa[i].gPos, (long long)fb[i].gPos, (int)fa[i].gLen, (int)fb[i].gLen, (int)fa[i].ops_len, (int)fb[i].ops_len);
                    }
                }
            }
        }
    }
    return ov;
}

static int run_selection(Emu &e, const Batch &b, const std::vector<uint32_t> &ids, const std::vector<int32_t> &est,
                         std::vector<AlnRec> &recs, std::vector<uint32_t> &cig, std::vector<PairOut> &pout, int64_t *stats)
{
    std::vector<uint32_t> ov = run_tier(e, 0, b, ids, est, recs, cig, pout, stats);
    if (ov.empty()) return 0;
    std::sort(ov.begin(), ov.end());
    std::vector<int32_t> ov_est(ov.size());
    for (size_t i = 0; i < ov.size(); i++) ov_est[i] = pout[ov[i]].est;
    if (stats) stats[9] += (int64_t)ov.size();
    std::vector<uint32_t> ov2 = run_tier(e, 1, b, ov, ov_est, recs, cig, pout, stats);
    return ov2.empty() ? 0 : -4;
}

} // namespace

// the comparison phase of the seeding walk in its three forms (mcx_fm.h: seed_compare 16 bases per fetch, seed_compare_wide64 64,
// seed_compare_wide 64 then 128 with the shared chunk kept): reads cut out of the text at random places — both strands, across the
// strand boundary and the text's end — with a base changed or an N somewhere, compared from a random offset: trials whose forms disagree
extern "C" int64_t hostemu_compare_check(const char *prefix, int64_t trials, uint64_t seed)
{
    Emu e;
    std::string err;
    if (!host_index_load(prefix, e.hix, err)) return -1;
    set_view(e);
    const IndexView &ix = e.view;
    auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
    int64_t bad = 0;
    for (int64_t t = 0; t < trials; t++) {
        const int rlen = 17 + (int)(rnd() % 400);
        int64_t j0; // text position of the read's base 0
        switch (rnd() % 4) {
        case 0: j0 = ix.G - (int64_t)(rnd() % (uint64_t)(rlen + 40)); break;              // across the strand boundary
        case 1: j0 = (int64_t)ix.seq_len - (int64_t)(rnd() % (uint64_t)(rlen + 40)); break; // up to (and past) the text's end
        default: j0 = (int64_t)(rnd() % ix.seq_len); break;
        }
        if (j0 < 0) j0 = 0;
        std::vector<uint8_t> ascii((size_t)rlen);
        for (int i = 0; i < rlen; i++) ascii[(size_t)i] = "ACGT"[j0 + i < (int64_t)ix.seq_len ? ref_code(ix, j0 + i) : (int)(rnd() & 3)];
        if (rnd() % 3) { const int at = (int)(rnd() % (uint64_t)rlen); ascii[(size_t)at] = rnd() % 4 == 0 ? 'N' : "ACGT"[(nt4_code(ascii[(size_t)at]) + 1 + (int)(rnd() % 3)) & 3]; }
        ReadRef rd; rd.ascii = ascii.data(); rd.rlen = rlen; rd.flipped = 0; rd.codes = nullptr;
        std::vector<uint32_t> words((size_t)packed_words(rlen) + 4);
        PackedRead pk; pk.w = words.data(); pk.stride = 1; pk.n_code = 0;
        pack_read(rd, pk);
        const int start = (int)(rnd() % 8), p0 = start + (int)(rnd() % (uint64_t)(rlen - start));
        int got[3];
        for (int form = 0; form < 3; form++) {
            SeedWalk w; w.x0 = w.x1 = 0; w.x2 = 1; w.tpos = j0 + start; w.carry = 0; w.carry_dir = 0; w.start = start; w.phase = 2; w.ended = 0;
            int p = p0;
            while (w.phase == 2) {
                if (form == 0) seed_compare(ix, pk, rlen, p, w, 1 << 30);
                else if (form == 1) seed_compare_wide64(ix, pk, rlen, p, w, 1 << 30);
                else seed_compare_wide(ix, pk, rlen, p, w, 1 << 30);
            }
            got[form] = p;
        }
        if (got[0] != got[1] || got[0] != got[2]) bad++;
    }
    return bad;
}

// pair_by_distance (mcx_glue.h: read 2's candidates swept in PosDiff order) against the plain scan of all n1 x n2 pairs
// (CheckPairedAlignmentDistance, ReadMapping.cpp:244-303) on random candidate lists in ascending PosDiff: trials that differ
extern "C" int64_t hostemu_pairing_check(int64_t trials, uint64_t seed)
{
    auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
    int64_t bad = 0;
    for (int64_t t = 0; t < trials; t++) {
        Cand a[2][24], b[2][24];
        int n[2];
        for (int s = 0; s < 2; s++) {
            n[s] = (int)(rnd() % 13);
            int64_t pd = (int64_t)(rnd() % 5000) + 1;
            for (int k = 0; k < n[s]; k++) {
                pd += (int64_t)(rnd() % 4 == 0 ? 0 : rnd() % 700);
                cand_init(a[s][k], rnd() % 5 == 0 ? 0 : (int)(rnd() % 150) + 1, k, 1, pd);
                b[s][k] = a[s][k];
            }
        }
        const int64_t est = (int64_t)(rnd() % 1500) + 1;
        int lo = 0, hi = 0;
        const int paired = pair_by_distance(est, a[0], n[0], a[1], n[1], lo, hi);
        // the plain scan
        Cand *c1 = b[0], *c2 = b[1];
        const int n1 = n[0], n2 = n[1];
        int64_t max_lt = -1, min_ge = 0x7fffffff;
        if (n1 * n2 > 100) { keep_top_scores(c1, n1); keep_top_scores(c2, n2); }
        auto partner = [&](const Cand &x, int &ps) {
            int pick = -1;
            ps = 0;
            for (int j = 0; j < n2; j++) {
                const int sj = c2[j].score;
                const int64_t pj = c2[j].pd0;
                if (sj == 0 || pj < x.pd0) continue;
                const int64_t d = pj - x.pd0;
                if (d < est) { if (d > max_lt) max_lt = d; if (sj > ps) { pick = j; ps = sj; } }
                else if (d < min_ge) min_ge = d;
            }
            return pick;
        };
        int64_t top = 0;
        for (int i = 0; i < n1; i++) { if (c1[i].score == 0) continue; int ps; if (partner(c1[i], ps) >= 0) { const int64_t sc = (int64_t)c1[i].score + ps; if (sc > top) top = sc; } }
        int want = 0;
        if (top > 0) for (int i = 0; i < n1; i++) { if (c1[i].score == 0) continue; int ps; const int pick = partner(c1[i], ps); if (pick >= 0 && (int64_t)c1[i].score + ps == top) { want++; c1[i].mate = pick; c2[pick].mate = i; } }
        bool same = want == paired && lo == (int)(max_lt + 1) && hi == (int)min_ge;
        for (int s = 0; s < 2 && same; s++) for (int k = 0; k < n[s]; k++) if (a[s][k].mate != b[s][k].mate || a[s][k].score != b[s][k].score) { same = false; break; }
        if (!same) bad++;
    }
    return bad;
}

// the pair records' self-check (mcx_fm.h fm_pair_step_agrees) on the host: trials that disagree
extern "C" int64_t hostemu_pair_check(const char *prefix, int64_t trials)
{
    Emu e;
    std::string err;
    if (!host_index_load(prefix, e.hix, err)) return -1;
    setenv("MCX_EMU_RANK2", "1", 1);
    set_view(e);
    unsetenv("MCX_EMU_RANK2");
    int64_t bad = 0;
    for (int64_t t = 0; t < trials; t++) bad += fm_pair_step_agrees(e.view, (uint64_t)t) ? 0 : 1;
    return bad;
}

extern "C" {

// MCX_EMU_DETAIL_CHECK runs so far: out[0] reads whose detail record was made both ways, out[1] those that differ; the counts start over
void hostemu_detail_counts(int64_t out[2]) { out[0] = g_detail_checked; out[1] = g_detail_bad; g_detail_checked = g_detail_bad = 0; }

// One problem through the product's one-problem-per-lane DP (mcx_dp_lane.h) for the reference's function-level vectors:
// q over ACGTN, t over ACGT (the 2-bit genome holds no N), strips of K = 8 or 16 columns.  ops_out: the column string
// ('M' / 'I' / 'D', front to back, NUL-terminated, room for qlen + tlen + 1); *score: nw's final score doubled (0 for ksw2).
// Returns the string's length, -1 for a target with a letter outside ACGT.
int hostemu_lane_dp(int use_nw, const char *q, int qlen, const char *t, int tlen, int K, char *ops_out, int *score)
{
    auto code = [](char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; };
    for (int i = 0; i < tlen; i++) if (code(t[i]) > 3) return -1;
    const int strips = (tlen + K - 1) / K;
    LaneLayout l;
    if (K == 8) l = use_nw ? lane_layout<8, true>(qlen, strips) : lane_layout<8, false>(qlen, strips);
    else l = use_nw ? lane_layout<16, true>(qlen, strips) : lane_layout<16, false>(qlen, strips);
    std::vector<uint32_t> words(l.words + 4, 0xDEADBEEFu);
    LaneMem mem; mem.base = words.data(); mem.stride = 1; mem.lane = 0;
    lane_stage_query(mem, l, qlen, [&](int p, uint32_t &codes, uint32_t &flags) {
        codes = 0; flags = 0;
        for (int k = 0; k < 16 && p + k < qlen; k++) { const int c = code(q[p + k]); codes |= (uint32_t)(c & 3) << (30 - 2 * k); flags |= (uint32_t)(c > 3) << (15 - k); }
    });
    auto tgt16 = [&](int b0) -> uint32_t {
        uint32_t v = 0;
        for (int k = 0; k < 16; k++) v = (v << 2) | (b0 + k < tlen ? (uint32_t)code(t[b0 + k]) : 0u);
        return v;
    };
    std::vector<uint8_t> ops((size_t)qlen + tlen + 1, 0);
    int w, sc = 0;
    if (K == 8) {
        if (use_nw) { sc = lane_sweep_nw<8>(mem, l, qlen, tlen, tgt16); w = lane_trace_nw<8>(mem, l, qlen, tlen, tgt16, ops.data(), nullptr, 0u); }
        else { lane_sweep_ksw2<8>(mem, l, qlen, tlen, tgt16); w = lane_trace_ksw2<8>(mem, l, qlen, tlen, tgt16, ops.data(), nullptr, 0u); }
    } else {
        if (use_nw) { sc = lane_sweep_nw<16>(mem, l, qlen, tlen, tgt16); w = lane_trace_nw<16>(mem, l, qlen, tlen, tgt16, ops.data(), nullptr, 0u); }
        else { lane_sweep_ksw2<16>(mem, l, qlen, tlen, tgt16); w = lane_trace_ksw2<16>(mem, l, qlen, tlen, tgt16, ops.data(), nullptr, 0u); }
    }
    const int L = qlen + tlen - w;
    memcpy(ops_out, ops.data() + w, (size_t)L);
    ops_out[L] = 0;
    if (score) *score = sc;
    return L;
}

// Two problems in ONE lane through the two-problems-per-lane DP (mcx_dp_lane2.h: 16-bit halves), for the same vectors: problem A in the low halves,
// problem B in the high ones, whatever their shapes.  ops_a / ops_b, score[2] as hostemu_lane_dp; len[2] = the strings' lengths.  Returns 0, -1 for a
// target with a letter outside ACGT.
int hostemu_lane_dp2(int use_nw, const char *qa, int qlen_a, const char *ta, int tlen_a, const char *qb, int qlen_b, const char *tb, int tlen_b, int K,
                     char *ops_a, char *ops_b, int *score, int *len)
{
    auto code = [](char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; };
    for (int i = 0; i < tlen_a; i++) if (code(ta[i]) > 3) return -1;
    for (int i = 0; i < tlen_b; i++) if (code(tb[i]) > 3) return -1;
    const int rows = std::max(qlen_a, qlen_b), strips = (std::max(tlen_a, tlen_b) + K - 1) / K;
    LaneLayout2 l;
    if (K == 8) l = use_nw ? lane_layout2<8, true>(rows, strips) : lane_layout2<8, false>(rows, strips);
    else l = use_nw ? lane_layout2<16, true>(rows, strips) : lane_layout2<16, false>(rows, strips);
    std::vector<uint32_t> words(l.words + 4, 0xDEADBEEFu);
    LaneMem mem; mem.base = words.data(); mem.stride = 1; mem.lane = 0;
    const char *qs[2] = {qa, qb}, *ts[2] = {ta, tb};
    const int ql[2] = {qlen_a, qlen_b}, tl[2] = {tlen_a, tlen_b};
    lane_stage_query2(mem, l, qlen_a, qlen_b, [&](int h, int p) { return code(qs[h][p]); });
    auto tgt = [&](int h, int b0) -> uint32_t {
        uint32_t v = 0;
        for (int k = 0; k < 16; k++) v = (v << 2) | (b0 + k < tl[h] ? (uint32_t)code(ts[h][b0 + k]) : 0u);
        return v;
    };
    auto tgt_a = [&](int b0) { return tgt(0, b0); };
    auto tgt_b = [&](int b0) { return tgt(1, b0); };
    score[0] = score[1] = 0;
    if (K == 8) { if (use_nw) lane_sweep_nw2<8>(mem, l, qlen_a, tlen_a, qlen_b, tlen_b, tgt_a, tgt_b, &score[0], &score[1], tgt_a(0), tgt_b(0)); else lane_sweep_ksw2_2<8>(mem, l, qlen_a, tlen_a, qlen_b, tlen_b, tgt_a, tgt_b, tgt_a(0), tgt_b(0)); }
    else { if (use_nw) lane_sweep_nw2<16>(mem, l, qlen_a, tlen_a, qlen_b, tlen_b, tgt_a, tgt_b, &score[0], &score[1], tgt_a(0), tgt_b(0)); else lane_sweep_ksw2_2<16>(mem, l, qlen_a, tlen_a, qlen_b, tlen_b, tgt_a, tgt_b, tgt_a(0), tgt_b(0)); }
    char *outs[2] = {ops_a, ops_b};
    std::vector<uint32_t> area[2]; // (word-aligned, rounded up: the walks store four columns at a time)
    for (int h = 0; h < 2; h++) area[h].assign(((size_t)ql[h] + tl[h]) / 4 + 4, 0u);
    auto walk = [&](auto &wa, auto &wb) {
        wa.begin(ql[0], tl[0], (uint8_t *)area[0].data(), nullptr);
        wb.begin(ql[1], tl[1], (uint8_t *)area[1].data(), nullptr);
        lane_walk2(wa, wb);
        wa.sink.end(0u, ql[0] + tl[0]); wb.sink.end(0u, ql[1] + tl[1]);
        const int ws[2] = {wa.sink.w, wb.sink.w};
        for (int h = 0; h < 2; h++) {
            len[h] = ql[h] + tl[h] - ws[h];
            memcpy(outs[h], (const uint8_t *)area[h].data() + ws[h], (size_t)len[h]);
            outs[h][len[h]] = 0;
        }
    };
    if (K == 8) {
        if (use_nw) { LaneWalk2<8, true> wa(mem, l, 0), wb(mem, l, 1); walk(wa, wb); }
        else { LaneWalk2<8, false> wa(mem, l, 0), wb(mem, l, 1); walk(wa, wb); }
    } else {
        if (use_nw) { LaneWalk2<16, true> wa(mem, l, 0), wb(mem, l, 1); walk(wa, wb); }
        else { LaneWalk2<16, false> wa(mem, l, 0), wb(mem, l, 1); walk(wa, wb); }
    }
    return 0;
}

// tier0 = {hit_cap, cand_cap, frag_cap, ops_cap, job_cap} or null for the product defaults.
// stats[12]: 0 reads 1 mapped 2 pairs 3 E 4 H 5 LF 6 dp jobs 7 dp cells 8 blocks 9 tier-1 pairs 10 replayed pairs
int64_t hostemu_map_files(const char *prefix, const char *fq1, const char *fq2, int alg, const char *sam_path,
                          int batch_reads, const int *tier0, int rlen_max, int64_t *stats)
{
    Emu e;
    std::string err;
    if (!host_index_load(prefix, e.hix, err)) { fprintf(stderr, "%s\n", err.c_str()); return -1; }
    set_view(e);
    e.pm.max_pos_diff = 30; e.pm.max_mm_rate = 0.05f; e.pm.use_nw = alg == 0; e.pm.paired = 1;
    e.caps[0].hit_cap = 48; e.caps[0].cand_cap = 12; e.caps[0].frag_cap = 96; e.caps[0].ops_cap = 1024; e.caps[0].job_cap = 16;
    e.caps[0].cig_cap = 32; e.caps[0].kmer_cap = 2048;
    if (tier0) { e.caps[0].hit_cap = tier0[0]; e.caps[0].cand_cap = tier0[1]; e.caps[0].frag_cap = tier0[2]; e.caps[0].ops_cap = tier0[3]; e.caps[0].job_cap = tier0[4]; }
    int seeds = rlen_max / (kMinSeedLength + 1) + 1;
    e.caps[1].hit_cap = seeds * kOccThr + rlen_max / 8 + 16; e.caps[1].cand_cap = e.caps[1].hit_cap;
    e.caps[1].frag_cap = 3 * e.caps[1].hit_cap + 16; e.caps[1].ops_cap = 96 * 1024; e.caps[1].job_cap = 2048;
    for (int t = 0; t < 2; t++) { e.caps[t].hit_seed = e.caps[t].hit_cap; e.caps[t].cand_seed = e.caps[t].cand_cap; }
    e.caps[1].cig_cap = 32; e.caps[1].kmer_cap = 4096;
    for (int t = 0; t < 2; t++) e.lay[t] = make_layout(e.caps[t]);
    e.mapq_rows = rlen_max + 64;
    e.mapq.assign((size_t)e.mapq_rows * 6, 0);
    for (int s = 1; s < e.mapq_rows; s++)
        for (int d = 1; d <= 5 && d < s; d++) {
            int sub = s - d;
            int q = (int)(30 * (1 - (float)(s - sub) / s) * log(s) + 0.4999);
            e.mapq[(size_t)s * 6 + d] = (uint8_t)(q > 60 ? 60 : q);
        }
    ReadFile f1, f2;
    const bool paired = fq2 && fq2[0];
    if (!f1.open(fq1, err) || (paired && !f2.open(fq2, err))) { fprintf(stderr, "%s\n", err.c_str()); return -1; }
    FILE *sam = nullptr;
    if (sam_path && sam_path[0]) {
        sam = fopen(sam_path, "w");
        std::string hdr; sam_header(e.hix, hdr); fputs(hdr.c_str(), sam);
    }
    int64_t avg[4] = {1000, 0, 0, 0};
    int64_t total = 0, mapped = 0;
    const size_t batch = std::max(200, batch_reads / 200 * 200);
    bool eof = false;
    std::string line;
    while (!eof) {
        Batch b;
        b.off.assign(1, 0);
        while (b.reads.size() < batch) {
            HostRead x, y;
            if (!f1.next(x)) { eof = true; break; }
            b.reads.push_back(x);
            if (paired) { f2.next(y); b.reads.push_back(y); }
        }
        if (b.reads.empty()) break;
        for (auto &r : b.reads) { b.bases.insert(b.bases.end(), r.seq.begin(), r.seq.end()); b.off.push_back((uint32_t)b.bases.size()); }
        b.bases.resize(b.bases.size() + 64, 'N');
        const uint32_t n = (uint32_t)b.reads.size();
        b.paired = paired && (n % 2 == 0);
        const uint32_t n_pairs = b.paired ? n / 2 : n;
        std::vector<AlnRec> recs(n);
        std::vector<uint32_t> cig((size_t)n * 64); // the batch's CIGAR pool (re-runs take new words)
        e.cig_used = 0;
        std::vector<PairOut> pout(n_pairs);
        std::vector<uint32_t> ids(n_pairs);
        for (uint32_t i = 0; i < n_pairs; i++) ids[i] = i;
        std::vector<int32_t> est(n_pairs, (int32_t)((uint32_t)avg[0] * 1.5));
        if (run_selection(e, b, ids, est, recs, cig, pout, stats)) return -4;
        if (b.paired) {
            std::vector<uint32_t> redo; std::vector<int32_t> redo_est;
            int64_t after[4];
            for (int iter = 0;; iter++) {
                avg_replay(pout.data(), n_pairs, avg, redo, redo_est, after);
                if (redo.empty()) break;
                if (iter == 63) return -5;
                if (stats) stats[10] += (int64_t)redo.size();
                if (run_selection(e, b, redo, redo_est, recs, cig, pout, stats)) return -4;
            }
            avg[0] = after[0]; avg[1] = after[1]; avg[2] = after[2];
        }
        avg[3] += n;
        total += n;
        for (uint32_t p = 0; p < n_pairs; p++) mapped += pout[p].mapped;
        if (sam)
            for (uint32_t i = 0; i < n; i++) {
                sam_line(e.hix, b.reads[i], b.paired && (i & 1), f1.fastq(), recs[i], cig.data() + (size_t)recs[i].pad[0], line);
                fputs(line.c_str(), sam); fputc('\n', sam);
            }
    }
    if (sam) fclose(sam);
    if (stats) { stats[0] = total; stats[1] = mapped; stats[2] = avg[1]; }
    return total;
}

}
