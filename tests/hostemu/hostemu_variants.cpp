// tests/hostemu/hostemu_variants.cpp — TEST HARNESS: the product's variant-calling host logic
// (mapcaller_amd/csrc/mcx_variants_host.h) over a CPU stand-in for the dense half.
//
// The product runs block depth, the per-position scan, column gathers and range reductions as HIP
// kernels over planes in HBM (mcx_variants.hip).  Here the same interface is served by plain loops
// over a host array filled from a .prof file (10 x u16 per position, as the reference's profile
// dump and the oracle write it), with the product's own eval_site() compiled for the host — so the
// sparse logic (indel calls, run pairing, gVCF blocks, break points, filters, VCF text) is checked
// against the golden VCFs without a GPU.  Not part of the product.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../mapcaller_amd/csrc/mcx_variants_host.h"

using namespace mcx;
using namespace mcx_vc;

static thread_local std::string g_err;
int mcx_set_error(int code, const std::string &msg) { g_err = msg; return code; }

namespace {

class HostProfile : public DenseProfile {
public:
    HostProfile(const HostIndex &h, std::vector<uint32_t> planes) : hix_(h), pl_(std::move(planes)), G_(h.G)
    {
        memset(&ix_, 0, sizeof ix_);
        ix_.pac = hix_.pac.data(); ix_.G = G_; ix_.G2 = 2 * G_; ix_.seq_len = hix_.seq_len;
    }
    int64_t genome_size() const override { return G_; }
    int cov(int64_t g) const { return (int)(pl_[g] + pl_[G_ + g] + pl_[2 * G_ + g] + pl_[3 * G_ + g]); }

    int scan(const ScanParams &sp, SiteVec &sites, double &ms_depth, double &ms_scan) override
    {
        ms_depth = ms_scan = 0;
        const int64_t nb = (G_ + kBlock - 1) / kBlock;
        depth_.assign((size_t)nb, 0);
        for (int64_t b = 0; b < nb; b++) { // k_vc_depth
            int64_t sum = 0;
            for (int64_t g = b * kBlock; g < std::min<int64_t>(G_, (b + 1) * kBlock); g++) sum += cov(g);
            depth_[b] = sum > 0 ? (int32_t)(sum / kBlock) : 0;
        }
        sites.clear();
        int prev_cls = 0; bool prev_cand = false; // k_vc_scan, one position at a time
        for (int64_t g = 0; g < G_; g++) {
            struct { const uint32_t *p; int64_t G; uint32_t get(int k, int64_t at) const { return p[(size_t)k * G + at]; } } planes = {pl_.data(), G_};
            const SiteEval e = eval_site(planes, depth_.data(), ix_, sp, g);
            SiteRec ev = e.rec; ev.geno = ev.qscore = 0; ev.alt = 0xFF; ev.DP = ev.AD_ref = ev.AD_alt = 0;
            if (prev_cls != 0 && e.cls != prev_cls) { ev.type = prev_cls == 1 ? eGapEnd : eDupEnd; sites.push_back(ev); }
            if (e.cls != 0 && e.cls != prev_cls) { ev.type = e.cls == 1 ? eGapStart : eDupStart; sites.push_back(ev); }
            if (e.call || (e.cand && sp.mono)) sites.push_back(e.rec);
            if (sp.gvcf && e.cand != prev_cand) { ev.type = e.cand ? eNormStart : eNormEnd; sites.push_back(ev); }
            prev_cls = e.cls; prev_cand = e.cand;
        }
        std::stable_sort(sites.begin(), sites.end(), [](const SiteRec &a, const SiteRec &b) { return a.pos != b.pos ? a.pos < b.pos : a.type < b.type; });
        return 0;
    }
    int gather(const std::vector<int64_t> &pos, ColVec &out) override
    {
        out.resize(pos.size());
        for (size_t i = 0; i < pos.size(); i++) {
            const int64_t g = pos[i];
            Column c; memset(&c, 0, sizeof c);
            if (g >= 0 && g < G_) {
                for (int k = 0; k < nPlanes; k++) c.v[k] = pl_[(size_t)k * G_ + g];
                c.depth = depth_[g / kBlock];
                c.ref = (uint32_t)ref_code(ix_, g);
            }
            out[i] = c;
        }
        return 0;
    }
    int ranges(const std::vector<RangeQ> &q, std::vector<unsigned long long> &out) override
    {
        out.resize(q.size());
        for (size_t i = 0; i < q.size(); i++) {
            unsigned long long acc = q[i].mode ? ~0ull : 0ull;
            for (int64_t g = std::max<int64_t>(q[i].beg, 0); g <= q[i].end && g < G_; g++) {
                const unsigned long long c = (unsigned long long)cov(g);
                if (q[i].mode) { if (c > 0 && c < acc) acc = c; } else acc += c;
            }
            out[i] = acc;
        }
        return 0;
    }

private:
    const HostIndex &hix_;
    std::vector<uint32_t> pl_; // [plane][G]
    int64_t G_;
    IndexView ix_;
    std::vector<int32_t> depth_;
};

} // namespace

extern "C" {

const char *hostemu_vc_error(void) { return g_err.c_str(); }

// prof: 10 x u16 per position (A C G T multi_hit readCount F1 R2 F2 R1); maps: the text the reference
// tool / the oracle write ("I pos seq n", "D pos seq n", "B pos n", "V pos dist", "T pos dist")
int hostemu_call_variants(const char *prefix, const char *prof_path, const char *maps_path, int64_t pairs, int64_t dist_sum, int64_t len_sum,
                          const mcx_vcf_opts *opts, const char *vcf_path)
{
    HostIndex hix;
    std::string err;
    if (!host_index_load(prefix, hix, err)) return mcx_set_error(MCX_ERR_IO, err);
    const int64_t G = hix.G;
    std::vector<uint16_t> raw((size_t)G * 10);
    FILE *f = fopen(prof_path, "rb");
    if (!f || fread(raw.data(), 2, raw.size(), f) != raw.size()) { if (f) fclose(f); return mcx_set_error(MCX_ERR_IO, std::string("cannot read ") + prof_path); }
    fclose(f);
    std::vector<uint32_t> planes((size_t)G * 10);
    for (int64_t g = 0; g < G; g++) for (int k = 0; k < 10; k++) planes[(size_t)k * G + g] = raw[(size_t)g * 10 + k];
    std::vector<mcx_sparse_rec> recs;
    f = fopen(maps_path, "r");
    if (!f) return mcx_set_error(MCX_ERR_IO, std::string("cannot read ") + maps_path);
    char type, seq[4096];
    long long pos, x;
    char line[8192];
    while (fgets(line, sizeof line, f)) {
        mcx_sparse_rec r; memset(&r, 0, sizeof r);
        if (sscanf(line, "%c %lld", &type, &pos) != 2) continue;
        r.type = (uint8_t)type; r.pos = pos;
        if (type == 'I' || type == 'D') {
            int n = 0;
            seq[0] = 0;
            const char *p = strchr(line + 2, ' ');
            if (!p) continue;
            // "<seq> <count>" — the sequence may be empty (two spaces)
            const char *q = strrchr(line, ' ');
            n = atoi(q + 1);
            size_t len = (size_t)(q - (p + 1));
            if (len > sizeof r.seq) len = sizeof r.seq;
            memcpy(r.seq, p + 1, len); r.len = (uint8_t)len;
            for (int k = 0; k < n; k++) recs.push_back(r);
        } else if (type == 'B') {
            if (sscanf(line, "%c %lld %lld", &type, &pos, &x) != 3) continue;
            for (long long k = 0; k < x; k++) recs.push_back(r);
        } else if (type == 'V' || type == 'T') {
            if (sscanf(line, "%c %lld %lld", &type, &pos, &x) != 3) continue;
            int64_t d = x; memcpy(r.seq, &d, 8);
            recs.push_back(r);
        }
    }
    fclose(f);
    mcx_vcf_opts o = *opts;
    if (o.gvcf && o.monomorphic) o.gvcf = 0;
    HostProfile prof(hix, std::move(planes));
    Caller c(hix, 2 * G, prof, o);
    return c.run(recs.data(), recs.size(), pairs, dist_sum, len_sum, vcf_path, nullptr);
}

}
