// mapcaller_amd/csrc/mcx_host.cpp — index files and the SAM header on the host.
//
// Index files are the byte-compatible BWA files `MapCaller index` writes (reference
// src/BWT_Index/bwtindex.c:77-160, src/BWT_Index/bwt.c:174-196, src/BWT_Index/bntseq.c:60-91);
// the loader mirrors src/bwt_index.cpp:16-124 and :232-258.  (Read files and SAM lines: mcx_files.cpp.)
#include "mcx_host.h"
#include "mcx_types.h"
#include "../../include/mcx.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <zlib.h>

namespace mcx {

static bool read_file(const std::string &path, std::vector<uint8_t> &out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize((size_t)n);
    size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

void host_index_finish(HostIndex &ix)
{
    const int64_t G2 = 2 * ix.G;
    ix.chr_fwd.clear(); ix.end_pos.clear(); ix.end_chr.clear();
    int64_t total = 0;
    std::vector<std::pair<int64_t, int32_t>> ends;
    for (size_t i = 0; i < ix.chr_len.size(); i++) {
        ix.chr_fwd.push_back(total);
        const int64_t fwd_end = total + ix.chr_len[i] - 1;
        total += ix.chr_len[i];
        const int64_t rev_end = (G2 - total) + ix.chr_len[i] - 1;
        ends.push_back(std::make_pair(fwd_end, (int32_t)i));
        ends.push_back(std::make_pair(rev_end, (int32_t)i));
    }
    std::sort(ends.begin(), ends.end());
    for (auto &e : ends) { ix.end_pos.push_back(e.first); ix.end_chr.push_back(e.second); }
}

bool host_index_load(const std::string &prefix, HostIndex &ix, std::string &err)
{
    std::vector<uint8_t> raw;
    if (!read_file(prefix + ".bwt", raw) || raw.size() < 40 + 64) { err = "cannot read " + prefix + ".bwt"; return false; }
    memcpy(&ix.primary, raw.data(), 8);
    memcpy(&ix.L2[1], raw.data() + 8, 32);
    ix.L2[0] = 0;
    ix.seq_len = ix.L2[4];
    ix.bwt.resize((raw.size() - 40) / 4);
    memcpy(ix.bwt.data(), raw.data() + 40, ix.bwt.size() * 4);
    const uint64_t want_words = ((ix.seq_len + 127) / 128) * 8 + ((ix.seq_len + 15) / 16) + 8;
    if (ix.bwt.size() < want_words) { err = prefix + ".bwt is truncated"; return false; }

    if (!read_file(prefix + ".sa", raw) || raw.size() < 56) { err = "cannot read " + prefix + ".sa"; return false; }
    uint64_t intv = 0, sl = 0;
    memcpy(&intv, raw.data() + 40, 8);
    memcpy(&sl, raw.data() + 48, 8);
    if (intv == 0 || (intv & (intv - 1)) || sl != ix.seq_len) { err = prefix + ".sa does not belong to " + prefix + ".bwt"; return false; }
    ix.sa_intv = (int)intv;
    const uint64_t n_sa = (ix.seq_len + intv) / intv;
    if ((raw.size() - 56) / 8 < n_sa - 1) { err = prefix + ".sa is truncated"; return false; }
    ix.sa.assign(n_sa, 0);
    ix.sa[0] = ~0ull;
    memcpy(ix.sa.data() + 1, raw.data() + 56, (n_sa - 1) * 8);

    FILE *f = fopen((prefix + ".ann").c_str(), "r");
    if (!f) { err = "cannot read " + prefix + ".ann"; return false; }
    long long l_pac = 0; int n_seqs = 0; unsigned seed = 0;
    if (fscanf(f, "%lld%d%u", &l_pac, &n_seqs, &seed) != 3) { fclose(f); err = prefix + ".ann is malformed"; return false; }
    ix.G = l_pac;
    for (int i = 0; i < n_seqs; i++) {
        unsigned gi; char name[1024]; long long off; int len, n_ambs, c;
        if (fscanf(f, "%u%1023s", &gi, name) != 2) { fclose(f); err = prefix + ".ann is malformed"; return false; }
        while ((c = fgetc(f)) != '\n' && c != EOF) {}
        if (fscanf(f, "%lld%d%d", &off, &len, &n_ambs) != 3) { fclose(f); err = prefix + ".ann is malformed"; return false; }
        ix.chr_name.push_back(name); ix.chr_len.push_back(len);
    }
    fclose(f);
    if ((uint64_t)ix.G * 2 != ix.seq_len) { err = prefix + ".ann and .bwt disagree on the genome size"; return false; }
    f = fopen((prefix + ".amb").c_str(), "r"); // presence is required by the reference (GetData.cpp:148-166)
    if (!f) { err = "cannot read " + prefix + ".amb"; return false; }
    fclose(f);
    if (!read_file(prefix + ".pac", ix.pac) || (int64_t)ix.pac.size() < ix.G / 4 + 1) { err = "cannot read " + prefix + ".pac"; return false; }
    ix.pac.resize(ix.pac.size() + 32, 0); // readers fetch aligned 16-byte chunks
    host_index_finish(ix);
    return true;
}

// ---------------------------------------------------------------------------------------------
// SAM header
// ---------------------------------------------------------------------------------------------
void sam_header(const HostIndex &ix, std::string &out) // OutputSamHeaders, ReadMapping.cpp:101-123
{
    out = "@PG\tID:MapCaller\tPN:MapCaller\tVN:0.9.9.41\n";
    char buf[1200];
    for (size_t i = 0; i < ix.chr_name.size(); i++) {
        snprintf(buf, sizeof buf, "@SQ\tSN:%s\tLN:%d\n", ix.chr_name[i].c_str(), ix.chr_len[i]);
        out += buf;
    }
}

} // namespace mcx

void mcx_disc_resolve(const mcx_sparse_rec *events, size_t n, int64_t G, std::vector<mcx_sparse_rec> &out)
{
    std::vector<const mcx_sparse_rec *> ev;
    for (size_t i = 0; i < n; i++) if (events[i].type == 'E') ev.push_back(events + i);
    std::stable_sort(ev.begin(), ev.end(), [](const mcx_sparse_rec *a, const mcx_sparse_rec *b) { return a->pos < b->pos; });
    const int64_t G2 = 2 * G;
    int64_t last[2] = {0, 0}; // DiscordPair {gPos, dist}
    auto push = [&](char type, int64_t gpos, int64_t dist) {
        mcx_sparse_rec r; memset(&r, 0, sizeof r);
        r.pos = gpos; r.type = (uint8_t)type; r.len = 0; memcpy(r.seq, &dist, 8);
        out.push_back(r);
    };
    for (const mcx_sparse_rec *e : ev) {
        int64_t g1, g2, dist;
        memcpy(&g1, e->seq, 8); memcpy(&g2, e->seq + 8, 8); memcpy(&dist, e->seq + 16, 8);
        const int kind = e->len;
        if (kind == 1) {
            int64_t d = G2 - g1 - g2; if (d < 0) d = -d;
            last[1] = d; // (.dist always, .gPos only inside the range: ReadMapping.cpp:492-496)
            if (d > 1000 && d < 10000000) { last[0] = g1; push('V', g1, d); }
        } else if (kind == 2) {
            int64_t d = G2 - g1 - g2; if (d < 0) d = -d;
            last[1] = d;
            if (d > 1000 && d < 10000000) last[0] = g2;
            push('V', last[0], last[1]);
        } else if (kind == 3) {
            push('T', g1, dist); push('T', g2, dist);
            last[0] = g2; last[1] = dist;
        } else {
            push('T', G2 - g1, dist); push('T', G2 - g2, dist);
            last[0] = G2 - g2; last[1] = dist;
        }
    }
}
