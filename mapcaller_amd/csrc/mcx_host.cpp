// mapcaller_amd/csrc/mcx_host.cpp — index files, read files and SAM text on the host.
//
// Index files are the byte-compatible BWA files `MapCaller index` writes (reference
// src/BWT_Index/bwtindex.c:77-160, src/BWT_Index/bwt.c:174-196, src/BWT_Index/bntseq.c:60-91);
// the loader mirrors src/bwt_index.cpp:16-124 and :232-258.  Read files follow
// src/GetData.cpp:3-146 (header trimming, 4-line FASTQ records, multi-line FASTA).  SAM lines
// follow the printf formats of src/SamReport.cpp:338,361,405,429,431.
#include "mcx_host.h"
#include "mcx_types.h"

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
// read files
// ---------------------------------------------------------------------------------------------
bool ReadFile::open(const std::string &path, std::string &err)
{
    gzFile g = gzopen(path.c_str(), "rb");
    if (!g) { err = "cannot open " + path; return false; }
    gzbuffer(g, 1 << 20);
    gz_ = g;
    int c = gzgetc(g);
    fastq_ = (c == '@'); // CheckReadFormat, GetData.cpp:22-31
    if (c != -1) gzungetc(c, g);
    have_pending_ = false;
    return true;
}

void ReadFile::close()
{
    if (gz_) gzclose((gzFile)gz_);
    gz_ = nullptr;
}

bool ReadFile::line(std::string &s)
{
    if (have_pending_) { s.swap(pending_); have_pending_ = false; return true; }
    s.clear();
    char buf[4096];
    for (;;) {
        if (!gzgets((gzFile)gz_, buf, sizeof buf)) return !s.empty();
        s += buf;
        if (!s.empty() && s.back() == '\n') return true;
    }
}

// IdentifyHeaderBegPos / IdentifyHeaderEndPos, GetData.cpp:3-20
static std::string header_of(const std::string &l)
{
    const int len = (int)l.size();
    int p1 = len - 1, lim = len > 100 ? 100 : len, p2 = lim - 1;
    for (int i = 1; i < len; i++) if (l[i] != '>' && l[i] != '@') { p1 = i; break; }
    for (int i = 1; i < lim; i++) if (l[i] == ' ' || l[i] == '/' || !isprint((unsigned char)l[i])) { p2 = i; break; }
    return p2 > p1 ? l.substr(p1, p2 - p1) : std::string();
}

bool ReadFile::next(HostRead &r)
{
    std::string l;
    r.name.clear(); r.seq.clear(); r.qual.clear();
    if (!line(l)) return false;
    r.name = header_of(l);
    if (fastq_) {
        if (!line(l)) return false;
        const size_t n = l.size(); // the last byte of the line is dropped (GetData.cpp:48-53)
        r.seq = l.substr(0, n ? n - 1 : 0);
        std::string plus, q;
        line(plus); line(q);
        q.resize(n, '\0');
        r.qual = q.substr(0, n ? n - 1 : 0);
    } else {
        for (;;) {
            if (!line(l)) break;
            if (l[0] == '>') { pending_ = l; have_pending_ = true; break; }
            if (!l.empty()) l.resize(l.size() - 1);
            r.seq += l;
        }
    }
    return !r.seq.empty();
}

// ---------------------------------------------------------------------------------------------
// avgDist feedback
// ---------------------------------------------------------------------------------------------
void avg_replay(const PairOut *po, uint32_t n_pairs, const int64_t avg[4], std::vector<uint32_t> &redo,
                std::vector<int32_t> &redo_est, int64_t avg_out[4])
{
    redo.clear(); redo_est.clear();
    const uint32_t chunk = kReadChunkSize / 2;
    int64_t tp = avg[1], td = avg[2];
    uint32_t cur = (uint32_t)avg[0];
    for (uint32_t p0 = 0; p0 < n_pairs; p0 += chunk) {
        const int32_t e = (int32_t)(cur * 1.5);
        const uint32_t p1 = n_pairs < p0 + chunk ? n_pairs : p0 + chunk;
        for (uint32_t p = p0; p < p1; p++) {
            const PairOut &o = po[p];
            const bool ok = (o.flags & kRescueUsedEst) ? o.est == e : (e >= o.est_lo && e <= o.est_hi);
            if (!ok) { redo.push_back(p); redo_est.push_back(e); }
            if (o.pair_ok) { tp++; td += o.pair_dist; }
        }
        if (tp > 1000) cur = (uint32_t)(int)(1. * td / tp + .5);
    }
    avg_out[0] = cur; avg_out[1] = tp; avg_out[2] = td; avg_out[3] = avg[3];
}

// ---------------------------------------------------------------------------------------------
// SAM text
// ---------------------------------------------------------------------------------------------
static inline char comp_char(char c) // GetComplementaryBase, tools.cpp:3-18
{
    switch (c) {
    case 'A': case 'a': return 'T';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    case 'T': case 't': return 'A';
    default: return 'N';
    }
}

static void revcomp(const std::string &in, std::string &out)
{
    out.resize(in.size());
    for (size_t i = 0, n = in.size(); i < n; i++) out[i] = comp_char(in[n - 1 - i]);
}

void sam_header(const HostIndex &ix, std::string &out) // OutputSamHeaders, ReadMapping.cpp:101-123
{
    out = "@PG\tID:MapCaller\tPN:MapCaller\tVN:0.9.9.41\n";
    char buf[1200];
    for (size_t i = 0; i < ix.chr_name.size(); i++) {
        snprintf(buf, sizeof buf, "@SQ\tSN:%s\tLN:%d\n", ix.chr_name[i].c_str(), ix.chr_len[i]);
        out += buf;
    }
}

void sam_line(const HostIndex &ix, const HostRead &rd, bool mate2_flipped, bool fastq, const AlnRec &rec,
              const uint32_t *cigar, std::string &out)
{
    // The reference reverse-complements mate 2 in place before mapping (ReadMapping.cpp:451) and
    // prints that string for forward-strand hits, its reverse complement otherwise.
    std::string cur_seq, cur_qual, tmp;
    if (mate2_flipped) { revcomp(rd.seq, cur_seq); cur_qual.assign(rd.qual.rbegin(), rd.qual.rend()); }
    else { cur_seq = rd.seq; cur_qual = rd.qual; }
    char num[128];
    out = rd.name;
    const bool mapped = rec.chr >= 0;
    if (!mapped) {
        snprintf(num, sizeof num, "\t%d\t*\t0\t0\t*\t*\t0\t0\t", rec.flag);
        out += num; out += cur_seq; out += '\t'; out += fastq ? cur_qual : std::string("*");
        out += "\tAS:i:0\tXS:i:0";
        return;
    }
    snprintf(num, sizeof num, "\t%d\t", rec.flag); out += num;
    out += ix.chr_name[rec.chr];
    snprintf(num, sizeof num, "\t%lld\t%d\t", (long long)rec.pos, rec.mapq); out += num;
    static const char opc[8] = {'M', 'I', 'D', 'N', 'S', 'H', 'P', '='};
    for (int i = 0; i < rec.n_cigar; i++) { snprintf(num, sizeof num, "%u%c", cigar[i] >> 4, opc[cigar[i] & 7]); out += num; }
    if (rec.has_mate) { snprintf(num, sizeof num, "\t=\t%lld\t%d\t", (long long)rec.mate_pos, rec.tlen); out += num; }
    else out += "\t*\t0\t0\t";
    if (rec.fwd) { out += cur_seq; out += '\t'; out += fastq ? cur_qual : std::string("*"); }
    else {
        revcomp(cur_seq, tmp); out += tmp; out += '\t';
        if (fastq) { tmp.assign(cur_qual.rbegin(), cur_qual.rend()); out += tmp; } else out += '*';
    }
    snprintf(num, sizeof num, "\tNM:i:%d\tAS:i:%d\tXS:i:%d", rec.nm, rec.as, rec.xs); out += num;
}

} // namespace mcx
