// mapcaller_amd/csrc/mcx_main.cpp — command line with the reference's flags for the path
// (reference src/main.cpp:154-396): MapCaller -i prefix | -r ref.fa, -f ..., -f2 ..., -alg nw|ksw2,
// -sam out, plus `index ref.fa prefix`.  Host code only: everything heavy goes through mcx.h.
#include "../../include/mcx.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>

static void usage(const char *prog)
{
    fprintf(stderr,
            "MapCaller seed-and-extend path on MI355X\n\n"
            "Usage: %s -i Index_Prefix -f <ReadFile_A1 ...> [-f2 <ReadFile_A2 ...>] [-alg nw|ksw2] [-sam out.sam]\n"
            "       %s index ref.fa prefix\n\n"
            "Options: -i STR        BWT_Index_Prefix\n"
            "         -r STR        Reference filename (format:fa); an index is built on the GPU first\n"
            "         -f            files with #1 mates reads (format:fa, fq, fq.gz)\n"
            "         -f2           files with #2 mates reads (format:fa, fq, fq.gz)\n"
            "         -alg STR      gapped alignment algorithm (option: nw|ksw2) [nw]\n"
            "         -sam          SAM output filename ('-' = stdout)\n"
            "         -indel INT    maximal indel size [30]\n"
            "         -maxmm FLOAT  maximal mismatch rate in read alignment [0.05]\n"
            "         -vcf / -no_vcf  variant calling is outside this path: -vcf is rejected, -no_vcf accepted\n"
            "         -t INT        accepted and ignored (the GPU path has no worker threads)\n"
            "         -gpu INT      device ordinal [0]\n", prog, prog);
}

int main(int argc, char **argv)
{
    if (argc == 1 || strcmp(argv[1], "-h") == 0) { usage(argv[0]); return 0; }
    if (strcmp(argv[1], "index") == 0) {
        if (argc != 4) { fprintf(stderr, "usage: %s index ref.fa prefix\n", argv[0]); return 0; }
        int rc = mcx_index_build(argv[2], argv[3], 0);
        if (rc) { fprintf(stderr, "index: %s (%d)\n", mcx_last_error(), rc); return 1; }
        return 0;
    }
    std::string prefix, ref, sam;
    std::vector<std::string> f1, f2;
    mcx_opts o;
    mcx_opts_default(&o);
    int gpu = 0;
    bool want_vcf = false, no_vcf = false;
    for (int i = 1; i < argc; i++) {
        std::string p = argv[i];
        if (p == "-i" && i + 1 < argc) prefix = argv[++i];
        else if (p == "-r" && i + 1 < argc) ref = argv[++i];
        else if (p == "-f") { while (++i < argc && argv[i][0] != '-') f1.push_back(argv[i]); i--; }
        else if (p == "-f2") { while (++i < argc && argv[i][0] != '-') f2.push_back(argv[i]); i--; }
        else if (p == "-alg" && i + 1 < argc) o.alg = strcmp(argv[++i], "ksw2") == 0 ? 1 : 0;
        else if (p == "-sam" && i + 1 < argc) sam = argv[++i];
        else if (p == "-indel" && i + 1 < argc) { o.max_pos_diff = atoi(argv[++i]); if (o.max_pos_diff > 100) { o.max_pos_diff = 100; fprintf(stderr, "Warning! The maximal indel size is 100!\n"); } }
        else if (p == "-maxmm" && i + 1 < argc) o.max_mismatch_rate = (float)atof(argv[++i]);
        else if (p == "-t" && i + 1 < argc) ++i;
        else if (p == "-gpu" && i + 1 < argc) gpu = atoi(argv[++i]);
        else if (p == "-vcf" && i + 1 < argc) { ++i; want_vcf = true; }
        else if (p == "-no_vcf") no_vcf = true;
        else { fprintf(stderr, "Warning! Unknow parameter: %s\n", argv[i]); usage(argv[0]); return 0; }
    }
    (void)no_vcf;
    if (want_vcf) { fprintf(stderr, "-vcf: variant calling is not part of the accelerated path yet; run with -no_vcf\n"); return 1; }
    if (f1.empty()) { fprintf(stderr, "Warning! Please specify a valid read input!\n"); usage(argv[0]); return 0; }
    if (!f2.empty() && f1.size() != f2.size()) { fprintf(stderr, "Warning! Paired-end reads input numbers do not match!\n"); return 0; }
    std::string tmp_prefix;
    if (!ref.empty()) {
        tmp_prefix = "/tmp/mcx_idx_" + std::to_string((long long)getpid());
        int rc = mcx_index_build(ref.c_str(), tmp_prefix.c_str(), gpu);
        if (rc) { fprintf(stderr, "index: %s (%d)\n", mcx_last_error(), rc); return 1; }
        prefix = tmp_prefix;
    }
    if (prefix.empty()) { fprintf(stderr, "Warning! Please specify a valid reference index!\n"); usage(argv[0]); return 0; }
    mcx_index *ix = nullptr;
    int rc = mcx_index_load(prefix.c_str(), gpu, 0, &ix);
    if (rc) { fprintf(stderr, "Error! %s\n", mcx_last_error()); return 1; }
    o.max_batch_reads = 1 << 20;
    mcx_ctx *cx = nullptr;
    rc = mcx_ctx_create(ix, &o, &cx);
    if (rc) { fprintf(stderr, "Error! %s\n", mcx_last_error()); return 1; }
    mcx_stats st;
    memset(&st, 0, sizeof st);
    for (size_t k = 0; k < f1.size() && rc == 0; k++) {
        // like the reference, every library appends to the same SAM stream; only the first writes the header
        rc = mcx_map_files(cx, f1[k].c_str(), f2.empty() ? nullptr : f2[k].c_str(), sam.empty() ? nullptr : sam.c_str(), &st);
        if (rc) fprintf(stderr, "Error! %s\n", mcx_last_error());
    }
    fprintf(stderr, "All the %lld %s reads have been processed.\n%12lld reads are mapped properly.\n%12lld reads are mapped in pairs.\n",
            (long long)st.reads, f2.empty() ? "single-end" : "paired-end", (long long)st.mapped, (long long)st.pairs * 2);
    mcx_ctx_free(cx);
    mcx_index_free(ix);
    if (!tmp_prefix.empty()) { std::string cmd = "rm -f " + tmp_prefix + ".*"; if (system(cmd.c_str())) {} }
    return rc ? 1 : 0;
}
