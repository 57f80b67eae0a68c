// oracle/mcx_oracle_main.cpp — TEST INFRASTRUCTURE ONLY: command-line front end of the CPU
// restatement, used to produce SAM for diffing against oracle/_ref/MapCaller and as the
// "port" CPU baseline of bench.py when the compiled reference is absent.
//   mcx_oracle -i <index prefix> -f r1.fq [-f2 r2.fq] [-alg nw|ksw2] [-sam out.sam] [-t N]
//   mcx_oracle -i <index prefix> -f r1.fq [-f2 r2.fq] [-alg nw|ksw2] -vcf out.vcf [-gvcf] [-monomorphic] [-filter]
//              [-somatic] [-ploidy N] [-ad N] [-min_cnv N] [-min_gap N] [-size N] [-dup N] [-maxclip N] [-id name]
#include "mcx_oracle.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>

int main(int argc, char **argv)
{
    std::string prefix, f1, f2, sam, vcf;
    int alg = 0, threads = 1;
    mcxo_vcf_opts vo;
    mcxo_vcf_defaults(&vo);
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "-i" && i + 1 < argc) prefix = argv[++i];
        else if (a == "-f" && i + 1 < argc) f1 = argv[++i];
        else if (a == "-f2" && i + 1 < argc) f2 = argv[++i];
        else if (a == "-sam" && i + 1 < argc) sam = argv[++i];
        else if (a == "-t" && i + 1 < argc) threads = atoi(argv[++i]);
        else if (a == "-vcf" && i + 1 < argc) vcf = argv[++i];
        else if (a == "-gvcf") vo.gvcf = 1;
        else if (a == "-monomorphic") vo.monomorphic = 1;
        else if (a == "-filter") vo.filter = 1;
        else if (a == "-somatic") vo.somatic = 1;
        else if (a == "-ploidy" && i + 1 < argc) { if ((vo.ploidy = atoi(argv[++i])) > 2) vo.ploidy = 2; }
        else if (a == "-ad" && i + 1 < argc) vo.min_allele_depth = atoi(argv[++i]);
        else if (a == "-min_cnv" && i + 1 < argc) vo.min_cnv = atoi(argv[++i]);
        else if (a == "-min_gap" && i + 1 < argc) vo.min_gap = atoi(argv[++i]);
        else if (a == "-size" && i + 1 < argc) vo.fragment_size = atoi(argv[++i]);
        else if (a == "-dup" && i + 1 < argc) { if (atoi(argv[++i]) <= 15) vo.max_dup = (int8_t)atoi(argv[i]); }
        else if (a == "-maxclip" && i + 1 < argc) vo.max_clip = atoi(argv[++i]);
        else if ((a == "-id" || a == "-label") && i + 1 < argc) vo.sample_id = argv[++i];
        else if (a == "-alg" && i + 1 < argc) alg = strcmp(argv[++i], "ksw2") == 0 ? 1 : 0;
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (prefix.empty() || f1.empty()) { fprintf(stderr, "usage: %s -i prefix -f r1.fq [-f2 r2.fq] [-alg nw|ksw2] [-sam out] [-t N]\n", argv[0]); return 2; }
    mcxo_index *ix = mcxo_index_load(prefix.c_str());
    if (!ix) { fprintf(stderr, "cannot load index %s\n", prefix.c_str()); return 1; }
    if (!vcf.empty()) {
        vo.ref_name = prefix.c_str();
        int64_t n = mcxo_map_files_vcf(ix, f1.c_str(), f2.c_str(), alg, vcf.c_str(), &vo);
        mcxo_index_free(ix);
        return n < 0;
    }
    int64_t st[8];
    auto t0 = std::chrono::steady_clock::now();
    int64_t n = mcxo_map_files(ix, f1.c_str(), f2.c_str(), alg, sam.c_str(), threads, st);
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "reads=%lld mapped=%lld pairs=%lld E=%lld H=%lld LF=%lld dp_calls=%lld dp_cells=%lld sec=%.3f reads_per_s=%.0f\n",
            (long long)n, (long long)st[1], (long long)st[2], (long long)st[3], (long long)st[4], (long long)st[5],
            (long long)st[6], (long long)st[7], sec, n / sec);
    mcxo_index_free(ix);
    return n < 0;
}
