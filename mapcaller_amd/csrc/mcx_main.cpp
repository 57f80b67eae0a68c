// mapcaller_amd/csrc/mcx_main.cpp — command line with the reference's flags for the path
// (reference src/main.cpp:154-396): MapCaller -i prefix | -r ref.fa, -f ..., -f2 ..., -alg nw|ksw2,
// -sam out, -vcf out (on by default, like the reference) with the variant-calling switches, plus
// `index ref.fa prefix`.  Host code only: everything heavy goes through mcx.h.
#include "../../include/mcx.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>
#include <zlib.h>

// longest sequence line among the first records of a read file (FASTQ: every 4th line from the 2nd;
// FASTA: any non-header line) — sizes the contexts; a longer read further down is an error that names -maxlen
static int sample_read_length(const std::string &path)
{
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) return 0;
    static char line[1 << 16];
    int longest = 0, n = 0;
    bool fastq = false;
    while (n < 40000 && gzgets(f, line, sizeof line)) {
        const int len = (int)strcspn(line, "\r\n");
        if (n == 0) fastq = line[0] == '@';
        if (fastq ? (n % 4 == 1) : (line[0] != '>')) longest = len > longest ? len : longest;
        n++;
    }
    gzclose(f);
    return longest;
}

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
            "         -lib STR      file with one 'reads1 [reads2]' pair of filenames per line\n"
            "         -alg STR      gapped alignment algorithm (option: nw|ksw2) [nw]\n"
            "         -sam          SAM output filename ('-' = stdout)\n"
            "         -indel INT    maximal indel size [30]\n"
            "         -maxmm FLOAT  maximal mismatch rate in read alignment [0.05]\n"
            "         -vcf          VCF output filename [output.vcf]\n"
            "         -no_vcf       No VCF output\n"
            "         -gvcf         GVCF mode\n"
            "         -monomorphic  report all loci which do not have any potential alternates\n"
            "         -ploidy INT   number of sets of chromosomes in a cell (1:monoploid, 2:diploid) [2]\n"
            "         -size INT     sequencing fragment size [500]\n"
            "         -ad INT       minimal ALT allele count [5]\n"
            "         -dup INT      maximal PCR duplicates [5]\n"
            "         -maxclip INT  maximal clip size at either ends [5]\n"
            "         -min_cnv INT  minimal cnv size to be reported [50]\n"
            "         -min_gap INT  minimal gap(unmapped) size to be reported [50]\n"
            "         -filter       apply variant filters (under test)\n"
            "         -somatic      detect somatic mutations\n"
            "         -id STR       sample id [unknown]\n"
            "         -p            paired-end reads are interlaced in the same file\n"
            "         -t INT        host threads that parse reads and format SAM lines [half the cores, at most 32]\n"
            "         -maxlen INT   longest read the run has to take [sampled from the first reads, at least 256, at most 1000]\n"
            "         -gpu INT      device ordinal [0]\n"
            "         -sampled_sa   keep only the sampled suffix array in HBM (saves 8 bytes per text position, slower seeding)\n", prog, prog);
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
    int gpu = 0, full_sa = 1, maxlen = 0;
    mcx_file_opts fo;
    mcx_file_opts_default(&fo);
    bool want_vcf = true; // bVCFoutput, main.cpp:171
    std::string vcf = "output.vcf", cmdline = argv[0];
    mcx_vcf_opts vo;
    mcx_vcf_defaults(&vo);
    for (int i = 1; i < argc; i++) cmdline += std::string(" ") + argv[i];
    for (int i = 1; i < argc; i++) {
        std::string p = argv[i];
        if (p == "-i" && i + 1 < argc) prefix = argv[++i];
        else if (p == "-r" && i + 1 < argc) ref = argv[++i];
        else if (p == "-f") { while (++i < argc && argv[i][0] != '-') f1.push_back(argv[i]); i--; }
        else if (p == "-f2") { while (++i < argc && argv[i][0] != '-') f2.push_back(argv[i]); i--; }
        else if (p == "-lib" && i + 1 < argc) { // ReadLibInput, main.cpp:136-152: one "file1 [file2]" per line, '#' comments, an empty line ends the list
            FILE *lf = fopen(argv[++i], "r");
            char line[4096], a[2048], b[2048];
            while (lf && fgets(line, sizeof line, lf)) {
                if (line[0] == '\n' || line[0] == '\0') break;
                if (line[0] == '#') continue;
                a[0] = b[0] = 0;
                sscanf(line, "%2047s %2047s", a, b);
                if (a[0]) f1.push_back(a);
                if (b[0]) f2.push_back(b);
            }
            if (lf) fclose(lf);
        }
        else if (p == "-alg" && i + 1 < argc) o.alg = strcmp(argv[++i], "ksw2") == 0 ? 1 : 0;
        else if (p == "-sam" && i + 1 < argc) sam = argv[++i];
        else if (p == "-indel" && i + 1 < argc) { o.max_pos_diff = atoi(argv[++i]); if (o.max_pos_diff > 100) { o.max_pos_diff = 100; fprintf(stderr, "Warning! The maximal indel size is 100!\n"); } }
        else if (p == "-maxmm" && i + 1 < argc) o.max_mismatch_rate = (float)atof(argv[++i]);
        else if (p == "-t" && i + 1 < argc) { if ((fo.host_threads = atoi(argv[++i])) <= 0) { fprintf(stderr, "Warning! The thread number should be positive!\n"); fo.host_threads = 4; } }
        else if (p == "-pair" || p == "-p") fo.interleaved_pairs = 1;
        else if (p == "-gpu" && i + 1 < argc) gpu = atoi(argv[++i]);
        else if (p == "-maxlen" && i + 1 < argc) maxlen = atoi(argv[++i]);
        else if (p == "-sampled_sa") full_sa = 0;
        else if (p == "-vcf" && i + 1 < argc) vcf = argv[++i];
        else if (p == "-no_vcf") want_vcf = false;
        else if (p == "-gvcf") vo.gvcf = 1;
        else if (p == "-monomorphic") vo.monomorphic = 1;
        else if (p == "-filter") vo.filter = 1;
        else if (p == "-somatic") vo.somatic = 1;
        else if (p == "-ploidy" && i + 1 < argc) { if ((vo.ploidy = atoi(argv[++i])) > 2) { vo.ploidy = 2; fprintf(stderr, "Warning! MapCaller only supports monoploid and diploid!\n"); } }
        else if (p == "-size" && i + 1 < argc) vo.fragment_size = atoi(argv[++i]);
        else if (p == "-ad" && i + 1 < argc) vo.min_allele_depth = atoi(argv[++i]);
        else if (p == "-min_cnv" && i + 1 < argc) vo.min_cnv = atoi(argv[++i]);
        else if (p == "-min_gap" && i + 1 < argc) vo.min_gap = atoi(argv[++i]);
        else if (p == "-maxclip" && i + 1 < argc) vo.max_clip = atoi(argv[++i]);
        else if (p == "-dup" && i + 1 < argc) { if (atoi(argv[++i]) <= 15) vo.max_dup = (int8_t)atoi(argv[i]); else fprintf(stderr, "Warning! The PCR-duplicate range is [1-15]!\n"); }
        else if ((p == "-id" || p == "-label") && i + 1 < argc) vo.sample_id = argv[++i];
        else if (p == "-log" && i + 1 < argc) ++i;
        else { fprintf(stderr, "Warning! Unknow parameter: %s\n", argv[i]); usage(argv[0]); return 0; }
    }
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
    int rc = mcx_index_load(prefix.c_str(), gpu, full_sa, &ix);
    if (rc) { fprintf(stderr, "Error! %s\n", mcx_last_error()); return 1; }
    o.max_batch_reads = 1 << 19; // per batch of the parse | map | format pipeline
    if (maxlen <= 0) {
        for (const std::string &f : f1) maxlen = std::max(maxlen, sample_read_length(f));
        for (const std::string &f : f2) maxlen = std::max(maxlen, sample_read_length(f));
        maxlen = (maxlen + 63) / 64 * 64;
    }
    o.max_read_len = std::min(1000, std::max(256, maxlen));
    mcx_ctx *cx = nullptr;
    rc = mcx_ctx_create(ix, &o, &cx);
    if (rc) { fprintf(stderr, "Error! %s\n", mcx_last_error()); return 1; }
    mcx_stats st;
    memset(&st, 0, sizeof st);
    uint32_t *planes = nullptr;
    if (want_vcf) { // MappingRecordArr, main.cpp:366-370
        fprintf(stderr, "Initialize the alignment profile...\n");
        if ((rc = mcx_planes_alloc(ix, &planes)) || (rc = mcx_profile_attach(cx, planes, vo.max_dup, vo.max_clip))) { fprintf(stderr, "Error! %s\n", mcx_last_error()); return 1; }
    }
    int64_t avg[4];
    mcx_avg_init(avg); // avgDist and its totals are globals of the reference: they carry over from library to library
    fo.avg_state = avg;
    for (size_t k = 0; k < f1.size() && rc == 0; k++) {
        // like the reference, every library appends to the same SAM stream; only the first writes the header
        fo.append_sam = k > 0;
        if (avg[3] % 200) avg[3] += 200 - avg[3] % 200; // a new library starts a new chunk
        rc = mcx_map_files_ex(cx, f1[k].c_str(), f2.empty() ? nullptr : f2[k].c_str(), &fo, sam.empty() ? nullptr : sam.c_str(), &st);
        if (rc) fprintf(stderr, "Error! %s\n", mcx_last_error());
    }
    fprintf(stderr, "All the %lld %s reads have been processed.\n%12lld reads are mapped properly.\n%12lld reads are mapped in pairs.\n",
            (long long)st.reads, (f2.empty() && !fo.interleaved_pairs) ? "single-end" : "paired-end", (long long)st.mapped, (long long)st.pairs * 2);
    if (want_vcf && rc == 0) { // VariantCalling(), main.cpp:379
        const mcx_sparse_rec *recs = nullptr;
        uint64_t n_recs = 0;
        mcx_vcf_stats vs;
        vo.ref_name = ref.empty() ? prefix.c_str() : ref.c_str();
        vo.cmdline = cmdline.c_str();
        fprintf(stderr, "Identify all variants (min_alt_allele_depth=%d)...\n", vo.min_allele_depth);
        if ((rc = mcx_profile_finalize(cx, planes)) || (rc = mcx_profile_sparse(cx, &recs, &n_recs)) ||
            (rc = mcx_call_variants(ix, planes, recs, n_recs, st.pairs, st.pair_dist_sum, st.pair_len_sum, &vo, vcf.c_str(), &vs)))
            fprintf(stderr, "Error! %s\n", mcx_last_error());
        else
            fprintf(stderr, "\tWrite all the predicted sample variations to file [%s]...\n\t%lld(snp); %lld(ins); %lld(del); %lld(trans); %lld(inversion)\n",
                    vcf.c_str(), (long long)vs.n_snv, (long long)vs.n_ins, (long long)vs.n_del, (long long)(vs.n_tnl >> 1), (long long)(vs.n_inv >> 1));
    }
    mcx_planes_free(planes);
    mcx_ctx_free(cx);
    mcx_index_free(ix);
    if (!tmp_prefix.empty()) { std::string cmd = "rm -f " + tmp_prefix + ".*"; if (system(cmd.c_str())) {} }
    return rc ? 1 : 0;
}
