// mapcaller_amd/csrc/mcx_main.cpp — command line with the reference's flags for the path
// (reference src/main.cpp:154-396): MapCaller -i prefix | -r ref.fa, -f ..., -f2 ..., -alg nw|ksw2,
// -sam out, -vcf out (on by default, like the reference) with the variant-calling switches, plus
// `index ref.fa prefix`.  Host code only: everything heavy goes through mcx.h.
#include "../../include/mcx.h"
#include "../../include/mcx_comm.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
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
            "         -gpus INT     number of GPUs of this node to spread the reads over (devices 0..INT-1) [1]\n"
            "         -devices LIST device ordinals, comma separated (instead of -gpus)\n"
            "         -batch INT    reads per batch (the unit dealt to the GPUs) [524288]\n"
            "         -sampled_sa   keep only the sampled suffix array in HBM (saves 8 bytes per text position, slower seeding)\n"
            "         -two_base     keep the pair records in HBM too (4 bytes per text position: the seeding walk takes two bases per step);\n"
            "                       implied by -no_vcf, which leaves the room\n", prog, prog);
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
    int gpu = 0, full_sa = 1, maxlen = 0, n_gpus = 1;
    long long batch_reads = 0;
    std::vector<int32_t> devices;
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
        else if (p == "-gpus" && i + 1 < argc) n_gpus = std::max(1, atoi(argv[++i]));
        else if (p == "-devices" && i + 1 < argc) { for (char *t = strtok(argv[++i], ","); t; t = strtok(nullptr, ",")) devices.push_back(atoi(t)); }
        else if (p == "-batch" && i + 1 < argc) batch_reads = atoll(argv[++i]);
        else if (p == "-maxlen" && i + 1 < argc) maxlen = atoi(argv[++i]);
        else if (p == "-sampled_sa") full_sa = 0;
        else if (p == "-two_base") full_sa = 2;
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
        else if (p == "-v" || p == "--version") { fprintf(stderr, "MapCaller v0.9.9.41 (mapcaller-mi355x)\n\n"); return 0; } // main.cpp:310-314
        else if (p == "-m" || p == "-bam") { fprintf(stderr, "Error! %s is not supported by mapcaller-mi355x (DESIGN.md, deliberate deviations)\n", argv[i]); return 1; }
        else { fprintf(stderr, "Warning! Unknow parameter: %s\n", argv[i]); usage(argv[0]); return 0; }
    }
    if (f1.empty()) { fprintf(stderr, "Warning! Please specify a valid read input!\n"); usage(argv[0]); return 0; }
    if (!f2.empty() && f1.size() != f2.size()) { fprintf(stderr, "Warning! Paired-end reads input numbers do not match!\n"); return 0; }
    if (devices.empty()) for (int r = 0; r < n_gpus; r++) devices.push_back(n_gpus == 1 ? gpu : r);
    n_gpus = (int)devices.size();
    // -r ref.fa: the index is built for this run and removed again (main.cpp:344-349, :385-391) — on every way out, also the
    // early ones (a failed build leaves partial files behind it)
    struct TmpIndex {
        std::string prefix;
        ~TmpIndex() { if (!prefix.empty()) for (const char *ext : {".bwt", ".sa", ".pac", ".ann", ".amb"}) unlink((prefix + ext).c_str()); }
    } tmp_index;
    if (!ref.empty()) {
        const char *tmp_dir = getenv("TMPDIR");
        tmp_index.prefix = std::string(tmp_dir && tmp_dir[0] ? tmp_dir : "/tmp") + "/mcx_idx_" + std::to_string((long long)getpid());
        int rc = mcx_index_build(ref.c_str(), tmp_index.prefix.c_str(), devices[0]);
        if (rc) { fprintf(stderr, "index: %s (%d)\n", mcx_last_error(), rc); return 1; }
        prefix = tmp_index.prefix;
    }
    if (prefix.empty()) { fprintf(stderr, "Warning! Please specify a valid reference index!\n"); usage(argv[0]); return 0; }
    o.max_batch_reads = batch_reads > 0 ? batch_reads : 1 << 21; // per batch of the parse | map | format pipeline (a batch costs a few ms of its own: fewer, larger ones)
    if (maxlen <= 0) {
        for (const std::string &f : f1) maxlen = std::max(maxlen, sample_read_length(f));
        for (const std::string &f : f2) maxlen = std::max(maxlen, sample_read_length(f));
        maxlen = (maxlen + 63) / 64 * 64;
    }
    o.max_read_len = std::min(1000, std::max(256, maxlen));

    // One shard per GPU, one host thread each (the whole run on this thread when there is one GPU).  The input stream is
    // cut into batches that are dealt to the shards in turn; between them the shards exchange what keeps the run equal
    // to the single-stream one (insert-size trajectory, duplicate cap: mcx_file_opts.exchange); every shard writes
    // its batches' lines at their final place in the one SAM file; with -vcf the counter planes are summed onto shard 0
    // over RCCL (mcx_profile_reduce) and the sparse tallies of all shards go to its VariantCalling().
    struct Shard {
        int rank = 0, device = 0, rc = 0;
        std::string err;
        mcx_index *ix = nullptr; mcx_ctx *cx = nullptr; uint32_t *planes = nullptr;
        mcx_stats st;
        const mcx_sparse_rec *recs = nullptr; uint64_t n_recs = 0;
        double reduce_s = 0;
    };
    std::vector<Shard> shards((size_t)n_gpus);
    std::vector<mcx_exchange> links((size_t)n_gpus);
    std::vector<mcx_comm *> comms((size_t)n_gpus, nullptr);
    if (n_gpus > 1) {
        if (mcx_exchange_local(n_gpus, links.data())) { fprintf(stderr, "Error! %s\n", mcx_last_error()); return 1; }
        if (want_vcf && mcx_comm_init_all(n_gpus, devices.data(), comms.data())) { fprintf(stderr, "Error! %s\n", mcx_last_error()); return 1; }
    }
    const bool paired_run = !f2.empty() || fo.interleaved_pairs;
    if (!want_vcf && full_sa == 1) full_sa = MCX_INDEX_PAIRS_IF_ROOM; // (no planes: room for the pair records, usually — a device that is too full keeps the one-base walk; -two_base insists)
    auto work = [&](int r) {
        Shard &sh = shards[(size_t)r];
        sh.rank = r; sh.device = devices[(size_t)r];
        memset(&sh.st, 0, sizeof sh.st);
        auto bad = [&](int rc) { sh.rc = rc; sh.err = mcx_last_error(); };
        int rc = mcx_index_load(prefix.c_str(), sh.device, full_sa, &sh.ix);
        mcx_fit fit;
        memset(&fit, 0, sizeof fit);
        if (rc == 0) {
            // the context, tier 0's pair records and — MappingRecordArr, main.cpp:366-370 — the planes and the bookkeeping's buffers, all taken here, fitted to
            // what the device has left (a genome larger than GRCh38, -two_base, a long -maxlen: smaller batches instead of a failed allocation in the first batch)
            if (want_vcf && r == 0) fprintf(stderr, "Initialize the alignment profile...\n");
            rc = mcx_ctx_create_fit(sh.ix, &o, want_vcf ? 1 : 0, paired_run ? 1 : 0, vo.max_dup, vo.max_clip, &sh.cx, want_vcf ? &sh.planes : nullptr, &fit);
        }
        if (rc) bad(rc);
        if (n_gpus > 1) { // the shards cut ONE input stream into batches: they must agree on the batch
            int64_t mine = sh.rc ? -1 : fit.max_batch_reads;
            std::vector<int64_t> all((size_t)n_gpus);
            links[(size_t)r].allgather(links[(size_t)r].user, &mine, all.data(), sizeof mine);
            for (int64_t v : all) if (v >= 0 && mine >= 0 && v != mine && !sh.rc) { sh.rc = MCX_ERR_DEVICE; sh.err = "the shards' devices have room for different batch sizes (" + std::to_string((long long)mine) + " / " + std::to_string((long long)v) + " reads): give -batch"; }
        }
        mcx_file_opts my = fo;
        int64_t avg[4];
        mcx_avg_init(avg); // avgDist and its totals are globals of the reference: they carry over from library to library
        my.avg_state = avg;
        if (n_gpus > 1) { my.shard_rank = r; my.shard_count = n_gpus; my.exchange = &links[(size_t)r]; }
        for (size_t k = 0; k < f1.size(); k++) {
            // like the reference, every library appends to the same SAM stream; only the first writes the header
            if (avg[3] % 200) avg[3] += 200 - avg[3] % 200; // a new library starts a new chunk
            // (the shards write into the one SAM file, every batch's text at its final place: mcx_map_files_ex)
            const std::string out = sam;
            my.append_sam = k > 0;
            if (n_gpus > 1) { // a shard that could not even start must not leave the others waiting in the first exchange
                int32_t mine = sh.rc;
                std::vector<int32_t> all((size_t)n_gpus);
                links[(size_t)r].allgather(links[(size_t)r].user, &mine, all.data(), sizeof mine);
                bool stop = false;
                for (int32_t v : all) if (v) stop = true;
                if (stop) { if (!sh.rc) { sh.rc = MCX_ERR_DEVICE; sh.err = "another shard failed"; } break; }
            } else if (sh.rc) break;
            rc = mcx_map_files_ex(sh.cx, f1[k].c_str(), f2.empty() ? nullptr : f2[k].c_str(), &my, out.empty() ? nullptr : out.c_str(), &sh.st);
            if (rc) bad(rc);
            if (n_gpus > 1) { // every part of this library is complete before shard 0 merges them
                int32_t mine = sh.rc;
                std::vector<int32_t> all((size_t)n_gpus);
                links[(size_t)r].allgather(links[(size_t)r].user, &mine, all.data(), sizeof mine);
                bool stop = false;
                for (int32_t v : all) if (v) stop = true;
                if (stop) { if (!sh.rc) { sh.rc = MCX_ERR_DEVICE; sh.err = "another shard failed"; } break; }
            }
        }
        if (want_vcf && sh.rc == 0 && (rc = mcx_profile_settle(sh.cx))) bad(rc); // differences -> counts, before anything sums or reads the planes
        if (n_gpus > 1 && want_vcf) { // the one collective of the run; every shard takes part even after a failure elsewhere
            bool all_ok = true;
            int32_t mine = sh.rc;
            std::vector<int32_t> all((size_t)n_gpus);
            links[(size_t)r].allgather(links[(size_t)r].user, &mine, all.data(), sizeof mine);
            for (int32_t v : all) if (v) all_ok = false;
            if (all_ok && (rc = mcx_profile_reduce(comms[(size_t)r], sh.planes, mcx_index_genome_size(sh.ix), 0, &sh.reduce_s))) bad(rc);
        }
        if (sh.rc == 0 && want_vcf && (rc = (n_gpus > 1 ? mcx_profile_sparse_shard : mcx_profile_sparse)(sh.cx, &sh.recs, &sh.n_recs))) bad(rc);
    };
    if (n_gpus == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int r = 0; r < n_gpus; r++) pool.emplace_back(work, r);
        for (auto &t : pool) t.join();
    }
    int rc = 0;
    mcx_stats st;
    memset(&st, 0, sizeof st);
    for (const Shard &sh : shards) {
        if (sh.rc) { fprintf(stderr, "Error! %s%s\n", sh.err.c_str(), n_gpus > 1 ? (" (GPU " + std::to_string(sh.device) + ")").c_str() : ""); rc = sh.rc; }
        st.reads += sh.st.reads; st.mapped += sh.st.mapped; st.pairs += sh.st.pairs; st.pair_dist_sum += sh.st.pair_dist_sum; st.pair_len_sum += sh.st.pair_len_sum;
    }
    fprintf(stderr, "All the %lld %s reads have been processed.\n%12lld reads are mapped properly.\n%12lld reads are mapped in pairs.\n",
            (long long)st.reads, paired_run ? "paired-end" : "single-end", (long long)st.mapped, (long long)st.pairs * 2);
    if (want_vcf && rc == 0) { // VariantCalling(), main.cpp:379
        std::vector<mcx_sparse_rec> merged;
        const mcx_sparse_rec *recs = shards[0].recs;
        uint64_t n_recs = shards[0].n_recs;
        if (n_gpus > 1) {
            for (const Shard &sh : shards) merged.insert(merged.end(), sh.recs, sh.recs + sh.n_recs);
            recs = merged.data(); n_recs = merged.size();
            if (getenv("MCX_TIMING")) fprintf(stderr, "[mcx_profile_reduce] %.3f s on shard 0\n", shards[0].reduce_s);
        }
        mcx_vcf_stats vs;
        vo.ref_name = ref.empty() ? prefix.c_str() : ref.c_str();
        vo.cmdline = cmdline.c_str();
        fprintf(stderr, "Identify all variants (min_alt_allele_depth=%d)...\n", vo.min_allele_depth);
        if ((rc = mcx_profile_finalize(shards[0].cx, shards[0].planes)) ||
            (rc = mcx_call_variants(shards[0].ix, shards[0].planes, recs, n_recs, st.pairs, st.pair_dist_sum, st.pair_len_sum, &vo, vcf.c_str(), &vs)))
            fprintf(stderr, "Error! %s\n", mcx_last_error());
        else
            fprintf(stderr, "\tWrite all the predicted sample variations to file [%s]...\n\t%lld(snp); %lld(ins); %lld(del); %lld(trans); %lld(inversion)\n",
                    vcf.c_str(), (long long)vs.n_snv, (long long)vs.n_ins, (long long)vs.n_del, (long long)(vs.n_tnl >> 1), (long long)(vs.n_inv >> 1));
    }
    for (Shard &sh : shards) { mcx_planes_free(sh.planes); mcx_ctx_free(sh.cx); mcx_index_free(sh.ix); }
    for (mcx_comm *c : comms) mcx_comm_free(c);
    if (n_gpus > 1) mcx_exchange_local_free(links.data());
    return rc ? 1 : 0;
}
