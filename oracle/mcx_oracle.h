// oracle/mcx_oracle.h — TEST INFRASTRUCTURE ONLY (checker, never the thing measured or shipped).
//
// C surface of the CPU restatement of MapCaller's seed-and-extend path (oracle/mcx_oracle.cpp).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
// Parity status: PINNED — every function below is checked against the real reference compiled
// into oracle/_ref (tests/test_oracle_vs_ref.py) and against the golden vectors in tests/golden.
#ifndef MCX_ORACLE_H
#define MCX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcxo_index mcxo_index;

// Loads <prefix>.bwt/.sa/.pac/.ann/.amb (reference src/bwt_index.cpp:150-258). NULL on failure.
mcxo_index *mcxo_index_load(const char *prefix);
void mcxo_index_free(mcxo_index *);
int64_t mcxo_genome_size(const mcxo_index *);

// BWT_Search (reference src/bwt_search.cpp:121-164). seq: codes 0..4; loc: room for 50.
int mcxo_bwt_search(const mcxo_index *, const uint8_t *seq, int start, int stop, int *len, int *freq, uint64_t *loc);
// Same, but also returns the final forward interval start x0 and size x2 (for kernel debugging).
int mcxo_bwt_search_iv(const mcxo_index *, const uint8_t *seq, int start, int stop, int *len, uint64_t *x0, uint64_t *x2);
// bwt_sa (reference src/bwt_search.cpp:109-119)
uint64_t mcxo_bwt_sa(const mcxo_index *, uint64_t k);
// bwt_occ4 (reference src/bwt_search.cpp:49-66)
void mcxo_occ4(const mcxo_index *, uint64_t k, uint64_t cnt[4]);

// ksw2_alignment / nw_alignment (reference src/ksw2_alignment.cpp:250, src/nw_alignment.cpp:18).
// s1 = read fragment (m chars), s2 = genome fragment (n chars), ASCII. o1/o2 receive the gapped
// strings (NUL-terminated, capacity cap). Returns aligned length, <0 on overflow.
int mcxo_ksw2(const char *s1, int m, const char *s2, int n, char *o1, char *o2, int cap);
int mcxo_nw(const char *s1, int m, const char *s2, int n, char *o1, char *o2, int cap);
// ksw_extz2_sse restated cell by cell: codes 0..4 in, reversed M/I/D op string and ez.score out.
int mcxo_ksw2_extz(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int *score, char *ops, int cap);

// Whole path: MapCaller -i <prefix> -f fq1 [-f2 fq2] -alg nw|ksw2 -sam <sam> -no_vcf -t 1.
// alg: 0 = nw, 1 = ksw2. threads>1 pulls 200-read chunks like the reference (line order then
// differs; avgDist trajectory is no longer the -t 1 one). Returns reads processed, <0 on error.
// stats (may be NULL): [0] reads, [1] mapped, [2] paired, [3] FM extension steps (E),
// [4] SA hits resolved (H), [5] LF steps, [6] DP calls, [7] DP cells.
int64_t mcxo_map_files(const mcxo_index *, const char *fq1, const char *fq2, int alg, const char *sam_path,
                       int threads, int64_t *stats);

// The run totals VariantCalling() takes over from Mapping() (reference src/ReadMapping.cpp:782-790):
// out = {iTotalPairedNum, TotalPairedDistance, ReadLengthSum}.  Returns reads processed.
int64_t mcxo_pair_totals(const mcxo_index *, const char *fq1, const char *fq2, int alg, int64_t out[3]);

// MapCaller -p: both mates alternate in one file (reference src/main.cpp:300, src/GetData.cpp:85-99)
int64_t mcxo_map_files_interleaved(const mcxo_index *, const char *fq, int alg, const char *sam_path, int64_t *stats);

// The same run with the -vcf bookkeeping of UpdateProfile / UpdateMultiHitCount (reference
// src/AlignmentProfile.cpp:41-271) and the discordant-site lists (src/ReadMapping.cpp:486-521):
// writes <out>.prof and <out>.maps in the format of oracle/_ref/mcref_tool's P command.
int64_t mcxo_map_files_profile(const mcxo_index *, const char *fq1, const char *fq2, int alg, const char *out_prefix);

// The same run followed by VariantCalling() (reference src/VariantCalling.cpp:696-740): writes the VCF
// of `MapCaller ... -vcf <vcf_path> -t 1`.  Option fields carry the reference's defaults when
// filled by mcxo_vcf_defaults (src/main.cpp:157-187).  The ##reference and ##command_line header
// lines hold whatever ref_name / cmdline say.
typedef struct mcxo_vcf_opts {
    int ploidy, min_allele_depth, min_cnv, min_gap, fragment_size; // -ploidy -ad -min_cnv -min_gap -size
    int filter, gvcf, monomorphic, somatic;                        // -filter -gvcf -monomorphic -somatic
    int max_dup, max_clip;                                         // -dup -maxclip
    float freq_thr;                                                // FrequencyThr (0.2, no option)
    const char *sample_id, *ref_name, *cmdline;                    // -id
} mcxo_vcf_opts;
void mcxo_vcf_defaults(mcxo_vcf_opts *);
int64_t mcxo_map_files_vcf(const mcxo_index *, const char *fq1, const char *fq2, int alg, const char *vcf_path, const mcxo_vcf_opts *);

#ifdef __cplusplus
}
#endif
#endif
