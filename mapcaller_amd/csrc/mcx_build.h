// mapcaller_amd/csrc/mcx_build.h — interface between the GPU index builder and the pipeline
#ifndef MCX_BUILD_H
#define MCX_BUILD_H
#include <stdint.h>
#include <string>

struct DevIndexArrays {         // device arrays in the on-disk layout of the reference's index
    uint32_t *bwt = nullptr; uint64_t bwt_words = 0;   // words after the 40-byte .bwt header
    uint64_t *sa = nullptr; uint64_t n_sa = 0;         // sampled every 32 rows, sa[0] = ~0
    uint64_t *sa_full = nullptr;                       // optional: all N + 1 rows
    uint64_t primary = 0, L2[5] = {0, 0, 0, 0, 0}, seq_len = 0;
};

// d_fwd: forward genome, one code (0..3) per byte, in HBM
int mcx_build_suffix_index(const uint8_t *d_fwd, uint64_t G, bool want_full_sa, DevIndexArrays &out, double *seconds);
int mcx_set_error(int code, const std::string &msg);
// the pair records of the seeding walk (mcx_fm.h PairSlot) for an index whose full suffix array, packed genome and .bwt blocks are in
// HBM: fills v.rank2 / rank2_c2 / rank2_lone / rank2_t0 (d_rec, d_c2: what to free); *bytes = 0 and nothing set when some pair of bases
// occurs 2^32 times or more.  check_trials > 0: that many random intervals extended both ways on the device, an error if any differs.
namespace mcx { struct IndexView; }
int mcx_build_pair_records(mcx::IndexView &v, void **d_rec, void **d_c2, int64_t *bytes, int check_trials);
#endif
