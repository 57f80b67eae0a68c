// mapcaller_amd/csrc/mcx_internal.h — what the translation units of libmcx.so share (not part of the ABI)
#ifndef MCX_INTERNAL_H
#define MCX_INTERNAL_H
#include "mcx_types.h"
#include "mcx_host.h"
#include "mcx_build.h"
#include "../../include/mcx.h"
#include <atomic>

struct mcx_index {
    mcx::IndexView view;
    mcx::HostIndex host;
    int device = 0;
    void *d_bwt = nullptr, *d_sa = nullptr, *d_sa_full = nullptr, *d_pac = nullptr;
    void *d_end_pos = nullptr, *d_end_chr = nullptr, *d_chr_fwd = nullptr, *d_ktab = nullptr, *d_rank = nullptr;
    void *d_rank2 = nullptr, *d_rank2_c2 = nullptr; // pair records (mcx_fm.h PairSlot): built when the index is made with full_sa = 2
    int64_t rank2_bytes = 0;
    int pair_records = 0;
    int64_t hbm_bytes = 0;
    uint64_t n_bwt_words = 0, n_sa = 0; // set for indexes built in HBM (mcx_index_from_codes)
    mutable std::atomic<int> n_ctx{0};  // contexts alive on this index (mcx_ctx_create / mcx_ctx_free): mcx_index_trim refuses while there are any
    mutable std::atomic<bool> orphan{false}; // mcx_index_free was called while contexts were alive: the last mcx_ctx_free deletes this object
};

// what the file front end (mcx_files.cpp) needs to know about a context
const mcx_index *mcx_ctx_index(const mcx_ctx *);
int mcx_ctx_max_read_len(const mcx_ctx *);
uint64_t mcx_ctx_max_reads(const mcx_ctx *);
// host buffers <-> the context's staging arrays in HBM, on the context's stream (stage_out waits for it)
int mcx_stage_in(mcx_ctx *, const uint8_t *bases, const uint32_t *off, uint32_t n_reads, const uint8_t **d_bases, const uint32_t **d_off,
                 mcx_aln **d_aln, uint32_t **d_cigar);
int mcx_stage_out(mcx_ctx *, uint32_t n_reads, mcx_aln *aln, uint32_t *cigar);
bool mcx_ctx_has_profile(const mcx_ctx *);
// something the file front end keeps with the context from call to call (its page-locked batch buffers): *slot, freed with
// `drop` when the context goes
void **mcx_ctx_files_slot(mcx_ctx *, void (*drop)(void *));
void *mcx_pinned_alloc(size_t bytes); // page-locked host memory (null on failure); mcx_pinned_free accepts null
void mcx_pinned_free(void *);

#endif
