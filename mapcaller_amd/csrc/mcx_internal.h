// mapcaller_amd/csrc/mcx_internal.h — what the translation units of libmcx.so share (not part of the ABI)
#ifndef MCX_INTERNAL_H
#define MCX_INTERNAL_H
#include "mcx_types.h"
#include "mcx_host.h"
#include "mcx_build.h"

struct mcx_index {
    mcx::IndexView view;
    mcx::HostIndex host;
    int device = 0;
    void *d_bwt = nullptr, *d_sa = nullptr, *d_sa_full = nullptr, *d_pac = nullptr;
    void *d_end_pos = nullptr, *d_end_chr = nullptr, *d_chr_fwd = nullptr, *d_ktab = nullptr;
    int64_t hbm_bytes = 0;
    uint64_t n_bwt_words = 0, n_sa = 0; // set for indexes built in HBM (mcx_index_from_codes)
};

#endif
