// integration/mapcaller_dropin.cpp — what a MapCaller maintainer adds to link the reference's own
// driver against libmcx.so: definitions of the three seam functions with the reference's exact
// signatures (reference src/structure.h:279, :289, :292), replacing src/bwt_search.o,
// src/nw_alignment.o and src/ksw2_alignment.o at link time.  Compile with the reference's
// structure.h on the include path:
//
//   g++ -O2 -I<MapCaller>/src -I<this repo>/include -c mapcaller_dropin.cpp
//   g++ main.o GetData.o ... (all reference objects except the three above) mapcaller_dropin.o \
//       BWT_Index/libbwa.a htslib/libhts.a -L<this repo>/mapcaller_amd -lmcx -lz -lm -lpthread
//
// Per-call use pays a launch + copy per call; it exists to prove the boundary (oracle/Makefile
// target `dropin` builds exactly this against the compiled reference objects and
// tests/test_gpu_parity.py diffs its SAM).  The fast path is the batch API (mcx_map_batch).
#include "structure.h"
#include "mcx.h"

#include <mutex>

static std::mutex g_lock;          // one GPU context, many reference worker threads
static mcx_index *g_index = nullptr;
static mcx_ctx *g_ctx[2] = {nullptr, nullptr};

static mcx_ctx *context(int alg)
{
    if (!g_index) {
        // the reference keeps the index prefix in the global IndexFileName (src/main.cpp:216, :350)
        if (mcx_index_load(IndexFileName, 0, 0, &g_index) != 0) {
            fprintf(stderr, "mcx: %s\n", mcx_last_error());
            exit(1);
        }
    }
    if (!g_ctx[alg]) {
        mcx_opts o;
        mcx_opts_default(&o);
        o.alg = alg; o.max_pos_diff = MaxPosDiff; o.max_batch_reads = 1024; o.max_read_len = 1000;
        if (mcx_ctx_create(g_index, &o, &g_ctx[alg]) != 0) {
            fprintf(stderr, "mcx: %s\n", mcx_last_error());
            exit(1);
        }
    }
    return g_ctx[alg];
}

// src/bwt_search.cpp:121 — caller owns LocArr (delete[] at src/ReadMapping.cpp:147) when freq > 0
bwtSearchResult_t BWT_Search(uint8_t *seq, int start, int stop)
{
    std::lock_guard<std::mutex> guard(g_lock);
    bwtSearchResult_t r;
    uint32_t off[2] = {0, (uint32_t)stop};
    int32_t st = start, len = 0, freq = 0;
    uint64_t loc[50];
    if (mcx_bwt_search_batch(context(0), seq, off, &st, 1, &len, &freq, loc) != 0) {
        fprintf(stderr, "mcx: %s\n", mcx_last_error());
        exit(1);
    }
    r.len = len; r.freq = freq; r.LocArr = NULL;
    if (freq > 0) {
        r.LocArr = new bwtint_t[freq];
        for (int i = 0; i < freq; i++) r.LocArr[i] = loc[i];
    }
    return r;
}

static void extend(int alg, int m, string &s1, int n, string &s2)
{
    std::lock_guard<std::mutex> guard(g_lock);
    uint32_t qo[2] = {0, (uint32_t)m}, to[2] = {0, (uint32_t)n};
    std::vector<uint8_t> ops(m + n + 16);
    int32_t ops_len = 0, score = 0;
    if (mcx_extend_batch(context(alg), alg, (const uint8_t *)s1.data(), qo, (const uint8_t *)s2.data(), to, 1, ops.data(), &ops_len, &score) != 0) {
        fprintf(stderr, "mcx: %s\n", mcx_last_error());
        exit(1);
    }
    // turn the column string back into the two gapped strings the caller expects
    string a, b;
    int i = 0, j = 0;
    for (int k = 0; k < ops_len; k++) {
        if (ops[k] == 'M') { a.push_back(s1[i++]); b.push_back(s2[j++]); }
        else if (ops[k] == 'I') { a.push_back(s1[i++]); b.push_back('-'); }
        else { a.push_back('-'); b.push_back(s2[j++]); }
    }
    s1.swap(a); s2.swap(b);
}

void nw_alignment(int m, string &s1, int n, string &s2) { extend(0, m, s1, n, s2); }   // src/nw_alignment.cpp:18
void ksw2_alignment(int m, string &s1, int n, string &s2) { extend(1, m, s1, n, s2); } // src/ksw2_alignment.cpp:250
