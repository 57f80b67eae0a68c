// mapcaller_amd/csrc/mcx_host.h — host-side pieces around the GPU path: the index files
// (reference src/bwt_index.cpp) and the SAM header.
#ifndef MCX_HOST_H
#define MCX_HOST_H
#include <stdint.h>
#include <string>
#include <vector>

namespace mcx {

struct HostIndex {
    uint64_t primary = 0, L2[5] = {0, 0, 0, 0, 0}, seq_len = 0;
    int sa_intv = 32;
    int64_t G = 0;
    std::vector<uint32_t> bwt;
    std::vector<uint64_t> sa;
    std::vector<uint8_t> pac;
    std::vector<std::string> chr_name;
    std::vector<int32_t> chr_len;
    std::vector<int64_t> chr_fwd;
    std::vector<int64_t> end_pos;
    std::vector<int32_t> end_chr;
};

bool host_index_load(const std::string &prefix, HostIndex &out, std::string &err);
void host_index_finish(HostIndex &ix); // fills chr_fwd / end_pos / end_chr from chr_len

// @PG / @SQ lines (OutputSamHeaders, ReadMapping.cpp:101-123)
void sam_header(const HostIndex &ix, std::string &out);

} // namespace mcx

// Discordant-pair events ('E' records: pos = pair number in input order, len = branch, seq = g1, g2, dist) ->
// the inversion / translocation site records ('V' / 'T') the reference pushes at ReadMapping.cpp:486-521,
// appended to out.  Replayed in input order: the second branch pushes the DiscordPair variable with whatever
// the previous discordant pair of the stream left in it (ReadMapping.cpp:418, :499-505).
struct mcx_sparse_rec;
void mcx_disc_resolve(const mcx_sparse_rec *events, size_t n, int64_t G, std::vector<mcx_sparse_rec> &out);

namespace mcx {

} // namespace mcx
#endif
