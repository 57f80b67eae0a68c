// mapcaller_amd/csrc/mcx_host.h — host-side pieces around the GPU path: index files, read files,
// SAM text.  (reference src/bwt_index.cpp, src/GetData.cpp, src/SamReport.cpp:324-488 formatting)
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

struct HostRead {
    std::string name, seq, qual;
};

class ReadFile {
public:
    bool open(const std::string &path, std::string &err);
    bool next(HostRead &r); // false at end of file
    void close();
    bool fastq() const { return fastq_; }
    ~ReadFile() { close(); }
private:
    bool line(std::string &s);
    void *gz_ = nullptr;
    bool fastq_ = true;
    std::string pending_;
    bool have_pending_ = false;
};

// one SAM line for a read (no trailing newline); rec/cigar as produced by the GPU path
struct AlnRec;
void sam_line(const HostIndex &ix, const HostRead &rd, bool mate2_flipped, bool fastq, const struct AlnRec &rec,
              const uint32_t *cigar, std::string &out);
void sam_header(const HostIndex &ix, std::string &out);

// Replay of the reference's insert-size feedback over one batch (ReadMapping.cpp:462, :538-539):
// avgDist is re-estimated after every chunk of 100 pairs once more than 1000 proper pairs were
// seen, and the next chunk pairs its mates with EstiDistance = (int)(avgDist*1.5).  Each pair
// reports the interval of estimates that leaves its result unchanged; the pairs whose chunk
// estimate falls outside it are returned in redo/redo_est (to be re-run with the exact value).
// avg = {avgDist, iTotalPairedNum, TotalPairedDistance, reads seen}; avg_out gets the state
// after the batch (valid once redo comes back empty).
struct PairOut;
void avg_replay(const struct PairOut *po, uint32_t n_pairs, const int64_t avg[4], std::vector<uint32_t> &redo,
                std::vector<int32_t> &redo_est, int64_t avg_out[4]);

} // namespace mcx
#endif
