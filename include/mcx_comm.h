/* include/mcx_comm.h — the one collective of a multi-GPU run, over RCCL (libmcx_comm.so).
 *
 * Reads shard embarrassingly; what the GPUs of a node exchange is the alignment profile that
 * VariantCalling() reads (reference src/AlignmentProfile.cpp:41-242 accumulates MappingRecordArr under
 * ProfileLock, src/VariantCalling.cpp:696 consumes it): the per-position counter planes are summed
 * onto the GPU that calls the variants.  This library is separate from libmcx.so so that a process
 * which brings its own RCCL (PyTorch does) never loads a second copy; mapcaller-mi355x -gpus N links
 * it, mapcaller_amd/run.py uses torch.distributed instead.
 *
 * Plain pointers and sizes; functions return 0 or a negative mcx_status (mcx_last_error() has the text).
 */
#ifndef MCX_COMM_H
#define MCX_COMM_H
#include <stdint.h>
#include "mcx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcx_comm mcx_comm;

/* One process, n host threads, one GPU each (ncclCommInitAll): out[r] is rank r's handle, to be used
 * by the thread that owns devices[r].  When two ranks name the same device (tests on a single-GPU
 * box; RCCL refuses that) the handles fall back to a direct sum inside the process. */
int mcx_comm_init_all(int32_t n, const int32_t *devices, mcx_comm **out);
/* One process per GPU: rank 0 makes the id (ncclGetUniqueId) and hands it to the others by whatever
 * means the launcher has; every rank then joins (ncclCommInitRank). */
#define MCX_COMM_ID_BYTES 128
int mcx_comm_unique_id(uint8_t id[MCX_COMM_ID_BYTES]);
int mcx_comm_init_rank(const uint8_t id[MCX_COMM_ID_BYTES], int32_t rank, int32_t size, int32_t device, mcx_comm **out);
void mcx_comm_free(mcx_comm *);
int32_t mcx_comm_rank(const mcx_comm *);
int32_t mcx_comm_size(const mcx_comm *);

/* Sums the counter planes of every rank onto `root` (ncclReduce, in pieces that fit a 32-bit count):
 * d_planes = the memory given to mcx_profile_attach (mcx.h: multi_hit as u32, the other nine planes as
 * u16, 22 bytes per position).  The readCount plane is left alone — with the duplicate cap decided across
 * shards (mcx_batch_accumulate) every rank already holds the run's count.  RCCL sums no 16-bit integers, so
 * the 16-bit planes travel as the words they lie in, two positions to a word: A C G T clamped to 4095 on
 * every rank first (up to 16 ranks cannot carry into the neighbouring half, and the clamped sum finalises to
 * the same value), F1 R2 F2 R1 as they are when no half can carry (the largest one over all ranks times
 * the number of ranks is below 2^16; otherwise piece by piece with one counter per word) — 20 bytes per
 * position on the wire.  What the root holds afterwards equals the plain sum once finalised; the other
 * ranks' planes are scratch.  Call after
 * mcx_profile_settle (the planes must hold counts, not differences) and before
 * mcx_profile_finalize / mcx_call_variants on the root. */
int mcx_profile_reduce(mcx_comm *, uint32_t *d_planes, int64_t genome_size, int32_t root, double *seconds);

/* An mcx_exchange (mcx.h) over the communicator, for launchers that run one process per GPU without
 * another transport: host buffers staged through HBM, ncclAllGather.  Free with mcx_comm_exchange_free. */
int mcx_comm_exchange(mcx_comm *, mcx_exchange *out);
void mcx_comm_exchange_free(mcx_exchange *);

#ifdef __cplusplus
}
#endif
#endif
