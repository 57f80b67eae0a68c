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
 * d_planes = the array given to mcx_profile_attach, 10 * genome_size u32 [plane][position].  The
 * readCount plane is left alone — with the duplicate cap decided across shards (mcx_batch_accumulate)
 * every rank already holds the run's count.  The counters are 12- and 16-bit fields once finalised,
 * so two planes travel in one u32 where that is exact: A|C and G|T clamped to 4095 on every rank first
 * (up to 16 ranks), F1|R2 and F2|R1 when no low half can carry (largest low half over all ranks x ranks
 * < 2^16; otherwise the four travel alone) — five planes (20 bytes per position) on the wire instead of
 * nine.  What the root holds afterwards equals the plain sum once finalised; the other ranks' planes are
 * left in their packed form.  Collective: every rank calls it, after its own
 * mcx_profile_settle (the planes must hold counts, not differences) and before
 * mcx_profile_finalize on the root.  seconds (may be NULL): wall time of the call on this rank. */
int mcx_profile_reduce(mcx_comm *, uint32_t *d_planes, int64_t genome_size, int32_t root, double *seconds);

/* An mcx_exchange (mcx.h) over the communicator, for launchers that run one process per GPU without
 * another transport: host buffers staged through HBM, ncclAllGather.  Free with mcx_comm_exchange_free. */
int mcx_comm_exchange(mcx_comm *, mcx_exchange *out);
void mcx_comm_exchange_free(mcx_exchange *);

#ifdef __cplusplus
}
#endif
#endif
