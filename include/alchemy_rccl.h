/*
 * alchemy_rccl.h -- the native multi-GPU route of the MI355X ciphertext-arithmetic backend: RCCL collectives on the library's own
 * device buffers, for hosts that are not Python (a Haskell or C++ evaluator that shards a batch over the GPUs of one node).
 *
 * Scope (BASELINE.json north_star: "independent ciphertexts shard embarrassingly across the 8 GPUs with RCCL over xGMI for the
 * batch gather only"): ciphertexts are independent, so the data path has NO collective.  Two exchanges exist around it:
 *   - the key-switch / tunnel hints are generated once per circuit (Crypto/Alchemy/Interpreter/KeysHints.hs:101-129) and every GPU
 *     needs them: one broadcast before anything is timed (8 MiB at BASELINE config 3);
 *   - results may be collected after a pipeline (examples/HomomRLWR.hs:52-59 maps one `f` over many inputs): an all-gather of result
 *     ranges, outside the timed region -- gathering every result of the hot path into one GPU would be bound by its 7 x ~153 GB/s of
 *     xGMI ingress, far below what eight GPUs produce.
 *
 * Two launch models, one set of collectives:
 *   (1) ONE process over the node's GPUs (alch_comm_init_all), described next;
 *   (2) ONE PROCESS PER GPU -- the launch model of torchrun / mpirun, or one Haskell RTS per device -- (alch_comm_init_rank): rank 0
 *       obtains a 128-byte id (alch_comm_unique_id) and hands it to the other processes over whatever channel the host has (a file,
 *       a pipe, an environment variable set by the launcher, MPI_Bcast); every process then joins with its own rank, on the HIP
 *       device that is current in it.  Such a communicator holds ONE local rank, so the `bufs` / `src` / `dst` arrays of the
 *       collectives have ONE entry there: the calling process's buffer.
 *
 * Model (1): ONE process, one alch_ring family per device (a ring is bound to the HIP device that was current when it was created;
 * every entry point of include/alchemy_hip.h makes that device current, so one host thread per device -- or one thread for all --
 * may drive them).  A communicator spans devices 0 .. n_dev-1 of the process (ncclCommInitAll).  Collectives take one alch_buf per
 * rank, rank r's buffer living on device r; they are queued on the stream of each buffer's ring, i.e. ordered after everything the
 * library has queued for that ring and before everything queued later -- alch_sync (or any download) waits for them.
 * bench.py's torchrun path (one process per GPU, torch.distributed) stays the driver's scaling entry; this library is the same two
 * exchanges without Python.  Optional: libalchemy_rccl.so links librccl.so and libalchemy_hip.so; nothing in libalchemy_hip.so
 * depends on it.
 *
 * Status codes are those of alchemy_hip.h (ALCH_OK, ALCH_E_INVALID, ALCH_E_NO_DEVICE, ALCH_E_HIP for RCCL / HIP failures);
 * alch_rccl_last_error() holds the message (thread local).
 */
#ifndef ALCHEMY_RCCL_H
#define ALCHEMY_RCCL_H

#include "alchemy_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct alch_comm alch_comm;

const char *alch_rccl_last_error(void);
/* A communicator over devices 0 .. n_dev-1 (1 <= n_dev <= visible devices), rank r = device r. */
int alch_comm_init_all(int n_dev, alch_comm **out);
/* One process per GPU.  alch_comm_unique_id: call in ONE process, ship the ALCH_COMM_ID_BYTES bytes to the others.
 * alch_comm_init_rank: every process, with the same id and n_ranks and its own rank; the communicator's rank lives on the HIP device
 * current in the calling thread (hipSetDevice first; with one visible device per process that is device 0).  Blocks until all
 * n_ranks processes have called it. */
#define ALCH_COMM_ID_BYTES 128
int alch_comm_unique_id(unsigned char *id);
int alch_comm_init_rank(int n_ranks, int rank, const unsigned char *id, alch_comm **out);
int alch_comm_destroy(alch_comm *comm);
/* ranks of the communicator over all processes */
int alch_comm_size(const alch_comm *comm, int *n_dev);
/* ranks held by THIS process (= entries of every buffer array below: n_dev after alch_comm_init_all, 1 after alch_comm_init_rank)
 * and the first of them */
int alch_comm_local(const alch_comm *comm, int *n_local, int *first_rank);
/* bufs[r][first .. first+count) <- bufs[root][first .. first+count) for every rank r.  bufs holds one buffer per LOCAL rank (n_dev
 * of them in model (1), bufs[r] on device r; one in model (2), where `root` is still a rank of the whole communicator),
 * all of rings with the same dimension, limb count and word size (the same `Cyc t m' zq` type on every GPU).  The hint source of
 * alch_hint_from_buf / alch_tunnel_create: generate on rank `root`, broadcast, then build the resident hint on every rank. */
int alch_hint_broadcast(alch_comm *comm, int root, alch_buf *const *bufs, size_t first, size_t count);
/* dst[r][k * count .. (k+1) * count) <- src[k][first .. first+count) for every pair of ranks (r, k): every rank ends up with every
 * rank's range, in rank order.  dst[r] must hold (ranks of the communicator) * count elements; src and dst rings as above, one
 * entry per local rank. */
int alch_buf_all_gather(alch_comm *comm, alch_buf *const *src, size_t first, size_t count, alch_buf *const *dst);

#ifdef __cplusplus
}
#endif
#endif /* ALCHEMY_RCCL_H */
