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
 * Model: ONE process, one alch_ring family per device (a ring is bound to the HIP device that was current when it was created;
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
int alch_comm_destroy(alch_comm *comm);
int alch_comm_size(const alch_comm *comm, int *n_dev);
/* bufs[r][first .. first+count) <- bufs[root][first .. first+count) for every rank r.  bufs holds n_dev buffers, bufs[r] on device r,
 * all of rings with the same dimension, limb count and word size (the same `Cyc t m' zq` type on every GPU).  The hint source of
 * alch_hint_from_buf / alch_tunnel_create: generate on rank `root`, broadcast, then build the resident hint on every rank. */
int alch_hint_broadcast(alch_comm *comm, int root, alch_buf *const *bufs, size_t first, size_t count);
/* dst[r][k * count .. (k+1) * count) <- src[k][first .. first+count) for every pair of ranks (r, k): every rank ends up with every
 * rank's range, in rank order.  dst[r] must hold n_dev * count elements; src and dst rings as above. */
int alch_buf_all_gather(alch_comm *comm, alch_buf *const *src, size_t first, size_t count, alch_buf *const *dst);

#ifdef __cplusplus
}
#endif
#endif /* ALCHEMY_RCCL_H */
