// libalchemy_rccl.so -- RCCL collectives on the library's device buffers (include/alchemy_rccl.h).
// Uses only the public C ABI of libalchemy_hip.so (alch_buf_device_ptr, alch_buf_ring, alch_ring_device, alch_ring_n,
// alch_buf_elems), the HIP runtime and RCCL.  Two launch models share the collectives: ONE process over devices 0 .. n-1
// (alch_comm_init_all: rank r = device r, group calls from the calling thread) and ONE PROCESS PER GPU (alch_comm_init_rank: the
// process holds one rank on its current device; the 128-byte id of alch_comm_unique_id travels over the host's own channel).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/alchemy_rccl.h"

struct alch_comm {
    int n = 0;                          // ranks of the communicator (all processes)
    std::vector<ncclComm_t> comms;      // one per LOCAL rank
    std::vector<int> ranks;             // local rank i is rank ranks[i] ...
    std::vector<int> devices;           // ... and lives on HIP device devices[i]
    int n_local() const { return (int)comms.size(); }
};
static_assert(ALCH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id handed between processes is RCCL's ncclUniqueId");

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
extern "C" const char* alch_rccl_last_error(void) { return g_err.c_str(); }
// no C++ exception leaves the library: every entry point is a function-try-block ending here (as in alchemy_hip.hip)
static int abi_catch() noexcept {
    int code = ALCH_E_INTERNAL;
    try {
        throw;
    } catch (const std::bad_alloc&) {
        code = ALCH_E_NOMEM;
        try { g_err = "out of host memory (std::bad_alloc)"; } catch (...) {}
    } catch (const std::exception& e) {
        try { g_err = std::string("internal error: ") + e.what(); } catch (...) {}
    } catch (...) {
        try { g_err = "internal error: unknown exception"; } catch (...) {}
    }
    return code;
}

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t _r = (expr);                                                                        \
        if (_r != ncclSuccess) return fail(ALCH_E_HIP, std::string(#expr) + ": " + ncclGetErrorString(_r)); \
    } while (0)

extern "C" int alch_comm_init_all(int n_dev, alch_comm** out) try {
    if (!out) return fail(ALCH_E_INVALID, "alch_comm_init_all: null out");
    *out = nullptr;
    if (n_dev < 1) return fail(ALCH_E_INVALID, "alch_comm_init_all: n_dev must be >= 1");
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) return fail(ALCH_E_NO_DEVICE, "no HIP device");
    if (n_dev > visible)
        return fail(ALCH_E_NO_DEVICE, "alch_comm_init_all: " + std::to_string(n_dev) + " ranks but only " + std::to_string(visible) + " devices visible (one rank per GPU)");
    alch_comm* c = new alch_comm();
    c->n = n_dev;
    c->comms.resize((size_t)n_dev);
    c->ranks.resize((size_t)n_dev);
    c->devices.resize((size_t)n_dev);
    for (int r = 0; r < n_dev; ++r) c->ranks[(size_t)r] = c->devices[(size_t)r] = r;
    ncclResult_t rc = ncclCommInitAll(c->comms.data(), n_dev, c->devices.data());
    if (rc != ncclSuccess) { delete c; return fail(ALCH_E_HIP, std::string("ncclCommInitAll: ") + ncclGetErrorString(rc)); }
    *out = c;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_comm_unique_id(unsigned char* id) try {
    if (!id) return fail(ALCH_E_INVALID, "alch_comm_unique_id: null id");
    ncclUniqueId u;
    NCCL_TRY(ncclGetUniqueId(&u));
    std::memcpy(id, u.internal, ALCH_COMM_ID_BYTES);
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_comm_init_rank(int n_ranks, int rank, const unsigned char* id, alch_comm** out) try {
    if (!out) return fail(ALCH_E_INVALID, "alch_comm_init_rank: null out");
    *out = nullptr;
    if (!id) return fail(ALCH_E_INVALID, "alch_comm_init_rank: null id");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(ALCH_E_INVALID, "alch_comm_init_rank: need 0 <= rank < n_ranks");
    int visible = 0, dev = -1;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1 || hipGetDevice(&dev) != hipSuccess) return fail(ALCH_E_NO_DEVICE, "no HIP device");
    ncclUniqueId u;
    std::memcpy(u.internal, id, ALCH_COMM_ID_BYTES);
    alch_comm* c = new alch_comm();
    c->n = n_ranks;
    c->comms.resize(1);
    c->ranks.assign(1, rank);
    c->devices.assign(1, dev);
    ncclResult_t rc = ncclCommInitRank(c->comms.data(), n_ranks, u, rank);      // blocks until all n_ranks processes have called it
    if (rc != ncclSuccess) { delete c; return fail(ALCH_E_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(rc)); }
    *out = c;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_comm_local(const alch_comm* c, int* n_local, int* first_rank) try {
    if (!c || !n_local || !first_rank) return fail(ALCH_E_INVALID, "null argument");
    *n_local = c->n_local();
    *first_rank = c->ranks[0];
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_comm_destroy(alch_comm* c) try {
    if (!c) return ALCH_OK;
    for (ncclComm_t k : c->comms) (void)ncclCommDestroy(k);
    delete c;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_comm_size(const alch_comm* c, int* n_dev) try {
    if (!c || !n_dev) return fail(ALCH_E_INVALID, "null argument");
    *n_dev = c->n;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

namespace {
struct RankBuf {
    char* ptr = nullptr;
    size_t elems = 0, elem_bytes = 0;
    int device = -1;
    hipStream_t stream = nullptr;
};

// Resolves one buffer per rank and checks that rank r's buffer lives on device r and that all rings have one element size.
int resolve(const alch_comm* c, alch_buf* const* bufs, const char* what, std::vector<RankBuf>& out) {
    if (!c || !bufs) return fail(ALCH_E_INVALID, std::string(what) + ": null argument");
    out.resize((size_t)c->n_local());
    for (int r = 0; r < c->n_local(); ++r) {
        if (!bufs[r]) return fail(ALCH_E_INVALID, std::string(what) + ": null buffer for rank " + std::to_string(r));
        alch_ring* ring = nullptr;
        void* p = nullptr;
        size_t bytes = 0, elems = 0;
        uint32_t n = 0;
        int L = 0, word = 0;
        void* stream = nullptr;
        RankBuf& b = out[(size_t)r];
        if (alch_buf_ring(bufs[r], &ring) != ALCH_OK || alch_buf_device_ptr(bufs[r], &p, &bytes) != ALCH_OK ||
            alch_buf_elems(bufs[r], &elems) != ALCH_OK || alch_ring_n(ring, &n, &L, &word) != ALCH_OK ||
            alch_ring_device(ring, &b.device, &stream) != ALCH_OK)
            return fail(ALCH_E_INVALID, std::string(what) + ": bad buffer handle for rank " + std::to_string(r));
        b.ptr = static_cast<char*>(p);
        b.elems = elems;
        b.elem_bytes = (size_t)n * (size_t)L * (size_t)word;
        b.stream = static_cast<hipStream_t>(stream);
        if (b.device != c->devices[(size_t)r])
            return fail(ALCH_E_INVALID, std::string(what) + ": the buffer of rank " + std::to_string(c->ranks[(size_t)r]) + " lives on device " + std::to_string(b.device) +
                                            ", its rank on device " + std::to_string(c->devices[(size_t)r]) + " (create that rank's rings with its device current)");
        if (b.elem_bytes != out[0].elem_bytes) return fail(ALCH_E_INVALID, std::string(what) + ": the ranks' rings differ in dimension, limbs or word size");
    }
    return ALCH_OK;
}
}  // namespace

extern "C" int alch_hint_broadcast(alch_comm* c, int root, alch_buf* const* bufs, size_t first, size_t count) try {
    std::vector<RankBuf> b;
    int rc = resolve(c, bufs, "alch_hint_broadcast", b);
    if (rc != ALCH_OK) return rc;
    if (root < 0 || root >= c->n) return fail(ALCH_E_INVALID, "alch_hint_broadcast: root out of range");
    for (auto& x : b) if (count > x.elems || first > x.elems - count) return fail(ALCH_E_INVALID, "alch_hint_broadcast: element range out of bounds");
    if (count == 0) return ALCH_OK;
    const size_t bytes = count * b[0].elem_bytes, off = first * b[0].elem_bytes;
    NCCL_TRY(ncclGroupStart());
    for (int r = 0; r < c->n_local(); ++r) {      // in place on every rank: the root sends what it holds, the others' send buffers are not read
        ncclResult_t e = ncclBroadcast(b[(size_t)r].ptr + off, b[(size_t)r].ptr + off, bytes, ncclChar, root, c->comms[(size_t)r], b[(size_t)r].stream);
        if (e != ncclSuccess) { (void)ncclGroupEnd(); return fail(ALCH_E_HIP, std::string("ncclBroadcast: ") + ncclGetErrorString(e)); }
    }
    NCCL_TRY(ncclGroupEnd());
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_all_gather(alch_comm* c, alch_buf* const* src, size_t first, size_t count, alch_buf* const* dst) try {
    std::vector<RankBuf> s, d;
    int rc = resolve(c, src, "alch_buf_all_gather (src)", s);
    if (rc != ALCH_OK) return rc;
    if ((rc = resolve(c, dst, "alch_buf_all_gather (dst)", d)) != ALCH_OK) return rc;
    if (s[0].elem_bytes != d[0].elem_bytes) return fail(ALCH_E_INVALID, "alch_buf_all_gather: source and destination rings differ");
    for (int r = 0; r < c->n_local(); ++r) {
        if (count > s[(size_t)r].elems || first > s[(size_t)r].elems - count) return fail(ALCH_E_INVALID, "alch_buf_all_gather: source range out of bounds");
        if (count > d[(size_t)r].elems / (size_t)c->n) return fail(ALCH_E_INVALID, "alch_buf_all_gather: dst must hold n_dev * count elements");
        if (s[(size_t)r].stream != d[(size_t)r].stream)
            return fail(ALCH_E_INVALID, "alch_buf_all_gather: a rank's source and destination rings must share a stream (the same ring, or alch_ring_share_stream)");
    }
    if (count == 0) return ALCH_OK;
    const size_t bytes = count * s[0].elem_bytes, off = first * s[0].elem_bytes;
    NCCL_TRY(ncclGroupStart());
    for (int r = 0; r < c->n_local(); ++r) {
        ncclResult_t e = ncclAllGather(s[(size_t)r].ptr + off, d[(size_t)r].ptr, bytes, ncclChar, c->comms[(size_t)r], s[(size_t)r].stream);
        if (e != ncclSuccess) { (void)ncclGroupEnd(); return fail(ALCH_E_HIP, std::string("ncclAllGather: ") + ncclGetErrorString(e)); }
    }
    NCCL_TRY(ncclGroupEnd());
    return ALCH_OK;
} catch (...) { return abi_catch(); }
