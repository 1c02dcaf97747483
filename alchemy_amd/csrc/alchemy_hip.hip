// C ABI of the MI355X ciphertext-arithmetic backend (see include/alchemy_hip.h) and the element-wise
// kernels around the LDS-resident transforms of kernels_ntt.hpp.
//
// No CPU fallback lives here: every compute entry point needs a gfx950 device and reports
// ALCH_E_NO_DEVICE / ALCH_E_HIP otherwise.  Nothing in this library includes or links oracle/.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/alchemy_hip.h"
#include "kernels_ntt.hpp"
#include "ring_host.hpp"
#include "gen_host.hpp"

using namespace alch;

// ------------------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------------------
// Device tables of the Tensor methods between two indices m | m' (tensor_ext.inc.hpp), cached in the bigger ring.
struct ExtTab {
    u32 d_rel = 0, fibre = 0;
    int32_t* pow_gather = nullptr;     // [n_big]: source position in the small ring, or -1 (embedPow)
    int32_t* coeffs = nullptr;         // [d_rel][n_small]: source position in the big ring (coeffs; row 0 = twacePowDec)
    int32_t* slot_small = nullptr;     // [n_big]: CRT slot of the small ring (embedCRT)
    int32_t* fibres = nullptr;         // [n_small][fibre]: CRT slots of the big ring above a slot of the small ring (twaceCRT)
};

// live streams with a hardware queue of their own (ring option "stream_dedicated"): capped, because the HIP runtime does not survive
// running out of hardware queues (observed: a segmentation fault inside the runtime after ~250 such streams in one process)
static std::atomic<int> g_dedicated_streams{0};
constexpr int MAX_DEDICATED_STREAMS = 32;

struct StreamOwner {
    hipStream_t s;
    int device;
    bool dedicated;
    StreamOwner(hipStream_t s_, int d, bool dedicated_ = false) : s(s_), device(d), dedicated(dedicated_) {}
    ~StreamOwner() {
        if (s) { (void)hipSetDevice(device); (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        if (dedicated) --g_dedicated_streams;
    }
    StreamOwner(const StreamOwner&) = delete;
    StreamOwner& operator=(const StreamOwner&) = delete;
};

struct alch_ring {
    u32 m = 0, n = 0;
    int logn = 0, L = 0, word = 0;            // word = 4 or 8 bytes per residue on the device
    u64 q[MAXL] = {0};
    bool balanced = false;
    bool q30 = false;                          // 32-bit words and every modulus below 2^30 (4q fits a word)
    hipStream_t stream = nullptr;
    // Owner of `stream` when the library created it: shared by every ring that borrowed the stream through alch_ring_share_stream, so
    // the stream outlives the ring that created it for as long as another ring still queues on it (rings are destroyed in any order --
    // a garbage-collected host gives none).  Null when the stream is the caller's (alch_ring_set_stream).
    std::shared_ptr<StreamOwner> stream_owner;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void* tables = nullptr;                    // all twiddle tables, one allocation
    void* tables_p = nullptr;                  // Plantard forward constants (32-bit rings) / Shoup pairs (64-bit rings)
    DevRing<u32> d32;
    DevRing<u64> d64;
    void* ws_digits = nullptr;                 // digit scratch, two halves of [chunk][L][n] signed words
    size_t ws_digits_bytes = 0;
    hipStream_t aux = nullptr;                 // second pipeline of ct_mul_relin (odd chunks)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void* ws_in = nullptr;                     // crt scratch for ALCH_POW_IN (2 * 2*batch elements)
    size_t ws_in_bytes = 0;
    void* ws_host = nullptr;                   // staging for the host-buffer Tensor methods
    size_t ws_host_bytes = 0;
    u64* ws_sum = nullptr;                     // checksum accumulator
    size_t chunk = 1024;                       // ciphertexts per (tensor_intt, ks_accum) launch pair
    void* ws_full = nullptr;                   // ct_mul_full scratch: per pipeline digits + key-switched chunk + stash
    size_t ws_full_bytes = 0;
    hipEvent_t ev_x = nullptr;                 // cross-ring ordering (ct_mul_full)
    int device = 0;                            // HIP device the ring's streams, tables and buffers live on
    LaunchOpts opts;                           // launch-structure options (alch_ring_set_option)
    bool one_stream = false;
    int nstreams = 2;                          // alch_ct_mul_relin: independent (tensor, key switch) pipelines the chunks rotate over (1 .. 4)
    hipStream_t xs[2] = {nullptr, nullptr};    // pipelines 3 and 4, created on first use
    hipEvent_t ev_xs[2] = {nullptr, nullptr};
    int pipe = 0;                              // alch_ct_mul_relin: 1 = tensor kernels on the aux stream one chunk ahead of the key-switch kernels
    hipEvent_t ev_pa[2] = {nullptr, nullptr}, ev_pb[2] = {nullptr, nullptr};
    unsigned rs_slots = 512;                   // resident workgroups of k_rescale_out (each owns a stash slot)
    size_t scratch_mib = 4096;                 // scratch of the composed (unfused) paths: digits + intermediates of one chunk
                                               // (n = 2^16: 66.3 k op/s at 1 GiB, 73.3 k at 4 GiB, 77.7 k at 16 GiB -- small chunks leave CUs idle)
    alch_buf* scratch = nullptr;               // staging elements of the host-buffer Tensor methods
    // general cyclotomic index (kernel_gen.hpp); two-power rings with n >= 16 keep the radix-16 engine
    bool gen = false;
    bool has_crt = true;                       // false: ring created with alch_ring_create_nocrt (Pow / Dec operations only)
    bool zdom = false;                         // "modulus" 0: the integers, signed 64-bit words
    GenHost gh;
    GenDev<u32> g32;
    GenDev<u64> g64;
    void* gen_tables = nullptr;
    int* d_flag = nullptr;                     // divG failure flag
    std::deque<std::pair<u32, ExtTab>> ext;    // extension tables towards sub-rings of index .first (same moduli)
    // Free list of small device buffers: alch_buf_alloc / alch_buf_free of single ring elements are the allocation pattern of a
    // device-resident Tensor value (one buffer per `GT` value), and hipMalloc / hipFree cost 50-200 us and synchronise the device.
    // Reuse is stream-ordered: every consumer of a buffer on another ring's stream makes the owner's stream wait for it (ext_order,
    // alch_ct_mod_switch, alch_ct_tunnel, alch_ct_mul_full), so work queued later on the owner's stream may overwrite it.
    std::vector<std::pair<size_t, void*>> pool;                 // (n_elems, device pointer)
    size_t pool_bytes = 0;
    std::mutex pool_mu;                                         // alch_buf_free may come from a finalizer thread (Haskell ForeignPtr)
    // Pinned host staging of small transfers (alch_buf_upload / alch_buf_download of a few ring elements): no pageable-memory copy
    // inside the HIP runtime, no synchronisation on upload, exactly one on download.
    struct Pin { void* p = nullptr; size_t bytes = 0; hipEvent_t ev = nullptr; bool busy = false; };
    Pin pin[4];
    int pin_next = 0;
};

static const size_t POOL_MAX_BUF = (size_t)32 << 20;           // buffers up to 32 MiB are recycled
static const size_t POOL_CAP = (size_t)1 << 30;                // at most 1 GiB parked per ring
static const size_t PIN_MAX = (size_t)8 << 20;                 // transfers up to 8 MiB of int64 go through pinned staging

struct alch_buf {
    alch_ring* ring;
    size_t n_elems;
    void* dptr;
    bool view = false;                         // alch_buf_view: a non-owning alias into another buffer
};

struct alch_tunnel {
    alch_ring* rr;                             // R'_q (input ciphertexts)
    alch_ring* rs;                             // S'_q (hint, output ciphertexts)
    u32 d_rel;                                 // dim of R'/E' = number of E'-coefficients per ring element
    u32 linv_skip_mask;
    int32_t* table;                            // device: [d_rel][n_s] source positions in R', -1 = zero
    void* lin;                                 // device: f' values y_i, [d_rel][L][n_s], CRT basis, Montgomery form
    void* ks;                                  // device: [d_rel * D][2][L][n_s], CRT basis, Montgomery form (D gadget digits)
    int gadget;
    int digits;                                // D: L for TrivGad, sum_i ceil(log2 q_i) for BaseBGad 2
    // E'-level form (null when E' = S' or E' has no general-index ring): an embedded E'-element's CRT over S' is its CRT over E'
    // replicated (embedCRT), so the constant terms' and the digits' transforms run at dimension phi(e') and the products with
    // the linear function / the hints read them through slot_e
    alch_ring* re = nullptr;                   // E'_q, same moduli (owned)
    int32_t* table_e = nullptr;                // device: [d_rel][n_e] source positions in R'
    u32* slot_e = nullptr;                     // device: [n_s] CRT slot of E' behind every CRT slot of S'
    bool pieces_ok = false;                    // every aligned group of four S' slots reads four consecutive, aligned E' slots
};

struct alch_hint {
    alch_ring* ring;
    int gadget;
    int digits;
    void* dptr;                                // [digit][2][L][n] words, Montgomery form
};

// ------------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

// element ranges of the C ABI are checked without forming first + count or 2 * batch (a huge argument must not wrap past the check)
static inline bool range_ok(size_t first, size_t count, size_t n) { return count <= n && first <= n - count; }
static inline bool pairs_ok(size_t batch, size_t n) { return batch <= n / 2; }
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
// No C++ exception leaves the library (include/alchemy_hip.h: "no exceptions across the ABI" -- the callers are C, Haskell's FFI and
// ctypes, where an escaping exception is std::terminate at best): every `extern "C" int` entry point is a function-try-block that
// ends in this handler.  What can throw inside them is host-side allocation (std::vector / std::string / new: std::bad_alloc,
// std::length_error) -- reported as ALCH_E_NOMEM / ALCH_E_INTERNAL with the message in alch_last_error(), never as an abort.
// tests/test_abi_guard.py checks the mechanism through the real ABI (alch_debug_throw) and, lexically, that no entry point lacks it.
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
// test hook (not part of the Tensor surface): throws the chosen exception inside a guarded entry point
extern "C" int alch_debug_throw(int kind) try {
    if (kind == 1) throw std::bad_alloc();
    if (kind == 2) throw std::length_error("alch_debug_throw");
    if (kind == 3) throw 42;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(ALCH_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                 \
    } while (0)

// HIP's current device is per host thread: every entry point first makes the ring's device current, so a host that
// calls from another OS thread (Haskell `safe` FFI calls on a -threaded RTS) still launches on the right GPU.
// Every entry point that touches a ring also takes the lock of the ring's DEVICE for the length of the call (recursive: entry points
// call each other).  A ring's scratch, pools and events are plain members, and calls between rings (embed / twace, tunnels, mul_ over
// three rings) touch several rings at once, so the unit of mutual exclusion is the device: a host that forces tensors from several
// threads (a -threaded Haskell RTS, parallel sparks) gets serialised launches instead of corrupted scratch, and threads that drive
// different GPUs (examples/ringround_multi.cpp) never meet.  alch_buf_free / alch_hint_free / alch_tunnel_free stay outside it: they
// come from finalizer threads and must not wait behind a call that is synchronising the device.
static std::recursive_mutex& device_mutex(int dev) {
    static std::recursive_mutex mu[64];
    return mu[(dev >= 0 && dev < 64) ? dev : 0];
}
#define ALCH_CAT2(a, b) a##b
#define ALCH_CAT(a, b) ALCH_CAT2(a, b)
#ifndef ALCH_NO_DEVICE_LOCK
#define ALCH_DEVICE_LOCK(dev) std::lock_guard<std::recursive_mutex> ALCH_CAT(_alch_dev_lock_, __LINE__)(device_mutex(dev))
#else
#define ALCH_DEVICE_LOCK(dev) ((void)0)   // tests/test_gpu_threads.py's self-test only (tools/build_variant.sh nolock): never in a product build
#endif
#define BIND(ringp)                                                                                     \
    ALCH_DEVICE_LOCK((ringp)->device);                                                                  \
    do {                                                                                                \
        if (hipSetDevice((ringp)->device) != hipSuccess)                                                \
            return fail(ALCH_E_HIP, "hipSetDevice(" + std::to_string((ringp)->device) + ") failed");    \
    } while (0)

extern "C" const char* alch_last_error(void) { return g_err.c_str(); }
extern "C" uint32_t alch_version(void) { return (1u << 16) | 6u; }   // 1.6: alch_buf_checksum_at, shared streams owned by their last user; 1.5: device-resident Tensor values (alch_buf_tensor_op, alch_buf_copy, alch_ring_share_stream, pooled small buffers, pinned staging), status order of alch_ring_create; 1.4: general cyclotomic indices, l / lInv, real mulG / divG, mulPublic / addPublic, alch_ring_set_option; 1.3: + alch_decompose_base2, BaseBGad hints, alch_ct_mul_full, alch_buf_device_ptr, n = 2^16

// ------------------------------------------------------------------------------------------------------
// element-wise kernels (HBM-bound; 16 B per lane, grid-stride, ~2048 workgroups)
// ------------------------------------------------------------------------------------------------------
__host__ __device__ static inline u64 splitmix64(u64 x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

template <typename W>
__global__ void k_fill_uniform(DevRing<W> R, W* data, size_t words, u64 seed) {
    const size_t n = (size_t)R.n;
    for (size_t w = blockIdx.x * (size_t)blockDim.x + threadIdx.x; w < words; w += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)((w / n) % (size_t)R.L);
        data[w] = (W)(splitmix64(seed + w) % (u64)R.mod[j].q);
    }
}

// Lol's tuple-interleaved int64 (coefficient-major, limb-minor) <-> limb-major device words.
template <typename W, bool TO_DEVICE>
__global__ void k_transpose(DevRing<W> R, W* dev, int64_t* host, size_t elems) {
    const size_t n = (size_t)R.n;
    const size_t L = (size_t)R.L;
    const size_t total = elems * L * n;
    for (size_t w = blockIdx.x * (size_t)blockDim.x + threadIdx.x; w < total; w += (size_t)gridDim.x * blockDim.x) {
        const size_t k = w % n, j = (w / n) % L, e = w / (n * L);
        const size_t h = (e * n + k) * L + j;
        if (TO_DEVICE) dev[w] = (W)(u64)host[h];
        else host[h] = (int64_t)(u64)dev[w];
    }
}

// Position (outer, mid, k) of word w in an [outer][mid < M][k < n] array, advanced by the grid stride with adds and
// carries: n is a run-time value (and not a power of two for a general index), so w / n and w % n per word would be
// 64-bit software divisions -- several hundred instructions against the handful a word of element-wise work needs.
struct Walk3 {
    u32 k, mid; size_t outer;
    u32 sk, smid; size_t souter;
    u32 n, M;
    __device__ Walk3(size_t w0, size_t stride, u32 n_, u32 M_) : n(n_), M(M_) {
        k = (u32)(w0 % n); const size_t t = w0 / n; mid = (u32)(t % M); outer = t / M;
        sk = (u32)(stride % n); const size_t s = stride / n; smid = (u32)(s % M); souter = s / M;
    }
    __device__ void step() {
        k += sk; u32 c = k >= n ? 1u : 0u; k -= c ? n : 0u;
        mid += smid + c; c = mid >= M ? 1u : 0u; mid -= c ? M : 0u;
        outer += souter + c;
    }
};
#define ALCH_WALK(w, total, walk) \
    for (size_t w = blockIdx.x * (size_t)blockDim.x + threadIdx.x; w < (total); w += (size_t)gridDim.x * blockDim.x, walk.step())
#define ALCH_WALK_INIT(n_, M_) Walk3 wk(blockIdx.x * (size_t)blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x, (u32)(n_), (u32)(M_))

// VW consecutive words moved with one access (VW = 16 bytes' worth when the ring dimension is a multiple of it, else 1): the
// element-wise kernels of the general-index path are bound by the number of memory instructions long before HBM.
template <typename W, int VW> struct alignas(sizeof(W) * VW) Pack { W v[VW]; };
#define ALCH_LAUNCH_VW(kern, ringp, items, stream, ...)                                                                         \
    do {                                                                                                                        \
        constexpr int VL_ = Vec4<W>::LANES;                                                                                     \
        if ((ringp)->n % VL_ == 0)                                                                                              \
            hipLaunchKernelGGL((kern<W, VL_>), dim3(ew_grid((items) / VL_)), dim3(256), 0, stream, __VA_ARGS__);                \
        else                                                                                                                    \
            hipLaunchKernelGGL((kern<W, 1>), dim3(ew_grid(items)), dim3(256), 0, stream, __VA_ARGS__);                          \
    } while (0)

enum PwOp { PW_MUL = 0, PW_ADD = 1, PW_SUB = 2 };

// z mod q in [0, q) for a signed z.  |z| < q is the common case (a digit, or a centred residue of a modulus of q's
// size): one conditional add; the software division only runs for the lanes that need it.
template <typename W>
__device__ inline typename Signed<W>::type reduce_signed(typename Signed<W>::type z, W q) {
    typedef typename Signed<W>::type SW;
    if (z < (SW)q && z > -(SW)q) return z < 0 ? z + (SW)q : z;
    SW r = z % (SW)q;
    return r < 0 ? r + (SW)q : r;
}

template <typename W, int OP>
__global__ void k_pointwise(DevRing<W> R, W* dst, const W* a, const W* b, size_t words) {
    const size_t n = (size_t)R.n;
    typedef typename Vec4<W>::type V;
    constexpr int VL = Vec4<W>::LANES;
    const size_t nv = words / VL;
    ALCH_WALK_INIT(n / VL, R.L);
    ALCH_WALK(v, nv, wk) {
        const ModP<W> m = R.mod[wk.mid];
        V x = reinterpret_cast<const V*>(a)[v], y = reinterpret_cast<const V*>(b)[v], z;
#pragma unroll
        for (int e = 0; e < VL; ++e) {
            if (OP == PW_MUL) z[e] = mont_mul(mont_mul(x[e], m.r2, m), y[e], m);
            else if (OP == PW_ADD) z[e] = add_mod(x[e], y[e], m.q);
            else z[e] = sub_mod(x[e], y[e], m.q);
        }
        reinterpret_cast<V*>(dst)[v] = z;
    }
}

// The same, one word per lane: rings whose dimension is not a multiple of the vector width (general indices such as
// m = 9, n = 6), where a 16-byte piece would straddle two limbs.
template <typename W, int OP>
__global__ void k_pointwise_scalar(DevRing<W> R, W* dst, const W* a, const W* b, size_t words) {
    ALCH_WALK_INIT(R.n, R.L);
    ALCH_WALK(w, words, wk) {
        const ModP<W> m = R.mod[wk.mid];
        if (OP == PW_MUL) dst[w] = mont_mul(mont_mul(a[w], m.r2, m), b[w], m);
        else if (OP == PW_ADD) dst[w] = add_mod(a[w], b[w], m.q);
        else dst[w] = sub_mod(a[w], b[w], m.q);
    }
}

// dst = src * s_j (mod q_j); sm[j] = s_j in Montgomery form.  TO_MONT callers pass sm = R^2 mod q.
template <typename W, int VW = 1>
__global__ void k_scale(DevRing<W> R, W* dst, const W* src, size_t words, Scal<W> sm) {
    typedef Pack<W, VW> P;
    ALCH_WALK_INIT(R.n / VW, R.L);
    ALCH_WALK(w, words / VW, wk) {
        P x = reinterpret_cast<const P*>(src)[w];
#pragma unroll
        for (int c = 0; c < VW; ++c) x.v[c] = mont_mul(x.v[c], sm.v[wk.mid], R.mod[wk.mid]);
        reinterpret_cast<P*>(dst)[w] = x;
    }
}

// TrivGad decompose + reduce on one Pow-basis element: digits[i] (limb-major element i) limb j =
// centred(c limb i) mod q_j.
template <typename W>
__global__ void k_decompose_triv(DevRing<W> R, const W* c, W* digits, int balanced) {
    typedef typename Signed<W>::type SW;
    const size_t n = (size_t)R.n;
    const size_t L = (size_t)R.L;
    const size_t total = L * L * n;
    c += (size_t)blockIdx.y * L * n;                       // blockIdx.y: which ring element
    digits += (size_t)blockIdx.y * total;
    ALCH_WALK_INIT(R.n, R.L);
    ALCH_WALK(w, total, wk) {
        const size_t k = wk.k, j = wk.mid, i = wk.outer;
        const W qi = R.mod[i].q, qj = R.mod[j].q;
        const W v = c[i * n + k];
        const SW z = v > ((qi - 1) >> 1) ? (SW)v - (SW)qi : (SW)v;
        SW r;
        if (balanced) r = z < 0 ? z + (SW)qj : z;            // |z| < q_j for every pair of limbs: one add
        else r = reduce_signed<W>(z, qj);
        digits[w] = (W)r;
    }
}

// BaseBGad 2 decompose + reduce of one Pow-basis element.  One thread per (source limb i, coefficient k) walks
// the ceil(log2 q_i) digits (balanced remainder in {0,-1}, top digit absorbs the rest) and writes each one
// reduced into every limb.  first_digit[i] = index of limb i's first digit, kd[i] = its digit count.
template <typename W>
__global__ void k_decompose_base2(DevRing<W> R, const W* c, W* digits, Scal<u32> first_digit, Scal<u32> kd, u32 D) {
    typedef typename Signed<W>::type SW;
    const size_t n = (size_t)R.n;
    const size_t L = (size_t)R.L;
    c += (size_t)blockIdx.y * L * n;                       // blockIdx.y: which ring element
    digits += (size_t)blockIdx.y * D * L * n;
    ALCH_WALK_INIT(R.n, R.L);
    ALCH_WALK(w, L * n, wk) {
        const size_t k = wk.k, i = wk.mid;
        const W qi = R.mod[i].q;
        const W x = c[i * n + k];
        SW v = x > ((qi - 1) >> 1) ? (SW)x - (SW)qi : (SW)x;
        for (u32 t = 0; t < kd.v[i]; ++t) {
            SW d;
            if (t + 1 < kd.v[i]) {
                d = v & 1 ? (SW)-1 : (SW)0;          // v mod 2 in {0,1}; remainder 1 is taken as -1 (2r >= b)
                v = (v - d) / 2;
            } else {
                d = v;
            }
            W* out = digits + (size_t)(first_digit.v[i] + t) * L * n;
            for (size_t j = 0; j < L; ++j) out[j * n + k] = (W)reduce_signed<W>(d, R.mod[j].q);
        }
    }
}

// SymmSHE (*) on linear ciphertexts, element-wise on the CRT basis: c0 = a0 b0 s, c1 = (a0 b1 + a1 b0) s -> out,
// c2 = a1 b1 s -> c2buf (one element per ciphertext).  sr2 = s R^2 (Montgomery).  Used by the BaseBGad key switch.
template <typename W> struct GTab { const W* p[MAXL]; };     // per limb: CRT image of g (Montgomery form), or null

template <typename W, int VW = 1>
__global__ void k_tensor_ew(DevRing<W> R, const W* a, const W* b, W* out, W* c2buf, size_t nct, Scal<W> sr2, int dup,
                            W* c2crt, GTab<W> gt, int c2_compact = 0) {
    // c2_compact: c2buf holds the L - dup limbs of the operands' ring only ([ct][L - dup][n]): the added limbs of c2 are zero and
    // neither their crtInv nor their digits are ever computed
    // gt: general index -- SymmSHE's (*) applies mulG to every product coefficient (mulGCRT = pointwise product with the
    // CRT image of g); all-null for a two-power index, where g = 1
    // c2crt != null: a second copy of c2 that stays in the CRT basis (the diagonal digits of the key switch)
    // R: the ring of out / c2buf (L limbs); a, b live on its last L - dup limbs; the dup leading limbs of the
    // results are zero (modSwitch up: Rescale b -> (a,b), its q_a factor folded into sr2 by the host)
    typedef Pack<W, VW> P;
    const size_t n = (size_t)R.n;
    const size_t Ln = (size_t)R.L * n, Lsn = (size_t)(R.L - dup) * n, off = (size_t)dup * n;
    ALCH_WALK_INIT(R.n / VW, R.L);
    ALCH_WALK(w, nct * Ln / VW, wk) {
        const size_t ct = wk.outer, rem = (size_t)wk.mid * n + (size_t)wk.k * VW;
        P c0, c1, c2;
        if (rem < off) {
#pragma unroll
            for (int c = 0; c < VW; ++c) { c0.v[c] = 0; c1.v[c] = 0; c2.v[c] = 0; }
        } else {
            const size_t rs = rem - off, js = wk.mid - (u32)dup;
            const ModP<W> m = R.mod[wk.mid];
            const P a0 = *reinterpret_cast<const P*>(a + 2 * ct * Lsn + rs), a1 = *reinterpret_cast<const P*>(a + (2 * ct + 1) * Lsn + rs);
            const P b0 = *reinterpret_cast<const P*>(b + 2 * ct * Lsn + rs), b1 = *reinterpret_cast<const P*>(b + (2 * ct + 1) * Lsn + rs);
            const W* g = gt.p[wk.mid];
            P gv;
            if (g) gv = *reinterpret_cast<const P*>(g + (size_t)wk.k * VW);
#pragma unroll
            for (int c = 0; c < VW; ++c) {
                const W x0 = mont_mul(a0.v[c], sr2.v[js], m), x1 = mont_mul(a1.v[c], sr2.v[js], m);      // a s R
                W c0v = mont_mul(b0.v[c], x0, m);
                W c1v = add_mod(mont_mul(b1.v[c], x0, m), mont_mul(b0.v[c], x1, m), m.q);
                W c2v = mont_mul(b1.v[c], x1, m);
                if (g) { c0v = mont_mul(c0v, gv.v[c], m); c1v = mont_mul(c1v, gv.v[c], m); c2v = mont_mul(c2v, gv.v[c], m); }
                c0.v[c] = c0v; c1.v[c] = c1v; c2.v[c] = c2v;
            }
        }
        *reinterpret_cast<P*>(out + 2 * ct * Ln + rem) = c0;
        *reinterpret_cast<P*>(out + (2 * ct + 1) * Ln + rem) = c1;
        if (!c2_compact) *reinterpret_cast<P*>(c2buf + ct * Ln + rem) = c2;
        else if (rem >= off) *reinterpret_cast<P*>(c2buf + ct * Lsn + (rem - off)) = c2;
        if (c2crt) *reinterpret_cast<P*>(c2crt + ct * Ln + rem) = c2;
    }
}

// keySwitchQuadCirc's inner product for a many-digit gadget: out_c += sum_d digit_d * hint_{d,c} (CRT basis).
// digits: [ct][D][L][n]; hint: [D][2][L][n] in Montgomery form.
template <typename W>
__global__ void k_hint_mac(DevRing<W> R, W* out, const W* digits, const W* hint, size_t nct, u32 D, const W* diag = nullptr,
                           u32 grp = 0, u32 hskip = 0, const u32* slot_e = nullptr, u32 n_d = 0) {
    // slot_e != null (tunnel, E'-level transforms): the digits are CRT vectors of dimension n_d over E'; slot s of S' reads
    // entry slot_e[s] (embedCRT replication).
    // hskip != 0 (tunnel on ciphertexts `hskip` limbs below the hint's ring): the digits come in groups of grp = L - hskip
    // source limbs per embedded coefficient and the hint rows of the absent (zero) limbs are skipped: digit d uses hint row
    // d + (d / grp + 1) * hskip.
    // diag != null (TrivGad, D = L): digit d reduced into its own limb d is c2's limb d itself, read from the CRT-basis
    // copy `diag` [ct][L][n] instead of a transformed digit (those slots of `digits` are never written).
    // One thread owns one (limb, slot) of TILE consecutive ciphertexts, so a hint word is loaded once per TILE
    // products (the hint is the larger stream for a many-digit gadget: 2 D words against D per ciphertext).
    constexpr int TILE = 4;          // 8 was measured slower (HomomRLWR pipeline 18.6 k -> 17.3 k/s): fewer, longer threads
    const size_t n = (size_t)R.n;
    const size_t Ln = (size_t)R.L * n;
    const size_t ntile = (nct + TILE - 1) / TILE;
    ALCH_WALK_INIT(R.n, R.L);
    ALCH_WALK(w, ntile * Ln, wk) {
        const size_t ct0 = wk.outer * TILE, rem = (size_t)wk.mid * n + wk.k;
        const u32 limb = wk.mid;
        const ModP<W> m = R.mod[limb];
        W acc0[TILE], acc1[TILE];
#pragma unroll
        for (int c = 0; c < TILE; ++c) {
            const bool live = ct0 + c < nct;
            acc0[c] = live ? out[2 * (ct0 + c) * Ln + rem] : (W)0;
            acc1[c] = live ? out[(2 * (ct0 + c) + 1) * Ln + rem] : (W)0;
        }
        for (u32 d = 0; d < D; ++d) {
            const u32 hd = hskip ? d + (d / grp + 1) * hskip : d;
            const W h0 = hint[(size_t)(2 * hd) * Ln + rem], h1 = hint[(size_t)(2 * hd + 1) * Ln + rem];
#pragma unroll
            for (int c = 0; c < TILE; ++c) {
                if (ct0 + c >= nct) continue;
                const size_t ct = ct0 + c;
                const W x = slot_e ? digits[((ct * (size_t)D + d) * R.L + limb) * (size_t)n_d + slot_e[wk.k]]
                                   : (diag && hd == limb) ? diag[ct * Ln + rem] : digits[(ct * (size_t)D + d) * Ln + rem];
                acc0[c] = add_mod(acc0[c], mont_mul(x, h0, m), m.q);
                acc1[c] = add_mod(acc1[c], mont_mul(x, h1, m), m.q);
            }
        }
#pragma unroll
        for (int c = 0; c < TILE; ++c) {
            if (ct0 + c >= nct) continue;
            out[2 * (ct0 + c) * Ln + rem] = acc0[c];
            out[(2 * (ct0 + c) + 1) * Ln + rem] = acc1[c];
        }
    }
}

// The same with 16-byte accesses: a thread owns one piece of VL consecutive slots of TILE ciphertexts (n a multiple of VL).
// The one-word form moved 2.4 TB/s in the HomomRLWR pipeline -- bound by the number of memory instructions, not by HBM.
template <typename W, int TILE>
__global__ void k_hint_mac_v(DevRing<W> R, W* out, const W* digits, const W* hint, size_t nct, u32 D, const W* diag, u32 grp, u32 hskip,
                             const u32* slot_e, u32 n_d) {
    typedef typename Vec4<W>::type V;
    // TILE ciphertexts share every hint piece.  Measured on the HomomRLWR pipeline with E'-level digits (small vectors served by
    // L2, the hint rows being what comes from HBM): TILE 2 -> 29.3 k, 4 -> 29.5 k, 8 -> 23.5 k ringRounds/s
    constexpr int VL = Vec4<W>::LANES;
    const size_t n = (size_t)R.n;
    const size_t Ln = (size_t)R.L * n;
    const size_t ntile = (nct + TILE - 1) / TILE;
    const size_t nv = n / VL;                                  // 16-byte pieces per limb-polynomial
    ALCH_WALK_INIT(nv, R.L);
    ALCH_WALK(w, ntile * (size_t)R.L * nv, wk) {
        const size_t ct0 = wk.outer * TILE, rem = (size_t)wk.mid * n + (size_t)wk.k * VL;
        const u32 limb = wk.mid;
        const W q = R.mod[limb].q, qni = R.mod[limb].qni;
        // E'-level digits: the VL slots of this piece read entries se[0..VL) of the small CRT vector; they are consecutive and
        // 16-byte aligned whenever the innermost prime-power factor of s' divides e' with the same exponent (every hop of the
        // reference), else gathered one by one
        u32 se[VL];
        bool piece = false;
        if (slot_e) {
#pragma unroll
            for (int e = 0; e < VL; ++e) se[e] = slot_e[(size_t)wk.k * VL + e];
            piece = (se[0] % VL == 0);
#pragma unroll
            for (int e = 1; e < VL; ++e) piece = piece && se[e] == se[0] + (u32)e;
        }
        V acc0[TILE], acc1[TILE];
#pragma unroll
        for (int c = 0; c < TILE; ++c) {
            const size_t ct = ct0 + c < nct ? ct0 + c : ct0;   // a dead lane of the tile recomputes ciphertext ct0 and stores nothing
            acc0[c] = *reinterpret_cast<const V*>(out + 2 * ct * Ln + rem);
            acc1[c] = *reinterpret_cast<const V*>(out + (2 * ct + 1) * Ln + rem);
        }
        for (u32 d = 0; d < D; ++d) {
            const u32 hd = hskip ? d + (d / grp + 1) * hskip : d;
            const V h0 = *reinterpret_cast<const V*>(hint + (size_t)(2 * hd) * Ln + rem);
            const V h1 = *reinterpret_cast<const V*>(hint + (size_t)(2 * hd + 1) * Ln + rem);
            const bool dg = diag && hd == limb;
#pragma unroll
            for (int c = 0; c < TILE; ++c) {
                const size_t ct = ct0 + c < nct ? ct0 + c : ct0;
                V x;
                if (slot_e) {
                    const W* dv = digits + ((ct * (size_t)D + d) * R.L + limb) * (size_t)n_d;
                    if (piece) x = *reinterpret_cast<const V*>(dv + se[0]);
                    else {
#pragma unroll
                        for (int e = 0; e < VL; ++e) x[e] = dv[se[e]];
                    }
                } else {
                    x = dg ? *reinterpret_cast<const V*>(diag + ct * Ln + rem)
                           : *reinterpret_cast<const V*>(digits + (ct * (size_t)D + d) * Ln + rem);
                }
#pragma unroll
                for (int e = 0; e < VL; ++e) {
                    acc0[c][e] = csub((W)(acc0[c][e] + csub(mont_mul_lazy(x[e], h0[e], q, qni), q)), q);
                    acc1[c][e] = csub((W)(acc1[c][e] + csub(mont_mul_lazy(x[e], h1[e], q, qni), q)), q);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < TILE; ++c) {
            if (ct0 + c >= nct) continue;
            *reinterpret_cast<V*>(out + 2 * (ct0 + c) * Ln + rem) = acc0[c];
            *reinterpret_cast<V*>(out + (2 * (ct0 + c) + 1) * Ln + rem) = acc1[c];
        }
    }
}

// The E'-level form of k_hint_mac_v on its own (tunnels whose digits were transformed at dimension phi(e'), every 16-byte piece of S'
// slots reading one aligned piece of the small vector: alch_tunnel::pieces_ok): without the other forms' address paths the kernel
// needs far fewer registers than k_hint_mac_v's 128 + scratch.
template <typename W, int TILE>
__global__ void k_hint_mac_e(DevRing<W> R, W* out, const W* digits, const W* hint, size_t nct, u32 D, u32 grp, u32 hskip,
                             const u32* slot_e, u32 n_d) {
    typedef typename Vec4<W>::type V;
    constexpr int VL = Vec4<W>::LANES;
    const size_t n = (size_t)R.n;
    const size_t Ln = (size_t)R.L * n;
    const size_t ntile = (nct + TILE - 1) / TILE;
    const size_t nv = n / VL;
    ALCH_WALK_INIT(nv, R.L);
    ALCH_WALK(w, ntile * (size_t)R.L * nv, wk) {
        const size_t ct0 = wk.outer * TILE, rem = (size_t)wk.mid * n + (size_t)wk.k * VL;
        const u32 limb = wk.mid;
        const W q = R.mod[limb].q, qni = R.mod[limb].qni;
        const u32 se = slot_e[(size_t)wk.k * VL];
        V acc0[TILE], acc1[TILE];
        const W* dv[TILE];
#pragma unroll
        for (int c = 0; c < TILE; ++c) {
            const size_t ct = ct0 + c < nct ? ct0 + c : ct0;   // a dead lane of the tile recomputes ciphertext ct0 and stores nothing
            acc0[c] = *reinterpret_cast<const V*>(out + 2 * ct * Ln + rem);
            acc1[c] = *reinterpret_cast<const V*>(out + (2 * ct + 1) * Ln + rem);
            dv[c] = digits + (ct * (size_t)D * R.L + limb) * (size_t)n_d + se;
        }
        const size_t dstep = (size_t)R.L * n_d;
        for (u32 d = 0; d < D; ++d) {
            const u32 hd = hskip ? d + (d / grp + 1) * hskip : d;
            const V h0 = *reinterpret_cast<const V*>(hint + (size_t)(2 * hd) * Ln + rem);
            const V h1 = *reinterpret_cast<const V*>(hint + (size_t)(2 * hd + 1) * Ln + rem);
#pragma unroll
            for (int c = 0; c < TILE; ++c) {
                const V x = *reinterpret_cast<const V*>(dv[c] + (size_t)d * dstep);
#pragma unroll
                for (int e = 0; e < VL; ++e) {
                    acc0[c][e] = csub((W)(acc0[c][e] + csub(mont_mul_lazy(x[e], h0[e], q, qni), q)), q);
                    acc1[c][e] = csub((W)(acc1[c][e] + csub(mont_mul_lazy(x[e], h1[e], q, qni), q)), q);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < TILE; ++c) {
            if (ct0 + c >= nct) continue;
            *reinterpret_cast<V*>(out + 2 * (ct0 + c) * Ln + rem) = acc0[c];
            *reinterpret_cast<V*>(out + (2 * (ct0 + c) + 1) * Ln + rem) = acc1[c];
        }
    }
}

// Round 4: the tunnel's hint inner product with lazy 64-bit accumulation, and evalLin's constant term as its starting value.
//
// k_hint_mac_e spends 7 VALU instructions per (slot, digit, hint row): a Montgomery product (3), its conditional subtraction, the
// addition and the sum's conditional subtraction.  On the hops whose digits were transformed in E' (every S' slot re-reads an E'
// entry but multiplies it by its OWN hint word) that made the kernel VALU-bound: 5.7 10^11 slot-piece-digits/s x 56 lane
// instructions is 82 % of the chip's integer issue rate on Tunnel.hs's first hop, while the digits arrive at 2.3 TB/s, half of
// what a copy moves (profiles/r04_tunnel_hs_kernel_stats_before.csv).  Here a group of K products is summed in 64 bits and reduced
// once: with t the running sum in [0, 2q), acc = t * (R mod q) + sum_{K} x h stays below q 2^32 as long as (K + 2) q < 2^32, and one
// Montgomery reduction of acc is again the running sum in [0, 2q).  K = floor((2^32 - 1) / q) - 2: 5 for Tunnel.hs's moduli
// (examples/Tunnel.hs:34-39, ~2^29), 3 for HomomRLWR's three rounding moduli, 0 for its 30.5-bit ones -- those limbs keep the
// product-at-a-time form.  Per (slot, digit, row): (K + 1 + 3) / K instructions = 1.8 at K = 5.  Same residues: the sum is the
// same element of Z_q, reduced to [0, q) at the end.
// lin != null: c0' starts from sum_i crt(x0_i) y_i (evalLin on the constant term, what k_tunnel_lin computed into `out` in a pass
// of its own) and c1' from zero -- one kernel launch, one write and one read of the output ciphertexts less per tunnel.
template <int TILE>
__global__ void __launch_bounds__(256) k_tunnel_mac_e(DevRing<u32> R, u32* out, const u32* digits, const u32* hint, size_t nct, u32 D, u32 grp, u32 hskip,
                                                      const u32* slot_e, u32 n_d, const u32* x0crt, const u32* lin, u32 d_rel, u32 Lx, u32 xoff) {
    typedef u32 W;
    typedef typename Vec4<W>::type V;
    constexpr int VL = 4;
    const size_t n = (size_t)R.n;
    const size_t Ln = (size_t)R.L * n;
    const size_t ntile = (nct + TILE - 1) / TILE;
    const size_t nv = n / VL;
    ALCH_WALK_INIT(nv, R.L);
    ALCH_WALK(w, ntile * (size_t)R.L * nv, wk) {
        const size_t ct0 = wk.outer * TILE, rem = (size_t)wk.mid * n + (size_t)wk.k * VL;
        const u32 limb = wk.mid;
        const W q = R.mod[limb].q, qni = R.mod[limb].qni;
        const u32 se = slot_e ? slot_e[(size_t)wk.k * VL] : wk.k * (u32)VL;      // no slot table: the digits were transformed in S' itself
        const u32 kmax = 0xFFFFFFFFu / q;                       // (K + 2) q < 2^32
        const u32 K = kmax >= 4 ? (kmax - 2 > 8 ? 8u : kmax - 2) : 0u;
        const bool lazy = K >= 2;
        const W r1 = lazy ? R.mod[limb].r1 : (W)1;            // not lazy: the accumulators hold the running sum itself, in [0, q)
        u64 a0[TILE][VL], a1[TILE][VL];
        const W* dv[TILE];
        // starting values, as acc = value * (R mod q) < q^2
#pragma unroll
        for (int c = 0; c < TILE; ++c) {
            const size_t ct = ct0 + c < nct ? ct0 + c : ct0;   // a dead lane of the tile recomputes ciphertext ct0 and stores nothing
            dv[c] = digits + (ct * (size_t)D * R.L + limb) * (size_t)n_d + se;
            if (lin) {
#pragma unroll
                for (int e = 0; e < VL; ++e) { a0[c][e] = 0; a1[c][e] = 0; }
                if (limb >= xoff) {
                    for (u32 i = 0; i < d_rel; ++i) {
                        const V x = *reinterpret_cast<const V*>(x0crt + ((ct * d_rel + i) * (size_t)Lx + (limb - xoff)) * (size_t)n_d + se);
                        const V y = *reinterpret_cast<const V*>(lin + (size_t)i * Ln + rem);
#pragma unroll
                        for (int e = 0; e < VL; ++e) {
                            const W t = csub(mont_mul_lazy(x[e], y[e], q, qni), q);
                            a0[c][e] = (u64)csub((W)((W)a0[c][e] + t), q);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < VL; ++e) a0[c][e] = a0[c][e] * r1;
                }
            } else {
                const V o0 = *reinterpret_cast<const V*>(out + 2 * ct * Ln + rem);
                const V o1 = *reinterpret_cast<const V*>(out + (2 * ct + 1) * Ln + rem);
#pragma unroll
                for (int e = 0; e < VL; ++e) { a0[c][e] = (u64)o0[e] * r1; a1[c][e] = (u64)o1[e] * r1; }
            }
        }
        const size_t dstep = (size_t)R.L * n_d;
        auto redc = [&](u64 t) -> W { const W m = (W)t * qni; return (W)((t + (u64)m * q) >> 32); };     // t < q 2^32 -> [0, 2q)
        if (lazy) {
            for (u32 d0 = 0; d0 < D; d0 += K) {
                const u32 dend = d0 + K < D ? d0 + K : D;
                for (u32 d = d0; d < dend; ++d) {
                    const u32 hd = hskip ? d + (d / grp + 1) * hskip : d;
                    const V h0 = *reinterpret_cast<const V*>(hint + (size_t)(2 * hd) * Ln + rem);
                    const V h1 = *reinterpret_cast<const V*>(hint + (size_t)(2 * hd + 1) * Ln + rem);
#pragma unroll
                    for (int c = 0; c < TILE; ++c) {
                        const V x = *reinterpret_cast<const V*>(dv[c] + (size_t)d * dstep);
#pragma unroll
                        for (int e = 0; e < VL; ++e) {
                            a0[c][e] += (u64)x[e] * h0[e];
                            a1[c][e] += (u64)x[e] * h1[e];
                        }
                    }
                }
                if (dend < D) {                                 // carry the running sum into the next group: t (R mod q) < 2 q^2
#pragma unroll
                    for (int c = 0; c < TILE; ++c)
#pragma unroll
                        for (int e = 0; e < VL; ++e) { a0[c][e] = (u64)redc(a0[c][e]) * r1; a1[c][e] = (u64)redc(a1[c][e]) * r1; }
                }
            }
        } else {
            // moduli too close to 2^31 for a lazy group: one reduction per product, running sum as acc = t (R mod q)
            for (u32 d = 0; d < D; ++d) {
                const u32 hd = hskip ? d + (d / grp + 1) * hskip : d;
                const V h0 = *reinterpret_cast<const V*>(hint + (size_t)(2 * hd) * Ln + rem);
                const V h1 = *reinterpret_cast<const V*>(hint + (size_t)(2 * hd + 1) * Ln + rem);
#pragma unroll
                for (int c = 0; c < TILE; ++c) {
                    const V x = *reinterpret_cast<const V*>(dv[c] + (size_t)d * dstep);
#pragma unroll
                    for (int e = 0; e < VL; ++e) {
                        a0[c][e] = (u64)csub((W)((W)a0[c][e] + csub(mont_mul_lazy(x[e], h0[e], q, qni), q)), q);
                        a1[c][e] = (u64)csub((W)((W)a1[c][e] + csub(mont_mul_lazy(x[e], h1[e], q, qni), q)), q);
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < TILE; ++c) {
            if (ct0 + c >= nct) continue;
            V r0, r1v;
#pragma unroll
            for (int e = 0; e < VL; ++e) {
                r0[e] = lazy ? csub(redc(a0[c][e]), q) : (W)a0[c][e];
                r1v[e] = lazy ? csub(redc(a1[c][e]), q) : (W)a1[c][e];
            }
            *reinterpret_cast<V*>(out + 2 * (ct0 + c) * Ln + rem) = r0;
            *reinterpret_cast<V*>(out + (2 * (ct0 + c) + 1) * Ln + rem) = r1v;
        }
    }
}

// Rescale (a,b) -> b: dst limb j-1 = q_0^-1 (src_j - reduce(lift src_0)).  q0inv_m[j] = q_0^-1 mod q_j (Montgomery).
template <typename W>
__global__ void k_rescale_drop0(DevRing<W> R, const W* src, W* dst, size_t elems, Scal<W> q0inv_m) {
    typedef typename Signed<W>::type SW;
    const size_t n = (size_t)R.n;
    const size_t L = (size_t)R.L;
    const size_t total = elems * (L - 1) * n;
    const W q0 = R.mod[0].q;
    ALCH_WALK_INIT(R.n, R.L - 1);
    ALCH_WALK(w, total, wk) {
        const size_t k = wk.k, e = wk.outer;
        const size_t j = (size_t)wk.mid + 1;
        const ModP<W> m = R.mod[j];
        const W x0 = src[(e * L) * n + k];
        const SW z = x0 > ((q0 - 1) >> 1) ? (SW)x0 - (SW)q0 : (SW)x0;
        const SW zr = reduce_signed<W>(z, m.q);
        const W d = sub_mod(src[(e * L + j) * n + k], (W)zr, m.q);
        dst[w] = mont_mul(d, q0inv_m.v[j], m);
    }
}

// Rescale b -> (a,b): dst limb 0 = 0, dst limb j+1 = q_a * src limb j.  qa_m[j+1] = q_a mod q_{j+1} (Montgomery).
template <typename W>
__global__ void k_rescale_add0(DevRing<W> Rd, const W* src, W* dst, size_t elems, Scal<W> qa_m) {
    const size_t n = (size_t)Rd.n;
    const size_t L = (size_t)Rd.L;
    const size_t total = elems * L * n;
    ALCH_WALK_INIT(Rd.n, Rd.L);
    ALCH_WALK(w, total, wk) {
        const size_t k = wk.k, j = wk.mid, e = wk.outer;
        dst[w] = j == 0 ? (W)0 : mont_mul(src[(e * (L - 1) + (j - 1)) * n + k], qa_m.v[j], Rd.mod[j]);
    }
}

// mulGCRT / divGCRT: element-wise product with a per-limb table of n words (Montgomery form).
template <typename W>
__global__ void k_mul_table(DevRing<W> R, W* data, size_t words, GTab<W> gt) {
    ALCH_WALK_INIT(R.n, R.L);
    ALCH_WALK(w, words, wk) data[w] = mont_mul(data[w], gt.p[wk.mid][wk.k], R.mod[wk.mid]);
}

// SymmSHE mulPublic (Eval.hs:132): every ring element of src times one public ring element (CRT basis, pointwise).
template <typename W>
__global__ void k_mul_bcast(DevRing<W> R, W* dst, const W* src, const W* pub, size_t words) {
    const size_t n = (size_t)R.n;
    ALCH_WALK_INIT(R.n, R.L);
    ALCH_WALK(w, words, wk) {
        const ModP<W> m = R.mod[wk.mid];
        dst[w] = mont_mul(mont_mul(src[w], m.r2, m), pub[(size_t)wk.mid * n + wk.k], m);
    }
}

// SymmSHE addPublic (Eval.hs:131): one public ring element added to the c0 of every ciphertext (elements 0, 2, 4, ..).
template <typename W>
__global__ void k_add_bcast(DevRing<W> R, W* dst, const W* pub, size_t cts) {
    const size_t n = (size_t)R.n, Ln = (size_t)R.L * n;
    ALCH_WALK_INIT(R.n, R.L);
    ALCH_WALK(w, cts * Ln, wk) {
        const size_t ct = wk.outer, rem = (size_t)wk.mid * n + wk.k;
        W* p = dst + 2 * ct * Ln + rem;
        *p = add_mod(*p, pub[rem], R.mod[wk.mid].q);
    }
}

// SymmSHE addPublic in one pass: dst = src * s_j, c0 components (even elements) += pub.  Element-major walk with VW-word pieces.
template <typename W, int VW = 1>
__global__ void k_scale_add_bcast(DevRing<W> R, W* dst, const W* src, const W* pub, size_t words, Scal<W> sm) {
    typedef Pack<W, VW> P;
    ALCH_WALK_INIT(R.n / VW, R.L);
    ALCH_WALK(w, words / VW, wk) {
        P x = reinterpret_cast<const P*>(src)[w];
        const ModP<W> mp = R.mod[wk.mid];
#pragma unroll
        for (int c = 0; c < VW; ++c) x.v[c] = mont_mul(x.v[c], sm.v[wk.mid], mp);
        if ((wk.outer & 1u) == 0) {
            const P b = reinterpret_cast<const P*>(pub)[(size_t)wk.mid * (R.n / VW) + wk.k];
#pragma unroll
            for (int c = 0; c < VW; ++c) x.v[c] = add_mod(x.v[c], b.v[c], mp.q);
        }
        reinterpret_cast<P*>(dst)[w] = x;
    }
}

// Tunnel, step 1: the E'-coefficients of (c0, c1) embedded into S' (coeffs + embedPow as one index gather; toMSD's
// per-limb scalar folded in).  in: [ct][2][L - dup][n_r] -- the ciphertexts may live `dup` limbs below the tunnel's ring
// (PT2CT's modSwitch_ in front of tunnel_, PT2CT.hs:224-229: x -> (0, q_a x), the factor folded into s_m by the host);
// x0 / x1: [ct][d_rel][Lx][n_s] holding the limbs xoff .. xoff + Lx - 1 (compact: the zero limbs are left out).
template <typename W, int VW = 1>
__global__ void k_tunnel_gather(DevRing<W> Rs, const W* in0, const W* in1, W* x0, W* x1, const int32_t* table, u32 d_rel, u32 n_r, size_t nct,
                                Scal<W> s_m, int scale, u32 Lx, u32 xoff, u32 dup) {
    // in0 / in1: where the c0 / c1 components are read from (same [ct][2][..] layout: c0 may come from a scratch copy that went
    // through lInv while c1 is read in place)
    typedef Pack<W, VW> P;
    typedef Pack<int32_t, VW> PI;
    const size_t n = (size_t)Rs.n, Lin = (size_t)Rs.L - dup;
    const size_t per_ct = (size_t)d_rel * Lx * n;
    // words as [ct * 2 + comp][i * Lx + limb'][k]; a thread gathers VW consecutive k and stores them as one piece
    ALCH_WALK_INIT(Rs.n / VW, d_rel * Lx);
    ALCH_WALK(w, nct * 2 * per_ct / VW, wk) {
        const size_t ct = wk.outer >> 1, comp = wk.outer & 1, k = (size_t)wk.k * VW;
        const u32 i = wk.mid / Lx, limb = wk.mid - i * Lx + xoff;
        const size_t r2 = (size_t)wk.mid * n + k;
        const PI src = *reinterpret_cast<const PI*>(table + i * n + k);
        P v;
#pragma unroll
        for (int c = 0; c < VW; ++c) {
            W x = 0;
            if (src.v[c] >= 0 && limb >= dup) {
                x = (comp ? in1 : in0)[((2 * ct + comp) * Lin + (limb - dup)) * (size_t)n_r + (size_t)src.v[c]];
                if (scale) x = mont_mul(x, s_m.v[limb], Rs.mod[limb]);
            }
            v.v[c] = x;
        }
        *reinterpret_cast<P*>((comp ? x1 : x0) + ct * per_ct + r2) = v;
    }
}

// Tunnel, step 2: c0' = sum_i crt(x0_i) * y_i (evalLin on the constant term), c1' = 0.  out: [ct][2][L][n];
// x0crt: [ct][d_rel][Lx][n] holding the limbs xoff .. (the limbs in front of xoff are zero).
template <typename W, int VW = 1>
__global__ void k_tunnel_lin(DevRing<W> Rs, W* out, const W* x0crt, const W* lin, u32 d_rel, size_t nct, u32 Lx, u32 xoff,
                             const u32* slot_e, u32 n_x) {
    // slot_e != null: x0crt holds CRT vectors of dimension n_x over E' ([ct][d_rel][Lx][n_x]); slot s of S' reads entry slot_e[s]
    typedef Pack<W, VW> P;
    const size_t n = (size_t)Rs.n, Ln = (size_t)Rs.L * n, Lxn = (size_t)Lx * (slot_e ? (size_t)n_x : n);
    ALCH_WALK_INIT(Rs.n / VW, Rs.L);
    ALCH_WALK(w, nct * Ln / VW, wk) {
        const size_t ct = wk.outer, rem = (size_t)wk.mid * n + (size_t)wk.k * VW;
        const ModP<W> m = Rs.mod[wk.mid];
        P acc, zero;
#pragma unroll
        for (int c = 0; c < VW; ++c) { acc.v[c] = 0; zero.v[c] = 0; }
        if (wk.mid >= xoff) {
            const size_t xr = slot_e ? (size_t)(wk.mid - xoff) * n_x : (size_t)(wk.mid - xoff) * n + (size_t)wk.k * VW;
            for (u32 i = 0; i < d_rel; ++i) {
                P x;
                if (slot_e) {
#pragma unroll
                    for (int c = 0; c < VW; ++c) x.v[c] = x0crt[(ct * d_rel + i) * Lxn + xr + slot_e[(size_t)wk.k * VW + c]];
                } else {
                    x = *reinterpret_cast<const P*>(x0crt + (ct * d_rel + i) * Lxn + xr);
                }
                const P y = *reinterpret_cast<const P*>(lin + (size_t)i * Ln + rem);
#pragma unroll
                for (int c = 0; c < VW; ++c) acc.v[c] = add_mod(acc.v[c], mont_mul(x.v[c], y.v[c], m), m.q);
            }
        }
        *reinterpret_cast<P*>(out + 2 * ct * Ln + rem) = acc;
        *reinterpret_cast<P*>(out + (2 * ct + 1) * Ln + rem) = zero;
    }
}

// Rescale b -> (a_1, .., a_dup, b): dst limb j < dup = 0, dst limb j >= dup = (prod of the added moduli) * src limb j - dup.
template <typename W>
__global__ void k_rescale_up(DevRing<W> Rd, const W* src, W* dst, size_t elems, int dup, Scal<W> mult_m) {
    const size_t n = (size_t)Rd.n, L = (size_t)Rd.L, Ls = L - (size_t)dup;
    ALCH_WALK_INIT(Rd.n, Rd.L);
    ALCH_WALK(w, elems * L * n, wk) {
        const size_t k = wk.k, j = wk.mid, e = wk.outer;
        dst[w] = j < (size_t)dup ? (W)0 : mont_mul(src[(e * Ls + (j - dup)) * n + k], mult_m.v[j], Rd.mod[j]);
    }
}

template <typename W>
__global__ void k_checksum(const W* data, size_t words, u64* sum, u64 w0) {
    u64 acc = 0;
    for (size_t w = blockIdx.x * (size_t)blockDim.x + threadIdx.x; w < words; w += (size_t)gridDim.x * blockDim.x)
        acc += splitmix64(((u64)w + w0) ^ ((u64)data[w] << 20));
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long*)sum, (unsigned long long)acc);
}

static inline unsigned ew_grid(size_t items) {
    size_t g = (items + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ------------------------------------------------------------------------------------------------------
// ring construction
// ------------------------------------------------------------------------------------------------------
template <typename W>
static int build_dev_ring(alch_ring* r, DevRing<W>& d) {
    const size_t n = r->n;
    const int L = r->L;
    std::vector<W> all(2 * (size_t)L * n);
    HIP_TRY(hipMalloc(&r->tables, all.size() * sizeof(W)));
    memset(&d, 0, sizeof d);
    d.L = L;
    d.logn = r->logn;
    d.n = r->n;
    u64 maxhalf = 0;
    for (int j = 0; j < L; ++j) maxhalf = std::max(maxhalf, (r->q[j] - 1) / 2);
    for (int j = 0; j < L; ++j) {
        const u64 q = r->q[j];
        d.mod[j] = make_modp<W>(q);
        const u64 psi = h_root(q, r->m);
        std::vector<W> f, inv;
        h_build_twiddles<W>(q, psi, r->logn, f, inv);
        memcpy(&all[(size_t)(2 * j) * n], f.data(), n * sizeof(W));
        memcpy(&all[(size_t)(2 * j + 1) * n], inv.data(), n * sizeof(W));
        d.twf[j] = reinterpret_cast<W*>(r->tables) + (size_t)(2 * j) * n;
        d.twi[j] = reinterpret_cast<W*>(r->tables) + (size_t)(2 * j + 1) * n;
        const u64 ninv = h_powmod(n % q, q - 2, q);
        const u64 r1 = d.mod[j].r1;
        d.ninv_m[j] = (W)h_mulmod(ninv, r1, q);
        // inv[1] is tw^-1[1] * R; times n^-1 stays in Montgomery form
        d.w1ninv_m[j] = (W)h_mulmod((u64)inv[1], ninv, q);
        d.dig_off[j] = (W)(((maxhalf + q - 1) / q) * q);
    }
    HIP_TRY(hipMemcpy(r->tables, all.data(), all.size() * sizeof(W), hipMemcpyHostToDevice));
    if (sizeof(W) == 4) {
        // forward twiddles once more as Plantard constants: tw[k] = psi^brev(k)
        std::vector<u64> pl((size_t)L * n);
        for (int j = 0; j < L; ++j) {
            const u64 q = r->q[j];
            const u64 psi = h_root(q, r->m);
            std::vector<u64> pw(n);
            u64 acc = 1;
            for (size_t i = 0; i < n; ++i) { pw[i] = acc; acc = h_mulmod(acc, psi, q); }
            for (size_t k = 0; k < n; ++k) pl[(size_t)j * n + k] = h_plant_const(pw[h_brev((u32)k, r->logn)], q);
        }
        HIP_TRY(hipMalloc(&r->tables_p, pl.size() * sizeof(u64)));
        HIP_TRY(hipMemcpy(r->tables_p, pl.data(), pl.size() * sizeof(u64), hipMemcpyHostToDevice));
        for (int j = 0; j < L; ++j) d.twp[j] = reinterpret_cast<const u64*>(r->tables_p) + (size_t)j * n;
    } else {
        // 64-bit rings: the transforms multiply by Shoup pairs (plain twiddle, floor(w 2^64 / q)); the Montgomery
        // tables above still serve the hand-written stage 0 of the split kernels
        std::vector<Sh64> sh(2 * (size_t)L * n);
        for (int j = 0; j < L; ++j) {
            const u64 q = r->q[j];
            const u64 psi = h_root(q, r->m), ipsi = h_powmod(psi, q - 2, q);
            std::vector<u64> pw(n), ipw(n);
            u64 a = 1, b = 1;
            for (size_t i = 0; i < n; ++i) { pw[i] = a; ipw[i] = b; a = h_mulmod(a, psi, q); b = h_mulmod(b, ipsi, q); }
            for (size_t k = 0; k < n; ++k) {
                const u32 e = h_brev((u32)k, r->logn);
                sh[(size_t)(2 * j) * n + k] = h_shoup_const(pw[e], q);
                sh[(size_t)(2 * j + 1) * n + k] = h_shoup_const(ipw[e], q);
            }
        }
        HIP_TRY(hipMalloc(&r->tables_p, sh.size() * sizeof(Sh64)));
        HIP_TRY(hipMemcpy(r->tables_p, sh.data(), sh.size() * sizeof(Sh64), hipMemcpyHostToDevice));
        for (int j = 0; j < L; ++j) {
            d.tws[j] = reinterpret_cast<const Sh64*>(r->tables_p) + (size_t)(2 * j) * n;
            d.twsi[j] = reinterpret_cast<const Sh64*>(r->tables_p) + (size_t)(2 * j + 1) * n;
        }
    }
    return ALCH_OK;
}

template <typename W> static GenDev<W>& gen_dev(alch_ring* r);
template <> GenDev<u32>& gen_dev<u32>(alch_ring* r) { return r->g32; }
template <> GenDev<u64>& gen_dev<u64>(alch_ring* r) { return r->g64; }

// General-index ring: moduli / Montgomery constants into DevRing, pass plan and tables into GenDev.
template <typename W>
static int build_gen_ring(alch_ring* r, DevRing<W>& d, GenDev<W>& g) {
    const GenHost& h = r->gh;
    const int L = r->L;
    memset(&d, 0, sizeof d);
    memset(&g, 0, sizeof g);
    d.L = L; d.logn = 0; d.n = h.n;
    g.n = h.n; g.npass = h.npass; g.nfact = h.nfact; g.rad = h.rad;
    for (int i = 0; i < h.npass; ++i) g.pass[i] = h.pass[i];
    for (int i = 0; i < h.nfact; ++i) g.fact[i] = h.fact[i];
    g.plain = (!r->has_crt && !r->zdom) ? 1 : 0;
    g.smallq = r->has_crt ? 1 : 0;
    for (int j = 0; j < L; ++j) if (r->q[j] >= 1753413056ull) g.smallq = 0;      // 6 q^2 < 2^64
    if (g.smallq) {                                                                // 6 q < 2^32: examples/Tunnel.hs's moduli
        g.smallq = 2;
        for (int j = 0; j < L; ++j) if (r->q[j] >= 715827882ull) g.smallq = 1;
    }
    u64 maxhalf = 0;
    for (int j = 0; j < L; ++j) maxhalf = std::max(maxhalf, r->q[j] ? (r->q[j] - 1) / 2 : 0);
    if (!r->has_crt) {
        for (int j = 0; j < L; ++j) {
            d.mod[j].q = (W)r->q[j]; d.mod[j].qni = 0; d.mod[j].r1 = 1; d.mod[j].r2 = 1;
            // plain inverse of the odd radical (0 when it is not a unit); the integers divide exactly instead
            u64 inv = 0;
            if (r->q[j]) {
                // extended Euclid
                int64_t r0 = (int64_t)r->q[j], r1 = (int64_t)(h.rad % r->q[j]), t0 = 0, t1 = 1;
                while (r1) { int64_t k = r0 / r1, r2 = r0 - k * r1, t2 = t0 - k * t1; r0 = r1; r1 = r2; t0 = t1; t1 = t2; }
                inv = r0 == 1 ? (u64)(((t0 % (int64_t)r->q[j]) + (int64_t)r->q[j]) % (int64_t)r->q[j]) : 0;
            }
            g.radinv_m[j] = (W)inv;
        }
        return ALCH_OK;
    }
    const size_t per_limb = 2 * (size_t)h.block_words + 2 * (size_t)h.n;
    std::vector<W> all(per_limb * L);
    HIP_TRY(hipMalloc(&r->gen_tables, all.size() * sizeof(W)));
    const int bits = 8 * (int)sizeof(W);
    for (int j = 0; j < L; ++j) {
        const u64 q = r->q[j];
        d.mod[j] = make_modp<W>(q);
        d.dig_off[j] = (W)(((maxhalf + q - 1) / q) * q);
        const u64 r1 = h_powmod(2, (u64)bits, q);
        std::vector<u64> f, iv, gc, gci;
        u64 iscale = 1;
        if (!gen_tables(h, q, f, iv, gc, gci, iscale)) return fail(ALCH_E_INVALID, "general-index table construction failed");
        W* base = all.data() + per_limb * j;
        for (u32 k = 0; k < h.block_words; ++k) { base[k] = (W)h_mulmod(f[k], r1, q); base[h.block_words + k] = (W)h_mulmod(iv[k], r1, q); }
        for (u32 k = 0; k < h.n; ++k) { base[2 * h.block_words + k] = (W)h_mulmod(gc[k], r1, q); base[2 * h.block_words + h.n + k] = (W)h_mulmod(gci[k], r1, q); }
        W* dev = reinterpret_cast<W*>(r->gen_tables) + per_limb * j;
        g.tabf[j] = dev; g.tabi[j] = dev + h.block_words;
        g.gcrt[j] = dev + 2 * h.block_words; g.gcrt_inv[j] = dev + 2 * h.block_words + h.n;
        g.iscale_m[j] = (W)h_mulmod(iscale, r1, q);
        g.radinv_m[j] = (h.rad % q) ? (W)h_mulmod(h_invmod(h.rad % q, q), r1, q) : (W)0;
    }
    HIP_TRY(hipMemcpy(r->gen_tables, all.data(), all.size() * sizeof(W), hipMemcpyHostToDevice));
    return ALCH_OK;
}

static bool two_power_engine(uint32_t m) { return m >= 32 && (m & (m - 1)) == 0; }

// Status order of alch_ring_create (the order a Lol host needs, haskell/.../GT.hs `ringFor`):
//   1. malformed arguments                                  -> ALCH_E_INVALID
//   2. NO CRT BASIS over the base ring -- a modulus that is composite, 2, or a prime that is not 1 mod m: the cases in which Lol's
//      `crtFuncs` / `crtInfo` answer Nothing for ZqBasic (CRTrans Maybe needs a prime q with m | q - 1; pairs need both) -> ALCH_E_NO_CRT,
//      whatever the index: it is a property of (m, q), not of this backend; the caller falls back to alch_ring_create_nocrt
//   3. a CRT basis exists but this backend does not serve the ring (prime factor of m above 13, a limb-polynomial larger than the
//      LDS, q >= 2^62)                                      -> ALCH_E_UNSUPPORTED: the caller keeps that (m, r) on lol-cpp as a whole
// ALCH_E_NOT_PRIME is only returned by alch_host_root.
static int classify_crt_moduli(uint32_t m, int L, const uint64_t* q) {
    if (!q || L < 1 || L > MAXL) return fail(ALCH_E_INVALID, "alch_ring_create: need 1 <= L <= 8 moduli");
    if (m < 1) return fail(ALCH_E_INVALID, "cyclotomic index must be >= 1");
    for (int j = 0; j < L; ++j) {
        if (q[j] < 2) return fail(ALCH_E_INVALID, "alch_ring_create: moduli must be >= 2 (the integers: alch_ring_create_nocrt with q = 0)");
        for (int i = 0; i < j; ++i)
            if (q[i] == q[j]) return fail(ALCH_E_INVALID, "RNS moduli must be distinct");
    }
    for (int j = 0; j < L; ++j) {
        if (q[j] >= (1ull << 62)) continue;                     // classified below as unsupported (primality is not tested up there)
        if (q[j] < 3 || !h_is_prime(q[j]))
            return fail(ALCH_E_NO_CRT, "modulus " + std::to_string(q[j]) + " is not an odd prime: no CRT basis over Z_q (Lol: crtFuncs = Nothing)");
        if ((q[j] - 1) % m) return fail(ALCH_E_NO_CRT, "modulus " + std::to_string(q[j]) + " is not 1 mod m: no CRT basis (Lol: crtFuncs = Nothing)");
    }
    for (int j = 0; j < L; ++j)
        if (q[j] >= (1ull << 62)) return fail(ALCH_E_UNSUPPORTED, "modulus must be below 2^62");
    return ALCH_OK;
}

// Argument checks of a general-index ring (or of a ring without CRT basis).  Fills gh; word size out.
static int validate_gen_args(uint32_t m, int L, const uint64_t* q, bool nocrt, GenHost& gh, int* word_out, bool* zdom_out) {
    if (!q || L < 1 || L > MAXL) return fail(ALCH_E_INVALID, "alch_ring_create: need 1 <= L <= 8 moduli");
    if (m < 1) return fail(ALCH_E_INVALID, "cyclotomic index must be >= 1");
    if (!nocrt) { const int rc = classify_crt_moduli(m, L, q); if (rc != ALCH_OK) return rc; }
    if (!gen_plan(m, gh)) return fail(ALCH_E_UNSUPPORTED, "cyclotomic index " + std::to_string(m) + ": " + gh.error);
    bool all32 = true, zdom = false;
    for (int j = 0; j < L; ++j) {
        if (nocrt) {
            if (q[j] == 0) { zdom = true; continue; }
            if (q[j] < 2 || q[j] >= (1ull << 31)) return fail(ALCH_E_UNSUPPORTED, "a ring without CRT basis takes moduli 2 <= q < 2^31, or 0 for the integers");
            continue;
        }
        if (q[j] >= (1ull << 31)) all32 = false;
    }
    if (zdom) for (int j = 0; j < L; ++j) if (q[j] != 0) return fail(ALCH_E_INVALID, "modulus 0 (the integers) cannot be mixed with other moduli");
    const int word = (zdom || !all32) ? 8 : 4;
    if ((size_t)gh.n * (size_t)word > 163840) return fail(ALCH_E_UNSUPPORTED, "ring dimension too large: a limb-polynomial must fit the 160 KiB LDS");
    *word_out = word;
    *zdom_out = zdom;
    return ALCH_OK;
}

static int validate_ring_args(uint32_t m, int L, const uint64_t* q, int* logn_out, int* word_out) {
    if (!two_power_engine(m)) return fail(ALCH_E_UNSUPPORTED, "internal: not a two-power index >= 32");
    const int rc = classify_crt_moduli(m, L, q);
    if (rc != ALCH_OK) return rc;
    int logn = 0;
    while ((1u << logn) < m / 2) ++logn;
    bool all32 = true;
    for (int j = 0; j < L; ++j) if (q[j] >= (1ull << 31)) all32 = false;
    const int word = all32 ? 4 : 8;
    const int maxlog = all32 ? 16 : 15;        // the top size of each word runs as two LDS-resident halves
    if (logn > maxlog) return fail(ALCH_E_UNSUPPORTED, "ring dimension too large (n <= 2^16 for 32-bit, 2^15 for 64-bit residues)");
    *logn_out = logn;
    *word_out = word;
    return ALCH_OK;
}

// ------------------------------------------------------------------------------------------------------
// limb-count selection (host only): PT2CT's type-level modulus arithmetic, restated
// ------------------------------------------------------------------------------------------------------
// Crypto/Alchemy/Interpreter/PT2CT/Noise.hs:107-170 (units of a modulus, shortest prefix with enough units) and
// Crypto/Alchemy/Interpreter/PT2CT.hs:132-140 (KSPNoise), :160-177 (mul_), :207-229 (linearCyc_), :234-249
// (CTPNoise2Units, KSPNoise2Units, Units2CTPNoise), :281-296 (the constants).
extern "C" int alch_modulus_units(uint64_t q) try {
    if (q < 2) return 0;
    return (int)std::floor(std::log2((double)q) / 6.1);          // mkModulus: floor (logBase 2 q / pNoiseUnit)
} catch (...) { return abi_catch(); }

extern "C" int alch_select_limbs(const uint64_t* moduli, int n_moduli, int op, int gadget, int p_noise_out, int* L_in,
                                 int* L_hint, int* L_out, int* p_noise_in) try {
    if (!moduli || n_moduli < 1 || p_noise_out < 0) return fail(ALCH_E_INVALID, "alch_select_limbs: bad argument");
    if (op != ALCH_OP_MUL && op != ALCH_OP_TUNNEL) return fail(ALCH_E_INVALID, "alch_select_limbs: unknown op");
    if (gadget != ALCH_GAD_TRIV && gadget != ALCH_GAD_BASE2) return fail(ALCH_E_INVALID, "unknown gadget");
    const int MinUnits = (int)std::ceil(12 / 6.1), MulPNoise = (int)std::ceil(18 / 6.1), KSAccumPNoise = (int)std::ceil(12 / 6.1),
              Max32BitUnits = (int)std::ceil(30.5 / 6.1), TunnelPNoise = (int)std::ceil(6 / 6.1);
    // prefixLen: length of the shortest nonempty prefix whose units sum to >= h; total = that prefix's units
    auto prefix = [&](int h, int* total) -> int {
        int sum = 0;
        for (int i = 0; i < n_moduli; ++i) {
            sum += alch_modulus_units(moduli[i]);
            if (sum >= h) { if (total) *total = sum; return i + 1; }
        }
        return -1;
    };
    const int p = p_noise_out;
    const int lout = prefix(p + MinUnits, nullptr);                               // PNoise2Zq zqs p
    int tot_in = 0;
    const int lin = op == ALCH_OP_MUL ? prefix(p + MulPNoise + MinUnits, &tot_in)   // PreMul_: Units2CTPNoise (TotalUnits zqs (CTPNoise2Units (p :+ MulPNoise)))
                                      : prefix(p + TunnelPNoise + MinUnits, &tot_in);
    const int ks_units = p + KSAccumPNoise + (gadget == ALCH_GAD_TRIV ? Max32BitUnits : 0);   // KSPNoise, KSPNoise2Units
    const int lh = prefix(ks_units, nullptr);
    if (lout < 0 || lin < 0 || lh < 0)
        return fail(ALCH_E_INVALID, "alch_select_limbs: the moduli do not hold enough noise units (PT2CT: \"You need more/bigger moduli!\")");
    if (L_in) *L_in = lin;
    if (L_hint) *L_hint = lh;
    if (L_out) *L_out = lout;
    if (p_noise_in) *p_noise_in = op == ALCH_OP_MUL ? tot_in - MinUnits : p + TunnelPNoise;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_host_root(uint32_t m, uint64_t q, uint64_t* psi, uint64_t* generator) try {
    if (m < 1 || q < 2) return fail(ALCH_E_INVALID, "alch_host_root: bad (m, q)");
    if (q < 3 || !h_is_prime(q)) return fail(ALCH_E_NOT_PRIME, "alch_host_root: q is not an odd prime");
    if ((q - 1) % m) return fail(ALCH_E_NO_CRT, "q is not 1 mod m");
    if (generator) *generator = h_smallest_generator(q);
    if (psi) *psi = h_root(q, m);
    return ALCH_OK;
} catch (...) { return abi_catch(); }

static int ring_create_impl(uint32_t m, int L, const uint64_t* q, bool nocrt, alch_ring** out) {
    if (!out) return fail(ALCH_E_INVALID, "alch_ring_create: null out");
    *out = nullptr;
    int logn = 0, word = 0;
    const bool gen = nocrt || !two_power_engine(m);
    GenHost gh;
    bool zdom = false;
    int rc = gen ? validate_gen_args(m, L, q, nocrt, gh, &word, &zdom) : validate_ring_args(m, L, q, &logn, &word);
    if (rc != ALCH_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(ALCH_E_NO_DEVICE, "no HIP device: this backend has no CPU fallback");
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return fail(ALCH_E_NO_DEVICE, "cannot query HIP device");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(ALCH_E_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");

    alch_ring* r = new alch_ring();
    r->m = m;
    r->n = gen ? gh.n : m / 2;
    r->logn = logn;
    r->gen = gen;
    r->has_crt = !nocrt;
    r->zdom = zdom;
    r->gh = gh;
    r->L = L;
    r->word = word;
    for (int j = 0; j < L; ++j) r->q[j] = q[j];
    u64 qmin = ~0ull, qmax = 0;
    for (int j = 0; j < L; ++j) { qmin = std::min(qmin, q[j]); qmax = std::max(qmax, q[j]); }
    r->balanced = qmax > 0 && (qmax - 1) / 2 < qmin;
    r->q30 = r->word == 4 && qmax > 0 && qmax < ((u64)1 << 30);
    r->device = dev;
    hipError_t e = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete r; return fail(ALCH_E_HIP, "hipStreamCreate failed"); }
    r->stream_owner = std::make_shared<StreamOwner>(r->stream, dev);
    if (hipEventCreate(&r->ev0) != hipSuccess || hipEventCreate(&r->ev1) != hipSuccess) { delete r; return fail(ALCH_E_HIP, "hipEventCreate failed"); }
    if (hipStreamCreateWithFlags(&r->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->ev_join, hipEventDisableTiming) != hipSuccess) { delete r; return fail(ALCH_E_HIP, "aux stream/event creation failed"); }
    if (hipMalloc((void**)&r->ws_sum, sizeof(u64)) != hipSuccess) { delete r; return fail(ALCH_E_NOMEM, "hipMalloc failed"); }
    if (hipMalloc((void**)&r->d_flag, sizeof(int)) != hipSuccess) { alch_ring_destroy(r); return fail(ALCH_E_NOMEM, "hipMalloc failed"); }
    if (gen) rc = (word == 4) ? build_gen_ring<u32>(r, r->d32, r->g32) : build_gen_ring<u64>(r, r->d64, r->g64);
    else rc = (word == 4) ? build_dev_ring<u32>(r, r->d32) : build_dev_ring<u64>(r, r->d64);
    if (rc != ALCH_OK) { alch_ring_destroy(r); return rc; }
    *out = r;
    return ALCH_OK;
}

extern "C" int alch_ring_create(uint32_t m, int L, const uint64_t* q, alch_ring** out) try { return ring_create_impl(m, L, q, false, out); } catch (...) { return abi_catch(); }
extern "C" int alch_ring_create_nocrt(uint32_t m, int L, const uint64_t* q, alch_ring** out) try { return ring_create_impl(m, L, q, true, out); } catch (...) { return abi_catch(); }

extern "C" int alch_ring_destroy(alch_ring* r) try {
    if (!r) return ALCH_OK;
    ALCH_DEVICE_LOCK(r->device);
    (void)hipSetDevice(r->device);
    if (r->scratch) { alch_buf* b = r->scratch; r->scratch = nullptr; (void)hipFree(b->dptr); delete b; }
    if (r->stream) (void)hipStreamSynchronize(r->stream);
    for (auto& e : r->pool) (void)hipFree(e.second);
    r->pool.clear();
    for (auto& pn : r->pin) { if (pn.p) (void)hipHostFree(pn.p); if (pn.ev) (void)hipEventDestroy(pn.ev); pn = alch_ring::Pin(); }
    if (r->tables) (void)hipFree(r->tables);
    if (r->gen_tables) (void)hipFree(r->gen_tables);
    if (r->d_flag) (void)hipFree(r->d_flag);
    for (auto& e : r->ext) {
        if (e.second.pow_gather) (void)hipFree(e.second.pow_gather);
        if (e.second.coeffs) (void)hipFree(e.second.coeffs);
        if (e.second.slot_small) (void)hipFree(e.second.slot_small);
        if (e.second.fibres) (void)hipFree(e.second.fibres);
    }
    if (r->tables_p) (void)hipFree(r->tables_p);
    if (r->ws_digits) (void)hipFree(r->ws_digits);
    if (r->ws_in) (void)hipFree(r->ws_in);
    if (r->ws_full) (void)hipFree(r->ws_full);
    if (r->ev_x) (void)hipEventDestroy(r->ev_x);
    if (r->ws_host) (void)hipFree(r->ws_host);
    if (r->ws_sum) (void)hipFree(r->ws_sum);
    if (r->ev0) (void)hipEventDestroy(r->ev0);
    if (r->ev1) (void)hipEventDestroy(r->ev1);
    if (r->ev_fork) (void)hipEventDestroy(r->ev_fork);
    for (int e = 0; e < 2; ++e) { if (r->ev_pa[e]) (void)hipEventDestroy(r->ev_pa[e]); if (r->ev_pb[e]) (void)hipEventDestroy(r->ev_pb[e]); }
    if (r->ev_join) (void)hipEventDestroy(r->ev_join);
    if (r->aux) { (void)hipStreamSynchronize(r->aux); (void)hipStreamDestroy(r->aux); }
    for (int e = 0; e < 2; ++e) {
        if (r->xs[e]) { (void)hipStreamSynchronize(r->xs[e]); (void)hipStreamDestroy(r->xs[e]); }
        if (r->ev_xs[e]) (void)hipEventDestroy(r->ev_xs[e]);
    }
    r->stream_owner.reset();                       // destroys the stream with its last user
    delete r;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_ring_n(const alch_ring* r, uint32_t* n, int* L, int* word_bytes) try {
    if (!r) return fail(ALCH_E_INVALID, "null ring");
    if (n) *n = r->n;
    if (L) *L = r->L;
    if (word_bytes) *word_bytes = r->word;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_ring_set_stream(alch_ring* r, void* s) try {
    if (!r) return fail(ALCH_E_INVALID, "null ring");
    BIND(r);
    HIP_TRY(hipStreamSynchronize(r->stream));
    r->stream_owner.reset();
    r->stream = (hipStream_t)s;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_ring_set_option(alch_ring* r, const char* name, long value) try {
    if (!r || !name) return fail(ALCH_E_INVALID, "null argument");
    const std::string k(name);
    if (k == "chunk") { if (value < 8) return fail(ALCH_E_INVALID, "chunk must be >= 8"); r->chunk = (size_t)value; }
    else if (k == "one_stream") r->one_stream = value != 0;
    else if (k == "pipe") r->pipe = value != 0;
    else if (k == "ks_map") r->opts.ks_map = value != 0;
    else if (k == "ks_rev") r->opts.ks_rev = value != 0;
    else if (k == "nstreams") { if (value < 1 || value > 4) return fail(ALCH_E_INVALID, "nstreams: 1 .. 4"); r->nstreams = (int)value; }
    else if (k == "q30") r->opts.q30 = value != 0;
    else if (k == "ks_grid") { if (value < 1) return fail(ALCH_E_INVALID, "ks_grid must be >= 1"); r->opts.ks_grid = (unsigned)value; }
    else if (k == "ti_grid") r->opts.ti_grid = (int)value;
    else if (k == "ti_split") { if (value < 0) return fail(ALCH_E_INVALID, "ti_split must be >= 0"); r->opts.ti_split = (int)value; }
    else if (k == "rs_lin") r->opts.rs_lin = value != 0;
    else if (k == "crt_half") r->opts.crt_half = value != 0;
    else if (k == "rs_half") r->opts.rs_half = value != 0;
    else if (k == "tunnel_ep") r->opts.tunnel_ep = value != 0;
    else if (k == "gen_fused") r->opts.gen_fused = value != 0;
    else if (k == "tunnel_mac") r->opts.tunnel_mac = value != 0;
    else if (k == "tunnel_fused") { if (value < 0 || (value != 0 && value != 2 && value != 4)) return fail(ALCH_E_INVALID, "tunnel_fused: 0 (composed), 2 or 4 (digits side by side)"); r->opts.tunnel_fused = (int)value; }
    else if (k == "gen_nt") {
        if (value != 0 && value != 128 && value != 256 && value != 512) return fail(ALCH_E_INVALID, "gen_nt must be 0 (by ring size), 128, 256 or 512");
        r->g32.nt = (int)value; r->g64.nt = (int)value;
    }
    else if (k == "split_fused") { if (value < 0 || value > 2) return fail(ALCH_E_INVALID, "split_fused: 0, 1 or 2"); r->opts.split_fused = (int)value; }
    else if (k == "scratch_mib") { if (value < 1 || value > 65536) return fail(ALCH_E_INVALID, "scratch_mib must be 1 .. 65536"); r->scratch_mib = (size_t)value; }
    else if (k == "rs_slots") { if (value < 1) return fail(ALCH_E_INVALID, "rs_slots must be >= 1"); r->rs_slots = (unsigned)value; }
    else if (k == "stream_dedicated") {
        // A new stream with an (all-ones) CU mask replaces the ring's: the HIP runtime gives such a stream a hardware queue of its own.
        // Ordinary streams share the runtime's pool of hardware queues (4 by default) by a rule that depends on how many streams the
        // process holds, so two "independent" streams land on ONE queue every other time and their kernels run one after the other
        // (measured with tools/queue_probe.py: the two sub-batches of the HomomRLWR pipeline at 44.7 k instead of 50.5 k pipelines/s
        // for every odd number of other rings alive; stream priorities do not separate them reliably either: 46 k every other time).
        // For hosts that run independent sub-batches side by side (alchemy_amd/ringround.py); every dedicated stream costs a
        // hardware queue, so it is an option, not the default of the hundreds of rings a Lol host creates.
        if (value != 1) return fail(ALCH_E_INVALID, "stream_dedicated: 1");
        BIND(r);
        if (++g_dedicated_streams > MAX_DEDICATED_STREAMS) {
            --g_dedicated_streams;
            return fail(ALCH_E_UNSUPPORTED, "stream_dedicated: " + std::to_string(MAX_DEDICATED_STREAMS) + " dedicated streams are alive in this process already "
                                            "(each holds a hardware queue; the ring keeps its ordinary stream)");
        }
        struct Undo { bool armed = true; ~Undo() { if (armed) --g_dedicated_streams; } } undo;       // any failure below gives the slot back
        HIP_TRY(hipStreamSynchronize(r->stream));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, r->device));
        std::vector<uint32_t> mask((size_t)(prop.multiProcessorCount + 31) / 32, 0xffffffffu);
        if (prop.multiProcessorCount % 32) mask.back() = (1u << (prop.multiProcessorCount % 32)) - 1u;
        hipStream_t ns = nullptr;
        HIP_TRY(hipExtStreamCreateWithCUMask(&ns, (uint32_t)mask.size(), mask.data()));
        r->stream_owner = std::make_shared<StreamOwner>(ns, r->device, true);   // the old stream goes with its last user
        undo.armed = false;                                                      // the owner returns the slot when the stream dies
        r->stream = ns;
    }
    else return fail(ALCH_E_INVALID, "unknown option '" + k + "'");
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_sync(alch_ring* r) try {
    if (!r) return fail(ALCH_E_INVALID, "null ring");
    BIND(r);
    HIP_TRY(hipStreamSynchronize(r->stream));
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_timer_start(alch_ring* r) try {
    if (!r) return fail(ALCH_E_INVALID, "null ring");
    BIND(r);
    HIP_TRY(hipEventRecord(r->ev0, r->stream));
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_timer_stop(alch_ring* r, float* ms) try {
    if (!r || !ms) return fail(ALCH_E_INVALID, "null argument");
    BIND(r);
    HIP_TRY(hipEventRecord(r->ev1, r->stream));
    HIP_TRY(hipEventSynchronize(r->ev1));
    HIP_TRY(hipEventElapsedTime(ms, r->ev0, r->ev1));
    return ALCH_OK;
} catch (...) { return abi_catch(); }

// ------------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------------
static inline bool split_ring_fwd(const alch_ring* r) { return !r->gen && r->logn > (r->word == 4 ? 15 : 14); }   // == split_ring()
static inline size_t elem_words(const alch_ring* r) { return (size_t)r->L * r->n; }
static inline size_t elem_bytes(const alch_ring* r) { return elem_words(r) * (size_t)r->word; }

static int ensure_ws(void** p, size_t* have, size_t need) {
    if (*have >= need) return ALCH_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *have = 0; }
    if (hipMalloc(p, need) != hipSuccess) return fail(ALCH_E_NOMEM, "hipMalloc(" + std::to_string(need) + ") failed");
    *have = need;
    return ALCH_OK;
}

template <typename W> static const DevRing<W>& dev_ring(const alch_ring* r);
template <> const DevRing<u32>& dev_ring<u32>(const alch_ring* r) { return r->d32; }
template <> const DevRing<u64>& dev_ring<u64>(const alch_ring* r) { return r->d64; }

static hipError_t dispatch(int logn, const NttCall<u32>& c) {
    if (logn <= 9) return dispatch32_small(logn, c);
    if (logn <= 13) return dispatch32_mid(logn, c);
    if (logn == 14) return dispatch32_14(logn, c);
    if (logn == 15) return dispatch32_15(logn, c);
    return dispatch32_16(logn, c);
}
static hipError_t dispatch(int logn, const NttCall<u64>& c) {
    if (logn <= 11) return dispatch64_small(logn, c);
    if (logn <= 14) return dispatch64_big(logn, c);
    return dispatch64_15(logn, c);
}

template <typename W>
static int do_crt(alch_ring* r, void* data, size_t first_elem, size_t count, bool inverse, const void* src = nullptr,
                  hipStream_t stream = nullptr) {
    // src != null: transform src[first_elem ..) into data[first_elem ..) (LDS-resident sizes only)
    if (count == 0) return ALCH_OK;
    if (!r->has_crt) return fail(ALCH_E_NO_CRT, "this ring has no CRT basis (created with alch_ring_create_nocrt)");
    if (r->gen) {
        GenCall<W> g{};
        g.op = inverse ? GEN_CRTINV : GEN_CRT;
        g.ring = &dev_ring<W>(r);
        g.gen = &gen_dev<W>(r);
        g.stream = stream ? stream : r->stream;
        g.data = reinterpret_cast<W*>(data);
        g.src = reinterpret_cast<const W*>(src);
        const size_t polys = count * (size_t)r->L;
        for (size_t done = 0; done < polys;) {
            const size_t now = std::min<size_t>(polys - done, (size_t)1 << 30);
            g.first_poly = first_elem * (size_t)r->L + done;
            g.npoly = now;
            hipError_t e = gen_dispatch(g);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("general-index crt launch: ") + hipGetErrorString(e));
            done += now;
        }
        return ALCH_OK;
    }
    NttCall<W> c{};
    c.src = reinterpret_cast<const W*>(src);
    c.op = inverse ? OP_CRTINV : OP_CRT;
    c.ring = &dev_ring<W>(r);
    c.stream = stream ? stream : r->stream;     // another ring's stream when the call is part of that ring's pipeline
    c.data = reinterpret_cast<W*>(data);
    const size_t polys = count * (size_t)r->L;
    // grids are 32-bit: split very large batches
    size_t done = 0;
    while (done < polys) {
        size_t now = std::min<size_t>(polys - done, (size_t)1 << 30);
        c.first_poly = first_elem * (size_t)r->L + done;
        c.npoly = now;
        hipError_t e = dispatch(r->logn, c);
        if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("crt launch: ") + hipGetErrorString(e));
        done += now;
    }
    return ALCH_OK;
}

static int buf_crt(alch_buf* b, size_t first, size_t count, bool inverse) {
    if (!b) return fail(ALCH_E_INVALID, "null buffer");
    if (!range_ok(first, count, b->n_elems)) return fail(ALCH_E_INVALID, "element range out of bounds");
    alch_ring* r = b->ring;
    BIND(r);
    return r->word == 4 ? do_crt<u32>(r, b->dptr, first, count, inverse) : do_crt<u64>(r, b->dptr, first, count, inverse);
}

// ------------------------------------------------------------------------------------------------------
// device buffers
// ------------------------------------------------------------------------------------------------------
extern "C" int alch_buf_alloc(alch_ring* r, size_t n_elems, alch_buf** out) try {
    if (!r || !out || n_elems == 0) return fail(ALCH_E_INVALID, "alch_buf_alloc: bad argument");
    if (n_elems > (size_t)-1 / elem_bytes(r)) return fail(ALCH_E_INVALID, "alch_buf_alloc: n_elems * element size overflows size_t");
    BIND(r);
    const size_t bytes = n_elems * elem_bytes(r);
    if (bytes <= POOL_MAX_BUF) {
        std::lock_guard<std::mutex> lk(r->pool_mu);
        for (size_t i = r->pool.size(); i-- > 0;)
            if (r->pool[i].first == n_elems) {                     // stream-ordered reuse (see alch_ring::pool)
                void* p = r->pool[i].second;
                r->pool[i] = r->pool.back();
                r->pool.pop_back();
                r->pool_bytes -= bytes;
                *out = new alch_buf{r, n_elems, p};
                return ALCH_OK;
            }
    }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        // parked buffers may be what exhausts the device: release them and try once more
        std::vector<std::pair<size_t, void*>> parked;
        { std::lock_guard<std::mutex> lk(r->pool_mu); parked.swap(r->pool); r->pool_bytes = 0; }
        if (!parked.empty()) {
            (void)hipStreamSynchronize(r->stream);
            for (auto& e : parked) (void)hipFree(e.second);
            if (hipMalloc(&p, bytes) != hipSuccess) p = nullptr;
        }
        if (!p) return fail(ALCH_E_NOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes failed");
    }
    *out = new alch_buf{r, n_elems, p};
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_view(const alch_buf* parent, size_t first, size_t count, alch_buf** out) try {
    if (!parent || !out || count == 0) return fail(ALCH_E_INVALID, "alch_buf_view: bad argument");
    if (!range_ok(first, count, parent->n_elems)) return fail(ALCH_E_INVALID, "element range out of bounds");
    alch_buf* v = new alch_buf{parent->ring, count, reinterpret_cast<char*>(parent->dptr) + first * elem_bytes(parent->ring)};
    v->view = true;
    *out = v;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_free(alch_buf* b) try {
    if (!b) return ALCH_OK;
    if (b->view) { delete b; return ALCH_OK; }                      // an alias owns nothing
    alch_ring* r = b->ring;
    const size_t bytes = b->n_elems * elem_bytes(r);
    if (bytes <= POOL_MAX_BUF) {
        std::lock_guard<std::mutex> lk(r->pool_mu);
        if (r->pool_bytes + bytes <= POOL_CAP) {
            r->pool.emplace_back(b->n_elems, b->dptr);             // no synchronisation: reuse is ordered on the ring's stream
            r->pool_bytes += bytes;
            delete b;
            return ALCH_OK;
        }
    }
    (void)hipSetDevice(r->device);
    (void)hipStreamSynchronize(r->stream);
    (void)hipFree(b->dptr);
    delete b;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_elems(const alch_buf* b, size_t* n) try {
    if (!b || !n) return fail(ALCH_E_INVALID, "null argument");
    *n = b->n_elems;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_device_ptr(const alch_buf* b, void** ptr, size_t* bytes) try {
    if (!b || !ptr) return fail(ALCH_E_INVALID, "null argument");
    *ptr = b->dptr;
    if (bytes) *bytes = b->n_elems * elem_bytes(b->ring);
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_ring(const alch_buf* b, alch_ring** ring) try {
    if (!b || !ring) return fail(ALCH_E_INVALID, "null argument");
    *ring = b->ring;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_ring_device(const alch_ring* r, int* device, void** hip_stream) try {
    if (!r) return fail(ALCH_E_INVALID, "null ring");
    if (device) *device = r->device;
    if (hip_stream) *hip_stream = (void*)r->stream;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

template <typename W>
static int do_transfer(alch_ring* r, void* dev, size_t count, int64_t* host, bool to_device) {
    const size_t bytes = count * elem_words(r) * sizeof(int64_t);
    int rc = ensure_ws(&r->ws_host, &r->ws_host_bytes, bytes);
    if (rc != ALCH_OK) return rc;
    int64_t* stage = reinterpret_cast<int64_t*>(r->ws_host);
    const size_t total = count * elem_words(r);
    if (bytes <= PIN_MAX) {
        // small transfer (the per-Tensor-call path moves one ring element at a time): through a pinned slot
        alch_ring::Pin& pn = r->pin[r->pin_next];
        r->pin_next = (r->pin_next + 1) % 4;
        if (pn.busy) { HIP_TRY(hipEventSynchronize(pn.ev)); pn.busy = false; }     // the slot's last upload has left it (long ago, normally)
        if (pn.bytes < bytes) {
            if (pn.p) { (void)hipHostFree(pn.p); pn.p = nullptr; pn.bytes = 0; }
            if (hipHostMalloc(&pn.p, bytes, hipHostMallocDefault) != hipSuccess) return fail(ALCH_E_NOMEM, "hipHostMalloc(" + std::to_string(bytes) + ") failed");
            pn.bytes = bytes;
        }
        if (!pn.ev) HIP_TRY(hipEventCreateWithFlags(&pn.ev, hipEventDisableTiming));
        if (to_device) {
            memcpy(pn.p, host, bytes);                             // the caller's buffer is consumed here: no synchronisation needed
            HIP_TRY(hipMemcpyAsync(stage, pn.p, bytes, hipMemcpyHostToDevice, r->stream));
            hipLaunchKernelGGL((k_transpose<W, true>), dim3(ew_grid(total)), dim3(256), 0, r->stream, dev_ring<W>(r),
                               reinterpret_cast<W*>(dev), stage, count);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(pn.ev, r->stream));
            pn.busy = true;
        } else {
            hipLaunchKernelGGL((k_transpose<W, false>), dim3(ew_grid(total)), dim3(256), 0, r->stream, dev_ring<W>(r),
                               reinterpret_cast<W*>(dev), stage, count);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(pn.p, stage, bytes, hipMemcpyDeviceToHost, r->stream));
            HIP_TRY(hipStreamSynchronize(r->stream));              // the one synchronisation of a download
            memcpy(host, pn.p, bytes);
        }
        return ALCH_OK;
    }
    if (to_device) {
        HIP_TRY(hipMemcpyAsync(stage, host, bytes, hipMemcpyHostToDevice, r->stream));
        hipLaunchKernelGGL((k_transpose<W, true>), dim3(ew_grid(total)), dim3(256), 0, r->stream, dev_ring<W>(r),
                           reinterpret_cast<W*>(dev), stage, count);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(r->stream));      // host buffer is caller-owned: do not retain it
    } else {
        hipLaunchKernelGGL((k_transpose<W, false>), dim3(ew_grid(total)), dim3(256), 0, r->stream, dev_ring<W>(r),
                           reinterpret_cast<W*>(dev), stage, count);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(host, stage, bytes, hipMemcpyDeviceToHost, r->stream));
        HIP_TRY(hipStreamSynchronize(r->stream));
    }
    return ALCH_OK;
}

static int transfer(alch_ring* r, void* dev_base, size_t first, size_t count, int64_t* host, bool to_device) {
    BIND(r);
    // bounded staging: move at most 64 MiB of int64 at a time
    const size_t per = std::max<size_t>(1, ((size_t)64 << 20) / (elem_words(r) * sizeof(int64_t)));
    size_t done = 0;
    while (done < count) {
        const size_t now = std::min(per, count - done);
        char* dev = reinterpret_cast<char*>(dev_base) + (first + done) * elem_bytes(r);
        int64_t* h = host + done * elem_words(r);
        int rc = r->word == 4 ? do_transfer<u32>(r, dev, now, h, to_device) : do_transfer<u64>(r, dev, now, h, to_device);
        if (rc != ALCH_OK) return rc;
        done += now;
    }
    return ALCH_OK;
}

extern "C" int alch_buf_upload(alch_buf* b, size_t first, size_t count, const int64_t* host) try {
    if (!b || !host) return fail(ALCH_E_INVALID, "null argument");
    if (!range_ok(first, count, b->n_elems)) return fail(ALCH_E_INVALID, "element range out of bounds");
    return transfer(b->ring, b->dptr, first, count, const_cast<int64_t*>(host), true);
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_download(const alch_buf* b, size_t first, size_t count, int64_t* host) try {
    if (!b || !host) return fail(ALCH_E_INVALID, "null argument");
    if (!range_ok(first, count, b->n_elems)) return fail(ALCH_E_INVALID, "element range out of bounds");
    return transfer(b->ring, b->dptr, first, count, host, false);
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_fill_uniform(alch_buf* b, uint64_t seed) try {
    if (!b) return fail(ALCH_E_INVALID, "null buffer");
    alch_ring* r = b->ring;
    BIND(r);
    const size_t words = b->n_elems * elem_words(r);
    if (r->word == 4)
        hipLaunchKernelGGL((k_fill_uniform<u32>), dim3(ew_grid(words)), dim3(256), 0, r->stream, r->d32, (u32*)b->dptr, words, seed);
    else
        hipLaunchKernelGGL((k_fill_uniform<u64>), dim3(ew_grid(words)), dim3(256), 0, r->stream, r->d64, (u64*)b->dptr, words, seed);
    HIP_TRY(hipGetLastError());
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_crt(alch_buf* b, size_t first, size_t count) try { return buf_crt(b, first, count, false); } catch (...) { return abi_catch(); }
extern "C" int alch_buf_crtinv(alch_buf* b, size_t first, size_t count) try { return buf_crt(b, first, count, true); } catch (...) { return abi_catch(); }

template <typename W, int OP>
static int do_pointwise(alch_ring* r, void* dst, const void* a, const void* b, size_t count) {
    const size_t words = count * elem_words(r);
    if (r->n % Vec4<W>::LANES)
        hipLaunchKernelGGL((k_pointwise_scalar<W, OP>), dim3(ew_grid(words)), dim3(256), 0, r->stream, dev_ring<W>(r),
                           (W*)dst, (const W*)a, (const W*)b, words);
    else
    hipLaunchKernelGGL((k_pointwise<W, OP>), dim3(ew_grid(words / Vec4<W>::LANES)), dim3(256), 0, r->stream, dev_ring<W>(r),
                       (W*)dst, (const W*)a, (const W*)b, words);
    HIP_TRY(hipGetLastError());
    return ALCH_OK;
}

static int buf_pointwise(alch_buf* dst, const alch_buf* a, const alch_buf* b, size_t count, int op) {
    if (!dst || !a || !b) return fail(ALCH_E_INVALID, "null buffer");
    if (a->ring != dst->ring || b->ring != dst->ring) return fail(ALCH_E_INVALID, "buffers belong to different rings");
    if (count > dst->n_elems || count > a->n_elems || count > b->n_elems) return fail(ALCH_E_INVALID, "count out of bounds");
    alch_ring* r = dst->ring;
    BIND(r);
    if (!r->has_crt && op == PW_MUL) return fail(ALCH_E_NO_CRT, "pointwise products need the CRT basis; this ring has none");
    if (r->word == 4) {
        if (op == PW_MUL) return do_pointwise<u32, PW_MUL>(r, dst->dptr, a->dptr, b->dptr, count);
        if (op == PW_ADD) return do_pointwise<u32, PW_ADD>(r, dst->dptr, a->dptr, b->dptr, count);
        return do_pointwise<u32, PW_SUB>(r, dst->dptr, a->dptr, b->dptr, count);
    }
    if (op == PW_MUL) return do_pointwise<u64, PW_MUL>(r, dst->dptr, a->dptr, b->dptr, count);
    if (op == PW_ADD) return do_pointwise<u64, PW_ADD>(r, dst->dptr, a->dptr, b->dptr, count);
    return do_pointwise<u64, PW_SUB>(r, dst->dptr, a->dptr, b->dptr, count);
}

extern "C" int alch_buf_mul(alch_buf* d, const alch_buf* a, const alch_buf* b, size_t count) try { return buf_pointwise(d, a, b, count, PW_MUL); } catch (...) { return abi_catch(); }
extern "C" int alch_buf_add(alch_buf* d, const alch_buf* a, const alch_buf* b, size_t count) try { return buf_pointwise(d, a, b, count, PW_ADD); } catch (...) { return abi_catch(); }
extern "C" int alch_buf_sub(alch_buf* d, const alch_buf* a, const alch_buf* b, size_t count) try { return buf_pointwise(d, a, b, count, PW_SUB); } catch (...) { return abi_catch(); }

static int buf_checksum(const alch_buf* b, size_t first, size_t count, uint64_t position, uint64_t* sum) {
    if (!b || !sum) return fail(ALCH_E_INVALID, "null argument");
    if (!range_ok(first, count, b->n_elems)) return fail(ALCH_E_INVALID, "element range out of bounds");
    alch_ring* r = b->ring;
    BIND(r);
    const size_t words = count * elem_words(r);
    const char* base = reinterpret_cast<const char*>(b->dptr) + first * elem_bytes(r);
    HIP_TRY(hipMemsetAsync(r->ws_sum, 0, sizeof(u64), r->stream));
    if (r->word == 4) hipLaunchKernelGGL((k_checksum<u32>), dim3(ew_grid(words)), dim3(256), 0, r->stream, (const u32*)base, words, r->ws_sum, (u64)position * elem_words(r));
    else hipLaunchKernelGGL((k_checksum<u64>), dim3(ew_grid(words)), dim3(256), 0, r->stream, (const u64*)base, words, r->ws_sum, (u64)position * elem_words(r));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(sum, r->ws_sum, sizeof(u64), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return ALCH_OK;
}

extern "C" int alch_buf_checksum(const alch_buf* b, size_t first, size_t count, uint64_t* sum) try { return buf_checksum(b, first, count, 0, sum); } catch (...) { return abi_catch(); }
extern "C" int alch_buf_checksum_at(const alch_buf* b, size_t first, size_t count, uint64_t position, uint64_t* sum) try { return buf_checksum(b, first, count, position, sum); } catch (...) { return abi_catch(); }

// ------------------------------------------------------------------------------------------------------
// host-buffer Tensor methods: stage one element through a scratch device buffer
// ------------------------------------------------------------------------------------------------------
// The host-buffer Tensor methods stage their operands through a per-ring scratch array that is kept between
// calls (no hipMalloc / hipFree per ring element); it only ever grows.
struct ScratchBuf {
    alch_buf* b = nullptr;
};
static int scratch_get(alch_ring* r, size_t n_elems, alch_buf** out) {
    BIND(r);
    if (r->scratch && r->scratch->n_elems >= n_elems) { *out = r->scratch; return ALCH_OK; }
    if (r->scratch) { alch_buf* old = r->scratch; r->scratch = nullptr; alch_buf_free(old); }
    int rc = alch_buf_alloc(r, n_elems, &r->scratch);
    if (rc != ALCH_OK) return rc;
    *out = r->scratch;
    return ALCH_OK;
}

static int host_unary(alch_ring* r, int64_t* data, int which) {
    if (!r || !data) return fail(ALCH_E_INVALID, "null argument");
    ScratchBuf s;
    int rc = scratch_get(r, 1, &s.b);
    if (rc != ALCH_OK) return rc;
    if ((rc = alch_buf_upload(s.b, 0, 1, data)) != ALCH_OK) return rc;
    if ((rc = buf_crt(s.b, 0, 1, which == 1)) != ALCH_OK) return rc;
    return alch_buf_download(s.b, 0, 1, data);
}

extern "C" int alch_crt(alch_ring* r, int64_t* data) try { return host_unary(r, data, 0); } catch (...) { return abi_catch(); }
extern "C" int alch_crtinv(alch_ring* r, int64_t* data) try { return host_unary(r, data, 1); } catch (...) { return abi_catch(); }

static int host_binary(alch_ring* r, int64_t* a, const int64_t* b, int op) {
    if (!r || !a || !b) return fail(ALCH_E_INVALID, "null argument");
    ScratchBuf s;
    int rc = scratch_get(r, 2, &s.b);
    if (rc != ALCH_OK) return rc;
    if ((rc = alch_buf_upload(s.b, 0, 1, a)) != ALCH_OK) return rc;
    if ((rc = alch_buf_upload(s.b, 1, 1, b)) != ALCH_OK) return rc;
    alch_buf second{r, 1, reinterpret_cast<char*>(s.b->dptr) + elem_bytes(r)};
    if ((rc = buf_pointwise(s.b, s.b, &second, 1, op)) != ALCH_OK) return rc;
    return alch_buf_download(s.b, 0, 1, a);
}

extern "C" int alch_mul(alch_ring* r, int64_t* a, const int64_t* b) try { return host_binary(r, a, b, PW_MUL); } catch (...) { return abi_catch(); }
extern "C" int alch_add(alch_ring* r, int64_t* a, const int64_t* b) try { return host_binary(r, a, b, PW_ADD); } catch (...) { return abi_catch(); }
extern "C" int alch_sub(alch_ring* r, int64_t* a, const int64_t* b) try { return host_binary(r, a, b, PW_SUB); } catch (...) { return abi_catch(); }

template <typename W>
static int scal_to_mont(const alch_ring* r, const uint64_t* s, int power_of_R, Scal<W>& out) {
    // out[j] = s[j] * R^power mod q_j  (s == NULL means 1)
    const int bits = 8 * (int)sizeof(W);
    for (int j = 0; j < MAXL; ++j) out.v[j] = 0;
    for (int j = 0; j < r->L; ++j) {
        const u64 q = r->q[j];
        u64 v = s ? s[j] % q : 1;
        const u64 r1 = h_powmod(2, (u64)bits, q);
        for (int p = 0; p < power_of_R; ++p) v = h_mulmod(v, r1, q);
        out.v[j] = (W)v;
    }
    return ALCH_OK;
}

template <typename W>
static int do_scale(alch_ring* r, void* dst, const void* src, size_t count, const uint64_t* s) {
    if (!r->has_crt) return fail(ALCH_E_UNSUPPORTED, "scalar products are implemented for rings with Montgomery constants (prime moduli) only");
    Scal<W> sm;
    scal_to_mont<W>(r, s, 1, sm);
    const size_t words = count * elem_words(r);
    ALCH_LAUNCH_VW(k_scale, r, words, r->stream, dev_ring<W>(r), (W*)dst, (const W*)src, words, sm);
    HIP_TRY(hipGetLastError());
    return ALCH_OK;
}

extern "C" int alch_scale(alch_ring* r, int64_t* a, const uint64_t* s) try {
    if (!r || !a || !s) return fail(ALCH_E_INVALID, "null argument");
    ScratchBuf t;
    int rc = scratch_get(r, 1, &t.b);
    if (rc != ALCH_OK) return rc;
    if ((rc = alch_buf_upload(t.b, 0, 1, a)) != ALCH_OK) return rc;
    rc = r->word == 4 ? do_scale<u32>(r, t.b->dptr, t.b->dptr, 1, s) : do_scale<u64>(r, t.b->dptr, t.b->dptr, 1, s);
    if (rc != ALCH_OK) return rc;
    return alch_buf_download(t.b, 0, 1, a);
} catch (...) { return abi_catch(); }

// mulG / divG / l / lInv on one host ring element: staged through the per-ring scratch element.
static int buf_mulg_divg(alch_buf* b, size_t first, size_t count, int basis, bool divide);
static int buf_l(alch_buf* b, size_t first, size_t count, bool inverse);
static int host_g(alch_ring* r, int64_t* a, int basis, int which /* 0 mulG, 1 divG, 2 l, 3 lInv */) {
    if (!r || !a) return fail(ALCH_E_INVALID, "null argument");
    if (basis == ALCH_BASIS_CRT && !r->has_crt) return fail(ALCH_E_NO_CRT, "this ring has no CRT basis");
    if (!r->gen || r->gh.rad == 1) return ALCH_OK;                    // two-power index: g = 1, L = identity
    ScratchBuf s;
    int rc = scratch_get(r, 1, &s.b);
    if (rc != ALCH_OK) return rc;
    if ((rc = alch_buf_upload(s.b, 0, 1, a)) != ALCH_OK) return rc;
    if (which == 0) rc = buf_mulg_divg(s.b, 0, 1, basis, false);
    else if (which == 1) rc = buf_mulg_divg(s.b, 0, 1, basis, true);
    else rc = buf_l(s.b, 0, 1, which == 3);
    if (rc != ALCH_OK) return rc;                                     // includes ALCH_NOT_DIVISIBLE: the host data stay untouched
    return alch_buf_download(s.b, 0, 1, a);
}
extern "C" int alch_mulg_pow(alch_ring* r, int64_t* a) try { return host_g(r, a, ALCH_BASIS_POW, 0); } catch (...) { return abi_catch(); }
extern "C" int alch_mulg_dec(alch_ring* r, int64_t* a) try { return host_g(r, a, ALCH_BASIS_DEC, 0); } catch (...) { return abi_catch(); }
extern "C" int alch_mulg_crt(alch_ring* r, int64_t* a) try { return host_g(r, a, ALCH_BASIS_CRT, 0); } catch (...) { return abi_catch(); }
extern "C" int alch_divg_pow(alch_ring* r, int64_t* a) try { return host_g(r, a, ALCH_BASIS_POW, 1); } catch (...) { return abi_catch(); }
extern "C" int alch_divg_dec(alch_ring* r, int64_t* a) try { return host_g(r, a, ALCH_BASIS_DEC, 1); } catch (...) { return abi_catch(); }
extern "C" int alch_divg_crt(alch_ring* r, int64_t* a) try { return host_g(r, a, ALCH_BASIS_CRT, 1); } catch (...) { return abi_catch(); }
extern "C" int alch_l(alch_ring* r, int64_t* a) try { return host_g(r, a, ALCH_BASIS_POW, 2); } catch (...) { return abi_catch(); }
extern "C" int alch_linv(alch_ring* r, int64_t* a) try { return host_g(r, a, ALCH_BASIS_POW, 3); } catch (...) { return abi_catch(); }

extern "C" int alch_decompose_triv(alch_ring* r, const int64_t* c_pow, int64_t* digits) try {
    if (!r || !c_pow || !digits) return fail(ALCH_E_INVALID, "null argument");
    ScratchBuf s;
    int rc = scratch_get(r, 1 + (size_t)r->L, &s.b);
    if (rc != ALCH_OK) return rc;
    if ((rc = alch_buf_upload(s.b, 0, 1, c_pow)) != ALCH_OK) return rc;
    char* dig = reinterpret_cast<char*>(s.b->dptr) + elem_bytes(r);
    const size_t total = (size_t)r->L * elem_words(r);
    if (r->word == 4) hipLaunchKernelGGL((k_decompose_triv<u32>), dim3(ew_grid(total)), dim3(256), 0, r->stream, r->d32, (const u32*)s.b->dptr, (u32*)dig, r->balanced ? 1 : 0);
    else hipLaunchKernelGGL((k_decompose_triv<u64>), dim3(ew_grid(total)), dim3(256), 0, r->stream, r->d64, (const u64*)s.b->dptr, (u64*)dig, r->balanced ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return alch_buf_download(s.b, 1, (size_t)r->L, digits);
} catch (...) { return abi_catch(); }

// BaseBGad 2 layout: limb i owns ceil(log2 q_i) digits starting at first[i]; returns their total.
static int base2_layout(const alch_ring* r, Scal<u32>& first, Scal<u32>& kd) {
    int D = 0;
    for (int j = 0; j < MAXL; ++j) { first.v[j] = 0; kd.v[j] = 0; }
    for (int i = 0; i < r->L; ++i) {
        int k = 0;
        for (u64 v = 1; v < r->q[i]; v <<= 1) ++k;
        first.v[i] = (u32)D;
        kd.v[i] = (u32)k;
        D += k;
    }
    return D;
}

static int gadget_digits(const alch_ring* r, int gadget) {
    if (gadget == ALCH_GAD_TRIV) return r->L;
    Scal<u32> f, k;
    return base2_layout(r, f, k);
}

extern "C" int alch_decompose_base2(alch_ring* r, const int64_t* c_pow, int64_t* digits, int* n_digits) try {
    if (!r) return fail(ALCH_E_INVALID, "null ring");
    Scal<u32> first, kd;
    const int D = base2_layout(r, first, kd);
    if (n_digits) *n_digits = D;
    if (!digits) return ALCH_OK;
    if (!c_pow) return fail(ALCH_E_INVALID, "null argument");
    ScratchBuf s;
    int rc = scratch_get(r, 1 + (size_t)D, &s.b);
    if (rc != ALCH_OK) return rc;
    if ((rc = alch_buf_upload(s.b, 0, 1, c_pow)) != ALCH_OK) return rc;
    char* dig = reinterpret_cast<char*>(s.b->dptr) + elem_bytes(r);
    const size_t total = elem_words(r);
    if (r->word == 4) hipLaunchKernelGGL((k_decompose_base2<u32>), dim3(ew_grid(total)), dim3(256), 0, r->stream, r->d32, (const u32*)s.b->dptr, (u32*)dig, first, kd, (u32)D);
    else hipLaunchKernelGGL((k_decompose_base2<u64>), dim3(ew_grid(total)), dim3(256), 0, r->stream, r->d64, (const u64*)s.b->dptr, (u64*)dig, first, kd, (u32)D);
    HIP_TRY(hipGetLastError());
    return alch_buf_download(s.b, 1, (size_t)D, digits);
} catch (...) { return abi_catch(); }

// ------------------------------------------------------------------------------------------------------
// hint
// ------------------------------------------------------------------------------------------------------
static int hint_from_device(alch_ring* r, int gadget, const void* src_crt, alch_hint** out) {
    BIND(r);
    if (gadget != ALCH_GAD_TRIV && gadget != ALCH_GAD_BASE2) return fail(ALCH_E_INVALID, "unknown gadget");
    const int digits = gadget_digits(r, gadget);
    const size_t elems = 2 * (size_t)digits;
    void* p = nullptr;
    if (hipMalloc(&p, elems * elem_bytes(r)) != hipSuccess) return fail(ALCH_E_NOMEM, "hipMalloc(hint) failed");
    // Montgomery form: multiply by R^2 under mont_mul  -> x * R
    int rc;
    if (r->word == 4) {
        Scal<u32> sm; scal_to_mont<u32>(r, nullptr, 2, sm);
        const size_t words = elems * elem_words(r);
        hipLaunchKernelGGL((k_scale<u32>), dim3(ew_grid(words)), dim3(256), 0, r->stream, r->d32, (u32*)p, (const u32*)src_crt, words, sm);
    } else {
        Scal<u64> sm; scal_to_mont<u64>(r, nullptr, 2, sm);
        const size_t words = elems * elem_words(r);
        hipLaunchKernelGGL((k_scale<u64>), dim3(ew_grid(words)), dim3(256), 0, r->stream, r->d64, (u64*)p, (const u64*)src_crt, words, sm);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { (void)hipFree(p); return fail(ALCH_E_HIP, std::string("hint conversion: ") + hipGetErrorString(e)); }
    rc = ALCH_OK;
    *out = new alch_hint{r, gadget, digits, p};
    return rc;
}

extern "C" int alch_buf_scale(alch_buf* dst, const alch_buf* src, size_t count, const uint64_t* s) try {
    if (!dst || !src || !s) return fail(ALCH_E_INVALID, "null argument");
    if (dst->ring != src->ring) return fail(ALCH_E_INVALID, "buffers belong to different rings");
    if (count > dst->n_elems || count > src->n_elems) return fail(ALCH_E_INVALID, "count out of bounds");
    alch_ring* r = dst->ring;
    BIND(r);
    return r->word == 4 ? do_scale<u32>(r, dst->dptr, src->dptr, count, s) : do_scale<u64>(r, dst->dptr, src->dptr, count, s);
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_decompose_triv(const alch_buf* src, size_t src_index, alch_buf* dst, size_t dst_first) try {
    if (!src || !dst) return fail(ALCH_E_INVALID, "null buffer");
    if (src->ring != dst->ring) return fail(ALCH_E_INVALID, "buffers belong to different rings");
    alch_ring* r = src->ring;
    BIND(r);
    if (src_index >= src->n_elems || !range_ok(dst_first, (size_t)r->L, dst->n_elems)) return fail(ALCH_E_INVALID, "element range out of bounds");
    const char* c = reinterpret_cast<const char*>(src->dptr) + src_index * elem_bytes(r);
    char* dig = reinterpret_cast<char*>(dst->dptr) + dst_first * elem_bytes(r);
    const size_t total = (size_t)r->L * elem_words(r);
    if (r->word == 4) hipLaunchKernelGGL((k_decompose_triv<u32>), dim3(ew_grid(total)), dim3(256), 0, r->stream, r->d32, (const u32*)c, (u32*)dig, r->balanced ? 1 : 0);
    else hipLaunchKernelGGL((k_decompose_triv<u64>), dim3(ew_grid(total)), dim3(256), 0, r->stream, r->d64, (const u64*)c, (u64*)dig, r->balanced ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_hint_load(alch_ring* r, int gadget, const int64_t* host_crt, alch_hint** out) try {
    if (!r || !host_crt || !out) return fail(ALCH_E_INVALID, "null argument");
    if (gadget != ALCH_GAD_TRIV && gadget != ALCH_GAD_BASE2) return fail(ALCH_E_INVALID, "unknown gadget");
    const size_t elems = 2 * (size_t)gadget_digits(r, gadget);
    ScratchBuf s;
    int rc = scratch_get(r, elems, &s.b);
    if (rc != ALCH_OK) return rc;
    if ((rc = alch_buf_upload(s.b, 0, elems, host_crt)) != ALCH_OK) return rc;
    rc = hint_from_device(r, gadget, s.b->dptr, out);
    if (rc == ALCH_OK) HIP_TRY(hipStreamSynchronize(r->stream));
    return rc;
} catch (...) { return abi_catch(); }

extern "C" int alch_hint_from_buf(alch_ring* r, int gadget, const alch_buf* src, alch_hint** out) try {
    if (!r || !src || !out) return fail(ALCH_E_INVALID, "null argument");
    if (gadget != ALCH_GAD_TRIV && gadget != ALCH_GAD_BASE2) return fail(ALCH_E_INVALID, "unknown gadget");
    if (src->ring != r || src->n_elems < 2 * (size_t)gadget_digits(r, gadget))
        return fail(ALCH_E_INVALID, "hint source needs 2 ring elements per gadget digit");
    return hint_from_device(r, gadget, src->dptr, out);
} catch (...) { return abi_catch(); }

extern "C" int alch_hint_free(alch_hint* h) try {
    if (!h) return ALCH_OK;
    (void)hipSetDevice(h->ring->device);
    (void)hipStreamSynchronize(h->ring->stream);
    (void)hipFree(h->dptr);
    delete h;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_ct_add_public(alch_buf* dst, const alch_buf* src, size_t batch, const uint64_t* s, const alch_buf* pub, size_t pub_index) try {
    if (!dst || !src || !pub) return fail(ALCH_E_INVALID, "null buffer");
    alch_ring* r = dst->ring;
    if (src->ring != r || pub->ring != r) return fail(ALCH_E_INVALID, "buffers belong to different rings");
    if (!r->has_crt) return fail(ALCH_E_UNSUPPORTED, "scalar products are implemented for rings with Montgomery constants (prime moduli) only");
    if (!pairs_ok(batch, dst->n_elems) || !pairs_ok(batch, src->n_elems) || pub_index >= pub->n_elems) return fail(ALCH_E_INVALID, "count out of bounds");
    if (batch == 0) return ALCH_OK;
    BIND(r);
    const size_t words = 2 * batch * elem_words(r);
    const char* pp = reinterpret_cast<const char*>(pub->dptr) + pub_index * elem_bytes(r);
    if (r->word == 4) {
        typedef u32 W;
        Scal<W> sm; scal_to_mont<W>(r, s, 1, sm);
        ALCH_LAUNCH_VW(k_scale_add_bcast, r, words, r->stream, r->d32, (W*)dst->dptr, (const W*)src->dptr, (const W*)pp, words, sm);
    } else {
        typedef u64 W;
        Scal<W> sm; scal_to_mont<W>(r, s, 1, sm);
        ALCH_LAUNCH_VW(k_scale_add_bcast, r, words, r->stream, r->d64, (W*)dst->dptr, (const W*)src->dptr, (const W*)pp, words, sm);
    }
    HIP_TRY(hipGetLastError());
    return ALCH_OK;
} catch (...) { return abi_catch(); }

// ------------------------------------------------------------------------------------------------------
// general index: l / lInv, mulG / divG on device buffers
// ------------------------------------------------------------------------------------------------------
// Runs one column operator (kernel_gen.hpp) on elements first, first + stride, ... (count of them) of `data`.
template <typename W>
static int do_columns(alch_ring* r, GenOp op, void* data, size_t first, size_t count, size_t stride, hipStream_t stream = nullptr,
                      const void* src = nullptr) {
    // src != null: out of place, src[e] -> data[e] (same element layout and stride)
    if (count == 0) return ALCH_OK;
    GenCall<W> g{};
    g.src = reinterpret_cast<const W*>(src);
    g.op = op;
    g.ring = &dev_ring<W>(r);
    g.gen = &gen_dev<W>(r);
    g.stream = stream ? stream : r->stream;
    g.data = reinterpret_cast<W*>(data);
    g.elem_stride = stride;
    g.zdom = r->zdom;
    g.fail_flag = r->d_flag;
    for (size_t done = 0; done < count;) {
        const size_t now = std::min<size_t>(count - done, ((size_t)1 << 30) / (size_t)r->L);
        g.first_poly = first + done * stride;
        g.npoly = now * (size_t)r->L;
        hipError_t e = gen_dispatch(g);
        if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("general-index column launch: ") + hipGetErrorString(e));
        done += now;
    }
    return ALCH_OK;
}

static int columns(alch_ring* r, GenOp op, void* data, size_t first, size_t count, size_t stride, hipStream_t stream = nullptr,
                   const void* src = nullptr) {
    return r->word == 4 ? do_columns<u32>(r, op, data, first, count, stride, stream, src)
                        : do_columns<u64>(r, op, data, first, count, stride, stream, src);
}

// mulG (divide = false) or divG on elements [first, first + count) of a buffer, basis ALCH_BASIS_*.
// Returns ALCH_OK, ALCH_NOT_DIVISIBLE (Lol's Nothing; the data are then unspecified) or an error.
static int buf_mulg_divg(alch_buf* b, size_t first, size_t count, int basis, bool divide) {
    if (!b) return fail(ALCH_E_INVALID, "null buffer");
    if (!range_ok(first, count, b->n_elems)) return fail(ALCH_E_INVALID, "element range out of bounds");
    if (basis != ALCH_BASIS_POW && basis != ALCH_BASIS_DEC && basis != ALCH_BASIS_CRT) return fail(ALCH_E_INVALID, "unknown basis");
    alch_ring* r = b->ring;
    BIND(r);
    if (basis == ALCH_BASIS_CRT && !r->has_crt) return fail(ALCH_E_NO_CRT, "this ring has no CRT basis");
    if (!r->gen || r->gh.rad == 1 || count == 0) return ALCH_OK;      // g = 1: two-power index
    if (basis == ALCH_BASIS_CRT) {
        const size_t words = count * elem_words(r);
        char* base = reinterpret_cast<char*>(b->dptr) + first * elem_bytes(r);
        if (r->word == 4) {
            GTab<u32> gt{};
            for (int j = 0; j < r->L; ++j) gt.p[j] = divide ? r->g32.gcrt_inv[j] : r->g32.gcrt[j];
            hipLaunchKernelGGL((k_mul_table<u32>), dim3(ew_grid(words)), dim3(256), 0, r->stream, r->d32, (u32*)base, words, gt);
        } else {
            GTab<u64> gt{};
            for (int j = 0; j < r->L; ++j) gt.p[j] = divide ? r->g64.gcrt_inv[j] : r->g64.gcrt[j];
            hipLaunchKernelGGL((k_mul_table<u64>), dim3(ew_grid(words)), dim3(256), 0, r->stream, r->d64, (u64*)base, words, gt);
        }
        HIP_TRY(hipGetLastError());
        return ALCH_OK;
    }
    if (!divide) return columns(r, basis == ALCH_BASIS_POW ? GEN_MULG_POW : GEN_MULG_DEC, b->dptr, first, count, 1);
    HIP_TRY(hipMemsetAsync(r->d_flag, 0, sizeof(int), r->stream));
    int rc = columns(r, basis == ALCH_BASIS_POW ? GEN_DIVG_POW : GEN_DIVG_DEC, b->dptr, first, count, 1);
    if (rc != ALCH_OK) return rc;
    int flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, r->d_flag, sizeof(int), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return flag ? ALCH_NOT_DIVISIBLE : ALCH_OK;
}

extern "C" int alch_buf_mulg(alch_buf* b, size_t first, size_t count, int basis) try { return buf_mulg_divg(b, first, count, basis, false); } catch (...) { return abi_catch(); }
extern "C" int alch_buf_divg(alch_buf* b, size_t first, size_t count, int basis) try { return buf_mulg_divg(b, first, count, basis, true); } catch (...) { return abi_catch(); }

static int buf_l(alch_buf* b, size_t first, size_t count, bool inverse) {
    if (!b) return fail(ALCH_E_INVALID, "null buffer");
    if (!range_ok(first, count, b->n_elems)) return fail(ALCH_E_INVALID, "element range out of bounds");
    alch_ring* r = b->ring;
    BIND(r);
    if (!r->gen || r->gh.rad == 1) return ALCH_OK;                    // L = identity for a two-power index
    return columns(r, inverse ? GEN_LINV : GEN_L, b->dptr, first, count, 1);
}
extern "C" int alch_buf_l(alch_buf* b, size_t first, size_t count) try { return buf_l(b, first, count, false); } catch (...) { return abi_catch(); }
extern "C" int alch_buf_linv(alch_buf* b, size_t first, size_t count) try { return buf_l(b, first, count, true); } catch (...) { return abi_catch(); }

// ------------------------------------------------------------------------------------------------------
// device-resident Tensor values: what a `GT m r` holds when it stays on the GPU between Tensor calls
// ------------------------------------------------------------------------------------------------------
// E issues one Lol call per op (Crypto/Alchemy/Interpreter/Eval.hs:120-134) and Lol one Tensor call per basis change, so a
// ciphertext operation reaches this library as a chain of single-element calls.  With the host-buffer entry points every link
// of that chain crosses PCIe twice; with these the element stays in HBM and a link costs one kernel launch.
extern "C" int alch_ring_share_stream(alch_ring* r, alch_ring* with) try {
    if (!r || !with) return fail(ALCH_E_INVALID, "null ring");
    if (r->device != with->device) return fail(ALCH_E_INVALID, "alch_ring_share_stream: the rings live on different devices");
    if (r == with || r->stream == with->stream) return ALCH_OK;
    BIND(r);
    HIP_TRY(hipStreamSynchronize(r->stream));
    r->stream_owner = with->stream_owner;          // releases r's own stream (destroyed with its last user), keeps with's alive
    r->stream = with->stream;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_copy(alch_buf* dst, size_t dst_first, const alch_buf* src, size_t src_first, size_t count) try {
    if (!dst || !src) return fail(ALCH_E_INVALID, "null buffer");
    if (dst->ring != src->ring) return fail(ALCH_E_INVALID, "buffers belong to different rings");
    if (!range_ok(dst_first, count, dst->n_elems) || !range_ok(src_first, count, src->n_elems)) return fail(ALCH_E_INVALID, "element range out of bounds");
    if (count == 0) return ALCH_OK;
    alch_ring* r = dst->ring;
    BIND(r);
    char* d = reinterpret_cast<char*>(dst->dptr) + dst_first * elem_bytes(r);
    const char* f = reinterpret_cast<const char*>(src->dptr) + src_first * elem_bytes(r);
    if (d == f) return ALCH_OK;
    if ((d < f ? f - d : d - f) < (ptrdiff_t)(count * elem_bytes(r))) return fail(ALCH_E_INVALID, "alch_buf_copy: overlapping ranges");
    HIP_TRY(hipMemcpyAsync(d, f, count * elem_bytes(r), hipMemcpyDeviceToDevice, r->stream));
    return ALCH_OK;
} catch (...) { return abi_catch(); }

// dst[dst_first + i] = op(src[src_first + i]), i < count.  Transforms and column operators read `src` and write `dst` in one
// kernel; the CRT-basis g products copy first.  dst and src may be the same range (in place).
extern "C" int alch_buf_tensor_op(alch_buf* dst, size_t dst_first, const alch_buf* src, size_t src_first, size_t count, int op) try {
    if (!dst || !src) return fail(ALCH_E_INVALID, "null buffer");
    if (dst->ring != src->ring) return fail(ALCH_E_INVALID, "buffers belong to different rings");
    if (!range_ok(dst_first, count, dst->n_elems) || !range_ok(src_first, count, src->n_elems)) return fail(ALCH_E_INVALID, "element range out of bounds");
    if (op < ALCH_T_CRT || op > ALCH_T_DIVG_CRT) return fail(ALCH_E_INVALID, "alch_buf_tensor_op: unknown op");
    alch_ring* r = dst->ring;
    BIND(r);
    const bool needs_crt = op == ALCH_T_CRT || op == ALCH_T_CRTINV || op == ALCH_T_MULG_CRT || op == ALCH_T_DIVG_CRT;
    if (needs_crt && !r->has_crt) return fail(ALCH_E_NO_CRT, "this ring has no CRT basis (created with alch_ring_create_nocrt)");
    if (count == 0) return ALCH_OK;
    char* d = reinterpret_cast<char*>(dst->dptr) + dst_first * elem_bytes(r);
    const char* f = reinterpret_cast<const char*>(src->dptr) + src_first * elem_bytes(r);
    const bool in_place = d == f;
    if (!in_place && (d < f ? f - d : d - f) < (ptrdiff_t)(count * elem_bytes(r))) return fail(ALCH_E_INVALID, "alch_buf_tensor_op: overlapping ranges");
    auto copy_first = [&]() -> int {
        if (!in_place) HIP_TRY(hipMemcpyAsync(d, f, count * elem_bytes(r), hipMemcpyDeviceToDevice, r->stream));
        return ALCH_OK;
    };
    int rc;
    if (op == ALCH_T_CRT || op == ALCH_T_CRTINV) {
        const bool inv = op == ALCH_T_CRTINV;
        if (in_place || split_ring_fwd(r)) {                       // the split transforms (n = 2^16 / 64-bit 2^15) work in place only
            if ((rc = copy_first()) != ALCH_OK) return rc;
            return r->word == 4 ? do_crt<u32>(r, d, 0, count, inv) : do_crt<u64>(r, d, 0, count, inv);
        }
        return r->word == 4 ? do_crt<u32>(r, d, 0, count, inv, f) : do_crt<u64>(r, d, 0, count, inv, f);
    }
    const bool trivial = !r->gen || r->gh.rad == 1;                   // two-power index: g = 1, L = identity
    if (trivial) return copy_first();
    if (op == ALCH_T_MULG_CRT || op == ALCH_T_DIVG_CRT) {
        if ((rc = copy_first()) != ALCH_OK) return rc;
        alch_buf view{r, count, d};
        return buf_mulg_divg(&view, 0, count, ALCH_BASIS_CRT, op == ALCH_T_DIVG_CRT);
    }
    GenOp g = GEN_L;
    bool divide = false;
    switch (op) {
    case ALCH_T_L: g = GEN_L; break;
    case ALCH_T_LINV: g = GEN_LINV; break;
    case ALCH_T_MULG_POW: g = GEN_MULG_POW; break;
    case ALCH_T_MULG_DEC: g = GEN_MULG_DEC; break;
    case ALCH_T_DIVG_POW: g = GEN_DIVG_POW; divide = true; break;
    default: g = GEN_DIVG_DEC; divide = true; break;
    }
    if (divide) HIP_TRY(hipMemsetAsync(r->d_flag, 0, sizeof(int), r->stream));
    rc = columns(r, g, d, 0, count, 1, nullptr, in_place ? nullptr : f);
    if (rc != ALCH_OK || !divide) return rc;
    int flag = 0;                                                     // Lol's Maybe: the answer is needed now
    HIP_TRY(hipMemcpyAsync(&flag, r->d_flag, sizeof(int), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return flag ? ALCH_NOT_DIVISIBLE : ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_mul_public(alch_buf* dst, const alch_buf* src, const alch_buf* pub, size_t pub_index, size_t count) try {
    if (!dst || !src || !pub) return fail(ALCH_E_INVALID, "null buffer");
    alch_ring* r = dst->ring;
    if (src->ring != r || pub->ring != r) return fail(ALCH_E_INVALID, "buffers belong to different rings");
    if (!r->has_crt) return fail(ALCH_E_NO_CRT, "this ring has no CRT basis");
    if (count > dst->n_elems || count > src->n_elems || pub_index >= pub->n_elems) return fail(ALCH_E_INVALID, "count out of bounds");
    BIND(r);
    const size_t words = count * elem_words(r);
    const char* pp = reinterpret_cast<const char*>(pub->dptr) + pub_index * elem_bytes(r);
    if (r->word == 4) hipLaunchKernelGGL((k_mul_bcast<u32>), dim3(ew_grid(words)), dim3(256), 0, r->stream, r->d32, (u32*)dst->dptr, (const u32*)src->dptr, (const u32*)pp, words);
    else hipLaunchKernelGGL((k_mul_bcast<u64>), dim3(ew_grid(words)), dim3(256), 0, r->stream, r->d64, (u64*)dst->dptr, (const u64*)src->dptr, (const u64*)pp, words);
    HIP_TRY(hipGetLastError());
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_add_public(alch_buf* cts, const alch_buf* pub, size_t pub_index, size_t batch) try {
    if (!cts || !pub) return fail(ALCH_E_INVALID, "null buffer");
    alch_ring* r = cts->ring;
    if (pub->ring != r) return fail(ALCH_E_INVALID, "buffers belong to different rings");
    if (!pairs_ok(batch, cts->n_elems) || pub_index >= pub->n_elems) return fail(ALCH_E_INVALID, "count out of bounds");
    BIND(r);
    const size_t words = batch * elem_words(r);
    const char* pp = reinterpret_cast<const char*>(pub->dptr) + pub_index * elem_bytes(r);
    if (r->word == 4) hipLaunchKernelGGL((k_add_bcast<u32>), dim3(ew_grid(words)), dim3(256), 0, r->stream, r->d32, (u32*)cts->dptr, (const u32*)pp, batch);
    else hipLaunchKernelGGL((k_add_bcast<u64>), dim3(ew_grid(words)), dim3(256), 0, r->stream, r->d64, (u64*)cts->dptr, (const u64*)pp, batch);
    HIP_TRY(hipGetLastError());
    return ALCH_OK;
} catch (...) { return abi_catch(); }

// ------------------------------------------------------------------------------------------------------
// the hot path
// ------------------------------------------------------------------------------------------------------
// rings whose limb-polynomial does not fit one LDS-resident transform (k_crt_split)
static bool split_ring(const alch_ring* r) { return r->logn > (r->word == 4 ? 15 : 14); }

// keySwitchQuadCirc hint (a * b), unfused: element-wise tensor product, batched crtInv of c2, decompose, batched crt
// of the digits, hint inner product.  Serves
//   * BaseBGad 2 hints (PT2CT.hs:140; Tunnel.hs:24 / HomomRLWR.hs:46 pick that gadget): D = sum_i ceil(log2 q_i)
//     digits, each reduced into every limb and transformed -- D*L crt per ciphertext against L*(L-1) for TrivGad,
//     two orders of magnitude heavier by construction;
//   * TrivGad on rings too large for one LDS-resident transform (n = 2^16 / 2^15), where the fused kernels do
//     not exist and crt runs as k_crt_split.
// The fused general-index key switch (k_gen_tensor_inv + k_gen_ks, kernel_gen.hpp): TrivGad, 32-bit words, n <= 12288.
static bool gen_ks_fused(const alch_ring* r, const alch_hint* hint) {
    return r->gen && r->word == 4 && hint->gadget == ALCH_GAD_TRIV && r->n <= (u32)(GEN_KS_T * GEN_KS_NPT) && r->opts.gen_fused;
}

// keySwitchQuadCirc hint (a * b) for `now` ciphertexts through the two fused general-index kernels; a, b live on the last
// L - dup limbs of r's moduli (Ls limbs, ring_in), ks receives [now][2][L][n].
static int launch_gen_ks(alch_ring* r, const alch_hint* hint, const u32* a, const u32* b, u32* ks, u32* c2pow, size_t now, int dup,
                         const Scal<u32>& sr2, hipStream_t stream) {
    GenKsArgs<u32> A{};
    A.a = a; A.b = b; A.c2pow = c2pow; A.hint = reinterpret_cast<const u32*>(hint->dptr); A.out = ks;
    A.sr2 = sr2; A.dup = dup; A.balanced = r->balanced ? 1 : 0; A.use_g = r->gh.rad > 1 ? 1 : 0;
    hipError_t e = gen_ks_dispatch(r->d32, r->g32, A, now, stream);
    if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("general-index fused key switch launch: ") + hipGetErrorString(e));
    return ALCH_OK;
}

template <typename W>
static void launch_hint_mac(alch_ring* r, hipStream_t stream, W* out, const W* digits, const W* hint, size_t nct, u32 D,
                            const W* diag = nullptr, u32 grp = 0, u32 hskip = 0, const u32* slot_e = nullptr, u32 n_d = 0,
                            bool pieces_ok = false) {
    if (slot_e && pieces_ok && !diag && r->n % Vec4<W>::LANES == 0) {
        const size_t pieces = (nct + 3) / 4 * (size_t)r->L * (r->n / Vec4<W>::LANES);
        hipLaunchKernelGGL((k_hint_mac_e<W, 4>), dim3(ew_grid(pieces)), dim3(256), 0, stream, dev_ring<W>(r), out, digits, hint, nct, D, grp, hskip, slot_e, n_d);
    } else if (r->n % Vec4<W>::LANES == 0) {
        const size_t pieces = (nct + 3) / 4 * (size_t)r->L * (r->n / Vec4<W>::LANES);
        hipLaunchKernelGGL((k_hint_mac_v<W, 4>), dim3(ew_grid(pieces)), dim3(256), 0, stream, dev_ring<W>(r), out, digits, hint, nct, D, diag, grp, hskip, slot_e, n_d);
    } else {
        hipLaunchKernelGGL((k_hint_mac<W>), dim3(ew_grid(nct * elem_words(r))), dim3(256), 0, stream, dev_ring<W>(r), out, digits, hint, nct, D, diag, grp, hskip, slot_e, n_d);
    }
}

template <typename W>
static int do_mul_relin_unfused(alch_ring* r, const alch_hint* hint, const void* a, const void* b, void* out, size_t batch,
                                const uint64_t* s_pre) {
    Scal<u32> first, kd;
    const bool base2 = hint->gadget == ALCH_GAD_BASE2;
    const u32 D = base2 ? (u32)base2_layout(r, first, kd) : (u32)r->L;
    const size_t eb = elem_bytes(r);
    // scratch: c2 (1 element) + digits (D elements) per ciphertext of a chunk, at most ~1 GiB
    const bool fused_digits = !base2 && (split_ring(r) || r->gen);
    // the one-kernel forms (k_ks_accum_split, k_gen_ks) keep no digits in HBM: only c2 in both bases
    const bool no_digits = fused_digits && ((!r->gen && r->opts.split_fused) || gen_ks_fused(r, hint));
    const size_t per_ct = (no_digits ? 2 : D + 2) * eb;
    size_t chunk = std::max<size_t>(1, (r->scratch_mib << 20) / per_ct);
    chunk = std::min(chunk, batch);
    int rc = ensure_ws(&r->ws_digits, &r->ws_digits_bytes, chunk * per_ct);
    if (rc != ALCH_OK) return rc;
    char* c2 = reinterpret_cast<char*>(r->ws_digits);
    char* c2crt = c2 + chunk * eb;                       // CRT-basis copy of c2 (diagonal digits, split rings)
    char* dig = c2crt + chunk * eb;
    Scal<W> sr2;
    scal_to_mont<W>(r, s_pre, 2, sr2);
    GTab<W> gt{};
    if (r->gen) for (int j = 0; j < r->L; ++j) gt.p[j] = r->gh.rad > 1 ? gen_dev<W>(r).gcrt[j] : nullptr;
    const size_t ct_bytes = 2 * eb;
    for (size_t done = 0; done < batch; done += chunk) {
        const size_t now = std::min(chunk, batch - done);
        const W* pa = reinterpret_cast<const W*>(reinterpret_cast<const char*>(a) + done * ct_bytes);
        const W* pb = reinterpret_cast<const W*>(reinterpret_cast<const char*>(b) + done * ct_bytes);
        W* po = reinterpret_cast<W*>(reinterpret_cast<char*>(out) + done * ct_bytes);
        const size_t words = now * elem_words(r);
        if constexpr (sizeof(W) == 4) {
            if (gen_ks_fused(r, hint)) {                          // two fused launches: tensor + crtInv, digit transforms + hint products
                if ((rc = launch_gen_ks(r, hint, pa, pb, po, reinterpret_cast<u32*>(c2), now, 0, sr2, r->stream)) != ALCH_OK) return rc;
                continue;
            }
        }
        ALCH_LAUNCH_VW(k_tensor_ew, r, words, r->stream, dev_ring<W>(r), pa, pb, po,
                           (W*)c2, now, sr2, 0, fused_digits ? (W*)c2crt : (W*)nullptr, gt);
        HIP_TRY(hipGetLastError());
        if ((rc = do_crt<W>(r, c2, 0, now, true)) != ALCH_OK) return rc;
        if (fused_digits && r->gen) {                               // general index: the same, on the pass engine
            GenCall<W> g{};
            g.op = GEN_CRT_DIGITS;
            g.ring = &dev_ring<W>(r);
            g.gen = &gen_dev<W>(r);
            g.stream = r->stream;
            g.src = reinterpret_cast<const W*>(c2);
            g.data = reinterpret_cast<W*>(dig);
            g.npoly = now * (size_t)r->L * (size_t)r->L;
            g.balanced = r->balanced;
            hipError_t e = gen_dispatch(g);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("general-index crt_digits launch: ") + hipGetErrorString(e));
            launch_hint_mac<W>(r, r->stream, po, (const W*)dig, (const W*)hint->dptr, now, D, (const W*)c2crt);
            HIP_TRY(hipGetLastError());
            continue;
        }
        if (fused_digits && r->opts.split_fused) {                  // split rings: digit transforms + hint products in one kernel
            NttCall<W> dc{};
            dc.op = OP_KS_SPLIT;
            dc.ring = &dev_ring<W>(r);
            dc.stream = r->stream;
            dc.src = reinterpret_cast<const W*>(c2);
            dc.a = reinterpret_cast<const W*>(c2crt);
            dc.hint = reinterpret_cast<const W*>(hint->dptr);
            dc.out = po;
            dc.nct = now;
            dc.dup = 0;
            dc.balanced = r->balanced;
            hipError_t e = dispatch(r->logn, dc);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("ks_accum_split launch: ") + hipGetErrorString(e));
            continue;
        }
        if (fused_digits) {                                         // decompose fused into the digit transforms
            NttCall<W> dc{};
            dc.op = OP_CRT_DIGITS;
            dc.ring = &dev_ring<W>(r);
            dc.stream = r->stream;
            dc.src = reinterpret_cast<const W*>(c2);
            dc.data = reinterpret_cast<W*>(dig);
            dc.npoly = now * (size_t)r->L * (size_t)r->L;
            dc.balanced = r->balanced;
            hipError_t e = dispatch(r->logn, dc);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("crt_digits launch: ") + hipGetErrorString(e));
            launch_hint_mac<W>(r, r->stream, po, (const W*)dig, (const W*)hint->dptr, now, D, (const W*)c2crt);
            HIP_TRY(hipGetLastError());
            continue;
        }
        if (base2 && !split_ring(r) && !r->gen) {                   // BaseBGad: digits computed in the transforms' loader
            NttCall<W> dc{};
            dc.op = OP_CRT_BASE2;
            dc.ring = &dev_ring<W>(r);
            dc.stream = r->stream;
            dc.src = reinterpret_cast<const W*>(c2);
            dc.data = reinterpret_cast<W*>(dig);
            dc.npoly = now * (size_t)D * (size_t)r->L;
            dc.b2_first = first; dc.b2_kd = kd; dc.b2_D = D;
            hipError_t e = dispatch(r->logn, dc);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("crt_base2 launch: ") + hipGetErrorString(e));
            launch_hint_mac<W>(r, r->stream, po, (const W*)dig, (const W*)hint->dptr, now, D);
            HIP_TRY(hipGetLastError());
            continue;
        }
        for (size_t y0 = 0; y0 < now; y0 += 32768) {             // grid.y is 16-bit
            const unsigned ny = (unsigned)std::min<size_t>(32768, now - y0);
            const W* src = reinterpret_cast<const W*>(c2 + y0 * eb);
            W* dst = reinterpret_cast<W*>(dig + y0 * D * eb);
            if (base2)
                hipLaunchKernelGGL((k_decompose_base2<W>), dim3(ew_grid(elem_words(r)), ny), dim3(256), 0, r->stream,
                                   dev_ring<W>(r), src, dst, first, kd, D);
            else
                hipLaunchKernelGGL((k_decompose_triv<W>), dim3(ew_grid((size_t)r->L * elem_words(r)), ny), dim3(256), 0,
                                   r->stream, dev_ring<W>(r), src, dst, r->balanced ? 1 : 0);
            HIP_TRY(hipGetLastError());
        }
        if ((rc = do_crt<W>(r, dig, 0, now * D, false)) != ALCH_OK) return rc;
        launch_hint_mac<W>(r, r->stream, po, (const W*)dig, (const W*)hint->dptr, now, D);
        HIP_TRY(hipGetLastError());
    }
    return ALCH_OK;
}

template <typename W>
static int do_mul_relin(alch_ring* r, const alch_hint* hint, const void* a, const void* b, void* out, size_t batch,
                        const uint64_t* s_pre) {
    typedef typename Signed<W>::type SW;
    // The batch runs as chunks of (tensor_intt, ks_accum) launch pairs.  A chunk's digits (chunk * L * n
    // signed words) are written by the first kernel and read 2(L-1) times by the second, so the chunk is kept
    // small enough for them to stay in the 256 MiB Infinity Cache; to keep the GPU full across the kernel
    // boundaries of such small launches, even and odd chunks run as two independent pipelines on two streams
    // (each with its own digit scratch), so one pipeline's tail overlaps the other's head.
    // the key-switch kernel addresses each array with 32-bit byte offsets: a chunk's ciphertexts stay below 4 GiB
    const size_t cap = std::max<size_t>(8, (((size_t)1 << 32) - 1) / (2 * elem_bytes(r)) / 8 * 8);
    const size_t chunk = std::min(std::min(r->chunk, cap), (batch + 7) / 8 * 8);
    const size_t dig_bytes = chunk * elem_words(r) * sizeof(SW);
    const int ns = r->one_stream ? 1 : r->nstreams;
    int rc = ensure_ws(&r->ws_digits, &r->ws_digits_bytes, (size_t)std::max(2, ns) * dig_bytes);
    if (rc != ALCH_OK) return rc;
    for (int e = 0; e + 2 < ns; ++e) {
        if (!r->xs[e]) HIP_TRY(hipStreamCreateWithFlags(&r->xs[e], hipStreamNonBlocking));
        if (!r->ev_xs[e]) HIP_TRY(hipEventCreateWithFlags(&r->ev_xs[e], hipEventDisableTiming));
    }
    hipStream_t lanes[4] = {r->stream, r->aux, r->xs[0], r->xs[1]};
    NttCall<W> c{};
    c.ring = &dev_ring<W>(r);
    c.hint = reinterpret_cast<const W*>(hint->dptr);
    c.balanced = r->balanced;
    c.opts = r->opts;
    c.q30 = r->q30 && r->opts.q30;
    c.partials = true;                              // read by ALCH_A_PARTIALS builds only (kernel_tensor_split.hpp)
    scal_to_mont<W>(r, s_pre, 2, c.spre_r2);
    const size_t ct_words = 2 * elem_words(r);
    const bool two = batch > chunk && !r->one_stream;
    if (two) {
        HIP_TRY(hipEventRecord(r->ev_fork, r->stream));
        for (int e = 1; e < ns; ++e) HIP_TRY(hipStreamWaitEvent(lanes[e], r->ev_fork, 0));
    }
    size_t idx = 0;
    if (two && r->pipe) {
        // Anti-phase pipeline: every tensor kernel on the aux stream, every key-switch kernel on the ring's stream, the tensor
        // kernel of chunk k+1 released as soon as chunk k-1's key switch has freed its digit scratch -- it shares the CUs with
        // chunk k's key switch (a latency-bound kernel next to a VALU-bound one) instead of with another tensor kernel.
        for (int e = 0; e < 2; ++e) {
            if (!r->ev_pa[e]) HIP_TRY(hipEventCreateWithFlags(&r->ev_pa[e], hipEventDisableTiming));
            if (!r->ev_pb[e]) HIP_TRY(hipEventCreateWithFlags(&r->ev_pb[e], hipEventDisableTiming));
        }
        // the tensor kernels run on the aux stream whatever `nstreams` says: order it behind everything queued on the ring's stream
        // (the producers of a and b, and the previous call's key-switch kernels, which still read the digit scratch)
        if (ns < 2) HIP_TRY(hipStreamWaitEvent(r->aux, r->ev_fork, 0));
        for (size_t done = 0; done < batch; done += chunk, ++idx) {
            const size_t now = std::min(chunk, batch - done);
            const int par = (int)(idx & 1);
            c.digits = reinterpret_cast<char*>(r->ws_digits) + (par ? dig_bytes : 0);
            c.a = reinterpret_cast<const W*>(a) + done * ct_words;
            c.b = reinterpret_cast<const W*>(b) + done * ct_words;
            c.out = reinterpret_cast<W*>(out) + done * ct_words;
            c.nct = now;
            if (idx >= 2) HIP_TRY(hipStreamWaitEvent(r->aux, r->ev_pb[par], 0));
            c.stream = r->aux;
            c.op = OP_TENSOR_INTT;
            hipError_t e = dispatch(r->logn, c);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("tensor_intt launch: ") + hipGetErrorString(e));
            HIP_TRY(hipEventRecord(r->ev_pa[par], r->aux));
            HIP_TRY(hipStreamWaitEvent(r->stream, r->ev_pa[par], 0));
            c.stream = r->stream;
            c.op = OP_KS_ACCUM;
            e = dispatch(r->logn, c);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("ks_accum launch: ") + hipGetErrorString(e));
            HIP_TRY(hipEventRecord(r->ev_pb[par], r->stream));
        }
        return ALCH_OK;                                    // the last key switch is on the ring's stream, behind every tensor kernel
    }
    for (size_t done = 0; done < batch; done += chunk, ++idx) {
        const size_t now = std::min(chunk, batch - done);
        const int lane = two ? (int)(idx % (size_t)ns) : 0;
        c.stream = lanes[lane];
        c.digits = reinterpret_cast<char*>(r->ws_digits) + (size_t)lane * dig_bytes;
        c.a = reinterpret_cast<const W*>(a) + done * ct_words;
        c.b = reinterpret_cast<const W*>(b) + done * ct_words;
        c.out = reinterpret_cast<W*>(out) + done * ct_words;
        c.nct = now;
        c.op = OP_TENSOR_INTT;
        hipError_t e = dispatch(r->logn, c);
        if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("tensor_intt launch: ") + hipGetErrorString(e));
        c.op = OP_KS_ACCUM;
        e = dispatch(r->logn, c);
        if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("ks_accum launch: ") + hipGetErrorString(e));
    }
    if (two) {
        HIP_TRY(hipEventRecord(r->ev_join, r->aux));
        HIP_TRY(hipStreamWaitEvent(r->stream, r->ev_join, 0));
        for (int e = 2; e < ns; ++e) {
            HIP_TRY(hipEventRecord(r->ev_xs[e - 2], lanes[e]));
            HIP_TRY(hipStreamWaitEvent(r->stream, r->ev_xs[e - 2], 0));
        }
    }
    return ALCH_OK;
}

extern "C" int alch_ct_mul_relin(alch_ring* r, const alch_hint* hint, const alch_buf* a, const alch_buf* b, alch_buf* out,
                                 size_t batch, const uint64_t* s_pre, unsigned flags) try {
    if (!r || !hint || !a || !b || !out) return fail(ALCH_E_INVALID, "null argument");
    if (!r->has_crt) return fail(ALCH_E_NO_CRT, "this ring has no CRT basis");
    if (hint->ring != r || a->ring != r || b->ring != r || out->ring != r) return fail(ALCH_E_INVALID, "handles belong to different rings");
    if (batch == 0) return ALCH_OK;
    BIND(r);
    if (!pairs_ok(batch, a->n_elems) || !pairs_ok(batch, b->n_elems) || !pairs_ok(batch, out->n_elems))
        return fail(ALCH_E_INVALID, "buffers must hold 2*batch ring elements");
    if (out == a || out == b) return fail(ALCH_E_INVALID, "out must not alias an input");
    const void* pa = a->dptr;
    const void* pb = b->dptr;
    int rc;
    if (flags & ALCH_POW_IN) {
        const size_t bytes = 2 * batch * elem_bytes(r);
        if ((rc = ensure_ws(&r->ws_in, &r->ws_in_bytes, 2 * bytes)) != ALCH_OK) return rc;
        char* wa = reinterpret_cast<char*>(r->ws_in);
        char* wb = wa + bytes;
        if (split_ring(r)) {                 // the split transform works in place: copy first (general-index kernels take src)
            HIP_TRY(hipMemcpyAsync(wa, a->dptr, bytes, hipMemcpyDeviceToDevice, r->stream));
            HIP_TRY(hipMemcpyAsync(wb, b->dptr, bytes, hipMemcpyDeviceToDevice, r->stream));
            rc = r->word == 4 ? do_crt<u32>(r, wa, 0, 4 * batch, false) : do_crt<u64>(r, wa, 0, 4 * batch, false);
        } else {                             // out of place, straight from the operands
            rc = r->word == 4 ? do_crt<u32>(r, wa, 0, 2 * batch, false, a->dptr) : do_crt<u64>(r, wa, 0, 2 * batch, false, a->dptr);
            if (rc == ALCH_OK)
                rc = r->word == 4 ? do_crt<u32>(r, wb, 0, 2 * batch, false, b->dptr) : do_crt<u64>(r, wb, 0, 2 * batch, false, b->dptr);
        }
        if (rc != ALCH_OK) return rc;
        pa = wa;
        pb = wb;
    }
    // split rings (a limb-polynomial is twice an LDS-resident transform), TrivGad: the same two-launch structure as the LDS-resident sizes
    // since round 3 (k_tensor_crtinv_split + k_ks_accum_split<FROM_OPS>); option split_fused < 2 keeps the composed forms
    const bool split_two = split_ring(r) && !r->gen && hint->gadget == ALCH_GAD_TRIV && r->opts.split_fused >= 2;
    if (!split_two && (hint->gadget == ALCH_GAD_BASE2 || split_ring(r) || r->gen))
        rc = r->word == 4 ? do_mul_relin_unfused<u32>(r, hint, pa, pb, out->dptr, batch, s_pre)
                          : do_mul_relin_unfused<u64>(r, hint, pa, pb, out->dptr, batch, s_pre);
    else
        rc = r->word == 4 ? do_mul_relin<u32>(r, hint, pa, pb, out->dptr, batch, s_pre)
                          : do_mul_relin<u64>(r, hint, pa, pb, out->dptr, batch, s_pre);
    if (rc != ALCH_OK) return rc;
    if (flags & ALCH_POW_OUT) return buf_crt(out, 0, 2 * batch, true);
    return ALCH_OK;
} catch (...) { return abi_catch(); }

// ---- the complete mul_: (*) . modSwitch up . keySwitchQuadCirc . modSwitch down -------------------------------
// Constants of the closing modSwitch that drops the first ddn limbs of `r` (outermost first): q_u^-1 mod q_t for the
// limb-at-a-time form, prod_{w >= u} q_w^-1 mod q_t for the forms that keep the surviving limbs in the CRT basis.
template <typename W>
static void fill_drop_tab(const alch_ring* r, int ddn, DropTab<W>& d) {
    d.ddn = ddn;
    d.balanced = 1;
    const int bits = 8 * (int)sizeof(W);
    for (int u = 0; u < MAXDROP; ++u)
        for (int t = 0; t < MAXL; ++t) { d.qinv_m[u][t] = 0; d.comb_m[u][t] = 0; }
    for (int u = 0; u < ddn; ++u)
        for (int t = u + 1; t < r->L; ++t) {
            const u64 qt = r->q[t], qu = r->q[u];
            if ((qu - 1) / 2 >= qt) d.balanced = 0;
            const u64 inv = h_powmod(qu % qt, qt - 2, qt);
            d.qinv_m[u][t] = (W)h_mulmod(inv, h_powmod(2, (u64)bits, qt), qt);
        }
    for (int u = 0; u < ddn; ++u)
        for (int t = ddn; t < r->L; ++t) {
            const u64 qt = r->q[t];
            u64 v = 1;
            for (int w = u; w < ddn; ++w) v = h_mulmod(v, h_powmod(r->q[w] % qt, qt - 2, qt), qt);
            d.comb_m[u][t] = (W)h_mulmod(v, h_powmod(2, (u64)bits, qt), qt);
        }
}

template <typename W>
static int do_mul_full(alch_ring* rh, alch_ring* rin, alch_ring* rout, const alch_hint* hint, const void* a, const void* b,
                       void* out, size_t batch, const uint64_t* s_pre, bool pow_out) {
    typedef typename Signed<W>::type SW;
    const int dup = rh->L - rin->L, ddn = rh->L - rout->L;
    const size_t n = rh->n;
    // per pipeline: digits [chunk][L_in][n] signed | key-switched chunk [chunk][2][Lh][n] | stash [slots][ddn][n] signed
    const size_t cap = std::max<size_t>(8, (((size_t)1 << 32) - 1) / (2 * elem_bytes(rh)) / 8 * 8);   // 32-bit byte offsets per array
    const size_t chunk = std::min(std::min(rh->chunk, cap), (batch + 7) / 8 * 8);
    const unsigned slots = rh->rs_slots;       // resident workgroups of k_rescale_out (each owns a stash slot)
    const size_t dig_bytes = chunk * (size_t)rin->L * n * sizeof(SW);
    const size_t ks_bytes = chunk * 2 * (size_t)rh->L * n * sizeof(W);
    // stash: one slot per resident workgroup (k_rescale_out / _lin) or the lifted residues of the whole chunk (kernel_rescale_half.hpp)
    const size_t stash_bytes = std::max<size_t>(slots, 2 * chunk) * (size_t)ddn * n * sizeof(SW);
    const size_t pipe_bytes = dig_bytes + ks_bytes + stash_bytes;
    int rc = ensure_ws(&rh->ws_full, &rh->ws_full_bytes, 2 * pipe_bytes);
    if (rc != ALCH_OK) return rc;

    // s_pre times the moduli the first modSwitch adds
    uint64_t s_eff[MAXL];
    for (int j = 0; j < rin->L; ++j) {
        u64 v = s_pre ? s_pre[j] % rin->q[j] : 1;
        for (int u = 0; u < dup; ++u) v = h_mulmod(v, rh->q[u] % rin->q[j], rin->q[j]);
        s_eff[j] = v;
    }
    NttCall<W> c{};
    c.hint = reinterpret_cast<const W*>(hint->dptr);
    c.balanced = rh->balanced;
    c.opts = rh->opts;
    c.q30 = rh->q30 && rh->opts.q30;
    scal_to_mont<W>(rin, s_eff, 2, c.spre_r2);
    fill_drop_tab<W>(rh, ddn, c.drop);
    c.stash_slots = slots;
    c.pow_out = pow_out;

    const bool two = batch > chunk && !rh->one_stream;
    if (two) {
        HIP_TRY(hipEventRecord(rh->ev_fork, rh->stream));
        HIP_TRY(hipStreamWaitEvent(rh->aux, rh->ev_fork, 0));
    }
    const size_t in_words = 2 * (size_t)rin->L * n, out_words = 2 * (size_t)rout->L * n;
    size_t idx = 0;
    for (size_t done = 0; done < batch; done += chunk, ++idx) {
        const size_t now = std::min(chunk, batch - done);
        const bool odd = two && (idx & 1);
        char* base = reinterpret_cast<char*>(rh->ws_full) + (odd ? pipe_bytes : 0);
        c.stream = odd ? rh->aux : rh->stream;
        c.nct = now;
        // (*) 's quadratic coefficient, scaled, crtInv, TrivGad digits -- on the operands' ring
        c.op = OP_TENSOR_INTT;
        c.ring = &dev_ring<W>(rin);
        c.a = reinterpret_cast<const W*>(a) + done * in_words;
        c.b = reinterpret_cast<const W*>(b) + done * in_words;
        c.digits = base;
        hipError_t e = dispatch(rh->logn, c);
        if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("tensor_intt launch: ") + hipGetErrorString(e));
        // key switch on the hint's ring
        c.op = OP_KS_ACCUM;
        c.ring = &dev_ring<W>(rh);
        c.dup = dup;
        c.out = reinterpret_cast<W*>(base + dig_bytes);
        e = dispatch(rh->logn, c);
        if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("ks_accum launch: ") + hipGetErrorString(e));
        // modSwitch down
        c.op = OP_RESCALE_OUT;
        c.a = reinterpret_cast<const W*>(base + dig_bytes);
        c.out = reinterpret_cast<W*>(out) + done * out_words;
        c.stash = base + dig_bytes + ks_bytes;
        e = dispatch(rh->logn, c);
        if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("rescale_out launch: ") + hipGetErrorString(e));
    }
    if (two) {
        HIP_TRY(hipEventRecord(rh->ev_join, rh->aux));
        HIP_TRY(hipStreamWaitEvent(rh->stream, rh->ev_join, 0));
    }
    return ALCH_OK;
}

template <typename W> static void gen_suffix_view(alch_ring* r, int u, DevRing<W>& d, GenDev<W>& g);

// The same mul_ for rings whose polynomial does not fit one LDS-resident transform (split crt, n = 2^16 / 2^15):
// composed from the element-wise kernels and batched transforms, one ciphertext chunk at a time.
template <typename W>
static int do_mul_full_unfused(alch_ring* rh, alch_ring* rin, alch_ring* rout, const alch_hint* hint, const void* a,
                               const void* b, void* out, size_t batch, const uint64_t* s_pre, bool pow_out) {
    const int L = rh->L, dup = L - rin->L, ddn = L - rout->L;
    const size_t eb = elem_bytes(rh);
    // scratch per ciphertext: key-switched pair (2) + c2 in both bases (2) + digits (L) + rescale ping-pong (2 + 2), in ring_h elements
    const bool no_digits = (!rh->gen && rh->opts.split_fused) || gen_ks_fused(rh, hint);    // one-kernel key switch: no digits in HBM
    const size_t dig_elems = no_digits ? 0 : (size_t)L;
    const size_t per_ct = (size_t)(2 + 2 + 4 + dig_elems) * eb;
    size_t chunk = std::max<size_t>(1, (rh->scratch_mib << 20) / per_ct);
    chunk = std::min(chunk, batch);
    int rc = ensure_ws(&rh->ws_full, &rh->ws_full_bytes, chunk * per_ct);
    if (rc != ALCH_OK) return rc;
    char* ks = reinterpret_cast<char*>(rh->ws_full);
    char* c2 = ks + chunk * 2 * eb;
    char* c2crt = c2 + chunk * eb;
    char* dig = c2crt + chunk * eb;
    char* ping = dig + chunk * dig_elems * eb;
    char* pong = ping + chunk * 2 * eb;
    uint64_t s_eff[MAXL];
    for (int j = 0; j < rin->L; ++j) {
        u64 v = s_pre ? s_pre[j] % rin->q[j] : 1;
        for (int u = 0; u < dup; ++u) v = h_mulmod(v, rh->q[u] % rin->q[j], rin->q[j]);
        s_eff[j] = v;
    }
    Scal<W> sr2;
    scal_to_mont<W>(rin, s_eff, 2, sr2);
    GTab<W> gt{};
    if (rh->gen) for (int j = 0; j < L; ++j) gt.p[j] = rh->gh.rad > 1 ? gen_dev<W>(rh).gcrt[j] : nullptr;
    const bool dec_c0 = rh->gen && rh->gh.rad > 1;     // general index: modSwitch rescales c0 on the Dec basis (rescaleDec), c1 on Pow
    // DevRing of the suffix ring that starts at limb u of ring_h
    auto suffix = [&](int u) {
        DevRing<W> d = dev_ring<W>(rh);
        d.L = L - u;
        for (int j = 0; j + u < L; ++j) {
            d.mod[j] = d.mod[j + u]; d.ninv_m[j] = d.ninv_m[j + u]; d.w1ninv_m[j] = d.w1ninv_m[j + u];
            d.twf[j] = d.twf[j + u]; d.twi[j] = d.twi[j + u]; d.twp[j] = d.twp[j + u];
            d.tws[j] = d.tws[j + u]; d.twsi[j] = d.twsi[j + u];
        }
        return d;
    };
    const size_t in_bytes = 2 * elem_bytes(rin), out_bytes = 2 * elem_bytes(rout);
    for (size_t done = 0; done < batch; done += chunk) {
        const size_t now = std::min(chunk, batch - done);
        const W* pa = reinterpret_cast<const W*>(reinterpret_cast<const char*>(a) + done * in_bytes);
        const W* pb = reinterpret_cast<const W*>(reinterpret_cast<const char*>(b) + done * in_bytes);
        const size_t words = now * elem_words(rh);
        // (*) and modSwitch up
        bool fused_ks = false, split_done = false;
        if constexpr (sizeof(W) == 4) {
            if (gen_ks_fused(rh, hint)) {
                if ((rc = launch_gen_ks(rh, hint, pa, pb, reinterpret_cast<u32*>(ks), reinterpret_cast<u32*>(c2), now, dup, sr2, rh->stream)) != ALCH_OK) return rc;
                fused_ks = true;
            }
        }
        if (!fused_ks) {
        // general index below the hint's ring: the limbs the first modSwitch adds are zero in c2, so c2, its crtInv and its digits
        // are kept on the operands' L - dup limbs only (a fifth to a half of the digit transforms of PT2CT's products)
        const bool c2_compact = rh->gen && dup > 0;
        const int Ls = L - dup;
        ALCH_LAUNCH_VW(k_tensor_ew, rh, words, rh->stream, dev_ring<W>(rh), pa, pb, (W*)ks,
                           (W*)c2, now, sr2, dup, (W*)c2crt, gt, c2_compact ? 1 : 0);
        HIP_TRY(hipGetLastError());
        // keySwitchQuadCirc on ring_h
        if (c2_compact) {
            DevRing<W> dv; GenDev<W> gv;
            gen_suffix_view<W>(rh, dup, dv, gv);
            GenCall<W> g{};
            g.op = GEN_CRTINV; g.ring = &dv; g.gen = &gv; g.stream = rh->stream;
            g.data = reinterpret_cast<W*>(c2); g.first_poly = 0; g.npoly = now * (size_t)Ls;
            hipError_t e = gen_dispatch(g);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("general-index crtInv launch: ") + hipGetErrorString(e));
        } else if ((rc = do_crt<W>(rh, c2, 0, now, true)) != ALCH_OK) return rc;
        if (rh->gen) {
            GenCall<W> g{};
            g.op = GEN_CRT_DIGITS;
            g.ring = &dev_ring<W>(rh);
            g.gen = &gen_dev<W>(rh);
            g.stream = rh->stream;
            g.src = reinterpret_cast<const W*>(c2);
            g.data = reinterpret_cast<W*>(dig);
            g.npoly = now * (size_t)(c2_compact ? Ls : L) * (size_t)L;
            g.balanced = rh->balanced;
            if (c2_compact) { g.src_limbs = Ls; g.src_first = dup; }
            hipError_t e = gen_dispatch(g);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("general-index crt_digits launch: ") + hipGetErrorString(e));
        } else if (rh->opts.split_fused) {   // digit transforms + hint products in one kernel (k_ks_accum_split)
            NttCall<W> dc{};
            dc.op = OP_KS_SPLIT;
            dc.ring = &dev_ring<W>(rh);
            dc.stream = rh->stream;
            dc.src = reinterpret_cast<const W*>(c2);
            dc.a = reinterpret_cast<const W*>(c2crt);
            dc.hint = reinterpret_cast<const W*>(hint->dptr);
            dc.out = reinterpret_cast<W*>(ks);
            dc.nct = now;
            dc.dup = dup;
            dc.balanced = rh->balanced;
            hipError_t e = dispatch(rh->logn, dc);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("ks_accum_split launch: ") + hipGetErrorString(e));
            split_done = true;
        } else {   // decompose + reduce fused into the digit transforms (k_crt_split_digits)
            NttCall<W> dc{};
            dc.op = OP_CRT_DIGITS;
            dc.ring = &dev_ring<W>(rh);
            dc.stream = rh->stream;
            dc.src = reinterpret_cast<const W*>(c2);
            dc.data = reinterpret_cast<W*>(dig);
            dc.npoly = now * (size_t)L * (size_t)L;
            dc.balanced = rh->balanced;
            hipError_t e = dispatch(rh->logn, dc);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("crt_digits launch: ") + hipGetErrorString(e));
        }
        if (!split_done) {
        if (c2_compact) launch_hint_mac<W>(rh, rh->stream, (W*)ks, (const W*)dig, (const W*)hint->dptr, now, (u32)Ls, (const W*)c2crt, (u32)Ls, (u32)dup);
        else launch_hint_mac<W>(rh, rh->stream, (W*)ks, (const W*)dig, (const W*)hint->dptr, now, (u32)L, (const W*)c2crt);
        HIP_TRY(hipGetLastError());
        }
        }
        if (rh->gen && rh->opts.rs_lin && !pow_out) {
            // modSwitch down with the kept limbs in the CRT basis (k_gen_rescale_drop / k_gen_rescale_keep): ddn inverse +
            // (L - ddn) forward transforms per component instead of L + (L - ddn)
            DropTab<W> dt;
            fill_drop_tab<W>(rh, ddn, dt);
            hipError_t e = gen_rescale_lin_dispatch(dev_ring<W>(rh), gen_dev<W>(rh), reinterpret_cast<const W*>(ks), reinterpret_cast<W*>(ping),
                                                    reinterpret_cast<W*>(reinterpret_cast<char*>(out) + done * out_bytes), dt, dec_c0 ? 1 : 0,
                                                    2 * now, rh->stream);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("rescale launch: ") + hipGetErrorString(e));
            continue;
        }
        // modSwitch down: Pow basis (c0: Dec basis for a general index), one limb at a time, then back to the CRT basis on ring_out
        if ((rc = do_crt<W>(rh, ks, 0, 2 * now, true)) != ALCH_OK) return rc;
        if (dec_c0 && (rc = do_columns<W>(rh, GEN_LINV, ks, 0, now, 2)) != ALCH_OK) return rc;
        char* cur = ks;
        for (int u = 0; u < ddn; ++u) {
            char* nxt = (u + 1 == ddn) ? reinterpret_cast<char*>(out) + done * out_bytes : ((u & 1) ? pong : ping);
            const DevRing<W> rs = suffix(u);
            uint64_t inv[MAXL] = {0};
            Scal<W> sm;
            for (int j = 0; j < MAXL; ++j) sm.v[j] = 0;
            const int bits = 8 * (int)sizeof(W);
            for (int j = 1; j < rs.L; ++j) {
                const u64 qj = rh->q[u + j];
                inv[j] = h_powmod(rh->q[u] % qj, qj - 2, qj);
                sm.v[j] = (W)h_mulmod(inv[j], h_powmod(2, (u64)bits, qj), qj);
            }
            const size_t total = 2 * now * (size_t)(rs.L - 1) * rh->n;
            hipLaunchKernelGGL((k_rescale_drop0<W>), dim3(ew_grid(total)), dim3(256), 0, rh->stream, rs, (const W*)cur, (W*)nxt,
                               2 * now, sm);
            HIP_TRY(hipGetLastError());
            cur = nxt;
        }
        if (dec_c0 && (rc = do_columns<W>(rout, GEN_L, out, 2 * done, now, 2, rh->stream)) != ALCH_OK) return rc;
        if (!pow_out) {
            // crt on ring_out, queued on ring_h's stream (same device tables: ring_out is a suffix of ring_h)
            if ((rc = do_crt<W>(rout, out, 2 * done, 2 * now, false, nullptr, rh->stream)) != ALCH_OK) return rc;
        }
    }
    return ALCH_OK;
}

static bool is_suffix_ring(const alch_ring* small, const alch_ring* big) {
    if (small->m != big->m || small->word != big->word || small->L >= big->L) return false;
    for (int j = 0; j < small->L; ++j)
        if (small->q[j] != big->q[big->L - small->L + j]) return false;
    return true;
}

// PT2CT's mul_ with a BaseBGad 2 hint whose ring has at least as many limbs as the operands': composed from the entry points'
// own implementations.  modSwitch up is linear and the tensor product bilinear, so  modSwitch (a * b) = (up a) * (up b) / q_a  on the
// old limbs and 0 on the added ones: both operands are switched up (alch_ct_mod_switch's kernel), the factor q_a^-1 rides on the
// tensor product's scalar, then the BaseBGad key switch of alch_ct_mul_relin and the closing alch_ct_mod_switch.
// A hint on FEWER limbs than the operands (KSPNoise (BaseBGad 2) = p + KSAccumPNoise can sit one limb below the product's
// p + MulPNoise): mul_full_base2_down below.
template <typename W> static int do_mod_switch(alch_ring* rin, alch_ring* rout, const void* in, void* out, size_t batch, unsigned flags, int per = 2);

// keySwitchQuadCirc's second half for BaseBGad 2, from a c2 that is already on the powerful basis of the hint's ring:
// ks (c0, c1 on the CRT basis) += sum_d crt(digit_d(c2)) * hint_d.  The same kernels as do_mul_relin_unfused's base-2 branches.
template <typename W>
static int ks_base2_from_pow(alch_ring* r, const alch_hint* hint, void* ks, const void* c2pow, size_t batch) {
    Scal<u32> first, kd;
    const u32 D = (u32)base2_layout(r, first, kd);
    const size_t eb = elem_bytes(r);
    size_t chunk = std::min(batch, std::max<size_t>(1, (r->scratch_mib << 20) / ((size_t)D * eb)));
    int rc = ensure_ws(&r->ws_digits, &r->ws_digits_bytes, chunk * D * eb);
    if (rc != ALCH_OK) return rc;
    char* dig = reinterpret_cast<char*>(r->ws_digits);
    for (size_t done = 0; done < batch; done += chunk) {
        const size_t now = std::min(chunk, batch - done);
        const char* c2 = reinterpret_cast<const char*>(c2pow) + done * eb;
        W* po = reinterpret_cast<W*>(reinterpret_cast<char*>(ks) + done * 2 * eb);
        if (!split_ring(r) && !r->gen) {                             // digits computed in the transforms' loader
            NttCall<W> dc{};
            dc.op = OP_CRT_BASE2;
            dc.ring = &dev_ring<W>(r);
            dc.stream = r->stream;
            dc.src = reinterpret_cast<const W*>(c2);
            dc.data = reinterpret_cast<W*>(dig);
            dc.npoly = now * (size_t)D * (size_t)r->L;
            dc.b2_first = first; dc.b2_kd = kd; dc.b2_D = D;
            hipError_t e = dispatch(r->logn, dc);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("crt_base2 launch: ") + hipGetErrorString(e));
        } else {
            for (size_t y0 = 0; y0 < now; y0 += 32768) {             // grid.y is 16-bit
                const unsigned ny = (unsigned)std::min<size_t>(32768, now - y0);
                hipLaunchKernelGGL((k_decompose_base2<W>), dim3(ew_grid(elem_words(r)), ny), dim3(256), 0, r->stream, dev_ring<W>(r),
                                   reinterpret_cast<const W*>(c2 + y0 * eb), reinterpret_cast<W*>(dig + y0 * D * eb), first, kd, D);
                HIP_TRY(hipGetLastError());
            }
            if ((rc = do_crt<W>(r, dig, 0, now * D, false)) != ALCH_OK) return rc;
        }
        launch_hint_mac<W>(r, r->stream, po, (const W*)dig, (const W*)hint->dptr, now, D);
        HIP_TRY(hipGetLastError());
    }
    return ALCH_OK;
}

// mul_ under a BaseBGad 2 hint that lives on FEWER limbs than the operands (PT2CT.hs:140,164: the product sits at p + MulPNoise units
// rounded up to whole limbs, the hint at p + KSAccumPNoise): the leading modSwitch goes DOWN on the quadratic ciphertext -- c0 on the
// decoding basis, c1 and c2 on the powerful basis -- before the key switch.
template <typename W>
static int mul_full_base2_down(const alch_hint* hint, alch_ring* rin, alch_ring* rh, const void* a, const void* b, void* ks, size_t batch,
                               const uint64_t* s_pre, alch_buf* t, alch_buf* c2, alch_buf* c2h) {
    Scal<W> sr2;
    scal_to_mont<W>(rin, s_pre, 2, sr2);
    GTab<W> gt{};
    if (rin->gen) for (int j = 0; j < rin->L; ++j) gt.p[j] = rin->gh.rad > 1 ? gen_dev<W>(rin).gcrt[j] : nullptr;
    const size_t words = batch * elem_words(rin);
    // the tensor product and both rescales run on the operands' stream, the key switch on the hint's
    ALCH_LAUNCH_VW(k_tensor_ew, rin, words, rin->stream, dev_ring<W>(rin), (const W*)a, (const W*)b, (W*)t->dptr, (W*)c2->dptr, batch, sr2, 0,
                   (W*)nullptr, gt);
    HIP_TRY(hipGetLastError());
    int rc;
    if ((rc = do_mod_switch<W>(rin, rh, t->dptr, ks, batch, 0)) != ALCH_OK) return rc;
    if ((rc = do_mod_switch<W>(rin, rh, c2->dptr, c2h->dptr, batch, ALCH_POW_OUT, 1)) != ALCH_OK) return rc;
    HIP_TRY(hipStreamSynchronize(rin->stream));
    return ks_base2_from_pow<W>(rh, hint, ks, c2h->dptr, batch);
}
static int mul_full_base2(const alch_hint* hint, const alch_buf* a, const alch_buf* b, alch_buf* out, size_t batch, const uint64_t* s_pre,
                          unsigned flags) {
    alch_ring* rh = hint->ring;
    alch_ring* rin = a->ring;
    alch_ring* rout = out->ring;
    const bool in_down = rin->L > rh->L;                       // the leading modSwitch goes down, on the quadratic ciphertext
    auto same_ring = [](const alch_ring* x, const alch_ring* y) {           // another handle of the same (index, moduli)
        if (x->m != y->m || x->word != y->word || x->L != y->L) return false;
        for (int j = 0; j < x->L; ++j) if (x->q[j] != y->q[j]) return false;
        return true;
    };
    if (in_down ? !is_suffix_ring(rh, rin) : (!same_ring(rin, rh) && !is_suffix_ring(rin, rh)))
        return fail(ALCH_E_INVALID, "operand moduli must be the last limbs of the hint's ring, or the hint's the last limbs of the operands' (same word size)");
    if (in_down && !rin->has_crt) return fail(ALCH_E_NO_CRT, "the operands' ring has no CRT basis");
    if (!same_ring(rout, rh) && !is_suffix_ring(rout, rh)) return fail(ALCH_E_INVALID, "output moduli must be the last limbs of the hint's ring (same word size)");
    if (flags & ~(unsigned)ALCH_POW_OUT) return fail(ALCH_E_UNSUPPORTED, "only ALCH_POW_OUT is accepted");
    if (!rh->has_crt) return fail(ALCH_E_NO_CRT, "the hint's ring has no CRT basis");
    if (batch == 0) return ALCH_OK;
    if (!pairs_ok(batch, a->n_elems) || !pairs_ok(batch, b->n_elems) || !pairs_ok(batch, out->n_elems)) return fail(ALCH_E_INVALID, "buffers must hold 2*batch ring elements");
    BIND(rh);
    const int dup = in_down ? 0 : rh->L - rin->L;
    alch_buf *ua = nullptr, *ub = nullptr, *ks = nullptr, *c2h = nullptr;
    int rc = ALCH_OK;
    auto done = [&](int code) {
        if (ua) alch_buf_free(ua); if (ub) alch_buf_free(ub); if (c2h) alch_buf_free(c2h); if (ks && ks != out) alch_buf_free(ks);
        return code;
    };
    // everything below is queued on ring_h's stream, behind the work of the other two rings
    HIP_TRY(hipStreamSynchronize(rin->stream));
    HIP_TRY(hipStreamSynchronize(rout->stream));
    const alch_buf *pa = a, *pb = b;
    uint64_t s_eff[MAXL];
    for (int j = 0; j < rh->L; ++j) s_eff[j] = 1;
    for (int j = dup; j < rh->L; ++j) {
        u64 v = s_pre ? s_pre[j - dup] % rh->q[j] : 1;
        for (int u = 0; u < dup; ++u) v = h_mulmod(v, h_invmod(rh->q[u] % rh->q[j], rh->q[j]), rh->q[j]);
        s_eff[j] = v;
    }
    if (dup > 0) {
        if ((rc = alch_buf_alloc(rh, 2 * batch, &ua)) != ALCH_OK || (rc = alch_buf_alloc(rh, 2 * batch, &ub)) != ALCH_OK) return done(rc);
        // alch_ct_mod_switch up works on rout's stream = rh's here
        rc = rh->word == 4 ? do_mod_switch<u32>(rin, rh, a->dptr, ua->dptr, batch, 0) : do_mod_switch<u64>(rin, rh, a->dptr, ua->dptr, batch, 0);
        if (rc == ALCH_OK) rc = rh->word == 4 ? do_mod_switch<u32>(rin, rh, b->dptr, ub->dptr, batch, 0) : do_mod_switch<u64>(rin, rh, b->dptr, ub->dptr, batch, 0);
        if (rc != ALCH_OK) return done(rc);
        pa = ua; pb = ub;
    }
    const bool down = rout->L < rh->L;
    if (down || (flags & ALCH_POW_OUT)) { if ((rc = alch_buf_alloc(rh, 2 * batch, &ks)) != ALCH_OK) return done(rc); }
    else ks = out;
    if (in_down) {
        // ua: the (c0, c1) pairs, ub: c2, both on the operands' ring; c2h: c2 on the hint's ring, powerful basis
        if ((rc = alch_buf_alloc(rin, 2 * batch, &ua)) != ALCH_OK || (rc = alch_buf_alloc(rin, batch, &ub)) != ALCH_OK ||
            (rc = alch_buf_alloc(rh, batch, &c2h)) != ALCH_OK) return done(rc);
        HIP_TRY(hipStreamSynchronize(rh->stream));
        rc = rh->word == 4 ? mul_full_base2_down<u32>(hint, rin, rh, a->dptr, b->dptr, ks->dptr, batch, s_pre, ua, ub, c2h)
                           : mul_full_base2_down<u64>(hint, rin, rh, a->dptr, b->dptr, ks->dptr, batch, s_pre, ua, ub, c2h);
    } else
    rc = rh->word == 4 ? do_mul_relin_unfused<u32>(rh, hint, pa->dptr, pb->dptr, ks->dptr, batch, s_eff)
                       : do_mul_relin_unfused<u64>(rh, hint, pa->dptr, pb->dptr, ks->dptr, batch, s_eff);
    if (rc != ALCH_OK) return done(rc);
    if (down) {
        rc = rh->word == 4 ? do_mod_switch<u32>(rh, rout, ks->dptr, out->dptr, batch, flags & ALCH_POW_OUT)
                           : do_mod_switch<u64>(rh, rout, ks->dptr, out->dptr, batch, flags & ALCH_POW_OUT);
    } else if (flags & ALCH_POW_OUT) {
        HIP_TRY(hipMemcpyAsync(out->dptr, ks->dptr, 2 * batch * elem_bytes(rh), hipMemcpyDeviceToDevice, rh->stream));
        rc = rh->word == 4 ? do_crt<u32>(rh, out->dptr, 0, 2 * batch, true) : do_crt<u64>(rh, out->dptr, 0, 2 * batch, true);
    }
    if (rc != ALCH_OK) return done(rc);
    HIP_TRY(hipStreamSynchronize(rh->stream));                 // the scratch buffers are freed on return
    return done(ALCH_OK);
}

extern "C" int alch_ct_mul_full(const alch_hint* hint, const alch_buf* a, const alch_buf* b, alch_buf* out, size_t batch,
                                const uint64_t* s_pre, unsigned flags) try {
    if (!hint || !a || !b || !out) return fail(ALCH_E_INVALID, "null argument");
    alch_ring* rh = hint->ring;
    alch_ring* rin = a->ring;
    alch_ring* rout = out->ring;
    if (b->ring != rin) return fail(ALCH_E_INVALID, "operands belong to different rings");
    if (hint->gadget != ALCH_GAD_TRIV) return mul_full_base2(hint, a, b, out, batch, s_pre, flags);
    if (!is_suffix_ring(rin, rh)) return fail(ALCH_E_INVALID, "operand moduli must be the last limbs of the hint's ring (same word size)");
    if (!is_suffix_ring(rout, rh)) return fail(ALCH_E_INVALID, "output moduli must be the last limbs of the hint's ring (same word size)");
    if (rh->L - rout->L > MAXDROP) return fail(ALCH_E_UNSUPPORTED, "at most 3 limbs dropped per call");
    if (flags & ~(unsigned)ALCH_POW_OUT) return fail(ALCH_E_UNSUPPORTED, "only ALCH_POW_OUT is accepted");
    if (batch == 0) return ALCH_OK;
    BIND(rh);
    if (!pairs_ok(batch, a->n_elems) || !pairs_ok(batch, b->n_elems) || !pairs_ok(batch, out->n_elems))
        return fail(ALCH_E_INVALID, "buffers must hold 2*batch ring elements");
    if (!rh->ev_x) HIP_TRY(hipEventCreateWithFlags(&rh->ev_x, hipEventDisableTiming));
    if (rin->stream != rh->stream) {
        HIP_TRY(hipEventRecord(rh->ev_x, rin->stream));
        HIP_TRY(hipStreamWaitEvent(rh->stream, rh->ev_x, 0));
    }
    if (rout->stream != rh->stream && rout->stream != rin->stream) {
        HIP_TRY(hipEventRecord(rh->ev_x, rout->stream));
        HIP_TRY(hipStreamWaitEvent(rh->stream, rh->ev_x, 0));
    }
    const bool pow_out = (flags & ALCH_POW_OUT) != 0;
    int rc;
    if (!rh->has_crt) return fail(ALCH_E_NO_CRT, "the hint's ring has no CRT basis");
    if (split_ring(rh) || rh->gen)
        rc = rh->word == 4 ? do_mul_full_unfused<u32>(rh, rin, rout, hint, a->dptr, b->dptr, out->dptr, batch, s_pre, pow_out)
                           : do_mul_full_unfused<u64>(rh, rin, rout, hint, a->dptr, b->dptr, out->dptr, batch, s_pre, pow_out);
    else
        rc = rh->word == 4 ? do_mul_full<u32>(rh, rin, rout, hint, a->dptr, b->dptr, out->dptr, batch, s_pre, pow_out)
                           : do_mul_full<u64>(rh, rin, rout, hint, a->dptr, b->dptr, out->dptr, batch, s_pre, pow_out);
    if (rc != ALCH_OK) return rc;
    HIP_TRY(hipEventRecord(rh->ev_x, rh->stream));
    if (rin->stream != rh->stream) HIP_TRY(hipStreamWaitEvent(rin->stream, rh->ev_x, 0));
    if (rout->stream != rh->stream) HIP_TRY(hipStreamWaitEvent(rout->stream, rh->ev_x, 0));
    return ALCH_OK;
} catch (...) { return abi_catch(); }

// ------------------------------------------------------------------------------------------------------
// ring tunnelling: tunnel_ hint between two modSwitch_ (PT2CT.hs:224-229, Eval.hs:134)
// ------------------------------------------------------------------------------------------------------
extern "C" int alch_tunnel_info(const alch_ring* rr, const alch_ring* rs, uint32_t* e_prime, uint32_t* d_rel) try {
    if (!rr || !rs) return fail(ALCH_E_INVALID, "null ring");
    u32 a = rr->m, b = rs->m;
    while (b) { const u32 t = a % b; a = b; b = t; }
    GenHost ge, gr, gs;
    if (!gen_plan(a, ge) || !gen_plan(rr->m, gr) || !gen_plan(rs->m, gs)) return fail(ALCH_E_UNSUPPORTED, "index not served");
    u32 d = 0, mask = 0;
    std::vector<int32_t> tab;
    if (!gen_tunnel_table(ge, gr, gs, d, tab, mask)) return fail(ALCH_E_INVALID, "indices do not form a tunnel");
    if (e_prime) *e_prime = a;
    if (d_rel) *d_rel = d;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

template <typename W>
static int tunnel_to_mont(alch_ring* rs, void* dst, const void* src, size_t elems) {
    Scal<W> sm;
    scal_to_mont<W>(rs, nullptr, 2, sm);
    const size_t words = elems * elem_words(rs);
    ALCH_LAUNCH_VW(k_scale, rs, words, rs->stream, dev_ring<W>(rs), (W*)dst, (const W*)src, words, sm);
    HIP_TRY(hipGetLastError());
    return ALCH_OK;
}

extern "C" int alch_tunnel_create(alch_ring* rr, alch_ring* rs, int gadget, const alch_buf* lin_crt, const alch_buf* ks_crt, alch_tunnel** out) try {
    if (!rr || !rs || !lin_crt || !ks_crt || !out) return fail(ALCH_E_INVALID, "null argument");
    *out = nullptr;
    if (gadget != ALCH_GAD_TRIV && gadget != ALCH_GAD_BASE2) return fail(ALCH_E_INVALID, "unknown gadget");
    if (!rr->gen || !rs->gen || !rr->has_crt || !rs->has_crt) return fail(ALCH_E_UNSUPPORTED, "tunnelling runs on general-index rings with a CRT basis");
    if (rr->L != rs->L || rr->word != rs->word) return fail(ALCH_E_INVALID, "both rings must have the same moduli");
    for (int j = 0; j < rr->L; ++j) if (rr->q[j] != rs->q[j]) return fail(ALCH_E_INVALID, "both rings must have the same moduli");
    if (lin_crt->ring != rs || ks_crt->ring != rs) return fail(ALCH_E_INVALID, "the linear function and the hints live in the target ring");
    u32 ep = 0, d_rel = 0;
    int rc = alch_tunnel_info(rr, rs, &ep, &d_rel);
    if (rc != ALCH_OK) return rc;
    GenHost ge;
    gen_plan(ep, ge);
    std::vector<int32_t> tab, tab_e;
    std::vector<u32> slot_e;
    u32 mask = 0;
    if (!gen_tunnel_table(ge, rr->gh, rs->gh, d_rel, tab, mask, &tab_e, &slot_e)) return fail(ALCH_E_INVALID, "indices do not form a tunnel");
    const int digits = gadget_digits(rs, gadget);
    const size_t nlin = d_rel, nks = (size_t)d_rel * (size_t)digits * 2;
    if (lin_crt->n_elems < nlin || ks_crt->n_elems < nks)
        return fail(ALCH_E_INVALID, "need d_rel linear-function values and 2 * d_rel * (gadget digits) hint elements");
    BIND(rs);
    alch_tunnel* t = new alch_tunnel{rr, rs, d_rel, mask, nullptr, nullptr, nullptr, gadget, digits};
    if (hipMalloc((void**)&t->table, tab.size() * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(&t->lin, nlin * elem_bytes(rs)) != hipSuccess || hipMalloc(&t->ks, nks * elem_bytes(rs)) != hipSuccess) {
        alch_tunnel_free(t);
        return fail(ALCH_E_NOMEM, "hipMalloc(tunnel) failed");
    }
    if (hipMemcpy(t->table, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { alch_tunnel_free(t); return fail(ALCH_E_HIP, "tunnel table upload failed"); }
    if (ep != rs->m && rs->opts.tunnel_ep) {           // E' smaller than S': run the transforms there when E' has a general-index ring
        alch_ring* re = nullptr;
        if (alch_ring_create(ep, rs->L, rs->q, &re) == ALCH_OK && re->gen && re->word == rs->word && re->n % 4 == 0 && rs->n % 4 == 0) {
            t->re = re;
            re->g32.nt = rs->g32.nt; re->g64.nt = rs->g64.nt;      // launch-structure options of the target ring apply to its E' ring
            t->pieces_ok = true;
            for (size_t k = 0; k + 3 < slot_e.size(); k += 4)
                if (slot_e[k] % 4 || slot_e[k + 1] != slot_e[k] + 1 || slot_e[k + 2] != slot_e[k] + 2 || slot_e[k + 3] != slot_e[k] + 3) { t->pieces_ok = false; break; }
            if (hipMalloc((void**)&t->table_e, tab_e.size() * sizeof(int32_t)) != hipSuccess ||
                hipMalloc((void**)&t->slot_e, slot_e.size() * sizeof(u32)) != hipSuccess ||
                hipMemcpy(t->table_e, tab_e.data(), tab_e.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(t->slot_e, slot_e.data(), slot_e.size() * sizeof(u32), hipMemcpyHostToDevice) != hipSuccess) {
                alch_tunnel_free(t);
                return fail(ALCH_E_NOMEM, "hipMalloc(tunnel, E' tables) failed");
            }
        } else if (re) {
            alch_ring_destroy(re);
        }
        BIND(rs);
    }
    rc = rs->word == 4 ? tunnel_to_mont<u32>(rs, t->lin, lin_crt->dptr, nlin) : tunnel_to_mont<u64>(rs, t->lin, lin_crt->dptr, nlin);
    if (rc == ALCH_OK) rc = rs->word == 4 ? tunnel_to_mont<u32>(rs, t->ks, ks_crt->dptr, nks) : tunnel_to_mont<u64>(rs, t->ks, ks_crt->dptr, nks);
    if (rc != ALCH_OK) { alch_tunnel_free(t); return rc; }
    if (hipStreamSynchronize(rs->stream) != hipSuccess) { alch_tunnel_free(t); return fail(ALCH_E_HIP, "tunnel setup failed"); }
    *out = t;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_tunnel_free(alch_tunnel* t) try {
    if (!t) return ALCH_OK;
    (void)hipSetDevice(t->rs->device);
    (void)hipStreamSynchronize(t->rs->stream);
    if (t->table) (void)hipFree(t->table);
    if (t->lin) (void)hipFree(t->lin);
    if (t->ks) (void)hipFree(t->ks);
    if (t->table_e) (void)hipFree(t->table_e);
    if (t->slot_e) (void)hipFree(t->slot_e);
    if (t->re) alch_ring_destroy(t->re);
    delete t;
    return ALCH_OK;
} catch (...) { return abi_catch(); }

// View of the last L - u limbs of a general-index ring (device tables shared with the ring itself).
template <typename W>
static void gen_suffix_view(alch_ring* r, int u, DevRing<W>& d, GenDev<W>& g) {
    d = dev_ring<W>(r);
    g = gen_dev<W>(r);
    d.L = r->L - u;
    for (int j = 0; j + u < r->L; ++j) {
        d.mod[j] = d.mod[j + u];
        g.tabf[j] = g.tabf[j + u]; g.tabi[j] = g.tabi[j + u]; g.iscale_m[j] = g.iscale_m[j + u];
        g.gcrt[j] = g.gcrt[j + u]; g.gcrt_inv[j] = g.gcrt_inv[j + u]; g.radinv_m[j] = g.radinv_m[j + u];
    }
}

// SymmSHE.tunnel on a batch of linear ciphertexts (k = 0):  out = (f'(c0), 0) + sum_i switch(hint_i, embed(c1_i)).
// rin: the ring the ciphertexts live in -- the tunnel's R' ring or its last limbs (PT2CT emits modSwitch_ .: tunnel_ hint .: modSwitch_,
// PT2CT.hs:224-229; the leading modSwitch up, x -> (0, q_a x), is then folded in: the added limbs are zero, so neither their
// crtInv, nor the crt of their embedded constant terms, nor their digits' transforms and hint products are computed).
template <typename W>
static int do_tunnel(const alch_tunnel* t, alch_ring* rin, const void* in, void* out, size_t batch, const uint64_t* s_pre, unsigned flags) {
    alch_ring* rr = t->rr;
    alch_ring* rs = t->rs;
    const int L = rs->L, dup = rr->L - rin->L;
    const u32 D = t->d_rel;
    const bool base2 = t->gadget == ALCH_GAD_BASE2;
    // compact: embedded coefficients and digits hold the L - dup non-zero limbs only (TrivGad; BaseBGad 2 keeps the
    // full layout with explicit zero limbs)
    const bool compact = dup > 0 && !base2;
    const u32 Lx = compact ? (u32)(L - dup) : (u32)L, xoff = compact ? (u32)dup : 0u;
    const u32 GD = base2 ? (u32)t->digits : Lx;     // gadget digits per embedded coefficient
    Scal<u32> b2first, b2kd;
    if (base2) base2_layout(rs, b2first, b2kd);
    // rx: the ring the E'-coefficients are transformed in -- E' itself when the tunnel has its ring (the CRT over S' of an embedded
    // element is its CRT over E' replicated: k_tunnel_lin / k_hint_mac read it through slot_e), else S' with the coefficients embedded
    alch_ring* rx = t->re ? t->re : rs;
    const u32* slot_e = t->re ? t->slot_e : nullptr;
    const size_t ebr = elem_bytes(rin), ebs = elem_bytes(rs), ebd = elem_bytes(rx), ebx = ebd / (size_t)L * Lx;
    // TrivGad, 32-bit words, E'-level transforms with piece-aligned slot table: digit transforms + hint products fused (k_gen_tunnel_ks)
    const bool fused = sizeof(W) == 4 && !base2 && t->re && t->pieces_ok && rs->opts.tunnel_fused && rs->n % 4 == 0 && rx->n % 4 == 0 &&
                       rs->n <= (u32)(GEN_TUN_T * 4 * 6) && 2 * (size_t)rx->n * sizeof(W) <= 65536;
    const int tun_ng = (rs->opts.tunnel_fused == 4 && 4 * (size_t)rx->n * sizeof(W) <= 65536) ? 4 : 2;     // two workgroups per CU: 64 KiB of LDS each
    // scratch per ciphertext: Pow copy of the input (2 R'-elements), x0, x1 (D coefficients each), digits (D * GD elements of rx)
    const size_t per_ct = 2 * ebr + 2 * (size_t)D * ebx + (fused ? 0 : (size_t)D * GD * ebd);
    size_t chunk = std::max<size_t>(1, (rs->scratch_mib << 20) / per_ct);
    chunk = std::min(chunk, batch);
    int rc = ensure_ws(&rs->ws_full, &rs->ws_full_bytes, chunk * per_ct);
    if (rc != ALCH_OK) return rc;
    char* win = reinterpret_cast<char*>(rs->ws_full);
    char* x0 = win + chunk * 2 * ebr;
    char* x1 = x0 + chunk * D * ebx;
    char* dig = x1 + chunk * D * ebx;
    uint64_t s_eff[MAXL] = {0};
    for (int j = dup; j < L; ++j) {
        u64 v = s_pre ? s_pre[j] % rs->q[j] : 1;
        for (int u = 0; u < dup; ++u) v = h_mulmod(v, rs->q[u] % rs->q[j], rs->q[j]);      // modSwitch up: times the added moduli
        s_eff[j] = v;
    }
    const bool scale = s_pre != nullptr || dup > 0;
    Scal<W> sm;
    scal_to_mont<W>(rs, s_eff, 1, sm);
    const bool dec_c0 = rr->gh.rad > 1 && t->linv_skip_mask != ((1u << rr->gh.nfact) - 1);
    for (size_t done = 0; done < batch; done += chunk) {
        const size_t now = std::min(chunk, batch - done);
        const char* src = reinterpret_cast<const char*>(in) + done * 2 * ebr;
        // Pow basis of R' (a copy: the caller's ciphertexts are left alone); c0 onto relative-Dec (x) Pow(E')
        // Pow-basis input is read in place (c1 always; c0 through an out-of-place lInv into the scratch when the tunnel needs one)
        const bool pin = (flags & ALCH_POW_IN) != 0;
        if (!pin && (rc = do_crt<W>(rin, win, 0, 2 * now, true, src, rs->stream)) != ALCH_OK) return rc;
        if (dec_c0) {
            GenCall<W> g{};
            g.op = GEN_LINV; g.ring = &dev_ring<W>(rin); g.gen = &gen_dev<W>(rin); g.stream = rs->stream;
            g.data = reinterpret_cast<W*>(win); g.src = pin ? reinterpret_cast<const W*>(src) : nullptr;
            g.elem_stride = 2; g.first_poly = 0; g.npoly = now * (size_t)rin->L;
            g.skip_mask = t->linv_skip_mask; g.fail_flag = rin->d_flag;
            hipError_t e = gen_dispatch(g);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("tunnel lInv launch: ") + hipGetErrorString(e));
        }
        const W* in0 = (pin && !dec_c0) ? reinterpret_cast<const W*>(src) : reinterpret_cast<const W*>(win);
        const W* in1 = pin ? reinterpret_cast<const W*>(src) : reinterpret_cast<const W*>(win);
        const size_t gw = now * 2 * (size_t)D * Lx * rx->n;
        ALCH_LAUNCH_VW(k_tunnel_gather, rx, gw, rs->stream, dev_ring<W>(rx), in0, in1, (W*)x0, (W*)x1,
                           t->re ? t->table_e : t->table, D, rr->n, now, sm, scale ? 1 : 0, Lx, xoff, (u32)dup);
        HIP_TRY(hipGetLastError());
        // constant term: evalLin
        if (compact) {
            DevRing<W> dv; GenDev<W> gv;
            gen_suffix_view<W>(rx, dup, dv, gv);
            GenCall<W> g{};
            g.op = GEN_CRT; g.ring = &dv; g.gen = &gv; g.stream = rs->stream;
            g.data = reinterpret_cast<W*>(x0); g.first_poly = 0; g.npoly = now * (size_t)D * Lx;
            hipError_t e = gen_dispatch(g);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("tunnel crt launch: ") + hipGetErrorString(e));
        } else if ((rc = do_crt<W>(rx, x0, 0, now * D, false, nullptr, rs->stream)) != ALCH_OK) return rc;
        W* po = reinterpret_cast<W*>(reinterpret_cast<char*>(out) + done * 2 * ebs);
        // k_tunnel_mac_e starts its accumulators from evalLin's constant term itself: no pass for it then
        const bool mac_e = sizeof(W) == 4 && !fused && (slot_e ? t->pieces_ok : true) && rs->n % 4 == 0 && rx->n % 4 == 0 && rs->opts.tunnel_mac;
        if (!mac_e) {
            ALCH_LAUNCH_VW(k_tunnel_lin, rs, now * elem_words(rs), rs->stream, dev_ring<W>(rs), po, (const W*)x0,
                           (const W*)t->lin, D, now, Lx, xoff, slot_e, rx->n);
            HIP_TRY(hipGetLastError());
        }
        // linear term: decompose + reduce + crt of every embedded coefficient, inner product with the hints
        if (fused) {                                  // one kernel, no digits in HBM (k_gen_tunnel_ks)
            if constexpr (sizeof(W) == 4) {
                GenTunArgs<u32> A{};
                A.x1 = reinterpret_cast<const u32*>(x1); A.hint = reinterpret_cast<const u32*>(t->ks); A.out = reinterpret_cast<u32*>(po);
                A.slot_e = slot_e; A.D = D; A.Lx = Lx; A.xoff = xoff; A.n_s = rs->n; A.balanced = rs->balanced ? 1 : 0;
                hipError_t e = gen_tunnel_ks_dispatch(rs->d32, rx->g32, A, now, tun_ng, rs->stream);
                if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("fused tunnel key switch launch: ") + hipGetErrorString(e));
            }
            continue;
        }
        if (base2) {                                 // BaseBGad 2 (examples/Tunnel.hs:24): decompose + reduce in the transforms' loader
            GenCall<W> g{};
            g.op = GEN_CRT_BASE2; g.ring = &dev_ring<W>(rx); g.gen = &gen_dev<W>(rx); g.stream = rs->stream;
            g.src = reinterpret_cast<const W*>(x1); g.data = reinterpret_cast<W*>(dig);
            g.npoly = now * (size_t)D * (size_t)GD * (size_t)L; g.b2_first = b2first; g.b2_kd = b2kd; g.b2_D = GD;
            hipError_t e = gen_dispatch(g);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("tunnel base2 crt_digits launch: ") + hipGetErrorString(e));
        } else {
            GenCall<W> g{};
            g.op = GEN_CRT_DIGITS; g.ring = &dev_ring<W>(rx); g.gen = &gen_dev<W>(rx); g.stream = rs->stream;
            g.src = reinterpret_cast<const W*>(x1); g.data = reinterpret_cast<W*>(dig);
            g.npoly = now * (size_t)D * (size_t)Lx * (size_t)L; g.balanced = rs->balanced; g.with_diag = true;
            g.src_limbs = (int)Lx; g.src_first = (int)xoff;
            hipError_t e = gen_dispatch(g);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("tunnel crt_digits launch: ") + hipGetErrorString(e));
        }
        if (mac_e) {
            if constexpr (sizeof(W) == 4) {
                const size_t pieces = (now + 3) / 4 * (size_t)rs->L * (rs->n / 4);
                hipLaunchKernelGGL((k_tunnel_mac_e<4>), dim3(ew_grid(pieces)), dim3(256), 0, rs->stream, rs->d32, (u32*)po, (const u32*)dig, (const u32*)t->ks,
                                   now, D * GD, Lx, compact ? (u32)dup : 0u, slot_e, rx->n, (const u32*)x0, (const u32*)t->lin, D, Lx, xoff);
            }
        } else {
            launch_hint_mac<W>(rs, rs->stream, po, (const W*)dig, (const W*)t->ks, now, D * GD, (const W*)nullptr, Lx, compact ? (u32)dup : 0u,
                               slot_e, rx->n, t->pieces_ok);
        }
        HIP_TRY(hipGetLastError());
    }
    return ALCH_OK;
}

extern "C" int alch_ct_tunnel(const alch_tunnel* t, const alch_buf* in, alch_buf* out, size_t batch, const uint64_t* s_pre, unsigned flags) try {
    if (!t || !in || !out) return fail(ALCH_E_INVALID, "null argument");
    if ((in->ring != t->rr && !is_suffix_ring(in->ring, t->rr)) || out->ring != t->rs)
        return fail(ALCH_E_INVALID, "input / output buffers must belong to the tunnel's rings (the input may live on the last limbs of the R' ring)");
    if (flags & ~(unsigned)(ALCH_POW_IN | ALCH_POW_OUT)) return fail(ALCH_E_INVALID, "unknown flag");
    if (batch == 0) return ALCH_OK;
    if (!pairs_ok(batch, in->n_elems) || !pairs_ok(batch, out->n_elems)) return fail(ALCH_E_INVALID, "buffers must hold 2*batch ring elements");
    alch_ring* rs = t->rs;
    alch_ring* rr = t->rr;
    BIND(rs);
    if (!rs->ev_x) HIP_TRY(hipEventCreateWithFlags(&rs->ev_x, hipEventDisableTiming));
    if (rr->stream != rs->stream) {                       // everything runs on the target ring's stream
        HIP_TRY(hipEventRecord(rs->ev_x, rr->stream));
        HIP_TRY(hipStreamWaitEvent(rs->stream, rs->ev_x, 0));
    }
    alch_ring* rin = in->ring;
    if (rin->stream != rs->stream && rin != rr) {
        HIP_TRY(hipEventRecord(rs->ev_x, rin->stream));
        HIP_TRY(hipStreamWaitEvent(rs->stream, rs->ev_x, 0));
    }
    int rc = rs->word == 4 ? do_tunnel<u32>(t, rin, in->dptr, out->dptr, batch, s_pre, flags)
                           : do_tunnel<u64>(t, rin, in->dptr, out->dptr, batch, s_pre, flags);
    if (rc != ALCH_OK) return rc;
    if (flags & ALCH_POW_OUT) if ((rc = buf_crt(out, 0, 2 * batch, true)) != ALCH_OK) return rc;
    if (rr->stream != rs->stream) {
        HIP_TRY(hipEventRecord(rs->ev_x, rs->stream));
        HIP_TRY(hipStreamWaitEvent(rr->stream, rs->ev_x, 0));
    }
    if (rin != rr && rin->stream != rs->stream) {
        HIP_TRY(hipEventRecord(rs->ev_x, rs->stream));
        HIP_TRY(hipStreamWaitEvent(rin->stream, rs->ev_x, 0));
    }
    return ALCH_OK;
} catch (...) { return abi_catch(); }

// ------------------------------------------------------------------------------------------------------
// SymmSHE modSwitch on batches of linear ciphertexts (Eval.hs:130; PT2CT.hs:177,224-229)
// ------------------------------------------------------------------------------------------------------
// per = 2: linear ciphertexts (c0 rescaled on the decoding basis, c1 on the powerful basis); per = 1: `batch` single ring elements, all on
// the powerful basis (the c2 of a quadratic ciphertext: SymmSHE's modSwitch rescales every coefficient above c0 with rescalePow)
template <typename W>
static int do_mod_switch(alch_ring* rin, alch_ring* rout, const void* in, void* out, size_t batch, unsigned flags, int per) {
    const size_t n = rin->n;
    const size_t P = (size_t)per;
    if (rout->L > rin->L) {                                   // up: Rescale b -> (a, b), any basis
        const int dup = rout->L - rin->L;
        uint64_t mult[MAXL] = {0};
        for (int j = dup; j < rout->L; ++j) {
            u64 v = 1;
            for (int u = 0; u < dup; ++u) v = h_mulmod(v, rout->q[u] % rout->q[j], rout->q[j]);
            mult[j] = v;
        }
        Scal<W> sm;
        scal_to_mont<W>(rout, mult, 1, sm);
        const size_t total = P * batch * elem_words(rout);
        hipLaunchKernelGGL((k_rescale_up<W>), dim3(ew_grid(total)), dim3(256), 0, rout->stream, dev_ring<W>(rout), (const W*)in, (W*)out,
                           P * batch, dup, sm);
        HIP_TRY(hipGetLastError());
        return ALCH_OK;
    }
    // down: Pow basis (c0 on the Dec basis for a general index), one limb at a time, outermost first
    const int L = rin->L, ddn = L - rout->L;
    const size_t eb = elem_bytes(rin);
    const bool dec_c0 = per == 2 && rin->gen && rin->gh.rad > 1;
    if (rin->gen && rin->opts.rs_lin && ddn <= MAXDROP && !(flags & ALCH_POW_IN)) {
        // kept limbs stay in the CRT basis (k_gen_rescale_drop / k_gen_rescale_keep)
        DropTab<W> dt;
        fill_drop_tab<W>(rin, ddn, dt);
        const size_t per_b = P * (size_t)ddn * n * sizeof(W);
        size_t chunk = std::min(batch, std::max<size_t>(1, (rin->scratch_mib << 20) / per_b));
        int rc = ensure_ws(&rin->ws_full, &rin->ws_full_bytes, chunk * per_b);
        if (rc != ALCH_OK) return rc;
        for (size_t done = 0; done < batch; done += chunk) {
            const size_t now = std::min(chunk, batch - done);
            hipError_t e = gen_rescale_lin_dispatch(dev_ring<W>(rin), gen_dev<W>(rin),
                                                    reinterpret_cast<const W*>(reinterpret_cast<const char*>(in) + done * P * eb),
                                                    reinterpret_cast<W*>(rin->ws_full),
                                                    reinterpret_cast<W*>(reinterpret_cast<char*>(out) + done * P * elem_bytes(rout)), dt,
                                                    dec_c0 ? 1 : 0, P * now, rin->stream, (flags & ALCH_POW_OUT) != 0);
            if (e != hipSuccess) return fail(ALCH_E_HIP, std::string("rescale launch: ") + hipGetErrorString(e));
        }
        return ALCH_OK;
    }
    const size_t per_ct = 3 * P * eb;                          // Pow copy + ping + pong
    size_t chunk = std::max<size_t>(1, (rin->scratch_mib << 20) / per_ct);
    chunk = std::min(chunk, batch);
    int rc = ensure_ws(&rin->ws_full, &rin->ws_full_bytes, chunk * per_ct);
    if (rc != ALCH_OK) return rc;
    char* cur0 = reinterpret_cast<char*>(rin->ws_full);
    char* ping = cur0 + chunk * P * eb;
    char* pong = ping + chunk * P * eb;
    auto suffix = [&](int u) {
        DevRing<W> d = dev_ring<W>(rin);
        d.L = L - u;
        for (int j = 0; j + u < L; ++j) d.mod[j] = d.mod[j + u];
        return d;
    };
    const size_t in_bytes = P * eb, out_bytes = P * elem_bytes(rout);
    for (size_t done = 0; done < batch; done += chunk) {
        const size_t now = std::min(chunk, batch - done);
        const char* src = reinterpret_cast<const char*>(in) + done * in_bytes;
        if (flags & ALCH_POW_IN) HIP_TRY(hipMemcpyAsync(cur0, src, now * in_bytes, hipMemcpyDeviceToDevice, rin->stream));
        else if (rin->gen || !split_ring(rin)) { if ((rc = do_crt<W>(rin, cur0, 0, P * now, true, src)) != ALCH_OK) return rc; }
        else {                                                 // split transforms work in place
            HIP_TRY(hipMemcpyAsync(cur0, src, now * in_bytes, hipMemcpyDeviceToDevice, rin->stream));
            if ((rc = do_crt<W>(rin, cur0, 0, P * now, true)) != ALCH_OK) return rc;
        }
        if (dec_c0 && (rc = do_columns<W>(rin, GEN_LINV, cur0, 0, now, 2)) != ALCH_OK) return rc;
        char* cur = cur0;
        for (int u = 0; u < ddn; ++u) {
            char* nxt = (u + 1 == ddn) ? reinterpret_cast<char*>(out) + done * out_bytes : ((u & 1) ? pong : ping);
            const DevRing<W> rs = suffix(u);
            Scal<W> sm;
            for (int j = 0; j < MAXL; ++j) sm.v[j] = 0;
            const int bits = 8 * (int)sizeof(W);
            for (int j = 1; j < rs.L; ++j) {
                const u64 qj = rin->q[u + j];
                sm.v[j] = (W)h_mulmod(h_powmod(rin->q[u] % qj, qj - 2, qj), h_powmod(2, (u64)bits, qj), qj);
            }
            const size_t total = P * now * (size_t)(rs.L - 1) * n;
            hipLaunchKernelGGL((k_rescale_drop0<W>), dim3(ew_grid(total)), dim3(256), 0, rin->stream, rs, (const W*)cur, (W*)nxt, P * now, sm);
            HIP_TRY(hipGetLastError());
            cur = nxt;
        }
        if (dec_c0 && (rc = do_columns<W>(rout, GEN_L, out, 2 * done, now, 2, rin->stream)) != ALCH_OK) return rc;
        if (!(flags & ALCH_POW_OUT) && (rc = do_crt<W>(rout, out, P * done, P * now, false, nullptr, rin->stream)) != ALCH_OK) return rc;
    }
    return ALCH_OK;
}

extern "C" int alch_ct_mod_switch(const alch_buf* in, alch_buf* out, size_t batch, unsigned flags) try {
    if (!in || !out) return fail(ALCH_E_INVALID, "null buffer");
    alch_ring* rin = in->ring;
    alch_ring* rout = out->ring;
    if (flags & ~(unsigned)(ALCH_POW_IN | ALCH_POW_OUT)) return fail(ALCH_E_INVALID, "unknown flag");
    if (!rin->has_crt || !rout->has_crt) return fail(ALCH_E_NO_CRT, "modSwitch runs on rings with a CRT basis");
    const bool up = rout->L > rin->L;
    if (rin->L == rout->L || !(up ? is_suffix_ring(rin, rout) : is_suffix_ring(rout, rin)))
        return fail(ALCH_E_INVALID, "the smaller ring's moduli must be the last limbs of the bigger ring's (same index and word size)");
    if (batch == 0) return ALCH_OK;
    if (!pairs_ok(batch, in->n_elems) || !pairs_ok(batch, out->n_elems)) return fail(ALCH_E_INVALID, "buffers must hold 2*batch ring elements");
    alch_ring* rw = up ? rout : rin;                          // the ring whose stream carries the work
    alch_ring* ro = up ? rin : rout;
    BIND(rw);
    if (!rw->ev_x) HIP_TRY(hipEventCreateWithFlags(&rw->ev_x, hipEventDisableTiming));
    if (ro->stream != rw->stream) { HIP_TRY(hipEventRecord(rw->ev_x, ro->stream)); HIP_TRY(hipStreamWaitEvent(rw->stream, rw->ev_x, 0)); }
    int rc = rin->word == 4 ? do_mod_switch<u32>(rin, rout, in->dptr, out->dptr, batch, flags)
                            : do_mod_switch<u64>(rin, rout, in->dptr, out->dptr, batch, flags);
    if (rc != ALCH_OK) return rc;
    if (ro->stream != rw->stream) { HIP_TRY(hipEventRecord(rw->ev_x, rw->stream)); HIP_TRY(hipStreamWaitEvent(ro->stream, rw->ev_x, 0)); }
    return ALCH_OK;
} catch (...) { return abi_catch(); }

// ------------------------------------------------------------------------------------------------------
// modSwitch building block
// ------------------------------------------------------------------------------------------------------
extern "C" int alch_buf_rescale_drop0(const alch_buf* src, alch_buf* dst, size_t count) try {
    if (!src || !dst) return fail(ALCH_E_INVALID, "null buffer");
    alch_ring* rs = src->ring;
    alch_ring* rd = dst->ring;
    BIND(rs);
    if (rs->L < 2 || rd->L != rs->L - 1 || rd->n != rs->n || rd->word != rs->word)
        return fail(ALCH_E_INVALID, "destination ring must be the source ring minus limb 0");
    for (int j = 1; j < rs->L; ++j)
        if (rs->q[j] != rd->q[j - 1]) return fail(ALCH_E_INVALID, "destination limbs must equal source limbs 1..L-1");
    if (count > src->n_elems || count > dst->n_elems) return fail(ALCH_E_INVALID, "count out of bounds");
    uint64_t inv[MAXL] = {0};
    for (int j = 1; j < rs->L; ++j) inv[j] = h_powmod(rs->q[0] % rs->q[j], rs->q[j] - 2, rs->q[j]);
    const size_t total = count * (size_t)rd->L * rd->n;
    HIP_TRY(hipStreamSynchronize(rd->stream));
    if (rs->word == 4) {
        Scal<u32> sm; scal_to_mont<u32>(rs, inv, 1, sm);
        hipLaunchKernelGGL((k_rescale_drop0<u32>), dim3(ew_grid(total)), dim3(256), 0, rs->stream, rs->d32, (const u32*)src->dptr, (u32*)dst->dptr, count, sm);
    } else {
        Scal<u64> sm; scal_to_mont<u64>(rs, inv, 1, sm);
        hipLaunchKernelGGL((k_rescale_drop0<u64>), dim3(ew_grid(total)), dim3(256), 0, rs->stream, rs->d64, (const u64*)src->dptr, (u64*)dst->dptr, count, sm);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(rs->stream));
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_rescale_add0(const alch_buf* src, alch_buf* dst, size_t count) try {
    if (!src || !dst) return fail(ALCH_E_INVALID, "null buffer");
    alch_ring* rs = src->ring;
    alch_ring* rd = dst->ring;
    BIND(rd);
    if (rd->L != rs->L + 1 || rd->n != rs->n || rd->word != rs->word)
        return fail(ALCH_E_INVALID, "destination ring must be the source ring plus one limb in front");
    for (int j = 0; j < rs->L; ++j)
        if (rs->q[j] != rd->q[j + 1]) return fail(ALCH_E_INVALID, "destination limbs 1..L must equal the source limbs");
    if (count > src->n_elems || count > dst->n_elems) return fail(ALCH_E_INVALID, "count out of bounds");
    uint64_t qa[MAXL] = {0};
    for (int j = 1; j < rd->L; ++j) qa[j] = rd->q[0] % rd->q[j];
    const size_t total = count * (size_t)rd->L * rd->n;
    HIP_TRY(hipStreamSynchronize(rs->stream));
    if (rd->word == 4) {
        Scal<u32> sm; scal_to_mont<u32>(rd, qa, 1, sm);
        hipLaunchKernelGGL((k_rescale_add0<u32>), dim3(ew_grid(total)), dim3(256), 0, rd->stream, rd->d32, (const u32*)src->dptr, (u32*)dst->dptr, count, sm);
    } else {
        Scal<u64> sm; scal_to_mont<u64>(rd, qa, 1, sm);
        hipLaunchKernelGGL((k_rescale_add0<u64>), dim3(ew_grid(total)), dim3(256), 0, rd->stream, rd->d64, (const u64*)src->dptr, (u64*)dst->dptr, count, sm);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(rd->stream));
    return ALCH_OK;
} catch (...) { return abi_catch(); }

#include "tensor_ext.inc.hpp"
