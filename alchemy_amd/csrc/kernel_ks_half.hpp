// k_ks_accum_half: the key-switch kernel split so that TWO workgroups fit one CU.
//
// One workgroup owns one HALF of the output slots of (ciphertext, limb j): slots [half*n/2, (half+1)*n/2).
// After the first Cooley-Tukey stage the two halves of a transform are independent, so each workgroup
//   * runs global stages 0..2 straight from HBM/L2 into registers (it reads the whole digit, computes only
//     its own half of the stage-0 outputs -- one extra modular product per coefficient, +6.7 % of a
//     transform's multiplies -- and writes n/2 words to LDS), then
//   * finishes the remaining log n - 3 stages as a sub-transform of size n/2 in 64 (+4 padding) KiB of LDS
//     (prefix = 2 + half, see ntt_pass); the result takes one more trip through LDS into the lane-contiguous
//     layout in which the hint rows, tensor inputs and results are read and written (1 KiB per wave access).
// With 68 KiB LDS, 512 threads and <= 128 VGPRs per workgroup a CU holds two workgroups whose HBM phases
// (tensor inputs, digits, hint, result stores) and barrier stalls hide under each other's butterflies;
// the one-workgroup-per-CU form (k_ks_accum) idles the VALU during those phases.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "ntt_engine.hpp"

// experiment switches (tools/build_variant.sh)
#ifndef ALCH_KS_SERIAL
#define ALCH_KS_SERIAL true
#endif
#ifndef ALCH_KS_GBARRIER
#define ALCH_KS_GBARRIER 1
#endif
#ifndef ALCH_KS_INIT_DEPTH
#define ALCH_KS_INIT_DEPTH 2
#endif
#ifndef ALCH_KS_HINT_DEPTH
#define ALCH_KS_HINT_DEPTH 5
#endif
#ifndef ALCH_A_PARTIALS
#define ALCH_A_PARTIALS 0
#endif
// Ablation switches (ALCH_EXP_FLAGS; wrong results, timing only) exist in -DALCH_ABLATE builds alone: in the product
// kernel they cost real instructions (the compiler hoists e.g. the "skip the tensor part" zero-fill in front of the branch).
#ifdef ALCH_ABLATE
#define KS_DBG(bit) (dbg_mask & (bit))
#else
#define KS_DBG(bit) false
#endif
// timing experiment only (wrong results): drop every workgroup barrier of the kernel
#ifdef ALCH_EXP_NOBARRIER
#define KS_SYNC() ((void)0)
#else
#define KS_SYNC() lds_barrier()
#endif

namespace alch {

// Diagnostic build only (-DALCH_STAMPS): wave 0 of every workgroup adds the shader-clock time it spent in
// each phase to g_ks_stamps[phase]; read back by tools/stamp_report.py.  No stamp executes in the product.
#ifdef ALCH_STAMPS
__device__ unsigned long long g_ks_stamps[1024 * 16];
#define KS_STAMP(ph)                                                                              \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long _t;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        _t_acc[ph] += _t - _t_prev;                                                               \
        _t_prev = _t;                                                                             \
    } while (0)
#define KS_STAMP_FLUSH()                                                                          \
    do {                                                                                          \
        if (threadIdx.x == KS_STAMP_LANE)                                                         \
            for (int _p = 0; _p < 11; ++_p) atomicAdd(&g_ks_stamps[(blockIdx.x & 1023) * 16 + _p], _t_acc[_p]); \
        unsigned long long _rt_last;                                                              \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_rt_last)::"memory");       \
        if (threadIdx.x == KS_STAMP_LANE) {                                                       \
            atomicAdd(&g_ks_stamps[(blockIdx.x & 1023) * 16 + 11], _t_prev - _t_first);           \
            atomicAdd(&g_ks_stamps[(blockIdx.x & 1023) * 16 + 12], _rt_last - _rt_first);         \
        }                                                                                         \
    } while (0)
}  // namespace alch
extern "C" __attribute__((visibility("default"), used)) int alch_debug_stamps(unsigned long long* out16) {
    static unsigned long long host[1024 * 16];
    if (hipDeviceSynchronize() != hipSuccess) return -6;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(alch::g_ks_stamps), sizeof host) != hipSuccess) return -6;
    for (int p = 0; p < 16; ++p) { out16[p] = 0; for (int w = 0; w < 1024; ++w) out16[p] += host[w * 16 + p]; }
    for (auto& v : host) v = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(alch::g_ks_stamps), host, sizeof host) != hipSuccess) return -6;
    return 0;
}
namespace alch {
#ifndef KS_STAMP_LANE
#define KS_STAMP_LANE 0
#endif
#define KS_STAMP_INIT()                                                                           \
    unsigned long long _t_acc[11] = {0};                                                          \
    unsigned long long _t_prev, _t_first, _rt_first;                                              \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_rt_first)::"memory");        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t_prev)::"memory");              \
    _t_first = _t_prev
#else
#define KS_STAMP(ph) do {} while (0)
#define KS_STAMP_INIT() do {} while (0)
#define KS_STAMP_FLUSH() do {} while (0)
#endif

// y * twiddle, fully reduced, for either twiddle representation
__device__ __forceinline__ u32 tw_mul(u32 y, u64 br, u32 q, u32) { return plant_mul(y, br, q); }
__device__ __forceinline__ u32 tw_mul(u32 y, u32 w, u32 q, u32 qni) { return csub(mont_mul_lazy(y, w, q, qni), q); }

__device__ __forceinline__ u32 mont_red_lazy(u64 p, u32 q, u32 qni) {       // p < 2^32 * q  ->  [0, 2q)
    u32 m = (u32)p * qni;
    return (u32)((p + (u64)m * q) >> 32);
}

// Global memory goes through buffer instructions: one descriptor per array (wave-uniform, from the kernel arguments),
// the per-lane part of every address is the single VGPR `lane16` = threadIdx.x * 16 bytes, everything else (work
// item, limb, slice rotation) is scalar and travels in the instruction's SGPR offset.  With flat loads the same
// addresses cost ~150 VALU instructions of 64-bit pointer arithmetic per digit transform.  All byte offsets are
// below 2^32 (the host caps the chunk size accordingly).
// CP: cache policy bits of the instruction (gfx940+: 1 = sc0, 2 = nt, 16 = sc1).  The tensor inputs and the results stream through once:
// marked non-temporal they do not displace the digits and hint rows (read by six items each) in L2 -- +1.7 % on the headline
// (same-box A/B, tools/ab_variants.sh: 532 k -> 541 k op/s; inputs alone +1 %, results alone +-0; the same on kernel A's operand loads +-0).
#ifndef ALCH_KS_NT_IN
#define ALCH_KS_NT_IN 2
#endif
#ifndef ALCH_KS_NT_OUT
#define ALCH_KS_NT_OUT 2
#endif
template <int CP = 0, typename Rsrc>
__device__ __forceinline__ u32 __attribute__((ext_vector_type(4))) buf_ld16(Rsrc r, u32 voff, u32 soff) {
    typedef u32 V4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, CP));
}
template <int CP = 0, typename Rsrc>
__device__ __forceinline__ void buf_st16(Rsrc r, u32 voff, u32 soff, u32 __attribute__((ext_vector_type(4))) v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0)), v), r, voff, soff, CP);
    ALCH_STORE_GUARD(v);
}

// EPT = coefficients (and accumulator pairs) per thread: 32 -> n/64 threads, <= 128 VGPRs, 4 waves per SIMD.
// EPT = 16 (n/32 threads, <= 64 VGPRs, 8 waves per SIMD) builds and is bit-exact, but spills 61 VGPRs and ran
// 30 % SLOWER on MI355X (294 k vs 418 k op/s), so it is not dispatched.
//
// UP (the full PT2CT mul_, PT2CT.hs:177): the operands live `dup` limbs below the hint's ring -- R is the hint's
// ring with L limbs, a/b/digits belong to its last L - dup limbs.  modSwitch up puts 0 into the added limbs
// and q_added * x into the others (a scalar, folded into spre by the host), so an added limb j < dup starts
// from c0 = c1 = 0, has a zero diagonal digit and transforms all L - dup digits; hint row of source digit i is
// i + dup (the digits of the added limbs are zero and are skipped).
// Q30 (every modulus below 2^30, so 4q fits a word): Harvey's forward butterflies (8 instructions instead of 10, values lazy in [0,4q)
// through pass G and the LDS passes -- the hint product takes any word), accumulators lazy in [0,2q) (one conditional subtraction per
// product instead of two) and brought to [0,q) when they are stored.
template <int LOGN, bool BALANCED, int EPT = 32, bool UP = false, bool Q30 = false>
__global__ void __launch_bounds__((1 << (LOGN - 1)) / EPT, EPT == 32 ? 4 : 8)
k_ks_accum_half(DevRing<u32> R, const u32* __restrict__ a, const u32* __restrict__ b,
                const int32_t* __restrict__ digits, const u32* __restrict__ hint, u32* __restrict__ out,
                unsigned nct, unsigned nitems, Scal<u32> spre, unsigned dbg_mask, int dup_) {
    typedef u32 W;
    constexpr int LOGM = LOGN - 1, M = 1 << LOGM, N = 1 << LOGN, LT = (EPT == 32) ? LOGN - 6 : LOGN - 5, T = 1 << LT;
    typedef Geo<LOGM, LT> G;
    static_assert(G::E == EPT && (EPT == 32 || EPT == 16), "16 or 32 coefficients per thread");
    static_assert((LOGM - 2) % 4 == 0, "remaining stages must split into radix-16 passes");
    typedef u32 V __attribute__((ext_vector_type(4)));
    typedef int32_t SV __attribute__((ext_vector_type(4)));
    constexpr int NG = EPT / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L;
    const int dup = UP ? dup_ : 0;
    const int Ls = L - dup;                         // limbs of the operands and number of digits
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(a), 0, (u32)((size_t)nct * 2 * Ls * N * 4), 0x00020000);
    const auto rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(b), 0, (u32)((size_t)nct * 2 * Ls * N * 4), 0x00020000);
    const auto rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(digits), 0, (u32)((size_t)nct * Ls * N * 4), 0x00020000);
    const auto rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(hint), 0, (u32)((size_t)2 * L * L * N * 4), 0x00020000);
    const auto ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, (u32)((size_t)nct * 2 * L * N * 4), 0x00020000);
    const u32 lane16 = threadIdx.x * 16u;
    constexpr u32 SLICE = (u32)T * 16u;             // bytes between two lane-contiguous slices
    // Persistent workgroups: the grid is two workgroups per CU and each loops over work items
    // (ciphertext, limb j, half): no partial last wave of workgroups, and the result stores of one item can be
    // issued behind the first loads of the next.  (An LDS-DMA prefetch of the next item's inputs towards L2 was
    // tried and measured 1 % slower late in the item, 6 % slower early: it only adds traffic.)
    // Item numbering is XCD-aware (speed only): the 2L items of one ciphertext agree mod 8, and gridDim is a
    // multiple of 8, so they run on one XCD and its L2 serves the digits they all read.
    const unsigned per = 16u * (unsigned)L;
    // dbg_mask bit 31 (launch option ks_map = 1, L | 8 only): limb-per-XCD numbering instead -- XCD x = item mod 8 serves limb x mod L
    // of every (8 / L)-th ciphertext, so an XCD's L2 sees the hint rows of ONE limb (1 MiB at L = 4) at the price of each digit being
    // fetched by the L - 1 XCDs that transform it.
    const bool limb_map = (dbg_mask >> 31) != 0;
    // dbg_mask bit 30 (launch option ks_rev = 1): walk the items from the last ciphertext of the chunk to the first -- the digits the
    // tensor kernel wrote last are the ones still in the Infinity Cache
    const bool rev = ((dbg_mask >> 30) & 1u) != 0;
    auto decode = [&](unsigned item, int& j_, int& hf_, size_t& ct_) {
        if (rev) item = nitems - 1u - item;
        if (limb_map) {
            const unsigned x = item & 7u, t = item >> 3, pl = 8u / (unsigned)L;
            j_ = (int)(x % (unsigned)L);
            hf_ = (int)(t & 1u);
            ct_ = (size_t)(t >> 1) * pl + x / (unsigned)L;
            return;
        }
        const unsigned grp = item / per, rem = item % per;
        const unsigned which = rem >> 3;
        j_ = (int)(which >> 1);
        hf_ = (int)(which & 1u);
        ct_ = (size_t)grp * 8u + (rem & 7u);
    };
    // Results of an item are stored at the start of the NEXT item, right after that item's first two slices of
    // tensor-input loads have been issued: vmcnt retires in order, so stores issued first would have to drain
    // to HBM before the next item's loads could be consumed.  (Spreading the stores over all eight slices was
    // measured 15 % slower.)
    W acc0[EPT], acc1[EPT];
    u32 po0 = 0, po1 = 0;                          // byte offsets of the previous item's two result rows
    bool pending = false;
    int prot = 0;
    W pq = 0;                                      // modulus of the previous item (Q30: its accumulators are reduced at the store)
    auto store_slice = [&](int r) {               // slice r of the previous item's results
        if (!pending) return;
        const u32 so = SLICE * (u32)((r + prot) & (EPT / 4 - 1));
        V v0, v1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v0[e] = Q30 ? csub(acc0[r * 4 + e], pq) : acc0[r * 4 + e];
            v1[e] = Q30 ? csub(acc1[r * 4 + e], pq) : acc1[r * 4 + e];
        }
        buf_st16<ALCH_KS_NT_OUT>(ro, lane16, po0 + so, v0);
        buf_st16<ALCH_KS_NT_OUT>(ro, lane16, po1 + so, v1);
    };
    auto flush_stores = [&]() {
#pragma unroll
        for (int r = 0; r < EPT / 4; ++r) store_slice(r);
        pending = false;
    };
    for (unsigned item = blockIdx.x; item < nitems; item += gridDim.x) {
    int j, hf;
    size_t ct;
    decode(item, j, hf, ct);
    if (ct >= nct) continue;
    // Workgroups run identical code in lockstep over rows that are 64 KiB-aligned, so reading every row from
    // its start makes all of them hit the same HBM channels at the same time (measured: the tensor-input and
    // result streams of this kernel moved only ~2 TB/s).  Each item therefore walks its EPT/4 lane-contiguous
    // 16-byte slices starting from a different one: slice r lives at chunk (r + rot) mod (EPT/4).
    const int rot = (int)((item ^ (item >> 3)) & (EPT / 4 - 1));

    const ModP<W> m = R.mod[j];
    const W q = m.q, qni = m.qni;
    const int js = j - dup;                         // this limb in the operands' ring; < 0: added by modSwitch
    // an added limb runs the same tensor code on limb 0's operands with scalar 0 (c0 = c1 = c2 = 0): a fifth of
    // the items waste ~10 % of their time, and the kernel keeps one copy of the load/store pipeline
    const W sr2 = js < 0 ? (W)0 : spre.v[js];
    const u32 slot0 = (u32)hf * (u32)M * 4u;                                   // byte offsets from here on
    const u32 cti = KS_DBG(1u) ? (u32)(ct & 7) : (u32)ct;                      // traffic experiment: alias the inputs
    const u32 jsz = (u32)(js < 0 ? 0 : js);
    constexpr u32 ROW = (u32)N * 4u;                                           // one limb-polynomial
    const u32 a0 = ((2 * cti) * (u32)Ls + jsz) * ROW + slot0, a1 = ((2 * cti + 1) * (u32)Ls + jsz) * ROW + slot0;
    const u32 hj = (u32)j * ROW + slot0;                                       // + ((i*2 + c)*L) rows
    const u32 hstride = (u32)L * ROW;

    KS_STAMP_INIT();
#if ALCH_A_PARTIALS
    if constexpr (!UP && !Q30) {
        // experiment (kernel_tensor_split.hpp, ALCH_A_PARTIALS): the starting values were formed by the tensor kernel and wait in the
        // result rows -- two loads per slice instead of six, no products
        const u32 o0 = ((2 * cti) * (u32)L + (u32)j) * ROW + slot0, o1 = ((2 * cti + 1) * (u32)L + (u32)j) * ROW + slot0;
        V st[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const u32 so = SLICE * (u32)((s + rot) & (EPT / 4 - 1));
            st[s][0] = buf_ld16(ro, lane16, o0 + so); st[s][1] = buf_ld16(ro, lane16, o1 + so);
        }
        flush_stores();                       // previous item's results: behind this item's first loads
#pragma unroll
        for (int s = 2; s < EPT / 4; ++s) {
            const u32 so = SLICE * (u32)((s + rot) & (EPT / 4 - 1));
            const V v0 = buf_ld16(ro, lane16, o0 + so), v1 = buf_ld16(ro, lane16, o1 + so);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc0[s * 4 + e] = v0[e]; acc1[s * 4 + e] = v1[e]; }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc0[s * 4 + e] = st[s][0][e]; acc1[s * 4 + e] = st[s][1][e]; }
    } else
#endif
    {   // c0, c1 and the diagonal digit (i == j): d_j = c2_j (mod q_j), no transform needed.
        // Eight 4-coefficient slices, software-pipelined: the six 16-byte loads of slice s+1 are in flight
        // while slice s is multiplied out (issuing them one slice at a time exposed the HBM latency eight
        // times per workgroup: 20 % of the kernel in the phase stamps).
        const u32 h0 = hj + (u32)(2 * j) * hstride, h1 = hj + (u32)(2 * j + 1) * hstride;
        constexpr int ID = ALCH_KS_INIT_DEPTH;         // slices of tensor inputs in flight (6 x 16 B per lane each)
        V in[ID][6];
        auto issue = [&](int s, V (&v)[6]) {
            const u32 so = SLICE * (u32)((s + rot) & (EPT / 4 - 1));               // lane-contiguous 16-byte pieces
            v[0] = buf_ld16<ALCH_KS_NT_IN>(ra, lane16, a0 + so); v[1] = buf_ld16<ALCH_KS_NT_IN>(ra, lane16, a1 + so);
            v[2] = buf_ld16<ALCH_KS_NT_IN>(rb, lane16, a0 + so); v[3] = buf_ld16<ALCH_KS_NT_IN>(rb, lane16, a1 + so);
            v[4] = buf_ld16(rh, lane16, h0 + so); v[5] = buf_ld16(rh, lane16, h1 + so);
        };
        issue(0, in[0]);
        issue(1, in[1]);
        flush_stores();                       // previous item's results: behind this item's first loads
#pragma unroll
        for (int s = 2; s < ID; ++s) issue(s, in[s]);
        (void)0;
        if (KS_DBG(1024u)) {
#pragma unroll
            for (int s = 0; s < EPT; ++s) { acc0[s] = 0; acc1[s] = 0; }
        } else {
#pragma unroll
        for (int s = 0; s < EPT / 4; ++s) {
            const V(&v)[6] = in[s % ID];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const W x0 = csub(mont_mul_lazy(v[0][e], sr2, q, qni), q);          // a0 s R
                const W x1 = csub(mont_mul_lazy(v[1][e], sr2, q, qni), q);          // a1 s R
                const W c2 = csub(mont_mul_lazy(v[3][e], x1, q, qni), q);           // a1 b1 s
                // sums of two products (< 2 q^2 < 2^32 q) share one Montgomery reduction
                acc0[s * 4 + e] = csub(mont_red_lazy((u64)x0 * v[2][e] + (u64)c2 * v[4][e], q, qni), q);
                // three products (< 3 q^2 < 2^64): bring the high word below q first, then one reduction
                const u64 p1 = (u64)x0 * v[3][e] + (u64)x1 * v[2][e] + (u64)c2 * v[5][e];
                const u64 p1r = ((u64)csub((W)(p1 >> 32), q) << 32) | (u32)p1;        // high word < 1.5 q -> < q
                acc1[s * 4 + e] = csub(mont_red_lazy(p1r, q, qni), q);
            }
            if (s + ID < EPT / 4) issue(s + ID, in[s % ID]);  // refill the buffer just consumed
            __builtin_amdgcn_sched_barrier(0);   // at most two slices of loads live
        }
        }
    }
    KS_STAMP(0);                                  // tensor part (c0, c1, diagonal digit)
    for (int i = 0; i < Ls; ++i) {
        if (i == js || KS_DBG(512u)) continue;
        const u32 d = ((KS_DBG(2u) ? (u32)(ct & 7) : (u32)ct) * (u32)Ls + (u32)i) * ROW;   // byte offset of digit i
        // Nothing below depends on i except d and the hint rows; keep addresses and twiddles from being
        // hoisted out of the digit loop (that costs ~250 spilled VGPRs).
        auto twf = fwd_tw(R, j);                        // Plantard constants (shared-twiddle passes)
        const W* twm = R.twf[j];                        // Montgomery words (per-lane last pass)
        int tid = threadIdx.x;
        asm volatile("" : "+s"(twf), "+s"(twm), "+v"(tid));
        KS_SYNC();      // previous transform's last pass has finished reading LDS
        KS_STAMP(1);                              // barrier before pass G

        // ---- global stages 0..2, HBM/L2 -> registers -> LDS
        if (!KS_DBG(256u)) {
            // stage 0 gives this half x + w1 y (lower) or x - w1 y (upper): the upper half multiplies by -w1 instead,
            // so both run the same instructions (no select per coefficient)
#if ALCH_USE_PLANTARD
            const auto w1 = twf[1];
#else
            const W w1 = hf ? q - twf[1] : twf[1];
#endif
            const auto w2 = twf[2 + hf], w3a = twf[4 + 2 * hf], w3b = twf[5 + 2 * hf];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int lo4 = (tid + T * g) * 4;                    // coefficients lo4..lo4+3 of each eighth
                V u[4];
                SV zxs[4], zys[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    zxs[k] = __builtin_bit_cast(SV, buf_ld16(rd, lane16, d + (u32)(k * (N / 8) + T * g * 4) * 4u));
                    zys[k] = __builtin_bit_cast(SV, buf_ld16(rd, lane16, d + (u32)((k + 4) * (N / 8) + T * g * 4) * 4u));
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const SV zx = zxs[k], zy = zys[k];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        W xx, yr;
                        if constexpr (BALANCED) {
                            // |z| < q: z mod q = min(z, z + q) as unsigned words (a negative z is a huge word)
                            const W zq = (W)zx[e] + q;
                            xx = zq < (W)zx[e] ? zq : (W)zx[e];
                            yr = (W)zy[e] + q;                                                      // (0, 2q)
                        } else {
                            xx = csub(mont_mul_lazy((W)((W)zx[e] + R.dig_off[j]), m.r1, q, qni), q);
                            yr = mont_mul_lazy((W)((W)zy[e] + R.dig_off[j]), m.r1, q, qni);
                        }
#if ALCH_USE_PLANTARD
                        const W t = tw_mul(yr, w1, q, qni);
                        u[k][e] = hf ? xx + (q - t) : xx + t;
#else
                        const W t = Q30 ? mont_mul_lazy(yr, w1, q, qni) : tw_mul(yr, w1, q, qni);     // Q30: [0,2q), the sum below 3q
                        u[k][e] = xx + t;
#endif
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    W u0 = u[0][e], u1 = u[1][e], u2 = u[2][e], u3 = u[3][e];
                    if constexpr (Q30) {
                        bfly_fwd4(u0, u2, w2, q, qni);
                        bfly_fwd4(u1, u3, w2, q, qni);
                        bfly_fwd4(u0, u1, w3a, q, qni);
                        bfly_fwd4(u2, u3, w3b, q, qni);
                    } else {
                    bfly_fwd(u0, u2, w2, q, qni);
                    bfly_fwd(u1, u3, w2, q, qni);
                    bfly_fwd(u0, u1, w3a, q, qni);
                    bfly_fwd(u2, u3, w3b, q, qni);
                    }
                    u[0][e] = u0; u[1][e] = u1; u[2][e] = u2; u[3][e] = u3;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<V*>(&lds[swz<LOGM>(k * (N / 8) + lo4)]) = u[k];
#if ALCH_KS_GBARRIER
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
        KS_STAMP(2);                              // pass G (global loads + stages 0..2 + LDS write)
        KS_SYNC();
        KS_STAMP(3);                              // barrier after pass G

        // ---- remaining stages: sub-transform of size n/2, local stages 2 .. LOGM-1
        const int ih = KS_DBG(8u) ? j : i + dup;                         // traffic experiment: alias the hint rows
        const u32 h0 = hj + (u32)(2 * ih) * hstride, h1 = hj + (u32)(2 * ih + 1) * hstride;
        const int prefix = 2 + hf;
        NoEpilogue none;
        constexpr int NP = (LOGM - 2) / 4;
        typedef typename std::remove_cv<typename std::remove_pointer<decltype(twf)>::type>::type TWF;
        typedef typename std::remove_cv<typename std::remove_pointer<decltype(twm)>::type>::type TWM;
        if constexpr (NP == 1) {
            ntt_pass<LOGM, LT, W, 2, 4, false, false, true, TWM, NoEpilogue&, true, Q30>(lds, twm, q, qni, (W)0, (W)0, tid, prefix, none);
        } else if constexpr (NP == 2) {
            ntt_pass<LOGM, LT, W, 2, 4, false, false, ALCH_KS_SERIAL, TWF, NoEpilogue&, true, Q30>(lds, twf, q, qni, (W)0, (W)0, tid, prefix, none);
            pair_sync<LOGM>();                    // LB = 4 -> LB = 0: the hand-off stays inside each wave
            ntt_pass<LOGM, LT, W, 6, 4, false, false, true, TWM, NoEpilogue&, true, Q30>(lds, twm, q, qni, (W)0, (W)0, tid, prefix, none);
        } else {
            static_assert(NP <= 3, "at most 3 LDS passes");
            if (!KS_DBG(16u)) ntt_pass<LOGM, LT, W, 2, 4, false, false, ALCH_KS_SERIAL, TWF, NoEpilogue&, true, Q30>(lds, twf, q, qni, (W)0, (W)0, tid, prefix, none);
            KS_STAMP(4);                          // LDS pass 1
            KS_SYNC();
            KS_STAMP(5);
            if (!KS_DBG(32u)) ntt_pass<LOGM, LT, W, 6, 4, false, false, ALCH_KS_SERIAL, TWF, NoEpilogue&, true, Q30>(lds, twf, q, qni, (W)0, (W)0, tid, prefix, none);
            KS_STAMP(6);                          // LDS pass 2
            pair_sync<LOGM>();                    // LB = 4 -> LB = 0: the hand-off stays inside each wave
            KS_STAMP(7);
            if (!KS_DBG(64u)) ntt_pass<LOGM, LT, W, 10, 4, false, false, true, TWM, NoEpilogue&, true, Q30>(lds, twm, q, qni, (W)0, (W)0, tid, prefix, none);
        }
        // hint multiply-accumulate, in the lane-contiguous slot layout: the transform result goes through LDS
        // once more so that hint loads (and the tensor inputs / result stores, which share the layout) are
        // fully coalesced 1 KiB wave accesses instead of 16-byte pieces at a 64-byte lane stride.
        // The hint rows of the first HD slices are requested before the barrier (LDS-only barriers let global loads
        // stay in flight), the rest HD slices ahead of their use: the rows come from L2 / Infinity Cache, whose
        // latency a shallower pipeline does not cover (measured with the XOR-swizzled LDS layout: depth 2 -> 448k, 4 -> 463k,
        // 6 -> 468k op/s; with the padded layout depth 6 makes the register allocator rotate the accumulators and
        // spill, and 4..5 are best: 500k -> 511k op/s).
        // (Running the digit loads of pass G one group ahead the same way spilled ~150 VGPRs: -20 %.)
        constexpr int HD = ALCH_KS_HINT_DEPTH;          // slices of hint rows in flight (2 x 16 B per lane each)
        V ph0[HD], ph1[HD];
        auto hint_issue = [&](int r, V& a0v, V& a1v) {
            const u32 so = SLICE * (u32)((r + rot) & (EPT / 4 - 1));
            a0v = buf_ld16(rh, lane16, h0 + so);
            a1v = buf_ld16(rh, lane16, h1 + so);
        };
#pragma unroll
        for (int r = 0; r < HD; ++r) hint_issue(r, ph0[r], ph1[r]);
        KS_SYNC();
        if (!KS_DBG(128u))
#pragma unroll
        for (int r = 0; r < EPT / 4; ++r) {
            const int idx = (tid + T * ((r + rot) & (EPT / 4 - 1))) * 4;
            const V x = *reinterpret_cast<const V*>(&lds[swz<LOGM>(idx)]);
            const V vh0 = ph0[r % HD], vh1 = ph1[r % HD];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (Q30) {                    // x in [0,4q), accumulators lazy in [0,2q)
                    acc0[r * 4 + e] = csub(acc0[r * 4 + e] + mont_mul_lazy(x[e], vh0[e], q, qni), 2u * q);
                    acc1[r * 4 + e] = csub(acc1[r * 4 + e] + mont_mul_lazy(x[e], vh1[e], q, qni), 2u * q);
                } else {
                acc0[r * 4 + e] = csub(acc0[r * 4 + e] + csub(mont_mul_lazy(x[e], vh0[e], q, qni), q), q);
                acc1[r * 4 + e] = csub(acc1[r * 4 + e] + csub(mont_mul_lazy(x[e], vh1[e], q, qni), q), q);
                }
            }
            if (r + HD < EPT / 4) hint_issue(r + HD, ph0[r % HD], ph1[r % HD]);
        }
        KS_STAMP(8);                              // last pass + hint multiply-accumulate
    }

    KS_STAMP(9);
    const u32 cto = KS_DBG(4u) ? (u32)(ct & 7) : (u32)ct;                // traffic experiment: alias the outputs
    if (!KS_DBG(2048u)) {
        po0 = ((2 * cto) * (u32)L + (u32)j) * ROW + slot0;
        po1 = ((2 * cto + 1) * (u32)L + (u32)j) * ROW + slot0;
        pending = true;
        prot = rot;
        pq = q;
    }
    KS_STAMP(10);                                 // result stores issued
    KS_STAMP_FLUSH();
    }  // item loop: the next item touches LDS only after the barrier that opens its first pass G
    flush_stores();
}

}  // namespace alch
