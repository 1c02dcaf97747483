// LDS-resident NTT kernels: batched crt / crtInv and the two fused kernels of ct_mul_relin.
//
//   k_crt          Tensor crt / crtInv on a batch of limb-polynomials, in place.
//   k_tensor_intt  per (ciphertext, limb i):  c2_i = a1_i * b1_i * s_i  (the quadratic coefficient of
//                  SymmSHE's (*), Crypto/Alchemy/Interpreter/Eval.hs:65-67), crtInv in LDS, centred lift
//                  -> TrivGad digit d_i (signed words, Pow basis).  First half of keySwitchQuadCirc
//                  (Eval.hs:133): `decompose c2`.
//   k_ks_accum     per (ciphertext, limb j):  out_{0,1} = c_{0,1} + sum_i crt_j(reduce d_i) * hint_i,{0,1}
//                  with c0 = a0 b0 s, c1 = (a0 b1 + a1 b0) s; the i = j digit is c2_j itself (d_j = c2_j
//                  mod q_j), so only L-1 transforms run per limb.  Accumulators stay in registers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "ntt_engine.hpp"

namespace alch {

constexpr int MAXL = 8;

template <typename W>
struct DevRing {
    int L;
    int logn;            // two-power rings: log2 n; general-index rings: 0
    u32 n;               // ring dimension phi(m)
    ModP<W> mod[MAXL];
    W ninv_m[MAXL];      // n^-1 in Montgomery form
    W w1ninv_m[MAXL];    // tw_inv[1] * n^-1 in Montgomery form
    W dig_off[MAXL];     // multiple of q_j >= max_i (q_i-1)/2: makes a signed digit non-negative
    const W* twf[MAXL];  // forward twiddles (device, Montgomery form), n words
    const W* twi[MAXL];  // inverse twiddles
    const u64* twp[MAXL]; // forward twiddles as Plantard constants (32-bit rings only), n x 8 bytes
    const Sh64* tws[MAXL];  // 64-bit rings: forward twiddles as Shoup pairs (plain residue, floor(w 2^64 / q)), n x 16 bytes
    const Sh64* twsi[MAXL]; // 64-bit rings: inverse twiddles likewise
};

// Forward-transform twiddle table of limb j.  ALCH_USE_PLANTARD=1 switches 32-bit rings to Plantard constants
// (9-instruction butterfly instead of 10, two-word twiddles).  Measured on MI355X (interleaved A/B runs of
// bench.py): 9 % fewer VALU instructions in k_ks_accum_half but 2 % LOWER throughput -- the early passes get
// faster and the hint/digit-load phases slower by the same amount -- so the default stays Montgomery.
#ifndef ALCH_USE_PLANTARD
#define ALCH_USE_PLANTARD 0
#endif
#if ALCH_USE_PLANTARD
__device__ __forceinline__ const u64* fwd_tw(const DevRing<u32>& R, int j) { return R.twp[j]; }
#else
__device__ __forceinline__ const u32* fwd_tw(const DevRing<u32>& R, int j) { return R.twf[j]; }
#endif
__device__ __forceinline__ const Sh64* fwd_tw(const DevRing<u64>& R, int j) { return R.tws[j]; }
// table of the forward transform's last pass (per-lane twiddles) and of the inverse transform
__device__ __forceinline__ const u32* fwd_twm(const DevRing<u32>& R, int j) { return R.twf[j]; }
__device__ __forceinline__ const Sh64* fwd_twm(const DevRing<u64>& R, int j) { return R.tws[j]; }
__device__ __forceinline__ const u32* inv_tw(const DevRing<u32>& R, int j) { return R.twi[j]; }
__device__ __forceinline__ const Sh64* inv_tw(const DevRing<u64>& R, int j) { return R.twsi[j]; }

template <typename W> struct Scal { W v[MAXL]; };   // per-limb scalars passed by value

template <typename W> struct Signed;
template <> struct Signed<u32> { typedef int32_t type; };
template <> struct Signed<u64> { typedef int64_t type; };

}  // namespace alch
#include "kernel_ks_half.hpp"
#include "kernel_crt_half.hpp"
namespace alch {

enum OpKind { OP_CRT = 0, OP_CRTINV = 1, OP_TENSOR_INTT = 2, OP_KS_ACCUM = 3, OP_RESCALE_OUT = 4,
              OP_CRT_DIGITS = 5 /* split rings: src = c2 (Pow), data = digits [ct][L][L][n], npoly = ct*L*L */,
              OP_CRT_BASE2 = 6 /* src = c2 (Pow), data = digits [ct][D][L][n], npoly = ct*D*L; b2_first / b2_kd / b2_D */,
              OP_KS_SPLIT = 7 /* split rings: src = c2 (Pow), a = c2 (CRT copy), hint, out (holds c0, c1; updated in place), nct, dup */ };

constexpr int MAXDROP = 3;
template <typename W>
struct DropTab {
    int ddn;                       // limbs dropped (outermost first), 1..MAXDROP
    int balanced;                  // every centred residue of a dropped limb is smaller than every kept modulus
    W qinv_m[MAXDROP][MAXL];       // q_u^-1 mod q_t in Montgomery form (t > u)
    W comb_m[MAXDROP][MAXL];       // prod_{v >= u} q_v^-1 mod q_t in Montgomery form (t >= ddn): k_rescale_out_lin
};

// Launch-structure options of the fused kernels (alch_ring_set_option; defaults are the measured optima).
struct LaunchOpts {
    int ti_split = 1;        // n = 2^15 tensor kernel: 1 = split form, one workgroup per item; 0 = whole-polynomial kernel; > 1 = that many persistent workgroups
    int ti_grid = -1;        // whole-polynomial tensor kernel: -1 = one resident set of persistent workgroups, 0 = one workgroup per item, n > 0 = n workgroups
    int split_fused = 2;     // n = 2^16 (32-bit) / 2^15 (64-bit) key switch: 2 = two launches per chunk, tensor product in the loaders
                             // (k_tensor_crtinv_split + k_ks_accum_split<FROM_OPS>; alch_ct_mul_relin with TrivGad hints), 1 = element-wise tensor
                             // kernel + crtInv + digit transforms and hint products in one kernel (k_ks_accum_split; what alch_ct_mul_full
                             // uses), 0 = every step its own kernel
    int gen_fused = 1;       // general index: 1 = fused tensor + key switch kernels (kernel_gen.hpp).  Round 2 measured 0.69-0.82x the composed
                             // path (48 accumulators + the CRT_13 pass matrices in VGPRs spilled); with the pass matrices in SGPRs (round 3) the
                             // kernel needs 127 VGPRs, no scratch: 1.13x on H5', 1.24x on H3', 1.08x on H1', 0.99x on H0'  (key switch, L = 4)
    bool tunnel_mac = true;  // alch_ct_tunnel, E'-level digits: hint inner product with lazy 64-bit accumulation, evalLin's constant term as its start (k_tunnel_mac_e); 0 = k_tunnel_lin + k_hint_mac_e
    int tunnel_fused = 0;    // alch_ct_tunnel (TrivGad, E'-level transforms): 2 or 4 = digit transforms + hint products in one kernel with that many
                             // digits side by side per workgroup (k_gen_tunnel_ks); 0 = k_gen_crt_digits + k_hint_mac_v through HBM.  Measured on the
                             // HomomRLWR pipeline (round 3): 41.7 k ringRounds/s composed, 41.2 k fused (hop 1: 2.17 -> 2.36 ms, hop 3: 3.75 -> 4.1 ms,
                             // hop 2: 3.12 -> 3.05): the E'-size transforms fill a 512-thread workgroup poorly and the digit round trip they save
                             // is small (phi(e') words per digit) -- kept as a tested option, off
    int crt_half = 1;        // crt / crtInv of a 128-KiB limb-polynomial: 1 = two half-size workgroups per CU (k_crt_half), 0 = k_crt
    int rs_half = 0;         // closing modSwitch at n = 2^15 (32-bit): 1 = always the two launches of half-size workgroups (kernel_rescale_half.hpp);
                             // 0 = only where k_rescale_out_lin cannot serve (three dropped limbs, unbalanced two-limb drops).  Measured on the
                             // 4 -> 5 -> 3 mul_: 0.473 + 0.925 ms per 1024 ciphertexts against k_rescale_out_lin's 1.331 ms -- the combination is
                             // recomputed from the stash per kept limb and costs what the second workgroup per CU gains
    int tunnel_ep = 1;       // alch_tunnel_create: 1 = transforms of the embedded E'-coefficients at dimension phi(e') (embedCRT replication), 0 = at phi(s')
    int rs_lin = 1;          // closing modSwitch: 1 = kept limbs stay in the CRT basis (k_rescale_out_lin), 0 = every limb through the Pow basis
    int ks_map = 0;          // k_ks_accum_half item numbering: 0 = the items of a ciphertext share an XCD (digits in its L2), 1 = a limb per XCD (hint rows in its L2)
    int ks_rev = 0;          // k_ks_accum_half: 1 = items from the chunk's last ciphertext to its first
    int q30 = 1;             // 32-bit two-power rings whose moduli are all below 2^30: 1 = Harvey butterflies in the fused n = 2^15 / 2^11 kernels
    unsigned ks_grid = 4096; // persistent workgroups of k_ks_accum_half (measured, 1024-ciphertext chunks: 2048 -> 511k, 4096 -> 517k, 8192 -> 513k op/s)
};

template <typename W>
struct NttCall {
    OpKind op;
    LaunchOpts opts;
    const DevRing<W>* ring;
    hipStream_t stream;
    // OP_CRT / OP_CRTINV
    W* data;
    const W* src;          // null: in place
    bool partials = false; // ALCH_A_PARTIALS builds: the tensor kernel also writes the key-switch kernel's starting values into `out`
    size_t first_poly, npoly;
    // fused kernels
    const W* a;
    const W* b;
    void* digits;          // Signed<W>*, [ct][L][n]
    const W* hint;         // [digit][2][L][n], Montgomery form
    W* out;
    size_t nct;            // ciphertexts in this launch (a, b, out, digits already offset to the first)
    Scal<W> spre_r2;       // s_j * R^2 mod q_j
    bool balanced;         // every |digit| < every q_j  ->  reduce is one add
    bool q30;              // 32-bit ring, every modulus below 2^30: the Harvey-butterfly instantiations of the fused n = 2^15 kernels
    // full mul_ (modSwitch . keySwitchQuad . modSwitch): see kernel_rescale_out.hpp
    int dup;               // OP_KS_ACCUM: limbs the hint's ring has in front of the operands' ring
    DropTab<W> drop;       // OP_RESCALE_OUT: limbs to drop and the q_u^-1 tables
    void* stash;           // OP_RESCALE_OUT: Signed<W> [grid][ddn][n]
    unsigned stash_slots;
    bool pow_out;
    Scal<u32> b2_first, b2_kd;   // OP_CRT_BASE2: BaseBGad 2 layout (first digit and digit count of every limb)
    u32 b2_D;
};

// ---- staging helpers -------------------------------------------------------------------------------
template <int LOGN, typename W, typename F>
__device__ __forceinline__ void stage_in(W* lds, F&& load4) {
    typedef Geo<LOGN> G;
    typedef typename Vec4<W>::type V;
    constexpr int VL = Vec4<W>::LANES;
#pragma unroll
    for (int r = 0; r < G::E / VL; ++r) {
        const int idx = (threadIdx.x + G::T * r) * VL;
        V v = load4(idx);
        *reinterpret_cast<V*>(&lds[swz<LOGN>(idx)]) = v;
    }
}

// ---- batched crt / crtInv --------------------------------------------------------------------------
template <int LOGN, typename W, bool INVERSE>
__global__ void __launch_bounds__(Geo<LOGN>::T) k_crt(DevRing<W> R, W* data, const W* src, size_t first_poly) {
    typedef Geo<LOGN> G;
    typedef typename Vec4<W>::type V;
    constexpr int VL = Vec4<W>::LANES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const size_t p = first_poly + blockIdx.x;
    const int j = (int)(p % (size_t)R.L);
    W* poly = data + p * (size_t)G::N;
    const W* in = src ? src + p * (size_t)G::N : poly;      // src != null: out of place (same element layout)
    const W q = R.mod[j].q, qni = R.mod[j].qni;

    stage_in<LOGN, W>(lds, [&](int idx) { return *reinterpret_cast<const V*>(in + idx); });
    lds_barrier();
    if constexpr (!INVERSE) ntt_forward<LOGN, W, false>(lds, fwd_tw(R, j), fwd_twm(R, j), q, qni, (int)threadIdx.x, NoEpilogue());
    else ntt_inverse<LOGN, W, false>(lds, inv_tw(R, j), q, qni, R.ninv_m[j], R.w1ninv_m[j], (int)threadIdx.x, NoEpilogue());
#pragma unroll
    for (int r = 0; r < G::E / VL; ++r) {
        const int idx = (threadIdx.x + G::T * r) * VL;
        V v = *reinterpret_cast<const V*>(&lds[swz<LOGN>(idx)]);
#pragma unroll
        for (int e = 0; e < VL; ++e) v[e] = csub(v[e], q);
        *reinterpret_cast<V*>(poly + idx) = v;
    }
}

// crt of the reduced BaseBGad 2 digits with decompose + reduce in the loader (unfused key switch, LDS-resident
// sizes): workgroup p = (ciphertext, digit d, target limb j).  Digit d is bit t of source limb i; with u = -(centred
// lift of c2_i) the balanced binary digits have the closed form  d_t = -((u >> t) & 1)  for t < k - 1 and the top
// digit is  -(u >> (k - 1))  (arithmetic shifts), so no digit depends on the ones below it.
template <int LOGN, typename W>
__global__ void __launch_bounds__(Geo<LOGN>::T) k_crt_base2_digits(DevRing<W> R, const W* __restrict__ c2pow, W* __restrict__ digits,
                                                                   Scal<u32> first_digit, Scal<u32> kd, u32 D) {
    typedef Geo<LOGN> G;
    typedef typename Vec4<W>::type V;
    typedef typename Signed<W>::type SW;
    constexpr int VL = Vec4<W>::LANES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const size_t p = blockIdx.x;
    const int L = R.L;
    const int j = (int)(p % (size_t)L);
    const u32 d = (u32)((p / (size_t)L) % D);
    const size_t ct = p / ((size_t)L * D);
    int i = 0;
    for (int c = 1; c < L; ++c) if (d >= first_digit.v[c]) i = c;
    const u32 t = d - first_digit.v[i];
    const bool top = t + 1 == kd.v[i];
    const W* src = c2pow + (ct * (size_t)L + i) * (size_t)G::N;
    W* dst = digits + p * (size_t)G::N;
    const W q = R.mod[j].q, qni = R.mod[j].qni, qi = R.mod[i].q, hqi = (qi - 1) >> 1;
    stage_in<LOGN, W>(lds, [&](int idx) {
        const V v = *reinterpret_cast<const V*>(src + idx);
        V o;
#pragma unroll
        for (int e = 0; e < VL; ++e) {
            const SW z = v[e] > hqi ? (SW)v[e] - (SW)qi : (SW)v[e];
            const SW u = -z;
            SW dg = top ? -(u >> t) : -((u >> t) & 1);
            if (top) { dg %= (SW)q; }                       // the top digit is tiny but its size depends on q_i vs 2^k
            o[e] = dg < 0 ? (W)(dg + (SW)q) : (W)dg;
        }
        return o;
    });
    lds_barrier();
    ntt_forward<LOGN, W, false>(lds, fwd_tw(R, j), fwd_twm(R, j), q, qni, (int)threadIdx.x, NoEpilogue());
#pragma unroll
    for (int r = 0; r < G::E / VL; ++r) {
        const int idx = (threadIdx.x + G::T * r) * VL;
        V v = *reinterpret_cast<const V*>(&lds[swz<LOGN>(idx)]);
#pragma unroll
        for (int e = 0; e < VL; ++e) v[e] = csub(v[e], q);
        *reinterpret_cast<V*>(dst + idx) = v;
    }
}

#ifndef ALCH_TI_STAMP_LANE
#define ALCH_TI_STAMP_LANE 960
#endif
#ifdef ALCH_STAMPS
__device__ unsigned long long g_ti_stamps[1024 * 16];
#define TI_STAMP(ph)                                                                              \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long _t;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        _ta[ph] += _t - _tp;                                                                      \
        _tp = _t;                                                                                 \
    } while (0)
}  // namespace alch
extern "C" __attribute__((visibility("default"), used)) int alch_debug_stamps_a(unsigned long long* out16) {
    static unsigned long long host[1024 * 16];
    if (hipDeviceSynchronize() != hipSuccess) return -6;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(alch::g_ti_stamps), sizeof host) != hipSuccess) return -6;
    for (int p = 0; p < 16; ++p) { out16[p] = 0; for (int w = 0; w < 1024; ++w) out16[p] += host[w * 16 + p]; }
    for (auto& v : host) v = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(alch::g_ti_stamps), host, sizeof host) != hipSuccess) return -6;
    return 0;
}
namespace alch {
#else
#define TI_STAMP(ph) do {} while (0)
#endif

// ablation switches of kernel A (ALCH_EXP_A): -DALCH_ABLATE builds only, see kernel_ks_half.hpp
#ifdef ALCH_ABLATE
#define TI_DBG(bit) (dbg & (bit))
#else
#define TI_DBG(bit) false
#endif

// ---- fused kernel A: tensor c2 + crtInv + centred lift ---------------------------------------------
template <int LOGN, typename W>
__global__ void __launch_bounds__(Geo<LOGN>::T)
k_tensor_intt(DevRing<W> R, const W* __restrict__ a, const W* __restrict__ b,
              typename Signed<W>::type* __restrict__ digits, unsigned nitems, Scal<W> spre, unsigned dbg) {
    typedef Geo<LOGN> G;
    typedef typename Vec4<W>::type V;
    typedef typename Signed<W>::type SW;
    constexpr int VL = Vec4<W>::LANES;
    constexpr int NV = G::E / VL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L;
    // Persistent workgroups (one polynomial fills the CU's LDS, so one workgroup per CU is resident): the
    // a1/b1 coefficients of the NEXT item are loaded into registers while the current transform runs, which
    // hides the HBM latency that a one-workgroup CU cannot hide by switching workgroups.
    // Global memory through buffer instructions (see kernel_ks_half.hpp): per-lane part of every address is one
    // VGPR, the rest is scalar.  Byte offsets stay below 2^32 (the host caps the chunk).
    const u32 nct = nitems / (unsigned)L;
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<W*>(a), 0, (u32)((size_t)nct * 2 * L * G::N * sizeof(W)), 0x00020000);
    const auto rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<W*>(b), 0, (u32)((size_t)nct * 2 * L * G::N * sizeof(W)), 0x00020000);
    const auto rd = __builtin_amdgcn_make_buffer_rsrc(digits, 0, (u32)((size_t)nct * L * G::N * sizeof(W)), 0x00020000);
    const u32 lane16 = threadIdx.x * 16u;
    constexpr u32 ROW = (u32)G::N * (u32)sizeof(W);
    V pa[NV], pb[NV];
    auto issue = [&](unsigned item) {
        const u32 ct = TI_DBG(1u) ? ((item / (unsigned)L) & 7) : item / (unsigned)L;   // dbg: timing experiments
        const u32 i = item % (unsigned)L;
        const u32 row = ((2 * ct + 1) * (u32)L + i) * ROW;
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            // rotated start per item: lockstep workgroups must not all read the same offset of their
            // 128 KiB-aligned rows at the same time (HBM channel conflicts)
            const u32 so = row + (u32)G::T * 16u * ((r + (item ^ (item >> 3))) & (NV - 1));
            pa[r] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(ra, lane16, so, 0));
            pb[r] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rb, lane16, so, 0));
        }
    };
#ifdef ALCH_STAMPS
    unsigned long long _ta[8] = {0}, _tp;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_tp)::"memory");
#endif
    unsigned item = blockIdx.x;
    if (item < nitems) issue(item);
    for (; item < nitems; item += gridDim.x) {
        const size_t ct = item / (unsigned)L;
        const int i = (int)(item % (unsigned)L);
        const ModP<W> m = R.mod[i];
        const W q = m.q, qni = m.qni;
        // The scalar s (and the R that turns the Montgomery product a1 b1 R^-1 back) rides on the n^-1 constants the
        // last inverse stage multiplies by anyway: the transform is linear.
        const W ninv_s = mont_mul(R.ninv_m[i], spre.v[i], m), w1ninv_s = mont_mul(R.w1ninv_m[i], spre.v[i], m);
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));          // keep LDS address arithmetic inside the item loop (VGPR pressure)
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int idx = (tid + G::T * ((r + (int)(item ^ (item >> 3))) & (NV - 1))) * VL;
            V v;
#pragma unroll
            for (int e = 0; e < VL; ++e) v[e] = mont_mul_lazy(pa[r][e], pb[r][e], q, qni);   // a1 b1 R^-1 in [0,2q)
            *reinterpret_cast<V*>(&lds[swz<LOGN>(idx)]) = v;
        }
        TI_STAMP(0);                            // loads + c2 + LDS write
        lds_barrier();
        TI_STAMP(1);
        const u32 drow = ((u32)ct * (u32)L + (u32)i) * ROW;
        const W half = (q - 1) >> 1;
        // The digit is stored straight from the last (strided) pass: 32 dword stores per lane, each wave store
        // 256 contiguous bytes.  Routing the result through LDS for 16-byte stores was measured 4 % slower.
        constexpr int RR = 1 << G::NS0;
        constexpr int STRIDE = G::N / RR;
        auto epi = [&](int, int base, W* x) {
#pragma unroll
            for (int k = 0; k < RR; ++k) {
                W v = csub(x[k], q);
                SW z = v > half ? (SW)v - (SW)q : (SW)v;
                if (!TI_DBG(4u)) {
                    if constexpr (sizeof(W) == 4)
                        __builtin_amdgcn_raw_buffer_store_b32((u32)z, rd, (u32)base * 4u, drow + (u32)(k * STRIDE) * 4u, 0);
                    else
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b64(rd, 0, 0, 0)), z),
                                                              rd, (u32)base * 8u, drow + (u32)(k * STRIDE) * 8u, 0);
                }
            }
        };
#ifdef ALCH_STAMPS
        if constexpr (LOGN == 15) {            // ntt_inverse spelled out so that the passes can be stamped
            NoEpilogue none;
            auto twi = inv_tw(R, i);
            ntt_pass<LOGN, G::LOGT, W, 11, 4, true, false, false>(lds, twi, q, qni, R.ninv_m[i], R.w1ninv_m[i], tid, 1, none);
            TI_STAMP(6); pair_sync<LOGN>(); TI_STAMP(3);
            ntt_pass<LOGN, G::LOGT, W, 7, 4, true, false, false>(lds, twi, q, qni, R.ninv_m[i], R.w1ninv_m[i], tid, 1, none);
            TI_STAMP(7); lds_barrier(); TI_STAMP(3);
            if (item + gridDim.x < nitems) issue(item + gridDim.x);
            ntt_pass<LOGN, G::LOGT, W, 3, 4, true, false, false>(lds, twi, q, qni, R.ninv_m[i], R.w1ninv_m[i], tid, 1, none);
            TI_STAMP(2); lds_barrier(); TI_STAMP(3);
            ntt_pass<LOGN, G::LOGT, W, 0, 3, true, true, false>(lds, twi, q, qni, ninv_s, w1ninv_s, tid, 1, epi);
            TI_STAMP(4);
        } else
#endif
        if (!TI_DBG(2u))
        ntt_inverse<LOGN, W, true>(lds, inv_tw(R, i), q, qni, ninv_s, w1ninv_s, tid, epi,
                                   [&]() { if (item + gridDim.x < nitems) issue(item + gridDim.x); });
        lds_barrier();                       // every lane has read its last-pass inputs before LDS is refilled
        TI_STAMP(5);
    }
#ifdef ALCH_STAMPS
    if (threadIdx.x == ALCH_TI_STAMP_LANE)
        for (int p = 0; p < 8; ++p) atomicAdd(&g_ti_stamps[(blockIdx.x & 1023) * 16 + p], _ta[p]);
#endif
}

// ---- fused kernel B: digit transforms + key-switch inner product ------------------------------------
template <int LOGN, typename W, bool BALANCED>
__global__ void __launch_bounds__(Geo<LOGN>::T)
k_ks_accum(DevRing<W> R, const W* __restrict__ a, const W* __restrict__ b,
           const typename Signed<W>::type* __restrict__ digits, const W* __restrict__ hint, W* __restrict__ out,
           unsigned nct, Scal<W> spre, int dup) {
    typedef Geo<LOGN> G;
    typedef typename Vec4<W>::type V;
    typedef typename Signed<W>::type SW;
    typedef SW SV __attribute__((ext_vector_type(Vec4<W>::LANES)));
    constexpr int VL = Vec4<W>::LANES;
    constexpr int NG = G::E / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L;
    // XCD-aware placement: workgroups b and b+8 share an XCD (round-robin dispatch), so the L limb
    // workgroups of one ciphertext -- which all read the same digits d_i -- are given ids that agree
    // mod 8 and hit the same L2.  Purely a speed choice; any placement is correct.
    const unsigned grp = blockIdx.x / (8u * (unsigned)L);
    const unsigned rem = blockIdx.x % (8u * (unsigned)L);
    const int j = (int)(rem >> 3);
    const size_t ct = (size_t)grp * 8u + (rem & 7u);
    if (ct >= nct) return;

    const ModP<W> m = R.mod[j];
    const W q = m.q, qni = m.qni;
    // dup > 0: the operands live dup limbs below the hint's ring (see k_ks_accum_half)
    const int Ls = L - dup, js = j - dup;
    const size_t jsz = (size_t)(js < 0 ? 0 : js);
    const W sr2 = spre.v[jsz];
    const size_t n = (size_t)G::N;
    const W* a0 = a + ((2 * ct) * (size_t)Ls + jsz) * n;
    const W* a1 = a + ((2 * ct + 1) * (size_t)Ls + jsz) * n;
    const W* b0 = b + ((2 * ct) * (size_t)Ls + jsz) * n;
    const W* b1 = b + ((2 * ct + 1) * (size_t)Ls + jsz) * n;
    const W* hj = hint + (size_t)j * n;                       // + ((i*2 + c)*L)*n
    const size_t hstride = (size_t)L * n;

    W acc0[G::E], acc1[G::E];
    // c0, c1 and the diagonal digit (i == j): d_j = c2_j (mod q_j), no transform needed
    if (js < 0) {
#pragma unroll
        for (int s = 0; s < G::E; ++s) { acc0[s] = 0; acc1[s] = 0; }
    } else {
        const W* h0 = hj + (size_t)(2 * j) * hstride;
        const W* h1 = hj + (size_t)(2 * j + 1) * hstride;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int k = 0; k < 16; k += VL) {
                const int idx = (threadIdx.x + G::T * g) * 16 + k;
                V va0 = *reinterpret_cast<const V*>(a0 + idx), va1 = *reinterpret_cast<const V*>(a1 + idx);
                V vb0 = *reinterpret_cast<const V*>(b0 + idx), vb1 = *reinterpret_cast<const V*>(b1 + idx);
                V vh0 = *reinterpret_cast<const V*>(h0 + idx), vh1 = *reinterpret_cast<const V*>(h1 + idx);
#pragma unroll
                for (int e = 0; e < VL; ++e) {
                    W x0 = csub(mont_mul_lazy(va0[e], sr2, q, qni), q);       // a0 s R
                    W x1 = csub(mont_mul_lazy(va1[e], sr2, q, qni), q);       // a1 s R
                    W c0 = csub(mont_mul_lazy(vb0[e], x0, q, qni), q);        // a0 b0 s
                    W c1 = csub(csub(mont_mul_lazy(vb1[e], x0, q, qni), q) +
                                csub(mont_mul_lazy(vb0[e], x1, q, qni), q), q);   // (a0 b1 + a1 b0) s
                    W c2 = mont_mul_lazy(vb1[e], x1, q, qni);                 // a1 b1 s, lazy
                    acc0[g * 16 + k + e] = csub(c0 + csub(mont_mul_lazy(c2, vh0[e], q, qni), q), q);
                    acc1[g * 16 + k + e] = csub(c1 + csub(mont_mul_lazy(c2, vh1[e], q, qni), q), q);
                }
                __builtin_amdgcn_sched_barrier(0);   // keep one 4-coefficient slice of loads live at a time
            }
        }
    }

    for (int i = 0; i < Ls; ++i) {
        if (i == js) continue;
        const SW* d = digits + (ct * (size_t)Ls + i) * n;
        lds_barrier();      // previous transform's last pass has finished reading LDS
        stage_in<LOGN, W>(lds, [&](int idx) {
            SV z = *reinterpret_cast<const SV*>(d + idx);
            V v;
#pragma unroll
            for (int e = 0; e < VL; ++e) {
                if constexpr (BALANCED) v[e] = (W)z[e] + q;                                   // (0, 2q)
                else v[e] = mont_mul_lazy((W)((W)z[e] + R.dig_off[j]), m.r1, q, qni);        // [0, 2q)
            }
            return v;
        });
        lds_barrier();
        const W* h0 = hj + (size_t)(2 * (i + dup)) * hstride;
        const W* h1 = hj + (size_t)(2 * (i + dup) + 1) * hstride;
        // Neither the twiddles nor the LDS addresses depend on i; unless both are made opaque here the
        // compiler hoists every pass's address arithmetic and twiddle loads out of the digit loop and
        // spills ~250 VGPRs per lane.
        auto twf = fwd_tw(R, j);
        auto twm = fwd_twm(R, j);
        int tid = threadIdx.x;
        asm volatile("" : "+s"(twf), "+s"(twm), "+v"(tid));
        ntt_forward<LOGN, W, true, true>(lds, twf, twm, q, qni, tid, [&acc0, &acc1, h0, h1, q, qni](int g, int base, W* x) {
#pragma unroll
            for (int k = 0; k < 16; k += VL) {
                V vh0 = *reinterpret_cast<const V*>(h0 + base + k), vh1 = *reinterpret_cast<const V*>(h1 + base + k);
#pragma unroll
                for (int e = 0; e < VL; ++e) {
                    acc0[g * 16 + k + e] = csub(acc0[g * 16 + k + e] + csub(mont_mul_lazy(x[k + e], vh0[e], q, qni), q), q);
                    acc1[g * 16 + k + e] = csub(acc1[g * 16 + k + e] + csub(mont_mul_lazy(x[k + e], vh1[e], q, qni), q), q);
                }
            }
        });
    }

    W* o0 = out + ((2 * ct) * (size_t)L + j) * n;
    W* o1 = out + ((2 * ct + 1) * (size_t)L + j) * n;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int k = 0; k < 16; k += VL) {
            const int idx = (threadIdx.x + G::T * g) * 16 + k;
            V v0, v1;
#pragma unroll
            for (int e = 0; e < VL; ++e) { v0[e] = acc0[g * 16 + k + e]; v1[e] = acc1[g * 16 + k + e]; }
            *reinterpret_cast<V*>(o0 + idx) = v0;
            *reinterpret_cast<V*>(o1 + idx) = v1;
        }
    }
}

}  // namespace alch
#include "kernel_rescale_out.hpp"
#include "kernel_rescale_half.hpp"
#include "kernel_tensor_split.hpp"
namespace alch {

// ---- launcher ----------------------------------------------------------------------------------------
template <typename K>
inline hipError_t set_lds(K kernel, size_t bytes) {
    if (bytes <= 65536) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

template <typename W, int LOGN>
inline hipError_t run_call(const NttCall<W>& c) {
    typedef Geo<LOGN> G;
    typedef typename Signed<W>::type SW;
    const size_t lds_bytes = (size_t)lds_words<LOGN>() * sizeof(W);
    const DevRing<W>& R = *c.ring;
    hipError_t e;
    switch (c.op) {
    case OP_CRT: {
        if constexpr ((sizeof(W) == 4 && LOGN == 15) || (sizeof(W) == 8 && LOGN == 14)) {
            if (c.opts.crt_half) {             // 128-KiB polynomials: two half-size workgroups per CU (kernel_crt_half.hpp)
                auto k = k_crt_half<LOGN, W, false>;
                const size_t hb = (size_t)lds_words<LOGN - 1>() * sizeof(W);
                if ((e = set_lds(k, hb)) != hipSuccess) return e;
                hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(1 << CrtHalfGeo<LOGN, W>::LT), hb, c.stream, R, c.data, c.src, c.first_poly);
                break;
            }
        }
        auto k = k_crt<LOGN, W, false>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(G::T), lds_bytes, c.stream, R, c.data, c.src, c.first_poly);
        break;
    }
    case OP_CRTINV: {
        if constexpr ((sizeof(W) == 4 && LOGN == 15) || (sizeof(W) == 8 && LOGN == 14)) {
            if (c.opts.crt_half) {
                auto k = k_crt_half<LOGN, W, true>;
                const size_t hb = (size_t)lds_words<LOGN - 1>() * sizeof(W);
                if ((e = set_lds(k, hb)) != hipSuccess) return e;
                hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(1 << CrtHalfGeo<LOGN, W>::LT), hb, c.stream, R, c.data, c.src, c.first_poly);
                break;
            }
        }
        auto k = k_crt<LOGN, W, true>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(G::T), lds_bytes, c.stream, R, c.data, c.src, c.first_poly);
        break;
    }
    case OP_TENSOR_INTT: {
        if constexpr (std::is_same<W, u32>::value && LOGN == 15) {
            // default: two sequential half-size sub-transforms, two workgroups per CU (kernel_tensor_split.hpp): +0.5..1 %
            // on the op, +2 % on the full mul_ against the whole-polynomial kernel below (ALCH_TI_SPLIT=0 selects that;
            // a value > 1 = that many persistent workgroups)
            const int split = c.opts.ti_split;
            if (split) {
                auto k = c.q30 ? k_tensor_intt_split<LOGN, true> : k_tensor_intt_split<LOGN, false>;
                const size_t half_lds = (size_t)lds_words<LOGN - 1>() * sizeof(W);
                if ((e = set_lds(k, half_lds)) != hipSuccess) return e;
                const unsigned nitems = (unsigned)(c.nct * (size_t)R.L);
                const unsigned grid = split > 1 && (unsigned)split < nitems ? (unsigned)split : nitems;
#if ALCH_A_PARTIALS
                hipLaunchKernelGGL(k, dim3(grid), dim3(1 << (LOGN - 6)), half_lds, c.stream, R, c.a, c.b, (int32_t*)c.digits, nitems,
                                   c.spre_r2, c.hint, c.partials ? c.out : nullptr);   // only alch_ct_mul_relin's launch pair asks for them
#else
                hipLaunchKernelGGL(k, dim3(grid), dim3(1 << (LOGN - 6)), half_lds, c.stream, R, c.a, c.b, (int32_t*)c.digits, nitems,
                                   c.spre_r2);
#endif
                break;
            }
        }
        auto k = k_tensor_intt<LOGN, W>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        const unsigned nitems = (unsigned)(c.nct * (size_t)R.L);
        // resident workgroups per CU: LDS (160 KiB), waves (32 per CU), at most 8
        unsigned per_cu = (unsigned)(163840 / (lds_bytes ? lds_bytes : 1));
        const unsigned by_waves = 32u / (unsigned)((G::T + 63) / 64);
        if (per_cu > by_waves) per_cu = by_waves;
        if (per_cu > 8) per_cu = 8;
        if (per_cu < 1) per_cu = 1;
        // ALCH_TI_GRID: -1 = one resident set of persistent workgroups (default: kernel time -4 % once the next
        // item's loads are issued behind the last vector-twiddle pass), 0 = one workgroup per item, n > 0 = n
        // persistent workgroups
        const int ti_grid = c.opts.ti_grid;
        unsigned grid = nitems < 256u * per_cu ? nitems : 256u * per_cu;
        if (ti_grid == 0) grid = nitems;
        else if (ti_grid > 0 && (unsigned)ti_grid < nitems) grid = (unsigned)ti_grid;
#ifdef ALCH_ABLATE
        static const unsigned dbg_a = getenv("ALCH_EXP_A") ? (unsigned)atoi(getenv("ALCH_EXP_A")) : 0u;   // diagnostic builds only
#else
        constexpr unsigned dbg_a = 0u;
#endif
        hipLaunchKernelGGL(k, dim3(grid), dim3(G::T), lds_bytes, c.stream, R, c.a, c.b, (SW*)c.digits, nitems,
                           c.spre_r2, dbg_a);
        break;
    }
    case OP_KS_ACCUM: {
        if constexpr (std::is_same<W, u32>::value && (LOGN == 15 || LOGN == 11)) {
            // two workgroups per (ciphertext, limb): see kernel_ks_half.hpp
#ifdef ALCH_ABLATE
            static const unsigned dbg_env = getenv("ALCH_EXP_FLAGS") ? (unsigned)strtoul(getenv("ALCH_EXP_FLAGS"), nullptr, 0) : 0u;   // diagnostic builds only: 1 inputs, 2 digits, 4 outputs, 8 hints aliased (wrong results, timing only)
#else
            constexpr unsigned dbg_env = 0u;
#endif
            const unsigned dbg_mask = dbg_env | ((c.opts.ks_map && 8 % R.L == 0) ? 0x80000000u : 0u) | (c.opts.ks_rev ? 0x40000000u : 0u);
            if ((size_t)c.nct * 2 * (size_t)R.L * G::N * sizeof(W) >= ((size_t)1 << 32)) return hipErrorInvalidValue;   // 32-bit byte offsets
            const size_t groups = (c.nct + 7) / 8;
            const unsigned nitems = (unsigned)(groups * 16 * (size_t)R.L);
            const unsigned persist = c.opts.ks_grid ? c.opts.ks_grid : 4096u;
            const unsigned grid = nitems < persist ? nitems : persist;
            constexpr int TH = 1 << (LOGN - 6);
            const size_t half_lds = (size_t)lds_words<LOGN - 1>() * sizeof(W);
            if (c.q30 && c.balanced) {                       // every modulus below 2^30: Harvey butterflies, lazy accumulators
                auto k = c.dup > 0 ? k_ks_accum_half<LOGN, true, 32, true, true> : k_ks_accum_half<LOGN, true, 32, false, true>;
                if ((e = set_lds(k, half_lds)) != hipSuccess) return e;
                hipLaunchKernelGGL(k, dim3(grid), dim3(TH), half_lds, c.stream, R, c.a, c.b, (const int32_t*)c.digits,
                                   c.hint, c.out, (unsigned)c.nct, nitems, c.spre_r2, dbg_mask, c.dup);
            } else if (c.dup > 0) {
                if (c.balanced) {
                    auto k = k_ks_accum_half<LOGN, true, 32, true>;
                    if ((e = set_lds(k, half_lds)) != hipSuccess) return e;
                    hipLaunchKernelGGL(k, dim3(grid), dim3(TH), half_lds, c.stream, R, c.a, c.b, (const int32_t*)c.digits,
                                       c.hint, c.out, (unsigned)c.nct, nitems, c.spre_r2, dbg_mask, c.dup);
                } else {
                    auto k = k_ks_accum_half<LOGN, false, 32, true>;
                    if ((e = set_lds(k, half_lds)) != hipSuccess) return e;
                    hipLaunchKernelGGL(k, dim3(grid), dim3(TH), half_lds, c.stream, R, c.a, c.b, (const int32_t*)c.digits,
                                       c.hint, c.out, (unsigned)c.nct, nitems, c.spre_r2, dbg_mask, c.dup);
                }
            } else if (c.balanced) {
                auto k = k_ks_accum_half<LOGN, true>;
                if ((e = set_lds(k, half_lds)) != hipSuccess) return e;
                hipLaunchKernelGGL(k, dim3(grid), dim3(TH), half_lds, c.stream, R, c.a, c.b, (const int32_t*)c.digits,
                                   c.hint, c.out, (unsigned)c.nct, nitems, c.spre_r2, dbg_mask, 0);
            } else {
                auto k = k_ks_accum_half<LOGN, false>;
                if ((e = set_lds(k, half_lds)) != hipSuccess) return e;
                hipLaunchKernelGGL(k, dim3(grid), dim3(TH), half_lds, c.stream, R, c.a, c.b, (const int32_t*)c.digits,
                                   c.hint, c.out, (unsigned)c.nct, nitems, c.spre_r2, dbg_mask, 0);
            }
            break;
        }
        const size_t groups = (c.nct + 7) / 8;
        const unsigned grid = (unsigned)(groups * 8 * (size_t)R.L);
        if (c.balanced) {
            auto k = k_ks_accum<LOGN, W, true>;
            if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
            hipLaunchKernelGGL(k, dim3(grid), dim3(G::T), lds_bytes, c.stream, R, c.a, c.b, (const SW*)c.digits,
                               c.hint, c.out, (unsigned)c.nct, c.spre_r2, c.dup);
        } else {
            auto k = k_ks_accum<LOGN, W, false>;
            if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
            hipLaunchKernelGGL(k, dim3(grid), dim3(G::T), lds_bytes, c.stream, R, c.a, c.b, (const SW*)c.digits,
                               c.hint, c.out, (unsigned)c.nct, c.spre_r2, c.dup);
        }
        break;
    }
    case OP_CRT_BASE2: {
        auto k = k_crt_base2_digits<LOGN, W>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(G::T), lds_bytes, c.stream, R, c.src, c.data, c.b2_first, c.b2_kd, c.b2_D);
        break;
    }
    case OP_RESCALE_OUT: {
        if constexpr (sizeof(W) == 4 && LOGN == 15) {
            const bool lin_serves = c.drop.ddn == 1 || (c.drop.ddn == 2 && c.drop.balanced);
            if (!c.pow_out && c.opts.rs_lin && (c.opts.rs_half || !lin_serves)) {
                // two half-size workgroups per CU; c.stash holds [2 nct][ddn][n] lifted residues
                const unsigned nitems = (unsigned)(c.nct * 2);
                const size_t hb = (size_t)lds_words<LOGN - 1>() * sizeof(W);
                auto k1 = k_rescale_drop_half<LOGN, W>;
                auto k2 = k_rescale_keep_half<LOGN, W>;
                if ((e = set_lds(k1, hb)) != hipSuccess) return e;
                if ((e = set_lds(k2, hb)) != hipSuccess) return e;
                hipLaunchKernelGGL(k1, dim3(nitems), dim3(1 << CrtHalfGeo<LOGN, W>::LT), hb, c.stream, R, c.a, (SW*)c.stash, c.drop);
                if ((e = hipGetLastError()) != hipSuccess) return e;
                hipLaunchKernelGGL(k2, dim3((nitems + 7u) / 8u * 8u * (unsigned)(R.L - c.drop.ddn)), dim3(1 << CrtHalfGeo<LOGN, W>::LT), hb, c.stream,
                                   R, c.a, (const SW*)c.stash, c.out, c.drop, nitems);
                break;
            }
        }
        if (!c.pow_out && c.opts.rs_lin && (c.drop.ddn == 1 || (c.drop.ddn == 2 && c.drop.balanced))) {   // (unbalanced two-limb drops would spill)
            // kept limbs stay in the CRT basis: ddn inverse + (L - ddn) forward transforms per component (kernel_rescale_out.hpp)
            const unsigned nitems = (unsigned)(c.nct * 2);
            const unsigned grid = nitems < c.stash_slots ? nitems : c.stash_slots;
            auto go = [&](auto k) -> hipError_t {
                hipError_t e2 = set_lds(k, lds_bytes);
                if (e2 != hipSuccess) return e2;
                hipLaunchKernelGGL(k, dim3(grid), dim3(G::T), lds_bytes, c.stream, R, c.a, c.out, nitems, c.drop);
                return hipSuccess;
            };
            if (c.drop.ddn == 1) e = c.drop.balanced ? go(k_rescale_out_lin<LOGN, W, 1, true>) : go(k_rescale_out_lin<LOGN, W, 1, false>);
            else e = go(k_rescale_out_lin<LOGN, W, 2, true>);
            if (e != hipSuccess) return e;
            break;
        }
        auto k = k_rescale_out<LOGN, W>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        const unsigned nitems = (unsigned)(c.nct * 2);
        const unsigned grid = nitems < c.stash_slots ? nitems : c.stash_slots;
        hipLaunchKernelGGL(k, dim3(grid), dim3(G::T), lds_bytes, c.stream, R, c.a, c.out, (SW*)c.stash, nitems, c.drop,
                           c.pow_out ? 1 : 0);
        break;
    }
    }
    return hipGetLastError();
}

// One of these per translation unit (the unrolled radix-16 passes make each instantiation large, so
// the LOGN range is split over several .hip files that build in parallel).
hipError_t dispatch32_small(int logn, const NttCall<u32>& c);   // log n = 4..9
hipError_t dispatch32_mid(int logn, const NttCall<u32>& c);     // 10..13
hipError_t dispatch32_14(int logn, const NttCall<u32>& c);
hipError_t dispatch32_15(int logn, const NttCall<u32>& c);
hipError_t dispatch64_small(int logn, const NttCall<u64>& c);   // 4..11
hipError_t dispatch64_big(int logn, const NttCall<u64>& c);     // 12..14
hipError_t dispatch32_16(int logn, const NttCall<u32>& c);      // split transform: crt / crtInv only
hipError_t dispatch64_15(int logn, const NttCall<u64>& c);

}  // namespace alch
