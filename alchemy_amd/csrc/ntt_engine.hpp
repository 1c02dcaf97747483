// Negacyclic NTT engine: one workgroup transforms one limb-polynomial that lives entirely in LDS.
//
// Serves Lol's Tensor `crt` / `crtInv` for a two-power index (the transforms under every Cyc product in
// SymmSHE's (*) and keySwitchQuadCirc, Crypto/Alchemy/Interpreter/Eval.hs:65-67,133).
//
// Algorithm.  crt = merged-twist Cooley-Tukey: stage s (s = 0..log n - 1) has 2^s groups of stride
// t = n / 2^(s+1); group i multiplies its upper half by tw[2^s + i] with tw[k] = psi^brev_logn(k)
// (natural order in, bit-reversed evaluation order out: slot k = a(psi^(2 brev(k)+1))).  crtInv runs the
// same stages backwards with Gentleman-Sande butterflies and tw^-1, folding n^-1 into the last stage.
//
// Mapping to CDNA4.  A limb-polynomial of n = 2^15 32-bit residues is 128 KiB and fits the 160 KiB LDS
// of one CU, so a transform costs one HBM read and one HBM write.  T = n/32 (<= 1024) threads each own
// E = n/T coefficients per pass and run up to four butterfly stages on them in registers (radix-16
// groups), so the 15 stages take four LDS round trips instead of fifteen.  A pass over stages
// [S0, S0+NS) owns groups of R = 2^NS coefficients {base + K * 2^LB} (LB = log n - S0 - NS): lanes
// walk consecutive low bits, so LDS accesses are lane-contiguous, and the last pass (LB = 0) owns
// R contiguous words per lane and moves them with 128-bit LDS accesses.  The XOR swizzle swz() keeps
// every one of those access patterns bank-conflict free without padding.
#pragma once
#include "modarith.hpp"

namespace alch {

template <int LOGN, int LOGT_ = -1>
struct Geo {
    static_assert(LOGN >= 4 && LOGN <= 15, "ring dimension 16 .. 32768 per LDS-resident transform");
    static constexpr int N = 1 << LOGN;
    static constexpr int LOGT = LOGT_ >= 0 ? LOGT_ : ((LOGN - 4 > 10) ? 10 : (LOGN - 4));
    static constexpr int T = 1 << LOGT;               // threads per workgroup
    static constexpr int E = N / T;                   // coefficients per thread per pass (16 or 32)
    static constexpr int NPASS = (LOGN + 3) / 4;
    static constexpr int NS0 = LOGN - 4 * (NPASS - 1);   // stages of the first forward pass (1..4)
    static constexpr bool SWZ = LOGN >= 10;
};

// Logical coefficient index -> LDS word index.
// ALCH_LDS_PAD = 0: XOR swizzle, no extra LDS.  Only bits >= 2 move, so aligned groups of four words stay
//   contiguous; linear over XOR (swz(a ^ b) = swz(a) ^ swz(b)), so a pass swizzles its group base once and
//   reaches element k with one v_xor against the compile-time constant swz(k << LB).
// ALCH_LDS_PAD = 1: four padding words after every 64 (LDS grows by 1/16).  Additive over operands with
//   disjoint bits (pad(a | b) = pad(a) + pad(b)), so element k is the group base plus a compile-time constant,
//   which the LDS instructions take as an immediate offset: no address arithmetic per access.
// Both are bank-conflict free for the four access shapes the passes use (LB = 0: 16 contiguous words per lane,
// 128-bit accesses; LB = 4: runs of 16 lanes 256 words apart; LB >= 6: 64 consecutive words; lane-contiguous
// 16-byte pieces): see DESIGN.md.
#ifndef ALCH_LDS_PAD
#define ALCH_LDS_PAD 1
#endif
template <int LOGN>
__host__ __device__ constexpr int swz(int idx) {
#if ALCH_LDS_PAD
    return (LOGN >= 10) ? idx + ((idx >> 6) << 2) : idx;
#else
    return (LOGN >= 10) ? (idx ^ (((idx >> 6) & 3) << 2) ^ (((idx >> 8) & 3) << 4)) : idx;
#endif
}
// address of element (base | k << LB) from the mapped base and the mapped constant
__host__ __device__ constexpr int lds_join(int mapped_base, int mapped_k) {
#if ALCH_LDS_PAD
    return mapped_base + mapped_k;
#else
    return mapped_base ^ mapped_k;
#endif
}
// words of LDS a 2^LOGN-point transform needs
template <int LOGN>
__host__ __device__ constexpr int lds_words() {
#if ALCH_LDS_PAD
    return (LOGN >= 10) ? (1 << LOGN) + (1 << (LOGN - 4)) : (1 << LOGN);
#else
    return 1 << LOGN;
#endif
}

template <typename W> struct Vec4;
template <> struct Vec4<u32> { typedef u32 type __attribute__((ext_vector_type(4))); static constexpr int LANES = 4; };
template <> struct Vec4<u64> { typedef u64 type __attribute__((ext_vector_type(2))); static constexpr int LANES = 2; };

// Store-data guard for 16-byte buffer stores.  A buffer_store_dwordx4 reads its four data VGPRs over several cycles.
// hipcc (ROCm 7.2, gfx950) inserts no wait state behind such a store when it carries an SGPR offset, and a VALU
// instruction issued right behind it that writes the first data register was observed to win the race: k_rescale_out_lin
// lost element 12 of lanes 12-15 / 28-31 / 44-47 / 60-63 of one wave in ~0.5 % of its polynomials until its store data
// were kept live across one s_nop (found by the whole-batch checksum of tests/test_gpu_bench_shape.py).  Every 16-byte
// buffer store in this library is followed by this guard; it costs one issue slot.
#ifndef ALCH_NO_STORE_GUARD
#define ALCH_STORE_GUARD(v) asm volatile("s_nop 0" ::"v"(v))
#else
#define ALCH_STORE_GUARD(v) ((void)0)   // lint self-test only (tests/test_store_hazard_lint.py): never in a product build
#endif

// Workgroup barrier for LDS hand-offs only.  __syncthreads() is also a fence for global memory, so hipcc puts
// `s_waitcnt vmcnt(0)` in front of it whenever vector-memory operations are outstanding -- which drains every
// prefetch (register loads for the next work item, LDS-DMA touches) at the next pass boundary.  The transforms
// exchange data through LDS only, so they wait for LDS traffic alone and let global loads stay in flight.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- twiddles of one butterfly stage ---------------------------------------------------------------
// Local stage r of a pass needs the CNT = 2^r consecutive table words tw[first .. first+CNT), first a
// multiple of CNT: one vector load per four words instead of one dword load per butterfly.  When every
// lane of the wave wants the same words (UNIFORM), the index is made wave-uniform so the loads become
// scalar (s_load) and the twiddles live in SGPRs.
template <typename W, int CNT, bool UNIFORM>
__device__ __forceinline__ void load_tw(const W* __restrict__ tw, int first, W (&w)[CNT]) {   // W: table word
    if constexpr (UNIFORM) {
        const int f = __builtin_amdgcn_readfirstlane(first);
#pragma unroll
        for (int c = 0; c < CNT; ++c) w[c] = tw[f + c];
    } else if constexpr (sizeof(W) == 16) {               // two-word Shoup twiddles: one 16-byte load each
#pragma unroll
        for (int c = 0; c < CNT; ++c) w[c] = tw[first + c];
    } else if constexpr (sizeof(W) == 4 && CNT >= 4) {
        typedef u32 V __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int c = 0; c < CNT; c += 4) {
            V v = *reinterpret_cast<const V*>(tw + first + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) w[c + e] = v[e];
        }
    } else if constexpr (CNT >= 2) {
        typedef W V __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int c = 0; c < CNT; c += 2) {
            V v = *reinterpret_cast<const V*>(tw + first + c);
            w[c] = v[0];
            w[c + 1] = v[1];
        }
    } else {
        w[0] = tw[first];
    }
}

// ---- one register-resident pass ------------------------------------------------------------------
struct NoEpilogue {
    template <typename W> __device__ __forceinline__ void operator()(int, int, W*) const {}
};

// Runs stages [S0, S0+NS) of a 2^LOGN-point transform on the thread's groups: loads them from LDS, NS
// butterfly stages in registers, writes them back.  With KEEP nothing is written back: epi(g, base, x)
// receives each finished group while its R values are still in registers (x[k] = logical index
// base | k << LB, lazy in [0,2q)), so a fused epilogue never needs more than one group live.
// `prefix`: the transform may be a sub-transform of a larger one whose leading stages ran elsewhere
// (k_ks_accum_half: prefix = 2 + half); group h of stage s then uses twiddle (prefix << s) + h.  A whole
// transform has prefix 1.
// TW is the twiddle table's word type: W for Montgomery twiddles, u64 for the Plantard constants the
// forward transform of 32-bit rings uses (bfly_fwd overloads pick the arithmetic from it).
// FOLD: an inverse pass that contains stage 0 folds n^-1 into it; a sub-transform (prefix != 1) whose local
// stage 0 is not the transform's stage 0 passes FOLD = false.
// Q30 (32-bit rings whose moduli are all below 2^30, Montgomery twiddles): Harvey's butterflies -- forward values stay lazy in [0,4q)
// from pass to pass (callers' epilogues multiply them, which takes any word, or reduce them), inverse values in [0,2q) as otherwise.
template <int LOGN, int LOGT, typename W, int S0, int NS, bool INVERSE, bool KEEP, bool SERIAL, typename TW, typename Epi,
          bool FOLD = true, bool Q30 = false>
__device__ __forceinline__ void ntt_pass(W* __restrict__ lds, const TW* __restrict__ tw, W q, W qni,
                                         W ninv_m, W w1ninv_m, int t, int prefix, Epi&& epi) {
    typedef Geo<LOGN, LOGT> G;
    constexpr int R = 1 << NS;
    constexpr int LB = LOGN - S0 - NS;
    constexpr int NG = G::E / R;                     // groups per thread
    static_assert(NG >= 1, "group larger than the per-thread coefficient budget");
    static_assert(NS >= 1 && NS <= 5, "one to five stages per pass");
    typedef typename Vec4<W>::type V;
    constexpr int VL = Vec4<W>::LANES;
    // all 64 lanes of a wave share h when a run of equal h covers a whole wave
    constexpr bool UNIFORM = (LB >= 6) && (G::T >= 64);

#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int gamma = t + G::T * g;
        const int lo = gamma & ((1 << LB) - 1);
        const int h = gamma >> LB;
        const int base = (h << (NS + LB)) | lo;
        const int sb = swz<LOGN>(base);
        W x[R];
        // ---- load
        if constexpr (LB == 0 && R >= VL) {
#pragma unroll
            for (int k = 0; k < R; k += VL) {
                V v = *reinterpret_cast<const V*>(&lds[lds_join(sb, swz<LOGN>(k))]);
#pragma unroll
                for (int e = 0; e < VL; ++e) x[k + e] = v[e];
            }
        } else {
#pragma unroll
            for (int k = 0; k < R; ++k) x[k] = lds[lds_join(sb, swz<LOGN>(k << LB))];
        }
        // ---- butterflies
        const int gm = (prefix << S0) + h;
        if constexpr (!INVERSE) {
#pragma unroll
            for (int r = 0; r < NS; ++r) {
                const int half = R >> (r + 1);
                TW w[1 << 4];                    // up to five stages per pass (radix 32)
                if (r == 0) { TW t1[1]; load_tw<TW, 1, UNIFORM>(tw, gm, t1); w[0] = t1[0]; }
                else if (r == 1) { TW t2[2]; load_tw<TW, 2, UNIFORM>(tw, gm << 1, t2); w[0] = t2[0]; w[1] = t2[1]; }
                else if (r == 2) { TW t4[4]; load_tw<TW, 4, UNIFORM>(tw, gm << 2, t4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) w[c] = t4[c]; }
                else if (r == 3) { TW t8[8]; load_tw<TW, 8, UNIFORM>(tw, gm << 3, t8);
#pragma unroll
                    for (int c = 0; c < 8; ++c) w[c] = t8[c]; }
                else { TW t16[16]; load_tw<TW, 16, UNIFORM>(tw, gm << 4, t16);
#pragma unroll
                    for (int c = 0; c < 16; ++c) w[c] = t16[c]; }
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    if (k & half) continue;
                    if constexpr (Q30) bfly_fwd4(x[k], x[k + half], w[k >> (NS - r)], q, qni);
                    else bfly_fwd_st(x[k], x[k + half], w[k >> (NS - r)], q, qni, r == 0, r == NS - 1);
                }
            }
        } else {
#pragma unroll
            for (int r = NS - 1; r >= 0; --r) {
                const int half = R >> (r + 1);
                TW w[1 << 4];                    // up to five stages per pass (radix 32)
                if (r == 0) { TW t1[1]; load_tw<TW, 1, UNIFORM>(tw, gm, t1); w[0] = t1[0]; }
                else if (r == 1) { TW t2[2]; load_tw<TW, 2, UNIFORM>(tw, gm << 1, t2); w[0] = t2[0]; w[1] = t2[1]; }
                else if (r == 2) { TW t4[4]; load_tw<TW, 4, UNIFORM>(tw, gm << 2, t4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) w[c] = t4[c]; }
                else if (r == 3) { TW t8[8]; load_tw<TW, 8, UNIFORM>(tw, gm << 3, t8);
#pragma unroll
                    for (int c = 0; c < 8; ++c) w[c] = t8[c]; }
                else { TW t16[16]; load_tw<TW, 16, UNIFORM>(tw, gm << 4, t16);
#pragma unroll
                    for (int c = 0; c < 16; ++c) w[c] = t16[c]; }
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    if (k & half) continue;
                    if (FOLD && S0 == 0 && r == 0) {
                        // last stage of crtInv (single twiddle tw[1]): fold in n^-1
                        W a = csub(x[k], q), b = csub(x[k + half], q);
                        x[k] = mont_mul_lazy((W)(a + b), ninv_m, q, qni);
                        x[k + half] = mont_mul_lazy((W)(a - b + q), w1ninv_m, q, qni);
                    } else if constexpr (Q30) {
                        bfly_inv4(x[k], x[k + half], w[k >> (NS - r)], q, qni);
                    } else {
                        bfly_inv(x[k], x[k + half], w[k >> (NS - r)], q, qni);
                    }
                }
            }
        }
        // ---- store
        if constexpr (KEEP) {
            epi(g, base, x);
            if constexpr (NG > 1) __builtin_amdgcn_sched_barrier(0);
        } else {
            if constexpr (LB == 0 && R >= VL) {
#pragma unroll
                for (int k = 0; k < R; k += VL) {
                    V v;
#pragma unroll
                    for (int e = 0; e < VL; ++e) v[e] = x[k + e];
                    *reinterpret_cast<V*>(&lds[lds_join(sb, swz<LOGN>(k))]) = v;
                }
            } else {
#pragma unroll
                for (int k = 0; k < R; ++k) lds[lds_join(sb, swz<LOGN>(k << LB))] = x[k];
            }
            // One group at a time: callers (k_ks_accum*) hold 2*E accumulators across the transform, and
            // interleaving groups would push them over the 128-VGPR budget.
            if constexpr (NG > 1 && SERIAL) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- full transforms -----------------------------------------------------------------------------
// crt: data in LDS (logical order, values < 2q) -> CRT slots (bit-reversed evaluation order).
// KEEP_LAST = false: results end up back in LDS (a trailing __syncthreads() is included).
// KEEP_LAST = true : the final pass (stages [LOGN-4, LOGN); all stages when LOGN == 4) hands each group of
//                    16 consecutive slots to epi(g, base, x): x[k] = slot base + k, base = (tid + T*g)*16.
// tid = threadIdx.x (passed in so that a caller looping over transforms can make it opaque per iteration
// and stop the compiler from hoisting every pass's LDS addresses out of its loop).
// Callers must __syncthreads() after filling LDS; the function syncs between passes.
// The passes with LB = 4 and LB = 0 (the last two of crt, the first two of crtInv) give every wave the same 1024
// contiguous coefficients (64 lanes x 16: [1024 w, 1024 w + 1024) per group index), so the hand-off between them
// stays inside a wave: LDS executes one wave's instructions in order, no workgroup barrier is needed.
#ifndef ALCH_WAVE_LOCAL_PAIR
#define ALCH_WAVE_LOCAL_PAIR 1
#endif
template <int LOGN>
__device__ __forceinline__ void pair_sync() {
    if constexpr (ALCH_WAVE_LOCAL_PAIR && LOGN >= 10) asm volatile("" ::: "memory");
    else lds_barrier();
}

template <int LOGN, typename W, bool KEEP_LAST, bool SERIAL = false, typename TW, typename TWM, typename Epi>
__device__ __forceinline__ void ntt_forward(W* lds, const TW* tw, const TWM* twm, W q, W qni, int tid, Epi&& epi,
                                            int prefix = 1) {
    // prefix != 1: the transform is one half of a transform twice its size whose stage 0 ran elsewhere
    // (k_crt_split); tw / twm are then the tables of the big ring.
    // tw : twiddle table for the passes whose twiddles are shared by many lanes (Plantard constants on 32-bit
    //      rings: one instruction less per butterfly, fetched by scalar or broadcast loads);
    // twm: Montgomery table for the last pass, where every lane needs its own 15 twiddles and two-word
    //      constants would double the per-lane load traffic and register pressure (64-bit rings: Shoup pairs in every
    //      pass -- a two-word twiddle still beats a 27-instruction Montgomery product).
    typedef Geo<LOGN> G;
    constexpr int P = G::NPASS, F = G::NS0, LT = G::LOGT;
    NoEpilogue none;
    if constexpr (P == 1) {
        ntt_pass<LOGN, LT, W, 0, F, false, KEEP_LAST, SERIAL>(lds, twm, q, qni, (W)0, (W)0, tid, prefix, epi);
    } else {
        ntt_pass<LOGN, LT, W, 0, F, false, false, SERIAL>(lds, tw, q, qni, (W)0, (W)0, tid, prefix, none);
        lds_barrier();
        if constexpr (P == 2) {
            ntt_pass<LOGN, LT, W, F, 4, false, KEEP_LAST, SERIAL>(lds, twm, q, qni, (W)0, (W)0, tid, prefix, epi);
        } else {
            ntt_pass<LOGN, LT, W, F, 4, false, false, SERIAL>(lds, tw, q, qni, (W)0, (W)0, tid, prefix, none);
            if constexpr (P == 3) pair_sync<LOGN>(); else lds_barrier();
            if constexpr (P == 3) {
                ntt_pass<LOGN, LT, W, F + 4, 4, false, KEEP_LAST, SERIAL>(lds, twm, q, qni, (W)0, (W)0, tid, prefix, epi);
            } else {
                ntt_pass<LOGN, LT, W, F + 4, 4, false, false, SERIAL>(lds, tw, q, qni, (W)0, (W)0, tid, prefix, none);
                pair_sync<LOGN>();
                ntt_pass<LOGN, LT, W, F + 8, 4, false, KEEP_LAST, SERIAL>(lds, twm, q, qni, (W)0, (W)0, tid, prefix, epi);
            }
        }
    }
    if constexpr (!KEEP_LAST) lds_barrier();
}

// crtInv: CRT slots in LDS -> coefficients, n^-1 applied.  With KEEP_LAST the final pass (stages [0, NS0),
// groups of R = 2^NS0 coefficients of stride n/R) hands each group to epi(g, base, x):
// x[k] = coefficient base + k * (n/R), lazy in [0,2q).
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `hook()` runs once, right before the first pass whose twiddles are wave-uniform (scalar loads): from there on
// the transform issues no vector-memory loads, so global loads started in the hook (a prefetch for the next
// work item) are never waited for by this transform -- vmcnt retires in order, and a later twiddle load would
// otherwise drag the whole prefetch's HBM latency into the pass.
template <int LOGN, typename W, bool KEEP_LAST, bool SERIAL = false, bool FOLD = true, typename TWI, typename Epi, typename Hook = NoHook>
__device__ __forceinline__ void ntt_inverse(W* lds, const TWI* twi, W q, W qni, W ninv_m, W w1ninv_m, int tid,
                                            Epi&& epi, Hook&& hook = NoHook(), int prefix = 1) {
    // FOLD = false, prefix = 2 + half: the stages 1.. of a transform twice this size on one half of its slots
    // (k_crt_split); the caller runs stage 0 and the n^-1 scaling itself.
    typedef Geo<LOGN> G;
    constexpr int P = G::NPASS, F = G::NS0, LT = G::LOGT;
    NoEpilogue none;
    // pass over stages [S0, S0+4) has LB = LOGN - S0 - 4; uniform twiddles need LB >= 6 (and a full wave)
    constexpr bool U3 = (LOGN - (F + 8) - 4 >= 6), U2 = (LOGN - (F + 4) - 4 >= 6), U1 = (LOGN - F - 4 >= 6);
    bool hooked = false;
    if constexpr (P >= 4) { if (U3 && !hooked) { hook(); hooked = true; }
        ntt_pass<LOGN, LT, W, F + 8, 4, true, false, SERIAL, TWI, NoEpilogue&, FOLD>(lds, twi, q, qni, ninv_m, w1ninv_m, tid, prefix, none); pair_sync<LOGN>(); }
    if constexpr (P >= 3) { if (U2 && !hooked) { hook(); hooked = true; }
        ntt_pass<LOGN, LT, W, F + 4, 4, true, false, SERIAL, TWI, NoEpilogue&, FOLD>(lds, twi, q, qni, ninv_m, w1ninv_m, tid, prefix, none);
        if constexpr (P == 3) pair_sync<LOGN>(); else lds_barrier(); }
    if constexpr (P >= 2) { if (U1 && !hooked) { hook(); hooked = true; }
        ntt_pass<LOGN, LT, W, F, 4, true, false, SERIAL, TWI, NoEpilogue&, FOLD>(lds, twi, q, qni, ninv_m, w1ninv_m, tid, prefix, none); lds_barrier(); }
    if (!hooked) hook();
    ntt_pass<LOGN, LT, W, 0, F, true, KEEP_LAST, SERIAL, TWI, Epi&, FOLD>(lds, twi, q, qni, ninv_m, w1ninv_m, tid, prefix, epi);
    if constexpr (!KEEP_LAST) lds_barrier();
}

}  // namespace alch
