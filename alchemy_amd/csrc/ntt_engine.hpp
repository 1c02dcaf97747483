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

template <int LOGN>
struct Geo {
    static_assert(LOGN >= 4 && LOGN <= 15, "ring dimension 16 .. 32768 per LDS-resident transform");
    static constexpr int N = 1 << LOGN;
    static constexpr int LOGT = (LOGN - 4 > 10) ? 10 : (LOGN - 4);
    static constexpr int T = 1 << LOGT;               // threads per workgroup
    static constexpr int E = N / T;                   // coefficients per thread per pass (16 or 32)
    static constexpr int NPASS = (LOGN + 3) / 4;
    static constexpr int NS0 = LOGN - 4 * (NPASS - 1);   // stages of the first forward pass (1..4)
    static constexpr bool SWZ = LOGN >= 10;
};

// Logical coefficient index -> LDS word index.  Only bits >= 2 move, so aligned groups of four words
// stay contiguous (128-bit accesses remain legal).
template <int LOGN>
__device__ __forceinline__ int swz(int idx) {
    if constexpr (Geo<LOGN>::SWZ) return idx ^ (((idx >> 6) & 3) << 2) ^ (((idx >> 8) & 3) << 4);
    else return idx;
}

template <typename W> struct Vec4;
template <> struct Vec4<u32> { typedef u32 type __attribute__((ext_vector_type(4))); static constexpr int LANES = 4; };
template <> struct Vec4<u64> { typedef u64 type __attribute__((ext_vector_type(2))); static constexpr int LANES = 2; };

// ---- one register-resident pass ------------------------------------------------------------------
// Loads the thread's groups from LDS, runs NS stages, writes them back (unless KEEP, in which case the
// results of the last pass stay in regs[][] for a fused epilogue and LDS is left stale).
struct NoEpilogue {
    template <typename W> __device__ __forceinline__ void operator()(int, int, W*) const {}
};

// When KEEP, nothing is written back to LDS: epi(g, base, x) receives each finished group while its R
// values x[0..R) are still in registers (x[k] = logical index base | k << LB, lazy in [0,2q)), so a fused
// epilogue never needs more than one group live.
template <int LOGN, typename W, int S0, int NS, bool INVERSE, bool KEEP, bool SERIAL, typename Epi>
__device__ __forceinline__ void ntt_pass(W* __restrict__ lds, const W* __restrict__ tw, W q, W qni,
                                         W ninv_m, W w1ninv_m, int t, Epi&& epi) {
    typedef Geo<LOGN> G;
    constexpr int R = 1 << NS;
    constexpr int LB = LOGN - S0 - NS;
    constexpr int NG = G::E / R;                     // groups per thread
    static_assert(NG >= 1, "group larger than the per-thread coefficient budget");
    typedef typename Vec4<W>::type V;
    constexpr int VL = Vec4<W>::LANES;

#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int gamma = t + G::T * g;
        const int lo = gamma & ((1 << LB) - 1);
        const int h = gamma >> LB;
        const int base = (h << (NS + LB)) | lo;
        W x[R];
        // ---- load
        if constexpr (LB == 0 && R >= VL) {
#pragma unroll
            for (int k = 0; k < R; k += VL) {
                V v = *reinterpret_cast<const V*>(&lds[swz<LOGN>(base + k)]);
#pragma unroll
                for (int e = 0; e < VL; ++e) x[k + e] = v[e];
            }
        } else {
#pragma unroll
            for (int k = 0; k < R; ++k) x[k] = lds[swz<LOGN>(base | (k << LB))];
        }
        // ---- butterflies
        const int gm = (1 << S0) + h;
        if constexpr (!INVERSE) {
#pragma unroll
            for (int r = 0; r < NS; ++r) {
                const int half = R >> (r + 1);
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    if (k & half) continue;
                    const W w = tw[(gm << r) + (k >> (NS - r))];
                    bfly_fwd(x[k], x[k + half], w, q, qni);
                }
            }
        } else {
#pragma unroll
            for (int r = NS - 1; r >= 0; --r) {
                const int half = R >> (r + 1);
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    if (k & half) continue;
                    if (S0 == 0 && r == 0) {
                        // last stage of crtInv (single twiddle tw[1]): fold in n^-1
                        W a = csub(x[k], q), b = csub(x[k + half], q);
                        x[k] = mont_mul_lazy((W)(a + b), ninv_m, q, qni);
                        x[k + half] = mont_mul_lazy((W)(a - b + q), w1ninv_m, q, qni);
                    } else {
                        const W w = tw[(gm << r) + (k >> (NS - r))];
                        bfly_inv(x[k], x[k + half], w, q, qni);
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
                    *reinterpret_cast<V*>(&lds[swz<LOGN>(base + k)]) = v;
                }
            } else {
#pragma unroll
                for (int k = 0; k < R; ++k) lds[swz<LOGN>(base | (k << LB))] = x[k];
            }
            // One group at a time: callers (k_ks_accum) hold 2*E accumulators across the transform, and
            // interleaving groups would push them over the 128-VGPR budget of a 1024-thread workgroup.
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
template <int LOGN, typename W, bool KEEP_LAST, bool SERIAL = false, typename Epi>
__device__ __forceinline__ void ntt_forward(W* lds, const W* tw, W q, W qni, int tid, Epi&& epi) {
    typedef Geo<LOGN> G;
    constexpr int P = G::NPASS, F = G::NS0;
    NoEpilogue none;
    if constexpr (P == 1) {
        ntt_pass<LOGN, W, 0, F, false, KEEP_LAST, SERIAL>(lds, tw, q, qni, (W)0, (W)0, tid, epi);
    } else {
        ntt_pass<LOGN, W, 0, F, false, false, SERIAL>(lds, tw, q, qni, (W)0, (W)0, tid, none);
        __syncthreads();
        if constexpr (P == 2) {
            ntt_pass<LOGN, W, F, 4, false, KEEP_LAST, SERIAL>(lds, tw, q, qni, (W)0, (W)0, tid, epi);
        } else {
            ntt_pass<LOGN, W, F, 4, false, false, SERIAL>(lds, tw, q, qni, (W)0, (W)0, tid, none);
            __syncthreads();
            if constexpr (P == 3) {
                ntt_pass<LOGN, W, F + 4, 4, false, KEEP_LAST, SERIAL>(lds, tw, q, qni, (W)0, (W)0, tid, epi);
            } else {
                ntt_pass<LOGN, W, F + 4, 4, false, false, SERIAL>(lds, tw, q, qni, (W)0, (W)0, tid, none);
                __syncthreads();
                ntt_pass<LOGN, W, F + 8, 4, false, KEEP_LAST, SERIAL>(lds, tw, q, qni, (W)0, (W)0, tid, epi);
            }
        }
    }
    if constexpr (!KEEP_LAST) __syncthreads();
}

// crtInv: CRT slots in LDS -> coefficients, n^-1 applied.  With KEEP_LAST the final pass (stages [0, NS0),
// groups of R = 2^NS0 coefficients of stride n/R) hands each group to epi(g, base, x):
// x[k] = coefficient base + k * (n/R), lazy in [0,2q).
template <int LOGN, typename W, bool KEEP_LAST, bool SERIAL = false, typename Epi>
__device__ __forceinline__ void ntt_inverse(W* lds, const W* twi, W q, W qni, W ninv_m, W w1ninv_m, int tid,
                                            Epi&& epi) {
    typedef Geo<LOGN> G;
    constexpr int P = G::NPASS, F = G::NS0;
    NoEpilogue none;
    if constexpr (P >= 4) { ntt_pass<LOGN, W, F + 8, 4, true, false, SERIAL>(lds, twi, q, qni, ninv_m, w1ninv_m, tid, none); __syncthreads(); }
    if constexpr (P >= 3) { ntt_pass<LOGN, W, F + 4, 4, true, false, SERIAL>(lds, twi, q, qni, ninv_m, w1ninv_m, tid, none); __syncthreads(); }
    if constexpr (P >= 2) { ntt_pass<LOGN, W, F, 4, true, false, SERIAL>(lds, twi, q, qni, ninv_m, w1ninv_m, tid, none); __syncthreads(); }
    ntt_pass<LOGN, W, 0, F, true, KEEP_LAST, SERIAL>(lds, twi, q, qni, ninv_m, w1ninv_m, tid, epi);
    if constexpr (!KEEP_LAST) __syncthreads();
}

}  // namespace alch
