// The closing modSwitch of PT2CT's mul_ (Rescale (a,b) -> b on both components, reference Eval.hs:130, PT2CT.hs:177) for
// 128-KiB limb-polynomials (n = 2^15, 32-bit residues), as TWO launches of half-size workgroups -- two 8-wave workgroups per
// CU like the tensor and key-switch kernels in front of it, instead of k_rescale_out_lin's one 16-wave workgroup per CU.
// Serves the shapes k_rescale_out_lin does not (three dropped limbs, unbalanced two-limb drops: its residues would not fit the
// register file) with 10-ish instead of 16+ transforms; on the shapes both serve it is NOT faster (0.473 + 0.925 ms against
// 1.331 ms per 1024 ciphertexts at 5 -> 3 limbs: the stash round trip and the per-kept-limb recomputation of the combination
// cost what the second workgroup per CU gains), so the dispatcher prefers k_rescale_out_lin there (option rs_half = 1 forces this form).
//
// Same identity as k_rescale_out_lin (kernel_rescale_out.hpp): the surviving limbs never leave the CRT basis,
//     crt(z_t) = crt(x_t) C_t - crt(sum_u reduce_t(R_u) c_{u,t}),
// R_u = centred lift of the u-th dropped residue after the drops in front of it.
//   k_rescale_drop_half  per (ciphertext, component): for every dropped limb, outermost first, crtInv as two half-size
//                        sub-transforms (k_crt_half's inverse), the chain of earlier drops applied in the lane-local stage-0
//                        epilogue, lifted residues -> stash (signed words; every stash word a lane re-reads was written by
//                        that lane);
//   k_rescale_keep_half  per (ciphertext, component, kept limb): the combination formed from the stash in the loader of
//                        k_crt_half's forward transform (stage 0 pairs coefficient k with k + n/2), the epilogue reads
//                        crt(x_t) from HBM and stores x_t C_t - . to the output ring.
// Any number of dropped limbs up to MAXDROP, balanced or not (no per-lane residue arrays: the stash holds them).
#pragma once
#include <hip/hip_runtime.h>
#include "kernel_crt_half.hpp"

namespace alch {

template <typename W>
__device__ __forceinline__ W reduce_lifted(typename Signed<W>::type z, W q) {       // z mod q; |z| < q is the common case
    typedef typename Signed<W>::type SW;
    if (z < (SW)q && z > -(SW)q) return z < 0 ? (W)(z + (SW)q) : (W)z;
    SW r = z % (SW)q;
    return r < 0 ? (W)(r + (SW)q) : (W)r;
}

template <int LOGN, typename W>
__global__ void __launch_bounds__((1 << CrtHalfGeo<LOGN, W>::LT), 4)
k_rescale_drop_half(DevRing<W> R, const W* __restrict__ src, typename Signed<W>::type* stash, DropTab<W> D) {
    typedef CrtHalfGeo<LOGN, W> H;
    typedef typename Signed<W>::type SW;
    constexpr int LOGM = H::LOGM, M = 1 << LOGM, LT = H::LT, T = 1 << LT, F = H::F;
    typedef Geo<LOGM, LT> G;
    typedef typename Vec4<W>::type V;
    constexpr int VL = Vec4<W>::LANES, NV = G::E / VL;
    typedef SW SV __attribute__((ext_vector_type(VL)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L, ddn = D.ddn;
    const size_t item = blockIdx.x;                               // 2 ct + component
    const W* x = src + item * (size_t)L * (size_t)(2 * M);
    SW* st = stash + item * (size_t)ddn * (size_t)(2 * M);
    NoEpilogue none;
    V keep[NV];
#pragma unroll 1
    for (int u = 0; u < ddn; ++u) {
        const W q = R.mod[u].q, qni = R.mod[u].qni, hq = (q - 1) >> 1;
        const auto twi = inv_tw(R, u);
        typedef typename std::remove_cv<typename std::remove_pointer<decltype(twi)>::type>::type TWI;
        const W ninv = R.ninv_m[u], w1ninv = R.w1ninv_m[u];
        const W* in = x + (size_t)u * (2 * M);
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const W* hsrc = in + (half ? M : 0);
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const int idx = (tid + T * r) * VL;
                *reinterpret_cast<V*>(&lds[swz<LOGM>(idx)]) = *reinterpret_cast<const V*>(hsrc + idx);
            }
            lds_barrier();
            const int prefix = 2 + half;
            ntt_pass<LOGM, LT, W, F + 8, 4, true, false, false, TWI, NoEpilogue&, false>(lds, twi, q, qni, (W)0, (W)0, tid, prefix, none);
            pair_sync<LOGM>();
            ntt_pass<LOGM, LT, W, F + 4, 4, true, false, false, TWI, NoEpilogue&, false>(lds, twi, q, qni, (W)0, (W)0, tid, prefix, none);
            lds_barrier();
            ntt_pass<LOGM, LT, W, F, 4, true, false, false, TWI, NoEpilogue&, false>(lds, twi, q, qni, (W)0, (W)0, tid, prefix, none);
            lds_barrier();
            ntt_pass<LOGM, LT, W, 0, F, true, false, false, TWI, NoEpilogue&, false>(lds, twi, q, qni, (W)0, (W)0, tid, prefix, none);
            lds_barrier();
            if (half == 0) {
#pragma unroll
                for (int r = 0; r < NV; ++r) keep[r] = *reinterpret_cast<const V*>(&lds[swz<LOGM>((tid + T * r) * VL)]);
            } else {
                // stage 0 with n^-1 folded in, then the drops in front of this limb, then the centred lift: all lane-local
#pragma unroll
                for (int r = 0; r < NV; ++r) {
                    const int idx = (tid + T * r) * VL;
                    const V hi = *reinterpret_cast<const V*>(&lds[swz<LOGM>(idx)]);
                    V c0, c1;
#pragma unroll
                    for (int e = 0; e < VL; ++e) {
                        const W a = csub(keep[r][e], q), b = csub(hi[e], q);
                        c0[e] = csub(mont_mul_lazy((W)(a + b), ninv, q, qni), q);
                        c1[e] = csub(mont_mul_lazy((W)(a - b + q), w1ninv, q, qni), q);
                    }
                    for (int v = 0; v < u; ++v) {
                        const W qim = D.qinv_m[v][u];
                        const SV z0 = *reinterpret_cast<const SV*>(st + (size_t)v * (2 * M) + idx);
                        const SV z1 = *reinterpret_cast<const SV*>(st + (size_t)v * (2 * M) + M + idx);
#pragma unroll
                        for (int e = 0; e < VL; ++e) {
                            c0[e] = csub(mont_mul_lazy((W)(c0[e] + (q - reduce_lifted<W>(z0[e], q))), qim, q, qni), q);
                            c1[e] = csub(mont_mul_lazy((W)(c1[e] + (q - reduce_lifted<W>(z1[e], q))), qim, q, qni), q);
                        }
                    }
                    SV l0, l1;
#pragma unroll
                    for (int e = 0; e < VL; ++e) {
                        l0[e] = c0[e] > hq ? (SW)c0[e] - (SW)q : (SW)c0[e];
                        l1[e] = c1[e] > hq ? (SW)c1[e] - (SW)q : (SW)c1[e];
                    }
                    *reinterpret_cast<SV*>(st + (size_t)u * (2 * M) + idx) = l0;
                    ALCH_STORE_GUARD(l0);
                    *reinterpret_cast<SV*>(st + (size_t)u * (2 * M) + M + idx) = l1;
                    ALCH_STORE_GUARD(l1);
                }
            }
            lds_barrier();                      // LDS is refilled next
        }
    }
}

template <int LOGN, typename W>
__global__ void __launch_bounds__((1 << CrtHalfGeo<LOGN, W>::LT), 4)
k_rescale_keep_half(DevRing<W> R, const W* __restrict__ src, const typename Signed<W>::type* __restrict__ stash, W* __restrict__ out,
                    DropTab<W> D, unsigned nitems) {
    typedef CrtHalfGeo<LOGN, W> H;
    typedef typename Signed<W>::type SW;
    constexpr int LOGM = H::LOGM, M = 1 << LOGM, LT = H::LT, T = 1 << LT, F = H::F;
    typedef Geo<LOGM, LT> G;
    typedef typename Vec4<W>::type V;
    constexpr int VL = Vec4<W>::LANES, NV = G::E / VL;
    typedef SW SV __attribute__((ext_vector_type(VL)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L, ddn = D.ddn, Lo = L - ddn;
    // XCD-aware numbering (speed only): workgroups go to the eight XCDs round-robin, so the Lo workgroups of one item -- which all
    // read that item's stash -- are given block ids that agree mod 8, the id k_rescale_drop_half wrote the stash under (its block
    // id is the item): the residues come from that XCD's L2 instead of three trips to HBM.
    const unsigned grp = blockIdx.x / (8u * (unsigned)Lo), rem = blockIdx.x % (8u * (unsigned)Lo);
    const size_t item = (size_t)grp * 8u + (rem & 7u);            // 2 ct + component
    if (item >= nitems) return;
    const int t = ddn + (int)(rem >> 3);
    const W q = R.mod[t].q, qni = R.mod[t].qni;
    const SW* st = stash + item * (size_t)ddn * (size_t)(2 * M);
    const W* xt = src + (item * (size_t)L + t) * (size_t)(2 * M);
    W* ot = out + (item * (size_t)Lo + (t - ddn)) * (size_t)(2 * M);
    NoEpilogue none;
    V keep[NV];
    {   // the combination sum_u reduce_t(R_u) c_{u,t} and stage 0 of its crt
        const W w1 = R.twf[t][1];
        const int tid = threadIdx.x;
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int idx = (tid + T * r) * VL;
            V x, y;
#pragma unroll
            for (int e = 0; e < VL; ++e) { x[e] = 0; y[e] = 0; }
            for (int u = 0; u < ddn; ++u) {
                const W cm = D.comb_m[u][t];
                const SV z0 = *reinterpret_cast<const SV*>(st + (size_t)u * (2 * M) + idx);
                const SV z1 = *reinterpret_cast<const SV*>(st + (size_t)u * (2 * M) + M + idx);
#pragma unroll
                for (int e = 0; e < VL; ++e) {
                    x[e] = csub((W)(x[e] + csub(mont_mul_lazy(reduce_lifted<W>(z0[e], q), cm, q, qni), q)), q);
                    y[e] = csub((W)(y[e] + csub(mont_mul_lazy(reduce_lifted<W>(z1[e], q), cm, q, qni), q)), q);
                }
            }
            V u0;
#pragma unroll
            for (int e = 0; e < VL; ++e) {
                const W tt = csub(mont_mul_lazy(y[e], w1, q, qni), q);
                u0[e] = x[e] + tt;
                keep[r][e] = x[e] + (q - tt);
            }
            *reinterpret_cast<V*>(&lds[swz<LOGM>(idx)]) = u0;
        }
    }
    const auto tw = fwd_tw(R, t);
    const auto twm = fwd_twm(R, t);
    const W Ct = D.comb_m[0][t];
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (half) {
#pragma unroll
            for (int r = 0; r < NV; ++r) *reinterpret_cast<V*>(&lds[swz<LOGM>((tid + T * r) * VL)]) = keep[r];
        }
        lds_barrier();
        const int prefix = 2 + half;
        ntt_pass<LOGM, LT, W, 0, F, false, false, false>(lds, tw, q, qni, (W)0, (W)0, tid, prefix, none);
        lds_barrier();
        ntt_pass<LOGM, LT, W, F, 4, false, false, false>(lds, tw, q, qni, (W)0, (W)0, tid, prefix, none);
        lds_barrier();
        ntt_pass<LOGM, LT, W, F + 4, 4, false, false, false>(lds, tw, q, qni, (W)0, (W)0, tid, prefix, none);
        pair_sync<LOGM>();
        ntt_pass<LOGM, LT, W, F + 8, 4, false, false, false>(lds, twm, q, qni, (W)0, (W)0, tid, prefix, none);
        lds_barrier();
        const W* xh = xt + (half ? M : 0);
        W* oh = ot + (half ? M : 0);
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int idx = (tid + T * r) * VL;
            const V v = *reinterpret_cast<const V*>(&lds[swz<LOGM>(idx)]);
            const V xin = *reinterpret_cast<const V*>(xh + idx);
            V res;
#pragma unroll
            for (int e = 0; e < VL; ++e) {
                const W a = csub(mont_mul_lazy(xin[e], Ct, q, qni), q);
                res[e] = csub((W)(a + (q - csub(v[e], q))), q);
            }
            *reinterpret_cast<V*>(oh + idx) = res;
            ALCH_STORE_GUARD(res);
        }
        lds_barrier();                          // LDS is refilled next
    }
}

}  // namespace alch
