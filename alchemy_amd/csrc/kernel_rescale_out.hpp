// k_rescale_out: the modSwitch that closes PT2CT's mul_ (PT2CT.hs:177, Eval.hs:130), fused with the transforms
// it forces.
//
// Input: a key-switched ciphertext component on the hint's ring (L limbs, CRT basis).  `Rescale (a,b) -> b`
// works coefficient-wise in the Pow (= Dec, two-power index) basis, one dropped limb at a time, outermost first:
//     y_t = (x_t - reduce(lift x_0)) * q_0^-1  (t = 1..L-1),   z_t = (y_t - reduce(lift y_1)) * q_1^-1  (t = 2..), ...
// One workgroup owns one (ciphertext, component) and walks its limbs in order, each limb-polynomial whole in LDS:
//   crtInv -> (last pass, values in registers) apply the drops of all earlier dropped limbs ->
//     dropped limb : keep the centred lift for the later limbs,
//     kept limb    : crt again in place and store (or store the Pow-basis value when the caller asked for it).
// The last inverse pass gives every lane the same coefficient indices for every limb, so the lifted residues of
// the dropped limbs are a per-lane private stash: written and read back by the same lane (global scratch,
// `ddn * n` signed words per resident workgroup, L2-resident), no cross-lane ordering needed.
// Cost per component: L crtInv + (L - ddn) crt, one read of L and one write of L - ddn limb-polynomials.
// (Loading the next limb into registers during the current transforms, as k_tensor_intt does, was measured 5-7 %
// slower here, twice: with the forward transform in the same kernel the 32 extra registers spill.)
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "ntt_engine.hpp"

// cache policy of k_rescale_out_lin's global accesses.  2 = non-temporal was measured at -15 % on the full mul_ (272 k -> 233 k op/s, same box):
// the limbs it reads were just written by the key-switch kernel and are still served from L2 / Infinity Cache under the default policy.
#ifndef ALCH_RS_NT
#define ALCH_RS_NT 0
#endif

namespace alch {

template <typename W, typename Rsrc>
__device__ __forceinline__ typename Signed<W>::type ld_word(Rsrc r, u32 voff, u32 soff) {
    if constexpr (sizeof(W) == 4) return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
    else return __builtin_bit_cast(int64_t, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
template <typename W, typename Rsrc>
__device__ __forceinline__ void st_word(Rsrc r, u32 voff, u32 soff, W v) {
    if constexpr (sizeof(W) == 4) __builtin_amdgcn_raw_buffer_store_b32((u32)v, r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b64(r, 0, 0, 0)), v), r, voff, soff, 0);
}

template <int LOGN, typename W>
__global__ void __launch_bounds__(Geo<LOGN>::T)
k_rescale_out(DevRing<W> R, const W* __restrict__ src, W* __restrict__ out, typename Signed<W>::type* stash,
              unsigned nitems, DropTab<W> D, int pow_out) {
    typedef Geo<LOGN> G;
    typedef typename Vec4<W>::type V;
    typedef typename Signed<W>::type SW;
    constexpr int VL = Vec4<W>::LANES;
    constexpr int RR = 1 << G::NS0;                 // coefficients per group of the last inverse pass
    constexpr int STRIDE = G::N / RR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L, ddn = D.ddn, Lo = L - ddn;
    // buffer instructions (see kernel_ks_half.hpp): one VGPR of per-lane address, the rest scalar; offsets < 2^32
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<W*>(src), 0, (u32)((size_t)nitems * L * G::N * sizeof(W)), 0x00020000);
    const auto rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, (u32)((size_t)nitems * Lo * G::N * sizeof(W)), 0x00020000);
    const auto rst = __builtin_amdgcn_make_buffer_rsrc(stash, 0, (u32)((size_t)gridDim.x * ddn * G::N * sizeof(W)), 0x00020000);
    constexpr u32 ROW = (u32)G::N * (u32)sizeof(W), WB = (u32)sizeof(W);
    const u32 st0 = blockIdx.x * (u32)ddn * ROW;
    const u32 lane16 = threadIdx.x * 16u;

    for (unsigned item = blockIdx.x; item < nitems; item += gridDim.x) {
        const u32 x = item * (u32)L * ROW;                           // item = 2*ct + component; byte offsets
        const u32 o = item * (u32)Lo * ROW;
        for (int t = 0; t < L; ++t) {
            const ModP<W> m = R.mod[t];
            const W q = m.q, qni = m.qni;
            const W half = (q - 1) >> 1;
            const u32 poly = x + (u32)t * ROW;
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));       // keep each limb's address arithmetic inside the loop (VGPR pressure)
            lds_barrier();                      // the previous limb's last pass / stores have finished with LDS
#pragma unroll
            for (int r = 0; r < G::E / VL; ++r)
                *reinterpret_cast<V*>(&lds[swz<LOGN>((tid + G::T * r) * VL)]) =
                    __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, poly + (u32)(G::T * r) * 16u, 0));
            lds_barrier();
            const int nd = t < ddn ? t : ddn;   // drops that apply to this limb
            const u32 ot = o + (u32)(t - ddn) * ROW;
            auto epi = [&](int, int base, W* v) {
#pragma unroll
                for (int k = 0; k < RR; ++k) v[k] = csub(v[k], q);
                for (int u = 0; u < nd; ++u) {
                    const W qim = D.qinv_m[u][t];
                    SW z[RR];
#pragma unroll
                    for (int k = 0; k < RR; ++k) z[k] = ld_word<W>(rst, (u32)base * WB, st0 + (u32)u * ROW + (u32)(k * STRIDE) * WB);
#pragma unroll
                    for (int k = 0; k < RR; ++k) {
                        W r;
                        if (D.balanced) r = z[k] < 0 ? (W)z[k] + q : (W)z[k];
                        else { SW rr = z[k] % (SW)q; r = rr < 0 ? (W)(rr + (SW)q) : (W)rr; }
                        v[k] = csub(mont_mul_lazy((W)(v[k] - r + q), qim, q, qni), q);
                    }
                }
                if (t < ddn) {
#pragma unroll
                    for (int k = 0; k < RR; ++k)
                        st_word(rst, (u32)base * WB, st0 + (u32)t * ROW + (u32)(k * STRIDE) * WB,
                                (W)(v[k] > half ? (SW)v[k] - (SW)q : (SW)v[k]));
                } else if (pow_out) {
#pragma unroll
                    for (int k = 0; k < RR; ++k) st_word(rout, (u32)base * WB, ot + (u32)(k * STRIDE) * WB, v[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < RR; ++k) lds[swz<LOGN>(base + k * STRIDE)] = v[k];   // the words this lane just read
                }
            };
            ntt_inverse<LOGN, W, true>(lds, inv_tw(R, t), q, qni, R.ninv_m[t], R.w1ninv_m[t], tid, epi);
            if (t >= ddn && !pow_out) {
                lds_barrier();
                ntt_forward<LOGN, W, false>(lds, fwd_tw(R, t), fwd_twm(R, t), q, qni, tid, NoEpilogue());
#pragma unroll
                for (int r = 0; r < G::E / VL; ++r) {
                    const int idx = (tid + G::T * r) * VL;
                    V v = *reinterpret_cast<const V*>(&lds[swz<LOGN>(idx)]);
#pragma unroll
                    for (int e = 0; e < VL; ++e) v[e] = csub(v[e], q);
                    __builtin_amdgcn_raw_buffer_store_b128(
                        __builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(rout, 0, 0, 0)), v), rout, lane16,
                        ot + (u32)(G::T * r) * 16u, 0);
                    ALCH_STORE_GUARD(v);
                }
            }
        }
    }
}

// k_rescale_out_lin: the same modSwitch with the kept limbs never leaving the CRT basis.
//
// Rescale is affine in the kept limb: with r_u = lift of the u-th dropped residue (after the drops before it),
//     z_t = (..((x_t - r_0) q_0^-1 - r_1) q_1^-1 ..)  =  x_t C_t - sum_u r_u c_{u,t}     (mod q_t),
//     c_{u,t} = prod_{v >= u} q_v^-1,  C_t = c_{0,t},
// and crt is linear, so  crt(z_t) = crt(x_t) C_t - crt(sum_u reduce(r_u) c_{u,t}):  the same residues bit for bit
// (both sides are the canonical representative of the same element of Z_{q_t}), with ddn inverse transforms (the dropped
// limbs) + (L - ddn) forward transforms per component instead of L inverse + (L - ddn) forward: 5 instead of 8 for the
// 5 -> 3 limb switch of PT2CT's mul_, 10 instead of 16 per ciphertext.
// One workgroup owns one (ciphertext, component): the lifted residues r_u live in registers (32 per lane and dropped
// limb, DDN <= 2), the combination sum_u reduce(r_u) c_{u,t} is formed lane-locally and written to LDS in the layout
// the first forward pass reads, and the closing pass's epilogue reads crt(x_t) straight from HBM, 16 consecutive slots
// per lane, and stores the result.  Serves CRT-basis output with one or two dropped limbs; k_rescale_out keeps the
// Pow-basis output and three-limb drops.
template <int LOGN, typename W, int DDN, bool BALANCED>
__global__ void __launch_bounds__(Geo<LOGN>::T)
k_rescale_out_lin(DevRing<W> R, const W* __restrict__ src, W* __restrict__ out, unsigned nitems, DropTab<W> D) {
    typedef Geo<LOGN> G;
    typedef typename Vec4<W>::type V;
    typedef typename Signed<W>::type SW;
    constexpr int VL = Vec4<W>::LANES;
    constexpr int RR = 1 << G::NS0;                 // coefficients per group of the last inverse pass
    constexpr int STRIDE = G::N / RR;
    constexpr int NG = G::E / RR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L, Lo = L - DDN;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<W*>(src), 0, (u32)((size_t)nitems * L * G::N * sizeof(W)), 0x00020000);
    const auto rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, (u32)((size_t)nitems * Lo * G::N * sizeof(W)), 0x00020000);
    constexpr u32 ROW = (u32)G::N * (u32)sizeof(W);
    const u32 lane16 = threadIdx.x * 16u;

    for (unsigned item = blockIdx.x; item < nitems; item += gridDim.x) {
        const u32 x = item * (u32)L * ROW;                           // item = 2*ct + component; byte offsets
        const u32 o = item * (u32)Lo * ROW;
        SW rr0[G::E], rr1[DDN > 1 ? G::E : 1];                        // lifted residues of the dropped limbs, per lane (registers)
        // ---- phase 1: the dropped limbs, outermost first (spelled out per limb: every index into rr0 / rr1 is static)
        auto drop_limb = [&](auto UC) {
            constexpr int u = decltype(UC)::value;
            const ModP<W> m = R.mod[u];
            const W q = m.q, qni = m.qni, half = (q - 1) >> 1;
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            lds_barrier();
#pragma unroll
            for (int r = 0; r < G::E / VL; ++r)
                *reinterpret_cast<V*>(&lds[swz<LOGN>((tid + G::T * r) * VL)]) =
                    __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, x + (u32)u * ROW + (u32)(G::T * r) * 16u, ALCH_RS_NT));
            lds_barrier();
            auto epi = [&](int g, int, W* v) {
#pragma unroll
                for (int k = 0; k < RR; ++k) {
                    W c = csub(v[k], q);
                    if constexpr (u == 1) {                          // the drop of limb 0 comes first
                        const SW z = rr0[g * RR + k];
                        W r;
                        if constexpr (BALANCED) r = z < 0 ? (W)z + q : (W)z;
                        else { SW t2 = z % (SW)q; r = t2 < 0 ? (W)(t2 + (SW)q) : (W)t2; }
                        c = csub(mont_mul_lazy((W)(c - r + q), D.qinv_m[0][1], q, qni), q);
                    }
                    const SW lifted = c > half ? (SW)c - (SW)q : (SW)c;
                    if constexpr (u == 0) rr0[g * RR + k] = lifted; else rr1[g * RR + k] = lifted;
                }
            };
            ntt_inverse<LOGN, W, true, (u > 0)>(lds, inv_tw(R, u), q, qni, R.ninv_m[u], R.w1ninv_m[u], tid, epi);
        };
        drop_limb(std::integral_constant<int, 0>());
        if constexpr (DDN > 1) drop_limb(std::integral_constant<int, 1>());
        // ---- phase 2: the kept limbs stay in the CRT basis
        for (int t = DDN; t < L; ++t) {
            const ModP<W> m = R.mod[t];
            const W q = m.q, qni = m.qni;
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            lds_barrier();                      // the previous transform has finished reading LDS
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int base = tid + G::T * g;
#pragma unroll
                for (int k = 0; k < RR; ++k) {
                    auto red = [&](SW z) -> W {
                        if constexpr (BALANCED) return z < 0 ? (W)z + q : (W)z;
                        else { SW t2 = z % (SW)q; return t2 < 0 ? (W)(t2 + (SW)q) : (W)t2; }
                    };
                    W acc = csub(mont_mul_lazy(red(rr0[g * RR + k]), D.comb_m[0][t], q, qni), q);
                    if constexpr (DDN > 1) acc = csub(acc + csub(mont_mul_lazy(red(rr1[g * RR + k]), D.comb_m[1][t], q, qni), q), q);
                    lds[swz<LOGN>(base + k * STRIDE)] = acc;
                }
            }
            lds_barrier();
            const u32 xt = x + (u32)t * ROW, ot = o + (u32)(t - DDN) * ROW;
            const W Ct = D.comb_m[0][t];
            auto twf = fwd_tw(R, t);
            auto twm = fwd_twm(R, t);
            ntt_forward<LOGN, W, true, true>(lds, twf, twm, q, qni, tid, [&](int, int base, W* v) {
#pragma unroll
                for (int k = 0; k < 16; k += VL) {
                    const V xin = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rs, (u32)(base + k) * (u32)sizeof(W), xt, ALCH_RS_NT));
                    V res;
#pragma unroll
                    for (int e = 0; e < VL; ++e) {
                        const W a = csub(mont_mul_lazy(xin[e], Ct, q, qni), q);
                        res[e] = csub((W)(a + (q - csub(v[k + e], q))), q);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(
                        __builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(rout, 0, 0, 0)), res), rout,
                        (u32)(base + k) * (u32)sizeof(W), ot, ALCH_RS_NT);
                    // see ALCH_STORE_GUARD (ntt_engine.hpp): this is the store the hazard was found on
                    asm volatile("s_nop 1" ::"v"(res));
                }
            });
        }
    }
}

}  // namespace alch
