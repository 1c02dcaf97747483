// k_crt_half: Tensor crt / crtInv of a limb-polynomial of 128 KiB (n = 2^15 with 32-bit residues, n = 2^14 with 64-bit
// residues) as TWO sequential half-size sub-transforms in 64 (+4) KiB of LDS, so that two independent 8-wave workgroups
// share a CU.  k_crt holds such a polynomial whole in LDS: one 16-wave workgroup per CU whose waves move through the
// load / transform / store phases in lock step, and nothing overlaps the HBM phases (0.20 us per transform against
// 0.125 us inside the split kernels of the key switch).  Same traffic as k_crt: one read and one write.
//
//   crt    : stage 0 pairs coefficient k with k + n/2: u0 = x + w1 y goes to LDS, u1 = x - w1 y waits in registers (one
//            group of 16-byte pieces per lane); stages 1.. act inside each half (sub-transform of size n/2 with the big
//            ring's twiddles, prefix = 2 + half); half 0 is stored, then half 1 takes the LDS.
//   crtInv : the mirror image (Gentleman-Sande): both halves of the slots through the inverse sub-transform, half 0's
//            result waits in registers, stage 0 with n^-1 folded in is lane-local.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "ntt_engine.hpp"

namespace alch {

template <int LOGN, typename W> struct CrtHalfGeo {
    static constexpr int LOGM = LOGN - 1;
    static constexpr int LT = LOGN - (sizeof(W) == 4 ? 6 : 5);      // 512 threads; 32 (u32) / 16 (u64) coefficients per lane
    static constexpr int F = LOGM - 12;                             // stages of the odd pass (the others run four)
    static_assert(F >= 1 && F <= 4, "13 to 16 stages per half");
};

template <int LOGN, typename W, bool INVERSE>
__global__ void __launch_bounds__((1 << CrtHalfGeo<LOGN, W>::LT), 4)
k_crt_half(DevRing<W> R, W* __restrict__ data, const W* __restrict__ src, size_t first_poly) {
    typedef CrtHalfGeo<LOGN, W> H;
    constexpr int LOGM = H::LOGM, M = 1 << LOGM, LT = H::LT, T = 1 << LT, F = H::F;
    typedef Geo<LOGM, LT> G;
    typedef typename Vec4<W>::type V;
    constexpr int VL = Vec4<W>::LANES, NV = G::E / VL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const size_t p = first_poly + blockIdx.x;
    const int j = (int)(p % (size_t)R.L);
    W* poly = data + p * (size_t)(2 * M);
    const W* in = src ? src + p * (size_t)(2 * M) : poly;
    const W q = R.mod[j].q, qni = R.mod[j].qni;
    NoEpilogue none;
    V keep[NV];

    if constexpr (!INVERSE) {
        const W w1 = R.twf[j][1];
        {
            const int tid = threadIdx.x;
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const int idx = (tid + T * r) * VL;
                const V x = *reinterpret_cast<const V*>(in + idx), y = *reinterpret_cast<const V*>(in + M + idx);
                V u0;
#pragma unroll
                for (int e = 0; e < VL; ++e) {
                    const W t = csub(mont_mul_lazy(y[e], w1, q, qni), q);
                    u0[e] = x[e] + t;                       // canonical inputs: both in [0, 2q)
                    keep[r][e] = x[e] + (q - t);
                }
                *reinterpret_cast<V*>(&lds[swz<LOGM>(idx)]) = u0;
            }
        }
        const auto tw = fwd_tw(R, j);
        const auto twm = fwd_twm(R, j);
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
            W* dst = poly + (half ? M : 0);
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const int idx = (tid + T * r) * VL;
                V v = *reinterpret_cast<const V*>(&lds[swz<LOGM>(idx)]);
#pragma unroll
                for (int e = 0; e < VL; ++e) v[e] = csub(v[e], q);
                *reinterpret_cast<V*>(dst + idx) = v;
                ALCH_STORE_GUARD(v);
            }
            lds_barrier();                      // LDS is refilled next
        }
    } else {
        const auto twi = inv_tw(R, j);
        typedef typename std::remove_cv<typename std::remove_pointer<decltype(twi)>::type>::type TWI;
        const W ninv = R.ninv_m[j], w1ninv = R.w1ninv_m[j];
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
                // stage 0 with n^-1 folded in: lane-local, canonical results
#pragma unroll
                for (int r = 0; r < NV; ++r) {
                    const int idx = (tid + T * r) * VL;
                    const V hi = *reinterpret_cast<const V*>(&lds[swz<LOGM>(idx)]);
                    V c0, c1;
#pragma unroll
                    for (int e = 0; e < VL; ++e) {
                        const W x = csub(keep[r][e], q), y = csub(hi[e], q);
                        c0[e] = csub(mont_mul_lazy((W)(x + y), ninv, q, qni), q);
                        c1[e] = csub(mont_mul_lazy((W)(x - y + q), w1ninv, q, qni), q);
                    }
                    *reinterpret_cast<V*>(poly + idx) = c0;
                    ALCH_STORE_GUARD(c0);
                    *reinterpret_cast<V*>(poly + M + idx) = c1;
                    ALCH_STORE_GUARD(c1);
                }
            }
            lds_barrier();                      // LDS is refilled next
        }
    }
}

}  // namespace alch
