// k_tensor_intt_split: kernel A (c2 = a1 b1 s, crtInv, centred lift -> TrivGad digit) with the inverse transform run
// as TWO sequential half-size sub-transforms in 64 (+4) KiB of LDS, so that two independent 8-wave workgroups share
// a CU (k_tensor_intt holds the whole polynomial in LDS: one 16-wave workgroup per CU whose waves move through the
// load / transform / store phases in lock step).
//
// Gentleman-Sande order: the stages 14..1 of the n-point inverse act inside each half of the slots (sub-transform of
// size n/2 with the big ring's twiddles, prefix = 2 + half, no n^-1); stage 0 pairs coefficient k of the two
// halves.  Half 0's result waits in 32 registers per lane while half 1 is transformed; both are read from LDS in the
// same lane-contiguous pattern, so stage 0, the n^-1 s scaling, the centred lift and the 16-byte digit stores are
// lane-local.
#pragma once
#include <hip/hip_runtime.h>
#include "ntt_engine.hpp"

#ifndef ALCH_TI_SPLIT_PASSES
#define ALCH_TI_SPLIT_PASSES 4
#endif

// experiment switch: cache policy of the operand loads (2 = non-temporal: read once by this kernel)
#ifndef ALCH_TI_NT_IN
#define ALCH_TI_NT_IN 0
#endif

// experiment (tools/build_variant.sh partials -DALCH_A_PARTIALS=1): this kernel also forms the key-switch
// kernel's starting values c0 = a0 b0 s + c2 h0_i, c1 = (a0 b1 + a1 b0) s + c2 h1_i for its limb (it holds a1, b1 and c2 anyway) and
// writes them to the result rows, so that the key-switch kernel reads two finished polynomials instead of four cold operands and
// two hint rows at the head of every item.  Measured, see DESIGN.md "dead ends"; off in the product.
#ifndef ALCH_A_PARTIALS
#define ALCH_A_PARTIALS 0
#endif

namespace alch {

// Q30: every modulus below 2^30 -- 8-instruction inverse butterflies (bfly_inv4); values stay in [0,2q) as otherwise
template <int LOGN, bool Q30 = false>
__global__ void __launch_bounds__(1 << (LOGN - 6), 4)
k_tensor_intt_split(DevRing<u32> R, const u32* __restrict__ a, const u32* __restrict__ b, int32_t* __restrict__ digits,
                    unsigned nitems, Scal<u32> spre
#if ALCH_A_PARTIALS
                    , const u32* __restrict__ hint, u32* __restrict__ out
#endif
                    ) {
    typedef u32 W;
    constexpr int LOGM = LOGN - 1, M = 1 << LOGM, LT = LOGN - 6, T = 1 << LT;
    typedef Geo<LOGM, LT> G;
    static_assert(G::E == 32 && (LOGM - 2) % 4 == 0, "32 coefficients per lane, stages 4+4+4+2");
    typedef u32 V __attribute__((ext_vector_type(4)));
    typedef int32_t SV __attribute__((ext_vector_type(4)));
    constexpr int NV = G::E / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L;
    const u32 nct = nitems / (unsigned)L;
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<W*>(a), 0, (u32)((size_t)nct * 2 * L * (2 * M) * 4), 0x00020000);
    const auto rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<W*>(b), 0, (u32)((size_t)nct * 2 * L * (2 * M) * 4), 0x00020000);
    const auto rd = __builtin_amdgcn_make_buffer_rsrc(digits, 0, (u32)((size_t)nct * L * (2 * M) * 4), 0x00020000);
    const u32 lane16 = threadIdx.x * 16u;
    constexpr u32 ROW = (u32)(2 * M) * 4u, SLICE = (u32)T * 16u, HALF = (u32)M * 4u;
#if ALCH_A_PARTIALS
    const auto rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<W*>(hint), 0, (u32)((size_t)2 * L * L * (2 * M) * 4), 0x00020000);
    const auto ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, (u32)((size_t)nct * 2 * L * (2 * M) * 4), 0x00020000);
#endif

    for (unsigned item = blockIdx.x; item < nitems; item += gridDim.x) {
        const u32 ct = item / (unsigned)L, i = item % (unsigned)L;
        const ModP<W> m = R.mod[i];
        const W q = m.q, qni = m.qni;
        const W ninv_s = mont_mul(R.ninv_m[i], spre.v[i], m), w1ninv_s = mont_mul(R.w1ninv_m[i], spre.v[i], m);
        const u32 row = ((2 * ct + 1) * (u32)L + i) * ROW;           // a1 / b1 of this ciphertext and limb
        const u32 drow = (ct * (u32)L + i) * ROW;
        const int rot = (int)((item ^ (item >> 3)) & (NV - 1));      // slice order rotated per item (HBM channels)
        const W* twi = R.twi[i];
        V keep[NV];
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            // c2 = a1 b1 R^-1 (lazy) for this half of the slots -> LDS
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const u32 so = row + (u32)half * HALF + SLICE * (u32)((r + rot) & (NV - 1));
                const V va = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(ra, lane16, so, ALCH_TI_NT_IN));
                const V vb = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rb, lane16, so, ALCH_TI_NT_IN));
#if ALCH_A_PARTIALS
                if (out) {   // the key-switch kernel's tensor part for this limb, with its very formulas (kernel_ks_half.hpp)
                    const u32 so0 = so - (u32)L * ROW;                                  // a0 / b0: the element in front of a1 / b1
                    const u32 piece = (u32)half * HALF + SLICE * (u32)((r + rot) & (NV - 1));
                    const u32 hrow = ((2 * i) * (u32)L + i) * ROW + piece;               // hint row (digit i, component 0, limb i)
                    const V a0v = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(ra, lane16, so0, ALCH_TI_NT_IN));
                    const V b0v = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rb, lane16, so0, ALCH_TI_NT_IN));
                    const V h0v = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rh, lane16, hrow, 0));
                    const V h1v = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rh, lane16, hrow + (u32)L * ROW, 0));
                    const W sr2 = spre.v[i];
                    V o0v, o1v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const W x0 = csub(mont_mul_lazy(a0v[e], sr2, q, qni), q);
                        const W x1 = csub(mont_mul_lazy(va[e], sr2, q, qni), q);
                        const W c2 = csub(mont_mul_lazy(vb[e], x1, q, qni), q);
                        const u64 p0 = (u64)x0 * b0v[e] + (u64)c2 * h0v[e];
                        const u32 m0 = (u32)p0 * qni;
                        o0v[e] = csub((u32)((p0 + (u64)m0 * q) >> 32), q);
                        const u64 p1 = (u64)x0 * vb[e] + (u64)x1 * b0v[e] + (u64)c2 * h1v[e];
                        const u64 p1r = ((u64)csub((W)(p1 >> 32), q) << 32) | (u32)p1;
                        const u32 m1 = (u32)p1r * qni;
                        o1v[e] = csub((u32)((p1r + (u64)m1 * q) >> 32), q);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(ro, 0, 0, 0)), o0v), ro, lane16, so0, 0);
                    ALCH_STORE_GUARD(o0v);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(ro, 0, 0, 0)), o1v), ro, lane16, so, 0);
                    ALCH_STORE_GUARD(o1v);
                }
#endif
                V v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = mont_mul_lazy(va[e], vb[e], q, qni);
                *reinterpret_cast<V*>(&lds[swz<LOGM>((tid + T * ((r + rot) & (NV - 1))) * 4)]) = v;
            }
            lds_barrier();
            NoEpilogue none;
            const int prefix = 2 + half;
#if ALCH_TI_SPLIT_PASSES == 3
            // 14 stages as 5 + 5 + 4: three LDS round trips per half instead of four (radix-32 groups: 32 coefficients and
            // up to 31 per-lane twiddles in registers -- affordable here, there are no accumulators to keep)
            ntt_pass<LOGM, LT, W, 9, 5, true, false, false, W, NoEpilogue&, false, Q30>(lds, twi, q, qni, (W)0, (W)0, tid, prefix, none);
            lds_barrier();
            ntt_pass<LOGM, LT, W, 4, 5, true, false, false, W, NoEpilogue&, false, Q30>(lds, twi, q, qni, (W)0, (W)0, tid, prefix, none);
            lds_barrier();
            ntt_pass<LOGM, LT, W, 0, 4, true, false, false, W, NoEpilogue&, false, Q30>(lds, twi, q, qni, (W)0, (W)0, tid, prefix, none);
            lds_barrier();
#else
            ntt_pass<LOGM, LT, W, 10, 4, true, false, false, W, NoEpilogue&, false, Q30>(lds, twi, q, qni, (W)0, (W)0, tid, prefix, none);
            pair_sync<LOGM>();
            ntt_pass<LOGM, LT, W, 6, 4, true, false, false, W, NoEpilogue&, false, Q30>(lds, twi, q, qni, (W)0, (W)0, tid, prefix, none);
            lds_barrier();
            ntt_pass<LOGM, LT, W, 2, 4, true, false, false, W, NoEpilogue&, false, Q30>(lds, twi, q, qni, (W)0, (W)0, tid, prefix, none);
            lds_barrier();
            ntt_pass<LOGM, LT, W, 0, 2, true, false, false, W, NoEpilogue&, false, Q30>(lds, twi, q, qni, (W)0, (W)0, tid, prefix, none);
            lds_barrier();
#endif
            if (half == 0) {
#pragma unroll
                for (int r = 0; r < NV; ++r) keep[r] = *reinterpret_cast<const V*>(&lds[swz<LOGM>((tid + T * r) * 4)]);
            } else {
                // stage 0 with n^-1 s folded in, centred lift, digit stores (lane-contiguous, 16 bytes per lane)
                const W hq = (q - 1) >> 1;
#pragma unroll
                for (int r = 0; r < NV; ++r) {
                    const V hi = *reinterpret_cast<const V*>(&lds[swz<LOGM>((tid + T * r) * 4)]);
                    SV z0, z1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const W x = csub(keep[r][e], q), y = csub(hi[e], q);
                        const W c0 = csub(mont_mul_lazy((W)(x + y), ninv_s, q, qni), q);
                        const W c1 = csub(mont_mul_lazy((W)(x - y + q), w1ninv_s, q, qni), q);
                        z0[e] = c0 > hq ? (int32_t)c0 - (int32_t)q : (int32_t)c0;
                        z1[e] = c1 > hq ? (int32_t)c1 - (int32_t)q : (int32_t)c1;
                    }
                    const u32 so = drow + SLICE * (u32)r;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(rd, 0, 0, 0)), z0), rd, lane16, so, 0);
                    ALCH_STORE_GUARD(z0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(rd, 0, 0, 0)), z1), rd, lane16, so + HALF, 0);
                    ALCH_STORE_GUARD(z1);
                }
            }
            lds_barrier();                      // LDS is refilled next
        }
    }
}

}  // namespace alch
