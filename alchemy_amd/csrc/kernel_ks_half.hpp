// k_ks_accum_half: the key-switch kernel split so that TWO workgroups fit one CU.
//
// One workgroup owns one HALF of the output slots of (ciphertext, limb j): slots [half*n/2, (half+1)*n/2).
// After the first Cooley-Tukey stage the two halves of a transform are independent, so each workgroup
//   * runs global stages 0..2 straight from HBM/L2 into registers (it reads the whole digit, computes only
//     its own half of the stage-0 outputs -- one extra modular product per coefficient, +6.7 % of a
//     transform's multiplies -- and writes n/2 words to LDS), then
//   * finishes the remaining log n - 3 stages as a sub-transform of size n/2 in 64 KiB of LDS
//     (prefix = 2 + half, see ntt_pass), the last pass feeding the hint multiply-accumulate from registers.
// With 64 KiB LDS, 512 threads and <= 128 VGPRs per workgroup a CU holds two workgroups whose HBM phases
// (tensor inputs, digits, hint, result stores) and barrier stalls hide under each other's butterflies;
// the one-workgroup-per-CU form (k_ks_accum) idles the VALU during those phases.
#pragma once
#include "ntt_engine.hpp"

namespace alch {

__device__ __forceinline__ u32 mont_red_lazy(u64 p, u32 q, u32 qni) {       // p < 2^32 * q  ->  [0, 2q)
    u32 m = (u32)p * qni;
    return (u32)((p + (u64)m * q) >> 32);
}

template <int LOGN, bool BALANCED>
__global__ void __launch_bounds__(1 << (LOGN - 6), 4)
k_ks_accum_half(DevRing<u32> R, const u32* __restrict__ a, const u32* __restrict__ b,
                const int32_t* __restrict__ digits, const u32* __restrict__ hint, u32* __restrict__ out,
                unsigned nct, Scal<u32> spre) {
    typedef u32 W;
    constexpr int LOGM = LOGN - 1, M = 1 << LOGM, N = 1 << LOGN, LT = LOGN - 6, T = 1 << LT;
    typedef Geo<LOGM, LT> G;
    static_assert(G::E == 32, "32 coefficients per thread");
    static_assert((LOGM - 2) % 4 == 0, "remaining stages must split into radix-16 passes");
    typedef u32 V __attribute__((ext_vector_type(4)));
    typedef int32_t SV __attribute__((ext_vector_type(4)));
    constexpr int NG = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L;
    // XCD-aware placement (speed only): the 2L workgroups of one ciphertext get ids that agree mod 8, so
    // they share an XCD and its L2 serves the digits they all read.
    const unsigned per = 16u * (unsigned)L;
    const unsigned grp = blockIdx.x / per, rem = blockIdx.x % per;
    const unsigned which = rem >> 3;
    const int j = (int)(which >> 1);
    const int hf = (int)(which & 1u);
    const size_t ct = (size_t)grp * 8u + (rem & 7u);
    if (ct >= nct) return;

    const ModP<W> m = R.mod[j];
    const W q = m.q, qni = m.qni;
    const W sr2 = spre.v[j];
    const size_t n = (size_t)N;
    const size_t slot0 = (size_t)hf * M;
    const W* a0 = a + ((2 * ct) * (size_t)L + j) * n + slot0;
    const W* a1 = a + ((2 * ct + 1) * (size_t)L + j) * n + slot0;
    const W* b0 = b + ((2 * ct) * (size_t)L + j) * n + slot0;
    const W* b1 = b + ((2 * ct + 1) * (size_t)L + j) * n + slot0;
    const W* hj = hint + (size_t)j * n + slot0;                // + ((i*2 + c)*L)*n
    const size_t hstride = (size_t)L * n;

    W acc0[32], acc1[32];
    {   // c0, c1 and the diagonal digit (i == j): d_j = c2_j (mod q_j), no transform needed
        const W* h0 = hj + (size_t)(2 * j) * hstride;
        const W* h1 = hj + (size_t)(2 * j + 1) * hstride;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int k = 0; k < 16; k += 4) {
                const int idx = ((int)threadIdx.x + T * g) * 16 + k;
                V va0 = *reinterpret_cast<const V*>(a0 + idx), va1 = *reinterpret_cast<const V*>(a1 + idx);
                V vb0 = *reinterpret_cast<const V*>(b0 + idx), vb1 = *reinterpret_cast<const V*>(b1 + idx);
                V vh0 = *reinterpret_cast<const V*>(h0 + idx), vh1 = *reinterpret_cast<const V*>(h1 + idx);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const W x0 = csub(mont_mul_lazy(va0[e], sr2, q, qni), q);          // a0 s R
                    const W x1 = csub(mont_mul_lazy(va1[e], sr2, q, qni), q);          // a1 s R
                    const W c2 = csub(mont_mul_lazy(vb1[e], x1, q, qni), q);           // a1 b1 s
                    // sums of two products (< 2 q^2 < 2^32 q) share one Montgomery reduction
                    acc0[g * 16 + k + e] = csub(mont_red_lazy((u64)x0 * vb0[e] + (u64)c2 * vh0[e], q, qni), q);
                    const W t1 = csub(mont_red_lazy((u64)x0 * vb1[e] + (u64)x1 * vb0[e], q, qni), q);
                    const W t2 = csub(mont_mul_lazy(c2, vh1[e], q, qni), q);
                    acc1[g * 16 + k + e] = csub(t1 + t2, q);
                }
                __builtin_amdgcn_sched_barrier(0);   // keep one 4-coefficient slice of loads live at a time
            }
        }
    }

    for (int i = 0; i < L; ++i) {
        if (i == j) continue;
        const int32_t* d = digits + (ct * (size_t)L + i) * n;
        // Nothing below depends on i except d and the hint rows; keep addresses and twiddles from being
        // hoisted out of the digit loop (that costs ~250 spilled VGPRs).
        const W* twf = R.twf[j];
        int tid = threadIdx.x;
        asm volatile("" : "+s"(twf), "+v"(tid));
        __syncthreads();      // previous transform's last pass has finished reading LDS

        // ---- global stages 0..2, HBM/L2 -> registers -> LDS
        {
            const W w1 = twf[1], w2 = twf[2 + hf], w3a = twf[4 + 2 * hf], w3b = twf[5 + 2 * hf];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int lo4 = (tid + T * g) * 4;                    // coefficients lo4..lo4+3 of each eighth
                V u[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    SV zx = *reinterpret_cast<const SV*>(d + k * (N / 8) + lo4);
                    SV zy = *reinterpret_cast<const SV*>(d + (k + 4) * (N / 8) + lo4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        W xr, yr;
                        if constexpr (BALANCED) { xr = (W)zx[e] + q; yr = (W)zy[e] + q; }           // (0, 2q)
                        else {
                            xr = mont_mul_lazy((W)((W)zx[e] + R.dig_off[j]), m.r1, q, qni);
                            yr = mont_mul_lazy((W)((W)zy[e] + R.dig_off[j]), m.r1, q, qni);
                        }
                        const W xx = csub(xr, q);
                        const W t = csub(mont_mul_lazy(yr, w1, q, qni), q);
                        u[k][e] = hf ? xx + (q - t) : xx + t;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    W u0 = u[0][e], u1 = u[1][e], u2 = u[2][e], u3 = u[3][e];
                    bfly_fwd(u0, u2, w2, q, qni);
                    bfly_fwd(u1, u3, w2, q, qni);
                    bfly_fwd(u0, u1, w3a, q, qni);
                    bfly_fwd(u2, u3, w3b, q, qni);
                    u[0][e] = u0; u[1][e] = u1; u[2][e] = u2; u[3][e] = u3;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<V*>(&lds[swz<LOGM>(k * (N / 8) + lo4)]) = u[k];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();

        // ---- remaining stages: sub-transform of size n/2, local stages 2 .. LOGM-1
        const W* h0 = hj + (size_t)(2 * i) * hstride;
        const W* h1 = hj + (size_t)(2 * i + 1) * hstride;
        auto epi = [&acc0, &acc1, h0, h1, q, qni](int g, int base, W* x) {
#pragma unroll
            for (int k = 0; k < 16; k += 4) {
                V vh0 = *reinterpret_cast<const V*>(h0 + base + k), vh1 = *reinterpret_cast<const V*>(h1 + base + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0[g * 16 + k + e] = csub(acc0[g * 16 + k + e] + csub(mont_mul_lazy(x[k + e], vh0[e], q, qni), q), q);
                    acc1[g * 16 + k + e] = csub(acc1[g * 16 + k + e] + csub(mont_mul_lazy(x[k + e], vh1[e], q, qni), q), q);
                }
            }
        };
        const int prefix = 2 + hf;
        NoEpilogue none;
        constexpr int NP = (LOGM - 2) / 4;
        if constexpr (NP == 1) {
            ntt_pass<LOGM, LT, W, 2, 4, false, true, true>(lds, twf, q, qni, (W)0, (W)0, tid, prefix, epi);
        } else if constexpr (NP == 2) {
            ntt_pass<LOGM, LT, W, 2, 4, false, false, true>(lds, twf, q, qni, (W)0, (W)0, tid, prefix, none);
            __syncthreads();
            ntt_pass<LOGM, LT, W, 6, 4, false, true, true>(lds, twf, q, qni, (W)0, (W)0, tid, prefix, epi);
        } else {
            static_assert(NP <= 3, "at most 3 LDS passes");
            ntt_pass<LOGM, LT, W, 2, 4, false, false, true>(lds, twf, q, qni, (W)0, (W)0, tid, prefix, none);
            __syncthreads();
            ntt_pass<LOGM, LT, W, 6, 4, false, false, true>(lds, twf, q, qni, (W)0, (W)0, tid, prefix, none);
            __syncthreads();
            ntt_pass<LOGM, LT, W, 10, 4, false, true, true>(lds, twf, q, qni, (W)0, (W)0, tid, prefix, epi);
        }
    }

    W* o0 = out + ((2 * ct) * (size_t)L + j) * n + slot0;
    W* o1 = out + ((2 * ct + 1) * (size_t)L + j) * n + slot0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
            const int idx = ((int)threadIdx.x + T * g) * 16 + k;
            V v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = acc0[g * 16 + k + e]; v1[e] = acc1[g * 16 + k + e]; }
            *reinterpret_cast<V*>(o0 + idx) = v0;
            *reinterpret_cast<V*>(o1 + idx) = v1;
        }
    }
}

}  // namespace alch
