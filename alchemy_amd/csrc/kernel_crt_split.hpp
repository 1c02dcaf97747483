// k_crt_split: Tensor crt / crtInv for a ring twice as large as one LDS-resident transform
// (n = 2^16 with 32-bit residues, n = 2^15 with 64-bit residues: a limb-polynomial is 256 KiB).
//
// After stage 0 of the merged-twist Cooley-Tukey transform the two halves of the data are independent
// sub-transforms of size n/2 that use the big ring's twiddles tw[(2 + half) << s + h]  (ntt_pass `prefix`);
// the inverse runs the same thing backwards.  One workgroup owns one limb-polynomial and works IN PLACE:
//   crt    : stage 0 from HBM (x, y) -> (x + w y) into LDS, (x - w y) back to HBM in place;
//            half 0: LDS sub-transform -> store;  half 1: reload, LDS sub-transform -> store.
//   crtInv : half 0: load, LDS inverse sub-transform (no n^-1), store in place;  half 1: load, inverse
//            sub-transform, result stays in LDS;  stage 0 + n^-1: a from HBM, b from LDS -> both halves.
// Every HBM word a lane re-reads was written by that same lane (same index pattern), so no cross-lane
// ordering is needed.  Traffic: 1.5 reads + 1.5 writes of the polynomial instead of 1 + 1.
#pragma once
#include <hip/hip_runtime.h>
#include "ntt_engine.hpp"

namespace alch {

template <int LOGN, typename W, bool INVERSE>
__global__ void __launch_bounds__(Geo<LOGN - 1>::T) k_crt_split(DevRing<W> R, W* __restrict__ data, size_t first_poly) {
    constexpr int LOGM = LOGN - 1;
    typedef Geo<LOGM> G;
    typedef typename Vec4<W>::type V;
    constexpr int VL = Vec4<W>::LANES;
    constexpr int M = G::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const size_t p = first_poly + blockIdx.x;
    const int j = (int)(p % (size_t)R.L);
    W* lo = data + p * (size_t)(2 * M);
    W* hi = lo + M;
    const ModP<W> m = R.mod[j];
    const W q = m.q, qni = m.qni;
    const int tid = threadIdx.x;

    if constexpr (!INVERSE) {
        const W w1 = R.twf[j][1];
#pragma unroll
        for (int r = 0; r < G::E / VL; ++r) {
            const int idx = (tid + G::T * r) * VL;
            const V x = *reinterpret_cast<const V*>(lo + idx), y = *reinterpret_cast<const V*>(hi + idx);
            V u0, u1;
#pragma unroll
            for (int e = 0; e < VL; ++e) {
                const W t = csub(mont_mul_lazy(y[e], w1, q, qni), q);
                u0[e] = x[e] + t;                                   // inputs are reduced: [0, 2q)
                u1[e] = csub((W)(x[e] + (q - t)), q);               // stored reduced
            }
            *reinterpret_cast<V*>(&lds[swz<LOGM>(idx)]) = u0;
            *reinterpret_cast<V*>(hi + idx) = u1;
        }
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            W* dst = half ? hi : lo;
            if (half) {
                stage_in<LOGM, W>(lds, [&](int idx) { return *reinterpret_cast<const V*>(hi + idx); });
            }
            lds_barrier();
            int t2 = tid;
            asm volatile("" : "+v"(t2));
            ntt_forward<LOGM, W, false>(lds, fwd_tw(R, j), fwd_twm(R, j), q, qni, t2, NoEpilogue(), 2 + half);
#pragma unroll
            for (int r = 0; r < G::E / VL; ++r) {
                const int idx = (t2 + G::T * r) * VL;
                V v = *reinterpret_cast<const V*>(&lds[swz<LOGM>(idx)]);
#pragma unroll
                for (int e = 0; e < VL; ++e) v[e] = csub(v[e], q);
                *reinterpret_cast<V*>(dst + idx) = v;
            }
            lds_barrier();                      // LDS is refilled by other lanes' loads next
        }
    } else {
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const W* src = half ? hi : lo;
            stage_in<LOGM, W>(lds, [&](int idx) { return *reinterpret_cast<const V*>(src + idx); });
            lds_barrier();
            int t2 = tid;
            asm volatile("" : "+v"(t2));
            ntt_inverse<LOGM, W, false, false, false>(lds, inv_tw(R, j), q, qni, (W)0, (W)0, t2, NoEpilogue(), NoHook(), 2 + half);
            if (half == 0) {
#pragma unroll
                for (int r = 0; r < G::E / VL; ++r) {
                    const int idx = (t2 + G::T * r) * VL;
                    V v = *reinterpret_cast<const V*>(&lds[swz<LOGM>(idx)]);
#pragma unroll
                    for (int e = 0; e < VL; ++e) v[e] = csub(v[e], q);
                    *reinterpret_cast<V*>(lo + idx) = v;
                }
                lds_barrier();
            }
        }
        // stage 0 with n^-1 folded in: c[k] = (a + b) n^-1,  c[k + n/2] = (a - b) tw_inv[1] n^-1
        const W ninv = R.ninv_m[j], w1ninv = R.w1ninv_m[j];
#pragma unroll
        for (int r = 0; r < G::E / VL; ++r) {
            const int idx = (tid + G::T * r) * VL;
            const V a = *reinterpret_cast<const V*>(lo + idx);
            const V bb = *reinterpret_cast<const V*>(&lds[swz<LOGM>(idx)]);
            V c0, c1;
#pragma unroll
            for (int e = 0; e < VL; ++e) {
                const W b = csub(bb[e], q);
                c0[e] = csub(mont_mul_lazy((W)(a[e] + b), ninv, q, qni), q);
                c1[e] = csub(mont_mul_lazy((W)(a[e] - b + q), w1ninv, q, qni), q);
            }
            *reinterpret_cast<V*>(lo + idx) = c0;
            *reinterpret_cast<V*>(hi + idx) = c1;
        }
    }
}

// Kernel A for split rings (round 3): c2 = a1 b1 s for limb i of one ciphertext with the product formed in the inverse transform's
// loader, crtInv as k_crt_split's inverse (two half-size sub-transforms, stage 0 + n^-1 s last), canonical Pow-basis residues to
// c2pow[ct][L][n] -- the element-wise tensor kernel, the CRT-basis copy of c2 and the separate crtInv launch of the composed path are
// gone (6 MiB written and 4.5 MiB re-read per op at n = 2^16, L = 6).  spre = s R^2 per limb; the scalar rides on the n^-1 constants.
template <int LOGN, typename W>
__global__ void __launch_bounds__(Geo<LOGN - 1>::T) k_tensor_crtinv_split(DevRing<W> R, const W* __restrict__ a, const W* __restrict__ b,
                                                                          W* __restrict__ c2pow, Scal<W> spre) {
    constexpr int LOGM = LOGN - 1;
    typedef Geo<LOGM> G;
    typedef typename Vec4<W>::type V;
    constexpr int VL = Vec4<W>::LANES;
    constexpr int M = G::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L;
    const size_t p = blockIdx.x, ct = p / (size_t)L;
    const int j = (int)(p % (size_t)L);
    const size_t n = (size_t)(2 * M);
    const W* a1 = a + ((2 * ct + 1) * (size_t)L + j) * n;
    const W* b1 = b + ((2 * ct + 1) * (size_t)L + j) * n;
    W* lo = c2pow + p * n;
    W* hi = lo + M;
    const ModP<W> m = R.mod[j];
    const W q = m.q, qni = m.qni;
    const int tid = threadIdx.x;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const W* sa = a1 + (size_t)half * M;
        const W* sb = b1 + (size_t)half * M;
        stage_in<LOGM, W>(lds, [&](int idx) {
            const V va = *reinterpret_cast<const V*>(sa + idx), vb = *reinterpret_cast<const V*>(sb + idx);
            V v;
#pragma unroll
            for (int e = 0; e < VL; ++e) v[e] = mont_mul_lazy(va[e], vb[e], q, qni);        // a1 b1 R^-1, lazy
            return v;
        });
        lds_barrier();
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        ntt_inverse<LOGM, W, false, false, false>(lds, inv_tw(R, j), q, qni, (W)0, (W)0, t2, NoEpilogue(), NoHook(), 2 + half);
        if (half == 0) {
#pragma unroll
            for (int r = 0; r < G::E / VL; ++r) {
                const int idx = (t2 + G::T * r) * VL;
                V v = *reinterpret_cast<const V*>(&lds[swz<LOGM>(idx)]);
#pragma unroll
                for (int e = 0; e < VL; ++e) v[e] = csub(v[e], q);
                *reinterpret_cast<V*>(lo + idx) = v;
            }
            lds_barrier();
        }
    }
    // stage 0 with n^-1 s R^2 folded in (the product above carries R^-1, the Montgomery product another)
    const W ninv = mont_mul(R.ninv_m[j], spre.v[j], m), w1ninv = mont_mul(R.w1ninv_m[j], spre.v[j], m);
#pragma unroll
    for (int r = 0; r < G::E / VL; ++r) {
        const int idx = (tid + G::T * r) * VL;
        const V x = *reinterpret_cast<const V*>(lo + idx);
        const V yy = *reinterpret_cast<const V*>(&lds[swz<LOGM>(idx)]);
        V c0, c1;
#pragma unroll
        for (int e = 0; e < VL; ++e) {
            const W y = csub(yy[e], q);
            c0[e] = csub(mont_mul_lazy((W)(x[e] + y), ninv, q, qni), q);
            c1[e] = csub(mont_mul_lazy((W)(x[e] - y + q), w1ninv, q, qni), q);
        }
        *reinterpret_cast<V*>(lo + idx) = c0;
        *reinterpret_cast<V*>(hi + idx) = c1;
    }
}

// crt of the reduced TrivGad digits for the unfused key switch on split rings, decompose fused into the loader:
// workgroup p = (ciphertext, digit i, target limb j) reads limb i of c2 (Pow basis, [0, q_i)), takes the centred
// lift, reduces it mod q_j on the fly (Lol: decompose, then reduce) and runs the forward transform of limb j into
// digits[p].  Saves the separate decompose pass (L^2 polynomials written and read back).
template <int LOGN, typename W>
__global__ void __launch_bounds__(Geo<LOGN - 1>::T) k_crt_split_digits(DevRing<W> R, const W* __restrict__ c2pow, W* __restrict__ digits,
                                                                       int balanced) {
    constexpr int LOGM = LOGN - 1;
    typedef Geo<LOGM> G;
    typedef typename Vec4<W>::type V;
    typedef typename Signed<W>::type SW;
    constexpr int VL = Vec4<W>::LANES;
    constexpr int M = G::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const size_t p = blockIdx.x;
    const int L = R.L;
    const int j = (int)(p % (size_t)L), i = (int)((p / (size_t)L) % (size_t)L);
    if (i == j) return;          // the diagonal digit is c2's own limb in the CRT basis: the inner product reads it there
    const size_t ct = p / ((size_t)L * L);
    const W* src = c2pow + (ct * (size_t)L + i) * (size_t)(2 * M);
    W* lo = digits + p * (size_t)(2 * M);
    W* hi = lo + M;
    const ModP<W> m = R.mod[j];
    const W q = m.q, qni = m.qni, qi = R.mod[i].q, hqi = (qi - 1) >> 1;
    const int tid = threadIdx.x;
    auto reduce = [&](W v) -> W {                         // centred lift of v mod q_i, reduced mod q_j
        const SW z = v > hqi ? (SW)v - (SW)qi : (SW)v;
        if (balanced) return z < 0 ? (W)(z + (SW)q) : (W)z;
        SW r = z % (SW)q;
        return r < 0 ? (W)(r + (SW)q) : (W)r;
    };
    const W w1 = R.twf[j][1];
#pragma unroll
    for (int r = 0; r < G::E / VL; ++r) {
        const int idx = (tid + G::T * r) * VL;
        const V x = *reinterpret_cast<const V*>(src + idx), y = *reinterpret_cast<const V*>(src + M + idx);
        V u0, u1;
#pragma unroll
        for (int e = 0; e < VL; ++e) {
            const W xr = reduce(x[e]);
            const W t = csub(mont_mul_lazy(reduce(y[e]), w1, q, qni), q);
            u0[e] = xr + t;
            u1[e] = csub((W)(xr + (q - t)), q);
        }
        *reinterpret_cast<V*>(&lds[swz<LOGM>(idx)]) = u0;
        *reinterpret_cast<V*>(hi + idx) = u1;
    }
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        W* dst = half ? hi : lo;
        if (half) {
            stage_in<LOGM, W>(lds, [&](int idx) { return *reinterpret_cast<const V*>(hi + idx); });
        }
        lds_barrier();
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        ntt_forward<LOGM, W, false>(lds, fwd_tw(R, j), fwd_twm(R, j), q, qni, t2, NoEpilogue(), 2 + half);
#pragma unroll
        for (int r = 0; r < G::E / VL; ++r) {
            const int idx = (t2 + G::T * r) * VL;
            V v = *reinterpret_cast<const V*>(&lds[swz<LOGM>(idx)]);
#pragma unroll
            for (int e = 0; e < VL; ++e) v[e] = csub(v[e], q);
            *reinterpret_cast<V*>(dst + idx) = v;
        }
        lds_barrier();
    }
}

// Fused digit transforms + hint inner product for split rings (n = 2^16 / 32-bit, n = 2^15 / 64-bit): the second half of
// keySwitchQuadCirc (Eval.hs:133) without the digit round trip through HBM.  One workgroup per (ciphertext, limb j, half of the
// slots), XCD-aware numbering as k_ks_accum: the 2 x 32 accumulators of the half stay in registers; per digit i != j the loader
// takes the centred lift of c2's limb i (Pow basis), reduces it mod q_j, runs stage 0 for this half (x +- w1 y: both halves of the
// coefficients are read, from L2 -- the 2L items of a ciphertext share them) and the half-size sub-transform (prefix 2 + half)
// in LDS; the closing pass's epilogue multiplies by the hint rows.  out holds c0, c1 on entry (element-wise tensor product) and
// the diagonal digit comes from c2's CRT-basis copy, as in the composed path this replaces (k_crt_split_digits + k_hint_mac:
// L (L-1) limb-polynomials of 256 KiB written and read back per ciphertext).  dup: leading limbs added by modSwitch (their c2 is 0).
// FROM_OPS (round 3, with k_tensor_crtinv_split in front): c0 = a0 b0 s, c1 = (a0 b1 + a1 b0) s and the diagonal digit c2_j = a1 b1 s are
// formed here from the operands (opa, opb: [ct][2][L][n], CRT basis; spre = s R^2 per limb) instead of being read back from out / c2crt.
template <int LOGN, typename W, bool BALANCED, bool FROM_OPS = false>
__global__ void __launch_bounds__(Geo<LOGN - 1>::T)
k_ks_accum_split(DevRing<W> R, const W* __restrict__ c2pow, const W* __restrict__ c2crt, const W* __restrict__ hint, W* __restrict__ out,
                 unsigned nct, int dup, const W* __restrict__ opa, const W* __restrict__ opb, Scal<W> spre) {
    constexpr int LOGM = LOGN - 1;
    typedef Geo<LOGM> G;
    typedef typename Vec4<W>::type V;
    typedef typename Signed<W>::type SW;
    constexpr int VL = Vec4<W>::LANES;
    constexpr int M = G::N, NG = G::E / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L;
    const unsigned per = 16u * (unsigned)L;
    const unsigned grp = blockIdx.x / per, rem = blockIdx.x % per, which = rem >> 3;
    const int j = (int)(which >> 1), half = (int)(which & 1u);
    const size_t ct = (size_t)grp * 8u + (rem & 7u);
    if (ct >= nct) return;
    const ModP<W> m = R.mod[j];
    const W q = m.q, qni = m.qni;
    const size_t n = (size_t)(2 * M), Ln = (size_t)L * n, hoff = (size_t)half * M;
    const W* hj = hint + (size_t)j * n + hoff;                 // + (2 i + c) * Ln
    W* o0 = out + ((2 * ct) * (size_t)L + j) * n + hoff;
    W* o1 = out + ((2 * ct + 1) * (size_t)L + j) * n + hoff;
    W acc0[G::E], acc1[G::E];
    if constexpr (FROM_OPS) {   // the tensor product itself, from the operands
        const W* a0 = opa + ((2 * ct) * (size_t)L + j) * n + hoff;
        const W* a1 = opa + ((2 * ct + 1) * (size_t)L + j) * n + hoff;
        const W* b0 = opb + ((2 * ct) * (size_t)L + j) * n + hoff;
        const W* b1 = opb + ((2 * ct + 1) * (size_t)L + j) * n + hoff;
        const W* h0 = hj + (size_t)(2 * j) * Ln;
        const W* h1 = hj + (size_t)(2 * j + 1) * Ln;
        const W s2 = spre.v[j];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int k = 0; k < 16; k += VL) {
                const int idx = (threadIdx.x + G::T * g) * 16 + k;
                const V va0 = *reinterpret_cast<const V*>(a0 + idx), va1 = *reinterpret_cast<const V*>(a1 + idx);
                const V vb0 = *reinterpret_cast<const V*>(b0 + idx), vb1 = *reinterpret_cast<const V*>(b1 + idx);
                const V vh0 = *reinterpret_cast<const V*>(h0 + idx), vh1 = *reinterpret_cast<const V*>(h1 + idx);
#pragma unroll
                for (int e = 0; e < VL; ++e) {
                    const W x0 = mont_mul(va0[e], s2, m), x1 = mont_mul(va1[e], s2, m);          // a s R
                    const W c2 = mont_mul(vb1[e], x1, m);
                    acc0[g * 16 + k + e] = add_mod(mont_mul(vb0[e], x0, m), mont_mul(c2, vh0[e], m), q);
                    acc1[g * 16 + k + e] = add_mod(add_mod(mont_mul(vb1[e], x0, m), mont_mul(vb0[e], x1, m), q), mont_mul(c2, vh1[e], m), q);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {   // c0, c1 (already in out) + the diagonal digit c2_j * hint_j
        const W* d = c2crt + (ct * (size_t)L + j) * n + hoff;
        const W* h0 = hj + (size_t)(2 * j) * Ln;
        const W* h1 = hj + (size_t)(2 * j + 1) * Ln;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int k = 0; k < 16; k += VL) {
                const int idx = (threadIdx.x + G::T * g) * 16 + k;
                const V v0 = *reinterpret_cast<const V*>(o0 + idx), v1 = *reinterpret_cast<const V*>(o1 + idx);
                const V vd = *reinterpret_cast<const V*>(d + idx);
                const V vh0 = *reinterpret_cast<const V*>(h0 + idx), vh1 = *reinterpret_cast<const V*>(h1 + idx);
#pragma unroll
                for (int e = 0; e < VL; ++e) {
                    acc0[g * 16 + k + e] = csub(v0[e] + csub(mont_mul_lazy(vd[e], vh0[e], q, qni), q), q);
                    acc1[g * 16 + k + e] = csub(v1[e] + csub(mont_mul_lazy(vd[e], vh1[e], q, qni), q), q);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const W w1 = half ? q - R.twf[j][1] : R.twf[j][1];          // upper half: x - w1 y = x + (-w1) y
    for (int i = dup; i < L; ++i) {
        if (i == j) continue;
        const W qi = R.mod[i].q, hqi = (qi - 1) >> 1;
        const W* src = c2pow + (ct * (size_t)L + i) * n;
        auto reduce = [&](W v) -> W {                         // centred lift of v mod q_i, reduced mod q_j
            const SW z = v > hqi ? (SW)v - (SW)qi : (SW)v;
            if constexpr (BALANCED) return z < 0 ? (W)(z + (SW)q) : (W)z;
            else { SW r = z % (SW)q; return r < 0 ? (W)(r + (SW)q) : (W)r; }
        };
        lds_barrier();                                        // the previous transform's last pass has finished reading LDS
#pragma unroll
        for (int r = 0; r < G::E / VL; ++r) {
            const int idx = (threadIdx.x + G::T * r) * VL;
            const V x = *reinterpret_cast<const V*>(src + idx), y = *reinterpret_cast<const V*>(src + M + idx);
            V u;
#pragma unroll
            for (int e = 0; e < VL; ++e) u[e] = reduce(x[e]) + csub(mont_mul_lazy(reduce(y[e]), w1, q, qni), q);     // [0, 2q)
            *reinterpret_cast<V*>(&lds[swz<LOGM>(idx)]) = u;
        }
        lds_barrier();
        const W* h0 = hj + (size_t)(2 * i) * Ln;
        const W* h1 = hj + (size_t)(2 * i + 1) * Ln;
        auto twf = fwd_tw(R, j);
        auto twm = fwd_twm(R, j);
        int tid = threadIdx.x;
        asm volatile("" : "+s"(twf), "+s"(twm), "+v"(tid));  // keep pass addresses / twiddles inside the digit loop (VGPR pressure)
        ntt_forward<LOGM, W, true, true>(lds, twf, twm, q, qni, tid, [&acc0, &acc1, h0, h1, q, qni](int g, int base, W* x) {
#pragma unroll
            for (int k = 0; k < 16; k += VL) {
                const V vh0 = *reinterpret_cast<const V*>(h0 + base + k), vh1 = *reinterpret_cast<const V*>(h1 + base + k);
#pragma unroll
                for (int e = 0; e < VL; ++e) {
                    acc0[g * 16 + k + e] = csub(acc0[g * 16 + k + e] + csub(mont_mul_lazy(x[k + e], vh0[e], q, qni), q), q);
                    acc1[g * 16 + k + e] = csub(acc1[g * 16 + k + e] + csub(mont_mul_lazy(x[k + e], vh1[e], q, qni), q), q);
                }
            }
        }, 2 + half);
    }
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

template <typename W, int LOGN>
inline hipError_t run_call_split(const NttCall<W>& c) {
    typedef Geo<LOGN - 1> G;
    const size_t lds_bytes = (size_t)lds_words<LOGN - 1>() * sizeof(W);
    hipError_t e;
    if (c.op == OP_CRT) {
        auto k = k_crt_split<LOGN, W, false>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(G::T), lds_bytes, c.stream, *c.ring, c.data, c.first_poly);
    } else if (c.op == OP_CRTINV) {
        auto k = k_crt_split<LOGN, W, true>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(G::T), lds_bytes, c.stream, *c.ring, c.data, c.first_poly);
    } else if (c.op == OP_CRT_DIGITS) {
        auto k = k_crt_split_digits<LOGN, W>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(G::T), lds_bytes, c.stream, *c.ring, c.src, c.data, c.balanced ? 1 : 0);
    } else if (c.op == OP_KS_SPLIT) {
        const size_t groups = (c.nct + 7) / 8;
        const unsigned grid = (unsigned)(groups * 16 * (size_t)c.ring->L);
        if (c.balanced) {
            auto k = k_ks_accum_split<LOGN, W, true>;
            if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
            hipLaunchKernelGGL(k, dim3(grid), dim3(G::T), lds_bytes, c.stream, *c.ring, c.src, c.a, c.hint, c.out, (unsigned)c.nct, c.dup,
                               (const W*)nullptr, (const W*)nullptr, Scal<W>{});
        } else {
            auto k = k_ks_accum_split<LOGN, W, false>;
            if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
            hipLaunchKernelGGL(k, dim3(grid), dim3(G::T), lds_bytes, c.stream, *c.ring, c.src, c.a, c.hint, c.out, (unsigned)c.nct, c.dup,
                               (const W*)nullptr, (const W*)nullptr, Scal<W>{});
        }
    } else if (c.op == OP_TENSOR_INTT) {   // split rings: tensor product in the inverse transform's loader (digits = canonical Pow residues)
        auto k = k_tensor_crtinv_split<LOGN, W>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)(c.nct * (size_t)c.ring->L)), dim3(G::T), lds_bytes, c.stream, *c.ring, c.a, c.b, (W*)c.digits,
                           c.spre_r2);
    } else if (c.op == OP_KS_ACCUM) {      // ... and the key switch with the tensor part from the operands
        if (c.dup != 0) return hipErrorInvalidValue;
        const size_t groups = (c.nct + 7) / 8;
        const unsigned grid = (unsigned)(groups * 16 * (size_t)c.ring->L);
        if (c.balanced) {
            auto k = k_ks_accum_split<LOGN, W, true, true>;
            if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
            hipLaunchKernelGGL(k, dim3(grid), dim3(G::T), lds_bytes, c.stream, *c.ring, (const W*)c.digits, (const W*)nullptr, c.hint, c.out,
                               (unsigned)c.nct, 0, c.a, c.b, c.spre_r2);
        } else {
            auto k = k_ks_accum_split<LOGN, W, false, true>;
            if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
            hipLaunchKernelGGL(k, dim3(grid), dim3(G::T), lds_bytes, c.stream, *c.ring, (const W*)c.digits, (const W*)nullptr, c.hint, c.out,
                               (unsigned)c.nct, 0, c.a, c.b, c.spre_r2);
        }
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace alch
