// General-index Tensor kernels: crt / crtInv, l / lInv, mulG / divG for an arbitrary cyclotomic index m
// (the reference's real ciphertext indices are composite: H0' = F11648 ... H5' = F20475,
// examples/Common.hs:38-54, used by examples/HomomRLWR.hs:29-35 and examples/Tunnel.hs:26-32).
//
// Mathematics (toolkit sparse decompositions; conventions documented in include/alchemy_hip.h):
//   a ring element is a phi(m)-vector viewed as a mixed-radix array [phi(m_1)] .. [phi(m_k)], m_l = p_l^e_l, primes
//   ascending, first factor outermost; every operator is a Kronecker product of per-factor operators, so it runs as a
//   sequence of PASSES, each applying one small operator along one strided sub-axis of the whole array:
//     CRT_{p^e} = (DFT_{m'} (x) I_{p-1}) . T . (I_{m'} (x) CRT_p)        m' = p^(e-1)
//       pass "dense p-1":  CRT_p[i0-1][j0] = w_p^(i0 j0) along j0                         (odd p only)
//       passes "dense p":  DFT_{m'} as e-1 radix-p decimation-in-frequency stages; T and the stage twiddles are
//                          per-axis-position tables multiplied in front of the stage that follows them
//       p = 2:             the merged-twiddle Cooley-Tukey stages of the two-power engine (one product per butterfly)
//     L_{p^e} = L_p (x) I_{m'}, G_{p^e} = G_p (x) I_{m'}: column recurrences of length p-1 (prefix sums / differences)
//
// Mapping to CDNA4.  phi(m) <= 11520 for every index of the reference, so one limb-polynomial (46 KiB of 32-bit
// words) lives whole in LDS: one workgroup per limb-polynomial, one HBM read and one write per transform, three
// workgroups per CU.  Inside a pass a lane owns one group of r elements (r <= 13) in registers; lanes walk the
// innermost index, so LDS accesses of a wave are consecutive words whenever the pass's stride is >= 64.  The small
// operator matrices are wave-uniform and come through scalar loads.  Everything is integer VALU work (Montgomery,
// R = 2^32 / 2^64); nothing here is a dense contraction worth MFMA (r <= 13, 31-bit exact integers).
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_ntt.hpp"

namespace alch {

constexpr int GEN_MAXPASS = 24;
constexpr int GEN_MAXFACT = 8;
constexpr int GEN_T = 256;                  // threads per workgroup of every kernel in this file

enum GenKind : int { GK_DENSE = 0, GK_RADIX2 = 1 };

struct GenPass {
    int kind;          // GK_DENSE: r x r matrix along the sub-axis; GK_RADIX2: one merged-twiddle butterfly stage
    int r;             // group size
    u32 stride;        // distance between consecutive elements of a group
    u32 axis_stride;   // stride of the prime-power axis the pass belongs to (its rts)
    u32 axis_len;      // phi(p^e): twiddle tables are indexed by the position along that axis
    u32 mat_off;       // offset of the r*r matrix inside a limb's table block (forward and inverse blocks alike)
    u32 tw_off;        // offset of the axis_len twiddles, or 0xffffffff
};

struct GenFact { int p, e; u32 mp, dim, rts; };

template <typename W>
struct GenDev {
    u32 n;
    int npass, nfact;
    GenPass pass[GEN_MAXPASS];           // forward order; crtInv walks them backwards
    GenFact fact[GEN_MAXFACT];
    const W* tabf[MAXL];                 // per limb: forward matrices / twiddles (Montgomery form)
    const W* tabi[MAXL];                 // per limb: inverse matrices / twiddles
    W iscale_m[MAXL];                    // crtInv's closing scalar (2^-k of the radix-2 stages), Montgomery form
    const W* gcrt[MAXL];                 // CRT image of g (Montgomery form), n words
    const W* gcrt_inv[MAXL];
    W radinv_m[MAXL];                    // (odd radical of m)^-1 mod q_j in Montgomery form; 0 = not a unit (divG fails)
    u32 rad;
    int plain;                           // ring without CRT over an arbitrary modulus 2 <= q < 2^31 (Lol: a plaintext ring
                                         // Z_p): no Montgomery constants, products by `%`; radinv_m is then a plain residue
};

enum GenOp {
    GEN_CRT = 0, GEN_CRTINV = 1,
    GEN_CRT_DIGITS = 2,        // src = c2 (Pow) [ct][L][n]; data = digits [ct][L(i)][L(j)][n]; TrivGad decompose + reduce in the loader
    GEN_L = 3, GEN_LINV = 4, GEN_MULG_POW = 5, GEN_MULG_DEC = 6, GEN_DIVG_POW = 7, GEN_DIVG_DEC = 8
};

template <typename W>
struct GenCall {
    GenOp op;
    const DevRing<W>* ring;
    const GenDev<W>* gen;
    hipStream_t stream;
    W* data;
    const W* src;              // null: in place
    size_t first_poly, npoly;
    size_t elem_stride;        // GEN_L .. GEN_DIVG_*: process every elem_stride-th ring element (1 = all; 2 = the c0 of ciphertexts)
    bool balanced;
    bool with_diag;            // GEN_CRT_DIGITS: also transform the digits i == j (tunnel: no CRT copy of the source exists)
    u32 skip_mask;             // GEN_L .. GEN_DIVG_*: bit l set = leave prime-power factor l alone (tunnel: partial lInv)
    bool zdom;                 // the ring's "modulus" is 0: signed 64-bit integers (Pow / Dec operations only)
    int* fail_flag;            // device int, set when a divG is not possible (Lol's Nothing)
};

hipError_t gen_dispatch(const GenCall<u32>& c);
hipError_t gen_dispatch(const GenCall<u64>& c);

// ------------------------------------------------------------------------------------------------------
// passes
// ------------------------------------------------------------------------------------------------------
// y = M (tw . x) along one sub-axis (forward) or y = twinv . (Minv x) (inverse).  Values canonical in [0, q).
template <typename W, int R, bool INV>
__device__ __forceinline__ void gen_dense_pass(W* __restrict__ lds, const GenPass& P, const W* __restrict__ tab, u32 n, W q, W qni) {
    const W* __restrict__ M = tab + P.mat_off;
    const bool has_tw = P.tw_off != 0xffffffffu;
    const W* __restrict__ tw = tab + (has_tw ? P.tw_off : 0u);
    const u32 step = P.stride / P.axis_stride;
    for (u32 w = threadIdx.x; w < n / (u32)R; w += GEN_T) {
        const u32 lo = w % P.stride, hi = w / P.stride;
        const u32 base = hi * (u32)R * P.stride + lo;
        const u32 pos0 = (base / P.axis_stride) % P.axis_len;
        W x[R], y[R];
#pragma unroll
        for (int t = 0; t < R; ++t) x[t] = lds[base + (u32)t * P.stride];
        if (!INV && has_tw) {
#pragma unroll
            for (int t = 0; t < R; ++t) x[t] = csub(mont_mul_lazy(x[t], tw[pos0 + (u32)t * step], q, qni), q);
        }
#pragma unroll
        for (int s = 0; s < R; ++s) {
            W acc = csub(mont_mul_lazy(x[0], M[s * R], q, qni), q);
#pragma unroll
            for (int t = 1; t < R; ++t) acc = csub(acc + csub(mont_mul_lazy(x[t], M[s * R + t], q, qni), q), q);
            y[s] = acc;
        }
        if (INV && has_tw) {
#pragma unroll
            for (int s = 0; s < R; ++s) y[s] = csub(mont_mul_lazy(y[s], tw[pos0 + (u32)s * step], q, qni), q);
        }
#pragma unroll
        for (int s = 0; s < R; ++s) lds[base + (u32)s * P.stride] = y[s];
    }
}

// one Cooley-Tukey stage on a two-power axis: (x, y) -> (x + w y, x - w y), w = tw[axis position of y];
// inverse: (x, y) -> (x + y, (x - y) w^-1), the factor 1/2 is collected in GenDev::iscale_m.
template <typename W, bool INV>
__device__ __forceinline__ void gen_radix2_pass(W* __restrict__ lds, const GenPass& P, const W* __restrict__ tab, u32 n, W q, W qni) {
    const W* __restrict__ tw = tab + P.tw_off;
    for (u32 w = threadIdx.x; w < n / 2u; w += GEN_T) {
        const u32 lo = w % P.stride, hi = w / P.stride;
        const u32 ix = hi * 2u * P.stride + lo, iy = ix + P.stride;
        const W tv = tw[(iy / P.axis_stride) % P.axis_len];
        const W x = lds[ix], y = lds[iy];
        if (!INV) {
            const W t = csub(mont_mul_lazy(y, tv, q, qni), q);
            lds[ix] = csub(x + t, q);
            lds[iy] = csub(x + (q - t), q);
        } else {
            lds[ix] = csub(x + y, q);
            lds[iy] = csub(mont_mul_lazy((W)(x + (q - y)), tv, q, qni), q);
        }
    }
}

template <typename W, bool INV>
__device__ __forceinline__ void gen_run_pass(W* lds, const GenPass& P, const W* tab, u32 n, W q, W qni) {
    if (P.kind == GK_RADIX2) { gen_radix2_pass<W, INV>(lds, P, tab, n, q, qni); return; }
    switch (P.r) {                                 // wave-uniform
    case 2: gen_dense_pass<W, 2, INV>(lds, P, tab, n, q, qni); break;
    case 3: gen_dense_pass<W, 3, INV>(lds, P, tab, n, q, qni); break;
    case 4: gen_dense_pass<W, 4, INV>(lds, P, tab, n, q, qni); break;
    case 5: gen_dense_pass<W, 5, INV>(lds, P, tab, n, q, qni); break;
    case 6: gen_dense_pass<W, 6, INV>(lds, P, tab, n, q, qni); break;
    case 7: gen_dense_pass<W, 7, INV>(lds, P, tab, n, q, qni); break;
    case 10: gen_dense_pass<W, 10, INV>(lds, P, tab, n, q, qni); break;
    case 11: gen_dense_pass<W, 11, INV>(lds, P, tab, n, q, qni); break;
    case 12: gen_dense_pass<W, 12, INV>(lds, P, tab, n, q, qni); break;
    case 13: gen_dense_pass<W, 13, INV>(lds, P, tab, n, q, qni); break;
    default: break;                                // the host refuses indices with other odd primes
    }
}

// whole transform on an LDS-resident polynomial (canonical values in, canonical values out)
template <typename W, bool INV>
__device__ __forceinline__ void gen_transform(W* lds, const GenDev<W>& G, int j, W q, W qni) {
    if (!INV) {
        for (int ps = 0; ps < G.npass; ++ps) { gen_run_pass<W, false>(lds, G.pass[ps], G.tabf[j], G.n, q, qni); lds_barrier(); }
    } else {
        for (int ps = G.npass - 1; ps >= 0; --ps) { gen_run_pass<W, true>(lds, G.pass[ps], G.tabi[j], G.n, q, qni); lds_barrier(); }
    }
}

// ------------------------------------------------------------------------------------------------------
// batched crt / crtInv: one workgroup per limb-polynomial
// ------------------------------------------------------------------------------------------------------
template <typename W, bool INV>
__global__ void __launch_bounds__(GEN_T) k_gen_crt(DevRing<W> R, GenDev<W> G, W* data, const W* src, size_t first_poly) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const size_t p = first_poly + blockIdx.x;
    const int j = (int)(p % (size_t)R.L);
    const u32 n = G.n;
    W* poly = data + p * (size_t)n;
    const W* in = src ? src + p * (size_t)n : poly;
    const W q = R.mod[j].q, qni = R.mod[j].qni;
    for (u32 i = threadIdx.x; i < n; i += GEN_T) lds[i] = in[i];
    lds_barrier();
    gen_transform<W, INV>(lds, G, j, q, qni);
    if (INV) {
        const W sc = G.iscale_m[j];
        for (u32 i = threadIdx.x; i < n; i += GEN_T) poly[i] = csub(mont_mul_lazy(lds[i], sc, q, qni), q);
    } else {
        for (u32 i = threadIdx.x; i < n; i += GEN_T) poly[i] = lds[i];
    }
}

// crt of the reduced TrivGad digits with decompose + reduce in the loader (keySwitchQuadCirc, Eval.hs:133):
// workgroup = (ciphertext, source limb i, target limb j); the diagonal i == j is skipped (that digit is c2's own
// limb j, which the caller kept in the CRT basis).
template <typename W>
__global__ void __launch_bounds__(GEN_T) k_gen_crt_digits(DevRing<W> R, GenDev<W> G, const W* __restrict__ c2pow, W* __restrict__ digits, int balanced, int with_diag) {
    typedef typename Signed<W>::type SW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L;
    const size_t p = blockIdx.x;
    const int j = (int)(p % (size_t)L), i = (int)((p / (size_t)L) % (size_t)L);
    if (i == j && !with_diag) return;      // key switch: that digit is c2's own limb j, kept in the CRT basis by the caller
    const size_t ct = p / ((size_t)L * L);
    const u32 n = G.n;
    const W* src = c2pow + (ct * (size_t)L + i) * (size_t)n;
    W* dst = digits + p * (size_t)n;
    const W q = R.mod[j].q, qni = R.mod[j].qni, qi = R.mod[i].q, hqi = (qi - 1) >> 1;
    for (u32 k = threadIdx.x; k < n; k += GEN_T) {
        const W v = src[k];
        const SW z = v > hqi ? (SW)v - (SW)qi : (SW)v;
        SW r;
        if (balanced) r = z < 0 ? z + (SW)q : z;
        else { r = z % (SW)q; if (r < 0) r += (SW)q; }
        lds[k] = (W)r;
    }
    lds_barrier();
    gen_transform<W, false>(lds, G, j, q, qni);
    for (u32 k = threadIdx.x; k < n; k += GEN_T) dst[k] = lds[k];
}

// ------------------------------------------------------------------------------------------------------
// l / lInv, mulG / divG on the Pow and Dec bases: column recurrences along every odd-prime axis
// ------------------------------------------------------------------------------------------------------
// Arithmetic policy: Z_q with canonical residues, or the integers (signed 64-bit, q = 0; Lol's Tensor over Int64).
template <typename W, bool ZDOM>
struct ColArith {
    W q, qni, r2;
    int plain;
    __device__ __forceinline__ W add(W a, W b) const { if constexpr (ZDOM) return a + b; else return csub((W)(a + b), q); }
    __device__ __forceinline__ W sub(W a, W b) const { if constexpr (ZDOM) return a - b; else return csub((W)(a + (q - b)), q); }
    __device__ __forceinline__ W muls(W a, u32 c) const {           // times a small non-negative constant
        if constexpr (ZDOM) return a * (W)c;
        else if (plain) return (W)(((u64)a * c) % (u64)q);
        else { const W cm = csub(mont_mul_lazy((W)c, r2, q, qni), q); return csub(mont_mul_lazy(a, cm, q, qni), q); }
    }
};

template <typename W, bool ZDOM, int OP>
__device__ __forceinline__ void gen_column(W* __restrict__ x, u32 b, u32 s, int p, const ColArith<W, ZDOM>& A) {
    if (OP == GEN_L) {                                  // prefix sums
        for (int i = 1; i < p - 1; ++i) x[b + i * s] = A.add(x[b + i * s], x[b + (i - 1) * s]);
    } else if (OP == GEN_LINV) {                        // differences
        for (int i = p - 2; i >= 1; --i) x[b + i * s] = A.sub(x[b + i * s], x[b + (i - 1) * s]);
    } else if (OP == GEN_MULG_POW) {                    // (1 - zeta_p): out_i = a_i - a_{i-1} + a_{p-2}
        const W last = x[b + (p - 2) * s];
        for (int i = p - 2; i >= 1; --i) x[b + i * s] = A.add(A.sub(x[b + i * s], x[b + (i - 1) * s]), last);
        x[b] = A.add(x[b], last);
    } else if (OP == GEN_MULG_DEC) {                    // out_0 = 2 c_0 + sum_{i>=1} c_i, out_i = c_i - c_{i-1}
        W sum = 0;
        for (int i = 0; i < p - 1; ++i) sum = A.add(sum, x[b + i * s]);
        for (int i = p - 2; i >= 1; --i) x[b + i * s] = A.sub(x[b + i * s], x[b + (i - 1) * s]);
        x[b] = A.add(x[b], sum);
    } else if (OP == GEN_DIVG_POW) {                    // p b_i = p A_i - (i+1) A_total (A = prefix sums); / rad later
        W tot = 0;
        for (int i = 0; i < p - 1; ++i) tot = A.add(tot, x[b + i * s]);
        W run = 0;
        for (int i = 0; i < p - 1; ++i) {
            run = A.add(run, x[b + i * s]);
            x[b + i * s] = A.sub(A.muls(run, (u32)p), A.muls(tot, (u32)(i + 1)));
        }
    } else {                                            // GEN_DIVG_DEC: p c_0 = y_0 - sum Y_i, p c_i = p c_0 + p Y_i
        W run = 0, acc = 0;
        for (int i = 1; i < p - 1; ++i) { run = A.add(run, x[b + i * s]); acc = A.add(acc, run); }
        const W c0 = A.sub(x[b], acc);
        run = 0;
        for (int i = 1; i < p - 1; ++i) {
            run = A.add(run, x[b + i * s]);
            x[b + i * s] = A.add(c0, A.muls(run, (u32)p));
        }
        x[b] = c0;
    }
}

template <typename W, bool ZDOM, int OP>
__global__ void __launch_bounds__(GEN_T) k_gen_columns(DevRing<W> R, GenDev<W> G, W* data, size_t first_elem, size_t elem_stride, int* fail_flag, u32 skip_mask) {
    typedef typename Signed<W>::type SW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L;
    const size_t e = first_elem + (blockIdx.x / (unsigned)L) * elem_stride;
    const int j = (int)(blockIdx.x % (unsigned)L);
    const u32 n = G.n;
    W* poly = data + (e * (size_t)L + j) * (size_t)n;
    ColArith<W, ZDOM> A{R.mod[j].q, R.mod[j].qni, R.mod[j].r2, G.plain};
    for (u32 i = threadIdx.x; i < n; i += GEN_T) lds[i] = poly[i];
    lds_barrier();
    for (int l = 0; l < G.nfact; ++l) {
        const GenFact f = G.fact[l];
        if (f.p == 2 || ((skip_mask >> l) & 1u)) continue;
        const u32 step = f.mp * f.rts, span = f.dim * f.rts, ncol = n / (u32)(f.p - 1);
        for (u32 c = threadIdx.x; c < ncol; c += GEN_T) {
            const u32 o = c / step, in = c % step;
            gen_column<W, ZDOM, OP>(lds, o * span + in, step, f.p, A);
        }
        lds_barrier();
    }
    if (OP == GEN_DIVG_POW || OP == GEN_DIVG_DEC) {      // divide by the odd radical of m (lol-cpp: Z_q multiplies by rad^-1, Z checks)
        bool bad = false;
        if (G.rad > 1) {
            if constexpr (ZDOM) {
                for (u32 i = threadIdx.x; i < n; i += GEN_T) {
                    const SW v = (SW)lds[i];
                    if (v % (SW)G.rad) bad = true; else lds[i] = (W)(v / (SW)G.rad);
                }
            } else {
                const W ri = G.radinv_m[j];
                if (ri == 0) bad = true;
                else if (G.plain) for (u32 i = threadIdx.x; i < n; i += GEN_T) lds[i] = (W)(((u64)lds[i] * (u64)ri) % (u64)A.q);
                else for (u32 i = threadIdx.x; i < n; i += GEN_T) lds[i] = csub(mont_mul_lazy(lds[i], ri, A.q, A.qni), A.q);
            }
        }
        if (bad) atomicOr(fail_flag, 1);
    }
    for (u32 i = threadIdx.x; i < n; i += GEN_T) poly[i] = lds[i];
}

// ------------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------------
template <typename W, bool ZDOM, int OP>
inline hipError_t gen_launch_columns(const GenCall<W>& c, size_t lds_bytes) {
    auto k = k_gen_columns<W, ZDOM, OP>;
    hipError_t e = set_lds(k, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(GEN_T), lds_bytes, c.stream, *c.ring, *c.gen, c.data, c.first_poly,
                       c.elem_stride ? c.elem_stride : (size_t)1, c.fail_flag, c.skip_mask);
    return hipGetLastError();
}

template <typename W>
inline hipError_t gen_run(const GenCall<W>& c) {
    const size_t lds_bytes = (size_t)c.gen->n * sizeof(W);
    hipError_t e;
    switch (c.op) {
    case GEN_CRT: {
        auto k = k_gen_crt<W, false>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(GEN_T), lds_bytes, c.stream, *c.ring, *c.gen, c.data, c.src, c.first_poly);
        break;
    }
    case GEN_CRTINV: {
        auto k = k_gen_crt<W, true>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(GEN_T), lds_bytes, c.stream, *c.ring, *c.gen, c.data, c.src, c.first_poly);
        break;
    }
    case GEN_CRT_DIGITS: {
        auto k = k_gen_crt_digits<W>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(GEN_T), lds_bytes, c.stream, *c.ring, *c.gen, c.src, c.data, c.balanced ? 1 : 0, c.with_diag ? 1 : 0);
        break;
    }
    // npoly = number of (element, limb) workgroups; first_poly = first ELEMENT here
    case GEN_L: return c.zdom ? gen_launch_columns<W, true, GEN_L>(c, lds_bytes) : gen_launch_columns<W, false, GEN_L>(c, lds_bytes);
    case GEN_LINV: return c.zdom ? gen_launch_columns<W, true, GEN_LINV>(c, lds_bytes) : gen_launch_columns<W, false, GEN_LINV>(c, lds_bytes);
    case GEN_MULG_POW: return c.zdom ? gen_launch_columns<W, true, GEN_MULG_POW>(c, lds_bytes) : gen_launch_columns<W, false, GEN_MULG_POW>(c, lds_bytes);
    case GEN_MULG_DEC: return c.zdom ? gen_launch_columns<W, true, GEN_MULG_DEC>(c, lds_bytes) : gen_launch_columns<W, false, GEN_MULG_DEC>(c, lds_bytes);
    case GEN_DIVG_POW: return c.zdom ? gen_launch_columns<W, true, GEN_DIVG_POW>(c, lds_bytes) : gen_launch_columns<W, false, GEN_DIVG_POW>(c, lds_bytes);
    case GEN_DIVG_DEC: return c.zdom ? gen_launch_columns<W, true, GEN_DIVG_DEC>(c, lds_bytes) : gen_launch_columns<W, false, GEN_DIVG_DEC>(c, lds_bytes);
    }
    return hipGetLastError();
}

}  // namespace alch
