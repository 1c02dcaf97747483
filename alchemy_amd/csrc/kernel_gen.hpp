// General-index Tensor kernels: crt / crtInv, l / lInv, mulG / divG for an arbitrary cyclotomic index m
// (the reference's real ciphertext indices are composite: H0' = F11648 ... H5' = F20475,
// examples/Common.hs:38-54, used by examples/HomomRLWR.hs:29-35 and examples/Tunnel.hs:26-32).
//
// Mathematics (toolkit sparse decompositions; conventions documented in include/alchemy_hip.h):
//   a ring element is a phi(m)-vector viewed as a mixed-radix array [phi(m_1)] .. [phi(m_k)], m_l = p_l^e_l, primes
//   ascending, first factor outermost; every operator is a Kronecker product of per-factor operators, so it runs as a
//   sequence of PASSES, each applying one small operator along one strided sub-axis of the whole array:
//     CRT_{p^e} = (DFT_{m'} (x) I_{p-1}) . T . (I_{m'} (x) CRT_p)        m' = p^(e-1)
//       pass "CRT_p":  CRT_p[i0-1][j0] = w_p^(i0 j0) along j0, computed through the +/- symmetry of the roots (odd p only)
//       passes "DFT_p": DFT_{m'} as e-1 radix-p decimation-in-frequency stages; T and the stage twiddles are
//                       per-axis-position tables multiplied in front of the stage that follows them
//       p = 2:          the merged-twiddle Cooley-Tukey stages of the two-power engine, up to three per pass on
//                       register-resident groups of eight (one product per butterfly)
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
constexpr int GEN_T_SMALL = 128;            // the transform kernels on rings of at most 18 KiB (gen_run)
constexpr int GEN_T = 256;                  // threads per workgroup of every kernel in this file

enum GenKind : int {
    GK_R2BLOCK = 1,    // K <= 3 merged-twiddle Cooley-Tukey stages of a two-power axis on register-resident groups of 2^K
    GK_SYM_CRT = 2,    // CRT_p (p odd, r = p - 1) through the +/- symmetry of the roots: half the products of a dense matrix
    GK_SYM_DFT = 3     // DFT_p (r = p) the same way, behind a per-axis-position twiddle table
};

struct GenPass {
    int kind;
    int r;             // group size
    int aux;           // GK_R2BLOCK: index s0 of the block's first stage
    u32 stride;        // distance between consecutive elements of a group
    u32 axis_stride;   // stride of the prime-power axis the pass belongs to (its rts)
    u32 axis_len;      // phi(p^e): twiddle tables are indexed by the position along that axis
    u32 mat_off;       // offset of the r*r matrix inside a limb's table block (forward and inverse blocks alike)
    u32 tw_off;        // offset of the axis_len twiddles, or 0xffffffff
    u32 rcp_stride, rcp_axis_stride, rcp_axis_len;   // floor(2^32 / d) + 1: w / d == mulhi(w, rcp) for w, d < 2^16
};

// Exact w / d for w * d < 2^32 (every index here is below phi(m) <= 40960 < 2^16): one v_mul_hi instead of a runtime
// division (~30 VALU instructions on gfx950), four of which a pass would otherwise pay per group.
__device__ __forceinline__ u32 fdiv(u32 w, u32 d, u32 rcp) { return d == 1 ? w : __umulhi(w, rcp); }

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
    int smallq;                          // 1: every modulus < 2^32 / sqrt(6): six-term lazy accumulation in the p = 13 passes (dense_row);
                                         // 2: every modulus < 2^32 / 6 as well: no conditional subtraction in front of the reductions
    int nt;                              // host side: threads per workgroup of the transform kernels (0 = by ring size: gen_threads)
    int plain;                           // ring without CRT over an arbitrary modulus 2 <= q < 2^31 (Lol: a plaintext ring
                                         // Z_p): no Montgomery constants, products by `%`; radinv_m is then a plain residue
};

enum GenOp {
    GEN_CRT_BASE2 = -1,        // src = Pow elements [el][L][n]; data = digits [el][D][L][n]; BaseBGad 2 decompose + reduce in the loader
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
    Scal<u32> b2_first, b2_kd; // GEN_CRT_BASE2: first digit and digit count of every limb
    u32 b2_D;                  // GEN_CRT_BASE2: digits per element
    int src_limbs, src_first;  // GEN_CRT_DIGITS: the source elements hold limbs src_first .. src_first + src_limbs - 1 (0 = all L)
    u32 skip_mask;             // GEN_L .. GEN_DIVG_*: bit l set = leave prime-power factor l alone (tunnel: partial lInv)
    bool zdom;                 // the ring's "modulus" is 0: signed 64-bit integers (Pow / Dec operations only)
    int* fail_flag;            // device int, set when a divG is not possible (Lol's Nothing)
};

hipError_t gen_dispatch(const GenCall<u32>& c);
hipError_t gen_dispatch(const GenCall<u64>& c);
template <typename W> struct GenKsArgs;
hipError_t gen_rescale_lin_dispatch(const DevRing<u32>& R, const GenDev<u32>& G, const u32* in, u32* res, u32* out, const DropTab<u32>& D, int dec_c0, size_t nelem, hipStream_t stream, bool pow_out = false);
hipError_t gen_rescale_lin_dispatch(const DevRing<u64>& R, const GenDev<u64>& G, const u64* in, u64* res, u64* out, const DropTab<u64>& D, int dec_c0, size_t nelem, hipStream_t stream, bool pow_out = false);
template <typename W> struct GenTunArgs;
hipError_t gen_tunnel_ks_dispatch(const DevRing<u32>& R, const GenDev<u32>& GE, const GenTunArgs<u32>& A, size_t nct, int ng, hipStream_t stream);
hipError_t gen_ks_dispatch(const DevRing<u32>& R, const GenDev<u32>& G, const GenKsArgs<u32>& A, size_t nct, hipStream_t stream);
hipError_t gen_ks_dispatch(const DevRing<u64>& R, const GenDev<u64>& G, const GenKsArgs<u64>& A, size_t nct, hipStream_t stream);

// ------------------------------------------------------------------------------------------------------
// passes
// ------------------------------------------------------------------------------------------------------
// sum_t x[t] * M[t] (Montgomery: M holds M R) for canonical x, M < q < 2^31.
// 32-bit words: four products are summed in 64 bits before one reduction -- 4 q^2 < 2^64, the sum's high word is < 2q, one
// conditional subtraction brings the sum below q 2^32, then a single Montgomery reduction: 10 instructions per four terms
// instead of 20.  64-bit words: one product at a time.
// LVL 1 (every modulus of the ring below 2^32 / sqrt(6) = 1 753 413 056: all of the reference's): up to SIX products are summed
// before one reduction -- 6 q^2 < 2^64, the sum's high word is < 2.2 q and takes two conditional subtractions (2q, q).  One
// Montgomery reduction per row of a CRT_13 / DFT_13 half-matrix instead of two: 14 instructions per row instead of 21.
// LVL 2 (every modulus below 2^32 / 6 = 715 827 882: all of examples/Tunnel.hs's): six q^2 stay below q 2^32, the sum's high word is
// already below q -- no conditional subtraction in front of the reduction (10 instructions per six-term row instead of 14).  Any level:
// a chunk of one or two products needs none either (2 q^2 < q 2^32 for every q < 2^31).
template <int R, int LVL = 0, typename MP>
__device__ __forceinline__ u32 dense_row(const u32* x, MP M, u32 q, u32 qni) {
    constexpr int CH = LVL ? 6 : 4;
    u32 acc = 0;
#pragma unroll
    for (int t0 = 0; t0 < R; t0 += CH) {
        const int terms = (R - t0 < CH) ? R - t0 : CH;
        u64 p = (u64)x[t0] * M[t0];
#pragma unroll
        for (int t = t0 + 1; t < t0 + CH && t < R; ++t) p += (u64)x[t] * M[t];
        u32 hi = (u32)(p >> 32);
        if (LVL == 1 && terms > 4) hi = csub(hi, 2u * q);                       // more than four terms: high word < 2.2 q
        if (terms > 2 && LVL != 2) hi = csub(hi, q);                             // high word < 2q -> < q
        const u64 pr = ((u64)hi << 32) | (u32)p;
        const u32 m = (u32)pr * qni;
        const u32 v = csub((u32)((pr + (u64)m * q) >> 32), q);
        acc = t0 ? csub(acc + v, q) : v;
    }
    return acc;
}
template <int R, int LVL = 0, typename MP>
__device__ __forceinline__ u64 dense_row(const u64* x, MP M, u64 q, u64 qni) {
    u64 acc = csub(mont_mul_lazy(x[0], M[0], q, qni), q);
#pragma unroll
    for (int t = 1; t < R; ++t) acc = csub(acc + csub(mont_mul_lazy(x[t], M[t], q, qni), q), q);
    return acc;
}
template <typename W> __device__ __forceinline__ W gadd(W a, W b, W q) { return csub((W)(a + b), q); }
// a - b mod q for canonical a, b: the difference wraps to a huge word exactly when a < b, and then adding q wraps back below it
template <typename W> __device__ __forceinline__ W gsub(W a, W b, W q) { const W d = a - b, e = d + q; return e < d ? e : d; }
template <typename W> __device__ __forceinline__ W gmul(W a, W b, W q, W qni) { return csub(mont_mul_lazy(a, b, q, qni), q); }

// CRT_p and DFT_p through the symmetry  w^(i (p-j)) = w^(-i j):  with u_j = x_j + x_{p-j}, v_j = x_j - x_{p-j} (j = 1..h,
// h = (p-1)/2), a_ij = (w^ij + w^-ij)/2, b_ij = (w^ij - w^-ij)/2:
//     y_i = A_i + B_i,  y_{p-i} = A_i - B_i,   A_i = x_0 + sum_j a_ij u_j,   B_i = sum_j b_ij v_j        (i = 1..h)
// i.e. 2 h^2 products for p - 1 outputs instead of (p-1)^2.  Table of a pass (same layout in the forward and inverse block):
//     a[h*h], b[h*h], negw[p-1], pinv      (negw, pinv: inverse CRT_p only / inverse passes only)
//   forward CRT_p : inputs x_0..x_{p-2} (x_{p-1} = 0), outputs y_1..y_{p-1} at slots 0..p-2
//   inverse CRT_p : y_0 = -sum_i y_i w^i (the condition x_{p-1} = 0), then the inverse DFT_p (w -> w^-1, a, b, carry 1/p)
//   forward DFT_p : y_0 = sum_j x_j as well;  inverse DFT_p : the same with w^-1 and 1/p
template <typename W, int NT, int P, bool IS_DFT, bool INV, int SMALLQ = 0>
__device__ __forceinline__ void gen_sym_pass(W* __restrict__ lds, const GenPass& Ps, const W* __restrict__ tab, u32 n, W q, W qni, u32 tid) {
    constexpr int H = (P - 1) / 2, R = IS_DFT ? P : P - 1;
    // CRT_p passes never carry twiddles, DFT_p passes always do (gen_plan): a compile-time fact, so the per-element loads below are
    // straight-line code -- as a run-time flag it cost a branch and ~10 VALU instructions per element loaded
    constexpr bool has_tw = IS_DFT;
    // the pass matrices are wave-uniform: read through the constant address space they become scalar loads and live in SGPRs
    // (as plain global loads hipcc hoisted all 2 h^2 + p of them into VGPRs: 85 registers for p = 13)
    typedef const W __attribute__((address_space(4)))* CP;
    CP T = (CP)(tab + Ps.mat_off);
    CP a = T;
    CP b = T + H * H;
    const W* __restrict__ tw = tab + (has_tw ? Ps.tw_off : 0u);
    const u32 stride = Ps.stride;
    const u32 step = has_tw ? fdiv(stride, Ps.axis_stride, Ps.rcp_axis_stride) : 0u;
    for (u32 w = tid; w < n / (u32)R; w += NT) {
        const u32 hi = fdiv(w, stride, Ps.rcp_stride), lo = w - hi * stride;
        const u32 base = hi * (u32)R * stride + lo;
        u32 pos0 = 0;
        if (has_tw) {
            const u32 ap = fdiv(base, Ps.axis_stride, Ps.rcp_axis_stride);
            pos0 = ap - fdiv(ap, Ps.axis_len, Ps.rcp_axis_len) * Ps.axis_len;
        }
        // the length-p input of the DFT_p behind the pass: in_t = element t - OFF of the group
        //   forward CRT_p: in_0..in_{p-2} = x, in_{p-1} = 0;  inverse CRT_p: in_0 = y_0 (below), in_1.. = x;  DFT_p: in = x
        constexpr int OFF = (!IS_DFT && INV) ? 1 : 0;
        W x[R];
        typedef typename Vec4<W>::type V;
        constexpr int VL = Vec4<W>::LANES;
        // innermost axis (stride 1, every index of the reference ends in p = 13: R = 12): the group is R contiguous words, moved as
        // 16-byte pieces -- the 4-byte form is 4-way bank-conflicted there (lane stride R words)
        const bool contiguous = R % VL == 0 && stride == 1;
        if (contiguous) {
#pragma unroll
            for (int t = 0; t < R; t += VL) {
                const V v = *reinterpret_cast<const V*>(lds + base + t);
#pragma unroll
                for (int c = 0; c < VL; ++c) x[t + c] = v[c];
            }
        } else {   // element t at base + t stride: one running address, one add per element
            W* pe = lds + base;
#pragma unroll
            for (int t = 0; t < R; ++t) { x[t] = *pe; pe += stride; }
        }
        if (!INV && has_tw) {
            const W* pt = tw + pos0;
#pragma unroll
            for (int t = 0; t < R; ++t) { x[t] = gmul(x[t], *pt, q, qni); pt += step; }
        }
        W in0;
        if (!IS_DFT && INV) in0 = dense_row<P - 1, SMALLQ>(x, T + 2 * H * H, q, qni);          // y_0 = sum_i y_i (-w^i)
        else in0 = x[0];
        auto in = [&](int t) -> W { return t == 0 ? in0 : (t - OFF < R ? x[t - OFF] : (W)0); };
        W u[H], v[H];
        W sum = in0;
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            const W xa = in(j), xb = in(P - j);
            u[j - 1] = gadd(xa, xb, q);
            v[j - 1] = gsub(xa, xb, q);
            sum = gadd(sum, u[j - 1], q);
        }
        W u0 = in0;
        if (INV) {                                 // 1/p rides on a, b; the constant terms take it explicitly
            const W pinv = T[2 * H * H + (P - 1)];
            u0 = gmul(u0, pinv, q, qni);
            sum = gmul(sum, pinv, q, qni);
        }
        // results: row r of the DFT_p lands at element r (DFT_p, inverse CRT_p) or r - 1 (forward CRT_p, which has no row 0);
        // inverse CRT_p drops row p - 1 (= 0).  Collected in registers in element order, then stored along one running address.
        W y[R];
        auto put = [&](int row, W val) {
            const int t = (!IS_DFT && !INV) ? row - 1 : row;
            if (t >= 0 && t < R) y[t] = val;
        };
        put(0, sum);
#pragma unroll
        for (int i = 1; i <= H; ++i) {
            const W A = gadd(u0, dense_row<H, SMALLQ>(u, a + (i - 1) * H, q, qni), q);
            const W B = dense_row<H, SMALLQ>(v, b + (i - 1) * H, q, qni);
            put(i, gadd(A, B, q));
            put(P - i, gsub(A, B, q));
        }
        if (INV && has_tw) {
            const W* pt = tw + pos0;
#pragma unroll
            for (int t = 0; t < R; ++t) { y[t] = gmul(y[t], *pt, q, qni); pt += step; }
        }
        if (contiguous) {
#pragma unroll
            for (int t = 0; t < R; t += VL) {
                V v;
#pragma unroll
                for (int c = 0; c < VL; ++c) v[c] = y[t + c];
                *reinterpret_cast<V*>(lds + base + t) = v;
            }
        } else {
            W* pe = lds + base;
#pragma unroll
            for (int t = 0; t < R; ++t) { *pe = y[t]; pe += stride; }
        }
    }
}

// K merged-twiddle Cooley-Tukey stages (s0 .. s0+K-1) of a two-power axis on register-resident groups of 2^K points
// (element k of a group at base + k stride).  Stage s0 + r uses the 2^r table entries tw[(gm << r) + c], gm = 2^s0 + group
// index -- the table is tw[k] = psi^brev(k), exactly the two-power engine's.  Inverse: Gentleman-Sande, the stages backwards with
// tw^-1; the factor 2^-K is collected in GenDev::iscale_m.
template <typename W, int NT, int K, bool INV>
__device__ __forceinline__ void gen_r2block_pass(W* __restrict__ lds, const GenPass& P, const W* __restrict__ tab, u32 n, W q, W qni, u32 tid) {
    constexpr int R = 1 << K;
    const W* __restrict__ tw = tab + P.tw_off;
    const u32 smask = (1u << P.aux) - 1u;
    for (u32 w = tid; w < n / (u32)R; w += NT) {
        const u32 hi = fdiv(w, P.stride, P.rcp_stride), lo = w - hi * P.stride;
        const u32 base = hi * (u32)R * P.stride + lo;
        const u32 gm = (1u << P.aux) + (hi & smask);
        W x[R];
#pragma unroll
        for (int k = 0; k < R; ++k) x[k] = lds[base + (u32)k * P.stride];
        if (!INV) {
#pragma unroll
            for (int r = 0; r < K; ++r) {
                const int half = R >> (r + 1);
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    if (k & half) continue;
                    const W t = gmul(x[k + half], tw[(gm << r) + (u32)(k >> (K - r))], q, qni);
                    const W a = x[k];
                    x[k] = gadd(a, t, q);
                    x[k + half] = gsub(a, t, q);
                }
            }
        } else {
#pragma unroll
            for (int r = K - 1; r >= 0; --r) {
                const int half = R >> (r + 1);
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    if (k & half) continue;
                    const W a = x[k], b = x[k + half];
                    x[k] = gadd(a, b, q);
                    x[k + half] = gmul(gsub(a, b, q), tw[(gm << r) + (u32)(k >> (K - r))], q, qni);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) lds[base + (u32)k * P.stride] = x[k];
    }
}

template <typename W, int NT, bool INV>
__device__ __forceinline__ void gen_run_pass(W* lds, const GenPass& P, const W* tab, u32 n, W q, W qni, int smallq, u32 tid) {
    if (P.kind == GK_R2BLOCK) {                    // every branch here is wave-uniform
        if (P.r == 8) gen_r2block_pass<W, NT, 3, INV>(lds, P, tab, n, q, qni, tid);
        else if (P.r == 4) gen_r2block_pass<W, NT, 2, INV>(lds, P, tab, n, q, qni, tid);
        else gen_r2block_pass<W, NT, 1, INV>(lds, P, tab, n, q, qni, tid);
        return;
    }
    const int p = P.kind == GK_SYM_DFT ? P.r : P.r + 1;
    if (P.kind == GK_SYM_CRT) {
        switch (p) {
        case 3: gen_sym_pass<W, NT, 3, false, INV>(lds, P, tab, n, q, qni, tid); break;
        case 5: gen_sym_pass<W, NT, 5, false, INV>(lds, P, tab, n, q, qni, tid); break;
        case 7:
            if (sizeof(W) == 4 && smallq == 2) gen_sym_pass<W, NT, 7, false, INV, 2>(lds, P, tab, n, q, qni, tid);
            else gen_sym_pass<W, NT, 7, false, INV>(lds, P, tab, n, q, qni, tid);
            break;
        case 11: gen_sym_pass<W, NT, 11, false, INV>(lds, P, tab, n, q, qni, tid); break;
        case 13:                                   // (wave-uniform: the ring's moduli decide)
            if (sizeof(W) == 4 && smallq == 2) gen_sym_pass<W, NT, 13, false, INV, 2>(lds, P, tab, n, q, qni, tid);
            else if (sizeof(W) == 4 && smallq) gen_sym_pass<W, NT, 13, false, INV, 1>(lds, P, tab, n, q, qni, tid);
            else gen_sym_pass<W, NT, 13, false, INV>(lds, P, tab, n, q, qni, tid);
            break;
        default: break;                            // the host refuses indices with other odd primes
        }
    } else {
        switch (p) {
        case 3: gen_sym_pass<W, NT, 3, true, INV>(lds, P, tab, n, q, qni, tid); break;
        case 5: gen_sym_pass<W, NT, 5, true, INV>(lds, P, tab, n, q, qni, tid); break;
        case 7:
            if (sizeof(W) == 4 && smallq == 2) gen_sym_pass<W, NT, 7, true, INV, 2>(lds, P, tab, n, q, qni, tid);
            else gen_sym_pass<W, NT, 7, true, INV>(lds, P, tab, n, q, qni, tid);
            break;
        case 11: gen_sym_pass<W, NT, 11, true, INV>(lds, P, tab, n, q, qni, tid); break;
        case 13:
            if (sizeof(W) == 4 && smallq == 2) gen_sym_pass<W, NT, 13, true, INV, 2>(lds, P, tab, n, q, qni, tid);
            else if (sizeof(W) == 4 && smallq) gen_sym_pass<W, NT, 13, true, INV, 1>(lds, P, tab, n, q, qni, tid);
            else gen_sym_pass<W, NT, 13, true, INV>(lds, P, tab, n, q, qni, tid);
            break;
        default: break;
        }
    }
}

// whole transform on an LDS-resident polynomial (canonical values in, canonical values out)
// NT threads with ids tid = 0 .. NT-1 work on the polynomial (the whole workgroup by default; k_gen_tunnel_ks runs several
// polynomials side by side, one per sub-group of threads: every thread of the workgroup reaches the same barriers)
template <typename W, bool INV, int NT = GEN_T>
__device__ __forceinline__ void gen_transform(W* lds, const GenDev<W>& G, int j, W q, W qni, u32 tid) {
    if (!INV) {
        for (int ps = 0; ps < G.npass; ++ps) { gen_run_pass<W, NT, false>(lds, G.pass[ps], G.tabf[j], G.n, q, qni, G.smallq, tid); lds_barrier(); }
    } else {
        for (int ps = G.npass - 1; ps >= 0; --ps) { gen_run_pass<W, NT, true>(lds, G.pass[ps], G.tabi[j], G.n, q, qni, G.smallq, tid); lds_barrier(); }
    }
}
template <typename W, bool INV, int NT = GEN_T>
__device__ __forceinline__ void gen_transform(W* lds, const GenDev<W>& G, int j, W q, W qni) {
    gen_transform<W, INV, NT>(lds, G, j, q, qni, threadIdx.x);
}

// ------------------------------------------------------------------------------------------------------
// batched crt / crtInv: one workgroup per limb-polynomial
// ------------------------------------------------------------------------------------------------------
// (second launch bound: four waves per SIMD = at most 128 VGPRs, so that rings of up to 36 KiB run four workgroups per CU)
template <typename W, bool INV, int NT = GEN_T>
__global__ void __launch_bounds__(NT, 4) k_gen_crt(DevRing<W> R, GenDev<W> G, W* data, const W* src, size_t first_poly) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const size_t p = first_poly + blockIdx.x;
    const int j = (int)(p % (size_t)R.L);
    const u32 n = G.n;
    W* poly = data + p * (size_t)n;
    const W* in = src ? src + p * (size_t)n : poly;
    const W q = R.mod[j].q, qni = R.mod[j].qni;
    typedef typename Vec4<W>::type V;
    constexpr u32 VL = Vec4<W>::LANES;
    const bool vec = n % VL == 0;                 // then every limb-polynomial starts 16-byte aligned
    if (vec) for (u32 i = threadIdx.x * VL; i < n; i += NT * VL) *reinterpret_cast<V*>(lds + i) = *reinterpret_cast<const V*>(in + i);
    else for (u32 i = threadIdx.x; i < n; i += NT) lds[i] = in[i];
    lds_barrier();
    gen_transform<W, INV, NT>(lds, G, j, q, qni);
    const W sc = G.iscale_m[j];
    if (vec) {
        for (u32 i = threadIdx.x * VL; i < n; i += NT * VL) {
            V v = *reinterpret_cast<const V*>(lds + i);
            if (INV) {
#pragma unroll
                for (u32 e = 0; e < VL; ++e) v[e] = csub(mont_mul_lazy(v[e], sc, q, qni), q);
            }
            *reinterpret_cast<V*>(poly + i) = v;
        }
    } else if (INV) {
        for (u32 i = threadIdx.x; i < n; i += NT) poly[i] = csub(mont_mul_lazy(lds[i], sc, q, qni), q);
    } else {
        for (u32 i = threadIdx.x; i < n; i += NT) poly[i] = lds[i];
    }
}

// crt of the reduced TrivGad digits with decompose + reduce in the loader (keySwitchQuadCirc, Eval.hs:133):
// workgroup = (ciphertext, source limb i, target limb j); the diagonal i == j is skipped (that digit is c2's own
// limb j, which the caller kept in the CRT basis).
template <typename W, int NT = GEN_T>
__global__ void __launch_bounds__(NT, 4) k_gen_crt_digits(DevRing<W> R, GenDev<W> G, const W* __restrict__ c2pow, W* __restrict__ digits, int balanced, int with_diag,
                 int Ls, int sfirst) {
    // Ls, sfirst: the source elements hold the limbs sfirst .. sfirst + Ls - 1 only (tunnel behind a modSwitch up: the added
    // limbs are zero and so are their digits); digits: [element][Ls][L][n].  Key switch: Ls = L, sfirst = 0.
    typedef typename Signed<W>::type SW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L;
    const size_t p = blockIdx.x;
    const int j = (int)(p % (size_t)L), is = (int)((p / (size_t)L) % (size_t)Ls), i = is + sfirst;
    if (i == j && !with_diag) return;      // key switch: that digit is c2's own limb j, kept in the CRT basis by the caller
    const size_t ct = p / ((size_t)L * Ls);
    const u32 n = G.n;
    const W* src = c2pow + (ct * (size_t)Ls + is) * (size_t)n;
    W* dst = digits + p * (size_t)n;
    const W q = R.mod[j].q, qni = R.mod[j].qni, qi = R.mod[i].q, hqi = (qi - 1) >> 1;
    typedef typename Vec4<W>::type V;
    constexpr u32 VL = Vec4<W>::LANES;
    auto digit = [&](W v) -> W {
        const SW z = v > hqi ? (SW)v - (SW)qi : (SW)v;
        SW r;
        if (balanced) r = z < 0 ? z + (SW)q : z;
        else { r = z % (SW)q; if (r < 0) r += (SW)q; }
        return (W)r;
    };
    if (n % VL == 0) {
        for (u32 k = threadIdx.x * VL; k < n; k += NT * VL) {
            V v = *reinterpret_cast<const V*>(src + k);
#pragma unroll
            for (u32 c = 0; c < VL; ++c) v[c] = digit(v[c]);
            *reinterpret_cast<V*>(lds + k) = v;
        }
    } else {
        for (u32 k = threadIdx.x; k < n; k += NT) lds[k] = digit(src[k]);
    }
    lds_barrier();
    gen_transform<W, false, NT>(lds, G, j, q, qni);
    if (n % VL == 0) for (u32 k = threadIdx.x * VL; k < n; k += NT * VL) *reinterpret_cast<V*>(dst + k) = *reinterpret_cast<const V*>(lds + k);
    else for (u32 k = threadIdx.x; k < n; k += NT) dst[k] = lds[k];
}

// crt of the reduced BaseBGad 2 digits with decompose + reduce in the loader (tunnels with the gadget of examples/Tunnel.hs:24):
// workgroup p = (element, digit d, target limb j); digit d is bit t of source limb i.  With u = -(centred lift) the balanced
// binary digits have the closed form  d_t = -((u >> t) & 1)  for t < k - 1 and the top digit is  -(u >> (k - 1))  (arithmetic
// shifts), so no digit depends on the ones below it and the digits never exist in HBM untransformed (k_crt_base2_digits is the
// two-power form).  digits: [element][D][L][n].
template <typename W, int NT = GEN_T>
__global__ void __launch_bounds__(NT, 4) k_gen_crt_base2_digits(DevRing<W> R, GenDev<W> G, const W* __restrict__ xpow, W* __restrict__ digits,
                                                                Scal<u32> first_digit, Scal<u32> kd, u32 D) {
    typedef typename Signed<W>::type SW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L;
    const size_t p = blockIdx.x;
    const int j = (int)(p % (size_t)L);
    const u32 d = (u32)((p / (size_t)L) % D);
    const size_t el = p / ((size_t)L * D);
    int i = 0;
    for (int c = 1; c < L; ++c) if (d >= first_digit.v[c]) i = c;
    const u32 t = d - first_digit.v[i];
    const bool top = t + 1 == kd.v[i];
    const u32 n = G.n;
    const W* src = xpow + (el * (size_t)L + i) * (size_t)n;
    W* dst = digits + p * (size_t)n;
    const W q = R.mod[j].q, qni = R.mod[j].qni, qi = R.mod[i].q, hqi = (qi - 1) >> 1;
    for (u32 k = threadIdx.x; k < n; k += NT) {
        const W v = src[k];
        const SW z = v > hqi ? (SW)v - (SW)qi : (SW)v;
        const SW u = -z;
        SW dg = top ? -(u >> t) : -((u >> t) & 1);
        if (top) dg %= (SW)q;                               // the top digit is tiny but its size depends on q_i vs 2^k
        lds[k] = dg < 0 ? (W)(dg + (SW)q) : (W)dg;
    }
    lds_barrier();
    gen_transform<W, false, NT>(lds, G, j, q, qni);
    typedef typename Vec4<W>::type V;
    constexpr u32 VL = Vec4<W>::LANES;
    if (n % VL == 0) for (u32 k = threadIdx.x * VL; k < n; k += NT * VL) *reinterpret_cast<V*>(dst + k) = *reinterpret_cast<const V*>(lds + k);
    else for (u32 k = threadIdx.x; k < n; k += NT) dst[k] = lds[k];
}

// ------------------------------------------------------------------------------------------------------
// fused key switch for a general index (SymmSHE (*) + keySwitchQuadCirc, Eval.hs:65-67,133), two launches per chunk:
//   k_gen_tensor_inv  per (ciphertext, operand limb i): c2_i = a1 b1 g s on the CRT basis (mulG = product with the CRT image
//                     of g), crtInv in LDS, canonical Pow-basis residues to the digit scratch
//   k_gen_ks          per (ciphertext, limb j of the hint's ring): c0 = a0 b0 g s, c1 = (a0 b1 + a1 b0) g s and the diagonal digit
//                     (c2_j itself) from the operands; for every other digit i: centred lift + reduce in the loader, crt in LDS,
//                     acc += crt(d_i) hint_i; the 2 n / T accumulators stay in registers
// Against the composed path (element-wise tensor, batched crtInv, digit transforms, hint inner product) this removes two
// HBM-bound element-wise kernels and the digit round trip through HBM (L (L-1) limb-polynomials written and read per op).
// `dup` > 0: the operands live dup limbs below the hint's ring (PT2CT's mul_, see k_ks_accum_half).
// ------------------------------------------------------------------------------------------------------
// experiment switch: non-temporal cache policy for k_gen_ks's operand loads and result stores (they stream through once)
#ifndef ALCH_GEN_NT
#define ALCH_GEN_NT 0
#endif
constexpr int GEN_KS_T = 512;
constexpr int GEN_KS_NPT = 24;              // slots per lane: n <= 12288 (every index of the reference)

template <typename W>
struct GenKsArgs {
    const W* a;              // operands [ct][2][Ls][n], CRT basis
    const W* b;
    W* c2pow;                // [ct][Ls][n], Pow basis
    const W* hint;           // [L][2][L][n], Montgomery form
    W* out;                  // [ct][2][L][n]
    Scal<W> sr2;             // s_i R^2 per operand limb
    int dup;
    int balanced;
    int use_g;               // index has odd prime factors: mulG is not the identity
};

// VEC: the ring dimension is a multiple of the 16-byte vector width (every index of the reference): all global and LDS traffic of
// both kernels moves 16-byte pieces, a lane owning VL consecutive slots; else one word at a time.
template <typename W, bool VEC>
__global__ void __launch_bounds__(GEN_KS_T, 4) k_gen_tensor_inv(DevRing<W> R, GenDev<W> G, GenKsArgs<W> A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    typedef typename Vec4<W>::type V;
    constexpr u32 VL = VEC ? Vec4<W>::LANES : 1;
    const int L = R.L, Ls = L - A.dup;
    const size_t ct = blockIdx.x / (unsigned)Ls;
    const int is = (int)(blockIdx.x % (unsigned)Ls), i = is + A.dup;
    const u32 n = G.n;
    const W q = R.mod[i].q, qni = R.mod[i].qni;
    const W* a1 = A.a + ((2 * ct + 1) * (size_t)Ls + is) * n;
    const W* b1 = A.b + ((2 * ct + 1) * (size_t)Ls + is) * n;
    const W* g = G.gcrt[i];
    const W sr2 = A.sr2.v[is];
    for (u32 k = threadIdx.x * VL; k < n; k += GEN_KS_T * VL) {
        if constexpr (VEC) {
            const V av = *reinterpret_cast<const V*>(a1 + k), bv = *reinterpret_cast<const V*>(b1 + k);
            V gv = av, o;
            if (A.use_g) gv = *reinterpret_cast<const V*>(g + k);
#pragma unroll
            for (u32 e = 0; e < VL; ++e) {
                W v = gmul(gmul(av[e], sr2, q, qni), bv[e], q, qni);
                if (A.use_g) v = gmul(v, gv[e], q, qni);
                o[e] = v;
            }
            *reinterpret_cast<V*>(lds + k) = o;
        } else {
            W v = gmul(gmul(a1[k], sr2, q, qni), b1[k], q, qni);
            if (A.use_g) v = gmul(v, g[k], q, qni);
            lds[k] = v;
        }
    }
    lds_barrier();
    gen_transform<W, true, GEN_KS_T>(lds, G, i, q, qni);
    const W sc = G.iscale_m[i];
    W* dst = A.c2pow + (ct * (size_t)Ls + is) * n;
    for (u32 k = threadIdx.x * VL; k < n; k += GEN_KS_T * VL) {
        if constexpr (VEC) {
            V v = *reinterpret_cast<const V*>(lds + k);
#pragma unroll
            for (u32 e = 0; e < VL; ++e) v[e] = gmul(v[e], sc, q, qni);
            *reinterpret_cast<V*>(dst + k) = v;
        } else {
            dst[k] = gmul(lds[k], sc, q, qni);
        }
    }
}

template <typename W, bool VEC>
__global__ void __launch_bounds__(GEN_KS_T, 2) k_gen_ks(DevRing<W> R, GenDev<W> G, GenKsArgs<W> A) {
    typedef typename Signed<W>::type SW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    typedef typename Vec4<W>::type V;
    constexpr int VL = VEC ? Vec4<W>::LANES : 1;
    constexpr int NP = GEN_KS_NPT / VL;                       // pieces per lane; slot of (piece kk, element e) = (tid + kk T) VL + e
    const int L = R.L, Ls = L - A.dup;
    const size_t ct = blockIdx.x / (unsigned)L;
    const int j = (int)(blockIdx.x % (unsigned)L), js = j - A.dup;
    const u32 n = G.n;
    const W q = R.mod[j].q, qni = R.mod[j].qni;
    const size_t Ln = (size_t)L * n;
    const W* hj = A.hint + (size_t)j * n;                    // + (2 i + c) * Ln
    auto ld = [&](const W* p, u32 s, W (&o)[VL]) {            // VL consecutive words at p + s
        if constexpr (VEC) {
            const V v = *reinterpret_cast<const V*>(p + s);
#pragma unroll
            for (int e = 0; e < VL; ++e) o[e] = v[e];
        } else o[0] = p[s];
    };
    auto ld_once = [&](const W* p, u32 s, W (&o)[VL]) {       // the same for data this kernel reads once (the operands)
        if constexpr (VEC && ALCH_GEN_NT) {
            const V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(p + s));
#pragma unroll
            for (int e = 0; e < VL; ++e) o[e] = v[e];
        } else ld(p, s, o);
    };
    W acc0[NP][VL], acc1[NP][VL];
#pragma unroll
    for (int kk = 0; kk < NP; ++kk)
#pragma unroll
        for (int e = 0; e < VL; ++e) { acc0[kk][e] = 0; acc1[kk][e] = 0; }
    if (js >= 0) {
        const size_t o = (size_t)js * n;
        const W* a0 = A.a + (2 * ct) * (size_t)Ls * n + o;
        const W* a1 = A.a + (2 * ct + 1) * (size_t)Ls * n + o;
        const W* b0 = A.b + (2 * ct) * (size_t)Ls * n + o;
        const W* b1 = A.b + (2 * ct + 1) * (size_t)Ls * n + o;
        const W* h0 = hj + (size_t)(2 * j) * Ln;
        const W* h1 = hj + (size_t)(2 * j + 1) * Ln;
        const W* g = G.gcrt[j];
        const W sr2 = A.sr2.v[js];
#pragma unroll
        for (int kk = 0; kk < NP; ++kk) {
            const u32 s = (threadIdx.x + (u32)kk * GEN_KS_T) * VL;
            if (s < n) {
                W va0[VL], va1[VL], vb0[VL], vb1[VL], vg[VL], vh0[VL], vh1[VL];
                ld_once(a0, s, va0); ld_once(a1, s, va1); ld_once(b0, s, vb0); ld_once(b1, s, vb1); ld(h0, s, vh0); ld(h1, s, vh1);
                if (A.use_g) ld(g, s, vg);
#pragma unroll
                for (int e = 0; e < VL; ++e) {
                    const W x0 = gmul(va0[e], sr2, q, qni), x1 = gmul(va1[e], sr2, q, qni);
                    W c0 = gmul(x0, vb0[e], q, qni);
                    W c1 = gadd(gmul(x0, vb1[e], q, qni), gmul(x1, vb0[e], q, qni), q);
                    W c2 = gmul(x1, vb1[e], q, qni);
                    if (A.use_g) { const W gv = vg[e]; c0 = gmul(c0, gv, q, qni); c1 = gmul(c1, gv, q, qni); c2 = gmul(c2, gv, q, qni); }
                    acc0[kk][e] = gadd(c0, gmul(c2, vh0[e], q, qni), q);
                    acc1[kk][e] = gadd(c1, gmul(c2, vh1[e], q, qni), q);
                }
            }
        }
    }
    for (int is = 0; is < Ls; ++is) {
        if (is == js) continue;
        const int i = is + A.dup;
        const W qi = R.mod[i].q, hqi = (qi - 1) >> 1;
        const W* src = A.c2pow + (ct * (size_t)Ls + is) * n;
        lds_barrier();                                      // the previous digit's products have been read
        for (u32 k = threadIdx.x * VL; k < n; k += GEN_KS_T * VL) {
            W v[VL];
            ld(src, k, v);
#pragma unroll
            for (int e = 0; e < VL; ++e) {
                const SW z = v[e] > hqi ? (SW)v[e] - (SW)qi : (SW)v[e];
                SW r;
                if (A.balanced) r = z < 0 ? z + (SW)q : z;
                else { r = z % (SW)q; if (r < 0) r += (SW)q; }
                v[e] = (W)r;
            }
            if constexpr (VEC) {
                V o;
#pragma unroll
                for (int e = 0; e < VL; ++e) o[e] = v[e];
                *reinterpret_cast<V*>(lds + k) = o;
            } else lds[k] = v[0];
        }
        lds_barrier();
        gen_transform<W, false, GEN_KS_T>(lds, G, j, q, qni);
        const W* h0 = hj + (size_t)(2 * i) * Ln;
        const W* h1 = hj + (size_t)(2 * i + 1) * Ln;
#pragma unroll
        for (int kk = 0; kk < NP; ++kk) {
            const u32 s = (threadIdx.x + (u32)kk * GEN_KS_T) * VL;
            if (s < n) {
                W x[VL], vh0[VL], vh1[VL];
                ld(lds, s, x); ld(h0, s, vh0); ld(h1, s, vh1);
#pragma unroll
                for (int e = 0; e < VL; ++e) {
                    acc0[kk][e] = gadd(acc0[kk][e], gmul(x[e], vh0[e], q, qni), q);
                    acc1[kk][e] = gadd(acc1[kk][e], gmul(x[e], vh1[e], q, qni), q);
                }
            }
        }
    }
    W* o0 = A.out + ((2 * ct) * (size_t)L + j) * n;
    W* o1 = A.out + ((2 * ct + 1) * (size_t)L + j) * n;
#pragma unroll
    for (int kk = 0; kk < NP; ++kk) {
        const u32 s = (threadIdx.x + (u32)kk * GEN_KS_T) * VL;
        if (s < n) {
            if constexpr (VEC) {
                V v0, v1;
#pragma unroll
                for (int e = 0; e < VL; ++e) { v0[e] = acc0[kk][e]; v1[e] = acc1[kk][e]; }
                if constexpr (ALCH_GEN_NT) {
                    __builtin_nontemporal_store(v0, reinterpret_cast<V*>(o0 + s));
                    __builtin_nontemporal_store(v1, reinterpret_cast<V*>(o1 + s));
                } else {
                    *reinterpret_cast<V*>(o0 + s) = v0;
                    *reinterpret_cast<V*>(o1 + s) = v1;
                }
            } else { o0[s] = acc0[kk][0]; o1[s] = acc1[kk][0]; }
        }
    }
}

template <typename W, bool VEC>
inline hipError_t gen_launch_ks_v(const DevRing<W>& R, const GenDev<W>& G, const GenKsArgs<W>& A, size_t nct, hipStream_t stream) {
    const size_t lds_bytes = (size_t)G.n * sizeof(W);
    auto k1 = k_gen_tensor_inv<W, VEC>;
    auto k2 = k_gen_ks<W, VEC>;
    hipError_t e;
    if ((e = set_lds(k1, lds_bytes)) != hipSuccess) return e;
    if ((e = set_lds(k2, lds_bytes)) != hipSuccess) return e;
    hipLaunchKernelGGL(k1, dim3((unsigned)(nct * (size_t)(R.L - A.dup))), dim3(GEN_KS_T), lds_bytes, stream, R, G, A);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(k2, dim3((unsigned)(nct * (size_t)R.L)), dim3(GEN_KS_T), lds_bytes, stream, R, G, A);
    return hipGetLastError();
}

template <typename W>
inline hipError_t gen_launch_ks(const DevRing<W>& R, const GenDev<W>& G, const GenKsArgs<W>& A, size_t nct, hipStream_t stream) {
    return G.n % Vec4<W>::LANES == 0 ? gen_launch_ks_v<W, true>(R, G, A, nct, stream) : gen_launch_ks_v<W, false>(R, G, A, nct, stream);
}

// ------------------------------------------------------------------------------------------------------
// Fused linear key switch of a ring tunnel (SymmSHE tunnel, Eval.hs:134; TrivGad): digit transforms + hint inner product in one
// kernel, the counterpart of k_gen_ks for tunnels.  Workgroup = (ciphertext, limb j of S'_q).  The digits are E'-coefficients of c1
// (dimension phi(e'), 2-5 times smaller than phi(s')): NG of them are transformed side by side, one per sub-group of T / NG
// threads, in NG phi(e') words of LDS; the CRT over S' of an embedded E'-element is its CRT over E' replicated (slot_e), so the
// inner product reads the transformed digit through that table, 16 bytes at a time, against the hint rows of the lane's own
// output slots; the 2 phi(s') / T accumulators per lane stay in registers.  No digit is written to or read from HBM
// (d_rel (L - dup) L limb-vectors of phi(e') words per ciphertext before).
// ------------------------------------------------------------------------------------------------------
constexpr int GEN_TUN_T = 512;

template <typename W>
struct GenTunArgs {
    const W* x1;             // [ct][D][Lx][n_e]: Pow-basis E'-coefficients of c1, limbs xoff .. xoff + Lx - 1 of the ring
    const W* hint;           // [i * L + limb][2][L][n_s], Montgomery form
    W* out;                  // [ct][2][L][n_s]: the c0 rows hold f'(c0) on entry; both rows are (over)written
    const u32* slot_e;       // [n_s]: CRT slot of E' behind every CRT slot of S'; consecutive and aligned within every 16-byte piece
    u32 D, Lx, xoff, n_s;
    int balanced;
};

template <typename W, int NG, int NP>                       // NP: 16-byte output pieces per lane, phi(s') <= NP * 4 * GEN_TUN_T words
__global__ void __launch_bounds__(GEN_TUN_T, 4) k_gen_tunnel_ks(DevRing<W> R, GenDev<W> GE, GenTunArgs<W> A) {
    typedef typename Signed<W>::type SW;
    typedef typename Vec4<W>::type V;
    constexpr int VL = Vec4<W>::LANES, TS = GEN_TUN_T / NG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L;
    const size_t ct = blockIdx.x / (unsigned)L;
    const int j = (int)(blockIdx.x % (unsigned)L);
    const u32 n_e = GE.n, n_s = A.n_s;
    const W q = R.mod[j].q, qni = R.mod[j].qni;
    const size_t Ln = (size_t)L * n_s;
    const u32 g = threadIdx.x / TS, lt = threadIdx.x - g * TS;
    W* mine = lds + (size_t)g * n_e;
    W* o0 = A.out + ((2 * ct) * (size_t)L + j) * n_s;
    W* o1 = A.out + ((2 * ct + 1) * (size_t)L + j) * n_s;
    V acc0[NP], acc1[NP];
    constexpr bool CACHE_SE = NP < 6;                      // six pieces per lane (phi(s') = 11520) leave no room for the slot indices
    u32 se[CACHE_SE ? NP : 1];
#pragma unroll
    for (int kk = 0; kk < NP; ++kk) {
        const u32 s = (threadIdx.x + (u32)kk * GEN_TUN_T) * VL;
        if (CACHE_SE) se[kk] = 0;
#pragma unroll
        for (int e = 0; e < VL; ++e) { acc0[kk][e] = 0; acc1[kk][e] = 0; }
        if (s < n_s) { acc0[kk] = *reinterpret_cast<const V*>(o0 + s); if (CACHE_SE) se[kk] = A.slot_e[s]; }
    }
    const u32 ndig = A.D * A.Lx;
    for (u32 d0 = 0; d0 < ndig; d0 += NG) {
        lds_barrier();                                      // the previous round's products have been read
        const u32 dd = d0 + g;
        if (dd < ndig) {                                    // decompose + reduce in the loader (TrivGad: centred lift of limb t)
            const u32 t = dd % A.Lx;
            const W qi = R.mod[t + A.xoff].q, hqi = (qi - 1) >> 1;
            const W* src = A.x1 + (ct * (size_t)ndig + dd) * (size_t)n_e;
            for (u32 k = lt * VL; k < n_e; k += TS * VL) {
                V v = *reinterpret_cast<const V*>(src + k);
#pragma unroll
                for (int e = 0; e < VL; ++e) {
                    const SW z = v[e] > hqi ? (SW)v[e] - (SW)qi : (SW)v[e];
                    SW r;
                    if (A.balanced) r = z < 0 ? z + (SW)q : z;
                    else { r = z % (SW)q; if (r < 0) r += (SW)q; }
                    v[e] = (W)r;
                }
                *reinterpret_cast<V*>(mine + k) = v;
            }
        } else {
            for (u32 k = lt * VL; k < n_e; k += TS * VL) {
                V z;
#pragma unroll
                for (int e = 0; e < VL; ++e) z[e] = 0;
                *reinterpret_cast<V*>(mine + k) = z;
            }
        }
        lds_barrier();
        gen_transform<W, false, TS>(mine, GE, j, q, qni, lt);       // every sub-group its own digit; ends with a barrier
        for (u32 gg = 0; gg < (u32)NG && d0 + gg < ndig; ++gg) {
            const u32 d = d0 + gg, i = d / A.Lx, t = d - i * A.Lx;
            const W* h0 = A.hint + (size_t)(2 * (i * (u32)L + t + A.xoff)) * Ln + (size_t)j * n_s;
            const W* h1 = h0 + Ln;
            const W* x = lds + (size_t)gg * n_e;
#pragma unroll
            for (int kk = 0; kk < NP; ++kk) {
                const u32 s = (threadIdx.x + (u32)kk * GEN_TUN_T) * VL;
                if (s < n_s) {
                    const V xv = *reinterpret_cast<const V*>(x + (CACHE_SE ? se[kk] : A.slot_e[s]));
                    const V v0 = *reinterpret_cast<const V*>(h0 + s), v1 = *reinterpret_cast<const V*>(h1 + s);
#pragma unroll
                    for (int e = 0; e < VL; ++e) {
                        acc0[kk][e] = gadd(acc0[kk][e], gmul(xv[e], v0[e], q, qni), q);
                        acc1[kk][e] = gadd(acc1[kk][e], gmul(xv[e], v1[e], q, qni), q);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int kk = 0; kk < NP; ++kk) {
        const u32 s = (threadIdx.x + (u32)kk * GEN_TUN_T) * VL;
        if (s < n_s) { *reinterpret_cast<V*>(o0 + s) = acc0[kk]; *reinterpret_cast<V*>(o1 + s) = acc1[kk]; }
    }
}

template <typename W, int NG, int NP>
inline hipError_t gen_launch_tunnel_ks_t(const DevRing<W>& R, const GenDev<W>& GE, const GenTunArgs<W>& A, size_t nct, hipStream_t stream) {
    const size_t lds_bytes = (size_t)NG * GE.n * sizeof(W);
    auto k = k_gen_tunnel_ks<W, NG, NP>;
    hipError_t e = set_lds(k, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)(nct * (size_t)R.L)), dim3(GEN_TUN_T), lds_bytes, stream, R, GE, A);
    return hipGetLastError();
}

template <typename W>
inline hipError_t gen_launch_tunnel_ks(const DevRing<W>& R, const GenDev<W>& GE, const GenTunArgs<W>& A, size_t nct, int ng, hipStream_t stream) {
    const u32 pieces = (A.n_s / Vec4<W>::LANES + GEN_TUN_T - 1) / GEN_TUN_T;
    if (pieces <= 3) return ng == 4 ? gen_launch_tunnel_ks_t<W, 4, 3>(R, GE, A, nct, stream) : gen_launch_tunnel_ks_t<W, 2, 3>(R, GE, A, nct, stream);
    if (pieces <= 5) return ng == 4 ? gen_launch_tunnel_ks_t<W, 4, 5>(R, GE, A, nct, stream) : gen_launch_tunnel_ks_t<W, 2, 5>(R, GE, A, nct, stream);
    return ng == 4 ? gen_launch_tunnel_ks_t<W, 4, 6>(R, GE, A, nct, stream) : gen_launch_tunnel_ks_t<W, 2, 6>(R, GE, A, nct, stream);
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

// the recurrence OP along every odd-prime axis of an LDS-resident limb-polynomial (ends with a barrier)
template <typename W, bool ZDOM, int OP, int NT = GEN_T>
__device__ __forceinline__ void gen_columns_lds(W* lds, const GenDev<W>& G, const ColArith<W, ZDOM>& A, u32 skip_mask) {
    const u32 n = G.n;
    typedef typename Vec4<W>::type V;
    constexpr int VL = Vec4<W>::LANES;
    for (int l = 0; l < G.nfact; ++l) {
        const GenFact f = G.fact[l];
        if (f.p == 2 || ((skip_mask >> l) & 1u)) continue;
        const u32 step = f.mp * f.rts, span = f.dim * f.rts, ncol = n / (u32)(f.p - 1);
        if (step == 1 && f.p == 13) {
            // innermost axis of every reference index: a column is 12 contiguous words -- three 16-byte accesses and the recurrence
            // in registers (the strided 4-byte form is 4-way bank-conflicted: lane stride 12 words)
            for (u32 c = threadIdx.x; c < ncol; c += NT) {
                W x[12];
#pragma unroll
                for (int t = 0; t < 12; t += VL) {
                    const V v = *reinterpret_cast<const V*>(lds + c * 12u + t);
#pragma unroll
                    for (int e = 0; e < VL; ++e) x[t + e] = v[e];
                }
                gen_column<W, ZDOM, OP>(x, 0u, 1u, 13, A);
#pragma unroll
                for (int t = 0; t < 12; t += VL) {
                    V v;
#pragma unroll
                    for (int e = 0; e < VL; ++e) v[e] = x[t + e];
                    *reinterpret_cast<V*>(lds + c * 12u + t) = v;
                }
            }
        } else {
            // column c = (o, in): o = c / step by a reciprocal product (exact: c * step < 2^32), not a run-time division per column
            const u32 rcp = step > 1 ? (u32)(((u64)1 << 32) / step) + 1u : 0u;
            for (u32 c = threadIdx.x; c < ncol; c += NT) {
                const u32 o = fdiv(c, step, rcp), in = c - o * step;
                gen_column<W, ZDOM, OP>(lds, o * span + in, step, f.p, A);
            }
        }
        lds_barrier();
    }
}

template <typename W, bool ZDOM, int OP>
__global__ void __launch_bounds__(GEN_T) k_gen_columns(DevRing<W> R, GenDev<W> G, W* data, size_t first_elem, size_t elem_stride, int* fail_flag, u32 skip_mask,
                                                       const W* src) {
    typedef typename Signed<W>::type SW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L;
    const size_t e = first_elem + (blockIdx.x / (unsigned)L) * elem_stride;
    const int j = (int)(blockIdx.x % (unsigned)L);
    const u32 n = G.n;
    W* poly = data + (e * (size_t)L + j) * (size_t)n;
    ColArith<W, ZDOM> A{R.mod[j].q, R.mod[j].qni, R.mod[j].r2, G.plain};
    const W* from = src ? src + (e * (size_t)L + j) * (size_t)n : poly;       // src != null: out of place (same element layout)
    for (u32 i = threadIdx.x; i < n; i += GEN_T) lds[i] = from[i];
    lds_barrier();
    gen_columns_lds<W, ZDOM, OP>(lds, G, A, skip_mask);
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
// SymmSHE modSwitch down (Rescale (a,b) -> b, Eval.hs:130; PT2CT.hs:177,224-229) with the kept limbs never leaving the
// CRT basis.  Lol rescales c0 on the Dec basis and c1 on the Pow basis, one limb at a time.  For a kept limb t the result
// is affine in its own residues,  z_t = x_t C_t - sum_u R_u c_{u,t}  (R_u: centred lift of the u-th dropped residue after
// the earlier drops, in the basis B the component is rescaled in; C_t, c_{u,t}: products of q_u^-1), and crt and the basis
// change B -> Pow are linear, so   crt(z_t) = crt(x_t) C_t - crt(toPow(sum_u reduce_t(R_u) c_{u,t}))
// -- the same residues bit for bit with ddn inverse + (L - ddn) forward transforms per component instead of
// L inverse + (L - ddn) forward (k_rescale_out_lin is the two-power, fused form).
//   k_gen_rescale_drop  per (element, dropped limb u):  crtInv (+ lInv for c0 when the index has odd factors) -> res
//   k_gen_rescale_keep  per (element, kept limb t):     the lift chain from res (element-wise, recomputed per kept limb),
//                       the combination into LDS, (l,) crt, epilogue x_t C_t - . from the CRT-basis input
// ------------------------------------------------------------------------------------------------------
template <typename W, int NT = GEN_T>
__global__ void __launch_bounds__(NT, 4) k_gen_rescale_drop(DevRing<W> R, GenDev<W> G, const W* __restrict__ in, W* __restrict__ res,
                                                            int ddn, int dec_c0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L;
    const size_t e = blockIdx.x / (unsigned)ddn;             // element = 2 ct + component
    const int u = (int)(blockIdx.x % (unsigned)ddn);
    const u32 n = G.n;
    const W* src = in + (e * (size_t)L + u) * (size_t)n;
    W* dst = res + (size_t)blockIdx.x * (size_t)n;
    const W q = R.mod[u].q, qni = R.mod[u].qni;
    typedef typename Vec4<W>::type V;
    constexpr u32 VL = Vec4<W>::LANES;
    const bool vec = n % VL == 0;                            // then every limb-polynomial starts 16-byte aligned
    if (vec) for (u32 i = threadIdx.x * VL; i < n; i += NT * VL) *reinterpret_cast<V*>(lds + i) = *reinterpret_cast<const V*>(src + i);
    else for (u32 i = threadIdx.x; i < n; i += NT) lds[i] = src[i];
    lds_barrier();
    gen_transform<W, true, NT>(lds, G, u, q, qni);
    if (dec_c0 && (e & 1) == 0) {                            // c0 is rescaled on the Dec basis
        ColArith<W, false> A{q, qni, R.mod[u].r2, 0};
        gen_columns_lds<W, false, GEN_LINV, NT>(lds, G, A, 0u);
    }
    const W sc = G.iscale_m[u];                              // crtInv's closing scalar commutes with lInv
    if (vec) {
        for (u32 i = threadIdx.x * VL; i < n; i += NT * VL) {
            V v = *reinterpret_cast<const V*>(lds + i);
#pragma unroll
            for (u32 c = 0; c < VL; ++c) v[c] = csub(mont_mul_lazy(v[c], sc, q, qni), q);
            *reinterpret_cast<V*>(dst + i) = v;
        }
    } else {
        for (u32 i = threadIdx.x; i < n; i += NT) dst[i] = csub(mont_mul_lazy(lds[i], sc, q, qni), q);
    }
}

template <typename W, int NT = GEN_T>
__global__ void __launch_bounds__(NT, 4) k_gen_rescale_keep(DevRing<W> R, GenDev<W> G, const W* __restrict__ in, const W* __restrict__ res,
                                                            W* __restrict__ out, DropTab<W> D, int dec_c0) {
    typedef typename Signed<W>::type SW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L, ddn = D.ddn, Lo = L - ddn;
    const size_t e = blockIdx.x / (unsigned)Lo;
    const int t = ddn + (int)(blockIdx.x % (unsigned)Lo);
    const u32 n = G.n;
    const W q = R.mod[t].q, qni = R.mod[t].qni;
    auto reduce = [](SW z, W qq) -> W {                      // z mod qq; |z| < qq is the common case
        if (z < (SW)qq && z > -(SW)qq) return z < 0 ? (W)(z + (SW)qq) : (W)z;
        SW r = z % (SW)qq;
        return r < 0 ? (W)(r + (SW)qq) : (W)r;
    };
    const W* r0 = res + (e * (size_t)ddn) * (size_t)n;
    typedef typename Vec4<W>::type V;
    constexpr u32 VL = Vec4<W>::LANES;
    const bool vec = n % VL == 0;
    auto combine = [&](const W* y0, W* acc) {                 // one coefficient: y0[u] = residue of dropped limb u
        SW lifted[MAXDROP];
        W a = 0;
#pragma unroll
        for (int u = 0; u < MAXDROP; ++u) {
            if (u >= ddn) continue;
            const W qu = R.mod[u].q, qniu = R.mod[u].qni;
            W y = y0[u];
#pragma unroll
            for (int v = 0; v < u; ++v)                       // the drops of the limbs in front of u come first
                y = csub(mont_mul_lazy((W)(y + (qu - reduce(lifted[v], qu))), D.qinv_m[v][u], qu, qniu), qu);
            lifted[u] = y > ((qu - 1) >> 1) ? (SW)y - (SW)qu : (SW)y;
            a = csub((W)(a + csub(mont_mul_lazy(reduce(lifted[u], q), D.comb_m[u][t], q, qni), q)), q);
        }
        *acc = a;
    };
    if (vec) {
        for (u32 k = threadIdx.x * VL; k < n; k += NT * VL) {
            V in[MAXDROP], o;
#pragma unroll
            for (int u = 0; u < MAXDROP; ++u) if (u < ddn) in[u] = *reinterpret_cast<const V*>(r0 + (size_t)u * n + k);
#pragma unroll
            for (u32 c = 0; c < VL; ++c) {
                W y0[MAXDROP], a;
#pragma unroll
                for (int u = 0; u < MAXDROP; ++u) y0[u] = u < ddn ? in[u][c] : (W)0;
                combine(y0, &a);
                o[c] = a;
            }
            *reinterpret_cast<V*>(lds + k) = o;
        }
    } else {
        for (u32 k = threadIdx.x; k < n; k += NT) {
            W y0[MAXDROP], a;
#pragma unroll
            for (int u = 0; u < MAXDROP; ++u) y0[u] = u < ddn ? r0[(size_t)u * n + k] : (W)0;
            combine(y0, &a);
            lds[k] = a;
        }
    }
    lds_barrier();
    if (dec_c0 && (e & 1) == 0) {                            // Dec -> Pow
        ColArith<W, false> A{q, qni, R.mod[t].r2, 0};
        gen_columns_lds<W, false, GEN_L, NT>(lds, G, A, 0u);
    }
    gen_transform<W, false, NT>(lds, G, t, q, qni);
    const W Ct = D.comb_m[0][t];
    const W* x = in + (e * (size_t)L + t) * (size_t)n;
    W* o = out + (e * (size_t)Lo + (t - ddn)) * (size_t)n;
    if (vec) {
        for (u32 k = threadIdx.x * VL; k < n; k += NT * VL) {
            const V xv = *reinterpret_cast<const V*>(x + k), lv = *reinterpret_cast<const V*>(lds + k);
            V r;
#pragma unroll
            for (u32 c = 0; c < VL; ++c) r[c] = csub((W)(csub(mont_mul_lazy(xv[c], Ct, q, qni), q) + (q - lv[c])), q);
            *reinterpret_cast<V*>(o + k) = r;
        }
    } else {
        for (u32 k = threadIdx.x; k < n; k += NT)
            o[k] = csub((W)(csub(mont_mul_lazy(x[k], Ct, q, qni), q) + (q - lds[k])), q);
    }
}

// The same modSwitch with the result left in the Pow basis (what Lol's rescale returns, and what the next tunnel reads: a hop
// handed over in the Pow basis saves that tunnel's crtInv): per (element, kept limb) crtInv of the limb itself (+ lInv for c0),
// z = x C_t - sum_u reduce_t(R_u) c_{u,t} coefficient-wise (crtInv's closing scalar folded into C_t), (l,) store.  Same number
// of transforms as k_gen_rescale_keep (one per kept limb-polynomial), inverse instead of forward.
template <typename W, int NT = GEN_T>
__global__ void __launch_bounds__(NT, 4) k_gen_rescale_keep_pow(DevRing<W> R, GenDev<W> G, const W* __restrict__ in, const W* __restrict__ res,
                                                                   W* __restrict__ out, DropTab<W> D, int dec_c0) {
    typedef typename Signed<W>::type SW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W* lds = reinterpret_cast<W*>(smem);
    const int L = R.L, ddn = D.ddn, Lo = L - ddn;
    const size_t e = blockIdx.x / (unsigned)Lo;
    const int t = ddn + (int)(blockIdx.x % (unsigned)Lo);
    const u32 n = G.n;
    const W q = R.mod[t].q, qni = R.mod[t].qni;
    auto reduce = [](SW z, W qq) -> W {
        if (z < (SW)qq && z > -(SW)qq) return z < 0 ? (W)(z + (SW)qq) : (W)z;
        SW r = z % (SW)qq;
        return r < 0 ? (W)(r + (SW)qq) : (W)r;
    };
    typedef typename Vec4<W>::type V;
    constexpr u32 VL = Vec4<W>::LANES;
    const bool vec = n % VL == 0;
    const W* x = in + (e * (size_t)L + t) * (size_t)n;
    if (vec) for (u32 i = threadIdx.x * VL; i < n; i += NT * VL) *reinterpret_cast<V*>(lds + i) = *reinterpret_cast<const V*>(x + i);
    else for (u32 i = threadIdx.x; i < n; i += NT) lds[i] = x[i];
    lds_barrier();
    gen_transform<W, true, NT>(lds, G, t, q, qni);
    const bool dec = dec_c0 && (e & 1) == 0;
    ColArith<W, false> A{q, qni, R.mod[t].r2, 0};
    if (dec) gen_columns_lds<W, false, GEN_LINV, NT>(lds, G, A, 0u);
    const W scCt = csub(mont_mul_lazy(G.iscale_m[t], D.comb_m[0][t], q, qni), q);      // crtInv's closing scalar times C_t (Montgomery form)
    const W* r0 = res + (e * (size_t)ddn) * (size_t)n;
    auto combine = [&](const W* y0) -> W {                     // one coefficient: y0[u] = residue of dropped limb u
        SW lifted[MAXDROP];
        W a = 0;
#pragma unroll
        for (int u = 0; u < MAXDROP; ++u) {
            if (u >= ddn) continue;
            const W qu = R.mod[u].q, qniu = R.mod[u].qni;
            W y = y0[u];
#pragma unroll
            for (int v = 0; v < u; ++v)
                y = csub(mont_mul_lazy((W)(y + (qu - reduce(lifted[v], qu))), D.qinv_m[v][u], qu, qniu), qu);
            lifted[u] = y > ((qu - 1) >> 1) ? (SW)y - (SW)qu : (SW)y;
            a = csub((W)(a + csub(mont_mul_lazy(reduce(lifted[u], q), D.comb_m[u][t], q, qni), q)), q);
        }
        return a;
    };
    // every lane rewrites the words it read: no barrier between the column pass above and this loop is needed beyond the one it ends with
    if (vec) {
        for (u32 k = threadIdx.x * VL; k < n; k += NT * VL) {
            V rin[MAXDROP], z = *reinterpret_cast<const V*>(lds + k);
#pragma unroll
            for (int u = 0; u < MAXDROP; ++u) if (u < ddn) rin[u] = *reinterpret_cast<const V*>(r0 + (size_t)u * n + k);
#pragma unroll
            for (u32 c = 0; c < VL; ++c) {
                W y0[MAXDROP];
#pragma unroll
                for (int u = 0; u < MAXDROP; ++u) y0[u] = u < ddn ? rin[u][c] : (W)0;
                z[c] = csub((W)(csub(mont_mul_lazy(z[c], scCt, q, qni), q) + (q - combine(y0))), q);
            }
            *reinterpret_cast<V*>(lds + k) = z;
        }
    } else {
        for (u32 k = threadIdx.x; k < n; k += NT) {
            W y0[MAXDROP];
#pragma unroll
            for (int u = 0; u < MAXDROP; ++u) y0[u] = u < ddn ? r0[(size_t)u * n + k] : (W)0;
            lds[k] = csub((W)(csub(mont_mul_lazy(lds[k], scCt, q, qni), q) + (q - combine(y0))), q);
        }
    }
    lds_barrier();
    if (dec) gen_columns_lds<W, false, GEN_L, NT>(lds, G, A, 0u);
    W* o = out + (e * (size_t)Lo + (t - ddn)) * (size_t)n;
    if (vec) for (u32 i = threadIdx.x * VL; i < n; i += NT * VL) *reinterpret_cast<V*>(o + i) = *reinterpret_cast<const V*>(lds + i);
    else for (u32 i = threadIdx.x; i < n; i += NT) o[i] = lds[i];
}

// Threads per workgroup of the LDS-resident transform kernels.  The pass matrices live in SGPRs (gen_sym_pass), so the kernels need
// ~50-60 VGPRs and four 512-thread workgroups (eight waves per SIMD) fit a CU whenever four polynomials fit its LDS; small rings
// keep 2-wave workgroups, eight or more to a CU.
template <typename W>
inline int gen_threads(const GenDev<W>& G) {
    if (G.nt == 128 || G.nt == 256 || G.nt == 512) return G.nt;
    const size_t bytes = (size_t)G.n * sizeof(W);
    // more than 40 KiB per polynomial (H3', phi = 11520): only three workgroups fit a CU's LDS, so they are made 8 waves wide
    // (measured on H3': crt 63 -> 59 ns, mul_ 302 k -> 338 k/s; on the 36-KiB rings 512 threads change nothing, below 18 KiB they lose)
    if (bytes > 40960) return 512;
#ifndef ALCH_GEN_SMALL_BYTES
#define ALCH_GEN_SMALL_BYTES 18431        // round 4: a polynomial of exactly 18 KiB (phi = 4608: H0', the E' of the H1' -> H2' hop) does better on 256 threads
#endif
    return bytes <= ALCH_GEN_SMALL_BYTES ? GEN_T_SMALL : GEN_T;
}

template <typename W, int NT>
inline hipError_t gen_launch_rescale_lin_nt(const DevRing<W>& R, const GenDev<W>& G, const W* in, W* res, W* out, const DropTab<W>& D,
                                            int dec_c0, size_t nelem, hipStream_t stream, bool pow_out) {
    const size_t lds_bytes = (size_t)G.n * sizeof(W);
    auto k1 = k_gen_rescale_drop<W, NT>;
    auto k2 = pow_out ? k_gen_rescale_keep_pow<W, NT> : k_gen_rescale_keep<W, NT>;
    hipError_t e;
    if ((e = set_lds(k1, lds_bytes)) != hipSuccess) return e;
    if ((e = set_lds(k2, lds_bytes)) != hipSuccess) return e;
    hipLaunchKernelGGL(k1, dim3((unsigned)(nelem * (size_t)D.ddn)), dim3(NT), lds_bytes, stream, R, G, in, res, D.ddn, dec_c0);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(k2, dim3((unsigned)(nelem * (size_t)(R.L - D.ddn))), dim3(NT), lds_bytes, stream, R, G, in, res, out, D, dec_c0);
    return hipGetLastError();
}

template <typename W>
inline hipError_t gen_launch_rescale_lin(const DevRing<W>& R, const GenDev<W>& G, const W* in, W* res, W* out, const DropTab<W>& D,
                                         int dec_c0, size_t nelem, hipStream_t stream, bool pow_out = false) {
    switch (gen_threads(G)) {
    case 128: return gen_launch_rescale_lin_nt<W, 128>(R, G, in, res, out, D, dec_c0, nelem, stream, pow_out);
    case 512: return gen_launch_rescale_lin_nt<W, 512>(R, G, in, res, out, D, dec_c0, nelem, stream, pow_out);
    default: return gen_launch_rescale_lin_nt<W, 256>(R, G, in, res, out, D, dec_c0, nelem, stream, pow_out);
    }
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
                       c.elem_stride ? c.elem_stride : (size_t)1, c.fail_flag, c.skip_mask, c.src);
    return hipGetLastError();
}

template <typename W, int NT>
inline hipError_t gen_run_nt(const GenCall<W>& c, size_t lds_bytes) {
    hipError_t e;
    switch (c.op) {
    case GEN_CRT: {
        auto k = k_gen_crt<W, false, NT>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(NT), lds_bytes, c.stream, *c.ring, *c.gen, c.data, c.src, c.first_poly);
        break;
    }
    case GEN_CRTINV: {
        auto k = k_gen_crt<W, true, NT>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(NT), lds_bytes, c.stream, *c.ring, *c.gen, c.data, c.src, c.first_poly);
        break;
    }
    case GEN_CRT_BASE2: {
        auto k = k_gen_crt_base2_digits<W, NT>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(NT), lds_bytes, c.stream, *c.ring, *c.gen, c.src, c.data, c.b2_first, c.b2_kd, c.b2_D);
        break;
    }
    case GEN_CRT_DIGITS: {
        auto k = k_gen_crt_digits<W, NT>;
        if ((e = set_lds(k, lds_bytes)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)c.npoly), dim3(NT), lds_bytes, c.stream, *c.ring, *c.gen, c.src, c.data, c.balanced ? 1 : 0, c.with_diag ? 1 : 0,
                           c.src_limbs ? c.src_limbs : c.ring->L, c.src_first);
        break;
    }
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <typename W>
inline hipError_t gen_run(const GenCall<W>& c) {
    const size_t lds_bytes = (size_t)c.gen->n * sizeof(W);
    switch (c.op) {
    case GEN_CRT: case GEN_CRTINV: case GEN_CRT_BASE2: case GEN_CRT_DIGITS:
        switch (gen_threads(*c.gen)) {
        case 128: return gen_run_nt<W, 128>(c, lds_bytes);
        case 512: return gen_run_nt<W, 512>(c, lds_bytes);
        default: return gen_run_nt<W, 256>(c, lds_bytes);
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
