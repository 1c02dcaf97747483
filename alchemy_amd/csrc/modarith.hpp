// Word-size modular arithmetic for gfx950 (and the host-side table builder).
//
// Montgomery multiplication with R = 2^32 (32-bit residues, q < 2^31) or R = 2^64 (q < 2^62).
// Measured on MI355X (tools/ubench): v_mul_lo/v_mul_hi/v_mad_u64_u32 all issue at ~4.5 cycles per
// wave64, v_add/v_sub at ~2.6, v_min_u32 at ~4.1.  The 32-bit Montgomery product below compiles to
// exactly three multiplier instructions (v_mad_u64_u32, v_mul_lo_u32, v_mad_u64_u32) with no
// separate add, one fewer instruction than Shoup's and with one-word twiddles.
//
// Ranges.  mont_mul_lazy(a, b): a any word, b < q  ->  a*b*R^-1 mod q in [0, 2q).
// Ring data are kept "lazy" in [0, 2q) between butterfly stages (2q < 2^32 / 2^63); csub() brings
// a [0,2q) value to [0,q).  4q does not fit a 32-bit word for ALCHEMY's 31-bit moduli
// (examples/HomomRLWR.hs:37-43), so Harvey's [0,4q) butterflies are not used.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ALCH_HD __host__ __device__ __forceinline__
#else
#define ALCH_HD inline
#endif

namespace alch {

typedef uint32_t u32;
typedef uint64_t u64;

template <typename W>
struct ModP {
    W q;     // modulus
    W qni;   // -q^-1 mod R
    W r1;    // R mod q          (Montgomery form of 1)
    W r2;    // R^2 mod q        (to_mont multiplier)
};

ALCH_HD u32 mont_mul_lazy(u32 a, u32 b, u32 q, u32 qni) {
    u64 p = (u64)a * b;
    u32 m = (u32)p * qni;
    return (u32)((p + (u64)m * q) >> 32);
}

ALCH_HD u64 mul_hi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

ALCH_HD u64 mont_mul_lazy(u64 a, u64 b, u64 q, u64 qni) {
    // One 128-bit accumulation (p + m q < 2 q 2^64 < 2^127): on gfx950 this form compiles to 18 % fewer VALU
    // instructions than hi(a b) + hi(m q) + carry (28 against 34 per product, mostly register-pair moves).
    const unsigned __int128 p = (unsigned __int128)a * b;
    const u64 m = (u64)p * qni;
    return (u64)((p + (unsigned __int128)m * q) >> 64);
}

// [0,2q) -> [0,q).  x - q wraps to a huge value when x < q, so the unsigned min picks x.
ALCH_HD u32 csub(u32 x, u32 q) { u32 y = x - q; return y < x ? y : x; }
ALCH_HD u64 csub(u64 x, u64 q) { u64 y = x - q; return y < x ? y : x; }

template <typename W>
ALCH_HD W mont_mul(W a, W b, const ModP<W>& m) { return csub(mont_mul_lazy(a, b, m.q, m.qni), m.q); }
template <typename W>
ALCH_HD W add_mod(W a, W b, W q) { return csub((W)(a + b), q); }            // a,b in [0,q)
template <typename W>
ALCH_HD W sub_mod(W a, W b, W q) { W d = a - b; W e = d + q; return e < d ? e : d; }   // a,b in [0,q)

// Forward (Cooley-Tukey) butterfly on lazy values: x,y in [0,2q), w = twiddle in Montgomery form.
//   x' = x + w*y, y' = x - w*y   (mod q), outputs in [0,2q).  10 VALU instructions for W = u32.
template <typename W>
ALCH_HD void bfly_fwd(W& x, W& y, W w, W q, W qni) {
    W xx = csub(x, q);
    W t = csub(mont_mul_lazy(y, w, q, qni), q);
    x = xx + t;
    y = xx + (q - t);
}

// q < 2^30 (4q fits a word): Harvey's lazy butterflies.  Forward: x, y in [0,4q) -> [0,4q); the product takes ANY word, so y is
// never reduced.  8 VALU instructions instead of 10.  Inverse: x, y in [0,2q) -> [0,2q); 8 instead of 10.
ALCH_HD void bfly_fwd4(u32& x, u32& y, u32 w, u32 q, u32 qni) {
    const u32 q2 = 2u * q;
    const u32 xx = csub(x, q2);
    const u32 t = mont_mul_lazy(y, w, q, qni);
    x = xx + t;
    y = xx + (q2 - t);
}
ALCH_HD void bfly_inv4(u32& x, u32& y, u32 w, u32 q, u32 qni) {
    const u32 q2 = 2u * q;
    const u32 s = x + y, d = x + (q2 - y);
    x = csub(s, q2);
    y = mont_mul_lazy(d, w, q, qni);
}

// Plantard multiplication by a precomputed constant (Plantard, "Efficient word size modular arithmetic",
// 2021).  For a constant c the table holds  br = (-c * 2^64 mod q) * q^-1 mod 2^64.  With T = a*br mod 2^64,
//   result = floor(((T >> 32) + 1) * q / 2^32)  ==  a*c mod q,   exactly reduced, for ANY 32-bit a
// provided q < 2^31 (then a * (-c 2^64 mod q) < 2^63 and q 2^32 + that product < 2^64, which is the
// condition for the rounded quotient to be exact).  Four instructions and no correction step, against five
// for Montgomery + conditional subtract; the price is a two-word constant.
ALCH_HD u32 plant_mul(u32 a, u64 br, u32 q) {
    const u64 lo = (u64)a * (u32)br;
    const u32 t1 = (u32)((u64)a * (u32)(br >> 32) + (lo >> 32));
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(t1 + 1u, q);
#else
    return (u32)(((u64)(u32)(t1 + 1u) * q) >> 32);
#endif
}

// Forward butterfly with a Plantard twiddle: 9 VALU instructions.  x,y in [0,2q) -> outputs in [0,2q].
ALCH_HD void bfly_fwd(u32& x, u32& y, u64 br, u32 q, u32 /*qni*/) {
    const u32 xx = csub(x, q);
    const u32 t = plant_mul(y, br, q);
    x = xx + t;
    y = xx + (q - t);
}

// Inverse (Gentleman-Sande) butterfly: x' = x + y, y' = (x - y) * w.  in/out in [0,2q).
template <typename W>
ALCH_HD void bfly_inv(W& x, W& y, W w, W q, W qni) {
    W a = csub(x, q), b = csub(y, q);
    x = a + b;
    y = mont_mul_lazy((W)(a - b + q), w, q, qni);
}

// ---- 64-bit rings: Shoup twiddles ----------------------------------------------------------------------
// A 64-bit Montgomery product is 11 multiplier instructions and, with the register-pair moves hipcc needs for the
// zero-extended addends, 27 VALU instructions.  A product by a CONSTANT w (a twiddle) with the precomputed quotient
// w' = floor(w 2^64 / q) (Shoup / Harvey, "Faster arithmetic for number-theoretic transforms", 2014):
//     Q = hi64(a w'),   r = a w - Q q  (mod 2^64)   in [0, 2q)   for ANY 64-bit a
// needs the high half of one product and the low halves of two: 17 VALU instructions (the subtraction rides on the
// multiply-add chain as + Q (2^64 - q)).  4q < 2^64 for every supported modulus (q < 2^62), so the forward butterflies
// are Harvey's: values in [0,4q) inside a register-resident pass, one conditional subtraction per butterfly, and
// [0,2q) again at the pass boundary (what every loader and epilogue of the engine expects).  w is a plain residue
// (ring data are never in Montgomery form), two words per twiddle.
struct alignas(16) Sh64 { u64 w, wp; };

ALCH_HD u64 shoup_mul_lazy(u64 a, Sh64 t, u64 q) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), p0 = (u32)t.wp, p1 = (u32)(t.wp >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    const u64 m1 = (u64)a1 * p0 + __umulhi(a0, p0);
#else
    const u64 m1 = (u64)a1 * p0 + (((u64)a0 * p0) >> 32);
#endif
    const u64 m2 = (u64)a0 * p1 + (u32)m1;
    const u64 Q = (u64)a1 * p1 + ((m1 >> 32) + (m2 >> 32));     // hi64(a w'), exact
    return a * t.w + Q * ((u64)0 - q);
}

// Forward butterfly of stage r of a register-resident pass (FIRST: inputs in [0,2q), else [0,4q); LAST: outputs
// reduced to [0,2q), else left in [0,4q)).
// (first / last are compile-time constants after the stage loops are unrolled)
ALCH_HD void bfly_fwd_st(u64& x, u64& y, Sh64 w, u64 q, u64 /*qni*/, bool FIRST, bool LAST) {
    const u64 q2 = 2 * q;
    const u64 xx = FIRST ? x : csub(x, q2);
    const u64 t = shoup_mul_lazy(y, w, q);
    x = xx + t;
    y = xx + (q2 - t);
    if (LAST) { x = csub(x, q2); y = csub(y, q2); }
}
// Inverse (Gentleman-Sande) butterfly, [0,2q) in and out: one conditional subtraction.
ALCH_HD void bfly_inv(u64& x, u64& y, Sh64 w, u64 q, u64 /*qni*/) {
    const u64 q2 = 2 * q;
    const u64 s = x + y, d = x + (q2 - y);
    x = csub(s, q2);
    y = shoup_mul_lazy(d, w, q);
}
// every other twiddle type: the stage position does not matter
template <typename W, typename TW>
ALCH_HD void bfly_fwd_st(W& x, W& y, TW w, W q, W qni, bool, bool) { bfly_fwd(x, y, w, q, qni); }

// ---- host-side helpers (table building) ------------------------------------------------------
inline u64 h_mulmod(u64 a, u64 b, u64 q) { return (u64)(((unsigned __int128)a * b) % q); }
inline u64 h_powmod(u64 b, u64 e, u64 q) {
    u64 r = 1 % q;
    b %= q;
    while (e) {
        if (e & 1) r = h_mulmod(r, b, q);
        b = h_mulmod(b, b, q);
        e >>= 1;
    }
    return r;
}

inline Sh64 h_shoup_const(u64 c, u64 q) {
    Sh64 t;
    t.w = c;
    t.wp = (u64)((((unsigned __int128)c) << 64) / q);
    return t;
}

// Plantard constant of c (< q) for modulus q < 2^31.
inline u64 h_plant_const(u64 c, u64 q) {
    u64 inv = 1;                                 // q^-1 mod 2^64
    for (int i = 0; i < 7; ++i) inv *= 2 - q * inv;
    const u64 r64 = h_powmod(2, 64, q);
    const u64 b = h_mulmod((q - c % q) % q, r64, q);      // -c * 2^64 mod q
    return b * inv;
}

template <typename W>
inline ModP<W> make_modp(u64 q) {
    ModP<W> m;
    m.q = (W)q;
    W inv = 1;                                  // Newton: inv = q^-1 mod 2^bits
    for (int i = 0; i < 7; ++i) inv *= (W)2 - (W)q * inv;
    m.qni = (W)0 - inv;
    const int bits = 8 * (int)sizeof(W);
    u64 r1 = h_powmod(2, (u64)bits, q);
    m.r1 = (W)r1;
    m.r2 = (W)h_mulmod(r1, r1, q);
    return m;
}

}  // namespace alch
