/*
 * oracle/lol_tensor_gen.c -- CPU restatement of Lol's Tensor operations for an ARBITRARY cyclotomic index
 * (SURVEY.md 8f N3): crt / crtInv, l / lInv, mulG / divG on the Pow, Dec and CRT bases, and the SymmSHE multiply +
 * key switch on top of them.  Companion of oracle/lol_tensor.c (two-power indices); same rules:
 *
 * TEST INFRASTRUCTURE ONLY -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 * PARITY UNPINNED -- the code restated lives in the un-vendored lol / lol-cpp packages
 * (/root/reference/stack.yaml:54-60); it is checked against oracle/model_gen.py (direct-evaluation definitions),
 * not against Lol.
 *
 * Reference call sites: the composite indices H0' .. H5' of examples/Common.hs:38-54 (HomomRLWR.hs:29-35,
 * Tunnel.hs:26-32); (*) on CT with mulG on every coefficient (Crypto/Alchemy/Interpreter/Eval.hs:65-67);
 * keySwitchQuadCirc (Eval.hs:133); modSwitch (Eval.hs:130); decrypt's divG (PT2CT.hs:91-99).
 *
 * Algorithm (the toolkit's sparse decompositions, as lol-cpp applies them: one prime-power factor at a time over a
 * vector viewed as [lts][phi(p^e)][rts], in place, tuple-interleaved Int64, a hardware % per product):
 *   CRT_m     = kron_l CRT_{p_l^e_l}
 *   CRT_{p^e} = (DFT_{m'} (x) I_{p-1}) . T . (I_{m'} (x) CRT_p),   m' = p^(e-1), rows (i0, i1), columns (j0, j1)
 *               CRT_p[i0][j0] = w_p^(i0 j0)  (i0 = 1..p-1, j0 = 0..p-2),  T = diag w_{p^e}^(i0 j1),
 *               DFT_{m'} = radix-p decimation in frequency, natural order in, digit-reversed order out
 *   L_{p^e}   = L_p (x) I_{m'}  (prefix sums),  G_{p^e} = G_p (x) I_{m'}
 * Conventions are those of oracle/model_gen.py (root rule, slot <-> unit map, Pow/Dec index order).
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;
typedef uint64_t u64;
typedef unsigned __int128 u128;

#define G_MAX_LIMBS 16
#define G_MAX_FACT 8

typedef struct {
    int p, e;
    i64 mp;          /* p^(e-1) */
    i64 dim;         /* phi(p^e) */
    i64 rts;         /* product of the later factors' dims */
    /* per limb tables (NULL when the limb has no CRT) */
    i64 *crtp[G_MAX_LIMBS], *crtp_inv[G_MAX_LIMBS];      /* (p-1)^2 */
    i64 *dftp[G_MAX_LIMBS], *dftp_inv[G_MAX_LIMBS];      /* p^2, inverse includes 1/p */
    i64 *wpe[G_MAX_LIMBS], *wpe_inv[G_MAX_LIMBS];        /* powers of w_{p^e} and of its inverse, p^e entries */
} gfact;

typedef struct {
    i64 m, n;
    int nf, L;
    int has_crt;
    i64 q[G_MAX_LIMBS];              /* 0 = the integers (Pow / Dec operations only) */
    gfact f[G_MAX_FACT];
    i64 *gcrt[G_MAX_LIMBS], *gcrt_inv[G_MAX_LIMBS];      /* CRT image of g and its inverse, n entries */
} orcg_ring;

/* ------------------------------------------------------------------ scalar arithmetic */
static inline i64 mulmod(i64 a, i64 b, i64 q) {
    if (q < ((i64)1 << 31)) return (a * b) % q;
    return (i64)(((u128)(u64)a * (u64)b) % (u64)q);
}
static inline i64 addmod(i64 a, i64 b, i64 q) { i64 s = a + b; return s >= q ? s - q : s; }
static inline i64 submod(i64 a, i64 b, i64 q) { i64 s = a - b; return s < 0 ? s + q : s; }
static i64 powmod(i64 b, u64 e, i64 q) {
    i64 r = 1 % q;
    b %= q;
    while (e) { if (e & 1) r = mulmod(r, b, q); b = mulmod(b, b, q); e >>= 1; }
    return r;
}
static i64 invmod(i64 a, i64 q) {            /* extended Euclid; returns 0 when a is not a unit */
    i64 r0 = q, r1 = ((a % q) + q) % q, t0 = 0, t1 = 1;
    while (r1) { i64 k = r0 / r1, r2 = r0 - k * r1, t2 = t0 - k * t1; r0 = r1; r1 = r2; t0 = t1; t1 = t2; }
    if (r0 != 1) return 0;
    return ((t0 % q) + q) % q;
}
extern i64 orc_smallest_generator(i64 q);      /* oracle/lol_tensor.c: the root rule */
static inline i64 centred(i64 x, i64 q) { return (2 * x < q) ? x : x - q; }

/* inverse of a d x d matrix mod prime q (Gauss-Jordan); returns 0 if singular */
static int mat_inv(const i64 *M, i64 *out, int d, i64 q) {
    i64 *A = malloc(sizeof(i64) * d * 2 * d);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < 2 * d; ++j) A[i * 2 * d + j] = j < d ? M[i * d + j] : (j - d == i);
    for (int c = 0; c < d; ++c) {
        int piv = -1;
        for (int r = c; r < d; ++r) if (A[r * 2 * d + c]) { piv = r; break; }
        if (piv < 0) { free(A); return 0; }
        for (int j = 0; j < 2 * d; ++j) { i64 t = A[c * 2 * d + j]; A[c * 2 * d + j] = A[piv * 2 * d + j]; A[piv * 2 * d + j] = t; }
        i64 inv = invmod(A[c * 2 * d + c], q);
        for (int j = 0; j < 2 * d; ++j) A[c * 2 * d + j] = mulmod(A[c * 2 * d + j], inv, q);
        for (int r = 0; r < d; ++r) {
            if (r == c || !A[r * 2 * d + c]) continue;
            i64 f = A[r * 2 * d + c];
            for (int j = 0; j < 2 * d; ++j) A[r * 2 * d + j] = submod(A[r * 2 * d + j], mulmod(f, A[c * 2 * d + j], q), q);
        }
    }
    for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) out[i * d + j] = A[i * 2 * d + d + j];
    free(A);
    return 1;
}

static i64 digitrev(i64 x, int p, int digits) {
    i64 r = 0;
    for (int t = 0; t < digits; ++t) { r = r * p + x % p; x /= p; }
    return r;
}

/* ------------------------------------------------------------------ ring context */
orcg_ring *orcg_ring_new(void) { return calloc(1, sizeof(orcg_ring)); }

void orcg_ring_delete(orcg_ring *r) {
    if (!r) return;
    for (int l = 0; l < r->nf; ++l)
        for (int j = 0; j < r->L; ++j) {
            free(r->f[l].crtp[j]); free(r->f[l].crtp_inv[j]); free(r->f[l].dftp[j]); free(r->f[l].dftp_inv[j]);
            free(r->f[l].wpe[j]); free(r->f[l].wpe_inv[j]);
        }
    for (int j = 0; j < r->L; ++j) { free(r->gcrt[j]); free(r->gcrt_inv[j]); }
    free(r);
}

i64 orcg_n(const orcg_ring *r) { return r->n; }
int orcg_has_crt(const orcg_ring *r) { return r->has_crt; }

/* unit of Z_m^* of CRT slot `lin` */
static i64 slot_unit(const orcg_ring *r, i64 lin) {
    i64 u = 0, mod = 1;
    for (int l = 0; l < r->nf; ++l) {
        const gfact *f = &r->f[l];
        i64 s = (lin / f->rts) % f->dim;
        i64 ml = f->mp * f->p;
        i64 i0 = s / f->mp + 1, i1 = digitrev(s % f->mp, f->p, f->e - 1);
        i64 ul = (i0 + f->p * i1) % ml;
        i64 t = mod > 1 ? mulmod(((ul - u) % ml + ml) % ml, invmod(mod % ml, ml), ml) : ul;
        u += mod * t;
        mod *= ml;
    }
    return u % r->m;
}

/* 0 ok, -1 bad argument, -2 modulus not prime.  Moduli that are not 1 mod m (or 0 = integers) give a ring without
 * CRT basis (Lol: crtFuncs = Nothing): Pow / Dec operations only. */
int orcg_ring_init(orcg_ring *r, i64 m, int L, const i64 *q) {
    extern int orc_is_prime(u64);
    memset(r, 0, sizeof *r);
    if (m < 1 || L < 1 || L > G_MAX_LIMBS) return -1;
    r->m = m; r->L = L;
    i64 rem = m;
    for (i64 p = 2; p * p <= rem; p += (p == 2 ? 1 : 2)) {
        if (rem % p) continue;
        if (r->nf == G_MAX_FACT) return -1;
        gfact *f = &r->f[r->nf++];
        f->p = (int)p; f->e = 0; f->mp = 1;
        while (rem % p == 0) { rem /= p; if (f->e++) f->mp *= p; }
    }
    if (rem > 1) { if (r->nf == G_MAX_FACT) return -1; gfact *f = &r->f[r->nf++]; f->p = (int)rem; f->e = 1; f->mp = 1; }
    r->n = 1;
    for (int l = 0; l < r->nf; ++l) { r->f[l].dim = (r->f[l].p - 1) * r->f[l].mp; r->n *= r->f[l].dim; }
    i64 s = r->n;
    for (int l = 0; l < r->nf; ++l) { s /= r->f[l].dim; r->f[l].rts = s; }
    r->has_crt = 1;
    for (int j = 0; j < L; ++j) {
        r->q[j] = q[j];
        if (q[j] == 0) { r->has_crt = 0; continue; }
        if (q[j] < 2) return -1;
        if ((q[j] - 1) % m) r->has_crt = 0;
    }
    if (!r->has_crt) return 0;
    for (int j = 0; j < L; ++j) {
        const i64 qj = q[j];
        if (!orc_is_prime((u64)qj)) return -2;
        const i64 wm = powmod(orc_smallest_generator(qj), (u64)(qj - 1) / (u64)m, qj);
        for (int l = 0; l < r->nf; ++l) {
            gfact *f = &r->f[l];
            const int p = f->p;
            const i64 pe = f->mp * p;
            const i64 w = powmod(wm, (u64)(m / pe), qj), wi = invmod(w, qj);
            f->wpe[j] = malloc(sizeof(i64) * pe); f->wpe_inv[j] = malloc(sizeof(i64) * pe);
            i64 a = 1, b = 1;
            for (i64 t = 0; t < pe; ++t) { f->wpe[j][t] = a; f->wpe_inv[j][t] = b; a = mulmod(a, w, qj); b = mulmod(b, wi, qj); }
            /* CRT_p[i0-1][j0] = w_p^(i0 j0), w_p = w_{p^e}^(m') */
            f->crtp[j] = malloc(sizeof(i64) * (p - 1) * (p - 1)); f->crtp_inv[j] = malloc(sizeof(i64) * (p - 1) * (p - 1));
            for (int i0 = 1; i0 < p; ++i0)
                for (int j0 = 0; j0 < p - 1; ++j0) f->crtp[j][(i0 - 1) * (p - 1) + j0] = f->wpe[j][((i64)i0 * j0 % p) * f->mp];
            if (!mat_inv(f->crtp[j], f->crtp_inv[j], p - 1, qj)) return -1;
            /* DFT_p[f][t] = w_p^(f t); inverse = conj / p */
            f->dftp[j] = malloc(sizeof(i64) * p * p); f->dftp_inv[j] = malloc(sizeof(i64) * p * p);
            const i64 pinv = invmod(p % qj, qj);
            for (int a2 = 0; a2 < p; ++a2)
                for (int b2 = 0; b2 < p; ++b2) {
                    f->dftp[j][a2 * p + b2] = f->wpe[j][((i64)a2 * b2 % p) * f->mp];
                    f->dftp_inv[j][a2 * p + b2] = mulmod(f->wpe_inv[j][((i64)a2 * b2 % p) * f->mp], pinv, qj);
                }
        }
        /* CRT image of g = prod_{odd p} (1 - zeta_p) */
        r->gcrt[j] = malloc(sizeof(i64) * r->n); r->gcrt_inv[j] = malloc(sizeof(i64) * r->n);
        for (i64 sl = 0; sl < r->n; ++sl) {
            const i64 u = slot_unit(r, sl);
            i64 v = 1;
            for (int l = 0; l < r->nf; ++l) {
                if (r->f[l].p == 2) continue;
                const i64 wp = powmod(wm, (u64)(((u128)(m / r->f[l].p) * (u64)u) % (u64)m), qj);
                v = mulmod(v, submod(1, wp, qj), qj);
            }
            r->gcrt[j][sl] = v;
            r->gcrt_inv[j][sl] = invmod(v, qj);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ one prime-power factor of crt / crtInv */
/* dense d x d matrix along a sub-axis: elements x[base + t*step], t < d */
static void dense(i64 *x, i64 L, i64 base, i64 step, int d, const i64 *M, i64 q) {
    i64 in[64], out[64];
    for (int t = 0; t < d; ++t) in[t] = x[(base + t * step) * L];
    for (int s = 0; s < d; ++s) {
        i64 acc = 0;
        for (int t = 0; t < d; ++t) acc = addmod(acc, mulmod(M[s * d + t], in[t], q), q);
        out[s] = acc;
    }
    for (int s = 0; s < d; ++s) x[(base + s * step) * L] = out[s];
}

static void ppcrt(const orcg_ring *r, int l, int j, i64 *x /* limb j: stride L */, int inverse) {
    const gfact *f = &r->f[l];
    const int p = f->p, L = r->L;
    const i64 q = r->q[j], mp = f->mp, rts = f->rts, dim = f->dim, lts = r->n / (dim * rts);
    for (i64 o = 0; o < lts; ++o)
        for (i64 in = 0; in < rts; ++in) {
            const i64 b0 = o * dim * rts + in;                  /* axis position a lives at b0 + a * rts */
            if (!inverse) {
                if (p > 2)
                    for (i64 j1 = 0; j1 < mp; ++j1) dense(x, L, b0 + j1 * rts, mp * rts, p - 1, f->crtp[j], q);
                for (i64 i0 = 1; i0 < p; ++i0) {
                    i64 *row = x;                               /* row i0: positions (i0-1) mp + j1 */
                    const i64 rb = b0 + (i0 - 1) * mp * rts;
                    if (mp > 1)
                        for (i64 j1 = 0; j1 < mp; ++j1)          /* T: w_{p^e}^(i0 j1) */
                            row[(rb + j1 * rts) * L] = mulmod(row[(rb + j1 * rts) * L], f->wpe[j][(i0 * j1) % (mp * p)], q);
                    /* DFT_{m'}: decimation in frequency, radix p */
                    for (i64 B = mp; B > 1; B /= p) {
                        const i64 sub = B / p;
                        for (i64 blk = 0; blk < mp; blk += B)
                            for (i64 off = 0; off < sub; ++off) {
                                dense(x, L, rb + (blk + off) * rts, sub * rts, p, f->dftp[j], q);
                                if (sub > 1)
                                    for (int fq = 1; fq < p; ++fq) {   /* w_B^(fq off) = w_{p^e}^(p (mp/B) fq off) */
                                        i64 *v = &x[(rb + (blk + fq * sub + off) * rts) * L];
                                        *v = mulmod(*v, f->wpe[j][(p * (mp / B) * fq * off) % (mp * p)], q);
                                    }
                            }
                    }
                }
            } else {
                for (i64 i0 = 1; i0 < p; ++i0) {
                    const i64 rb = b0 + (i0 - 1) * mp * rts;
                    for (i64 B = p; B <= mp; B *= p) {
                        const i64 sub = B / p;
                        for (i64 blk = 0; blk < mp; blk += B)
                            for (i64 off = 0; off < sub; ++off) {
                                if (sub > 1)
                                    for (int fq = 1; fq < p; ++fq) {
                                        i64 *v = &x[(rb + (blk + fq * sub + off) * rts) * L];
                                        *v = mulmod(*v, f->wpe_inv[j][(p * (mp / B) * fq * off) % (mp * p)], q);
                                    }
                                dense(x, L, rb + (blk + off) * rts, sub * rts, p, f->dftp_inv[j], q);
                            }
                    }
                    if (mp > 1)
                        for (i64 j1 = 0; j1 < mp; ++j1)
                            x[(rb + j1 * rts) * L] = mulmod(x[(rb + j1 * rts) * L], f->wpe_inv[j][(i0 * j1) % (mp * p)], q);
                }
                if (p > 2)
                    for (i64 j1 = 0; j1 < mp; ++j1) dense(x, L, b0 + j1 * rts, mp * rts, p - 1, f->crtp_inv[j], q);
            }
        }
}

/* crt: Pow -> CRT, in place, AoS.  Returns 0, or -3 when the ring has no CRT basis. */
int orcg_crt(const orcg_ring *r, i64 *data) {
    if (!r->has_crt) return -3;
    for (int j = 0; j < r->L; ++j)
        for (int l = 0; l < r->nf; ++l) ppcrt(r, l, j, data + j, 0);
    return 0;
}
int orcg_crtinv(const orcg_ring *r, i64 *data) {
    if (!r->has_crt) return -3;
    for (int j = 0; j < r->L; ++j)
        for (int l = r->nf - 1; l >= 0; --l) ppcrt(r, l, j, data + j, 1);
    return 0;
}

/* ------------------------------------------------------------------ l / lInv, mulG / divG (Pow, Dec) */
/* generic walker: calls fn on every length-(p-1) column (stride `step`) of every odd prime factor */
typedef void (*colfn)(i64 *x, i64 L, i64 base, i64 step, int p, i64 q);

static void for_columns(const orcg_ring *r, i64 *data, colfn fn) {
    for (int j = 0; j < r->L; ++j)
        for (int l = 0; l < r->nf; ++l) {
            const gfact *f = &r->f[l];
            if (f->p == 2) continue;
            const i64 step = f->mp * f->rts, span = f->dim * f->rts;
            for (i64 o = 0; o < r->n; o += span)
                for (i64 in = 0; in < step; ++in) fn(data + j, r->L, o + in, step, f->p, r->q[j]);
        }
}
static inline i64 zadd(i64 a, i64 b, i64 q) { return q ? addmod(a, b, q) : a + b; }
static inline i64 zsub(i64 a, i64 b, i64 q) { return q ? submod(a, b, q) : a - b; }
static inline i64 zmul(i64 a, i64 b, i64 q) { return q ? mulmod(((a % q) + q) % q, ((b % q) + q) % q, q) : a * b; }

static void col_l(i64 *x, i64 L, i64 b, i64 s, int p, i64 q) {          /* prefix sums */
    for (int i = 1; i < p - 1; ++i) x[(b + i * s) * L] = zadd(x[(b + i * s) * L], x[(b + (i - 1) * s) * L], q);
}
static void col_linv(i64 *x, i64 L, i64 b, i64 s, int p, i64 q) {       /* differences */
    for (int i = p - 2; i >= 1; --i) x[(b + i * s) * L] = zsub(x[(b + i * s) * L], x[(b + (i - 1) * s) * L], q);
}
static void col_gpow(i64 *x, i64 L, i64 b, i64 s, int p, i64 q) {       /* (1 - zeta_p): out_i = a_i - a_{i-1} + a_{p-2} */
    const i64 last = x[(b + (p - 2) * s) * L];
    for (int i = p - 2; i >= 1; --i) x[(b + i * s) * L] = zadd(zsub(x[(b + i * s) * L], x[(b + (i - 1) * s) * L], q), last, q);
    x[b * L] = zadd(x[b * L], last, q);
}
static void col_gdec(i64 *x, i64 L, i64 b, i64 s, int p, i64 q) {       /* out_0 = 2 c_0 + sum_{i>=1} c_i, out_i = c_i - c_{i-1} */
    i64 sum = 0;
    for (int i = 0; i < p - 1; ++i) sum = zadd(sum, x[(b + i * s) * L], q);
    for (int i = p - 2; i >= 1; --i) x[(b + i * s) * L] = zsub(x[(b + i * s) * L], x[(b + (i - 1) * s) * L], q);
    x[b * L] = zadd(x[b * L], sum, q);
}
/* p times the inverse of col_gpow: p b_i = p A_i - (i+1) A_total, A = prefix sums */
static void col_ginvpow(i64 *x, i64 L, i64 b, i64 s, int p, i64 q) {
    i64 tot = 0;
    for (int i = 0; i < p - 1; ++i) tot = zadd(tot, x[(b + i * s) * L], q);
    i64 run = 0;
    for (int i = 0; i < p - 1; ++i) {
        run = zadd(run, x[(b + i * s) * L], q);
        x[(b + i * s) * L] = zsub(zmul(p, run, q), zmul(i + 1, tot, q), q);
    }
}
/* p times the inverse of col_gdec: p c_0 = y_0 - sum_{i>=1} Y_i, p c_i = p c_0 + p Y_i, Y_i = y_1 + .. + y_i */
static void col_ginvdec(i64 *x, i64 L, i64 b, i64 s, int p, i64 q) {
    i64 run = 0, acc = 0;
    for (int i = 1; i < p - 1; ++i) { run = zadd(run, x[(b + i * s) * L], q); acc = zadd(acc, run, q); }
    const i64 c0 = zsub(x[b * L], acc, q);
    run = 0;
    for (int i = 1; i < p - 1; ++i) {
        run = zadd(run, x[(b + i * s) * L], q);
        x[(b + i * s) * L] = zadd(c0, zmul(p, run, q), q);
    }
    x[b * L] = c0;
}

void orcg_l(const orcg_ring *r, i64 *d) { for_columns(r, d, col_l); }
void orcg_linv(const orcg_ring *r, i64 *d) { for_columns(r, d, col_linv); }
void orcg_mulg_pow(const orcg_ring *r, i64 *d) { for_columns(r, d, col_gpow); }
void orcg_mulg_dec(const orcg_ring *r, i64 *d) { for_columns(r, d, col_gdec); }

static i64 odd_rad(const orcg_ring *r) {
    i64 v = 1;
    for (int l = 0; l < r->nf; ++l) if (r->f[l].p != 2) v *= r->f[l].p;
    return v;
}
/* divide every entry by rad; 1 = success (Lol's Just), 0 = Nothing.  As lol-cpp: a Z_q limb multiplies by rad^-1
 * (fails when rad is not a unit), an integer limb checks divisibility.  On failure the data are unspecified. */
static int div_rad(const orcg_ring *r, i64 *data) {
    const i64 rad = odd_rad(r);
    if (rad == 1) return 1;
    for (int j = 0; j < r->L; ++j) {
        const i64 q = r->q[j];
        if (q) {
            const i64 inv = invmod(rad % q, q);
            if (!inv) return 0;
            for (i64 k = 0; k < r->n; ++k) data[k * r->L + j] = mulmod(data[k * r->L + j], inv, q);
        } else {
            for (i64 k = 0; k < r->n; ++k) {
                if (data[k * r->L + j] % rad) return 0;
                data[k * r->L + j] /= rad;
            }
        }
    }
    return 1;
}
int orcg_divg_pow(const orcg_ring *r, i64 *d) { for_columns(r, d, col_ginvpow); return div_rad(r, d); }
int orcg_divg_dec(const orcg_ring *r, i64 *d) { for_columns(r, d, col_ginvdec); return div_rad(r, d); }

int orcg_mulg_crt(const orcg_ring *r, i64 *d) {
    if (!r->has_crt) return -3;
    for (int j = 0; j < r->L; ++j) for (i64 k = 0; k < r->n; ++k) d[k * r->L + j] = mulmod(d[k * r->L + j], r->gcrt[j][k], r->q[j]);
    return 0;
}
int orcg_divg_crt(const orcg_ring *r, i64 *d) {
    if (!r->has_crt) return -3;
    for (int j = 0; j < r->L; ++j) for (i64 k = 0; k < r->n; ++k) d[k * r->L + j] = mulmod(d[k * r->L + j], r->gcrt_inv[j][k], r->q[j]);
    return 0;
}

/* ------------------------------------------------------------------ element-wise (any basis) */
void orcg_mul(const orcg_ring *r, i64 *a, const i64 *b) { const i64 N = r->n * r->L; for (i64 t = 0; t < N; ++t) a[t] = mulmod(a[t], b[t], r->q[t % r->L]); }
void orcg_add(const orcg_ring *r, i64 *a, const i64 *b) { const i64 N = r->n * r->L; for (i64 t = 0; t < N; ++t) a[t] = addmod(a[t], b[t], r->q[t % r->L]); }
void orcg_sub(const orcg_ring *r, i64 *a, const i64 *b) { const i64 N = r->n * r->L; for (i64 t = 0; t < N; ++t) a[t] = submod(a[t], b[t], r->q[t % r->L]); }
void orcg_scale(const orcg_ring *r, i64 *a, const i64 *s) { const i64 N = r->n * r->L; for (i64 t = 0; t < N; ++t) a[t] = mulmod(a[t], s[t % r->L] % r->q[t % r->L], r->q[t % r->L]); }

void orcg_decompose_triv(const orcg_ring *r, const i64 *c, i64 *const *digits) {
    for (int i = 0; i < r->L; ++i)
        for (i64 k = 0; k < r->n; ++k) {
            const i64 z = centred(c[k * r->L + i], r->q[i]);
            for (int j = 0; j < r->L; ++j) { i64 v = z % r->q[j]; digits[i][k * r->L + j] = v < 0 ? v + r->q[j] : v; }
        }
}

/* Rescale (a,b) -> b: drops limb 0, coefficient-wise on whatever basis the caller put the element in
 * (modSwitch: rescaleDec for c0, rescalePow for c1).  out: AoS n*(L-1). */
void orcg_rescale_drop0(const orcg_ring *r, const i64 *in, i64 *out) {
    const int L = r->L;
    i64 qinv[G_MAX_LIMBS];
    for (int j = 1; j < L; ++j) qinv[j] = invmod(r->q[0] % r->q[j], r->q[j]);
    for (i64 k = 0; k < r->n; ++k) {
        const i64 z = centred(in[k * L], r->q[0]);
        for (int j = 1; j < L; ++j) {
            i64 zr = z % r->q[j];
            if (zr < 0) zr += r->q[j];
            out[k * (L - 1) + (j - 1)] = mulmod(submod(in[k * L + j], zr, r->q[j]), qinv[j], r->q[j]);
        }
    }
}

/* ------------------------------------------------------------------ SymmSHE hot path, general index */
/* keySwitchQuadCirc hint (a * b), CRT basis in and out.  As oracle/lol_tensor.c's orc_ct_mul_relin_crt, plus the
 * mulG that SymmSHE's (*) applies to every product coefficient (the identity for a two-power index). */
int orcg_ct_mul_relin_crt(const orcg_ring *r, const i64 *const *hint, const i64 *a0, const i64 *a1, const i64 *b0,
                          const i64 *b1, const i64 *s_pre, i64 *out0, i64 *out1) {
    if (!r->has_crt) return -3;
    const i64 N = r->n * r->L;
    const int L = r->L;
    i64 *c2 = malloc(sizeof(i64) * N), *tmp = malloc(sizeof(i64) * N);
    i64 **dig = malloc(sizeof(i64 *) * L);
    for (int i = 0; i < L; ++i) dig[i] = malloc(sizeof(i64) * N);
    memcpy(out0, a0, sizeof(i64) * N); orcg_mul(r, out0, b0);
    memcpy(out1, a0, sizeof(i64) * N); orcg_mul(r, out1, b1);
    memcpy(tmp, a1, sizeof(i64) * N);  orcg_mul(r, tmp, b0);
    orcg_add(r, out1, tmp);
    memcpy(c2, a1, sizeof(i64) * N);   orcg_mul(r, c2, b1);
    orcg_mulg_crt(r, out0); orcg_mulg_crt(r, out1); orcg_mulg_crt(r, c2);
    orcg_scale(r, out0, s_pre); orcg_scale(r, out1, s_pre); orcg_scale(r, c2, s_pre);
    orcg_crtinv(r, c2);                                   /* decompose works on the Pow basis */
    orcg_decompose_triv(r, c2, dig);
    for (int i = 0; i < L; ++i) {
        orcg_crt(r, dig[i]);
        memcpy(tmp, dig[i], sizeof(i64) * N); orcg_mul(r, tmp, hint[2 * i]);     orcg_add(r, out0, tmp);
        memcpy(tmp, dig[i], sizeof(i64) * N); orcg_mul(r, tmp, hint[2 * i + 1]); orcg_add(r, out1, tmp);
    }
    for (int i = 0; i < L; ++i) free(dig[i]);
    free(dig); free(tmp); free(c2);
    return 0;
}

static inline u64 splitmix64(u64 x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
/* same synthetic-residue rule as orc_fill_uniform / alch_buf_fill_uniform */
void orcg_fill_uniform(const orcg_ring *r, i64 *data, u64 seed, u64 elem) {
    for (int j = 0; j < r->L; ++j)
        for (i64 k = 0; k < r->n; ++k)
            data[k * r->L + j] = (i64)(splitmix64(seed + ((elem * (u64)r->L + (u64)j) * (u64)r->n + (u64)k)) % (u64)r->q[j]);
}
