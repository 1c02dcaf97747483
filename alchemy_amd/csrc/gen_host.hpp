// Host-side construction of the general-index pass plan and its tables (kernel_gen.hpp runs it).
// Lol computes the corresponding twiddles in Haskell and hands lol-cpp raw pointers on every call; here they are
// built once per ring from the documented root rule and stay device-resident.
#pragma once
#include <algorithm>
#include <string>
#include <vector>
#include "kernel_gen.hpp"
#include "ring_host.hpp"

namespace alch {

struct GenHost {
    u32 m = 0, n = 0;
    int nfact = 0, npass = 0;
    GenFact fact[GEN_MAXFACT];
    GenPass pass[GEN_MAXPASS];
    u32 block_words = 0;      // words of one limb's table block
    u32 rad = 1;              // product of the odd primes of m
    std::string error;
};

inline u64 h_invmod(u64 a, u64 q) { return h_powmod(a % q, q - 2, q); }      // q prime

inline u32 h_digitrev(u32 x, u32 p, int digits) {
    u32 r = 0;
    for (int t = 0; t < digits; ++t) { r = r * p + x % p; x /= p; }
    return r;
}

// Factor m: prime powers ascending, per-axis dimension and stride of the mixed-radix index (first factor outermost), odd radical.
inline bool gen_factor(u32 m, GenHost& g) {
    g = GenHost();
    g.m = m;
    u32 rem = m;
    for (u32 p = 2; (u64)p * p <= rem; p += (p == 2 ? 1 : 2)) {
        if (rem % p) continue;
        if (g.nfact == GEN_MAXFACT) { g.error = "too many prime factors"; return false; }
        GenFact& f = g.fact[g.nfact++];
        f.p = (int)p; f.e = 0; f.mp = 1;
        while (rem % p == 0) { rem /= p; if (f.e++) f.mp *= p; }
    }
    if (rem > 1) {
        if (g.nfact == GEN_MAXFACT) { g.error = "too many prime factors"; return false; }
        GenFact& f = g.fact[g.nfact++];
        f.p = (int)rem; f.e = 1; f.mp = 1;
    }
    u64 n = 1;
    for (int l = 0; l < g.nfact; ++l) {
        GenFact& f = g.fact[l];
        if (f.p != 2) g.rad *= (u32)f.p;
        f.dim = (u32)(f.p - 1) * f.mp;
        n *= f.dim;
    }
    if (n > 0x7fffffffull) { g.error = "ring dimension too large"; return false; }
    g.n = (u32)n;
    u32 s = g.n;
    for (int l = 0; l < g.nfact; ++l) { s /= g.fact[l].dim; g.fact[l].rts = s; }
    return true;
}

// Factor m and lay out the passes.  Returns false (with g.error) for an index this backend does not serve.
inline bool gen_plan(u32 m, GenHost& g) {
    if (!gen_factor(m, g)) return false;
    for (int l = 0; l < g.nfact; ++l)
        if (g.fact[l].p > 13) { g.error = "odd prime factors of the index must be <= 13 (the reference's indices use 3, 5, 7, 13)"; return false; }
    if (g.n > 65535) { g.error = "ring dimension too large (the pass engine indexes with 16-bit quantities)"; return false; }
    u32 off = 0;
    auto rcp = [](u32 d) { return d <= 1 ? 0u : (u32)(((u64)1 << 32) / d) + 1u; };
    // mat_words: size of the pass's own table; tw_off: 0xffffffff = no twiddles, 0xfffffffe = allocate f.dim words, else shared
    auto push = [&](int kind, int r, int aux, u32 stride, const GenFact& f, u32 mat_words, u32 tw_off) {
        if (g.npass == GEN_MAXPASS) { g.error = "too many passes"; return false; }
        GenPass& P = g.pass[g.npass++];
        P.kind = kind; P.r = r; P.aux = aux; P.stride = stride; P.axis_stride = f.rts; P.axis_len = f.dim;
        P.mat_off = off; off += mat_words;
        if (tw_off == 0xfffffffeu) { P.tw_off = off; off += f.dim; } else P.tw_off = tw_off;
        P.rcp_stride = rcp(P.stride); P.rcp_axis_stride = rcp(P.axis_stride); P.rcp_axis_len = rcp(P.axis_len);
        return true;
    };
    for (int l = 0; l < g.nfact; ++l) {
        const GenFact& f = g.fact[l];
        if (f.p == 2) {
            int a = 0;
            while ((1u << a) < f.dim) ++a;                       // dim = 2^a stages
            if (a == 0) continue;
            const u32 tw = off;                                  // one table tw[k] = psi^brev(k), k < dim, for every block
            off += f.dim;
            for (int s0 = 0; s0 < a;) {
                const int K = std::min(3, a - s0);
                const u32 t_last = f.dim >> (s0 + K);
                if (!push(GK_R2BLOCK, 1 << K, s0, t_last * f.rts, f, 0, tw)) return false;
                s0 += K;
            }
        } else {
            const u32 h = (u32)(f.p - 1) / 2, sym_words = 2 * h * h + (u32)f.p;
            if (!push(GK_SYM_CRT, f.p - 1, 0, f.mp * f.rts, f, sym_words, 0xffffffffu)) return false;
            for (u32 B = f.mp; B > 1; B /= (u32)f.p)
                if (!push(GK_SYM_DFT, f.p, 0, (B / (u32)f.p) * f.rts, f, sym_words, 0xfffffffeu)) return false;
        }
    }
    g.block_words = off ? off : 1;
    return true;
}

// Tables of one limb: forward and inverse blocks (plain residues; the caller converts to Montgomery form), the CRT
// image of g and its inverse, and crtInv's closing scalar.
inline bool gen_tables(const GenHost& g, u64 q, std::vector<u64>& fwd, std::vector<u64>& inv, std::vector<u64>& gcrt,
                       std::vector<u64>& gcrt_inv, u64& iscale) {
    const u32 m = g.m;
    const u64 wm = h_powmod(h_smallest_generator(q), (q - 1) / m, q);        // the root rule: omega_m
    fwd.assign(g.block_words, 0);
    inv.assign(g.block_words, 0);
    iscale = 1;
    int ps = 0;
    for (int l = 0; l < g.nfact; ++l) {
        const GenFact& f = g.fact[l];
        const u32 p = (u32)f.p, pe = f.mp * p;
        const u64 w = h_powmod(wm, m / pe, q), wi = h_invmod(w, q);        // omega_{p^e}
        std::vector<u64> pw(pe), pwi(pe);
        { u64 a = 1, b = 1; for (u32 t = 0; t < pe; ++t) { pw[t] = a; pwi[t] = b; a = h_mulmod(a, w, q); b = h_mulmod(b, wi, q); } }
        if (p == 2) {
            // tw[k] = psi^brev(k) on log2(dim) bits, psi = omega_{2^e}: stage s, group gidx uses tw[2^s + gidx]
            int lg = 0;
            while ((1u << lg) < f.dim) ++lg;
            if (lg == 0) continue;
            const u32 tw_off = g.pass[ps].tw_off;
            for (u32 k = 0; k < f.dim; ++k) {
                const u32 e = h_brev(k, lg);
                fwd[tw_off + k] = pw[e % pe];
                inv[tw_off + k] = pwi[e % pe];
            }
            for (int st = 0; st < lg; ++st) iscale = h_mulmod(iscale, h_invmod(2, q), q);
            while (ps < g.npass && g.pass[ps].kind == GK_R2BLOCK && g.pass[ps].tw_off == tw_off) ++ps;
            continue;
        }
        // symmetric tables of omega_p = omega_{p^e}^(m'):  a_ij = (w^ij + w^-ij)/2, b_ij = (w^ij - w^-ij)/2  (i, j = 1..h)
        const u32 h = (p - 1) / 2;
        const u64 half = h_invmod(2, q), pinv = h_invmod(p, q);
        auto wp = [&](u64 e) { return pw[(size_t)(e % p) * f.mp]; };
        auto wpi = [&](u64 e) { return pwi[(size_t)(e % p) * f.mp]; };
        auto fill_sym = [&](const GenPass& P) {
            for (u32 i = 1; i <= h; ++i)
                for (u32 j = 1; j <= h; ++j) {
                    const u64 c = h_mulmod((wp((u64)i * j) + wpi((u64)i * j)) % q, half, q);
                    const u64 sn = h_mulmod((wp((u64)i * j) + q - wpi((u64)i * j)) % q, half, q);
                    fwd[P.mat_off + (i - 1) * h + (j - 1)] = c;
                    fwd[P.mat_off + h * h + (i - 1) * h + (j - 1)] = sn;
                    // inverse: w -> w^-1 (the cosine part is even, the sine part changes sign) and 1/p
                    inv[P.mat_off + (i - 1) * h + (j - 1)] = h_mulmod(c, pinv, q);
                    inv[P.mat_off + h * h + (i - 1) * h + (j - 1)] = h_mulmod((q - sn) % q, pinv, q);
                }
            for (u32 i = 1; i < p; ++i) {                       // inverse CRT_p: y_0 = sum_i y_i (-w^i)
                fwd[P.mat_off + 2 * h * h + (i - 1)] = 0;
                inv[P.mat_off + 2 * h * h + (i - 1)] = (q - wp(i)) % q;
            }
            fwd[P.mat_off + 2 * h * h + (p - 1)] = 1;
            inv[P.mat_off + 2 * h * h + (p - 1)] = pinv;
        };
        fill_sym(g.pass[ps++]);                                  // CRT_p
        bool first = true;
        u32 Bprev = 0;
        for (u32 B = f.mp; B > 1; B /= p, ++ps) {
            const GenPass& P = g.pass[ps];
            fill_sym(P);
            // twiddles in front of this stage, by axis position a = (i0 - 1) m' + j1:
            //   first stage: T = omega_{p^e}^(i0 j1);  later stages: the previous stage's omega_{Bprev}^(fq off)
            for (u32 i0 = 1; i0 < p; ++i0)
                for (u32 j1 = 0; j1 < f.mp; ++j1) {
                    u32 e;
                    if (first) e = (u32)(((u64)i0 * j1) % pe);
                    else {
                        const u32 sub = Bprev / p, within = j1 % Bprev;
                        const u32 fq = within / sub, offv = within % sub;
                        e = (u32)(((u64)p * (f.mp / Bprev) * fq * offv) % pe);
                    }
                    fwd[P.tw_off + (i0 - 1) * f.mp + j1] = pw[e];
                    inv[P.tw_off + (i0 - 1) * f.mp + j1] = pwi[e];
                }
            first = false;
            Bprev = B;
        }
    }
    // CRT image of g = prod_{odd p | m} (1 - zeta_p): slot s holds prod (1 - omega_p^u(s))
    gcrt.assign(g.n, 1);
    gcrt_inv.assign(g.n, 1);
    for (u32 sl = 0; sl < g.n; ++sl) {
        u64 u = 0, mod = 1;
        for (int l = 0; l < g.nfact; ++l) {
            const GenFact& f = g.fact[l];
            const u64 ml = (u64)f.mp * f.p;
            const u32 s = (sl / f.rts) % f.dim;
            const u64 i0 = s / f.mp + 1, i1 = h_digitrev(s % f.mp, (u32)f.p, f.e - 1);
            const u64 ul = (i0 + (u64)f.p * i1) % ml;
            u64 t = ul;
            if (mod > 1) {
                // t = (ul - u) / mod  (mod ml); mod and ml are coprime
                u64 minv = 1;
                for (u64 c = 1; c < ml; ++c) if ((mod % ml) * c % ml == 1) { minv = c; break; }
                t = ((ul + ml - u % ml) % ml) * minv % ml;
            }
            u += mod * t;
            mod *= ml;
        }
        u %= m;
        u64 v = 1;
        for (int l = 0; l < g.nfact; ++l) {
            const u32 p = (u32)g.fact[l].p;
            if (p == 2) continue;
            const u64 wp = h_powmod(wm, ((u64)(m / p) * u) % m, q);
            v = h_mulmod(v, (1 + q - wp) % q, q);
        }
        gcrt[sl] = v;
        gcrt_inv[sl] = h_invmod(v, q);
    }
    return true;
}

// Tunnel index table (SymmSHE.tunnel): for the relative index i of R'/E' and every Pow/Dec position k of S',
// table[i * n_s + k] = the position in R' whose coefficient lands there -- coeffs (R' -> E', relative index i) followed by
// embedPow (E' -> S') -- or -1 where embedPow leaves a zero.  ep = gcd(r', s'); relative indices in mixed radix, first
// prime outermost (the order of oracle/model_gen.py coeffs_indices).
// table_e (optional): the same gather BEFORE embedPow, [d_rel][n_e] positions in R' of the Pow coefficients of E'-coefficient i.
// slot_e (optional): for every CRT slot of S' the CRT slot of E' that holds the same value for an element of E' embedded into S'
//   (Tensor embedCRT): crt_S'(embed d)[s] = crt_E'(d)[slot_e[s]].  With the slot rule of include/alchemy_hip.h (first factor
//   outermost, within a factor s = (i0 - 1) p^(e-1) + digitrev(i1) for the unit i0 + p i1) reducing a unit mod p^ee keeps i0 and
//   the low digits of i1, i.e. the HIGH digits of the reversed index: the factor's slot index is divided by p^(es - ee); a prime
//   that does not divide e' drops out.  Checked against the by-definition model in tests/test_oracle_tunnel.py.
inline bool gen_tunnel_table(const GenHost& ep, const GenHost& rp, const GenHost& sp, u32& d_rel, std::vector<int32_t>& table,
                             u32& linv_skip_mask, std::vector<int32_t>* table_e = nullptr, std::vector<u32>* slot_e = nullptr) {
    if (rp.m % ep.m || sp.m % ep.m) return false;
    auto expo = [](const GenHost& g, int p) { for (int l = 0; l < g.nfact; ++l) if (g.fact[l].p == p) return g.fact[l].e; return 0; };
    auto ipow = [](u32 b, int e) { u32 r = 1; while (e-- > 0) r *= b; return r; };
    // relative dimensions per factor of r'
    u32 rel_dim[GEN_MAXFACT];
    d_rel = 1;
    linv_skip_mask = 0;
    for (int l = 0; l < rp.nfact; ++l) {
        const int p = rp.fact[l].p, er = rp.fact[l].e, ee = expo(ep, p);
        rel_dim[l] = ee ? ipow((u32)p, er - ee) : rp.fact[l].dim;
        d_rel *= rel_dim[l];
        // lInv on R' followed by l on every E'-coefficient = lInv over the primes that do NOT divide e' only
        if (ee) linv_skip_mask |= 1u << l;
    }
    if ((u64)d_rel * ep.n != rp.n) return false;
    table.assign((size_t)d_rel * sp.n, -1);
    if (table_e) table_e->assign((size_t)d_rel * ep.n, -1);
    if (slot_e) {
        slot_e->assign(sp.n, 0);
        for (u32 s = 0; s < sp.n; ++s) {
            u32 se = 0;
            for (int l = 0; l < sp.nfact; ++l) {
                const int p = sp.fact[l].p, es = sp.fact[l].e, ee = expo(ep, p);
                if (!ee) continue;
                const u32 sf = (s / sp.fact[l].rts) % sp.fact[l].dim;
                for (int le = 0; le < ep.nfact; ++le) if (ep.fact[le].p == p) se += (sf / ipow((u32)p, es - ee)) * ep.fact[le].rts;
            }
            (*slot_e)[s] = se;
        }
    }
    for (u32 i = 0; i < d_rel; ++i) {
        u32 rel[GEN_MAXFACT], t = i;
        for (int l = rp.nfact - 1; l >= 0; --l) { rel[l] = t % rel_dim[l]; t /= rel_dim[l]; }
        for (u32 j = 0; j < ep.n; ++j) {
            u32 posr = 0, poss = 0;
            for (int l = 0; l < rp.nfact; ++l) {
                const int p = rp.fact[l].p, er = rp.fact[l].e, ee = expo(ep, p);
                u32 jp = 0;
                if (ee) for (int le = 0; le < ep.nfact; ++le) if (ep.fact[le].p == p) jp = (j / ep.fact[le].rts) % ep.fact[le].dim;
                posr += (ee ? rel[l] + ipow((u32)p, er - ee) * jp : rel[l]) * rp.fact[l].rts;
            }
            for (int l = 0; l < sp.nfact; ++l) {
                const int p = sp.fact[l].p, es = sp.fact[l].e, ee = expo(ep, p);
                if (!ee) continue;
                u32 jp = 0;
                for (int le = 0; le < ep.nfact; ++le) if (ep.fact[le].p == p) jp = (j / ep.fact[le].rts) % ep.fact[le].dim;
                poss += jp * ipow((u32)p, es - ee) * sp.fact[l].rts;
            }
            table[(size_t)i * sp.n + poss] = (int32_t)posr;
            if (table_e) (*table_e)[(size_t)i * ep.n + j] = (int32_t)posr;
        }
    }
    return true;
}

// Index tables of the Tensor methods between two indices m | m' (Lol: embedPow / embedDec / twacePowDec / coeffs / powBasisPow and
// crtExtFuncs = (twaceCRT, embedCRT)).  small = index m, big = index m'.
//   pow_pos[j]   (n_small)          position in the big ring's Pow (or Dec, for twace / coeffs) vector of the small ring's basis element j
//                                   (oracle/model_gen.py embed_indices)
//   coeffs[i][j] ([d_rel][n_small]) position in the big ring of coefficient j of the i-th E-coefficient w.r.t. the relative
//                                   powerful / decoding basis (coeffs_indices; relative indices mixed radix, first prime outermost);
//                                   coeffs[0] == pow_pos
//   slot_small[s] (n_big)           CRT slot of the small ring whose unit is the reduction mod m of the unit of the big ring's slot s:
//                                   crt_big(embed x)[s] = crt_small(x)[slot_small[s]]   (embedCRT), and twaceCRT sums over the fibres
//                                   {s : slot_small[s] = t}.  With the slot rule of include/alchemy_hip.h reducing a unit mod p^es keeps
//                                   i0 and the low digits of i1 = the HIGH digits of the reversed index: the factor's slot index is
//                                   divided by p^(eb - es); a prime that does not divide m drops out.
inline bool gen_ext_tables(const GenHost& sm, const GenHost& bg, u32& d_rel, std::vector<int32_t>& pow_pos, std::vector<int32_t>& coeffs,
                           std::vector<u32>& slot_small) {
    if (sm.m == 0 || bg.m % sm.m) return false;
    auto expo = [](const GenHost& g, int p) { for (int l = 0; l < g.nfact; ++l) if (g.fact[l].p == p) return g.fact[l].e; return 0; };
    auto ipow = [](u32 b, int e) { u32 r = 1; while (e-- > 0) r *= b; return r; };
    auto comp = [](const GenHost& g, int p, u32 lin) -> u32 {            // index of the factor with prime p inside the linear index
        for (int l = 0; l < g.nfact; ++l) if (g.fact[l].p == p) return (lin / g.fact[l].rts) % g.fact[l].dim;
        return 0;
    };
    u32 rel_dim[GEN_MAXFACT];
    d_rel = 1;
    for (int l = 0; l < bg.nfact; ++l) {
        const int p = bg.fact[l].p, es = expo(sm, p);
        rel_dim[l] = es ? ipow((u32)p, bg.fact[l].e - es) : bg.fact[l].dim;
        d_rel *= rel_dim[l];
    }
    if ((u64)d_rel * sm.n != bg.n) return false;
    coeffs.assign((size_t)d_rel * sm.n, -1);
    for (u32 i = 0; i < d_rel; ++i) {
        u32 rel[GEN_MAXFACT], t = i;
        for (int l = bg.nfact - 1; l >= 0; --l) { rel[l] = t % rel_dim[l]; t /= rel_dim[l]; }
        for (u32 j = 0; j < sm.n; ++j) {
            u32 pos = 0;
            for (int l = 0; l < bg.nfact; ++l) {
                const int p = bg.fact[l].p, es = expo(sm, p);
                pos += (es ? rel[l] + ipow((u32)p, bg.fact[l].e - es) * comp(sm, p, j) : rel[l]) * bg.fact[l].rts;
            }
            coeffs[(size_t)i * sm.n + j] = (int32_t)pos;
        }
    }
    pow_pos.assign(coeffs.begin(), coeffs.begin() + sm.n);
    slot_small.assign(bg.n, 0);
    for (u32 s = 0; s < bg.n; ++s) {
        u32 t = 0;
        for (int l = 0; l < bg.nfact; ++l) {
            const int p = bg.fact[l].p, es = expo(sm, p);
            if (!es) continue;
            const u32 sf = (s / bg.fact[l].rts) % bg.fact[l].dim;
            for (int ls = 0; ls < sm.nfact; ++ls) if (sm.fact[ls].p == p) t += (sf / ipow((u32)p, bg.fact[l].e - es)) * sm.fact[ls].rts;
        }
        slot_small[s] = t;
    }
    return true;
}

}  // namespace alch
