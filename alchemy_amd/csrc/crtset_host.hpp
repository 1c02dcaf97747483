// Host-side construction of Tensor `crtSetDec` (Lol: the relative mod-p CRT set of O_m' / O_m as coefficient vectors over
// F_p in the decoding basis).  Cyc's `crtSet` builds on it; the reference uses it through `decToCRT` (examples/Common.hs:65-75),
// the linear functions of the five ring switches of examples/HomomRLWR.hs and examples/Tunnel.hs.
//
// This is table construction (the counterpart of the twiddle tables of gen_host.hpp, which Lol also computes on the host),
// not a device compute path: finite-field arithmetic in GF(p^d), d = ord_{m'}(p), over index sets of Z_{m'}^*.
//
// Definition.  c_k in R_{m'} / p R_{m'} is the idempotent with sigma_i(c_k) = [i in I_k] (sigma_i: zeta_{m'} -> w^i, w a primitive
// m'-th root of unity in GF(p^d)); the I_k partition Z_{m'}^* into unions of <p>-cosets, exactly one coset above every <p>-coset
// of Z_m^*.  Order (this library's rule; Lol's own ordering of the set is not observable in the reference -- parity unpinned):
// the <p>-cosets of Z_{m'}^* are grouped by the <p>-coset of Z_m^* they reduce to, groups ordered by the smallest element of that
// coset, cosets inside a group by their smallest element; I_k = the union of every group's k-th coset.
// Coefficients on the decoding basis d = (mhat'/g') conj(p)^dual:   a_j = mhat'^-1 sum_{i in I} g'(w^i) w^(-i e(j)),
// e(j) the exponent of zeta_{m'} of the j-th powerful-basis element (oracle/model_gen.py checks the defining property).
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <string>
#include <vector>
#include "gen_host.hpp"

namespace alch {

// GF(p^d) = F_p[t] / (f), f the first monic irreducible polynomial of degree d in the order of its low coefficients read as a
// base-p number (constant term least significant); elements are d residues, constant term first.
struct GFpd {
    u32 p = 2;
    int d = 1;
    std::vector<u32> f;           // d + 1 coefficients, monic
    typedef std::vector<u32> El;

    static u64 ipow(u64 b, int e) { u64 r = 1; while (e-- > 0) r *= b; return r; }
    u32 inv_p(u32 a) const { return (u32)h_powmod(a % p, p - 2, p); }

    El zero() const { return El((size_t)d, 0); }
    El one() const { El e = zero(); e[0] = 1 % p; return e; }
    El add(const El& a, const El& b) const { El r((size_t)d); for (int i = 0; i < d; ++i) r[i] = (a[i] + b[i]) % p; return r; }
    El neg(const El& a) const { El r((size_t)d); for (int i = 0; i < d; ++i) r[i] = (p - a[i]) % p; return r; }
    bool is_one(const El& a) const { if (a[0] != 1 % p) return false; for (int i = 1; i < d; ++i) if (a[i]) return false; return true; }

    // a mod g over F_p (g monic or not), in place; polynomials as coefficient vectors without a fixed length
    void pmod(std::vector<u32>& a, const std::vector<u32>& g) const {
        const size_t dg = g.size() - 1;
        const u32 li = inv_p(g[dg]);
        while (!a.empty() && a.back() == 0) a.pop_back();
        while (a.size() > dg) {
            const u32 c = (u32)((u64)a.back() * li % p);
            const size_t sh = a.size() - 1 - dg;
            for (size_t i = 0; i <= dg; ++i) a[sh + i] = (u32)((a[sh + i] + (u64)(p - c) * g[i]) % p);
            while (!a.empty() && a.back() == 0) a.pop_back();
        }
    }
    std::vector<u32> pmul(const std::vector<u32>& a, const std::vector<u32>& b) const {
        if (a.empty() || b.empty()) return {};
        std::vector<u32> o(a.size() + b.size() - 1, 0);
        for (size_t i = 0; i < a.size(); ++i) if (a[i]) for (size_t j = 0; j < b.size(); ++j) o[i + j] = (u32)((o[i + j] + (u64)a[i] * b[j]) % p);
        return o;
    }
    std::vector<u32> ppow_x(u64 e, const std::vector<u32>& g) const {          // t^e mod g
        std::vector<u32> r{1}, b{0, 1};
        pmod(b, g);
        while (e) {
            if (e & 1) { r = pmul(r, b); pmod(r, g); }
            b = pmul(b, b); pmod(b, g);
            e >>= 1;
        }
        return r;
    }
    std::vector<u32> pgcd(std::vector<u32> a, std::vector<u32> b) const {
        while (!a.empty() && a.back() == 0) a.pop_back();
        while (!b.empty() && b.back() == 0) b.pop_back();
        while (!b.empty()) { pmod(a, b); std::swap(a, b); }
        return a;
    }
    bool irreducible(const std::vector<u32>& g) const {                        // Rabin's test
        const int dg = (int)g.size() - 1;
        auto minus_x = [&](std::vector<u32> h) { if (h.size() < 2) h.resize(2, 0); h[1] = (h[1] + p - 1) % p; while (!h.empty() && h.back() == 0) h.pop_back(); return h; };
        if (!minus_x(ppow_x(ipow(p, dg), g)).empty()) return false;
        for (int r = 2; r <= dg; ++r) {
            if (dg % r) continue;
            bool prime = true;
            for (int s = 2; s * s <= r; ++s) if (r % s == 0) prime = false;
            if (!prime) continue;
            std::vector<u32> h = minus_x(ppow_x(ipow(p, dg / r), g));
            if (h.empty() || pgcd(g, h).size() != 1) return false;
        }
        return true;
    }
    bool init(u32 p_, int d_) {
        p = p_; d = d_;
        if (d == 1) { f = {0, 1}; return true; }
        const u64 lim = ipow(p, d);
        for (u64 code = 0; code < lim; ++code) {
            std::vector<u32> g((size_t)d + 1);
            u64 c = code;
            for (int i = 0; i < d; ++i) { g[i] = (u32)(c % p); c /= p; }
            g[d] = 1;
            if (g[0] && irreducible(g)) { f = g; return true; }
        }
        return false;
    }
    El mul(const El& a, const El& b) const {
        std::vector<u32> r = pmul(a, b);
        pmod(r, f);
        r.resize((size_t)d, 0);
        return r;
    }
    El pow(El b, u64 e) const {
        El r = one();
        while (e) { if (e & 1) r = mul(r, b); b = mul(b, b); e >>= 1; }
        return r;
    }
    // the m-th root rule over GF(p^d): x^((p^d - 1)/m) for the first field element x (base-p count from 2) of order exactly m
    bool root_of_unity(u32 m, El& w) const {
        const u64 N = ipow(p, d) - 1;
        if (N % m) return false;
        std::vector<u32> rs;
        for (u32 t = m, r = 2; t > 1; ++r) if (t % r == 0) { rs.push_back(r); while (t % r == 0) t /= r; }
        for (u64 code = 2; code <= N; ++code) {
            El x((size_t)d);
            u64 c = code;
            for (int i = 0; i < d; ++i) { x[i] = (u32)(c % p); c /= p; }
            w = pow(x, N / m);
            bool ok = true;
            for (u32 r : rs) if (is_one(pow(w, m / r))) { ok = false; break; }
            if (ok) return true;
        }
        return m == 1 ? (w = one(), true) : false;
    }
};

inline u32 mult_order(u32 p, u32 m) {
    if (m == 1) return 1;
    u32 d = 1;
    u64 x = p % m;
    while (x != 1) { x = x * p % m; ++d; }
    return d;
}

// the index sets I_k of the rule above
inline bool crt_set_cosets(u32 m, u32 mb, u32 p, std::vector<std::vector<std::vector<u32>>>& sets, std::string& err) {
    auto gcd = [](u32 a, u32 b) { while (b) { u32 t = a % b; a = b; b = t; } return a; };
    if (m < 1 || mb % m || gcd(p % mb == 0 ? mb : p % mb, mb) != 1 || !h_is_prime(p)) { err = "crtSetDec needs m | m' and a prime p that does not divide m'"; return false; }
    std::vector<char> seen((size_t)mb + 1, 0);
    std::vector<std::vector<u32>> cosets;
    for (u32 u = (mb > 1 ? 1 : 0); u < std::max(mb, 1u); ++u) {
        if (mb > 1 && gcd(u, mb) != 1) continue;
        if (seen[u]) continue;
        std::vector<u32> c;
        u64 x = u;
        while (!seen[x]) { seen[x] = 1; c.push_back((u32)x); x = x * p % mb; }
        std::sort(c.begin(), c.end());
        cosets.push_back(c);
    }
    const u32 dm = mult_order(p, m);
    std::map<u32, std::vector<std::vector<u32>>> groups;
    for (auto& c : cosets) {
        u64 x = m > 1 ? c[0] % m : 0, key = x;
        for (u32 t = 0; t < dm; ++t) { key = std::min<u64>(key, x); x = m > 1 ? x * p % m : 0; }
        groups[(u32)key].push_back(c);
    }
    const size_t r = cosets.size() / groups.size();
    for (auto& g : groups) {
        if (g.second.size() != r) { err = "internal: uneven coset groups"; return false; }
        std::sort(g.second.begin(), g.second.end(), [](const std::vector<u32>& a, const std::vector<u32>& b) { return a[0] < b[0]; });
    }
    sets.assign(r, {});                                    // sets[k] = the <p>-cosets whose union is I_k
    for (size_t k = 0; k < r; ++k)
        for (auto& g : groups) sets[k].push_back(g.second[k]);
    return true;
}

// out: count vectors of phi(m') residues mod p (decoding basis of index m'), consecutive.
inline bool crt_set_dec(u32 m, u32 mb, u32 p, std::vector<int64_t>& out, size_t& count, std::string& err) {
    GenHost gb;
    if (!gen_factor(mb, gb)) { err = gb.error; return false; }
    std::vector<std::vector<std::vector<u32>>> sets;
    if (!crt_set_cosets(m, mb, p, sets, err)) return false;
    const int d = (int)mult_order(p, mb);
    if (d * std::log2((double)p) > 40) { err = "crtSetDec: GF(p^d) too large for this construction"; return false; }
    GFpd F;
    if (!F.init(p, d)) { err = "crtSetDec: no irreducible polynomial found"; return false; }
    GFpd::El w;
    if (!F.root_of_unity(mb, w)) { err = "crtSetDec: no primitive root of unity"; return false; }
    std::vector<GFpd::El> pw(mb);
    pw[0] = F.one();
    for (u32 t = 1; t < mb; ++t) pw[t] = F.mul(pw[t - 1], w);
    // exponent of zeta_{m'} of every powerful-basis element
    std::vector<u32> ex(gb.n);
    for (u32 j = 0; j < gb.n; ++j) {
        u64 e = 0;
        for (int l = 0; l < gb.nfact; ++l) {
            const GenFact& f = gb.fact[l];
            const u64 ml = (u64)f.mp * f.p;
            e += (u64)((j / f.rts) % f.dim) * (mb / ml);
        }
        ex[j] = (u32)(e % mb);
    }
    const u32 mh = mb % 2 == 0 ? mb / 2 : mb;
    const u32 inv_mhat = (u32)h_powmod(mh % p, p - 2, p);
    // x -> x^p permutes the terms y_i = g'(w^i) w^(-i e) of one <p>-coset (g'(w^(ip)) = g'(w^i)^p), and every <p>-coset of Z_{m'}^*
    // has d = ord_{m'}(p) elements, so a coset's sum is the absolute trace of its first term -- F_p-linear:
    // Tr(y) = sum_c y_c Tr(t^c).  One field product per coset instead of d.
    std::vector<u32> tr((size_t)d);
    for (int c = 0; c < d; ++c) {
        GFpd::El x = F.zero(), acc = F.zero();
        x[c] = 1;
        for (int k = 0; k < d; ++k) { acc = F.add(acc, x); x = F.pow(x, p); }
        for (int t = 1; t < d; ++t) if (acc[t]) { err = "internal: trace outside the prime field"; return false; }
        tr[c] = acc[0];
    }
    count = sets.size();
    out.assign(count * (size_t)gb.n, 0);
    for (size_t k = 0; k < count; ++k) {
        std::vector<u32> rep;
        std::vector<GFpd::El> gv;
        for (auto& coset : sets[k]) {
            if ((int)coset.size() != d) { err = "internal: coset size"; return false; }
            GFpd::El v = F.one();
            for (int l = 0; l < gb.nfact; ++l) {
                const u32 q = (u32)gb.fact[l].p;
                if (q == 2) continue;
                v = F.mul(v, F.add(F.one(), F.neg(pw[(u64)(mb / q) * coset[0] % mb])));
            }
            rep.push_back(coset[0]);
            gv.push_back(v);
        }
        for (u32 j = 0; j < gb.n; ++j) {
            u64 acc = 0;
            for (size_t t = 0; t < rep.size(); ++t) {
                const u64 e = ((u64)mb - (u64)rep[t] * ex[j] % mb) % mb;
                const GFpd::El y = F.mul(gv[t], pw[e]);
                for (int c = 0; c < d; ++c) acc += (u64)y[c] * tr[c];
            }
            out[k * (size_t)gb.n + j] = (int64_t)((acc % p) * inv_mhat % p);
        }
    }
    return true;
}

}  // namespace alch
