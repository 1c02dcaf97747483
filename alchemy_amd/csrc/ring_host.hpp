// Host-side number theory for ring contexts: primality, the root rule, twiddle tables.
// (Lol computes these on the Haskell side and hands lol-cpp raw twiddle pointers on every call; here
// they are built once per ring and kept device-resident.)
#pragma once
#include <vector>
#include "modarith.hpp"

namespace alch {

inline bool h_is_prime(u64 n) {
    static const u64 bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return false;
    for (u64 p : bases) if (n % p == 0) return n == p;
    u64 d = n - 1;
    int s = 0;
    while ((d & 1) == 0) { d >>= 1; ++s; }
    for (u64 a : bases) {
        u64 x = h_powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int r = 1; r < s; ++r) {
            x = h_mulmod(x, x, n);
            if (x == n - 1) { comp = false; break; }
        }
        if (comp) return false;
    }
    return true;
}

// Root rule, step 1: the smallest generator of Z_q^*.
inline u64 h_smallest_generator(u64 q) {
    std::vector<u64> fac;
    u64 m = q - 1;
    for (u64 p = 2; p * p <= m; p += (p == 2 ? 1 : 2)) {
        if (m % p == 0) {
            fac.push_back(p);
            while (m % p == 0) m /= p;
        }
    }
    if (m > 1) fac.push_back(m);
    for (u64 g = 2;; ++g) {
        bool ok = true;
        for (u64 f : fac) if (h_powmod(g, (q - 1) / f, q) == 1) { ok = false; break; }
        if (ok) return g;
    }
}

// Root rule, step 2: psi = g^((q-1)/m), a primitive m-th root of unity (m = 2n).
inline u64 h_root(u64 q, u64 m) { return h_powmod(h_smallest_generator(q), (q - 1) / m, q); }

inline u32 h_brev(u32 k, int bits) {
    u32 r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (k & 1); k >>= 1; }
    return r;
}

// tw[k] = psi^brev_logn(k) and its inverse, both in Montgomery form (x * R mod q), k in [0, n).
template <typename W>
inline void h_build_twiddles(u64 q, u64 psi, int logn, std::vector<W>& fwd, std::vector<W>& inv) {
    const u64 n = 1ull << logn;
    const int bits = 8 * (int)sizeof(W);
    const u64 r1 = h_powmod(2, (u64)bits, q);
    const u64 ipsi = h_powmod(psi, q - 2, q);
    std::vector<u64> pw(n), ipw(n);
    u64 a = 1, b = 1;
    for (u64 i = 0; i < n; ++i) {
        pw[i] = a;
        ipw[i] = b;
        a = h_mulmod(a, psi, q);
        b = h_mulmod(b, ipsi, q);
    }
    fwd.resize(n);
    inv.resize(n);
    for (u64 k = 0; k < n; ++k) {
        u32 e = h_brev((u32)k, logn);
        fwd[k] = (W)h_mulmod(pw[e], r1, q);
        inv[k] = (W)h_mulmod(ipw[e], r1, q);
    }
}

}  // namespace alch
