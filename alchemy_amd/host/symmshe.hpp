// Host-side mirror (C++, header-only) of the Lol / lol-apps interface ALCHEMY's evaluator calls on this
// path, written above the C ABI of include/alchemy_hip.h.  The reference's toolchain (GHC) is absent, so
// this C++ layer stands where `instance Tensor GT` + Lol's Cyc + SymmSHE would stand; names, argument
// meaning and failure behaviour follow the reference's call sites:
//
//   Cyc            ring element with basis tracking (Pow / CRT), like Lol's Cyc t m' zq: products bring both
//                  operands to the CRT basis (Tensor crt) and multiply pointwise; `adviseCRT`, `advisePow`.
//   CT             SymmSHE ciphertext  CT enc k l c        (Crypto/Alchemy/Interpreter/PT2CT.hs:251-254)
//   toMSD / toLSD  encoding changes (scalars l, c)
//   operator*      SymmSHE (*)  = E's mul_                 (Crypto/Alchemy/Interpreter/Eval.hs:65-67)
//   operator+      SymmSHE (+)  = E's add_                 (Eval.hs:58-60)
//   keySwitchQuadCirc                                      (Eval.hs:133)
//   modSwitchDrop0 one limb of modSwitch (rescale (a,b)->b)(Eval.hs:130)
//   modSwitchAdd0  the other direction (rescale b->(a,b))   (Eval.hs:130; PT2CT.hs:177)
//   genSK, ksQuadCircHint, encrypt, decrypt                (KeysHints.hs:93-96,101-113; PT2CT.hs:84-99)
//   mulRelinBatch  the fused device path for PT2CT's  keySwitchQuad_ hint $: (x *: y)  (PT2CT.hs:172-177)
//   mulFullBatch   the fused device path for the whole mul_ with the longer hint modulus (PT2CT.hs:139,160-177)
//
// Everything numeric goes through the C ABI (device kernels); this file only sequences calls and keeps the
// (enc, k, l) metadata.  Errors surface as std::runtime_error carrying alch_last_error().
// REPLAY AND TEST CODE, NOT A KEY GENERATOR: keys, errors and hints are drawn from a caller-supplied std::mt19937_64 with `rng() % q`
// for uniform residues (reproducible, not a CSPRNG, modulo-biased); a production host keeps Lol's sampling.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/alchemy_hip.h"

namespace alchemy {

inline void check(int rc, const char* what) {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + alch_last_error());
}

inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((unsigned __int128)a * b) % q); }
inline uint64_t powmod(uint64_t b, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    b %= q;
    for (; e; e >>= 1, b = mulmod(b, b, q)) if (e & 1) r = mulmod(r, b, q);
    return r;
}
inline uint64_t invmod(uint64_t a, uint64_t q) { return powmod(a % q, q - 2, q); }      // q prime

// ---- ring context (one `Cyc t m' zq` type) -------------------------------------------------------------
class Ring {
public:
    Ring(uint32_t m, std::vector<uint64_t> qs) : m_(m), n_(m / 2), qs_(std::move(qs)) {
        check(alch_ring_create(m, (int)qs_.size(), qs_.data(), &h_), "alch_ring_create");
    }
    ~Ring() { alch_ring_destroy(h_); }
    Ring(const Ring&) = delete;
    Ring& operator=(const Ring&) = delete;
    alch_ring* handle() const { return h_; }
    uint32_t n() const { return n_; }
    int L() const { return (int)qs_.size(); }
    const std::vector<uint64_t>& qs() const { return qs_; }
    size_t words() const { return (size_t)n_ * qs_.size(); }

private:
    uint32_t m_, n_;
    std::vector<uint64_t> qs_;
    alch_ring* h_ = nullptr;
};

enum class Basis { Pow, CRT };

// ---- ring element ---------------------------------------------------------------------------------------
// Host image in Lol's layout (int64, coefficient-major / limb-minor); every operation is one C-ABI call.
class Cyc {
public:
    Cyc(const Ring& r, Basis b) : r_(&r), basis_(b), v_(r.words(), 0) {}
    static Cyc fromIntegers(const Ring& r, const std::vector<int64_t>& z) {      // `reduce` an integer polynomial
        Cyc c(r, Basis::Pow);
        for (uint32_t k = 0; k < r.n(); ++k)
            for (int j = 0; j < r.L(); ++j) {
                int64_t q = (int64_t)r.qs()[j];
                int64_t v = z[k] % q;
                c.v_[(size_t)k * r.L() + j] = v < 0 ? v + q : v;
            }
        return c;
    }
    const Ring& ring() const { return *r_; }
    Basis basis() const { return basis_; }
    std::vector<int64_t>& data() { return v_; }
    const std::vector<int64_t>& data() const { return v_; }

    Cyc adviseCRT() const {                                   // Tensor crt
        if (basis_ == Basis::CRT) return *this;
        Cyc o = *this;
        check(alch_crt(r_->handle(), o.v_.data()), "alch_crt");
        o.basis_ = Basis::CRT;
        return o;
    }
    Cyc advisePow() const {                                   // Tensor crtInv
        if (basis_ == Basis::Pow) return *this;
        Cyc o = *this;
        check(alch_crtinv(r_->handle(), o.v_.data()), "alch_crtinv");
        o.basis_ = Basis::Pow;
        return o;
    }
    Cyc mulG() const {                                        // Tensor mulGPow / mulGCRT (identity: g = 1)
        Cyc o = *this;
        check(basis_ == Basis::Pow ? alch_mulg_pow(r_->handle(), o.v_.data()) : alch_mulg_crt(r_->handle(), o.v_.data()),
              "alch_mulg");
        return o;
    }
    friend Cyc operator*(const Cyc& a, const Cyc& b) {        // ring product: crt both, zipWithT (*)
        Cyc x = a.adviseCRT(), y = b.adviseCRT();
        check(alch_mul(x.r_->handle(), x.v_.data(), y.v_.data()), "alch_mul");
        return x;
    }
    friend Cyc operator+(const Cyc& a, const Cyc& b) {
        Cyc x = a, y = (b.basis_ == a.basis_) ? b : (a.basis_ == Basis::CRT ? b.adviseCRT() : b.advisePow());
        check(alch_add(x.r_->handle(), x.v_.data(), y.v_.data()), "alch_add");
        return x;
    }
    friend Cyc operator-(const Cyc& a, const Cyc& b) {
        Cyc x = a, y = (b.basis_ == a.basis_) ? b : (a.basis_ == Basis::CRT ? b.adviseCRT() : b.advisePow());
        check(alch_sub(x.r_->handle(), x.v_.data(), y.v_.data()), "alch_sub");
        return x;
    }
    Cyc scale(const std::vector<uint64_t>& s) const {         // scalarPow/scalarCRT product, per limb
        Cyc o = *this;
        check(alch_scale(r_->handle(), o.v_.data(), s.data()), "alch_scale");
        return o;
    }
    // Lol `decompose` (TrivGad) then `reduce <$>`: needs the Pow basis
    std::vector<Cyc> decomposeTrivReduced() const {
        Cyc p = advisePow();
        std::vector<int64_t> all((size_t)r_->L() * r_->words());
        check(alch_decompose_triv(r_->handle(), p.v_.data(), all.data()), "alch_decompose_triv");
        std::vector<Cyc> out;
        for (int i = 0; i < r_->L(); ++i) {
            Cyc d(*r_, Basis::Pow);
            std::copy(all.begin() + (size_t)i * r_->words(), all.begin() + (size_t)(i + 1) * r_->words(), d.v_.begin());
            out.push_back(std::move(d));
        }
        return out;
    }

private:
    const Ring* r_;
    Basis basis_;
    std::vector<int64_t> v_;
};

// ---- SymmSHE ---------------------------------------------------------------------------------------------
enum class Encoding { MSD, LSD };

struct CT {
    Encoding enc;
    int k;                 // accumulated power of g (numerically inert for a two-power index)
    uint64_t l;            // accumulated Z_p scalar
    uint64_t p;            // plaintext modulus
    std::vector<Cyc> c;    // polynomial in the secret key over R'_q
};

struct SK { std::vector<int64_t> s; double r; };
struct KSQuadCircHint { std::vector<std::pair<Cyc, Cyc>> h; };     // TrivGad: one (h0, h1) per limb, CRT basis

inline uint64_t qprod_mod(const Ring& r, uint64_t p) {
    uint64_t v = 1 % p;
    for (uint64_t q : r.qs()) v = mulmod(v, q % p, p);
    return v;
}

inline CT toLSD(const CT& ct) {
    if (ct.enc == Encoding::LSD) return ct;
    const Ring& r = ct.c[0].ring();
    uint64_t negq = (ct.p - qprod_mod(r, ct.p)) % ct.p;
    std::vector<uint64_t> s;
    for (uint64_t q : r.qs()) s.push_back(ct.p % q);
    CT o{Encoding::LSD, ct.k, mulmod(ct.l, invmod(negq, ct.p), ct.p), ct.p, {}};
    for (const Cyc& x : ct.c) o.c.push_back(x.scale(s));
    return o;
}

inline CT toMSD(const CT& ct) {
    if (ct.enc == Encoding::MSD) return ct;
    const Ring& r = ct.c[0].ring();
    uint64_t negq = (ct.p - qprod_mod(r, ct.p)) % ct.p;
    std::vector<uint64_t> s;
    for (uint64_t q : r.qs()) s.push_back(invmod(ct.p % q, q));
    CT o{Encoding::MSD, ct.k, mulmod(ct.l, negq, ct.p), ct.p, {}};
    for (const Cyc& x : ct.c) o.c.push_back(x.scale(s));
    return o;
}

// SymmSHE (*): both to LSD, polynomial product in S, mulG on every coefficient, k1+k2+1, l1*l2.
inline CT operator*(const CT& a_, const CT& b_) {
    CT a = toLSD(a_), b = toLSD(b_);
    const Ring& r = a.c[0].ring();
    std::vector<Cyc> out(a.c.size() + b.c.size() - 1, Cyc(r, Basis::CRT));
    for (size_t i = 0; i < a.c.size(); ++i)
        for (size_t j = 0; j < b.c.size(); ++j) out[i + j] = out[i + j] + a.c[i] * b.c[j];
    for (Cyc& x : out) x = x.mulG();
    return CT{Encoding::LSD, a.k + b.k + 1, mulmod(a.l, b.l, a.p), a.p, out};
}

// SymmSHE (+) for operands that agree in k and l (all this path produces).
inline CT operator+(const CT& a_, const CT& b_) {
    if (a_.k != b_.k || a_.l != b_.l) throw std::runtime_error("CT (+): operands not aligned (k, l)");
    CT a = a_, b = b_;
    if (a.enc != b.enc) { a = toMSD(a); b = toMSD(b); }
    const Ring& r = a.c[0].ring();
    size_t m = std::max(a.c.size(), b.c.size());
    CT o{a.enc, a.k, a.l, a.p, {}};
    for (size_t i = 0; i < m; ++i) {
        Cyc x = i < a.c.size() ? a.c[i] : Cyc(r, Basis::CRT);
        Cyc y = i < b.c.size() ? b.c[i] : Cyc(r, Basis::CRT);
        o.c.push_back(x + y);
    }
    return o;
}

// keySwitchQuadCirc: toMSD; [c0,c1] + sum_i reduce(d_i) *>> hint_i,  d = decompose c2.
inline CT keySwitchQuadCirc(const KSQuadCircHint& hint, const CT& ct_) {
    CT ct = toMSD(ct_);
    if (ct.c.size() < 3) return ct;
    if (ct.c.size() != 3) throw std::runtime_error("keySwitchQuadCirc: ciphertext degree > 2");
    std::vector<Cyc> digs = ct.c[2].decomposeTrivReduced();
    Cyc c0 = ct.c[0].adviseCRT(), c1 = ct.c[1].adviseCRT();
    for (size_t i = 0; i < digs.size(); ++i) {
        c0 = c0 + digs[i] * hint.h[i].first;
        c1 = c1 + digs[i] * hint.h[i].second;
    }
    return CT{Encoding::MSD, ct.k, ct.l, ct.p, {c0, c1}};
}

// One limb of modSwitch: Rescale (a,b) -> b on every coefficient of S, in the Pow (= Dec) basis.
inline CT modSwitchDrop0(const CT& ct_, const Ring& dst) {
    CT ct = toMSD(ct_);
    const Ring& src = ct.c[0].ring();
    alch_buf *bs = nullptr, *bd = nullptr;
    check(alch_buf_alloc(src.handle(), ct.c.size(), &bs), "alch_buf_alloc");
    check(alch_buf_alloc(dst.handle(), ct.c.size(), &bd), "alch_buf_alloc");
    CT o{Encoding::MSD, ct.k, ct.l, ct.p, {}};
    for (size_t i = 0; i < ct.c.size(); ++i) {
        Cyc p = ct.c[i].advisePow();
        check(alch_buf_upload(bs, i, 1, p.data().data()), "alch_buf_upload");
    }
    check(alch_buf_rescale_drop0(bs, bd, ct.c.size()), "alch_buf_rescale_drop0");
    for (size_t i = 0; i < ct.c.size(); ++i) {
        Cyc x(dst, Basis::Pow);
        check(alch_buf_download(bd, i, 1, x.data().data()), "alch_buf_download");
        o.c.push_back(std::move(x));
    }
    alch_buf_free(bs);
    alch_buf_free(bd);
    return o;
}

// The other direction of modSwitch: Rescale b -> (a,b) into a ring with one more limb in front (the first
// modSwitch_ of PT2CT's mul_, PT2CT.hs:177: zq_in -> the hint's modulus).
inline CT modSwitchAdd0(const CT& ct_, const Ring& dst) {
    CT ct = toMSD(ct_);
    const Ring& src = ct.c[0].ring();
    alch_buf *bs = nullptr, *bd = nullptr;
    check(alch_buf_alloc(src.handle(), ct.c.size(), &bs), "alch_buf_alloc");
    check(alch_buf_alloc(dst.handle(), ct.c.size(), &bd), "alch_buf_alloc");
    CT o{Encoding::MSD, ct.k, ct.l, ct.p, {}};
    for (size_t i = 0; i < ct.c.size(); ++i) {
        Cyc p = ct.c[i].advisePow();
        check(alch_buf_upload(bs, i, 1, p.data().data()), "alch_buf_upload");
    }
    check(alch_buf_rescale_add0(bs, bd, ct.c.size()), "alch_buf_rescale_add0");
    for (size_t i = 0; i < ct.c.size(); ++i) {
        Cyc x(dst, Basis::Pow);
        check(alch_buf_download(bd, i, 1, x.data().data()), "alch_buf_download");
        o.c.push_back(std::move(x));
    }
    alch_buf_free(bs);
    alch_buf_free(bd);
    return o;
}

// ---- keys, hints, encryption (setup time; KeysHints.hs) ----------------------------------------------------
inline std::vector<int64_t> gaussianPoly(uint32_t n, double r, std::mt19937_64& rng) {
    std::normal_distribution<double> g(0.0, r / std::sqrt(2.0 * M_PI));
    std::vector<int64_t> z(n);
    for (auto& v : z) v = (int64_t)std::llround(g(rng));
    return z;
}

inline SK genSK(const Ring& r, double rparam, std::mt19937_64& rng) { return SK{gaussianPoly(r.n(), rparam, rng), rparam}; }

inline Cyc uniformCRT(const Ring& r, std::mt19937_64& rng) {
    Cyc c(r, Basis::CRT);
    for (uint32_t k = 0; k < r.n(); ++k)
        for (int j = 0; j < r.L(); ++j) c.data()[(size_t)k * r.L() + j] = (int64_t)(rng() % r.qs()[j]);
    return c;
}

// hint_i = g_i s^2 + LWE sample under s:  h0_i + h1_i s = g_i s^2 + e_i   (g_i = unit vector of limb i)
inline KSQuadCircHint ksQuadCircHint(const Ring& r, const SK& sk, std::mt19937_64& rng) {
    Cyc s = Cyc::fromIntegers(r, sk.s).adviseCRT();
    Cyc s2 = s * s;
    KSQuadCircHint hint;
    for (int i = 0; i < r.L(); ++i) {
        std::vector<uint64_t> gi(r.L(), 0);
        gi[i] = 1;
        Cyc e = Cyc::fromIntegers(r, gaussianPoly(r.n(), sk.r, rng)).adviseCRT();
        Cyc h1 = uniformCRT(r, rng);
        Cyc h0 = s2.scale(gi) + e - h1 * s;
        hint.h.emplace_back(h0, h1);
    }
    return hint;
}

// LSD encryption of a plaintext of R_p (index 2*npt) embedded into R' (index 2n): c0 + c1 s = e, e = pt mod p.
inline CT encrypt(const Ring& r, const SK& sk, const std::vector<uint64_t>& pt, uint64_t p, std::mt19937_64& rng) {
    const uint32_t n = r.n(), d = n / (uint32_t)pt.size();
    std::vector<int64_t> e = gaussianPoly(n, sk.r, rng);
    for (auto& v : e) v *= (int64_t)p;
    for (size_t i = 0; i < pt.size(); ++i) e[i * d] += (int64_t)pt[i];
    Cyc s = Cyc::fromIntegers(r, sk.s).adviseCRT();
    Cyc c1 = uniformCRT(r, rng);
    Cyc c0 = Cyc::fromIntegers(r, e).adviseCRT() - c1 * s;
    return CT{Encoding::LSD, 0, 1, p, {c0, c1}};
}

// decrypt: mu = l * g^-k * (c(s) mod p), twaced to the plaintext ring (g = 1).  Up to 4 limbs of < 2^31.
inline std::vector<uint64_t> decrypt(const SK& sk, const CT& ct_, size_t npt) {
    CT ct = toLSD(ct_);
    const Ring& r = ct.c[0].ring();
    Cyc s = Cyc::fromIntegers(r, sk.s).adviseCRT();
    Cyc acc(r, Basis::CRT);
    for (size_t i = ct.c.size(); i-- > 0;) acc = acc * s + ct.c[i];                      // Horner in S
    Cyc e = acc.advisePow();
    typedef __int128 i128;
    i128 Q = 1;
    for (uint64_t q : r.qs()) Q *= (i128)q;
    std::vector<uint64_t> out(npt);
    const uint32_t d = r.n() / (uint32_t)npt;
    for (size_t t = 0; t < npt; ++t) {
        i128 v = 0;
        for (int j = 0; j < r.L(); ++j) {
            const uint64_t q = r.qs()[j];
            i128 Qi = Q / (i128)q;
            uint64_t inv = invmod((uint64_t)(Qi % (i128)q), q);
            uint64_t x = (uint64_t)e.data()[(t * d) * r.L() + j];
            v = (v + Qi * (i128)mulmod(x, inv, q)) % Q;
        }
        if (2 * v >= Q) v -= Q;                                                          // centred lift
        int64_t m = (int64_t)(v % (i128)ct.p);
        if (m < 0) m += (int64_t)ct.p;
        out[t] = mulmod(ct.l, (uint64_t)m, ct.p);
    }
    return out;
}

// ---- the fused device path ----------------------------------------------------------------------------------
// PT2CT's  keySwitchQuad_ hint $: (x *: y)  on whole batches of linear ciphertexts: one alch_ct_mul_relin
// call (two kernel launches per chunk) instead of the ~40 Tensor calls the per-op path above makes.
// Returns MSD ciphertexts with k = k1+k2+1, l = l1*l2*(-q mod p); inputs must share (enc, k, l).
inline std::vector<CT> mulRelinBatch(const Ring& r, const KSQuadCircHint& hint, const std::vector<CT>& xs,
                                     const std::vector<CT>& ys) {
    const size_t B = xs.size();
    if (B == 0 || ys.size() != B) throw std::runtime_error("mulRelinBatch: batch mismatch");
    // toLSD of both operands and keySwitchQuadCirc's toMSD as one per-limb scalar
    std::vector<uint64_t> s(r.L(), 1);
    uint64_t lx = xs[0].l, ly = ys[0].l;
    const uint64_t p = xs[0].p;
    const uint64_t negq = (p - qprod_mod(r, p)) % p;
    auto fold = [&](Encoding enc, uint64_t& l) {
        if (enc == Encoding::MSD) {
            for (int j = 0; j < r.L(); ++j) s[j] = mulmod(s[j], p % r.qs()[j], r.qs()[j]);
            l = mulmod(l, invmod(negq, p), p);
        }
    };
    fold(xs[0].enc, lx);
    fold(ys[0].enc, ly);
    for (int j = 0; j < r.L(); ++j) s[j] = mulmod(s[j], invmod(p % r.qs()[j], r.qs()[j]), r.qs()[j]);   // toMSD
    const uint64_t lout = mulmod(mulmod(lx, ly, p), negq, p);

    alch_buf *ba = nullptr, *bb = nullptr, *bo = nullptr, *bh = nullptr;
    alch_hint* dh = nullptr;
    check(alch_buf_alloc(r.handle(), 2 * B, &ba), "alch_buf_alloc");
    check(alch_buf_alloc(r.handle(), 2 * B, &bb), "alch_buf_alloc");
    check(alch_buf_alloc(r.handle(), 2 * B, &bo), "alch_buf_alloc");
    check(alch_buf_alloc(r.handle(), 2 * (size_t)r.L(), &bh), "alch_buf_alloc");
    for (size_t i = 0; i < B; ++i) {
        if (xs[i].c.size() != 2 || ys[i].c.size() != 2) throw std::runtime_error("mulRelinBatch: linear ciphertexts only");
        for (int c = 0; c < 2; ++c) {
            check(alch_buf_upload(ba, 2 * i + c, 1, xs[i].c[c].adviseCRT().data().data()), "alch_buf_upload");
            check(alch_buf_upload(bb, 2 * i + c, 1, ys[i].c[c].adviseCRT().data().data()), "alch_buf_upload");
        }
    }
    for (int i = 0; i < r.L(); ++i) {
        check(alch_buf_upload(bh, 2 * i, 1, hint.h[i].first.adviseCRT().data().data()), "alch_buf_upload");
        check(alch_buf_upload(bh, 2 * i + 1, 1, hint.h[i].second.adviseCRT().data().data()), "alch_buf_upload");
    }
    check(alch_hint_from_buf(r.handle(), ALCH_GAD_TRIV, bh, &dh), "alch_hint_from_buf");
    check(alch_ct_mul_relin(r.handle(), dh, ba, bb, bo, B, s.data(), 0), "alch_ct_mul_relin");
    std::vector<CT> out;
    for (size_t i = 0; i < B; ++i) {
        CT o{Encoding::MSD, xs[i].k + ys[i].k + 1, lout, p, {Cyc(r, Basis::CRT), Cyc(r, Basis::CRT)}};
        for (int c = 0; c < 2; ++c) check(alch_buf_download(bo, 2 * i + c, 1, o.c[c].data().data()), "alch_buf_download");
        out.push_back(std::move(o));
    }
    alch_hint_free(dh);
    alch_buf_free(ba); alch_buf_free(bb); alch_buf_free(bo); alch_buf_free(bh);
    return out;
}

// PT2CT's whole mul_ on batches:  modSwitch_ .: keySwitchQuad_ hint .: modSwitch_ $: (x *: y)  (PT2CT.hs:172-177)
// with the hint on a ring `rh` that has extra limbs in front of the operands' ring `rin` (KSPNoise, PT2CT.hs:139)
// and the result on `rout`, the last limbs of rh.  One alch_ct_mul_full call (three kernel launches per chunk).
// Returns MSD ciphertexts in the Pow basis (what Lol's rescale leaves), k = k1+k2+1, l = l1*l2*(-q_in mod p).
inline std::vector<CT> mulFullBatch(const Ring& rin, const Ring& rh, const Ring& rout, const KSQuadCircHint& hint,
                                    const std::vector<CT>& xs, const std::vector<CT>& ys) {
    const size_t B = xs.size();
    if (B == 0 || ys.size() != B) throw std::runtime_error("mulFullBatch: batch mismatch");
    std::vector<uint64_t> s(rin.L(), 1);
    uint64_t lx = xs[0].l, ly = ys[0].l;
    const uint64_t p = xs[0].p;
    const uint64_t negq = (p - qprod_mod(rin, p)) % p;
    auto fold = [&](Encoding enc, uint64_t& l) {               // toLSD
        if (enc == Encoding::MSD) {
            for (int j = 0; j < rin.L(); ++j) s[j] = mulmod(s[j], p % rin.qs()[j], rin.qs()[j]);
            l = mulmod(l, invmod(negq, p), p);
        }
    };
    fold(xs[0].enc, lx);
    fold(ys[0].enc, ly);
    for (int j = 0; j < rin.L(); ++j) s[j] = mulmod(s[j], invmod(p % rin.qs()[j], rin.qs()[j]), rin.qs()[j]);   // first modSwitch's toMSD
    const uint64_t lout = mulmod(mulmod(lx, ly, p), negq, p);

    alch_buf *ba = nullptr, *bb = nullptr, *bo = nullptr, *bh = nullptr;
    alch_hint* dh = nullptr;
    check(alch_buf_alloc(rin.handle(), 2 * B, &ba), "alch_buf_alloc");
    check(alch_buf_alloc(rin.handle(), 2 * B, &bb), "alch_buf_alloc");
    check(alch_buf_alloc(rout.handle(), 2 * B, &bo), "alch_buf_alloc");
    check(alch_buf_alloc(rh.handle(), 2 * (size_t)rh.L(), &bh), "alch_buf_alloc");
    for (size_t i = 0; i < B; ++i) {
        if (xs[i].c.size() != 2 || ys[i].c.size() != 2) throw std::runtime_error("mulFullBatch: linear ciphertexts only");
        for (int c = 0; c < 2; ++c) {
            check(alch_buf_upload(ba, 2 * i + c, 1, xs[i].c[c].adviseCRT().data().data()), "alch_buf_upload");
            check(alch_buf_upload(bb, 2 * i + c, 1, ys[i].c[c].adviseCRT().data().data()), "alch_buf_upload");
        }
    }
    for (int i = 0; i < rh.L(); ++i) {
        check(alch_buf_upload(bh, 2 * i, 1, hint.h[i].first.adviseCRT().data().data()), "alch_buf_upload");
        check(alch_buf_upload(bh, 2 * i + 1, 1, hint.h[i].second.adviseCRT().data().data()), "alch_buf_upload");
    }
    check(alch_hint_from_buf(rh.handle(), ALCH_GAD_TRIV, bh, &dh), "alch_hint_from_buf");
    check(alch_ct_mul_full(dh, ba, bb, bo, B, s.data(), ALCH_POW_OUT), "alch_ct_mul_full");
    std::vector<CT> out;
    for (size_t i = 0; i < B; ++i) {
        CT o{Encoding::MSD, xs[i].k + ys[i].k + 1, lout, p, {Cyc(rout, Basis::Pow), Cyc(rout, Basis::Pow)}};
        for (int c = 0; c < 2; ++c) check(alch_buf_download(bo, 2 * i + c, 1, o.c[c].data().data()), "alch_buf_download");
        out.push_back(std::move(o));
    }
    alch_hint_free(dh);
    alch_buf_free(ba); alch_buf_free(bb); alch_buf_free(bo); alch_buf_free(bh);
    return out;
}

}  // namespace alchemy
