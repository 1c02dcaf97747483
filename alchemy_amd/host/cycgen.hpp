// Host-side mirror (C++, header-only) of Lol's Cyc layer for ANY cyclotomic index, written above the C ABI of
// include/alchemy_hip.h -- the part of the Haskell host that examples/HomomRLWR.hs and examples/Tunnel.hs need beyond
// alchemy_amd/host/symmshe.hpp (which replays examples/Arithmetic.hs on a two-power index):
//
//   Ring / RingCache   one `Cyc t m r` type: (index, moduli); with CRT (ciphertext rings Z_q, q = 1 mod m) or without (plaintext
//                      rings Z_{2^k}, the integers): alch_ring_create / alch_ring_create_nocrt
//   Cyc                ring element with basis tracking Pow / Dec / CRT like Lol's UCyc; toPow / toDec / toCRT (Tensor crt, crtInv,
//                      l, lInv), (+), (-), (*) (CRT basis: zipWithT), scalar products, mulG / divG, embed / twace / coeffs
//                      (Tensor embedPow, embedDec, embedCRT, twacePowDec, twaceCRT, coeffs)
//   tGaussianDec       Lol's tweaked Gaussian t*D on the decoding basis (setup time: keys, hint errors, encryption errors):
//                      errorRounded, errorCoset        (KeysHints.hs:86-96, PT2CT.hs:84-87 via SymmSHE genSK / encrypt)
//   PtRing / PtCyc     plaintext ring elements over Z_{2^k} (Common.hs:32): exact products through a lifting (Lol: the CRT over an
//                      extension ring; here two word-size primes of the library), Dec / Pow changes, embed, coeffs, div2 = rescalePow
//   crtSet, decToCRT   Cyc's crtSet over Z_{p^e} (Hensel lifting of Tensor crtSetDec) and the linear functions of
//                      examples/Common.hs:65-75;  evalLin
//
// Everything numeric on ring elements goes through the C ABI; this file sequences calls, keeps bases and does the scalar /
// sampling work a Lol host does in Haskell.
//
// Three execution modes of the per-element layer (alchemy::gen::mode()), mirroring the representations of `GT` in
// haskell/Crypto/Lol/Cyclotomic/Tensor/GT.hs:
//   HostBuffers       a ring element is a host vector; every Tensor call stages it through the GPU (GT's constructor GTHost with
//                     the host-buffer entry points alch_crt, alch_mul, ...: two PCIe crossings and two synchronisations per call)
//   Resident          a ring element lives in HBM (GTDev: one pooled alch_buf element); a Tensor call is one asynchronous kernel
//                     launch (alch_buf_tensor_op, alch_buf_embed / twace / coeffs, alch_buf_mul / add / sub / scale,
//                     alch_buf_decompose_triv); the host sees data only when it asks (data(), lifts) -- what a Lol whose Cyc
//                     routes its pointwise ring operations to GT's mulGT / addGT / subGT gets
//   ResidentZipHost   the same, but pointwise operations run on the HOST as Lol's unchanged UCyc issues them (zipWithT / fmapT
//                     with an opaque Haskell function: GT downloads, lol-cpp computes; here plain C++ stands in for lol-cpp),
//                     so every product costs two downloads and its result a later upload
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/alchemy_hip.h"

namespace alchemy {
namespace gen {

typedef __int128 i128;

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
inline uint64_t invmod_prime(uint64_t a, uint64_t q) { return powmod(a % q, q - 2, q); }
inline int64_t invmod_any(int64_t a, int64_t m) {        // a^-1 mod m for gcd(a, m) = 1 (m need not be prime: plaintext moduli 2^k)
    int64_t r0 = m, r1 = ((a % m) + m) % m, t0 = 0, t1 = 1;
    while (r1) { int64_t k = r0 / r1, r2 = r0 - k * r1, t2 = t0 - k * t1; r0 = r1; r1 = r2; t0 = t1; t1 = t2; }
    if (r0 != 1) throw std::runtime_error("invmod: not a unit");
    return ((t0 % m) + m) % m;
}
inline int64_t centred(int64_t v, int64_t q) { v %= q; if (v < 0) v += q; return v > (q - 1) / 2 ? v - q : v; }
inline uint32_t totient(uint32_t m) {
    uint32_t r = m, t = m;
    for (uint32_t p = 2; (uint64_t)p * p <= t; ++p) if (t % p == 0) { r -= r / p; while (t % p == 0) t /= p; }
    return t > 1 ? r - r / t : r;
}
inline std::vector<std::pair<uint32_t, int>> factor(uint32_t m) {
    std::vector<std::pair<uint32_t, int>> f;
    for (uint32_t p = 2; (uint64_t)p * p <= m; ++p) if (m % p == 0) { int e = 0; while (m % p == 0) { m /= p; ++e; } f.push_back({p, e}); }
    if (m > 1) f.push_back({m, 1});
    return f;
}

// ---- ring context ---------------------------------------------------------------------------------------------
class Ring {
public:
    Ring(uint32_t m, std::vector<uint64_t> qs, bool crt) : m_(m), qs_(std::move(qs)), crt_(crt) {
        check((crt ? alch_ring_create : alch_ring_create_nocrt)(m, (int)qs_.size(), qs_.data(), &h_), "alch_ring_create");
        check(alch_ring_n(h_, &n_, nullptr, nullptr), "alch_ring_n");
    }
    ~Ring() { alch_ring_destroy(h_); }
    Ring(const Ring&) = delete;
    Ring& operator=(const Ring&) = delete;
    alch_ring* handle() const { return h_; }
    uint32_t m() const { return m_; }
    uint32_t n() const { return n_; }
    int L() const { return (int)qs_.size(); }
    bool hasCRT() const { return crt_; }
    const std::vector<uint64_t>& qs() const { return qs_; }
    size_t words() const { return (size_t)n_ * qs_.size(); }

private:
    uint32_t m_, n_ = 0;
    std::vector<uint64_t> qs_;
    bool crt_;
    alch_ring* h_ = nullptr;
};

// One handle per (index, moduli, crt): device buffers are tied to ring handles, so every user must see the same Ring.
class RingCache {
public:
    const Ring& get(uint32_t m, const std::vector<uint64_t>& qs, bool crt = true) {
        auto key = std::make_tuple(m, qs, crt);
        auto it = rings_.find(key);
        if (it == rings_.end()) {
            it = rings_.emplace(key, std::unique_ptr<Ring>(new Ring(m, qs, crt))).first;
            // one Tensor call at a time: all rings queue on one stream, so calls between two rings need no events
            if (!first_) first_ = it->second.get();
            else check(alch_ring_share_stream(it->second->handle(), first_->handle()), "alch_ring_share_stream");
        }
        return *it->second;
    }

private:
    std::map<std::tuple<uint32_t, std::vector<uint64_t>, bool>, std::unique_ptr<Ring>> rings_;
    const Ring* first_ = nullptr;
};

enum class Basis { Pow, Dec, CRT };

enum class Mode { HostBuffers, Resident, ResidentZipHost };
inline Mode& mode() { static Mode m = Mode::HostBuffers; return m; }
inline bool resident() { return mode() != Mode::HostBuffers; }

// One ring element (or a run of them) resident in HBM.  Buffers come from the library's per-ring free list; a view aliases part of
// a bigger buffer (the d_rel results of coeffs, the L digits of decompose) and keeps it alive.
struct DevElem {
    alch_buf* b = nullptr;
    std::shared_ptr<DevElem> parent;
    DevElem() {}
    DevElem(const DevElem&) = delete;
    DevElem& operator=(const DevElem&) = delete;
    ~DevElem() { if (b) alch_buf_free(b); }
    static std::shared_ptr<DevElem> make(alch_ring* r, size_t elems = 1) {
        auto d = std::make_shared<DevElem>();
        check(alch_buf_alloc(r, elems, &d->b), "alch_buf_alloc");
        return d;
    }
    static std::shared_ptr<DevElem> view(const std::shared_ptr<DevElem>& p, size_t first) {
        auto d = std::make_shared<DevElem>();
        check(alch_buf_view(p->b, first, 1, &d->b), "alch_buf_view");
        d->parent = p;
        return d;
    }
};

// ---- ring element ------------------------------------------------------------------------------------------------
// Immutable value semantics like Lol's UCyc: every operation returns a new element.  An element has a host copy (Lol's
// tuple-interleaved int64 vector), a device copy (one alch_buf element), or both; each is produced from the other on demand.
class Cyc {
public:
    Cyc() : r_(nullptr), basis_(Basis::Pow) {}
    Cyc(const Ring& r, Basis b) : r_(&r), basis_(b), hp_(std::make_shared<std::vector<int64_t>>(r.words(), 0)) {}
    // `reduce` of an integer vector given on basis b (Pow or Dec)
    static Cyc fromIntegers(const Ring& r, const std::vector<int64_t>& z, Basis b = Basis::Pow) {
        if (z.size() != r.n()) throw std::runtime_error("fromIntegers: wrong dimension");
        Cyc c(r, b);
        for (uint32_t k = 0; k < r.n(); ++k)
            for (int j = 0; j < r.L(); ++j) {
                const int64_t q = (int64_t)r.qs()[j];
                int64_t v = q ? z[k] % q : z[k];
                (*c.hp_)[(size_t)k * r.L() + j] = (q && v < 0) ? v + q : v;
            }
        return c;
    }
    static Cyc scalar(const Ring& r, int64_t s) {                 // scalarPow
        std::vector<int64_t> z(r.n(), 0);
        z[0] = s;
        return fromIntegers(r, z, Basis::Pow);
    }
    // an element that exists on the device only (the result of a Tensor call in the resident modes)
    static Cyc onDevice(const Ring& r, Basis b, std::shared_ptr<DevElem> d) {
        Cyc c;
        c.r_ = &r; c.basis_ = b; c.d_ = std::move(d);
        return c;
    }
    const Ring& ring() const { return *r_; }
    Basis basis() const { return basis_; }
    bool onDeviceOnly() const { return !hp_; }
    // host copy: downloaded on first use (pinned staging, one synchronisation)
    const std::vector<int64_t>& data() const {
        if (!hp_) {
            hp_ = std::make_shared<std::vector<int64_t>>(r_->words(), 0);
            check(alch_buf_download(d_->b, 0, 1, hp_->data()), "alch_buf_download");
        }
        return *hp_;
    }
    // writable host copy: private to this value, and the device copy (shared with other values) is dropped
    std::vector<int64_t>& data() {
        (void)static_cast<const Cyc*>(this)->data();
        if (hp_.use_count() > 1) hp_ = std::make_shared<std::vector<int64_t>>(*hp_);
        d_.reset();
        return *hp_;
    }
    // device copy: uploaded on first use (no synchronisation)
    alch_buf* dev() const {
        if (!d_) {
            d_ = DevElem::make(r_->handle());
            check(alch_buf_upload(d_->b, 0, 1, hp_->data()), "alch_buf_upload");
        }
        return d_->b;
    }
    int64_t at(uint32_t k, int j) const { return data()[(size_t)k * r_->L() + j]; }

    Cyc toPow() const {
        if (basis_ == Basis::Pow) return *this;
        if (resident()) return unary(basis_ == Basis::CRT ? ALCH_T_CRTINV : ALCH_T_L, Basis::Pow);
        Cyc o = *this;                                              // the writable data() below is private to o (copy on write)
        if (basis_ == Basis::CRT) check(alch_crtinv(r_->handle(), o.data().data()), "alch_crtinv");
        else check(alch_l(r_->handle(), o.data().data()), "alch_l");
        o.basis_ = Basis::Pow;
        return o;
    }
    Cyc toDec() const {
        if (basis_ == Basis::Dec) return *this;
        Cyc o = toPow();
        if (resident()) return o.unary(ALCH_T_LINV, Basis::Dec);
        check(alch_linv(r_->handle(), o.data().data()), "alch_linv");
        o.basis_ = Basis::Dec;
        return o;
    }
    Cyc toCRT() const {
        if (basis_ == Basis::CRT) return *this;
        Cyc o = toPow();
        if (resident()) return o.unary(ALCH_T_CRT, Basis::CRT);
        check(alch_crt(r_->handle(), o.data().data()), "alch_crt");
        o.basis_ = Basis::CRT;
        return o;
    }
    Cyc to(Basis b) const { return b == Basis::Pow ? toPow() : b == Basis::Dec ? toDec() : toCRT(); }

    friend Cyc operator*(const Cyc& a, const Cyc& b) {            // ring product: both to the CRT basis, zipWithT (*)
        Cyc x = a.toCRT(), y = b.toCRT();
        if (mode() == Mode::Resident) return x.binary(y, 0);
        if (mode() == Mode::ResidentZipHost) return x.zipHost(y, 0);
        x = x.hostCopy();
        check(alch_mul(x.r_->handle(), x.data().data(), y.data().data()), "alch_mul");
        return x;
    }
    friend Cyc operator+(const Cyc& a, const Cyc& b) {
        Cyc x = a, y = b.to(a.basis_);
        if (mode() == Mode::Resident) return x.binary(y, 1);
        if (mode() == Mode::ResidentZipHost) return x.zipHost(y, 1);
        x = x.hostCopy();
        check(alch_add(x.r_->handle(), x.data().data(), y.data().data()), "alch_add");
        return x;
    }
    friend Cyc operator-(const Cyc& a, const Cyc& b) {
        Cyc x = a, y = b.to(a.basis_);
        if (mode() == Mode::Resident) return x.binary(y, 2);
        if (mode() == Mode::ResidentZipHost) return x.zipHost(y, 2);
        x = x.hostCopy();
        check(alch_sub(x.r_->handle(), x.data().data(), y.data().data()), "alch_sub");
        return x;
    }
    Cyc scale(const std::vector<uint64_t>& s) const {             // per-limb scalar (any basis)
        if (mode() == Mode::Resident && r_->hasCRT()) {
            auto d = DevElem::make(r_->handle());
            check(alch_buf_scale(d->b, dev(), 1, s.data()), "alch_buf_scale");
            return onDevice(*r_, basis_, d);
        }
        if (mode() == Mode::ResidentZipHost) {                      // Lol: fmapT (* s) with an opaque function -> host
            Cyc o = hostCopy();
            for (uint32_t k = 0; k < r_->n(); ++k)
                for (int j = 0; j < r_->L(); ++j) {
                    int64_t& w = (*o.hp_)[(size_t)k * r_->L() + j];
                    w = (int64_t)mulmod((uint64_t)w, s[j] % r_->qs()[j], r_->qs()[j]);
                }
            return o;
        }
        Cyc o = hostCopy();
        check(alch_scale(r_->handle(), o.data().data(), s.data()), "alch_scale");
        return o;
    }
    Cyc mulG() const {
        if (resident()) return unary(basis_ == Basis::Pow ? ALCH_T_MULG_POW : basis_ == Basis::Dec ? ALCH_T_MULG_DEC : ALCH_T_MULG_CRT, basis_);
        Cyc o = hostCopy();
        check(basis_ == Basis::Pow ? alch_mulg_pow(r_->handle(), o.data().data())
              : basis_ == Basis::Dec ? alch_mulg_dec(r_->handle(), o.data().data()) : alch_mulg_crt(r_->handle(), o.data().data()), "alch_mulg");
        return o;
    }
    // Lol's divG: false = Nothing
    bool divG(Cyc& out) const {
        if (resident()) {
            auto d = DevElem::make(r_->handle());
            const int op = basis_ == Basis::Pow ? ALCH_T_DIVG_POW : basis_ == Basis::Dec ? ALCH_T_DIVG_DEC : ALCH_T_DIVG_CRT;
            const int rc = alch_buf_tensor_op(d->b, 0, dev(), 0, 1, op);
            check(rc, "alch_buf_tensor_op (divG)");
            out = onDevice(*r_, basis_, d);
            return rc != ALCH_NOT_DIVISIBLE;
        }
        out = hostCopy();
        int rc = basis_ == Basis::Pow ? alch_divg_pow(r_->handle(), out.data().data())
                 : basis_ == Basis::Dec ? alch_divg_dec(r_->handle(), out.data().data()) : alch_divg_crt(r_->handle(), out.data().data());
        check(rc, "alch_divg");
        return rc != ALCH_NOT_DIVISIBLE;
    }
    // embed into a ring of a multiple index with the same moduli (Cyc embed: stays on this element's basis)
    Cyc embed(const Ring& big) const {
        const int bs = basis_ == Basis::Pow ? ALCH_BASIS_POW : basis_ == Basis::Dec ? ALCH_BASIS_DEC : ALCH_BASIS_CRT;
        if (resident()) {
            auto d = DevElem::make(big.handle());
            check(alch_buf_embed(d->b, dev(), 1, bs), "alch_buf_embed");
            return onDevice(big, basis_, d);
        }
        Cyc o(big, basis_);
        auto f = basis_ == Basis::Pow ? alch_embed_pow : basis_ == Basis::Dec ? alch_embed_dec : alch_embed_crt;
        check(f(r_->handle(), big.handle(), data().data(), o.data().data()), "alch_embed");
        return o;
    }
    // tweaked trace to a ring of a divisor index (Cyc twace)
    Cyc twace(const Ring& small) const {
        if (resident()) {
            auto d = DevElem::make(small.handle());
            check(alch_buf_twace(d->b, dev(), 1, basis_ == Basis::CRT ? ALCH_BASIS_CRT : ALCH_BASIS_POW), "alch_buf_twace");
            return onDevice(small, basis_, d);
        }
        Cyc o(small, basis_);
        auto f = basis_ == Basis::CRT ? alch_twace_crt : alch_twace_pow_dec;
        check(f(small.handle(), r_->handle(), data().data(), o.data().data()), "alch_twace");
        return o;
    }
    // Cyc coeffsPow / coeffsDec: the d_rel coefficient vectors over the subring w.r.t. the relative powerful / decoding basis
    std::vector<Cyc> coeffs(const Ring& small, Basis b) const {
        Cyc src = to(b);
        if (b == Basis::CRT) throw std::runtime_error("coeffs: Pow or Dec");
        const uint32_t d = r_->n() / small.n();
        std::vector<Cyc> out;
        if (resident()) {
            auto all = DevElem::make(small.handle(), d);
            check(alch_buf_coeffs(all->b, src.dev(), 1), "alch_buf_coeffs");
            for (uint32_t i = 0; i < d; ++i) out.push_back(onDevice(small, b, DevElem::view(all, i)));
            return out;
        }
        std::vector<int64_t> all((size_t)d * small.words());
        check(alch_coeffs(small.handle(), r_->handle(), src.data().data(), all.data()), "alch_coeffs");
        for (uint32_t i = 0; i < d; ++i) {
            Cyc c(small, b);
            std::copy(all.begin() + (size_t)i * small.words(), all.begin() + (size_t)(i + 1) * small.words(), c.hp_->begin());
            out.push_back(std::move(c));
        }
        return out;
    }
    // Lol `decompose` (TrivGad) then `reduce <$>` (Pow basis)
    std::vector<Cyc> decomposeTrivReduced() const {
        Cyc p = toPow();
        std::vector<Cyc> out;
        if (mode() == Mode::Resident) {
            auto all = DevElem::make(r_->handle(), (size_t)r_->L());
            check(alch_buf_decompose_triv(p.dev(), 0, all->b, 0), "alch_buf_decompose_triv");
            for (int i = 0; i < r_->L(); ++i) out.push_back(onDevice(*r_, Basis::Pow, DevElem::view(all, (size_t)i)));
            return out;
        }
        if (mode() == Mode::ResidentZipHost) {                      // Lol: fmapT lift / reduce per coefficient -> host
            const std::vector<int64_t>& v = p.data();
            for (int i = 0; i < r_->L(); ++i) {
                Cyc dgt(*r_, Basis::Pow);
                const int64_t qi = (int64_t)r_->qs()[i];
                for (uint32_t k = 0; k < r_->n(); ++k) {
                    const int64_t c = centred(v[(size_t)k * r_->L() + i], qi);
                    for (int j = 0; j < r_->L(); ++j) {
                        const int64_t qj = (int64_t)r_->qs()[j];
                        int64_t w = c % qj;
                        (*dgt.hp_)[(size_t)k * r_->L() + j] = w < 0 ? w + qj : w;
                    }
                }
                out.push_back(std::move(dgt));
            }
            return out;
        }
        std::vector<int64_t> all((size_t)r_->L() * r_->words());
        check(alch_decompose_triv(r_->handle(), p.data().data(), all.data()), "alch_decompose_triv");
        for (int i = 0; i < r_->L(); ++i) {
            Cyc d(*r_, Basis::Pow);
            std::copy(all.begin() + (size_t)i * r_->words(), all.begin() + (size_t)(i + 1) * r_->words(), d.hp_->begin());
            out.push_back(std::move(d));
        }
        return out;
    }
    // Centred lift modulo Q = prod q of every coefficient of the current (Pow or Dec) basis (Lol liftPow / liftDec), as 128-bit
    // integers: rings of up to four ~31-bit limbs (Q < 2^126).
    std::vector<i128> liftCoeffs() const {
        if (basis_ == Basis::CRT) throw std::runtime_error("lift: Pow or Dec basis");
        long double bits = 0;
        for (uint64_t q : r_->qs()) bits += std::log2((long double)q);
        if (bits > 125) throw std::runtime_error("liftCoeffs: modulus too large for 128-bit lifts (use liftModP)");
        i128 Q = 1;
        for (uint64_t q : r_->qs()) Q *= (i128)q;
        std::vector<i128> Qi(r_->L());
        std::vector<uint64_t> inv(r_->L());
        for (int j = 0; j < r_->L(); ++j) {
            Qi[j] = Q / (i128)r_->qs()[j];
            inv[j] = invmod_prime((uint64_t)(Qi[j] % (i128)r_->qs()[j]), r_->qs()[j]);
        }
        std::vector<i128> out(r_->n());
        for (uint32_t k = 0; k < r_->n(); ++k) {
            i128 v = 0;
            for (int j = 0; j < r_->L(); ++j) v = (v + Qi[j] * (i128)mulmod((uint64_t)at(k, j), inv[j], r_->qs()[j])) % Q;
            if (2 * v >= Q) v -= Q;
            out[k] = v;
        }
        return out;
    }
    // The same lift for any number of limbs, reduced mod p, without big integers: mixed-radix (Garner) digits
    // x = d_0 + q_0 (d_1 + q_1 (d_2 + ..)), d_j in [0, q_j).  All q_j are odd, so (Q - 1)/2 has the digits (q_j - 1)/2 and the
    // centring test x > (Q - 1)/2 is a comparison of digit vectors from the top.  max_abs_over_q (optional): max |x| / Q.
    std::vector<int64_t> liftModP(int64_t p, double* max_abs_over_q = nullptr) const {
        if (basis_ == Basis::CRT) throw std::runtime_error("lift: Pow or Dec basis");
        const int L = r_->L();
        const std::vector<uint64_t>& q = r_->qs();
        std::vector<std::vector<uint64_t>> inv(L, std::vector<uint64_t>(L, 0));      // inv[j][i] = q_i^-1 mod q_j, i < j
        for (int j = 0; j < L; ++j) for (int i = 0; i < j; ++i) inv[j][i] = invmod_prime(q[i] % q[j], q[j]);
        std::vector<int64_t> wp(L);                                                    // prod_{i<j} q_i mod p
        std::vector<long double> wf(L);
        int64_t accp = 1 % p;
        long double accf = 1, Qf = 1;
        for (int j = 0; j < L; ++j) { wp[j] = accp; wf[j] = accf; accp = (int64_t)(((i128)accp * (i128)(q[j] % (uint64_t)p)) % p); accf *= (long double)q[j]; }
        Qf = accf;
        const int64_t Qp = accp;
        std::vector<int64_t> out(r_->n());
        long double worst = 0;
        std::vector<uint64_t> d(L);
        for (uint32_t k = 0; k < r_->n(); ++k) {
            for (int j = 0; j < L; ++j) {
                uint64_t t = (uint64_t)at(k, j);
                for (int i = 0; i < j; ++i) t = mulmod((t + q[j] - d[i] % q[j]) % q[j], inv[j][i], q[j]);
                d[j] = t;
            }
            bool neg = false;
            for (int j = L - 1; j >= 0; --j) {
                const uint64_t h = (q[j] - 1) / 2;
                if (d[j] != h) { neg = d[j] > h; break; }
            }
            i128 vp = 0;
            long double vf = 0;
            for (int j = 0; j < L; ++j) { vp += (i128)(d[j] % (uint64_t)p) * wp[j]; vf += (long double)d[j] * wf[j]; }
            int64_t r = (int64_t)(vp % p);
            if (neg) { r = ((r - Qp) % p + p) % p; vf = Qf - vf; }
            out[k] = r;
            worst = std::max(worst, vf / Qf);
        }
        if (max_abs_over_q) *max_abs_over_q = (double)worst;
        return out;
    }

private:
    // a value whose host vector may be modified in place (HostBuffers mode: the host-buffer entry points work in place)
    Cyc hostCopy() const {
        Cyc o;
        o.r_ = r_; o.basis_ = basis_; o.hp_ = std::make_shared<std::vector<int64_t>>(data());
        return o;
    }
    Cyc unary(int op, Basis nb) const {
        auto d = DevElem::make(r_->handle());
        check(alch_buf_tensor_op(d->b, 0, dev(), 0, 1, op), "alch_buf_tensor_op");
        return onDevice(*r_, nb, d);
    }
    Cyc binary(const Cyc& y, int op) const {                        // 0 mul, 1 add, 2 sub on the device
        auto d = DevElem::make(r_->handle());
        check((op == 0 ? alch_buf_mul : op == 1 ? alch_buf_add : alch_buf_sub)(d->b, dev(), y.dev(), 1), "alch_buf_mul/add/sub");
        return onDevice(*r_, basis_, d);
    }
    // zipWithT f a b with an opaque f: both operands come to the host and lol-cpp computes (plain C++ stands in for it here)
    Cyc zipHost(const Cyc& y, int op) const {
        Cyc o = hostCopy();
        const std::vector<int64_t>& w = y.data();
        const int L = r_->L();
        for (uint32_t k = 0; k < r_->n(); ++k)
            for (int j = 0; j < L; ++j) {
                const uint64_t q = r_->qs()[j];
                int64_t& x = (*o.hp_)[(size_t)k * L + j];
                const int64_t z = w[(size_t)k * L + j];
                if (op == 0) x = q ? (int64_t)mulmod((uint64_t)x, (uint64_t)z, q) : x * z;
                else if (op == 1) { x += z; if (q && (uint64_t)x >= q) x -= (int64_t)q; }
                else { x -= z; if (q && x < 0) x += (int64_t)q; }
            }
        return o;
    }

    const Ring* r_;
    Basis basis_;
    mutable std::shared_ptr<std::vector<int64_t>> hp_;      // host copy, shared between copies of the value (copy on write); null = none
    mutable std::shared_ptr<DevElem> d_;                    // device copy; null = none
};

// ---- the tweaked Gaussian on the decoding basis (Lol Tensor tGaussianDec; SymmSHE genSK / errorRounded / errorCoset) ----
// t*D for D the spherical Gaussian of SCALED variance v over K_R (true variance v / 2 pi per real coordinate of the canonical
// embedding), t = mhat / g: its coefficient vector on the decoding basis of R (= t x decoding basis of R^dual) is
//     sqrt(rad m) * D * x,   x iid N(0, v (m / rad m) / 2 pi),   D = kron_l (D_{p_l} (x) I_{m_l/p_l}),   D_p D_p^T = I - J/p
// (toolkit section 6.3; covariance of the coefficients of a spherical Gaussian on the dual of the conjugate powerful basis is
// sigma^2 (m_l/p)(p I - J) per prime power).  D_p is taken as the Cholesky factor: only the distribution matters.  PARITY UNPINNED
// against Lol's sampler (its source is not in the reference); what the reference fixes is the parameter: svar = r / sqrt(phi(m'))
// (KeysHints.hs:86-87) with r = 5.0 for HomomRLWR (examples/HomomRLWR.hs:56), 3.0 for Arithmetic / Tunnel.
inline std::vector<double> tGaussianDec(uint32_t m, double svar, std::mt19937_64& rng) {
    const auto fs = factor(m);
    uint32_t rad = 1, n = 1;
    for (auto& f : fs) { rad *= f.first; n *= (f.first - 1); for (int e = 1; e < f.second; ++e) n *= f.first; }
    std::normal_distribution<double> g(0.0, std::sqrt(svar * (double)(m / rad) / (2.0 * M_PI)));
    std::vector<double> x(n);
    for (auto& v : x) v = g(rng);
    // apply sqrt(p) D_p along every axis (mixed radix, first factor outermost; within an axis index = j0 * m' + j1)
    uint32_t stride = n;
    for (auto& f : fs) {
        const uint32_t p = f.first;
        uint32_t mp = 1;
        for (int e = 1; e < f.second; ++e) mp *= p;
        const uint32_t dim = (p - 1) * mp;
        stride /= dim;
        const uint32_t h = p - 1;
        // Cholesky of C = p I - J  (h x h): lower-triangular Lc with Lc Lc^T = C
        std::vector<double> Lc((size_t)h * h, 0.0);
        for (uint32_t i = 0; i < h; ++i)
            for (uint32_t j = 0; j <= i; ++j) {
                double s = (i == j ? (double)p - 1.0 : -1.0);
                for (uint32_t k = 0; k < j; ++k) s -= Lc[i * h + k] * Lc[j * h + k];
                Lc[i * h + j] = i == j ? std::sqrt(s) : s / Lc[j * h + j];
            }
        std::vector<double> col(h);
        for (uint32_t base = 0; base < n; ++base) {
            if ((base / stride) % dim >= mp) continue;               // one visit per column: j0 = 0
            for (uint32_t i = 0; i < h; ++i) col[i] = x[base + (size_t)i * mp * stride];
            for (uint32_t i = h; i-- > 0;) {
                double s = 0;
                for (uint32_t k = 0; k <= i; ++k) s += Lc[i * h + k] * col[k];
                x[base + (size_t)i * mp * stride] = s;
            }
        }
    }
    return x;
}

// errorRounded: the tweaked Gaussian rounded coefficient-wise on the decoding basis (integer Dec coefficients)
inline std::vector<int64_t> errorRoundedDec(uint32_t m, double svar, std::mt19937_64& rng) {
    std::vector<double> x = tGaussianDec(m, svar, rng);
    std::vector<int64_t> z(x.size());
    for (size_t i = 0; i < x.size(); ++i) z[i] = (int64_t)std::llround(x[i]);
    return z;
}

// errorCoset: scaled variance svar * p^2, rounded on the decoding basis into the coset pt + p R (pt: Dec coefficients mod p)
inline std::vector<int64_t> errorCosetDec(uint32_t m, double svar, int64_t p, const std::vector<int64_t>& pt_dec, std::mt19937_64& rng) {
    std::vector<double> x = tGaussianDec(m, svar * (double)p * (double)p, rng);
    std::vector<int64_t> z(x.size());
    for (size_t i = 0; i < x.size(); ++i) {
        const int64_t c = centred(pt_dec[i], p);
        z[i] = c + p * (int64_t)std::llround((x[i] - (double)c) / (double)p);
    }
    return z;
}

// integer Dec coefficients -> integer Pow coefficients (Tensor l over Int64: a ring of the integers, modulus 0)
inline std::vector<int64_t> decToPowZ(RingCache& rc, uint32_t m, const std::vector<int64_t>& dec) {
    const Ring& z = rc.get(m, {0}, false);
    std::vector<int64_t> v = dec;
    check(alch_l(z.handle(), v.data()), "alch_l (integers)");
    return v;
}

// ---- plaintext rings Z_{2^k} (examples/Common.hs:32) ---------------------------------------------------------------
// A plaintext ring element: coefficients mod p = 2^k on the Pow or the Dec basis of index m.  Sums and basis changes go through a
// ring without CRT (alch_ring_create_nocrt); products through a lifting to two word-size primes = 1 mod m (exact: the product's
// integer coefficients are far below their 2^60 range), as Lol multiplies over an extension ring when Z_p has no CRT basis.
struct PtCyc {
    uint32_t m = 1;
    int64_t p = 2;
    Basis basis = Basis::Pow;
    std::vector<int64_t> v;          // phi(m) residues in [0, p)
};

class PtOps {
public:
    PtOps(RingCache& rc, std::vector<uint64_t> lift_primes) : rc_(rc), lp_(std::move(lift_primes)) {
        if (lp_.size() != 2) throw std::runtime_error("PtOps: two lifting primes");
    }
    PtCyc to(const PtCyc& a, Basis b) const {
        if (a.basis == b) return a;
        if (b == Basis::CRT || a.basis == Basis::CRT) throw std::runtime_error("plaintext rings have no CRT basis");
        const Ring& r = rc_.get(a.m, {(uint64_t)a.p}, false);
        PtCyc o = a;
        check(b == Basis::Pow ? alch_l(r.handle(), o.v.data()) : alch_linv(r.handle(), o.v.data()), "alch_l/linv (plaintext)");
        o.basis = b;
        return o;
    }
    PtCyc add(const PtCyc& a, const PtCyc& b_) const {
        PtCyc b = to(b_, a.basis), o = a;
        for (size_t i = 0; i < o.v.size(); ++i) o.v[i] = (a.v[i] + b.v[i]) % a.p;
        return o;
    }
    PtCyc addScalar(const PtCyc& a_, int64_t s) const {
        PtCyc o = to(a_, Basis::Pow);
        o.v[0] = (((o.v[0] + s) % o.p) + o.p) % o.p;
        return o;
    }
    PtCyc mul(const PtCyc& a_, const PtCyc& b_) const {
        const PtCyc a = to(a_, Basis::Pow), b = to(b_, Basis::Pow);
        const Ring& r = rc_.get(a.m, lp_, true);
        std::vector<int64_t> za(a.v.size()), zb(b.v.size());
        for (size_t i = 0; i < za.size(); ++i) { za[i] = centred(a.v[i], a.p); zb[i] = centred(b.v[i], b.p); }
        Cyc prod = (Cyc::fromIntegers(r, za) * Cyc::fromIntegers(r, zb)).toPow();
        std::vector<i128> z = prod.liftCoeffs();
        PtCyc o{a.m, a.p, Basis::Pow, std::vector<int64_t>(a.v.size())};
        for (size_t i = 0; i < z.size(); ++i) { int64_t t = (int64_t)(z[i] % (i128)a.p); o.v[i] = t < 0 ? t + a.p : t; }
        return o;
    }
    // Cyc embed (Pow basis) into the ring of a multiple index
    PtCyc embed(const PtCyc& a_, uint32_t mbig) const {
        const PtCyc a = to(a_, Basis::Pow);
        const Ring &rs = rc_.get(a.m, {(uint64_t)a.p}, false), &rb = rc_.get(mbig, {(uint64_t)a.p}, false);
        PtCyc o{mbig, a.p, Basis::Pow, std::vector<int64_t>(rb.n())};
        check(alch_embed_pow(rs.handle(), rb.handle(), a.v.data(), o.v.data()), "alch_embed_pow (plaintext)");
        return o;
    }
    // coeffsDec: Dec-basis coefficient vectors over the subring of index msmall
    std::vector<PtCyc> coeffsDec(const PtCyc& a_, uint32_t msmall) const {
        const PtCyc a = to(a_, Basis::Dec);
        const Ring &rs = rc_.get(msmall, {(uint64_t)a.p}, false), &rb = rc_.get(a.m, {(uint64_t)a.p}, false);
        const uint32_t d = rb.n() / rs.n();
        std::vector<int64_t> all((size_t)d * rs.n());
        check(alch_coeffs(rs.handle(), rb.handle(), a.v.data(), all.data()), "alch_coeffs (plaintext)");
        std::vector<PtCyc> out;
        for (uint32_t i = 0; i < d; ++i)
            out.push_back(PtCyc{msmall, a.p, Basis::Dec, std::vector<int64_t>(all.begin() + (size_t)i * rs.n(), all.begin() + (size_t)(i + 1) * rs.n())});
        return out;
    }
    // E's div2_ on plaintexts = rescalePow Z_{2^(k+1)} -> Z_{2^k} (Eval.hs:70-78): every coefficient must be even
    // ("since input is divisible by two, it doesn't matter which basis we use"); returns false when one is not
    bool div2(const PtCyc& a_, PtCyc& out) const {
        out = to(a_, Basis::Pow);
        bool even = true;
        for (auto& c : out.v) { if (c & 1) even = false; c = (c >> 1) % (out.p / 2); }
        out.p /= 2;
        return even;
    }

private:
    RingCache& rc_;
    std::vector<uint64_t> lp_;
};

// Cyc crtSet over Z_{p^e} (p = 2 here): Tensor crtSetDec of the p-free parts of (m, m') over F_p, lifted, taken to the Pow basis,
// raised to the p^(e-1)-th power (Hensel: an idempotent mod p becomes one mod p^e) and embedded into index m'.
inline std::vector<PtCyc> crtSet(PtOps& ops, uint32_t m, uint32_t mbig, int64_t prime, int e) {
    uint32_t mo = m, mbo = mbig;
    while (mo % prime == 0) mo /= (uint32_t)prime;
    while (mbo % prime == 0) mbo /= (uint32_t)prime;
    size_t count = 0;
    check(alch_crt_set_dec(mo, mbo, (uint32_t)prime, nullptr, &count), "alch_crt_set_dec");
    const uint32_t nb = totient(mbo);
    std::vector<int64_t> dec(count * (size_t)nb);
    check(alch_crt_set_dec(mo, mbo, (uint32_t)prime, dec.data(), &count), "alch_crt_set_dec");
    int64_t pe = 1;
    for (int i = 0; i < e; ++i) pe *= prime;
    std::vector<PtCyc> out;
    for (size_t k = 0; k < count; ++k) {
        PtCyc c{mbo, pe, Basis::Dec, std::vector<int64_t>(dec.begin() + k * nb, dec.begin() + (k + 1) * nb)};
        c = ops.to(c, Basis::Pow);
        for (int64_t t = 1; t < pe / prime; t *= prime) {             // ^(p^(e-1)) as e-1 p-th powers
            PtCyc b = c;
            for (int64_t i = 1; i < prime; ++i) c = ops.mul(c, b);
        }
        out.push_back(ops.embed(c, mbig));
    }
    return out;
}

// An E-linear function R -> S given by its values on the relative decoding basis of R / E (Lol: linearDec).
struct Linear {
    uint32_t e = 1, r = 1, s = 1;
    std::vector<PtCyc> ys;           // over S, Pow basis
};

// decToCRT (examples/Common.hs:65-75): the relative decoding basis of R / E to the first dim = phi(r)/phi(e) elements of the
// mod-p CRT set of S / E, E = R cap S.
inline Linear decToCRT(PtOps& ops, uint32_t r, uint32_t s, int64_t prime, int e_exp) {
    uint32_t a = r, b = s;
    while (b) { uint32_t t = a % b; a = b; b = t; }
    const uint32_t e = a;
    std::vector<PtCyc> crts = crtSet(ops, e, s, prime, e_exp);
    const uint32_t dim = totient(r) / totient(e);
    if (crts.size() < dim) throw std::runtime_error("decToCRT: the CRT set is smaller than the relative dimension (linearDec fails)");
    crts.resize(dim);
    return Linear{e, r, s, crts};
}

// evalLin f x = sum_i y_i * embed(coeffsDec(x)_i)
inline PtCyc evalLin(PtOps& ops, const Linear& f, const PtCyc& x) {
    std::vector<PtCyc> cs = ops.coeffsDec(x, f.e);
    if (cs.size() != f.ys.size()) throw std::runtime_error("evalLin: dimension mismatch");
    PtCyc acc;
    for (size_t i = 0; i < cs.size(); ++i) {
        PtCyc term = ops.mul(f.ys[i], ops.embed(cs[i], f.s));
        acc = i ? ops.add(acc, term) : term;
    }
    return acc;
}

}  // namespace gen
}  // namespace alchemy
