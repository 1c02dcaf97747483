// Host-side mirror (C++, header-only) of lol-apps' SymmSHE for ANY cyclotomic index, above the C ABI of include/alchemy_hip.h and
// the Cyc layer of cycgen.hpp -- what ALCHEMY's evaluator calls when it runs examples/HomomRLWR.hs / examples/Tunnel.hs
// (Crypto/Alchemy/Interpreter/Eval.hs:120-134), plus the key / hint generation of Crypto/Alchemy/Interpreter/KeysHints.hs:
//
//   SK, genSK                                (KeysHints.hs:86-96:  svar = r / sqrt(phi(m')))
//   encrypt, decrypt, errorTermUnrestricted   (PT2CT.hs:84-99, Eval.hs:151-160)
//   toLSD / toMSD, (*), addPublic, mulPublic, modSwitchPT, modSwitch, keySwitchQuadCirc, tunnel   (Eval.hs:65-67,129-134)
//   ksQuadCircHint, tunnelHint               (KeysHints.hs:101-129)
//
// REPLAY AND TEST CODE, NOT A KEY GENERATOR: randomness comes from a caller-supplied std::mt19937_64 (reproducible runs) and uniform
// residues are drawn as `rng() % q` -- not a CSPRNG, with the 2^-33 modulo bias of a 31-bit q in a 64-bit draw.  A production host
// keeps Lol's own sampling (`getRandom`, Crypto/Alchemy/Interpreter/KeysHints.hs:86-129) and only the ring arithmetic comes here.
//
// Two forms of every ciphertext operation: the per-element form on host `CT` values (one Tensor call per step, as `eval` over
// `Tensor GT` would issue them), and -- for the ops PT2CT emits in bulk -- the batched device entry points on `DevBatch`
// (alch_ct_tunnel, alch_ct_mul_full, alch_ct_mod_switch, alch_buf_*), which carry the same (enc, k, l, p) metadata on the host.
#pragma once
#include "cycgen.hpp"

namespace alchemy {
namespace gen {

enum class Encoding { MSD, LSD };

struct CT {
    Encoding enc;
    int k;                     // accumulated power of g
    int64_t l;                 // accumulated Z_p scalar
    int64_t p;                 // plaintext modulus
    uint32_t m;                // plaintext index
    std::vector<Cyc> c;        // polynomial in the secret key over R'_q
};

struct SK {
    uint32_t mp;               // ciphertext index m'
    double svar;               // scaled variance r / sqrt(phi(m'))
    std::vector<int64_t> s;    // integer Pow coefficients
};

inline int64_t qprod_mod(const std::vector<uint64_t>& qs, int64_t p) {
    int64_t v = 1 % p;
    for (uint64_t q : qs) v = (int64_t)(((i128)v * (i128)(q % (uint64_t)p)) % p);
    return v;
}
inline int64_t neg_q_mod_p(const std::vector<uint64_t>& qs, int64_t p) { return (p - qprod_mod(qs, p)) % p; }

// per-limb scalars of the encoding changes
inline std::vector<uint64_t> lsdScalars(const std::vector<uint64_t>& qs, int64_t p) {       // MSD -> LSD: c * p
    std::vector<uint64_t> s;
    for (uint64_t q : qs) s.push_back((uint64_t)p % q);
    return s;
}
inline std::vector<uint64_t> msdScalars(const std::vector<uint64_t>& qs, int64_t p) {       // LSD -> MSD: c * p^-1
    std::vector<uint64_t> s;
    for (uint64_t q : qs) s.push_back(invmod_prime((uint64_t)p % q, q));
    return s;
}

inline CT toLSD(const CT& ct) {
    if (ct.enc == Encoding::LSD) return ct;
    const Ring& r = ct.c[0].ring();
    CT o{Encoding::LSD, ct.k, (int64_t)(((i128)ct.l * invmod_any(neg_q_mod_p(r.qs(), ct.p), ct.p)) % ct.p), ct.p, ct.m, {}};
    for (const Cyc& x : ct.c) o.c.push_back(x.scale(lsdScalars(r.qs(), ct.p)));
    return o;
}
inline CT toMSD(const CT& ct) {
    if (ct.enc == Encoding::MSD) return ct;
    const Ring& r = ct.c[0].ring();
    CT o{Encoding::MSD, ct.k, (int64_t)(((i128)ct.l * neg_q_mod_p(r.qs(), ct.p)) % ct.p), ct.p, ct.m, {}};
    for (const Cyc& x : ct.c) o.c.push_back(x.scale(msdScalars(r.qs(), ct.p)));
    return o;
}

// genSK: tweaked Gaussian of scaled variance svar = r / sqrt(phi(m')), rounded on the decoding basis
inline SK genSK(RingCache& rc, uint32_t mp, double r, std::mt19937_64& rng) {
    const double svar = r / std::sqrt((double)totient(mp));
    return SK{mp, svar, decToPowZ(rc, mp, errorRoundedDec(mp, svar, rng))};
}

inline Cyc uniformCRT(const Ring& r, std::mt19937_64& rng) {
    Cyc c(r, Basis::CRT);
    for (uint32_t k = 0; k < r.n(); ++k)
        for (int j = 0; j < r.L(); ++j) c.data()[(size_t)k * r.L() + j] = (int64_t)(rng() % r.qs()[j]);
    return c;
}

inline Cyc roundedError(RingCache& rc, const Ring& r, double svar, std::mt19937_64& rng) {   // errorRounded, reduced into r
    return Cyc::fromIntegers(r, decToPowZ(rc, r.m(), errorRoundedDec(r.m(), svar, rng)), Basis::Pow);
}

// SymmSHE encrypt: CT LSD 0 1 [e - c1 s, c1],  e = errorCoset svar (embed pt)
inline CT encrypt(RingCache& rc, PtOps& ops, const Ring& r, const SK& sk, const PtCyc& pt, std::mt19937_64& rng) {
    const PtCyc emb = ops.to(ops.embed(pt, r.m()), Basis::Dec);
    std::vector<int64_t> e = decToPowZ(rc, r.m(), errorCosetDec(r.m(), sk.svar, pt.p, emb.v, rng));
    Cyc s = Cyc::fromIntegers(r, sk.s).toCRT();
    Cyc c1 = uniformCRT(r, rng);
    Cyc c0 = Cyc::fromIntegers(r, e).toCRT() - c1 * s;
    return CT{Encoding::LSD, 0, 1, pt.p, pt.m, {c0, c1}};
}

// c(s) on the decoding basis for the LSD form: the object Lol's errorTermUnrestricted lifts (Eval.hs:151-160)
inline Cyc errorTermDec(const SK& sk, const CT& ct_) {
    CT ct = toLSD(ct_);
    const Ring& r = ct.c[0].ring();
    Cyc s = Cyc::fromIntegers(r, sk.s).toCRT();
    Cyc acc(r, Basis::CRT);
    for (size_t i = ct.c.size(); i-- > 0;) acc = acc * s + ct.c[i];                          // Horner in S
    return acc.toDec();
}

// decrypt: l * twace(g^-k (liftDec(c(s)) mod p)); false when a divG fails.  error_rate (optional): max |liftDec c(s)| / q, the
// figure the ERW interpreter logs after every ciphertext op (ErrorRateWriter.hs:70-75).
inline bool decrypt(RingCache& rc, PtOps& ops, const SK& sk, const CT& ct, PtCyc& out, double* error_rate = nullptr) {
    const Ring& r = ct.c[0].ring();
    PtCyc x{r.m(), ct.p, Basis::Dec, errorTermDec(sk, ct).liftModP(ct.p, error_rate)};
    const CT lsd = toLSD(ct);
    const Ring& zp = rc.get(r.m(), {(uint64_t)ct.p}, false);
    for (int i = 0; i < ct.k; ++i) {
        int rcg = alch_divg_dec(zp.handle(), x.v.data());
        check(rcg, "alch_divg_dec (plaintext)");
        if (rcg == ALCH_NOT_DIVISIBLE) return false;
    }
    const Ring& zs = rc.get(ct.m, {(uint64_t)ct.p}, false);
    out = PtCyc{ct.m, ct.p, Basis::Dec, std::vector<int64_t>(zs.n())};
    check(alch_twace_pow_dec(zs.handle(), zp.handle(), x.v.data(), out.v.data()), "alch_twace_pow_dec (plaintext)");
    for (auto& v : out.v) v = (int64_t)(((i128)v * lsd.l) % ct.p);
    out = ops.to(out, Basis::Pow);
    return true;
}

// embed(reduce(lift a)): a plaintext ring element as an element of the ciphertext ring (mulPublic / addPublic / tunnelHint)
inline Cyc liftEmbed(PtOps& ops, const Ring& r, const PtCyc& a) {
    const PtCyc e = ops.embed(a, r.m());
    std::vector<int64_t> z(e.v.size());
    for (size_t i = 0; i < z.size(); ++i) z[i] = centred(e.v[i], e.p);
    return Cyc::fromIntegers(r, z, Basis::Pow);
}

// SymmSHE mulPublic a ct: every coefficient times embed(reduce(liftPow a))
inline CT mulPublic(PtOps& ops, const PtCyc& a, const CT& ct) {
    const Cyc pub = liftEmbed(ops, ct.c[0].ring(), a).toCRT();
    CT o{ct.enc, ct.k, ct.l, ct.p, ct.m, {}};
    for (const Cyc& x : ct.c) o.c.push_back(x * pub);
    return o;
}

// The ring element addPublic adds to c0 of the LSD form: mulG^k (embed (reduce (liftPow (l^-1 b))))
inline Cyc addPublicTerm(PtOps& ops, const Ring& r, const PtCyc& b, int64_t l_lsd, int k) {
    PtCyc t = ops.to(b, Basis::Pow);
    const int64_t linv = invmod_any(l_lsd, b.p);
    for (auto& v : t.v) v = (int64_t)(((i128)v * linv) % b.p);
    Cyc x = liftEmbed(ops, r, t);
    for (int i = 0; i < k; ++i) x = x.mulG();
    return x;
}
inline CT addPublic(PtOps& ops, const PtCyc& b, const CT& ct_) {
    CT ct = toLSD(ct_);
    ct.c[0] = ct.c[0] + addPublicTerm(ops, ct.c[0].ring(), b, ct.l, ct.k).to(ct.c[0].basis());
    return ct;
}

// SymmSHE absorbGFactors: what `tunnel` runs first when k > 0 -- every component times reduce(liftPow d), d = g^-k in R'_p
// (`iterate divG one !! k`), then k = 0.  On this ABI: alch_divg_pow on a ring without CRT over Z_p, a centred lift, `reduce`,
// one ring product per component.
inline CT absorbGFactors(RingCache& rc, const CT& ct) {
    if (ct.k == 0) return ct;
    const Ring& r = ct.c[0].ring();
    const Ring& zp = rc.get(r.m(), {(uint64_t)ct.p}, false);
    std::vector<int64_t> d(zp.n(), 0);
    d[0] = 1;
    for (int i = 0; i < ct.k; ++i) {
        int rcg = alch_divg_pow(zp.handle(), d.data());
        check(rcg, "alch_divg_pow (plaintext)");
        if (rcg == ALCH_NOT_DIVISIBLE) throw std::runtime_error("absorbGFactors: g is not a unit modulo p");
    }
    for (auto& v : d) v = centred(v, ct.p);
    const Cyc rep = Cyc::fromIntegers(r, d).toCRT();
    CT o{ct.enc, 0, ct.l, ct.p, ct.m, {}};
    for (const Cyc& x : ct.c) o.c.push_back(x * rep);
    return o;
}

// modSwitchPT (PT2CT's div2_, PT2CT.hs:179-189): MSD form, plaintext modulus p -> p' | p, ring elements unchanged
inline CT modSwitchPT(const CT& ct_, int64_t p_new) {
    CT ct = toMSD(ct_);
    ct.l %= p_new;
    ct.p = p_new;
    return ct;
}

// SymmSHE (*)
inline CT operator*(const CT& a_, const CT& b_) {
    CT a = toLSD(a_), b = toLSD(b_);
    const Ring& r = a.c[0].ring();
    std::vector<Cyc> out(a.c.size() + b.c.size() - 1, Cyc(r, Basis::CRT));
    for (size_t i = 0; i < a.c.size(); ++i)
        for (size_t j = 0; j < b.c.size(); ++j) out[i + j] = out[i + j] + a.c[i] * b.c[j];
    for (Cyc& x : out) x = x.mulG();
    return CT{Encoding::LSD, a.k + b.k + 1, (int64_t)(((i128)a.l * b.l) % a.p), a.p, a.m, out};
}

// ---- hints -------------------------------------------------------------------------------------------------------
struct KSHint { std::vector<std::pair<Cyc, Cyc>> h; };          // one (b, a) per gadget entry, CRT basis; b + a s = g_t v + e

// The gadget vector as per-limb scalars (Lol Gadget for an RNS product: the per-limb gadgets placed in their own limb, zeros
// elsewhere): TrivGad = the unit vectors; BaseBGad 2 = 2^t on limb i for t < ceil(log2 q_i), limb 0's entries first.
inline std::vector<std::vector<uint64_t>> gadgetVector(const Ring& r, int gadget) {
    std::vector<std::vector<uint64_t>> g;
    for (int i = 0; i < r.L(); ++i) {
        int k = 1;
        if (gadget == ALCH_GAD_BASE2) { k = 0; for (uint64_t v = 1; v < r.qs()[i]; v <<= 1) ++k; }
        for (int t = 0; t < k; ++t) {
            std::vector<uint64_t> e(r.L(), 0);
            e[i] = powmod(2, (uint64_t)t, r.qs()[i]);
            g.push_back(e);
        }
    }
    return g;
}

// ksHint skout v: LWE samples under skout hiding g_t * v for every entry g_t of the gadget
inline KSHint ksHint(RingCache& rc, const Ring& r, const SK& skout, const Cyc& v_crt, std::mt19937_64& rng, int gadget = ALCH_GAD_TRIV) {
    const Cyc s = Cyc::fromIntegers(r, skout.s).toCRT();
    KSHint hint;
    for (const auto& gt : gadgetVector(r, gadget)) {
        const Cyc e = roundedError(rc, r, skout.svar, rng).toCRT();
        const Cyc a = uniformCRT(r, rng);
        hint.h.emplace_back(v_crt.scale(gt) + e - a * s, a);
    }
    return hint;
}
inline KSHint ksQuadCircHint(RingCache& rc, const Ring& r, const SK& sk, std::mt19937_64& rng) {
    const Cyc s = Cyc::fromIntegers(r, sk.s).toCRT();
    return ksHint(rc, r, sk, s * s, rng);
}

// keySwitchQuadCirc (per element)
inline CT keySwitchQuadCirc(const KSHint& hint, const CT& ct_) {
    CT ct = toMSD(ct_);
    if (ct.c.size() < 3) return ct;
    std::vector<Cyc> digs = ct.c[2].decomposeTrivReduced();
    Cyc c0 = ct.c[0].toCRT(), c1 = ct.c[1].toCRT();
    for (size_t i = 0; i < digs.size(); ++i) { c0 = c0 + digs[i] * hint.h[i].first; c1 = c1 + digs[i] * hint.h[i].second; }
    return CT{Encoding::MSD, ct.k, ct.l, ct.p, ct.m, {c0, c1}};
}

// tunnelHint f skout skin (KeysHints.hs:120-129): the E'-linear extension f' of f (values embedded into S', lifted, reduced mod q)
// and, for every element p_i of the relative powerful basis of R'/E', a key-switch hint for f'(s_in p_i) under s_out.
struct TunnelHint {
    uint32_t ep, rp, sp;
    int gadget = ALCH_GAD_TRIV;
    std::vector<Cyc> lin;                       // d_rel values f'(d_i) over S'_q, CRT basis
    std::vector<KSHint> ks;                     // d_rel hints
};

// f'(x) for x over R'_q: sum_i y'_i * embed(coeffsDec(x)_i)   (Lol evalLin on the extended function)
inline Cyc evalLinExt(const std::vector<Cyc>& lin_crt, const Ring& re, const Ring& rs, const Cyc& x) {
    std::vector<Cyc> cs = x.coeffs(re, Basis::Dec);
    Cyc acc(rs, Basis::CRT);
    for (size_t i = 0; i < cs.size(); ++i) acc = acc + lin_crt[i] * cs[i].embed(rs);       // embedDec, then to CRT inside (*)
    return acc;
}

inline TunnelHint tunnelHint(RingCache& rc, PtOps& ops, const Linear& f, uint32_t rp, uint32_t sp, const std::vector<uint64_t>& qs,
                             const SK& skout, const SK& skin, std::mt19937_64& rng, int gadget = ALCH_GAD_TRIV) {
    uint32_t a = rp, b = sp;
    while (b) { uint32_t t = a % b; a = b; b = t; }
    const uint32_t ep = a;
    const Ring &re = rc.get(ep, qs), &rr = rc.get(rp, qs), &rs = rc.get(sp, qs);
    uint32_t chk_e = 0, d_rel = 0;
    check(alch_tunnel_info(rr.handle(), rs.handle(), &chk_e, &d_rel), "alch_tunnel_info");
    if (d_rel != f.ys.size() || chk_e != ep) throw std::runtime_error("tunnelHint: the linear function does not match the rings");
    TunnelHint h{ep, rp, sp, gadget, {}, {}};
    for (const PtCyc& y : f.ys) h.lin.push_back(liftEmbed(ops, rs, y).toCRT());
    // relative powerful basis of R'/E': unit vectors at the positions table ALCH_EXT_COEFFS lists first
    size_t len = (size_t)d_rel * re.n();
    std::vector<int32_t> tab(len);
    check(alch_ext_table(ep, rp, ALCH_EXT_COEFFS, tab.data(), &len), "alch_ext_table");
    const Cyc sin = Cyc::fromIntegers(rr, skin.s).toCRT();
    for (uint32_t i = 0; i < d_rel; ++i) {
        std::vector<int64_t> unit(rr.n(), 0);
        unit[(size_t)tab[(size_t)i * re.n()]] = 1;
        const Cyc x = sin * Cyc::fromIntegers(rr, unit);                                   // s_in p_i over R'_q
        h.ks.push_back(ksHint(rc, rs, skout, evalLinExt(h.lin, re, rs, x), rng, gadget));
    }
    return h;
}

// SymmSHE tunnel (per element): linear ciphertext, k = 0, MSD
inline CT tunnel(RingCache& rc, const TunnelHint& h, const CT& ct_, uint32_t m_out) {
    CT ct = toMSD(absorbGFactors(rc, ct_));
    if (ct.c.size() != 2) throw std::runtime_error("tunnel: linear ciphertexts");
    if (h.gadget != ALCH_GAD_TRIV) throw std::runtime_error("tunnel (per element): TrivGad hints; BaseBGad hints run on the batched entry point");
    const Ring& rr = ct.c[0].ring();
    const Ring &re = rc.get(h.ep, rr.qs()), &rs = rc.get(h.sp, rr.qs());
    Cyc c0 = evalLinExt(h.lin, re, rs, ct.c[0]);
    Cyc c1(rs, Basis::CRT);
    std::vector<Cyc> parts = ct.c[1].coeffs(re, Basis::Pow);
    for (size_t i = 0; i < parts.size(); ++i) {
        std::vector<Cyc> digs = parts[i].embed(rs).decomposeTrivReduced();
        for (size_t t = 0; t < digs.size(); ++t) { c0 = c0 + digs[t] * h.ks[i].h[t].first; c1 = c1 + digs[t] * h.ks[i].h[t].second; }
    }
    return CT{Encoding::MSD, 0, ct.l, ct.p, m_out, {c0, c1}};
}

// ---- device-resident batches ---------------------------------------------------------------------------------------
// A batch of linear ciphertexts sharing their metadata, resident in HBM (elements (2b, 2b+1) = (c0, c1) of ciphertext b).
struct DevBatch {
    const Ring* ring = nullptr;
    alch_buf* buf = nullptr;
    size_t B = 0;
    Encoding enc = Encoding::LSD;
    int k = 0;
    int64_t l = 1, p = 2;
    uint32_t m = 1;
    Basis basis = Basis::CRT;

    DevBatch() {}
    DevBatch(const Ring& r, size_t batch) : ring(&r), B(batch) { check(alch_buf_alloc(r.handle(), 2 * batch, &buf), "alch_buf_alloc"); }
    DevBatch(const DevBatch&) = delete;
    DevBatch& operator=(const DevBatch&) = delete;
    DevBatch(DevBatch&& o) noexcept { *this = std::move(o); }
    DevBatch& operator=(DevBatch&& o) noexcept {
        if (this != &o) {
            if (buf) alch_buf_free(buf);
            ring = o.ring; buf = o.buf; B = o.B; enc = o.enc; k = o.k; l = o.l; p = o.p; m = o.m; basis = o.basis;
            o.buf = nullptr;
        }
        return *this;
    }
    ~DevBatch() { if (buf) alch_buf_free(buf); }
    void meta(const DevBatch& o) { enc = o.enc; k = o.k; l = o.l; p = o.p; m = o.m; basis = o.basis; }

    void upload(size_t b, const CT& ct) {
        if (ct.c.size() != 2) throw std::runtime_error("DevBatch: linear ciphertexts");
        for (int c = 0; c < 2; ++c) {
            const Cyc x = ct.c[c].to(basis);
            if (x.onDeviceOnly()) check(alch_buf_copy(buf, 2 * b + c, x.dev(), 0, 1), "alch_buf_copy");      // already in HBM
            else check(alch_buf_upload(buf, 2 * b + c, 1, x.data().data()), "alch_buf_upload");
        }
    }
    CT download(size_t b) const {
        CT ct{enc, k, l, p, m, {Cyc(*ring, basis), Cyc(*ring, basis)}};
        for (int c = 0; c < 2; ++c) check(alch_buf_download(buf, 2 * b + c, 1, ct.c[c].data().data()), "alch_buf_download");
        return ct;
    }
    // encoding changes: a per-limb scalar on the device
    void toLSD() {
        if (enc == Encoding::LSD) return;
        std::vector<uint64_t> s = lsdScalars(ring->qs(), p);
        check(alch_buf_scale(buf, buf, 2 * B, s.data()), "alch_buf_scale");
        l = (int64_t)(((i128)l * invmod_any(neg_q_mod_p(ring->qs(), p), p)) % p);
        enc = Encoding::LSD;
    }
    void toMSD() {
        if (enc == Encoding::MSD) return;
        std::vector<uint64_t> s = msdScalars(ring->qs(), p);
        check(alch_buf_scale(buf, buf, 2 * B, s.data()), "alch_buf_scale");
        l = (int64_t)(((i128)l * neg_q_mod_p(ring->qs(), p)) % p);
        enc = Encoding::MSD;
    }
};

// addPublic b on every ciphertext of the batch (LSD form)
inline void addPublicBatch(PtOps& ops, DevBatch& x, const PtCyc& b) {
    x.toLSD();
    const Cyc term = addPublicTerm(ops, *x.ring, b, x.l, x.k).to(x.basis);
    alch_buf* pub = nullptr;
    check(alch_buf_alloc(x.ring->handle(), 1, &pub), "alch_buf_alloc");
    check(alch_buf_upload(pub, 0, 1, term.data().data()), "alch_buf_upload");
    check(alch_buf_add_public(x.buf, pub, 0, x.B), "alch_buf_add_public");
    alch_buf_free(pub);
}

// modSwitchPT on the batch (div2_): toMSD, then only metadata
inline void modSwitchPTBatch(DevBatch& x, int64_t p_new) {
    x.toMSD();
    x.l %= p_new;
    x.p = p_new;
}

// Device-resident hint of keySwitchQuadCirc
struct DevQuadHint {
    alch_hint* h = nullptr;
    const Ring* ring = nullptr;
    DevQuadHint() {}
    DevQuadHint(const Ring& r, const KSHint& hint) : ring(&r) {
        alch_buf* b = nullptr;
        check(alch_buf_alloc(r.handle(), 2 * (size_t)r.L(), &b), "alch_buf_alloc");
        for (int i = 0; i < r.L(); ++i) {
            check(alch_buf_upload(b, 2 * i, 1, hint.h[i].first.toCRT().data().data()), "alch_buf_upload");
            check(alch_buf_upload(b, 2 * i + 1, 1, hint.h[i].second.toCRT().data().data()), "alch_buf_upload");
        }
        check(alch_hint_from_buf(r.handle(), ALCH_GAD_TRIV, b, &h), "alch_hint_from_buf");
        alch_buf_free(b);
    }
    DevQuadHint(const DevQuadHint&) = delete;
    DevQuadHint& operator=(const DevQuadHint&) = delete;
    DevQuadHint(DevQuadHint&& o) noexcept : h(o.h), ring(o.ring) { o.h = nullptr; }
    ~DevQuadHint() { if (h) alch_hint_free(h); }
};

// PT2CT's mul_ on batches (PT2CT.hs:160-177): modSwitch_ .: keySwitchQuad_ hint .: modSwitch_ $: (x *: y) as one alch_ct_mul_full
// call; operands on rin (CRT basis, any encoding), hint on its ring, result on rout (MSD, CRT basis).
inline DevBatch mulFullBatch(const DevQuadHint& hint, DevBatch& x, DevBatch& y, const Ring& rout) {
    if (x.ring != y.ring || x.B != y.B || x.p != y.p) throw std::runtime_error("mulFullBatch: operand mismatch");
    x.toLSD(); y.toLSD();                                     // (*) multiplies LSD forms
    const Ring& rin = *x.ring;
    // the product's toMSD (at the operands' modulus) rides on the tensor product as a per-limb scalar
    std::vector<uint64_t> s = msdScalars(rin.qs(), x.p);
    DevBatch out(rout, x.B);
    check(alch_ct_mul_full(hint.h, x.buf, y.buf, out.buf, x.B, s.data(), 0), "alch_ct_mul_full");
    out.enc = Encoding::MSD;
    out.k = x.k + y.k + 1;
    out.l = (int64_t)(((i128)x.l * y.l % x.p) * neg_q_mod_p(rin.qs(), x.p) % x.p);
    out.p = x.p; out.m = x.m; out.basis = Basis::CRT;
    return out;
}

// Device-resident tunnel (linear function + hints)
struct DevTunnel {
    alch_tunnel* t = nullptr;
    const Ring *rr = nullptr, *rs = nullptr;
    DevTunnel() {}
    DevTunnel(const Ring& r, const Ring& s, const TunnelHint& h) : rr(&r), rs(&s) {
        const size_t d = h.lin.size(), D = h.ks.empty() ? 0 : h.ks[0].h.size();
        alch_buf *lin = nullptr, *ks = nullptr;
        check(alch_buf_alloc(s.handle(), d, &lin), "alch_buf_alloc");
        check(alch_buf_alloc(s.handle(), 2 * d * D, &ks), "alch_buf_alloc");
        for (size_t i = 0; i < d; ++i) {
            check(alch_buf_upload(lin, i, 1, h.lin[i].toCRT().data().data()), "alch_buf_upload");
            for (size_t t2 = 0; t2 < D; ++t2) {
                check(alch_buf_upload(ks, (i * D + t2) * 2, 1, h.ks[i].h[t2].first.toCRT().data().data()), "alch_buf_upload");
                check(alch_buf_upload(ks, (i * D + t2) * 2 + 1, 1, h.ks[i].h[t2].second.toCRT().data().data()), "alch_buf_upload");
            }
        }
        check(alch_tunnel_create(r.handle(), s.handle(), h.gadget, lin, ks, &t), "alch_tunnel_create");
        alch_buf_free(lin);
        alch_buf_free(ks);
    }
    DevTunnel(const DevTunnel&) = delete;
    DevTunnel& operator=(const DevTunnel&) = delete;
    DevTunnel(DevTunnel&& o) noexcept : t(o.t), rr(o.rr), rs(o.rs) { o.t = nullptr; }
    ~DevTunnel() { if (t) alch_tunnel_free(t); }
};

// PT2CT's linearCyc_ on batches (PT2CT.hs:207-229): modSwitch_ .: tunnel_ hint .: modSwitch_.  x lives on the last limbs of the
// tunnel's R' ring (the leading modSwitch up is part of alch_ct_tunnel); the result is rescaled down to rout.
inline DevBatch tunnelBatch(const DevTunnel& tun, DevBatch& x, const Ring& rout, uint32_t m_out) {
    if (x.k != 0) throw std::runtime_error("tunnelBatch: k must be 0");
    if (x.ring->L() > tun.rs->L()) {
        // BaseBGad hints may sit on FEWER limbs than the input (KSPNoise, PT2CT.hs:140): the leading modSwitch goes down
        x.toMSD();
        DevBatch dn(*tun.rr, x.B);
        check(alch_ct_mod_switch(x.buf, dn.buf, x.B, x.basis == Basis::Pow ? ALCH_POW_IN : 0u), "alch_ct_mod_switch");
        dn.meta(x);
        dn.basis = Basis::CRT;
        return tunnelBatch(tun, dn, rout, m_out);
    }
    // toMSD rides on the call as its per-limb scalar (indexed by the tunnel ring's limbs; the input holds its last limbs)
    const Ring& rs = *tun.rs;
    const int dup = rs.L() - x.ring->L();
    std::vector<uint64_t> s(rs.L(), 1);
    int64_t l = x.l;
    if (x.enc == Encoding::LSD) {
        std::vector<uint64_t> ms = msdScalars(x.ring->qs(), x.p);
        for (int j = 0; j < x.ring->L(); ++j) s[dup + j] = ms[j];
        l = (int64_t)(((i128)l * neg_q_mod_p(x.ring->qs(), x.p)) % x.p);
    }
    const unsigned fin = x.basis == Basis::Pow ? ALCH_POW_IN : 0u;
    DevBatch mid(rs, x.B);
    check(alch_ct_tunnel(tun.t, x.buf, mid.buf, x.B, s.data(), fin), "alch_ct_tunnel");
    mid.enc = Encoding::MSD; mid.k = 0; mid.l = l; mid.p = x.p; mid.m = m_out; mid.basis = Basis::CRT;
    if (rout.L() == rs.L()) return mid;
    DevBatch out(rout, x.B);
    check(alch_ct_mod_switch(mid.buf, out.buf, x.B, 0), "alch_ct_mod_switch");
    out.meta(mid);
    return out;
}

// SymmSHE modSwitch on one host ciphertext (Eval.hs:130): toMSD, then Rescale up / down by whole limbs through alch_ct_mod_switch
// (c0 on the Dec basis, c1 on the Pow basis when going down).
inline CT modSwitch(const CT& ct_, const Ring& dst) {
    const CT ct = toMSD(ct_);
    if (&ct.c[0].ring() == &dst) return ct;
    if (mode() == Mode::Resident) {
        // the two components side by side in one pooled buffer, the result handed on as two aliases: nothing leaves HBM
        auto in = DevElem::make(ct.c[0].ring().handle(), 2), out = DevElem::make(dst.handle(), 2);
        for (int c = 0; c < 2; ++c) check(alch_buf_copy(in->b, (size_t)c, ct.c[c].toCRT().dev(), 0, 1), "alch_buf_copy");
        check(alch_ct_mod_switch(in->b, out->b, 1, 0), "alch_ct_mod_switch");
        return CT{Encoding::MSD, ct.k, ct.l, ct.p, ct.m,
                  {Cyc::onDevice(dst, Basis::CRT, DevElem::view(out, 0)), Cyc::onDevice(dst, Basis::CRT, DevElem::view(out, 1))}};
    }
    // HostBuffers, and ResidentZipHost (Lol's rescale is fmapT-class work: the components cross to the host and back)
    DevBatch in(ct.c[0].ring(), 1), out(dst, 1);
    in.basis = Basis::CRT;
    for (int c = 0; c < 2; ++c) check(alch_buf_upload(in.buf, (size_t)c, 1, ct.c[c].toCRT().data().data()), "alch_buf_upload");
    check(alch_ct_mod_switch(in.buf, out.buf, 1, 0), "alch_ct_mod_switch");
    out.enc = Encoding::MSD; out.k = ct.k; out.l = ct.l; out.p = ct.p; out.m = ct.m; out.basis = Basis::CRT;
    return out.download(0);
}

}  // namespace gen
}  // namespace alchemy
