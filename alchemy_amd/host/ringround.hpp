// The ciphertext-op sequence of the reference's HomomRLWR example (BASELINE config 4) on resident batches, in C++ above the C ABI --
// the compiled twin of alchemy_amd/ringround.py (same calls, same seeds, same scratch discipline, therefore the same result words:
// tests/golden/batch_checksums.json's per-ciphertext checksums apply to both).
//
// `eval (pt2ct ringRound)` in examples/HomomRLWR.hs:45-59 runs: mulPublic, the five ring tunnels switch1..5 over H0' .. H5'
// (examples/Common.hs:49-54,78-95), then rescaleTreePow2 (Language/RescaleTree.hs:64-87): x (1 + x), eight leaves (addPublic + div2),
// 4 + 2 + 1 pairwise mul_ each followed by div2 -- with the limb counts PT2CT's type-level rules pick (alch_select_limbs) and the
// HomomRLWR moduli (examples/HomomRLWR.hs:37-43).  Residues and hints are synthetic (seeded): throughput and bit-exactness do not
// need valid encryptions; examples/homomrlwr_replay.cpp runs the same sequence with real keys to the example's PASS.
//
// One RingRound object belongs to ONE device: construct it with that device current (its rings bind to it).  The hint sources are
// separate from the hints so that a multi-GPU host can generate them on one rank and broadcast them (include/alchemy_rccl.h):
//     RingRound rr(B);  rr.fillSources();  /* or: receive them by alch_hint_broadcast */  rr.buildHints();  rr.run();
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/alchemy_hip.h"

namespace alchemy {
namespace ringround {

inline void check(int rc, const char* what) {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + alch_last_error());
}
inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((unsigned __int128)a * b) % q); }
inline uint64_t powmod(uint64_t b, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    for (b %= q; e; e >>= 1, b = mulmod(b, b, q)) if (e & 1) r = mulmod(r, b, q);
    return r;
}
inline uint64_t invmod(uint64_t a, uint64_t q) { return powmod(a % q, q - 2, q); }

static const uint64_t QS[6] = {1543651201, 689270401, 718099201, 720720001, 1556755201, 1567238401};   // Zqs order
static const uint32_t HP[6] = {11648, 29120, 43680, 54600, 27300, 20475};                            // H0' .. H5'
static const uint64_t P = 32;                                                                          // plaintext modulus 2^5

inline std::vector<uint64_t> moduli(int L) { return std::vector<uint64_t>(std::reverse_iterator<const uint64_t*>(QS + L), std::reverse_iterator<const uint64_t*>(QS)); }

struct Limbs { int lin, lh, lout; };

class RingRound {
public:
    // first: index of this object's first ciphertext in the whole seeded batch (a lane of `Lanes` below, a shard of a multi-GPU run)
    // dedicated_stream: the chain's one stream gets a hardware queue of its own (Lanes: two ordinary streams share a queue every other
    // time and then run one after the other -- include/alchemy_hip.h, "stream_dedicated")
    explicit RingRound(size_t batch, size_t first_ct = 0, bool dedicated_stream = false) : B(batch), first(first_ct), dedicated(dedicated_stream) {
        int p = 0;
        Limbs m[4], t[5];
        for (int i = 3; i >= 0; --i) check(alch_select_limbs(QS, 6, ALCH_OP_MUL, ALCH_GAD_TRIV, p, &m[i].lin, &m[i].lh, &m[i].lout, &p), "alch_select_limbs");
        for (int i = 4; i >= 0; --i) check(alch_select_limbs(QS, 6, ALCH_OP_TUNNEL, ALCH_GAD_TRIV, p, &t[i].lin, &t[i].lh, &t[i].lout, &p), "alch_select_limbs");
        muls.assign(m, m + 4);
        tuns.assign(t, t + 5);
        // hint sources: (linear function, key-switch hints) per tunnel, one quadratic hint per product level; seeds as in ringround.py
        for (int k = 0; k < 5; ++k) {
            alch_ring *rr = ring(HP[k], tuns[k].lh), *rs = ring(HP[k + 1], tuns[k].lh);
            uint32_t e = 0, d = 0;
            check(alch_tunnel_info(rr, rs, &e, &d), "alch_tunnel_info");
            drel.push_back(d);
            sources.push_back({alloc(rs, d), (uint64_t)(100 + k)});
            sources.push_back({alloc(rs, 2 * (size_t)d * tuns[k].lh), (uint64_t)(200 + k)});
        }
        for (const Limbs& l : muls) sources.push_back({alloc(ring(HP[5], l.lh), 2 * (size_t)l.lh), (uint64_t)(300 + l.lh)});
    }
    RingRound(const RingRound&) = delete;
    RingRound& operator=(const RingRound&) = delete;
    ~RingRound() {
        for (alch_tunnel* t : tunnels) alch_tunnel_free(t);
        for (alch_hint* h : quads) alch_hint_free(h);
        for (auto& s : sources) alch_buf_free(s.first);
        for (alch_buf* b : pool) alch_buf_free(b);
        for (auto& p : pubs) alch_buf_free(p.second);
        for (auto& r : rings) alch_ring_destroy(r.second);
    }

    // The seeded hint sources, in a fixed order (tunnel k: linear function, hints; then the four quadratic hints): what a
    // multi-GPU host broadcasts from the rank that generated them.
    std::vector<std::pair<alch_buf*, uint64_t>> sources;
    void fillSources() { for (auto& s : sources) check(alch_buf_fill_uniform(s.first, s.second), "alch_buf_fill_uniform"); }
    void buildHints() {
        for (int k = 0; k < 5; ++k) {
            alch_tunnel* t = nullptr;
            check(alch_tunnel_create(ring(HP[k], tuns[k].lh), ring(HP[k + 1], tuns[k].lh), ALCH_GAD_TRIV, sources[2 * k].first, sources[2 * k + 1].first, &t), "alch_tunnel_create");
            tunnels.push_back(t);
        }
        for (size_t i = 0; i < muls.size(); ++i) {
            alch_hint* h = nullptr;
            check(alch_hint_from_buf(ring(HP[5], muls[i].lh), ALCH_GAD_TRIV, sources[10 + i].first, &h), "alch_hint_from_buf");
            quads.push_back(h);
        }
    }

    void sync() { for (auto& r : rings) check(alch_sync(r.second), "alch_sync"); }

    // One pass over the batch; returns the result buffer (2B elements over H5' on one limb), owned by this object.
    alch_buf* run() {
        cursor = 0;
        const int L_0 = tuns[0].lin;
        alch_ring* r0 = ring(HP[0], L_0);
        if (!pubs.count("x")) {
            // word (e, j, k) of the whole batch is splitmix64(1 + ((e L + j) n + k)) mod q_j: a part that starts at ciphertext `first`
            // = element 2 first shifts the seed by the words in front of it
            uint32_t n0 = 0;
            check(alch_ring_n(r0, &n0, nullptr, nullptr), "alch_ring_n");
            pubs["x"] = alloc(r0, 2 * B);
            check(alch_buf_fill_uniform(pubs["x"], 1 + 2 * (uint64_t)first * (uint64_t)L_0 * n0), "fill");
        }
        if (!pubs.count("pub_msd")) {
            // mulPublic's public element times toMSD's per-limb scalar P^-1: folded into one element
            alch_buf* ps = alloc(r0, 1);
            std::vector<uint64_t> s;
            for (uint64_t q : moduli(L_0)) s.push_back(invmod(P, q));
            check(alch_buf_scale(ps, publicElem(r0, 2), 1, s.data()), "alch_buf_scale");
            pubs["pub_msd"] = ps;
        }
        alch_buf* x1 = scratch(r0, 2 * B);
        check(alch_buf_mul_public(x1, pubs["x"], pubs["pub_msd"], 0, 2 * B), "alch_buf_mul_public");
        alch_buf* cur = x1;
        for (int k = 0; k < 5; ++k) {
            // modSwitch_ (up) .: tunnel_ hint as one call; hops hand the ciphertexts over in the Pow basis
            alch_ring *rs = ring(HP[k + 1], tuns[k].lh), *ro = ring(HP[k + 1], tuns[k].lout);
            const unsigned pin = k > 0 ? ALCH_POW_IN : 0u, pout = k < 4 ? ALCH_POW_OUT : 0u;
            alch_buf* mid = scratch(rs, 2 * B);
            if (tuns[k].lout < tuns[k].lh) {
                check(alch_ct_tunnel(tunnels[k], cur, mid, B, nullptr, pin), "alch_ct_tunnel");
                alch_buf* dn = scratch(ro, 2 * B);
                check(alch_ct_mod_switch(mid, dn, B, pout), "alch_ct_mod_switch");
                mid = dn;
            } else {
                check(alch_ct_tunnel(tunnels[k], cur, mid, B, nullptr, pin | pout), "alch_ct_tunnel");
            }
            cur = mid;
        }
        // rescale tree on H5'
        const uint32_t m5 = HP[5];
        auto ones = [](int L) { return std::vector<uint64_t>((size_t)L, 1); };
        auto mulv = [](const std::vector<uint64_t>& a, const std::vector<uint64_t>& b, int L) {
            std::vector<uint64_t> o, q = moduli(L);
            for (int j = 0; j < L; ++j) o.push_back(mulmod(a[(size_t)j] % q[(size_t)j], b[(size_t)j] % q[(size_t)j], q[(size_t)j]));
            return o;
        };
        auto pinv = [](int L) { std::vector<uint64_t> o; for (uint64_t q : moduli(L)) o.push_back(invmod(P, q)); return o; };
        auto pres = [](int L) { std::vector<uint64_t> o; for (uint64_t q : moduli(L)) o.push_back(P % q); return o; };
        // mul_ of (a, pending pa) and (b, pending pb): the product's own toMSD scalar P^-1 and both pending scalars ride on s_pre
        auto product = [&](int level, alch_buf* a, const std::vector<uint64_t>& pa, alch_buf* b, const std::vector<uint64_t>& pb) {
            const int lin = muls[(size_t)level].lin;
            alch_buf* o = scratch(ring(m5, muls[(size_t)level].lout), 2 * B);
            std::vector<uint64_t> s = mulv(mulv(pa, pb, lin), pinv(lin), lin);
            check(alch_ct_mul_full(quads[(size_t)level], a, b, o, B, s.data(), 0), "alch_ct_mul_full");
            return o;
        };
        // toLSD, addPublic (div2_'s modSwitchPT is metadata): one fused pass
        auto plusPublic = [&](alch_buf* src, const std::vector<uint64_t>& ps, int L, uint64_t seed) {
            alch_ring* r = ring(m5, L);
            alch_buf* o = scratch(r, 2 * B);
            std::vector<uint64_t> s = mulv(ps, pres(L), L);
            check(alch_ct_add_public(o, src, B, s.data(), publicElem(r, seed), 0), "alch_ct_add_public");
            return o;
        };
        const int L0 = muls[0].lin, L1 = muls[1].lin;
        alch_buf* y = product(0, cur, pres(L0), plusPublic(cur, ones(L0), L0, 50), ones(L0));     // x_lsd = P x stays pending on x itself
        std::vector<alch_buf*> t;
        for (int i = 0; i < 8; ++i) t.push_back(plusPublic(y, ones(L1), L1, (uint64_t)(60 + i)));
        std::vector<uint64_t> pend = ones(L1);
        for (int level = 1; level <= 3; ++level) {
            std::vector<alch_buf*> nx;
            for (size_t i = 0; i + 1 < t.size(); i += 2) nx.push_back(product(level, t[i], pend, t[i + 1], pend));
            t = nx;
            pend.clear();
            for (uint64_t q : moduli(muls[(size_t)level].lout)) pend.push_back(invmod(2, q));      // div2_: toMSD scalar; the plaintext modulus halves (metadata)
        }
        check(alch_buf_scale(t[0], t[0], 2 * B, pend.data()), "alch_buf_scale");                    // the last div2's scalar: nothing follows that could absorb it
        return t[0];
    }

    const size_t B, first;
    const bool dedicated;
    std::vector<Limbs> muls, tuns;
    std::vector<uint32_t> drel;

private:
    alch_ring* ring(uint32_t m, int L) {
        auto key = std::make_pair(m, L);
        auto it = rings.find(key);
        if (it == rings.end()) {
            alch_ring* r = nullptr;
            std::vector<uint64_t> q = moduli(L);
            check(alch_ring_create(m, L, q.data(), &r), "alch_ring_create");
            // the op sequence is one dependency chain: all rings queue on the first ring's stream (no event per ring-to-ring hand-off)
            if (!rings.empty()) check(alch_ring_share_stream(r, rings.begin()->second), "alch_ring_share_stream");
            else if (dedicated) {
                const int rc = alch_ring_set_option(r, "stream_dedicated", 1);
                if (rc != ALCH_OK && rc != ALCH_E_UNSUPPORTED) check(rc, "alch_ring_set_option");   // UNSUPPORTED: the process holds its share of dedicated streams -- an ordinary one is still correct
            }
            it = rings.emplace(key, r).first;
        }
        return it->second;
    }
    static alch_buf* alloc(alch_ring* r, size_t count) {
        alch_buf* b = nullptr;
        check(alch_buf_alloc(r, count, &b), "alch_buf_alloc");
        return b;
    }
    // buffer pool: the first pass allocates, later passes replay the same sequence of requests
    alch_buf* scratch(alch_ring* r, size_t count) {
        const size_t i = cursor++;
        if (i == pool.size()) pool.push_back(alloc(r, count));
        return pool[i];
    }
    alch_buf* publicElem(alch_ring* r, uint64_t seed) {
        const std::string key = std::to_string((uintptr_t)r) + ":" + std::to_string(seed);
        if (!pubs.count(key)) { pubs[key] = alloc(r, 1); check(alch_buf_fill_uniform(pubs[key], seed), "fill"); }
        return pubs[key];
    }

    std::map<std::pair<uint32_t, int>, alch_ring*> rings;
    std::vector<alch_buf*> pool;
    size_t cursor = 0;
    std::map<std::string, alch_buf*> pubs;
    std::vector<alch_tunnel*> tunnels;
    std::vector<alch_hint*> quads;
};

// The same batch as `lanes` contiguous sub-batches, each a RingRound of its own on its own HIP stream (alchemy_amd/ringround.py's
// RingRoundLanes: two chains fill each other's memory-bound passes and thin last waves; measured 46.6 k -> 51.8 k pipelines/s at 1024
// ciphertexts).  The union of the lanes' results is word for word RingRound(batch)'s.
class Lanes {
public:
    Lanes(size_t batch, int lanes, size_t first_ct = 0) : B(batch) {
        if (lanes < 1) lanes = 1;
        if ((size_t)lanes > batch) lanes = (int)batch;
        size_t at = 0;
        for (int i = 0; i < lanes; ++i) {
            const size_t b = batch / (size_t)lanes + ((size_t)i < batch % (size_t)lanes ? 1 : 0);
            lane.emplace_back(new RingRound(b, first_ct + at, true));
            firsts.push_back(at);
            at += b;
        }
    }
    void fillSources() { for (auto& l : lane) l->fillSources(); }
    void buildHints() { for (auto& l : lane) l->buildHints(); }
    void sync() { for (auto& l : lane) l->sync(); }
    // one pass over the whole batch: the lanes' result buffers in batch order
    std::vector<alch_buf*> run() { std::vector<alch_buf*> o; for (auto& l : lane) o.push_back(l->run()); return o; }
    // checksum of the first `count` result ciphertexts, each part taken at its position in the whole batch (alch_buf_checksum_at)
    uint64_t checksum(const std::vector<alch_buf*>& outs, size_t count) const {
        uint64_t total = 0;
        for (size_t i = 0; i < lane.size(); ++i) {
            if (count <= firsts[i]) break;
            const size_t take = std::min(lane[i]->B, count - firsts[i]);
            uint64_t s = 0;
            check(alch_buf_checksum_at(outs[i], 0, 2 * take, 2 * firsts[i], &s), "alch_buf_checksum_at");
            total += s;
        }
        return total;
    }
    const size_t B;
    std::vector<std::unique_ptr<RingRound>> lane;
    std::vector<size_t> firsts;
};

}  // namespace ringround
}  // namespace alchemy
