// Replays examples/HomomRLWR.hs (reference) on the MI355X backend at the reference's own parameters:
//
//   ringRound = rescaleTreePow2_ @P5 .: switch5                               (examples/HomomRLWR.hs:45-50)
//   f         = eval (pt2ct ringRound) . (`mulPublic` enc(s))                   (:52-59, Gaussian parameter 5.0)
//   main      : decrypt (f a) == eval ringRound (s * a)  ->  PASS / FAIL        (:62-71)
//
// plaintext indices H0 .. H5 and ciphertext indices H0' .. H5' of examples/Common.hs:38-54, the six moduli of
// examples/HomomRLWR.hs:37-43, TrivGad, plaintext modulus 2^5 (K = P5), the linear functions decToCRT @H_k (Common.hs:65-95).
// What `pt2ct` resolves at the type level is resolved here by alch_select_limbs (tunnels 5/6/5 .. 5/5/4, products 4/5/3, 3/4/2,
// 2/3/1, 1/2/1).  The op sequence runs on a batch of B ciphertexts through the batched device entry points (alch_ct_tunnel,
// alch_ct_mul_full, alch_buf_*), with real keys, hints and encryptions built by the host layer of alchemy_amd/host/*.hpp; after every
// stage ciphertext 0 is decrypted with the stage's key and compared with the plaintext computation, and its error rate
// max |c(s)| / q -- what the ERW interpreter logs (Crypto/Alchemy/Interpreter/ErrorRateWriter.hs:70-75, Eval.hs:151-160) -- is printed.
//
//   homomrlwr_replay [batch] [--seed N] [--dump DIR] [--per-element] [--per-element-resident] [--per-element-zip-host]
//                    [--host-mode buffers|resident] [--var-scale F] [--quiet-stages]
// --dump writes the final ciphertexts, the H5' key and the expected plaintexts for an independent decryption by the oracle
// (tests/test_gpu_homomrlwr_full.py).  The --per-element* flags also run the first hop (modSwitch . tunnel hint . modSwitch) through
// the per-Tensor-call path -- what `eval` over `instance Tensor GT` issues, one C-ABI call per Tensor method -- in the three
// representations of a ring element (alchemy_amd/host/cycgen.hpp): host buffers staged through the GPU per call, device-resident
// elements, device-resident with the pointwise operations on the host (unchanged Lol's zipWithT), and compare each with the batched
// result bit for bit.  --host-mode picks the representation the setup (keys, hints), the plaintext side and the decryptions use
// (default resident).  --var-scale multiplies the Gaussian parameter r = 5.0 of examples/HomomRLWR.hs:56 (svar = r / sqrt(phi(m'))):
// the noise-margin experiment of DESIGN.md section 5.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "../alchemy_amd/host/symmshe_gen.hpp"

using namespace alchemy::gen;

static const std::vector<uint64_t> QS = {1543651201, 689270401, 718099201, 720720001, 1556755201, 1567238401};   // Zqs order
static const uint32_t H[6] = {128, 448, 2912, 3640, 5460, 4095};
static const uint32_t HP[6] = {11648, 29120, 43680, 54600, 27300, 20475};
static const int64_t P = 32;                                       // plaintext modulus 2^5
static const int K_EXP = 5;

static std::vector<uint64_t> moduli(int L) { return std::vector<uint64_t>(QS.rend() - L, QS.rend()); }   // last-taken modulus outermost

struct Limbs { int lin, lh, lout; };

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void dump_i64(const std::string& path, const std::vector<int64_t>& v) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(int64_t)));
}

int main(int argc, char** argv) {
    size_t B = 4;
    uint64_t seed = 2026;
    std::string dump;
    bool per_element = false, per_resident = false, per_ziphost = false, quiet = false;
    double var_scale = 1.0;
    Mode host_mode = Mode::Resident;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = strtoull(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
        else if (!strcmp(argv[i], "--per-element")) per_element = true;
        else if (!strcmp(argv[i], "--per-element-resident")) per_resident = true;
        else if (!strcmp(argv[i], "--per-element-zip-host")) per_ziphost = true;
        else if (!strcmp(argv[i], "--quiet-stages")) quiet = true;
        else if (!strcmp(argv[i], "--var-scale") && i + 1 < argc) var_scale = atof(argv[++i]);
        else if (!strcmp(argv[i], "--host-mode") && i + 1 < argc) host_mode = !strcmp(argv[++i], "buffers") ? Mode::HostBuffers : Mode::Resident;
        else B = (size_t)atoi(argv[i]);
    }
    const double R_GAUSS = 5.0 * var_scale;                       // examples/HomomRLWR.hs:56
    mode() = host_mode;
    try {
        std::mt19937_64 rng(seed);
        RingCache rc;
        PtOps ops(rc, {QS[4], QS[5]});
        // ---- PT2CT's limb counts, resolved backwards from the output pNoise 0
        Limbs muls[4], tuns[5];
        int pn = 0;
        for (int i = 3; i >= 0; --i) check(alch_select_limbs(QS.data(), 6, ALCH_OP_MUL, ALCH_GAD_TRIV, pn, &muls[i].lin, &muls[i].lh, &muls[i].lout, &pn), "alch_select_limbs");
        for (int i = 4; i >= 0; --i) check(alch_select_limbs(QS.data(), 6, ALCH_OP_TUNNEL, ALCH_GAD_TRIV, pn, &tuns[i].lin, &tuns[i].lh, &tuns[i].lout, &pn), "alch_select_limbs");
        printf("limbs (in/hint/out): tunnels");
        for (auto& t : tuns) printf(" %d/%d/%d", t.lin, t.lh, t.lout);
        printf("; mul_");
        for (auto& t : muls) printf(" %d/%d/%d", t.lin, t.lh, t.lout);
        printf("\n");

        // ---- "Generating function": keys, linear functions, hints (examples/HomomRLWR.hs:52-59,64)
        double t0 = now();
        std::vector<SK> sk;
        for (int k = 0; k < 6; ++k) sk.push_back(genSK(rc, HP[k], R_GAUSS, rng));
        std::vector<Linear> lin;
        for (int k = 0; k < 5; ++k) lin.push_back(decToCRT(ops, H[k], H[k + 1], 2, K_EXP));
        std::vector<DevTunnel> dtun;
        std::vector<TunnelHint> thints;
        for (int k = 0; k < 5; ++k) {
            const std::vector<uint64_t> qs = moduli(tuns[k].lh);
            thints.push_back(tunnelHint(rc, ops, lin[k], HP[k], HP[k + 1], qs, sk[k + 1], sk[k], rng));
            dtun.emplace_back(rc.get(HP[k], qs), rc.get(HP[k + 1], qs), thints.back());
        }
        std::vector<DevQuadHint> dquad;
        for (int i = 0; i < 4; ++i) {
            const Ring& rh = rc.get(HP[5], moduli(muls[i].lh));
            dquad.emplace_back(rh, ksQuadCircHint(rc, rh, sk[5], rng));
        }
        // the RLWR secret s (encrypted) and the public a's
        auto randPt = [&](uint32_t m) { PtCyc x{m, P, Basis::Pow, std::vector<int64_t>(totient(m))}; for (auto& v : x.v) v = (int64_t)(rng() % (uint64_t)P); return x; };
        const PtCyc s = randPt(H[0]);
        const Ring& r0 = rc.get(HP[0], moduli(tuns[0].lin));
        const CT enc_s = encrypt(rc, ops, r0, sk[0], s, rng);
        printf("Generating function... %.2f s (6 keys, 5 linear functions with tunnel hints, 4 quadratic hints, enc(s))\n", now() - t0);

        std::vector<PtCyc> as;
        for (size_t b = 0; b < B; ++b) as.push_back(randPt(H[0]));

        // ---- plaintext result: eval ringRound (s * a)
        t0 = now();
        std::vector<PtCyc> expect(B);
        std::vector<std::vector<PtCyc>> stage_pt(B);                 // after mulPublic, every hop, x(1+x) -- for the stage checks
        bool all_even = true;
        const int64_t zs[8] = {0, -2, -6, -12, -20, -30, -42, -56};    // z (1 - z), z = 1 .. 8   (Language/RescaleTree.hs:69)
        for (size_t b = 0; b < B; ++b) {
            PtCyc x = ops.mul(s, as[b]);
            stage_pt[b].push_back(x);
            for (int k = 0; k < 5; ++k) { x = evalLin(ops, lin[k], x); stage_pt[b].push_back(x); }
            PtCyc y = ops.mul(x, ops.addScalar(x, 1));
            stage_pt[b].push_back(y);
            std::vector<PtCyc> t;
            for (int j = 0; j < 8; ++j) { PtCyc h; all_even &= ops.div2(ops.addScalar(y, zs[j]), h); t.push_back(h); }
            while (t.size() > 1) {
                std::vector<PtCyc> nx;
                for (size_t i = 0; i + 1 < t.size(); i += 2) { PtCyc h; all_even &= ops.div2(ops.mul(t[i], t[i + 1]), h); nx.push_back(h); }
                t = nx;
            }
            expect[b] = t[0];
        }
        printf("Computing plaintext result... %.2f s (%zu inputs; every div2 operand even: %s)\n", now() - t0, B, all_even ? "yes" : "NO");

        // ---- encrypted result: f a
        auto rate = [&](const char* what, const DevBatch& x, int key, const PtCyc* want) {
            // ERW: max |c(s)| / q of ciphertext 0 (LSD form), and the stage's decryption against the plaintext computation
            check(alch_sync(x.ring->handle()), "alch_sync");
            CT ct = x.download(0);
            PtCyc got;
            double er = 0;
            bool ok = decrypt(rc, ops, sk[key], ct, got, &er);
            bool match = ok && want && got.v == ops.to(*want, Basis::Pow).v && got.p == want->p;
            if (!quiet || (want && !match)) printf("  %-26s q has %d limbs, p = %2lld, k = %2d   error rate %.3e   decrypts to the plaintext stage: %s\n", what, x.ring->L(),
                   (long long)x.p, x.k, er, want ? (match ? "yes" : "NO") : "-");
            return !want || match;
        };
        t0 = now();
        bool stages_ok = true;
        DevBatch cur(r0, B);
        cur.enc = enc_s.enc; cur.k = 0; cur.l = 1; cur.p = P; cur.m = H[0]; cur.basis = Basis::CRT;
        {   // mulPublic a: every component of enc(s) times embed(reduce(lift a_b))
            alch_buf* pubs = nullptr;
            check(alch_buf_alloc(r0.handle(), 2 * B, &pubs), "alch_buf_alloc");
            for (size_t b = 0; b < B; ++b) {
                cur.upload(b, enc_s);
                const Cyc pub = liftEmbed(ops, r0, as[b]).toCRT();
                for (int c = 0; c < 2; ++c) check(alch_buf_upload(pubs, 2 * b + c, 1, pub.data().data()), "alch_buf_upload");
            }
            check(alch_buf_mul(cur.buf, cur.buf, pubs, 2 * B), "alch_buf_mul");
            alch_buf_free(pubs);
        }
        stages_ok &= rate("mulPublic a (H0')", cur, 0, &stage_pt[0][0]);
        CT first_in;
        if (per_element || per_resident || per_ziphost) first_in = cur.download(0);
        for (int k = 0; k < 5; ++k) {
            DevBatch nxt = tunnelBatch(dtun[k], cur, rc.get(HP[k + 1], moduli(tuns[k].lout)), H[k + 1]);
            if ((per_element || per_resident || per_ziphost) && k == 0) {
                // the same hop through the per-Tensor-call path (what `eval` over `instance Tensor GT` issues: one C-ABI call per
                // Tensor method): modSwitch up, tunnel, modSwitch down on ciphertext 0 -- must equal the batched result bit for bit
                check(alch_sync(nxt.ring->handle()), "alch_sync");
                const CT dev = nxt.download(0);
                const std::vector<uint64_t> qh = moduli(tuns[0].lh);
                struct Run { const char* name; Mode m; bool on; int reps; };
                const Run runs[3] = {{"host buffers (GTHost + host-buffer entry points)", Mode::HostBuffers, per_element, 2},
                                     {"device-resident elements (GTDev)", Mode::Resident, per_resident, 40},
                                     {"device-resident, pointwise ops on the host (unchanged Lol's zipWithT)", Mode::ResidentZipHost, per_ziphost, 4}};
                const Mode saved = mode();
                for (const Run& run : runs) {
                    if (!run.on) continue;
                    mode() = run.m;
                    // operands as a host would hold them in this representation; hint elements likewise (converted once, untimed)
                    CT in0 = first_in;
                    if (run.m != Mode::HostBuffers) { for (Cyc& x : in0.c) (void)x.dev(); for (auto& h : thints[0].ks) for (auto& pr : h.h) { (void)pr.first.dev(); (void)pr.second.dev(); } for (auto& y : thints[0].lin) (void)y.dev(); }
                    CT pe;
                    double best = 1e30, total = 0;
                    for (int rep = 0; rep < run.reps; ++rep) {
                        const double tp = now();
                        pe = modSwitch(tunnel(rc, thints[0], modSwitch(in0, rc.get(HP[0], qh)), H[1]), rc.get(HP[1], moduli(tuns[0].lout)));
                        check(alch_sync(pe.c[0].ring().handle()), "alch_sync");        // the hop has run, not just been queued
                        const double secs = now() - tp;
                        if (rep) { best = std::min(best, secs); total += secs; }
                    }
                    const double mean = total / (run.reps - 1);
                    const bool same = pe.c[0].toCRT().data() == dev.c[0].data() && pe.c[1].toCRT().data() == dev.c[1].data() && pe.l == dev.l;
                    printf("  per-Tensor-call path of switch1, %s: %.3f ms per hop (%.1f tunnels/s; best %.3f ms); equal to the batched result: %s\n",
                           run.name, mean * 1e3, 1.0 / mean, best * 1e3, same ? "yes" : "NO");
                    stages_ok &= same;
                }
                mode() = saved;
            }
            cur = std::move(nxt);
            char name[64];
            snprintf(name, sizeof name, "switch%d (H%d' -> H%d')", k + 1, k, k + 1);
            stages_ok &= rate(name, cur, k + 1, &stage_pt[0][k + 1]);
        }
        const Ring& r5in = rc.get(HP[5], moduli(muls[0].lin));
        if (cur.ring != &r5in) throw std::runtime_error("limb mismatch between switch5 and the first product");
        auto copyBatch = [&](const DevBatch& src) {
            DevBatch d(*src.ring, src.B);
            std::vector<uint64_t> one(src.ring->L(), 1);
            check(alch_buf_scale(d.buf, src.buf, 2 * src.B, one.data()), "alch_buf_scale");
            d.meta(src);
            return d;
        };
        PtCyc onePt{H[5], P, Basis::Pow, std::vector<int64_t>(totient(H[5]), 0)};
        onePt.v[0] = 1;
        // y = x * (1 + x)
        DevBatch x1 = copyBatch(cur);
        addPublicBatch(ops, x1, onePt);
        DevBatch y = mulFullBatch(dquad[0], cur, x1, rc.get(HP[5], moduli(muls[0].lout)));
        stages_ok &= rate("x * (1 + x)", y, 5, &stage_pt[0][6]);
        // leaves: div2 (y + z_j)
        std::vector<DevBatch> t;
        for (int j = 0; j < 8; ++j) {
            DevBatch leaf = copyBatch(y);
            PtCyc zj = onePt;
            zj.v[0] = ((zs[j] % P) + P) % P;
            addPublicBatch(ops, leaf, zj);
            modSwitchPTBatch(leaf, leaf.p / 2);
            t.push_back(std::move(leaf));
        }
        stages_ok &= rate("leaf div2 (y + z_1)", t[0], 5, nullptr);
        for (int level = 1; level <= 3; ++level) {
            std::vector<DevBatch> nx;
            for (size_t i = 0; i + 1 < t.size(); i += 2) {
                DevBatch pr = mulFullBatch(dquad[level], t[i], t[i + 1], rc.get(HP[5], moduli(muls[level].lout)));
                modSwitchPTBatch(pr, pr.p / 2);
                nx.push_back(std::move(pr));
            }
            t = std::move(nx);
            char name[64];
            snprintf(name, sizeof name, "tree level %d (mul_, div2)", level);
            stages_ok &= rate(name, t[0], 5, level == 3 ? &expect[0] : nullptr);
        }
        DevBatch& res = t[0];
        check(alch_sync(res.ring->handle()), "alch_sync");
        printf("Computing encrypted result... %.2f s (batch of %zu, including the per-stage decryptions above)\n", now() - t0, B);

        // ---- decrypt and compare (examples/HomomRLWR.hs:70-71)
        size_t good = 0;
        double worst = 0, mean = 0;
        std::vector<double> rates;
        for (size_t b = 0; b < B; ++b) {
            PtCyc got;
            double er = 0;
            if (decrypt(rc, ops, sk[5], res.download(b), got, &er) && got.p == expect[b].p && got.v == ops.to(expect[b], Basis::Pow).v) ++good;
            worst = std::max(worst, er);
            mean += er / (double)B;
            rates.push_back(er);
        }
        std::sort(rates.begin(), rates.end());
        const double p999 = rates[std::min(rates.size() - 1, (size_t)std::ceil(0.999 * (double)rates.size()) - 1)];
        const double med = rates[rates.size() / 2];
        printf("decrypted results equal to the plaintext results: %zu of %zu   (final error rate: mean %.3f, worst %.3f; decryption needs < 0.5)\n",
               good, B, mean, worst);
        printf("STATS batch %zu var_scale %.3f equal %zu mean %.4f median %.4f p99.9 %.4f max %.4f over_half %zu\n", B, var_scale, good, mean, med, p999, worst,
               (size_t)(rates.end() - std::lower_bound(rates.begin(), rates.end(), 0.5)));
        if (!dump.empty()) {
            std::vector<int64_t> meta = {(int64_t)B, (int64_t)res.ring->n(), (int64_t)res.ring->L(), res.enc == Encoding::MSD ? 1 : 0, res.k, res.l, res.p,
                                         (int64_t)res.ring->qs()[0]};
            dump_i64(dump + "/meta.i64", meta);
            dump_i64(dump + "/sk5_pow.i64", sk[5].s);
            std::vector<int64_t> cts, pts;
            for (size_t b = 0; b < B; ++b) {
                CT ct = res.download(b);
                for (int c = 0; c < 2; ++c) cts.insert(cts.end(), ct.c[c].data().begin(), ct.c[c].data().end());
                const PtCyc e = ops.to(expect[b], Basis::Pow);
                pts.insert(pts.end(), e.v.begin(), e.v.end());
            }
            dump_i64(dump + "/cts_crt.i64", cts);
            dump_i64(dump + "/expect_pow.i64", pts);
            // the first hop's linear function and plaintexts, for a by-definition check of decToCRT / evalLin by the model
            std::vector<int64_t> l0;
            for (const PtCyc& yv : lin[0].ys) l0.insert(l0.end(), yv.v.begin(), yv.v.end());
            dump_i64(dump + "/lin0_pow.i64", l0);
            dump_i64(dump + "/pt_h0.i64", stage_pt[0][0].v);
            dump_i64(dump + "/pt_h1.i64", ops.to(stage_pt[0][1], Basis::Pow).v);
        }
        const bool pass = good == B && stages_ok && all_even;
        printf("%s\n", pass ? "PASS" : "FAIL");
        return pass ? 0 : 1;
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
}
