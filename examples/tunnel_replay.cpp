// Replays examples/Tunnel.hs (reference) on the MI355X backend at the reference's own parameters: the ring switches
// switch1 .. switchN of examples/Common.hs:78-95 (decToCRT @H_k) as PT2CT lowers them -- modSwitch_ .: tunnel_ hint .: modSwitch_
// (PT2CT.hs:224-229) -- with BaseBGad 2 hints (examples/Tunnel.hs:24), its five ~30-bit moduli (:34-39), Gaussian parameter 3.0 (:59),
// plaintext modulus 2^3 (PT = ... (Zq PP8), :41), ciphertext indices H0' .. H5'.  The file at this commit names an undefined
// `tunnel3` (= switch3: H0 -> H3); `tunnel_replay [hops] [batch]` runs switch<hops> (default 3, up to 5).
// The reference example only prints error rates; this replay also decrypts and compares with the plaintext evaluation (PASS / FAIL),
// after every hop, and prints the same error rates (max |c(s)| / q on the LSD form, ErrorRateWriter.hs:70-75).
#include <cstdio>
#include <cstdlib>

#include "../alchemy_amd/host/symmshe_gen.hpp"

using namespace alchemy::gen;

static const std::vector<uint64_t> QS = {537264001, 539884801, 555609601, 560851201, 566092801};     // Zqs order
static const uint32_t H[6] = {128, 448, 2912, 3640, 5460, 4095};
static const uint32_t HP[6] = {11648, 29120, 43680, 54600, 27300, 20475};
static const int64_t P = 8;

static std::vector<uint64_t> moduli(int L) { return std::vector<uint64_t>(QS.rend() - L, QS.rend()); }

int main(int argc, char** argv) {
    const int hops = argc > 1 ? atoi(argv[1]) : 3;
    const size_t B = argc > 2 ? (size_t)atoi(argv[2]) : 4;
    if (hops < 1 || hops > 5) { fprintf(stderr, "hops: 1 .. 5\n"); return 2; }
    try {
        std::mt19937_64 rng(59);
        RingCache rc;
        PtOps ops(rc, {QS[3], QS[4]});
        struct Limbs { int lin, lh, lout; } tuns[5];
        int pn = 0;
        for (int i = hops - 1; i >= 0; --i)
            check(alch_select_limbs(QS.data(), 5, ALCH_OP_TUNNEL, ALCH_GAD_BASE2, pn, &tuns[i].lin, &tuns[i].lh, &tuns[i].lout, &pn), "alch_select_limbs");
        printf("switch%d, BaseBGad 2; limbs (in/hint/out):", hops);
        for (int i = 0; i < hops; ++i) printf(" %d/%d/%d", tuns[i].lin, tuns[i].lh, tuns[i].lout);
        printf("\n");
        std::vector<SK> sk;
        for (int k = 0; k <= hops; ++k) sk.push_back(genSK(rc, HP[k], 3.0, rng));
        std::vector<Linear> lin;
        std::vector<DevTunnel> dtun;
        for (int k = 0; k < hops; ++k) {
            lin.push_back(decToCRT(ops, H[k], H[k + 1], 2, 3));
            const std::vector<uint64_t> qs = moduli(tuns[k].lh);
            TunnelHint th = tunnelHint(rc, ops, lin[k], HP[k], HP[k + 1], qs, sk[k + 1], sk[k], rng, ALCH_GAD_BASE2);
            printf("  hop %d: d_rel = %zu, %zu gadget digits per coefficient\n", k + 1, th.lin.size(), th.ks[0].h.size());
            dtun.emplace_back(rc.get(HP[k], qs), rc.get(HP[k + 1], qs), th);
        }
        // plaintexts: the example encrypts the constant 2; here ciphertext 0 does, the others are random
        std::vector<PtCyc> pts;
        for (size_t b = 0; b < B; ++b) {
            PtCyc x{H[0], P, Basis::Pow, std::vector<int64_t>(totient(H[0]), 0)};
            if (b == 0) x.v[0] = 2;
            else for (auto& v : x.v) v = (int64_t)(rng() % (uint64_t)P);
            pts.push_back(x);
        }
        const Ring& r0 = rc.get(HP[0], moduli(tuns[0].lin));
        DevBatch cur(r0, B);
        cur.enc = Encoding::LSD; cur.k = 0; cur.l = 1; cur.p = P; cur.m = H[0]; cur.basis = Basis::CRT;
        for (size_t b = 0; b < B; ++b) cur.upload(b, encrypt(rc, ops, r0, sk[0], pts[b], rng));
        bool ok = true;
        for (int k = 0; k < hops; ++k) {
            DevBatch nxt = tunnelBatch(dtun[k], cur, rc.get(HP[k + 1], moduli(tuns[k].lout)), H[k + 1]);
            cur = std::move(nxt);
            check(alch_sync(cur.ring->handle()), "alch_sync");
            size_t good = 0;
            double worst = 0;
            for (size_t b = 0; b < B; ++b) {
                pts[b] = evalLin(ops, lin[k], pts[b]);
                PtCyc got;
                double er = 0;
                if (decrypt(rc, ops, sk[k + 1], cur.download(b), got, &er) && got.v == ops.to(pts[b], Basis::Pow).v) ++good;
                worst = std::max(worst, er);
            }
            printf("  switch%d (H%d' -> H%d')  q has %d limb(s)   error rate (worst of %zu) %.3e   decrypts to the plaintext evaluation: %zu of %zu\n",
                   k + 1, k, k + 1, cur.ring->L(), B, worst, good, B);
            ok = ok && good == B;
        }
        printf("%s\n", ok ? "PASS" : "FAIL");
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
}
