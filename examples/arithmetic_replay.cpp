// Replays examples/Arithmetic.hs (reference) on the MI355X backend: addMul = \x y -> (x + y) * y on
// PT = R_7 of index 4, ciphertext index 512 (n' = 256) -- or 32 with `arithmetic_replay 32`, the BASELINE
// wording -- moduli 268440577, 8392193, 1073750017, TrivGad, Gaussian parameter 3.0.
//
// What `eval (pt2ct addMul)` executes, in E's bindings (Eval.hs:58-67,129-134) and PT2CT's op order
// (PT2CT.hs:172-177), with the limb counts the type-level rules select (2 limbs in, 2-limb hint,
// 1 limb out):    s = x + y ;  prod = s * y ;  modSwitch (same modulus: identity) ;
//                 keySwitchQuadCirc hint ;  modSwitch (drop the outer limb).
// Runs the per-op path and the fused batch path, checks they agree bit for bit, decrypts and compares with
// the plaintext evaluation, and prints PASS / FAIL like the reference (examples/Arithmetic.hs:73-75).
#include <cstdio>
#include <cstdlib>

#include "../alchemy_amd/host/symmshe.hpp"

using namespace alchemy;

static std::vector<uint64_t> ptMul(const std::vector<uint64_t>& a, const std::vector<uint64_t>& b, uint64_t p) {
    const size_t n = a.size();
    std::vector<int64_t> acc(n, 0);
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) {
            int64_t v = (int64_t)(a[i] * b[j] % p);
            if (i + j < n) acc[i + j] += v; else acc[i + j - n] -= v;
        }
    std::vector<uint64_t> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = (uint64_t)(((acc[i] % (int64_t)p) + (int64_t)p) % (int64_t)p);
    return out;
}

int main(int argc, char** argv) {
    const uint32_t m = argc > 1 ? (uint32_t)atoi(argv[1]) : 512;
    const uint64_t p = 7;
    const size_t npt = 2;                                   // F4: phi = 2
    try {
        // PNoise2Zq picks (q2, q1) for the multiplication input, q1 alone for the output; outer limb first.
        Ring ring2(m, {8392193, 268440577});
        Ring ring1(m, {268440577});
        std::mt19937_64 rng(2026);
        SK sk = genSK(ring2, 3.0, rng);
        KSQuadCircHint hint = ksQuadCircHint(ring2, sk, rng);

        std::vector<uint64_t> pt1(npt), pt2(npt);
        for (auto& v : pt1) v = rng() % p;
        for (auto& v : pt2) v = rng() % p;
        std::vector<uint64_t> sum(npt);
        for (size_t i = 0; i < npt; ++i) sum[i] = (pt1[i] + pt2[i]) % p;
        std::vector<uint64_t> ptresult = ptMul(sum, pt2, p);
        printf("PT evaluation result: [%llu, %llu]\n", (unsigned long long)ptresult[0], (unsigned long long)ptresult[1]);

        CT arg1 = encrypt(ring2, sk, pt1, p, rng), arg2 = encrypt(ring2, sk, pt2, p, rng);

        // per-op path: exactly E's call sequence
        CT s = arg1 + arg2;
        CT prod = s * arg2;
        CT ks = keySwitchQuadCirc(hint, prod);
        CT result = modSwitchDrop0(ks, ring1);

        // fused device path for  keySwitchQuad_ hint $: (s *: y)
        std::vector<CT> fused = mulRelinBatch(ring2, hint, {s}, {arg2});
        bool same = fused[0].l == ks.l && fused[0].k == ks.k;
        for (int c = 0; c < 2; ++c) same = same && fused[0].c[c].adviseCRT().data() == ks.c[c].adviseCRT().data();
        printf("fused path == per-op path: %s\n", same ? "yes" : "NO");
        CT result2 = modSwitchDrop0(fused[0], ring1);

        // The same product with the hint modulus KSPNoise asks for (PT2CT.hs:139: one more limb, here q3 in front):
        // modSwitch up, key switch on three limbs, modSwitch down to q1 -- per op and as one alch_ct_mul_full call.
        Ring ring3(m, {1073750017, 8392193, 268440577});
        SK sk3{sk.s, sk.r};
        KSQuadCircHint hint3 = ksQuadCircHint(ring3, sk3, rng);
        CT up = modSwitchAdd0(prod, ring3);
        CT ks3 = keySwitchQuadCirc(hint3, up);
        CT result3 = modSwitchDrop0(modSwitchDrop0(ks3, ring2), ring1);
        std::vector<CT> full = mulFullBatch(ring2, ring3, ring1, hint3, {s}, {arg2});
        bool same3 = full[0].l == result3.l && full[0].k == result3.k;
        for (int c = 0; c < 2; ++c) same3 = same3 && full[0].c[c].advisePow().data() == result3.c[c].advisePow().data();
        printf("fused full mul_ (2 -> 3 -> 1 limbs) == per-op path: %s\n", same3 ? "yes" : "NO");

        SK sk1{sk.s, sk.r};
        std::vector<uint64_t> dec = decrypt(sk1, result, npt), dec2 = decrypt(sk1, result2, npt);
        printf("Decrypted evaluation result: [%llu, %llu]\n", (unsigned long long)dec[0], (unsigned long long)dec[1]);
        std::vector<uint64_t> dec3 = decrypt(sk1, full[0], npt);
        const bool ok = same && same3 && dec == ptresult && dec2 == ptresult && dec3 == ptresult;
        printf("%s\n", ok ? "PASS" : "FAIL");
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
}
