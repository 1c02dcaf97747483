// AddressSanitizer / UndefinedBehaviorSanitizer harness for the HOST-ONLY code of the product library:
// gen_host.hpp (index plans, ext tables), crtset_host.hpp (CRT sets over GF(p^d)), ring_host.hpp (primality, generators, roots),
// alch_select_limbs / alch_modulus_units, the argument checks of alch_ring_create* -- everything of libalchemy_hip that runs
// without a device.  GPU sanitizers are not available on this pool, so this is the sanitizer coverage the product gets.
//
// Build (tests/test_host_sanitizers.py):  hipcc --offload-host-only -fsanitize=address,undefined  on alchemy_hip.hip, linked with
// this file.  The HIP runtime and the kernel dispatchers of the other translation units are replaced by "no device" definitions
// below: every compute entry point then ends in ALCH_E_NO_DEVICE, which is also asserted.  Nothing here is a CPU fallback -- no
// arithmetic of the hot path exists in this program.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/alchemy_hip.h"

// ---- "no device" HIP runtime -------------------------------------------------------------------------------------
extern "C" {
typedef int hipError_t;
static const hipError_t NO_DEVICE = 100;                           // hipErrorNoDevice
hipError_t hipGetDeviceCount(int* n) { if (n) *n = 0; return NO_DEVICE; }
hipError_t hipGetDevice(int*) { return NO_DEVICE; }
hipError_t hipSetDevice(int) { return NO_DEVICE; }
hipError_t hipGetDevicePropertiesR0600(void*, int) { return NO_DEVICE; }
const char* hipGetErrorString(hipError_t) { return "no device (sanitizer harness)"; }
hipError_t hipGetLastError(void) { return NO_DEVICE; }
hipError_t hipMalloc(void**, size_t) { return NO_DEVICE; }
hipError_t hipFree(void*) { return NO_DEVICE; }
hipError_t hipHostMalloc(void**, size_t, unsigned) { return NO_DEVICE; }
hipError_t hipHostFree(void*) { return NO_DEVICE; }
hipError_t hipMemcpy(void*, const void*, size_t, int) { return NO_DEVICE; }
hipError_t hipMemcpyAsync(void*, const void*, size_t, int, void*) { return NO_DEVICE; }
hipError_t hipMemsetAsync(void*, int, size_t, void*) { return NO_DEVICE; }
hipError_t hipStreamCreateWithFlags(void**, unsigned) { return NO_DEVICE; }
hipError_t hipExtStreamCreateWithCUMask(void**, unsigned, const unsigned*) { return NO_DEVICE; }
hipError_t hipStreamDestroy(void*) { return NO_DEVICE; }
hipError_t hipStreamSynchronize(void*) { return NO_DEVICE; }
hipError_t hipStreamWaitEvent(void*, void*, unsigned) { return NO_DEVICE; }
hipError_t hipEventCreate(void**) { return NO_DEVICE; }
hipError_t hipEventCreateWithFlags(void**, unsigned) { return NO_DEVICE; }
hipError_t hipEventDestroy(void*) { return NO_DEVICE; }
hipError_t hipEventRecord(void*, void*) { return NO_DEVICE; }
hipError_t hipEventSynchronize(void*) { return NO_DEVICE; }
hipError_t hipEventElapsedTime(float*, void*, void*) { return NO_DEVICE; }
hipError_t hipLaunchKernel(const void*, ...) { return NO_DEVICE; }
hipError_t __hipPushCallConfiguration(...) { return NO_DEVICE; }
hipError_t __hipPopCallConfiguration(...) { return NO_DEVICE; }
void** __hipRegisterFatBinary(const void*) { static void* h = nullptr; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
}
// the kernel dispatchers of the other translation units (never reached without a device)
#define NO_DISPATCH(mangled, id) extern "C" hipError_t nodisp_##id() asm(mangled); extern "C" hipError_t nodisp_##id() { return NO_DEVICE; }
NO_DISPATCH("_ZN4alch12gen_dispatchERKNS_7GenCallIjEE", 0)
NO_DISPATCH("_ZN4alch12gen_dispatchERKNS_7GenCallImEE", 1)
NO_DISPATCH("_ZN4alch13dispatch32_14EiRKNS_7NttCallIjEE", 2)
NO_DISPATCH("_ZN4alch13dispatch32_15EiRKNS_7NttCallIjEE", 3)
NO_DISPATCH("_ZN4alch13dispatch32_16EiRKNS_7NttCallIjEE", 4)
NO_DISPATCH("_ZN4alch13dispatch64_15EiRKNS_7NttCallImEE", 5)
NO_DISPATCH("_ZN4alch14dispatch32_midEiRKNS_7NttCallIjEE", 6)
NO_DISPATCH("_ZN4alch14dispatch64_bigEiRKNS_7NttCallImEE", 7)
NO_DISPATCH("_ZN4alch15gen_ks_dispatchERKNS_7DevRingIjEERKNS_6GenDevIjEERKNS_9GenKsArgsIjEEmP12ihipStream_t", 8)
NO_DISPATCH("_ZN4alch16dispatch32_smallEiRKNS_7NttCallIjEE", 9)
NO_DISPATCH("_ZN4alch16dispatch64_smallEiRKNS_7NttCallImEE", 10)
NO_DISPATCH("_ZN4alch22gen_tunnel_ks_dispatchERKNS_7DevRingIjEERKNS_6GenDevIjEERKNS_10GenTunArgsIjEEmiP12ihipStream_t", 11)
NO_DISPATCH("_ZN4alch24gen_rescale_lin_dispatchERKNS_7DevRingIjEERKNS_6GenDevIjEEPKjPjSA_RKNS_7DropTabIjEEimP12ihipStream_tb", 12)
NO_DISPATCH("_ZN4alch24gen_rescale_lin_dispatchERKNS_7DevRingImEERKNS_6GenDevImEEPKmPmSA_RKNS_7DropTabImEEimP12ihipStream_tb", 13)

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { ++failures; fprintf(stderr, "FAILED %s:%d: %s  (%s)\n", __FILE__, __LINE__, #cond, alch_last_error()); } } while (0)

static uint32_t totient(uint32_t m) {
    uint32_t r = m, t = m;
    for (uint32_t p = 2; (uint64_t)p * p <= t; ++p) if (t % p == 0) { r -= r / p; while (t % p == 0) t /= p; }
    return t > 1 ? r - r / t : r;
}

int main() {
    const uint32_t H[6] = {128, 448, 2912, 3640, 5460, 4095}, HP[6] = {11648, 29120, 43680, 54600, 27300, 20475};
    // ---- the index pairs of tests/test_tensor_ext.py: all three tables, their documented lengths, entries in range
    std::vector<std::pair<uint32_t, uint32_t>> pairs = {{4, 12}, {3, 9}, {5, 15}, {8, 40}, {12, 60}, {7, 21}, {1, 7}, {9, 45}, {4, 8}, {16, 48},
                                                        {32, 96}, {1, 1}, {15, 15}, {64, 448}, {1365, 4095}, {2275, 20475}, {128, 11648}};
    for (int k = 0; k < 6; ++k) pairs.push_back({H[k], HP[k]});
    for (int k = 0; k < 5; ++k) { uint32_t a = HP[k], b = HP[k + 1]; while (b) { uint32_t t = a % b; a = b; b = t; } pairs.push_back({a, HP[k]}); pairs.push_back({a, HP[k + 1]}); }
    for (auto& pr : pairs) {
        const uint32_t ns = totient(pr.first), nb = totient(pr.second), d = nb / ns;
        const size_t want[3] = {ns, (size_t)d * ns, nb};
        for (int which = 0; which < 3; ++which) {
            size_t len = 0;
            EXPECT(alch_ext_table(pr.first, pr.second, which, nullptr, &len) == ALCH_OK && len == want[which]);
            std::vector<int32_t> t(len);                                    // exactly as large as reported: ASan sees any overrun
            EXPECT(alch_ext_table(pr.first, pr.second, which, t.data(), &len) == ALCH_OK);
            const int32_t bound = which == 2 ? (int32_t)ns : (int32_t)nb;
            for (int32_t v : t) EXPECT(v >= 0 && v < bound);
            if (len > 1) { size_t small = len - 1; EXPECT(alch_ext_table(pr.first, pr.second, which, t.data(), &small) == ALCH_E_INVALID); }
        }
    }
    { size_t len = 0; EXPECT(alch_ext_table(12, 40, 0, nullptr, &len) == ALCH_E_INVALID); EXPECT(alch_ext_table(4, 12, 7, nullptr, &len) == ALCH_E_INVALID);
      EXPECT(alch_ext_table(4, 12, 0, nullptr, nullptr) == ALCH_E_INVALID); }
    // ---- crtSetDec: the small cases of the test-suite and the reference's hops on their odd parts
    struct CS { uint32_t m, mb, p; };
    std::vector<CS> sets = {{1, 7, 2}, {3, 15, 2}, {1, 13, 3}, {5, 35, 3}, {9, 63, 2}, {7, 91, 2}, {4, 12, 5}, {1, 8, 3}, {1, 1, 2}, {7, 7, 2}, {1, 15, 2}, {1, 21, 2}};
    for (int k = 0; k < 5; ++k) {
        uint32_t a = H[k], b = H[k + 1]; while (b) { uint32_t t = a % b; a = b; b = t; }
        uint32_t e = a, s = H[k + 1];
        while (e % 2 == 0) e /= 2;
        while (s % 2 == 0) s /= 2;
        sets.push_back({e, s, 2});
    }
    for (auto& c : sets) {
        size_t count = 0;
        EXPECT(alch_crt_set_dec(c.m, c.mb, c.p, nullptr, &count) == ALCH_OK && count >= 1);
        const uint32_t nb = totient(c.mb);
        std::vector<int64_t> v(count * nb);
        size_t cap = count;
        EXPECT(alch_crt_set_dec(c.m, c.mb, c.p, v.data(), &cap) == ALCH_OK && cap == count);
        for (int64_t x : v) EXPECT(x >= 0 && x < (int64_t)c.p);
        if (count > 1) { size_t small = count - 1; EXPECT(alch_crt_set_dec(c.m, c.mb, c.p, v.data(), &small) == ALCH_E_INVALID); }
    }
    { size_t count = 0; EXPECT(alch_crt_set_dec(3, 7, 2, nullptr, &count) == ALCH_E_INVALID); EXPECT(alch_crt_set_dec(1, 14, 2, nullptr, &count) < 0);
      EXPECT(alch_crt_set_dec(1, 7, 2, nullptr, nullptr) == ALCH_E_INVALID); }
    // ---- limb-count selection: every (p_noise, op, gadget) over the three examples' modulus lists and their prefixes
    const std::vector<std::vector<uint64_t>> lists = {{268440577, 8392193, 1073750017}, {1543651201, 689270401, 718099201, 720720001, 1556755201, 1567238401},
                                                      {537264001, 539884801, 555609601, 560851201, 566092801}};
    for (auto& zqs : lists)
        for (int n = 1; n <= (int)zqs.size(); ++n)
            for (int p = 0; p < 40; ++p)
                for (int op = 0; op < 2; ++op)
                    for (int gad = 0; gad < 2; ++gad) {
                        int li = -1, lh = -1, lo = -1, pin = -1;
                        const int rc = alch_select_limbs(zqs.data(), n, op, gad, p, &li, &lh, &lo, &pin);
                        EXPECT(rc == ALCH_OK || rc == ALCH_E_INVALID);
                        if (rc == ALCH_OK) EXPECT(li >= 1 && li <= n && lh >= 1 && lh <= n && lo >= 1 && lo <= li && pin > p);
                        EXPECT(alch_select_limbs(zqs.data(), n, op, gad, p, nullptr, nullptr, nullptr, nullptr) == rc);
                    }
    EXPECT(alch_select_limbs(nullptr, 3, 0, 0, 0, nullptr, nullptr, nullptr, nullptr) == ALCH_E_INVALID);
    EXPECT(alch_select_limbs(lists[0].data(), 3, 2, 0, 0, nullptr, nullptr, nullptr, nullptr) == ALCH_E_INVALID);
    EXPECT(alch_modulus_units(0) == 0 && alch_modulus_units(1543651201) == 5 && alch_modulus_units(689270401) == 4 && alch_modulus_units(~0ull) == 10);
    // ---- the root rule
    for (uint64_t q : {268440577ull, 8392193ull, 1073750017ull, 2147352577ull, 1152921504606748673ull, 12289ull, 65537ull}) {
        uint64_t psi = 0, g = 0;
        for (uint32_t m : {32u, 512u, 4096u}) {
            const int rc = alch_host_root(m, q, &psi, &g);
            EXPECT(rc == ((q - 1) % m ? ALCH_E_NO_CRT : ALCH_OK));
            if (rc == ALCH_OK) EXPECT(psi > 1 && psi < q && g > 1 && g < q);
        }
    }
    EXPECT(alch_host_root(512, 268440579, nullptr, nullptr) == ALCH_E_NOT_PRIME);
    EXPECT(alch_host_root(0, 7, nullptr, nullptr) == ALCH_E_INVALID);
    // ---- alch_ring_create*: the whole argument classification runs before the device probe, then "no device"
    struct RC { uint32_t m; std::vector<uint64_t> q; bool nocrt; int want; };
    std::vector<RC> rcs = {{512, {268440579}, false, ALCH_E_NO_CRT}, {11648, {32}, false, ALCH_E_NO_CRT}, {4, {7}, false, ALCH_E_NO_CRT},
                           {68, {32}, false, ALCH_E_NO_CRT}, {68, {268435577}, false, ALCH_E_UNSUPPORTED}, {512, {0}, false, ALCH_E_INVALID},
                           {512, {268440577, 268440577}, false, ALCH_E_INVALID}, {1u << 18, {2146959361}, false, ALCH_E_UNSUPPORTED},
                           {3 * 5 * 7 * 11 * 13 * 64, {960961}, false, ALCH_E_UNSUPPORTED}, {512, {268440577, 8392193}, false, ALCH_E_NO_DEVICE},
                           {20475, {1543651201, 689270401}, false, ALCH_E_NO_DEVICE}, {11648, {32}, true, ALCH_E_NO_DEVICE},
                           {54600, {0}, true, ALCH_E_NO_DEVICE}, {68, {32}, true, ALCH_E_UNSUPPORTED}, {512, {0, 7}, true, ALCH_E_INVALID},
                           {1, {7}, true, ALCH_E_NO_DEVICE}, {0, {7}, false, ALCH_E_INVALID}};
    for (int k = 0; k < 6; ++k) for (int L = 1; L <= 6; ++L) {
        std::vector<uint64_t> q(lists[1].rend() - L, lists[1].rend());
        rcs.push_back({HP[k], q, false, ALCH_E_NO_DEVICE});
        rcs.push_back({H[k], {1ull << ((L % 5) + 1)}, false, ALCH_E_NO_CRT});
    }
    for (auto& r : rcs) {
        alch_ring* h = reinterpret_cast<alch_ring*>(1);
        const int rc = (r.nocrt ? alch_ring_create_nocrt : alch_ring_create)(r.m, (int)r.q.size(), r.q.data(), &h);
        EXPECT(rc == r.want);
        EXPECT(h == nullptr);
    }
    { alch_ring* h = nullptr; EXPECT(alch_ring_create(512, 0, lists[0].data(), &h) == ALCH_E_INVALID); EXPECT(alch_ring_create(512, 9, lists[0].data(), &h) == ALCH_E_INVALID);
      EXPECT(alch_ring_create(512, 1, nullptr, &h) == ALCH_E_INVALID); EXPECT(alch_ring_create(512, 1, lists[0].data(), nullptr) == ALCH_E_INVALID); }
    // ---- null handles never dereference
    EXPECT(alch_buf_free(nullptr) == ALCH_OK && alch_ring_destroy(nullptr) == ALCH_OK && alch_hint_free(nullptr) == ALCH_OK && alch_tunnel_free(nullptr) == ALCH_OK);
    EXPECT(alch_buf_alloc(nullptr, 1, nullptr) == ALCH_E_INVALID && alch_buf_view(nullptr, 0, 1, nullptr) == ALCH_E_INVALID);
    EXPECT(alch_buf_tensor_op(nullptr, 0, nullptr, 0, 1, 0) == ALCH_E_INVALID && alch_buf_copy(nullptr, 0, nullptr, 0, 1) == ALCH_E_INVALID);
    EXPECT(alch_ring_share_stream(nullptr, nullptr) == ALCH_E_INVALID && alch_sync(nullptr) == ALCH_E_INVALID);
    EXPECT(alch_tunnel_info(nullptr, nullptr, nullptr, nullptr) < 0 && alch_ct_mod_switch(nullptr, nullptr, 1, 0) < 0);
    printf("%s: %d failed expectation(s)\n", failures ? "FAIL" : "OK", failures);
    return failures ? 1 : 0;
}
