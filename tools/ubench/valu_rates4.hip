// Microbenchmark 4: Plantard vs Montgomery forward butterfly, register-resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../alchemy_amd/csrc/modarith.hpp"
using namespace alch;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096, UNROLL = 16;
template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, uint32_t seed, uint32_t q, uint32_t qni, u64 br) {
    uint32_t x[UNROLL], y[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { x[i] = seed * (threadIdx.x + 1 + i) + i; y[i] = x[i] ^ 0x9e3779b9u; }
    uint32_t w = (seed | 1u) % q;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) bfly_fwd(x[i], y[i], w, q, qni);
            else if (OP == 1) bfly_fwd(x[i], y[i], br, q, qni);
            else if (OP == 2) { x[i] = plant_mul(x[i], br, q); }
            else if (OP == 3) { x[i] = csub(mont_mul_lazy(x[i], w, q, qni), q); }
            else if (OP == 4) { x[i] = __umulhi(x[i], w) + y[i]; }
            else if (OP == 5) { unsigned long long p = (unsigned long long)x[i] * w + y[i]; x[i] = (uint32_t)(p >> 32); }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) r ^= x[i] + y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> static int run(const char* name, int wps) {
    int blocks = 256 * wps; uint32_t* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const uint32_t q = 2147352577u; ModP<u32> m = make_modp<u32>(q); u64 br = h_plant_const(123456789ull, q);
    k_rate<OP><<<blocks, 256>>>(out, 12345u, q, m.qni, br); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(a)); k_rate<OP><<<blocks, 256>>>(out, 12345u + rep, q, m.qni, br); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    double units = (double)blocks * 256 * ITERS * UNROLL;
    printf("%-24s w/SIMD=%d %8.3f ms %9.1f Gunit/s %6.2f cyc/unit\n", name, wps, best, units / (best * 1e-3) * 1e-9, (double)best * 1e-3 * 2.4e9 / ((double)ITERS * UNROLL) / wps);
    CK(hipFree(out)); return 0;
}
int main() { for (int w : {4, 8}) { run<0>("bfly fwd montgomery", w); run<1>("bfly fwd plantard", w); run<2>("plant_mul", w); run<3>("mont_mul+csub", w); run<4>("mul_hi+add", w); run<5>("mad64 hi", w);} return 0; }
