// Microbenchmark: integer VALU issue rates on gfx950 that decide the NTT butterfly
// formulation (Shoup vs Montgomery, mad_u64_u32 vs mul_lo/mul_hi).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;   // independent chains per thread

enum Op { MUL_LO, MUL_HI, MAD64, ADD, MINU, MUL24, SUBMIN, BF_SHOUP, BF_MONT, BF_MONT_LAZY30, MONTMUL, SHOUPMUL };

template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, uint32_t seed, uint32_t q, uint32_t qinv) {
    uint32_t x[UNROLL], y[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { x[i] = seed * (threadIdx.x + 1 + i) + i; y[i] = x[i] ^ 0x9e3779b9u; }
    uint32_t w = seed | 1u, wp = seed * 77u + 5u;
    const uint32_t q2 = 2u * q;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if constexpr (OP == MUL_LO) {
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(w));
            } else if constexpr (OP == MUL_HI) {
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(w));
            } else if constexpr (OP == MAD64) {
                unsigned long long acc = ((unsigned long long)y[i] << 32) | x[i];
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x[i]), "v"(w) : "vcc");
                x[i] = (uint32_t)acc; y[i] = (uint32_t)(acc >> 32);
            } else if constexpr (OP == ADD) {
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(w));
            } else if constexpr (OP == MINU) {
                asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
            } else if constexpr (OP == MUL24) {
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(w));
            } else if constexpr (OP == SUBMIN) {
                uint32_t t;
                asm volatile("v_sub_u32 %0, %1, %2\n\tv_min_u32 %1, %1, %0" : "=&v"(t), "+v"(x[i]) : "v"(q));
            } else if constexpr (OP == SHOUPMUL) {
                // r = y*w - mulhi(y,wp)*q  in [0,2q)
                uint32_t hi = __umulhi(y[i], wp);
                y[i] = y[i] * w - hi * q;
            } else if constexpr (OP == MONTMUL) {
                unsigned long long p = (unsigned long long)y[i] * w;
                uint32_t m = (uint32_t)p * qinv;
                y[i] = (uint32_t)((p + (unsigned long long)m * q) >> 32);
            } else if constexpr (OP == BF_SHOUP) {
                // strict [0,q) CT butterfly, Shoup multiply
                uint32_t hi = __umulhi(y[i], wp);
                uint32_t t = y[i] * w - hi * q;
                t = min(t, t - q);
                uint32_t s = x[i] + t; s = min(s, s - q);
                uint32_t d = x[i] - t; d = min(d, d + q);
                x[i] = s; y[i] = d;
            } else if constexpr (OP == BF_MONT) {
                unsigned long long p = (unsigned long long)y[i] * w;
                uint32_t m = (uint32_t)p * qinv;
                uint32_t t = (uint32_t)((p + (unsigned long long)m * q) >> 32);
                t = min(t, t - q);
                uint32_t s = x[i] + t; s = min(s, s - q);
                uint32_t d = x[i] - t; d = min(d, d + q);
                x[i] = s; y[i] = d;
            } else if constexpr (OP == BF_MONT_LAZY30) {
                // Harvey lazy butterfly, values in [0,4q), q < 2^30
                unsigned long long p = (unsigned long long)y[i] * w;
                uint32_t m = (uint32_t)p * qinv;
                uint32_t t = (uint32_t)((p + (unsigned long long)m * q) >> 32);   // [0,2q)
                uint32_t xx = min(x[i], x[i] - q2);                                // [0,2q)
                x[i] = xx + t; y[i] = xx - t + q2;
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) r ^= x[i] + y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
static int run(const char* name, int insts_per_iter, int waves_per_simd) {
    int dev_cus = 256;
    int blocks = dev_cus * waves_per_simd;        // 256 threads = 4 waves = 1 per SIMD
    uint32_t* out;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const uint32_t q = 2147352577u;
    uint32_t qinv = 1; for (int i = 0; i < 5; ++i) qinv *= 2u - q * qinv; qinv = 0u - qinv;
    k_rate<OP><<<blocks, 256>>>(out, 12345u, q, qinv);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a));
        k_rate<OP><<<blocks, 256>>>(out, 12345u + rep, q, qinv);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    double units = (double)blocks * 256 * ITERS * UNROLL;   // lane-units (one "OP" each)
    double per_s = units / (best * 1e-3);
    // cycles per wave-unit per SIMD at 2.4 GHz: waves per SIMD * time * clk / (ITERS*UNROLL)
    double cyc = (double)best * 1e-3 * 2.4e9 / ((double)ITERS * UNROLL) / waves_per_simd;
    printf("%-16s waves/SIMD=%d  %8.3f ms  %8.2f Gunit/s  ~%6.2f cyc/unit/wave@2.4GHz  (%d inst/unit)\n",
           name, waves_per_simd, best, per_s * 1e-9, cyc, insts_per_iter);
    CK(hipFree(out));
    return 0;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int w : {1, 2, 4, 8}) {
        run<ADD>("v_add_u32", 1, w);
        run<MINU>("v_min_u32", 1, w);
        run<MUL24>("v_mul_u32_u24", 1, w);
        run<MUL_LO>("v_mul_lo_u32", 1, w);
        run<MUL_HI>("v_mul_hi_u32", 1, w);
        run<MAD64>("v_mad_u64_u32", 1, w);
        run<SUBMIN>("sub+min", 2, w);
        run<SHOUPMUL>("shoup mulmod", 4, w);
        run<MONTMUL>("mont mulmod", 3, w);
        run<BF_SHOUP>("bfly shoup", 12, w);
        run<BF_MONT>("bfly mont", 11, w);
        run<BF_MONT_LAZY30>("bfly mont lazy", 8, w);
    }
    return 0;
}
