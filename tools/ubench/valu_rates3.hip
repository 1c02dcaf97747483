// Microbenchmark 3: does an SGPR operand slow a VALU op on gfx950?  Butterfly with q/qni/w in VGPRs vs SGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;
constexpr int UNROLL = 16;
enum Op { BF_SGPR, BF_VGPR, BF_VGPR_WS, MUL_VS, MUL_VV, MAD_VS, MAD_VV, SUB_VS, SUB_VV, MIN_VV2, BF_INV_V, PW_ACC_V };

__device__ __forceinline__ uint32_t mm(uint32_t a, uint32_t b, uint32_t q, uint32_t qni) {
    unsigned long long p = (unsigned long long)a * b; uint32_t m = (uint32_t)p * qni;
    return (uint32_t)((p + (unsigned long long)m * q) >> 32);
}
__device__ __forceinline__ uint32_t cs(uint32_t x, uint32_t q) { uint32_t y = x - q; return y < x ? y : x; }

template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, uint32_t seed, uint32_t q_s, uint32_t qni_s) {
    uint32_t x[UNROLL], y[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { x[i] = seed * (threadIdx.x + 1 + i) + i; y[i] = x[i] ^ 0x9e3779b9u; }
    uint32_t w_s = seed | 1u;
    uint32_t q = q_s, qni = qni_s, w = w_s;
    if (OP == BF_VGPR || OP == BF_VGPR_WS || OP == MUL_VV || OP == MAD_VV || OP == SUB_VV || OP == MIN_VV2 || OP == BF_INV_V || OP == PW_ACC_V) {
        asm volatile("v_mov_b32 %0, %1" : "=v"(q) : "s"(q_s));
        asm volatile("v_mov_b32 %0, %1" : "=v"(qni) : "s"(qni_s));
        if (OP != BF_VGPR_WS) asm volatile("v_mov_b32 %0, %1" : "=v"(w) : "s"(w_s));
    }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if constexpr (OP == BF_SGPR || OP == BF_VGPR || OP == BF_VGPR_WS) {
                uint32_t xx = cs(x[i], q);
                uint32_t t = cs(mm(y[i], w, q, qni), q);
                x[i] = xx + t; y[i] = xx + (q - t);
            } else if constexpr (OP == BF_INV_V) {
                uint32_t a = cs(x[i], q), b = cs(y[i], q);
                x[i] = a + b; y[i] = mm(a - b + q, w, q, qni);
            } else if constexpr (OP == PW_ACC_V) {
                x[i] = cs(x[i] + cs(mm(y[i], w, q, qni), q), q);
            } else if constexpr (OP == MUL_VS || OP == MUL_VV) { x[i] = x[i] * w;
            } else if constexpr (OP == MAD_VS || OP == MAD_VV) {
                unsigned long long p = (unsigned long long)x[i] * w + (((unsigned long long)y[i] << 32) | x[i]); x[i] = (uint32_t)p; y[i] = (uint32_t)(p >> 32);
            } else if constexpr (OP == SUB_VS || OP == SUB_VV) { x[i] = x[i] - w;
            } else if constexpr (OP == MIN_VV2) { x[i] = min(x[i], y[i]); y[i] ^= x[i]; }
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
    const uint32_t q = 2147352577u; uint32_t qinv = 1; for (int i = 0; i < 5; ++i) qinv *= 2u - q * qinv; qinv = 0u - qinv;
    k_rate<OP><<<blocks, 256>>>(out, 12345u, q, qinv); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(a)); k_rate<OP><<<blocks, 256>>>(out, 12345u + rep, q, qinv); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    double units = (double)blocks * 256 * ITERS * UNROLL;
    printf("%-22s w/SIMD=%d %8.3f ms %9.1f Gunit/s  %6.2f cyc/unit/wave-slot@2.4GHz\n", name, wps, best, units / (best * 1e-3) * 1e-9,
           (double)best * 1e-3 * 2.4e9 / ((double)ITERS * UNROLL) / wps);
    CK(hipFree(out)); return 0;
}
int main() {
    for (int w : {4, 8}) {
        run<SUB_VS>("sub v,s", w); run<SUB_VV>("sub v,v", w);
        run<MUL_VS>("mul_lo v,s", w); run<MUL_VV>("mul_lo v,v", w);
        run<MAD_VS>("mad64 v,s", w); run<MAD_VV>("mad64 v,v", w);
        run<MIN_VV2>("min v,v + xor", w);
        run<BF_SGPR>("bfly fwd q,w sgpr", w); run<BF_VGPR>("bfly fwd q,w vgpr", w); run<BF_VGPR_WS>("bfly fwd q vgpr w sgpr", w);
        run<BF_INV_V>("bfly inv vgpr", w); run<PW_ACC_V>("acc+=x*h vgpr", w);
    }
    return 0;
}
