// Microbenchmark 2: more VALU/cross-lane/LDS issue rates on gfx950, plus the effective shader clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;

enum Op { ADD_VV, ADD_VS, MIN_VV, MIN_VS, MAX_VS, MINI_VS, CMP_CND, SUBCO_CND, ADD3, LSHLADD, AND_VS, ASHR, MED3,
          DPP_ROR8, DPP_QP, DPP_MIRROR, PERMLANE32, SWIZZLE, BPERMUTE, MOV, BF_STRICT_MIN, BF_STRICT_CND, BF_X1Q, BF_LAZY30,
          LDS_R32, LDS_W32, LDS_R128, LDS_W128, NOPS };

template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, uint32_t seed, uint32_t q, uint32_t qinv, unsigned long long* clk) {
    __shared__ uint32_t lds[256 * 36];
    uint32_t x[UNROLL], y[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { x[i] = seed * (threadIdx.x + 1 + i) + i; y[i] = x[i] ^ 0x9e3779b9u; }
    for (int i = threadIdx.x; i < 256 * 36; i += 256) lds[i] = i * seed;
    __syncthreads();
    uint32_t w = seed | 1u;
    const uint32_t q2 = 2u * q;
    uint32_t laddr = threadIdx.x * 4;          // ds_*_b32 byte address: conflict-free
    uint32_t laddr128 = threadIdx.x * 144;     // padded row for b128
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if constexpr (OP == ADD_VV) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i])); }
            else if constexpr (OP == ADD_VS) { asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[i]) : "s"(q)); }
            else if constexpr (OP == MIN_VV) { asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[i]) : "v"(w)); }
            else if constexpr (OP == MIN_VS) { asm volatile("v_min_u32 %0, %1, %0" : "+v"(x[i]) : "s"(q)); }
            else if constexpr (OP == MAX_VS) { asm volatile("v_max_u32 %0, %1, %0" : "+v"(x[i]) : "s"(q)); }
            else if constexpr (OP == MINI_VS) { asm volatile("v_min_i32 %0, %1, %0" : "+v"(x[i]) : "s"(q)); }
            else if constexpr (OP == CMP_CND) {
                uint32_t t;
                asm volatile("v_subrev_u32 %0, %2, %1\n\tv_cmp_le_u32 vcc, %2, %1\n\tv_cndmask_b32 %1, %1, %0, vcc" : "=&v"(t), "+v"(x[i]) : "s"(q) : "vcc");
            }
            else if constexpr (OP == SUBCO_CND) {
                uint32_t t;
                asm volatile("v_subrev_co_u32 %0, vcc, %2, %1\n\tv_cndmask_b32 %1, %0, %1, vcc" : "=&v"(t), "+v"(x[i]) : "s"(q) : "vcc");
            }
            else if constexpr (OP == ADD3) { asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(w), "s"(q)); }
            else if constexpr (OP == LSHLADD) { asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x[i]) : "v"(w)); }
            else if constexpr (OP == AND_VS) { asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[i]) : "s"(q)); }
            else if constexpr (OP == ASHR) { asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(x[i])); }
            else if constexpr (OP == MED3) { asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(w), "s"(q)); }
            else if constexpr (OP == MOV) { asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(y[i])); }
            else if constexpr (OP == DPP_ROR8) { asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(x[i]) : "v"(y[i])); }
            else if constexpr (OP == DPP_QP) { asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(y[i])); }
            else if constexpr (OP == DPP_MIRROR) { asm volatile("v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(x[i]) : "v"(y[i])); }
            else if constexpr (OP == PERMLANE32) { asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[i]), "+v"(y[i])); }
            else if constexpr (OP == SWIZZLE) { x[i] = __builtin_amdgcn_ds_swizzle(x[i], 0x041F | (16 << 10)); }   // xor 16 within 32
            else if constexpr (OP == BPERMUTE) { x[i] = __builtin_amdgcn_ds_bpermute(laddr ^ 128, x[i]); }
            else if constexpr (OP == NOPS) { asm volatile("s_nop 0"); }
            else if constexpr (OP == LDS_R32) { uint32_t v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(laddr), "n"(i * 1024)); x[i] ^= v; }
            else if constexpr (OP == LDS_W32) { asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(laddr), "v"(x[i]), "n"(i * 1024) : "memory"); }
            else if constexpr (OP == LDS_R128) {
                if (i < 8) { u32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(laddr128), "n"((i & 7) * 16)); x[i] ^= v.x; y[i] ^= v.y; x[i + 8] ^= v.z; y[i + 8] ^= v.w; }
            }
            else if constexpr (OP == LDS_W128) {
                if (i < 8) { u32x4 v = {x[i], y[i], x[i + 8], y[i + 8]}; asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(laddr128), "v"(v), "n"((i & 7) * 16) : "memory"); }
            }
            else if constexpr (OP == BF_STRICT_MIN) {
                unsigned long long p = (unsigned long long)y[i] * w;
                uint32_t m = (uint32_t)p * qinv;
                uint32_t t = (uint32_t)((p + (unsigned long long)m * q) >> 32);
                t = min(t, t - q);
                uint32_t s = x[i] + t; s = min(s, s - q);
                uint32_t d = x[i] - t; d = min(d, d + q);
                x[i] = s; y[i] = d;
            }
            else if constexpr (OP == BF_STRICT_CND) {
                unsigned long long p = (unsigned long long)y[i] * w;
                uint32_t m = (uint32_t)p * qinv;
                uint32_t t = (uint32_t)((p + (unsigned long long)m * q) >> 32);
                t = t >= q ? t - q : t;
                uint32_t s = x[i] + t; s = s >= q ? s - q : s;
                uint32_t d = x[i] >= t ? x[i] - t : x[i] - t + q;
                x[i] = s; y[i] = d;
            }
            else if constexpr (OP == BF_X1Q) {
                // X reduced on entry to [0,q); Y arbitrary u32; outputs in [0,2q) (no output csub)
                unsigned long long p = (unsigned long long)y[i] * w;
                uint32_t m = (uint32_t)p * qinv;
                uint32_t t = (uint32_t)((p + (unsigned long long)m * q) >> 32);
                t = min(t, t - q);
                uint32_t xx = min(x[i], x[i] - q);
                x[i] = xx + t; y[i] = xx + (q - t);
            }
            else if constexpr (OP == BF_LAZY30) {
                unsigned long long p = (unsigned long long)y[i] * w;
                uint32_t m = (uint32_t)p * qinv;
                uint32_t t = (uint32_t)((p + (unsigned long long)m * q) >> 32);
                uint32_t xx = min(x[i], x[i] - q2);
                x[i] = xx + t; y[i] = xx - t + q2;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) r ^= x[i] + y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + lds[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int OP>
static int run(const char* name, int waves_per_simd, double unit_scale = 1.0) {
    int blocks = 256 * waves_per_simd;
    uint32_t* out; unsigned long long* clk;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CK(hipMalloc(&clk, 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const uint32_t q = 2147352577u;
    uint32_t qinv = 1; for (int i = 0; i < 5; ++i) qinv *= 2u - q * qinv; qinv = 0u - qinv;
    k_rate<OP><<<blocks, 256>>>(out, 12345u, q, qinv, clk);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        k_rate<OP><<<blocks, 256>>>(out, 12345u + rep, q, qinv, clk);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double ghz = (double)h[0] / ((double)h[1] / 100e6) * 1e-9;
    double units = (double)ITERS * UNROLL * unit_scale;
    double cyc_wave = (double)h[0] / units;                  // shader cycles per unit for one wave (incl. co-resident waves' share)
    double per_s = (double)blocks * 256 * units / (best * 1e-3);
    printf("%-14s w/SIMD=%d %8.3f ms %9.1f Gunit/s  in-kernel %6.2f cyc/unit/wave -> %5.2f cyc/unit/SIMD  clk %.3f GHz\n",
           name, waves_per_simd, best, per_s * 1e-9, cyc_wave, cyc_wave / waves_per_simd, ghz);
    CK(hipFree(out)); CK(hipFree(clk));
    return 0;
}

int main() {
    for (int w : {1, 4, 8}) {
        run<NOPS>("s_nop", w);
        run<MOV>("v_mov", w);
        run<ADD_VV>("add v,v", w);
        run<ADD_VS>("add v,s", w);
        run<MIN_VV>("min_u32 v,v", w);
        run<MIN_VS>("min_u32 v,s", w);
        run<MAX_VS>("max_u32 v,s", w);
        run<MINI_VS>("min_i32 v,s", w);
        run<AND_VS>("and v,s", w);
        run<ASHR>("ashr", w);
        run<ADD3>("add3", w);
        run<LSHLADD>("lshl_add", w);
        run<MED3>("med3_u32", w);
        run<CMP_CND>("sub+cmp+cnd", w);
        run<SUBCO_CND>("subco+cnd", w);
        run<DPP_ROR8>("mov dpp ror8", w);
        run<DPP_QP>("add dpp qperm", w);
        run<DPP_MIRROR>("mov dpp mirror", w);
        run<PERMLANE32>("permlane32swap", w);
        run<SWIZZLE>("ds_swizzle", w);
        run<BPERMUTE>("ds_bpermute", w);
        run<LDS_R32>("ds_read_b32", w);
        run<LDS_W32>("ds_write_b32", w);
        run<LDS_R128>("ds_read_b128", w, 0.5);
        run<LDS_W128>("ds_write_b128", w, 0.5);
        run<BF_STRICT_MIN>("bf strict min", w);
        run<BF_STRICT_CND>("bf strict cnd", w);
        run<BF_X1Q>("bf x<q lazy", w);
        run<BF_LAZY30>("bf lazy30", w);
    }
    return 0;
}
