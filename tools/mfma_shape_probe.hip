// Probe (diagnostic): issue cost of the bf16 MFMA shapes and of a polynomial exp2 against v_exp_f32 on gfx950, one wave per SIMD. Backs
// DESIGN.md section 7 item 1 (why the head-dim-72 attention keeps its 32x32x16 formulation and its v_exp_f32).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_shape_probe.hip -o /tmp/mfma_shape_probe && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* cyc, float seed) {
    f32x4 a8 = {seed, seed, seed, seed}, b8 = a8;       // 8 bf16 per lane (k = 16 / 32 forms)
    f32x2 a4 = {seed, seed}, b4 = a4;                   // 4 bf16 per lane (the legacy _1k forms)
    f32x16 c0, c1;
    f32x4 d0 = {0, 0, 0, 0}, d1 = d0;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1) * 0.01f - 1.0f;
    const float k0 = 0.6565f, k1 = 0.3435f, k2 = 0.05f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
        if constexpr (MODE == 0) asm volatile(REP16("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t") : "+v"(c0), "+v"(c1) : "v"(a8), "v"(b8));
        else if constexpr (MODE == 1) asm volatile(REP16("v_mfma_f32_32x32x8bf16_1k %0, %2, %3, %0\n\tv_mfma_f32_32x32x8bf16_1k %1, %2, %3, %1\n\t") : "+v"(c0), "+v"(c1) : "v"(a4), "v"(b4));
        else if constexpr (MODE == 2) asm volatile(REP16("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n\t") : "+v"(d0), "+v"(d1) : "v"(a8), "v"(b8));
        else if constexpr (MODE == 3) asm volatile(REP16("v_mfma_f32_16x16x16bf16_1k %0, %2, %3, %0\n\tv_mfma_f32_16x16x16bf16_1k %1, %2, %3, %1\n\t") : "+v"(d0), "+v"(d1) : "v"(a4), "v"(b4));
        else if constexpr (MODE == 4)   // 32 v_exp_f32 on 8 independent registers
            asm volatile(REP16("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\t") : "+v"(v[0]), "+v"(v[1]));
        else if constexpr (MODE == 5)   // 32 polynomial exp2: fract, floor, three FMAs (Horner), ldexp = 6 instructions per value (two independent chains)
            asm volatile(REP16("v_fract_f32 %2, %0\n\tv_floor_f32 %3, %0\n\tv_fma_f32 %4, %2, %8, %7\n\tv_fma_f32 %4, %4, %2, %6\n\tv_fma_f32 %4, %4, %2, 1.0\n\tv_cvt_i32_f32 %3, %3\n\tv_ldexp_f32 %0, %4, %3\n\t"
                               "v_fract_f32 %2, %1\n\tv_floor_f32 %3, %1\n\tv_fma_f32 %5, %2, %8, %7\n\tv_fma_f32 %5, %5, %2, %6\n\tv_fma_f32 %5, %5, %2, 1.0\n\tv_cvt_i32_f32 %3, %3\n\tv_ldexp_f32 %1, %5, %3\n\t")
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]) : "v"(k0), "v"(k1), "v"(k2));
        else if constexpr (MODE == 6)   // 32 v_exp_f32 interleaved with 32 independent v_fma_f32: do the transcendental and the plain VALU overlap?
            asm volatile(REP16("v_exp_f32 %0, %0\n\tv_fma_f32 %2, %2, %4, %5\n\tv_exp_f32 %1, %1\n\tv_fma_f32 %3, %3, %4, %5\n\t") : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(k0), "v"(k1));
        else if constexpr (MODE == 7)   // 64 independent v_fma_f32 alone
            asm volatile(REP16("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t") : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(k0), "v"(k1));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i];
    acc += d0[0] + d1[0] + v[0] + v[1] + v[2] + v[3] + v[4] + v[5];
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 4); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(256), 0, 0, out, cyc, 0.37f);
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(256), 0, 0, out, cyc, 0.37f);
    unsigned long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on this part: convert with the measured ratio of mode 0 below if needed; report raw counts per instruction
    printf("%-62s %8.2f memtime ticks per instruction (%d per iteration x 64)\n", name, (double)h / (64.0 * per_iter), per_iter);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("v_mfma_f32_32x32x16_bf16 (k = 16), two accumulators", 32);
    run<1>("v_mfma_f32_32x32x8bf16_1k (k = 8, legacy), two accumulators", 32);
    run<2>("v_mfma_f32_16x16x32_bf16 (k = 32), two accumulators", 32);
    run<3>("v_mfma_f32_16x16x16bf16_1k (k = 16, legacy), two accumulators", 32);
    run<4>("v_exp_f32", 32);
    run<5>("polynomial exp2 (7 instructions per value)", 32);
    run<6>("v_exp_f32 + independent v_fma_f32, alternating (per pair)", 32);
    run<7>("v_fma_f32", 64);
    return 0;
}
