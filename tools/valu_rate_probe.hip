// Probe (diagnostic): issue cost of the instructions the fp8 attention softmax is made of, one wave per SIMD, and how much of them hides
// behind a v_mfma_scale_f32_32x32x64_f8f6f4. Prints cycles per instruction (s_memtime around 64 x 16 unrolled instructions).
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate_probe.hip -o tools/valu_rate_probe && ./tools/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* cyc, float seed) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + 0.001f * (threadIdx.x + i);
    i32x8 a = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838}, b = a;
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
    int sc = 127;
    unsigned int pk = 0;
    float scale = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
        if constexpr (MODE == 0) {   // 16 v_exp_f32 (independent)
            asm volatile(REP16("v_exp_f32 %0, %0\n\t") : "+v"(v[0]));
        } else if constexpr (MODE == 1) {   // 16 independent v_exp
            asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t"
                         "v_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_exp_f32 %12, %12\n\tv_exp_f32 %13, %13\n\tv_exp_f32 %14, %14\n\tv_exp_f32 %15, %15"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]),
                           "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
        } else if constexpr (MODE == 2) {   // 16 v_cvt_scalef32_pk_fp8_f32
            asm volatile(REP16("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3\n\t") : "+v"(pk) : "v"(v[0]), "v"(v[1]), "v"(scale));
        } else if constexpr (MODE == 3) {   // 16 v_max3_f32 (dependent chain on %0)
            asm volatile(REP16("v_max3_f32 %0, %0, %1, %2\n\t") : "+v"(v[0]) : "v"(v[1]), "v"(v[2]));
        } else if constexpr (MODE == 4) {   // 16 v_cvt_pk_fp8_f32 (unscaled)
            asm volatile(REP16("v_cvt_pk_fp8_f32 %0, %1, %2\n\t") : "+v"(pk) : "v"(v[0]), "v"(v[1]));
        } else if constexpr (MODE == 5) {   // 16 e4m3 MFMAs back to back, two accumulators
            asm volatile(REP16("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %2, %3, %0, %4, %4 op_sel_hi:[0,0,0]\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %1, %2, %3, %1, %4, %4 op_sel_hi:[0,0,0]\n\t")
                         : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(sc));
        } else if constexpr (MODE == 6) {   // MFMA + 8 v_exp, 16 times
            asm volatile(REP16("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %2, %3, %0, %4, %4 op_sel_hi:[0,0,0]\n\t"
                               "v_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_exp_f32 %12, %12\n\t")
                         : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(sc), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
        } else if constexpr (MODE == 7) {   // MFMA + 4 v_exp + 4 cvt + 4 max3
            asm volatile(REP16("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %2, %3, %0, %4, %4 op_sel_hi:[0,0,0]\n\t"
                               "v_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\t"
                               "v_cvt_scalef32_pk_fp8_f32 %13, %9, %10, %14\n\tv_cvt_scalef32_pk_fp8_f32 %13, %11, %12, %14\n\tv_cvt_scalef32_pk_fp8_f32 %13, %9, %10, %14\n\tv_cvt_scalef32_pk_fp8_f32 %13, %11, %12, %14\n\t"
                               "v_max3_f32 %9, %9, %10, %11\n\tv_max3_f32 %9, %9, %10, %11\n\tv_max3_f32 %9, %9, %10, %11\n\tv_max3_f32 %9, %9, %10, %11\n\t")
                         : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(sc), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(pk), "v"(scale));
        } else if constexpr (MODE == 8) {   // bf16 32x32x16 MFMA + 4 v_exp
            asm volatile(REP16("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t")
                         : "+v"(c0), "+v"(c1) : "v"(*(float __attribute__((ext_vector_type(4)))*)&a), "v"(*(float __attribute__((ext_vector_type(4)))*)&b), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
        } else if constexpr (MODE == 9) {   // e4m3 MFMA with an accumulator in AGPRs + 8 v_exp
            asm volatile(REP16("v_mfma_scale_f32_32x32x64_f8f6f4 a[0:15], %0, %1, a[0:15], %2, %2 op_sel_hi:[0,0,0]\n\t"
                               "v_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\t")
                         :: "v"(a), "v"(b), "v"(sc), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7])
                         : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
        } else if constexpr (MODE == 11) {  // e4m3 MFMA, fresh VGPR destination, C = 0 + 8 v_exp
            asm volatile(REP16("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %2, %3, 0, %4, %4 op_sel_hi:[0,0,0]\n\t"
                               "v_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_exp_f32 %12, %12\n\t")
                         : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(sc), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
        } else if constexpr (MODE == 12) {  // the phase-A pattern: F8(c0) 8 exp F8(c1) 8 exp B16(c0) 4 exp B16(c1) 4 exp
            asm volatile(REP16("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %2, %3, 0, %4, %4 op_sel_hi:[0,0,0]\n\t"
                               "v_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_exp_f32 %12, %12\n\t"
                               "v_mfma_scale_f32_32x32x64_f8f6f4 %1, %2, %3, 0, %4, %4 op_sel_hi:[0,0,0]\n\t"
                               "v_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_exp_f32 %12, %12\n\t"
                               "v_mfma_f32_32x32x16_bf16 %0, %13, %14, %0\n\t"
                               "v_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\t"
                               "v_mfma_f32_32x32x16_bf16 %1, %13, %14, %1\n\t"
                               "v_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_exp_f32 %12, %12\n\t")
                         : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(sc), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
                           "v"(*(float __attribute__((ext_vector_type(4)))*)&a), "v"(*(float __attribute__((ext_vector_type(4)))*)&b));
        } else if constexpr (MODE == 13) {  // the same with 6 x v_max3 in place of the exps behind the bf16 MFMAs (the phase's 24 exp per 4 MFMAs kept)
            asm volatile(REP16("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %2, %3, 0, %4, %4 op_sel_hi:[0,0,0]\n\t"
                               "v_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_exp_f32 %12, %12\n\t"
                               "v_mfma_f32_32x32x16_bf16 %0, %13, %14, %0\n\t"
                               "v_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\t")
                         : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(sc), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
                           "v"(*(float __attribute__((ext_vector_type(4)))*)&a), "v"(*(float __attribute__((ext_vector_type(4)))*)&b));
        } else if constexpr (MODE == 10) {  // 16 v_permlane32_swap with s_nop 1
            asm volatile(REP16("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\t") : "+v"(v[0]), "+v"(v[1]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i] + c0[i] + c1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + pk;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
static void run(const char* what, double per) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, 0.5f);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, 0.5f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += x;
    printf("%-70s %8.1f cycles per %s\n", what, s / 256 / 64 / per, per == 16 ? "instruction" : "group");
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("v_exp_f32, dependent chain", 16);
    run<1>("v_exp_f32, independent", 16);
    run<2>("v_cvt_scalef32_pk_fp8_f32", 16);
    run<3>("v_max3_f32 (chain)", 16);
    run<4>("v_cvt_pk_fp8_f32", 16);
    run<5>("2 x v_mfma_scale_f32_32x32x64_f8f6f4 back to back", 16);
    run<6>("e4m3 MFMA (VGPR acc) + 8 v_exp", 16);
    run<7>("e4m3 MFMA (VGPR acc) + 4 v_exp + 4 cvt_scalef32 + 4 v_max3", 16);
    run<8>("bf16 32x32x16 MFMA + 4 v_exp", 16);
    run<9>("e4m3 MFMA (AGPR acc) + 8 v_exp", 16);
    run<10>("s_nop 1 + v_permlane32_swap", 16);
    run<11>("e4m3 MFMA (fresh VGPR dst, C = 0) + 8 v_exp", 16);
    run<12>("F8(c0) 8exp F8(c1) 8exp B16(c0) 4exp B16(c1) 4exp  [24 exp = 211 cycles]", 16);
    run<13>("F8(c0) 8exp B16(c0) 4exp (dependent, adjacent)  [12 exp = 106 cycles]", 16);
    return 0;
}
