// Do two INDEPENDENT 4-wave workgroups on one CU (two waves per SIMD) overlap their phases - one's vector-only stretch (an epilogue) under the
// other's MFMA stream - or do they run in lockstep (round 1's observation on the 4-wave conv / attention kernels: "their times add")? And does
// a static s_setprio difference between the two break the lockstep?
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/coresident_probe tools/coresident_probe.hip && tools/bin/coresident_probe
// Each workgroup repeats: [64 x v_mfma_f32_16x16x32_bf16, back to back, operands in registers] [E dependent-free VALU instructions]. One
// workgroup per CU (grid 256) gives the single-resident time per iteration; two per CU (grid 512, 72 KB of LDS each so that exactly two fit)
// the co-resident time. Perfect overlap: the pair takes max(2 x 1024, 1024 + 4E) matrix-cycles per iteration pair; lockstep: 2 x (1024 + 4E).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int PRIO>   // 0: no priority; 1: workgroups 256.. run at s_setprio 3, the first 256 at 0; 2: the other way round
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters, int valu_blocks, unsigned long long* clk) {
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const bool second = blockIdx.x >= 256;
    if (PRIO == 1 && second) __builtin_amdgcn_s_setprio(3);
    if (PRIO == 2 && !second) __builtin_amdgcn_s_setprio(3);
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * ((lane * 7 + i * 3) % 13 - 6)); b[i] = (__bf16)(0.02f * ((lane * 5 + i) % 11 - 5)); }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + lane * 1e-3f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        for (int e = 0; e < valu_blocks; ++e) {   // 32 independent-ish VALU instructions per block
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0000001f, 1e-7f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + smem[threadIdx.x];
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int PRIO>
static void run(const char* name, int grid, int iters, int valu_blocks, float* out, unsigned long long* clk) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = 72 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<PRIO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(probe<PRIO>, dim3(grid), dim3(256), lds, 0, out, iters / 4, valu_blocks, clk);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<PRIO>, dim3(grid), dim3(256), lds, 0, out, iters, valu_blocks, clk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CK(hipMemcpy(h.data(), clk, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto c : h) mean += (double)c;
    mean /= grid;
    const double wg_per_cu = grid / 256.0;
    // matrix cycles one SIMD spends per iteration of ONE workgroup: 64 x 16 = 1024
    printf("%-34s grid %4d  VALU %4d/iter  %8.3f ms  %9.0f cycles per iteration and workgroup  MFMA pipe busy %.3f\n", name, grid, valu_blocks * 32, ms,
           mean / iters, wg_per_cu * 1024.0 * iters / mean);
    fflush(stdout);
}

int main() {
    float* out;
    unsigned long long* clk;
    CK(hipMalloc(&out, 512 * 256 * 4));
    CK(hipMalloc(&clk, 512 * 8));
    const int iters = 4000;
    for (int vb : {0, 4, 8, 16}) {
        run<0>("one workgroup per CU", 256, iters, vb, out, clk);
        run<0>("two per CU, equal priority", 512, iters, vb, out, clk);
        run<1>("two per CU, second at prio 3", 512, iters, vb, out, clk);
        run<2>("two per CU, first at prio 3", 512, iters, vb, out, clk);
    }
    return 0;
}
