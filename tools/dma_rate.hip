// Diagnostic: per-CU throughput of the global -> LDS paths from L2-resident data (what bounds a GEMM tile's bytes per MFMA).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/dma_rate.hip -o tools/dma_rate ; run: tools/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// MODE 0: global_load_lds_dwordx4 (LDS-DMA); 1: global_load_dwordx4 into registers only; 2: registers then ds_write_b128
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void rate_kernel(const uint4* src, int reps, unsigned long long* cyc, uint4* sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[WAVES * 8 * 1024];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint4* g = src + (blockIdx.x % 64) * 4096 + w * 512 + lane;   // 64 KB window per block, 8 KB per wave: L2 / L1 resident
    unsigned char* l = smem + w * 8 * 1024;
    uint4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) __builtin_amdgcn_global_load_lds(g + i * 64, (lds_ptr_t)(l + i * 1024), 16, 0, 0);
            else {
                uint4 v = g[i * 64];
                if (MODE == 2) *reinterpret_cast<uint4*>(l + i * 1024 + lane * 16) = v;
                else { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
            }
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (MODE != 0 && acc.x == 0x12345678) sink[threadIdx.x] = acc;
    if (MODE == 2 && reinterpret_cast<uint4*>(smem)[threadIdx.x].x == 0x12345678) sink[threadIdx.x] = acc;
}

template <int MODE, int WAVES>
int run(const uint4* src, unsigned long long* cyc, uint4* sink, const char* name) {
    const int reps = 2000, blocks = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rate_kernel<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, src, 200, cyc, sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate_kernel<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, src, reps, cyc, sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks);
    CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    double c = 0; for (auto v : h) c += v; c /= blocks;
    const double bytes = (double)reps * 8 * 1024 * WAVES;
    printf("%-34s %d waves/CU: %7.1f B/clk/CU (s_memtime ticks), %6.2f TB/s chip, %5.0f cycles per 1 KB piece per wave\n", name, WAVES, bytes / c,
           bytes * blocks / (ms * 1e-3) / 1e12, c / (reps * 8.0));
    return 0;
}

int main() {
    uint4 *src, *sink; unsigned long long* cyc;
    CK(hipMalloc(&src, 64 * 65536)); CK(hipMemset(src, 1, 64 * 65536)); CK(hipMalloc(&sink, 65536)); CK(hipMalloc(&cyc, 256 * 8));
    if (run<0, 4>(src, cyc, sink, "global_load_lds_dwordx4")) return 1;
    if (run<0, 8>(src, cyc, sink, "global_load_lds_dwordx4")) return 1;
    if (run<1, 4>(src, cyc, sink, "global_load_dwordx4 -> VGPR")) return 1;
    if (run<1, 8>(src, cyc, sink, "global_load_dwordx4 -> VGPR")) return 1;
    if (run<2, 4>(src, cyc, sink, "global_load_dwordx4 -> ds_write_b128")) return 1;
    if (run<2, 8>(src, cyc, sink, "global_load_dwordx4 -> ds_write_b128")) return 1;
    return 0;
}
