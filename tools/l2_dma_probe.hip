// Probe (diagnostic): L2 -> LDS bandwidth of LDS-DMA pieces by piece shape, on an L2-resident table (every workgroup streams the same 2 MB
// region again and again, as the B operand of gemm_pp_kernel is streamed): is a 64-byte row piece (BK = 32 bf16) served at half the rate of
// a 128-byte one? Prints TB/s aggregate over 256 workgroups of 8 waves.
//   hipcc --offload-arch=gfx950 -O3 tools/l2_dma_probe.hip -o tools/l2_dma_probe && ./tools/l2_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int ROW_BYTES = 2304;            // K = 1152 bf16
constexpr int ROWS = 896;                  // 2.06 MB: L2-resident per XCD
constexpr int REPS = 64;

// MODE 0: contiguous 1 KB per instruction; 1: 16 rows x 64 B; 2: 8 rows x 128 B; 3: 4 rows x 256 B
template <int MODE>
__global__ __launch_bounds__(512) void probe(const unsigned char* src, uint4* sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[8 * 8 * 1024];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int RPP = MODE == 0 ? 1 : (MODE == 1 ? 16 : (MODE == 2 ? 8 : 4));   // rows per piece
    constexpr int BPR = 1024 / RPP;                                                // bytes per row and piece
    constexpr int LPR = BPR / 16;                                                  // lanes per row
    unsigned char* l = smem + w * 8192;
    for (int rep = 0; rep < REPS; ++rep) {
        // the wave walks row groups w, w + 8, ... ; within a group all column pieces
        for (int rg = w; rg < ROWS / RPP; rg += 8) {
            const unsigned char* g = MODE == 0 ? src + (long)rg * 1024 + lane * 16
                                               : src + ((long)rg * RPP + lane / LPR) * ROW_BYTES + (lane % LPR) * 16;
            constexpr int NP = MODE == 0 ? 1 : ROW_BYTES / BPR;
#pragma unroll
            for (int i = 0; i < NP; ++i) __builtin_amdgcn_global_load_lds(g + i * BPR, (lds_ptr_t)(l + (i & 7) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (reinterpret_cast<uint4*>(smem)[threadIdx.x].x == 0x12345678u) sink[threadIdx.x] = reinterpret_cast<uint4*>(smem)[threadIdx.x];
}

template <int MODE>
void run(const char* name, const unsigned char* src, uint4* sink, double bytes_per_wg) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, src, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, src, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.3f ms  %.2f TB/s L2 -> LDS aggregate (%.1f GB/s per CU)\n", name, ms, 256 * bytes_per_wg / (ms * 1e-3) / 1e12, bytes_per_wg / (ms * 1e-3) / 1e9);
}
int main() {
    unsigned char* src; uint4* sink;
    hipMalloc(&src, (size_t)ROWS * ROW_BYTES + 65536); hipMalloc(&sink, 8192);
    hipMemset(src, 1, (size_t)ROWS * ROW_BYTES + 65536);
    const double full = (double)REPS * ROWS * ROW_BYTES;   // modes 1-3 read every byte of the table per rep; mode 0 reads ROWS KB per rep
    run<0>("contiguous 1 KB", src, sink, (double)REPS * (ROWS) * 1024);
    run<1>("16 rows x 64 B", src, sink, full);
    run<2>("8 rows x 128 B", src, sink, full);
    run<3>("4 rows x 256 B", src, sink, full);
    return 0;
}
