// Known-bytes calibration of rocprofv3's FETCH_SIZE for the access shapes the conv / GEMM kernels use.
// Every kernel below reads each byte of one 604 MB table exactly once (table > 256 MiB Infinity Cache, fresh pages between kernels
// through a 604 MB memset of a second table), so FETCH_SIZE x correction should equal ROWS * ROW_BYTES for each of them.
//   calib_coalesced   : global_load_dwordx4, one contiguous KB per wave instruction (gn_apply / layernorm shape)
//   calib_dma_contig  : global_load_lds_dwordx4, one contiguous KB per instruction
//   calib_dma_rows64  : global_load_lds_dwordx4, a piece = 16 rows x 64 B (gemm_pp_kernel's A / B pieces, row pitch 2304 B)
//   calib_dma_rows128 : global_load_lds_dwordx4, a piece = 8 rows x 128 B (halo-conv pixel rows at Cin = 64 per chunk pair)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/fetch_calib.hip -o tools/fetch_calib
// Run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib -o c -- tools/fetch_calib ; python tools/pmc_sum.py gpurun_out/calib calib
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int ROW_BYTES = 2304;      // K = 1152 bf16
constexpr int ROWS = 262144;         // 604 MB
constexpr int STEPS = ROW_BYTES / 64;  // 36 k-steps of 64 B per row = 36 KB per wave in every kernel

__global__ __launch_bounds__(256) void calib_coalesced(const unsigned char* src, uint4* sink) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned char* g = src + wave * (16L * ROW_BYTES) + lane * 16;
    uint4 acc = {0, 0, 0, 0};
#pragma unroll 4
    for (int i = 0; i < STEPS; ++i) {
        uint4 v = *reinterpret_cast<const uint4*>(g + i * 1024);
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[threadIdx.x] = acc;
}

// MODE 0: contiguous KB; 1: 16 rows x 64 B; 2: 8 rows x 128 B
template <int MODE>
__global__ __launch_bounds__(256) void calib_dma(const unsigned char* src, uint4* sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * 4 * 1024];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long wave = (long)blockIdx.x * 4 + w;
    const unsigned char* base = src + wave * (16L * ROW_BYTES);
    const unsigned char* g;
    long step;
    if (MODE == 0) { g = base + lane * 16; step = 1024; }
    else if (MODE == 1) { g = base + (long)(lane >> 2) * ROW_BYTES + (lane & 3) * 16; step = 64; }
    else { g = base + (long)(lane >> 3) * ROW_BYTES + (lane & 7) * 16; step = 128; }
    unsigned char* l = smem + w * 4096;
    if (MODE == 2) {
        // rows 0-7 over 18 steps of 128 B, then rows 8-15
#pragma unroll 1
        for (int half = 0; half < 2; ++half)
#pragma unroll 2
            for (int i = 0; i < STEPS / 2; ++i)
                __builtin_amdgcn_global_load_lds(g + (long)half * 8 * ROW_BYTES + i * step, (lds_ptr_t)(l + (i & 3) * 1024), 16, 0, 0);
    } else {
#pragma unroll 4
        for (int i = 0; i < STEPS; ++i) __builtin_amdgcn_global_load_lds(g + i * step, (lds_ptr_t)(l + (i & 3) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (reinterpret_cast<uint4*>(smem)[threadIdx.x].x == 0x12345678u) sink[threadIdx.x] = reinterpret_cast<uint4*>(smem)[threadIdx.x];
}

int main() {
    unsigned char *src, *flush; uint4* sink;
    const size_t bytes = (size_t)ROWS * ROW_BYTES;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&flush, bytes)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(src, 1, bytes));
    const int blocks = ROWS / 16 / 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"calib_coalesced", "calib_dma<0> contiguous KB", "calib_dma<1> 16 rows x 64 B", "calib_dma<2> 8 rows x 128 B"};
    for (int rep = 0; rep < 2; ++rep)
        for (int k = 0; k < 4; ++k) {
            CK(hipMemset(flush, rep + k, bytes));   // evict the table from the Infinity Cache
            CK(hipEventRecord(e0));
            if (k == 0) hipLaunchKernelGGL(calib_coalesced, dim3(blocks), dim3(256), 0, 0, src, sink);
            if (k == 1) hipLaunchKernelGGL(calib_dma<0>, dim3(blocks), dim3(256), 0, 0, src, sink);
            if (k == 2) hipLaunchKernelGGL(calib_dma<1>, dim3(blocks), dim3(256), 0, 0, src, sink);
            if (k == 3) hipLaunchKernelGGL(calib_dma<2>, dim3(blocks), dim3(256), 0, 0, src, sink);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-32s known bytes %.1f MB (%.0f KB)  %.3f ms  %.2f TB/s\n", names[k], bytes / 1e6, bytes / 1024.0, ms, bytes / (ms * 1e-3) / 1e12);
        }
    return 0;
}
