// What does the memory system give a read + write pass of GroupNorm-apply's shape (bf16 in, bf16 out, 16-byte vectors, 1-2 GB tensors)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/copy_rate tools/copy_rate.hip && tools/bin/copy_rate
// Variants: plain copy / scale+shift+SiLU arithmetic; grid-stride (the product kernel's order) / one contiguous span per workgroup;
// temporal / non-temporal loads and stores; in place / out of place; 4 / 8 loads in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    uint32_t r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

template <bool ARITH>
__device__ __forceinline__ uint4 work(uint4 v, const float (&sc)[8], const float (&sh)[8]) {
    if constexpr (!ARITH) return v;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a = silu(bflo(w[e]) * sc[2 * e] + sh[2 * e]);
        float b = silu(bfhi(w[e]) * sc[2 * e + 1] + sh[2 * e + 1]);
        o[e] = pack2(a, b);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ uint4 ld(const uint4* p) {
    if constexpr (NT) { const u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); return make_uint4(r.x, r.y, r.z, r.w); }
    return *p;
}
template <bool NT> __device__ __forceinline__ void st(uint4* p, uint4 v) {
    if constexpr (NT) __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(p));
    else *p = v;
}

// order 0: grid-stride (thread i, i + stride, ...), U loads in flight. order 1: workgroup b owns the contiguous span [b * span, (b+1) * span).
template <bool ARITH, bool NTL, bool NTS, int U, int ORDER>
__global__ __launch_bounds__(256) void pass_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, long nvec, const float* __restrict__ tab) {
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = tab[(threadIdx.x & 15) * 8 + e]; sh[e] = tab[128 + (threadIdx.x & 15) * 8 + e]; }
    long i, end, stride;
    if (ORDER == 0) { stride = (long)gridDim.x * 256; i = (long)blockIdx.x * 256 + threadIdx.x; end = nvec; }
    else { const long span = (nvec + gridDim.x - 1) / gridDim.x; i = (long)blockIdx.x * span + threadIdx.x; end = min(nvec, (long)(blockIdx.x + 1) * span); stride = 256; }
    for (; i + (U - 1) * stride < end; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NTL>(x + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) st<NTS>(y + i + u * stride, work<ARITH>(v[u], sc, sh));
    }
    for (; i < end; i += stride) st<NTS>(y + i, work<ARITH>(ld<NTL>(x + i), sc, sh));
}

template <bool ARITH, bool NTL, bool NTS, int U, int ORDER>
static void run(const char* name, const uint4* x, uint4* y, long nvec, const float* tab, int blocks_per_cu) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * blocks_per_cu;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((pass_kernel<ARITH, NTL, NTS, U, ORDER>), dim3(grid), dim3(256), 0, 0, x, y, nvec, tab);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((pass_kernel<ARITH, NTL, NTS, U, ORDER>), dim3(grid), dim3(256), 0, 0, x, y, nvec, tab);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-58s %2d blk/CU  %8.1f us  %6.2f TB/s (read + write)\n", name, blocks_per_cu, ms * 1e3, 2.0 * nvec * 16 / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const long bytes = argc > 1 ? atol(argv[1]) : (1L << 30);   // 2048 x 2048 x 128 bf16 = 1 GiB
    const long nvec = bytes / 16;
    uint4 *x, *y;
    float* tab;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&tab, 256 * 4));
    std::vector<uint16_t> h(bytes / 2);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((s >> 16) & 0x3ff) + ((s >> 31) << 15)); }   // bf16 values around +-1
    CK(hipMemcpy(x, h.data(), bytes, hipMemcpyHostToDevice));
    std::vector<float> t(256, 1.0f);
    for (int i = 128; i < 256; ++i) t[i] = 0.1f;
    CK(hipMemcpy(tab, t.data(), 1024, hipMemcpyHostToDevice));
    printf("tensor %.2f GiB, 16-byte vectors\n", bytes / 1073741824.0);
    for (int b : {8, 16, 32}) {
        run<false, false, false, 4, 0>("copy, grid-stride, 4 in flight", x, y, nvec, tab, b);
        run<true, false, false, 4, 0>("gn+silu, grid-stride, 4 in flight (product shape)", x, y, nvec, tab, b);
        run<true, false, false, 8, 0>("gn+silu, grid-stride, 8 in flight", x, y, nvec, tab, b);
        run<false, false, false, 4, 1>("copy, contiguous span per workgroup, 4 in flight", x, y, nvec, tab, b);
        run<true, false, false, 4, 1>("gn+silu, contiguous span per workgroup, 4 in flight", x, y, nvec, tab, b);
        run<true, false, false, 8, 1>("gn+silu, contiguous span per workgroup, 8 in flight", x, y, nvec, tab, b);
        run<true, true, true, 4, 0>("gn+silu, grid-stride, nt loads + nt stores", x, y, nvec, tab, b);
        run<true, true, false, 4, 0>("gn+silu, grid-stride, nt loads", x, y, nvec, tab, b);
        run<true, false, true, 4, 0>("gn+silu, grid-stride, nt stores", x, y, nvec, tab, b);
        run<true, true, true, 8, 1>("gn+silu, contiguous span, nt both, 8 in flight", x, y, nvec, tab, b);
    }
    run<true, false, false, 4, 0>("gn+silu IN PLACE, grid-stride, 4 in flight", x, x, nvec, tab, 16);
    run<true, false, false, 8, 1>("gn+silu IN PLACE, contiguous span, 8 in flight", x, x, nvec, tab, 16);
    run<false, false, false, 4, 0>("copy IN PLACE, grid-stride", x, x, nvec, tab, 16);
    return 0;
}
