// Standalone probe: semantics of __builtin_amdgcn_global_load_lds (16-byte LDS-DMA) on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ void probe(const uint4* __restrict__ src, uint4* dst, const uint4* zero_page) {
    __shared__ __attribute__((aligned(16))) uint4 lds[4 * 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // each wave writes 64 chunks to lds[w*64 ..]; lane l fetches source chunk perm(l); odd lanes of wave 3 fetch the zero page
    const int perm = (lane * 7 + 3) & 63;
    const uint4* g = src + w * 64 + perm;
    if (w == 3 && (lane & 1)) g = zero_page;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)(&lds[wu * 64]), 16, 0, 0);
    __syncthreads();   // emits vmcnt(0) + barrier
    dst[tid] = lds[tid];
}

int main() {
    const int N = 256;
    uint4 h[N], out[N];
    for (int i = 0; i < N; ++i) h[i] = make_uint4(i, i * 2, i * 3, 0xabc00000u + i);
    uint4 *d, *o, *z;
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof out); hipMalloc(&z, 256);
    hipMemset(z, 0, 256);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, o, z);
    hipMemcpy(out, o, sizeof out, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < N; ++t) {
        int w = t >> 6, lane = t & 63, perm = (lane * 7 + 3) & 63;
        uint32_t want = (w == 3 && (lane & 1)) ? 0 : (uint32_t)(w * 64 + perm);
        if (out[t].x != want) { if (bad < 8) printf("mismatch t=%d got %u want %u\n", t, out[t].x, want); ++bad; }
    }
    printf("glds probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK: LDS[base + lane*16] <- per-lane global address", bad);
    return bad != 0;
}
