// Probe (diagnostic, round 6): does the MFMA SHAPE change what the chip delivers once it runs at its power-limited clock? MI355X_MICROARCH.md ("DVFS
// give-back", item 7) reports 1.12-1.15 x the FLOP/s for v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16 at equal cycles per FLOP on random
// data; the attention kernels of this repository use the 32x32x16 shape and run at the power-limited clock (a truncating pack that removes 128 issue
// cycles per tile changes nothing inside the pipeline: profiles/r06_ab_pp2_trunc.txt). Whole chip (256 workgroups x 4 waves, one wave per SIMD), about
// one second of back-to-back launches per arm on RANDOM bf16 operands, wall time by HIP events; in-kernel clock = delta s_memtime / delta
// s_memrealtime x 100 MHz of workgroup 0. Arms: operands in registers; every A operand re-read from LDS (ds_read_b128) as the attention kernels do.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_power_probe.hip -o /tmp/mfma_power_probe && /tmp/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// SHAPE 0: 32x32x16 (32 cycles, 32 KFLOP per wave-instruction); 1: 16x16x32 (16 cycles, 16 KFLOP). LDS: A operand re-read from LDS per MFMA.
template <int SHAPE, bool LDS>
__global__ __launch_bounds__(256, 1) void probe(const uint4* __restrict__ rnd, float* out, unsigned long long* stamps, int iters) {
    __shared__ __attribute__((aligned(16))) uint4 lds[2048];   // 32 KB of random operands
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2048; i += 256) lds[i] = rnd[(blockIdx.x * 2048 + i) & 65535];
    __syncthreads();
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, rnd[(tid * 8 + i) & 65535]);
        b[i] = __builtin_bit_cast(bf16x8, rnd[(tid * 8 + 4 + i) & 65535]);
    }
    f32x16 c[4];
    f32x4 d[8];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
    for (int i = 0; i < 8; ++i) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint4* lp = lds + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (LDS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, lp[((it * 8 + u) * 4 + i) * 64 & 1984]);
            }
            if constexpr (SHAPE == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], c[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) d[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i + (i >> 2)) & 3], d[i], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc += c[i][e];
    for (int i = 0; i < 8; ++i) acc += d[i][0] + d[i][1] + d[i][2] + d[i][3];
    out[blockIdx.x * 256 + tid] = acc;
    if (tid == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

template <int SHAPE, bool LDS>
void run(const char* name, const uint4* rnd, float* out, unsigned long long* stamps) {
    const int iters = 4096, launches = 60;
    // FLOP per wave and iteration: 8 x 4 MFMAs of 32x32x16 (32768 FLOP each) or 8 x 8 of 16x16x32 (16384 each) = 1 048 576 either way
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((probe<SHAPE, LDS>), dim3(256), dim3(256), 0, 0, rnd, out, stamps, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((probe<SHAPE, LDS>), dim3(256), dim3(256), 0, 0, rnd, out, stamps, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost);
    const double flop = 1048576.0 * iters * 4.0 * 256.0 * launches;
    printf("%-58s %7.1f ms  %8.1f TFLOP/s  in-kernel clock %6.0f MHz  cycles per 32 KFLOP %6.2f\n", name, ms, flop / ms / 1e9, (double)h[0] / (double)h[1] * 100.0,
           (double)h[0] / (iters * 32.0));
}

int main() {
    std::vector<uint16_t> hbits(65536 * 8);
    srand(7);
    for (auto& v : hbits) {   // random bf16 in (-2, 2): random sign, exponent 118..127, random mantissa
        const unsigned s = rand() & 1, e = 118 + rand() % 10, m = rand() & 127;
        v = (uint16_t)((s << 15) | (e << 7) | m);
    }
    uint4* rnd; float* out; unsigned long long* stamps;
    hipMalloc(&rnd, 65536 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&stamps, 16);
    hipMemcpy(rnd, hbits.data(), 65536 * 16, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, false>("32x32x16, operands in registers", rnd, out, stamps);
        run<1, false>("16x16x32, operands in registers", rnd, out, stamps);
        run<0, true>("32x32x16, A operand re-read from LDS per MFMA", rnd, out, stamps);
        run<1, true>("16x16x32, A operand re-read from LDS per 2 MFMAs", rnd, out, stamps);
    }
    return 0;
}
