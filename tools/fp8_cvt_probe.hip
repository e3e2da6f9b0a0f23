// Probe (diagnostic, not part of the product): semantics of v_cvt_scalef32_pk_fp8_f32 and of the per-lane E8M0 block scales of
// v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950 - what flash_attn_fp8_kernel relies on.
//   hipcc --offload-arch=gfx950 -O2 tools/fp8_cvt_probe.hip -o tools/fp8_cvt_probe && ./tools/fp8_cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) short s16x2;

__global__ void cvt_probe(const float* x, const float* scale, uint32_t* out) {
    const int l = threadIdx.x;
    s16x2 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x[2 * l], x[2 * l + 1], scale[l], false);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, -x[2 * l], 2.f * x[2 * l + 1], scale[l], true);
    out[l] = __builtin_bit_cast(uint32_t, r);
}
// A = 1.0 (0x38) everywhere, B = 1.0 everywhere, scale_a byte per lane sa[l], scale_b byte sb[l] (byte 0 of the register):
// C[i][j] = sum_k 2^(sa(i, k/32) - 127) * 2^(sb(j, k/32) - 127)
__global__ void scale_probe(const int* sa, const int* sb, float* C) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) { a[w] = 0x38383838; b[w] = 0x38383838; }
    f32x16 c;
    for (int g = 0; g < 16; ++g) c[g] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[l], 0, sb[l]);
    for (int g = 0; g < 16; ++g) C[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];
}
// Which bytes of a lane belong to which 32-k scale block? A = 1.0 in bytes 0..15 of every lane and 0 in bytes 16..31, B = 1.0:
//   blocks by BYTE RANGE (block b = bytes 16b .. 16b+15 of both lane halves): C[i][j] = 32 * 2^(sa[i] - 127) * 2^(sb[j] - 127)
//   blocks by LANE HALF  (block h = the 32 bytes of lane half h):             C[i][j] = 16 * (2^(sa[i]+sb[j]-254) + 2^(sa[i+32]+sb[j+32]-254))
__global__ void block_probe(const int* sa, const int* sb, float* C) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) { a[w] = w < 4 ? 0x38383838 : 0; b[w] = 0x38383838; }
    f32x16 c;
    for (int g = 0; g < 16; ++g) c[g] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[l], 0, sb[l]);
    for (int g = 0; g < 16; ++g) C[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];
}
static float f8_to_f(uint8_t v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m, -9) : ((e == 15 && m == 7) ? NAN : ldexpf(1.f + m / 8.f, e - 7));
    return s ? -x : x;
}
int main() {
    std::vector<float> x(128), sc(64);
    const float vals[16] = {1.f, 3.f, 0.3f, 447.f, 448.f, 500.f, 1e-3f, 100.f, 0.f, 17.f, 1e4f, 255.9f, 1.0625f, 1.1875f, 2e-3f, 64.f};
    for (int i = 0; i < 64; ++i) { x[2 * i] = vals[i % 16]; x[2 * i + 1] = vals[(i + 5) % 16]; sc[i] = i < 16 ? 1.f : (i < 32 ? 4.f : (i < 48 ? 0.25f : 6.f)); }
    float *dx, *ds; uint32_t* dout;
    hipMalloc(&dx, 512); hipMalloc(&ds, 256); hipMalloc(&dout, 256);
    hipMemcpy(dx, x.data(), 512, hipMemcpyHostToDevice); hipMemcpy(ds, sc.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, dx, ds, dout);
    std::vector<uint32_t> out(64);
    hipMemcpy(out.data(), dout, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; i += 3) {
        printf("x = (%g, %g) scale %g -> lo (%g, %g) hi (%g, %g)   [x/scale = %g, %g]\n", x[2 * i], x[2 * i + 1], sc[i], f8_to_f(out[i] & 255),
               f8_to_f((out[i] >> 8) & 255), f8_to_f((out[i] >> 16) & 255), f8_to_f(out[i] >> 24), x[2 * i] / sc[i], x[2 * i + 1] / sc[i]);
    }
    std::vector<int> sa(64), sb(64);
    for (int l = 0; l < 64; ++l) { sa[l] = 127 + (l & 31) % 3 + 4 * (l >> 5); sb[l] = 127 - (l & 31) % 2 + 2 * (l >> 5); }
    int *dsa, *dsb; float* dC;
    hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dC, 4096);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(scale_probe, dim3(1), dim3(64), 0, 0, dsa, dsb, dC);
    std::vector<float> C(1024);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float ref = 0.f;
        for (int hb = 0; hb < 2; ++hb) ref += 32.f * ldexpf(1.f, (sa[i + 32 * hb] - 127) + (sb[j + 32 * hb] - 127));
        if (ref != C[i * 32 + j]) ++bad;
    }
    printf("per-lane block scales (lane = row/col r + 32 * k-block): %d / 1024 mismatches; C[0][0] = %g C[1][1] = %g C[2][0] = %g\n", bad, C[0], C[33], C[64]);
    hipLaunchKernelGGL(block_probe, dim3(1), dim3(64), 0, 0, dsa, dsb, dC);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad_range = 0, bad_half = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        const float by_range = 32.f * ldexpf(1.f, sa[i] + sb[j] - 254);
        const float by_half = 16.f * (ldexpf(1.f, sa[i] + sb[j] - 254) + ldexpf(1.f, sa[i + 32] + sb[j + 32] - 254));
        if (C[i * 32 + j] != by_range) ++bad_range;
        if (C[i * 32 + j] != by_half) ++bad_half;
    }
    printf("scale-block membership: %d mismatches if block b = bytes [16b, 16b+16) of every lane; %d mismatches if block h = lane half h (C[0][0] = %g)\n",
           bad_range, bad_half, C[0]);
    return 0;
}
