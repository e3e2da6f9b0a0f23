// Probe of the MX-scaled fp8 MFMA on gfx950 with exact small-integer data (diagnostic, not part of the product):
//   v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands (cbsz = blgp = 0) and unit block scales (e8m0 exponent 127).
// Hypothesis checked: lane l (r = l & 31, h = l >> 5) holds A[row r][k = 32h + j] and B[k = 32h + j][col r] in byte j of its 8 dwords;
// C/D as for the bf16 32x32 form. Prints the number of mismatches against an integer matmul.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_fp8_probe.hip -o tools/mfma_fp8_probe && ./tools/mfma_fp8_probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp8.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void probe32(const uint8_t* A, const uint8_t* B, float* C, int scale_word) {  // A [32][64], B [64][32] fp8 bytes; C [32][32]
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) {
        uint32_t av = 0, bv = 0;
        for (int e = 0; e < 4; ++e) {
            const int k = 32 * h + 4 * w + e;
            av |= (uint32_t)A[r * 64 + k] << (8 * e);
            bv |= (uint32_t)B[k * 32 + r] << (8 * e);
        }
        a[w] = (int)av; b[w] = (int)bv;
    }
    f32x16 c;
    for (int g = 0; g < 16; ++g) c[g] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_word, 0, scale_word);
    for (int g = 0; g < 16; ++g) C[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];
}
__global__ void probe16(const uint8_t* A, const uint8_t* B, float* C, int scale_word) {  // A [16][128], B [128][16]; C [16][16]
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) {
        uint32_t av = 0, bv = 0;
        for (int e = 0; e < 4; ++e) {
            const int k = 32 * q + 4 * w + e;
            av |= (uint32_t)A[r * 128 + k] << (8 * e);
            bv |= (uint32_t)B[k * 16 + r] << (8 * e);
        }
        a[w] = (int)av; b[w] = (int)bv;
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_word, 0, scale_word);
    for (int g = 0; g < 4; ++g) C[(4 * q + g) * 16 + r] = c[g];
}
static uint8_t f8(int v) {  // small integers are exact in e4m3: sign | exp(4, bias 7) | mant(3)
    if (v == 0) return 0;
    uint8_t s = v < 0 ? 0x80 : 0;
    int a = v < 0 ? -v : v, e = 0;
    while ((a >> (e + 1)) != 0) ++e;                 // a in [2^e, 2^(e+1))
    int mant = ((a << 3) >> e) & 7;                   // exact for a < 16
    return s | (uint8_t)((e + 7) << 3) | (uint8_t)mant;
}
int main() {
    for (int shape = 0; shape < 2; ++shape) {
        const int M = shape ? 16 : 32, K = shape ? 128 : 64, N = M;
        std::vector<uint8_t> A(M * K), B(K * N);
        std::vector<int> Ai(M * K), Bi(K * N);
        for (int i = 0; i < M; ++i) for (int k = 0; k < K; ++k) { Ai[i * K + k] = ((i * 3 + k * 5) % 7) - 3; A[i * K + k] = f8(Ai[i * K + k]); }
        for (int k = 0; k < K; ++k) for (int j = 0; j < N; ++j) { Bi[k * N + j] = ((k * 2 + j * 7) % 5) - 2; B[k * N + j] = f8(Bi[k * N + j]); }
        uint8_t *dA, *dB; float* dC;
        hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, M * N * 4);
        hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
        for (int sw : {(int)0x7F7F7F7F, (int)0x80808080}) {
            if (shape) hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, dA, dB, dC, sw);
            else hipLaunchKernelGGL(probe32, dim3(1), dim3(64), 0, 0, dA, dB, dC, sw);
            std::vector<float> C(M * N);
            hipMemcpy(C.data(), dC, M * N * 4, hipMemcpyDeviceToHost);
            int bad = 0; double ratio = 0;
            for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) {
                long ref = 0;
                for (int k = 0; k < K; ++k) ref += (long)Ai[i * K + k] * Bi[k * N + j];
                if ((float)ref != C[i * N + j]) ++bad;
                if (ref != 0) ratio = C[i * N + j] / (double)ref;
            }
            printf("%dx%dx%d scale word 0x%08X: %d / %d mismatches (last got/ref ratio %.3f); C[0][0..3] = %g %g %g %g\n", M, N, K, (unsigned)sw, bad, M * N, ratio,
                   C[0], C[1], C[2], C[3]);
        }
    }
    return 0;
}
