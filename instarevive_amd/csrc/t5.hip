// T5 v1.1 encoder kernels (prompt-embedding producer: reference diffusion/model/t5.py:82-101 -> transformers.T5EncoderModel; SURVEY.md
// section 8(f) N3). The encoder runs ONCE per prompt on <= 512 tokens, so only its linears (igemm.hip) are MFMA kernels; what is here
// is the glue around them, written for clarity: embedding gather, RMSNorm, self-attention with the additive relative-position bias
// (no 1/sqrt(d) scaling) and padding mask, and the gated-GELU product.
#include <hip/hip_runtime.h>
#include "common.h"
#include "kernels.h"

// x[row][:] = table[ids[row]][:] (bf16 table -> fp32 residual stream); ids outside [0, vocab) read row 0 and set *bad
__global__ __launch_bounds__(256) void t5_embed_kernel(const int* __restrict__ ids, const bf16_t* __restrict__ table, float* __restrict__ x,
                                                       int D, int vocab, int* __restrict__ bad) {
    const long row = blockIdx.x;
    int id = ids[row];
    if (id < 0 || id >= vocab) {
        if (threadIdx.x == 0) *bad = 1;
        id = 0;
    }
    const bf16_t* src = table + (long)id * D;
    for (int c = threadIdx.x * 2; c < D; c += 512) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(src + c);
        *reinterpret_cast<float2*>(x + row * D + c) = make_float2(bflo(w), bfhi(w));
    }
}

// T5LayerNorm: y = w * x * rsqrt(mean(x^2) + eps), statistics in fp32; one wave per row; yb (bf16) and / or yf (fp32) may be null
__global__ __launch_bounds__(256) void t5_rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w, bf16_t* __restrict__ yb,
                                                         float* __restrict__ yf, long rows, int D, float eps) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * D;
    float ss = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)D + eps);
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        const f32x4 g = *reinterpret_cast<const f32x4*>(w + c);
        const f32x4 o = {g[0] * (v[0] * rs), g[1] * (v[1] * rs), g[2] * (v[2] * rs), g[3] * (v[3] * rs)};
        if (yb) *reinterpret_cast<uint2*>(yb + row * D + c) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
        if (yf) *reinterpret_cast<f32x4*>(yf + row * D + c) = o;
    }
}

// Self-attention of one (batch, head, 64-query tile): scores = q.k + bias[h][q][k] (+ mask), softmax over keys, out = P v.
// qkv: [B][T][3*H*dk] bf16 (q | k | v column blocks); bias: [H][T][T] fp32; key_mask: [B][T] fp32, 1 = token (null: no padding);
// out: [B][T][H*dk] bf16. K (rows padded by 2 elements: conflict-free column walks) and V of the head live in LDS as bf16; a wave
// takes one query at a time: lane j owns keys j, j+64, ... for the scores, lane d owns output dim d for P v.
__global__ __launch_bounds__(256) void t5_attn_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ bias,
                                                      const float* __restrict__ key_mask, bf16_t* __restrict__ out, int T, int H, int dk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char t5_smem[];
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int ld = 3 * H * dk, kp = dk + 2;
    bf16_t* Ks = reinterpret_cast<bf16_t*>(t5_smem);            // [T][dk + 2]
    bf16_t* Vs = Ks + (long)T * kp;                              // [T][dk]
    float* Ps = reinterpret_cast<float*>(Vs + (long)T * dk);    // [4 waves][T]
    float* Qs = Ps + 4 * T;                                      // [4 waves][dk]
    const bf16_t* base = qkv + (long)b * T * ld + h * dk;
    for (int i = threadIdx.x; i < T * dk; i += 256) {
        const int j = i / dk, d = i - j * dk;
        Ks[j * kp + d] = base[(long)j * ld + H * dk + d];
        Vs[j * dk + d] = base[(long)j * ld + 2 * H * dk + d];
    }
    __syncthreads();
    float* P = Ps + wid * T;
    float* Q = Qs + wid * dk;
    const int nkeys = (T + 63) / 64;
    for (int qi = wid; qi < 64; qi += 4) {
        const int q = q0 + qi;
        if (q >= T) break;  // uniform per wave
        if (lane < dk) Q[lane] = bf2f(base[(long)q * ld + lane]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float s[8];
        float mx = -3.0e38f;
        for (int t = 0; t < nkeys; ++t) {
            const int j = t * 64 + lane;
            float a = -3.0e38f;
            if (j < T) {
                a = 0.f;
                for (int d = 0; d < dk; ++d) a += Q[d] * bf2f(Ks[j * kp + d]);
                a += bias[((long)h * T + q) * T + j];
                if (key_mask && key_mask[(long)b * T + j] < 0.5f) a = -3.0e38f;   // transformers adds finfo.min: exp underflows to exactly 0
            }
            s[t] = a;
            mx = fmaxf(mx, a);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int t = 0; t < nkeys; ++t) {
            const int j = t * 64 + lane;
            const float e = (j < T && s[t] > -1.0e38f) ? __expf(s[t] - mx) : 0.f;
            if (j < T) P[j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < dk) {
            float o = 0.f;
            for (int j = 0; j < T; ++j) o += P[j] * bf2f(Vs[j * dk + lane]);
            out[((long)b * T + q) * (H * dk) + h * dk + lane] = f2bf(o / sum);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// T5DenseGatedActDense: out = gelu_new(a) * b with a | b the two column halves of the fused wi_0 | wi_1 projection
__global__ __launch_bounds__(256) void t5_gated_gelu_kernel(const bf16_t* __restrict__ ab, bf16_t* __restrict__ out, long rows, int F) {
    const long nv = rows * (F / 2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        const long row = i / (F / 2);
        const int c = (int)(i - row * (F / 2)) * 2;
        const uint32_t a = *reinterpret_cast<const uint32_t*>(ab + row * 2 * F + c);
        const uint32_t g = *reinterpret_cast<const uint32_t*>(ab + row * 2 * F + F + c);
        *reinterpret_cast<uint32_t*>(out + row * F + c) = pack2bf(gelu_tanh(bflo(a)) * bflo(g), gelu_tanh(bfhi(a)) * bfhi(g));
    }
}

int ir_launch_t5_embed(const int* ids, const bf16_t* table, float* x, long rows, int D, int vocab, int* bad, hipStream_t s) {
    if (rows <= 0 || (D & 1)) return -2;
    hipLaunchKernelGGL(t5_embed_kernel, dim3((unsigned)rows), dim3(256), 0, s, ids, table, x, D, vocab, bad);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int ir_launch_t5_rmsnorm(const float* x, const float* w, bf16_t* yb, float* yf, long rows, int D, float eps, hipStream_t s) {
    if (rows <= 0 || (D & 3)) return -2;
    hipLaunchKernelGGL(t5_rmsnorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, w, yb, yf, rows, D, eps);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
size_t ir_t5_attn_lds(int T, int dk) { return (size_t)T * (dk + 2) * 2 + (size_t)T * dk * 2 + 4 * (size_t)T * 4 + 4 * (size_t)dk * 4; }
int ir_launch_t5_attn(const bf16_t* qkv, const float* bias, const float* key_mask, bf16_t* out, int B, int T, int H, int dk, hipStream_t s) {
    if (T <= 0 || T > 512 || dk > 64 || (dk & 1)) return -2;   // s[8] keys per lane; one output dim per lane
    const size_t lds = ir_t5_attn_lds(T, dk);
    if (lds > 160 * 1024) return -3;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(t5_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return -4;
    hipLaunchKernelGGL(t5_attn_kernel, dim3((T + 63) / 64, H, B), dim3(256), lds, s, qkv, bias, key_mask, out, T, H, dk);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int ir_launch_t5_gated_gelu(const bf16_t* ab, bf16_t* out, long rows, int F, hipStream_t s) {
    if (rows <= 0 || (F & 1)) return -2;
    const long nv = rows * (F / 2);
    long blocks = (nv + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(t5_gated_gelu_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ab, out, rows, F);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
