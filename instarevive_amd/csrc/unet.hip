// Memory-bound kernels of the ControlLDM one-step path (diffusion/cldm.py:568-588, ControlNet :58-292, the SD-2.1
// UNet of ldm/modules/diffusionmodules/openaimodel.py:411-786 and the SpatialTransformer of ldm/modules/attention.py:262-350). The
// matrix work of that path runs on the kernels of igemm.hip / attention.hip; what is here is what those do not cover:
//   * GroupNorm(32) over ANY channel count that is a multiple of 32 and 8 (320 ... 2560: 10 ... 80 channels per group; the VAE kernels of
//     norm.hip need C = 8 * 2^k <= 512), eps as an argument (1e-5 in the ResBlocks, 1e-6 in the SpatialTransformer);
//   * GEGLU (attention.py:48-56): a * gelu(g) over the two column halves of the ff.net.0.proj output;
//   * the latent <-> NHWC ends: cat(x, hint) -> 32-channel bf16 rows, and zT + v back to NCHW fp32;
//   * skip-connection rows into a concatenation buffer (cat([h, hs.pop()], dim=1), openaimodel.py:779).
// All tensors here are small next to the 2048 x 2048 VAE activations (latent resolution, <= 2560 channels): HBM-bound, 16-byte accesses.
#include "common.h"
#include "kernels.h"

namespace {

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : -1)

// ---- GroupNorm, stage 1: partial sums of one (pixel chunk, group, image). x: [N][HW][C] bf16, part: [N][G][chunks][2]
__global__ __launch_bounds__(256) void gn_any_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, long HW, int C, int cpg,
                                                             int chunks, long pix_per_chunk) {
    const int chunk = blockIdx.x, g = blockIdx.y, n = blockIdx.z, G = C / cpg;
    const long p0 = (long)chunk * pix_per_chunk;
    const long np = min(pix_per_chunk, HW - p0);
    const int c2 = cpg >> 1;   // channel pairs per group (cpg is even: C % 64 == 0 or cpg % 2 == 0 checked by the launcher)
    const bf16_t* base = x + ((long)n * HW + p0) * C + (long)g * cpg;
    float s = 0.f, q = 0.f;
    for (long e = threadIdx.x; e < np * c2; e += 256) {
        const long pix = e / c2;
        const int c = (int)(e - pix * c2) * 2;
        const uint32_t u = *reinterpret_cast<const uint32_t*>(base + pix * C + c);
        const float a = bflo(u), b = bfhi(u);
        s += a + b;
        q += a * a + b * b;
    }
    __shared__ float red[8];
    s = wave_sum(s);
    q = wave_sum(q);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w] = s; red[4 + w] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* dst = part + (((long)n * G + g) * chunks + chunk) * 2;
        dst[0] = (red[0] + red[1]) + (red[2] + red[3]);
        dst[1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}
// ---- stage 2: per-channel scale / shift tables [N][C] each (scale = gamma * rstd, shift = beta - mean * scale)
__global__ __launch_bounds__(256) void gn_any_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ scale, float* __restrict__ shift, long HW, int C, int cpg, int chunks,
                                                              float eps) {
    const int n = blockIdx.x, G = C / cpg;
    __shared__ float mean_s[64], rstd_s[64];
    for (int g = threadIdx.x; g < G; g += 256) {
        const float* p = part + ((long)n * G + g) * chunks * 2;
        double s = 0.0, q = 0.0;
        for (int i = 0; i < chunks; ++i) { s += p[2 * i]; q += p[2 * i + 1]; }
        const double cnt = (double)HW * cpg, m = s / cnt;
        double var = q / cnt - m * m;
        if (var < 0.0) var = 0.0;
        mean_s[g] = (float)m;
        rstd_s[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c / cpg;
        const float sc = gamma[c] * rstd_s[g];
        scale[(long)n * C + c] = sc;
        shift[(long)n * C + c] = beta[c] - mean_s[g] * sc;
    }
}
// ---- stage 3: y = x * scale[c] + shift[c] (optionally SiLU), 8 channels per thread
__global__ __launch_bounds__(256) void gn_any_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, long HW, int C, int do_silu, long total_vec) {
    const int vpp = C >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (long)gridDim.x * 256) {
        const long pix = i / vpp;
        const int c0 = (int)(i - pix * vpp) * 8;
        const long n = pix / HW;
        const uint4 u = *reinterpret_cast<const uint4*>(x + pix * C + c0);
        const float* sc = scale + n * C + c0;
        const float* sh = shift + n * C + c0;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh), h1 = *reinterpret_cast<const f32x4*>(sh + 4);
        float v[8] = {bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y), bflo(u.z), bfhi(u.z), bflo(u.w), bfhi(u.w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = v[j] * s0[j] + h0[j];
            v[4 + j] = v[4 + j] * s1[j] + h1[j];
        }
        if (do_silu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = silu(v[j]);
        }
        *reinterpret_cast<uint4*>(y + pix * C + c0) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    }
}

// ---- stages 2 + 3 in one launch (round 6: the ControlLDM step is bound by its launch count - 61 of these per step): every workgroup finalises the
// statistics of all N x G groups itself (a few hundred partials, the arithmetic of gn_any_finalize_kernel in the same order) and applies. The
// per-channel scale / shift are formed per element with the expressions of stage 2, so the result is that of the three-launch form bit for bit.
constexpr int GN_ANY_MAX_GROUPS = 2048;
__global__ __launch_bounds__(256) void gn_any_finalize_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, const float* __restrict__ part,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta, long HW, int C, int cpg,
                                                                    int chunks, int NG, float eps, int do_silu, long total_vec) {
    __shared__ float mean_s[GN_ANY_MAX_GROUPS], rstd_s[GN_ANY_MAX_GROUPS];
    for (int i = threadIdx.x; i < NG; i += 256) {
        const float* p = part + (long)i * chunks * 2;
        double s = 0.0, q = 0.0;
        for (int k = 0; k < chunks; ++k) { s += p[2 * k]; q += p[2 * k + 1]; }
        const double cnt = (double)HW * cpg, m = s / cnt;
        double var = q / cnt - m * m;
        if (var < 0.0) var = 0.0;
        mean_s[i] = (float)m;
        rstd_s[i] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int vpp = C >> 3, G = C / cpg;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (long)gridDim.x * 256) {
        const long pix = i / vpp;
        const int c0 = (int)(i - pix * vpp) * 8;
        const int n = (int)(pix / HW);
        const uint4 u = *reinterpret_cast<const uint4*>(x + pix * C + c0);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c0), g1 = *reinterpret_cast<const f32x4*>(gamma + c0 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c0), b1 = *reinterpret_cast<const f32x4*>(beta + c0 + 4);
        float v[8] = {bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y), bflo(u.z), bfhi(u.z), bflo(u.w), bfhi(u.w)};
        int g = c0 / cpg, rem = c0 - g * cpg;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float ga = j < 4 ? g0[j & 3] : g1[j & 3], be = j < 4 ? b0[j & 3] : b1[j & 3];
            const float sc = ga * rstd_s[n * G + g];
            const float sh = be - mean_s[n * G + g] * sc;
            v[j] = v[j] * sc + sh;
            if (++rem == cpg) { rem = 0; ++g; }
        }
        if (do_silu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = silu(v[j]);
        }
        *reinterpret_cast<uint4*>(y + pix * C + c0) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    }
}

// ---- GEGLU: out[r][c] = ag[r][c] * gelu_erf(ag[r][F + c]); ag: [rows][2F] bf16 (value half | gate half), out: [rows][F]
__global__ __launch_bounds__(256) void geglu_kernel(const bf16_t* __restrict__ ag, bf16_t* __restrict__ out, long rows, int F) {
    const int vpr = F >> 3;
    const long nv = rows * vpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        const long row = i / vpr;
        const int c = (int)(i - row * vpr) * 8;
        const uint4 a = *reinterpret_cast<const uint4*>(ag + row * 2 * F + c);
        const uint4 g = *reinterpret_cast<const uint4*>(ag + row * 2 * F + F + c);
        uint4 o;
        o.x = pack2bf(bflo(a.x) * gelu_erf(bflo(g.x)), bfhi(a.x) * gelu_erf(bfhi(g.x)));
        o.y = pack2bf(bflo(a.y) * gelu_erf(bflo(g.y)), bfhi(a.y) * gelu_erf(bfhi(g.y)));
        o.z = pack2bf(bflo(a.z) * gelu_erf(bflo(g.z)), bfhi(a.z) * gelu_erf(bfhi(g.z)));
        o.w = pack2bf(bflo(a.w) * gelu_erf(bflo(g.w)), bfhi(a.w) * gelu_erf(bfhi(g.w)));
        *reinterpret_cast<uint4*>(out + row * F + c) = o;
    }
}

// ---- cat((x, hint), dim=1) of NCHW fp32 latents (4 channels each; hint may be null) -> NHWC bf16 rows of 32 channels (zero padded)
__global__ __launch_bounds__(256) void cldm_in_kernel(const float* __restrict__ x, const float* __restrict__ hint, bf16_t* __restrict__ out, long HW, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW, p = i - n * HW;
        const float* xp = x + n * 4 * HW + p;
        float v[8] = {xp[0], xp[HW], xp[2 * HW], xp[3 * HW], 0.f, 0.f, 0.f, 0.f};
        if (hint) {
            const float* hp = hint + n * 4 * HW + p;
            v[4] = hp[0]; v[5] = hp[HW]; v[6] = hp[2 * HW]; v[7] = hp[3 * HW];
        }
        uint4* o = reinterpret_cast<uint4*>(out + i * 32);
        o[0] = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
        o[1] = o[2] = o[3] = make_uint4(0, 0, 0, 0);
    }
}
// ---- out[n][c][p] = zT[n][c][p] + v[n*HW + p][c]  (cldm.py:588; zT null: v alone = apply_model's eps; v: the fp32 NHWC rows of the UNet's last conv, 4 of v_cs columns)
__global__ __launch_bounds__(256) void cldm_out_kernel(const float* __restrict__ zT, const float* __restrict__ v, int v_cs, float* __restrict__ out, long HW,
                                                       long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW, p = i - n * HW;
        const float* vp = v + i * v_cs;
#pragma unroll
        for (int c = 0; c < 4; ++c) out[(n * 4 + c) * HW + p] = (zT ? zT[(n * 4 + c) * HW + p] : 0.f) + vp[c];
    }
}
// ---- rows of C bf16 channels from src (row stride src_cs) into dst (row stride dst_cs), optionally dst = src + add (add row stride add_cs)
__global__ __launch_bounds__(256) void copy_rows_kernel(const bf16_t* __restrict__ src, int src_cs, const bf16_t* __restrict__ add, int add_cs,
                                                        bf16_t* __restrict__ dst, int dst_cs, int C, long total_vec) {
    const int vpr = C >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (long)gridDim.x * 256) {
        const long row = i / vpr;
        const int c = (int)(i - row * vpr) * 8;
        uint4 u = *reinterpret_cast<const uint4*>(src + row * src_cs + c);
        if (add) {
            const uint4 a = *reinterpret_cast<const uint4*>(add + row * add_cs + c);
            u.x = pack2bf(bflo(u.x) + bflo(a.x), bfhi(u.x) + bfhi(a.x));
            u.y = pack2bf(bflo(u.y) + bflo(a.y), bfhi(u.y) + bfhi(a.y));
            u.z = pack2bf(bflo(u.z) + bflo(a.z), bfhi(u.z) + bfhi(a.z));
            u.w = pack2bf(bflo(u.w) + bflo(a.w), bfhi(u.w) + bfhi(a.w));
        }
        *reinterpret_cast<uint4*>(dst + row * dst_cs + c) = u;
    }
}

unsigned grid1d(long n) {
    long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace

int ir_gn_any_chunks(long HW) {
    long c = (HW + 1023) / 1024;
    return (int)(c < 1 ? 1 : (c > 64 ? 64 : c));
}
long ir_gn_any_ws_floats(int N, long HW, int C) { return (long)N * 32 * ir_gn_any_chunks(HW) * 2 + 2L * N * C; }

int ir_launch_groupnorm_any(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, float* ws, int N, long HW, int C, int G, float eps,
                            int do_silu, hipStream_t s) {
    if (N <= 0 || HW <= 0) return 0;
    if (G <= 0 || G > 64 || C % G || (C & 7) || ((C / G) & 1)) return -2;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15)) return -3;
    const int cpg = C / G, chunks = ir_gn_any_chunks(HW);
    const long ppc = (HW + chunks - 1) / chunks;
    float* part = ws;                               // [N][G][chunks][2]
    float* scale = ws + (long)N * G * chunks * 2;   // [N][C]
    float* shift = scale + (long)N * C;             // [N][C]
    hipLaunchKernelGGL(gn_any_partial_kernel, dim3(chunks, G, N), dim3(256), 0, s, x, part, HW, C, cpg, chunks, ppc);
    const long nv = (long)N * HW * (C / 8);
    static const bool three = getenv("IR_GN_ANY_3") != nullptr;   // experiment knob: finalise as a launch of its own
    if (!three && (long)N * G <= GN_ANY_MAX_GROUPS && !(reinterpret_cast<uintptr_t>(gamma) & 15) && !(reinterpret_cast<uintptr_t>(beta) & 15)) {
        hipLaunchKernelGGL(gn_any_finalize_apply_kernel, dim3(grid1d(nv)), dim3(256), 0, s, x, y, part, gamma, beta, HW, C, cpg, chunks, N * G, eps, do_silu, nv);
        return LAUNCH_OK();
    }
    hipLaunchKernelGGL(gn_any_finalize_kernel, dim3(N), dim3(256), 0, s, part, gamma, beta, scale, shift, HW, C, cpg, chunks, eps);
    hipLaunchKernelGGL(gn_any_apply_kernel, dim3(grid1d(nv)), dim3(256), 0, s, x, y, scale, shift, HW, C, do_silu, nv);
    return LAUNCH_OK();
}
int ir_launch_geglu(const bf16_t* ag, bf16_t* out, long rows, int F, hipStream_t s) {
    if (rows <= 0) return 0;
    if (F & 7) return -2;
    hipLaunchKernelGGL(geglu_kernel, dim3(grid1d(rows * (F / 8))), dim3(256), 0, s, ag, out, rows, F);
    return LAUNCH_OK();
}
int ir_launch_cldm_in(const float* x, const float* hint, bf16_t* out, int N, long HW, hipStream_t s) {
    const long total = (long)N * HW;
    hipLaunchKernelGGL(cldm_in_kernel, dim3(grid1d(total)), dim3(256), 0, s, x, hint, out, HW, total);
    return LAUNCH_OK();
}
int ir_launch_cldm_out(const float* zT, const float* v, int v_cs, float* out, int N, long HW, hipStream_t s) {
    const long total = (long)N * HW;
    hipLaunchKernelGGL(cldm_out_kernel, dim3(grid1d(total)), dim3(256), 0, s, zT, v, v_cs, out, HW, total);
    return LAUNCH_OK();
}
int ir_launch_copy_rows(const bf16_t* src, int src_cs, const bf16_t* add, int add_cs, bf16_t* dst, int dst_cs, long rows, int C, hipStream_t s) {
    if (rows <= 0) return 0;
    if ((C | src_cs | dst_cs | add_cs) & 7) return -2;
    const long nv = rows * (C / 8);
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid1d(nv)), dim3(256), 0, s, src, src_cs, add, add_cs, dst, dst_cs, C, nv);
    return LAUNCH_OK();
}

