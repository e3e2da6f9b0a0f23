// HBM-bound normalisation kernels (gfx950): GroupNorm(32)+SiLU for the VAE, LayerNorm (+affine or
// adaLN-single modulate) for SwinIR / PixArt-DiT. All statistics accumulate in fp32 (finalised in fp64).
//
// GroupNorm follows reference ldm/modules/diffusionmodules/model.py:43-49 (Normalize = GroupNorm(32,
// eps=1e-6, affine) followed by x*sigmoid(x)); it is split in three launches:
//   gn_partial  : per (image, pixel-chunk) per-channel sum / sum-of-squares        (1 read of x)
//   gn_finalize : per (image, group) mean / rstd in fp64 -> per (image, channel) scale & shift
//   gn_apply    : y = silu(x * scale + shift), 16-byte vectors                      (1 read, 1 write)
// LayerNorm follows swinir.py:210,216,256,288,769 (affine, eps 1e-5) and PixArtMS.py:58,64,74-77 +
// PixArt_blocks.py:24-25 (no affine, eps 1e-6, then x*(1+scale)+shift).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

// ---------------------------------------------------------------- GroupNorm
// x: [N][HW][C] bf16 (C multiple of 8, C <= 512). grid = (chunks, N), block = 256.
// part: [N][chunks][2][C] fp32.
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ part,
                                                         int HW, int C, int chunks) {
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int vpp = C >> 3;             // 16-byte vectors per pixel
    const int ppi = 256 / vpp;          // pixels per block iteration (C=128:16, 256:8, 512:4)
    const int cv = tid % vpp, pl = tid / vpp;
    const long per = (HW + chunks - 1) / chunks;
    const long p0 = (long)chunk * per;
    const long p1 = p0 + per < HW ? p0 + per : HW;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    if (pl < ppi) {
        const bf16_t* base = x + (long)n * HW * C + cv * 8;
        auto acc8 = [&](const uint4& v) {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = bflo(w[e]), b = bfhi(w[e]);
                s[2 * e] += a; q[2 * e] += a * a;
                s[2 * e + 1] += b; q[2 * e + 1] += b * b;
            }
        };
        long pix = p0 + pl;
        for (; pix + 3L * ppi < p1; pix += 4L * ppi) {  // 4 independent 16-byte loads in flight per lane
            uint4 v0 = *reinterpret_cast<const uint4*>(base + pix * C);
            uint4 v1 = *reinterpret_cast<const uint4*>(base + (pix + ppi) * C);
            uint4 v2 = *reinterpret_cast<const uint4*>(base + (pix + 2L * ppi) * C);
            uint4 v3 = *reinterpret_cast<const uint4*>(base + (pix + 3L * ppi) * C);
            acc8(v0); acc8(v1); acc8(v2); acc8(v3);
        }
        for (; pix < p1; pix += ppi) acc8(*reinterpret_cast<const uint4*>(base + pix * C));
    }
    // reduce over the ppi pixel lanes that share a channel vector, in a FIXED order (no atomics: results must be bit-identical
    // run to run): every lane parks its 8+8 partials in LDS, then one thread per channel adds the ppi partials in sequence
    __shared__ float s_red[2][2048];  // [sum|sumsq][pl * C + c], ppi * C == 2048
    if (pl < ppi) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s_red[0][pl * C + cv * 8 + e] = s[e];
            s_red[1][pl * C + cv * 8 + e] = q[e];
        }
    }
    __syncthreads();
    float* out = part + ((long)n * chunks + chunk) * 2 * C;
    for (int c = tid; c < C; c += 256) {
        float a = 0.f, b = 0.f;
        for (int i = 0; i < ppi; ++i) { a += s_red[0][i * C + c]; b += s_red[1][i * C + c]; }
        out[c] = a;
        out[C + c] = b;
    }
}

// grid = N, block = 256. scale/shift: [N][C] fp32. gamma/beta: [C] fp32.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ scale,
                                                          float* __restrict__ shift, int HW, int C, int G, int chunks,
                                                          float eps) {
    __shared__ double s_mean[64], s_rstd[64];
    __shared__ double s_part[2][256];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int cpg = C / G;
    const int spg = 256 / G;                 // slices (threads) per group: 8 for G = 32
    const int g = tid / spg, sl = tid % spg;
    double s = 0.0, q = 0.0;
    if (g < G) {
        for (int ch = sl; ch < chunks; ch += spg) {
            const float* pp = part + ((long)n * chunks + ch) * 2 * C;
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
                s += (double)pp[c];
                q += (double)pp[C + c];
            }
        }
    }
    s_part[0][tid] = s;
    s_part[1][tid] = q;
    __syncthreads();
    if (tid < G) {
        s = 0.0; q = 0.0;
        for (int i = 0; i < spg; ++i) { s += s_part[0][tid * spg + i]; q += s_part[1][tid * spg + i]; }
        const double cnt = (double)HW * cpg;
        const double mean = s / cnt;
        double var = q / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[tid] = mean;
        s_rstd[tid] = 1.0 / sqrt(var + (double)eps);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int g = c / cpg;
        const double a = s_rstd[g] * (double)gamma[c];
        scale[(long)n * C + c] = (float)a;
        shift[(long)n * C + c] = (float)((double)beta[c] - s_mean[g] * a);
    }
}

// Finalise from per-GROUP partials written by the producing conv's epilogue (igemm.hip): part [N][chunks][2][G]. Same fixed-order
// fp64 reduction as above (thread (g, slice) sums every spg-th chunk, then the slices are added in order). grid = N, block = 256.
__global__ __launch_bounds__(256) void gn_finalize_groups_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ scale,
                                                                 float* __restrict__ shift, int HW, int C, int G, int chunks,
                                                                 float eps) {
    __shared__ double s_mean[64], s_rstd[64];
    __shared__ double s_part[2][256];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int cpg = C / G;
    const int spg = 256 / G;
    const int g = tid / spg, sl = tid % spg;
    double s = 0.0, q = 0.0;
    if (g < G) {
        for (int ch = sl; ch < chunks; ch += spg) {
            const float* pp = part + ((long)n * chunks + ch) * 2 * G;
            s += (double)pp[g];
            q += (double)pp[G + g];
        }
    }
    s_part[0][tid] = s;
    s_part[1][tid] = q;
    __syncthreads();
    if (tid < G) {
        s = 0.0; q = 0.0;
        for (int i = 0; i < spg; ++i) { s += s_part[0][tid * spg + i]; q += s_part[1][tid * spg + i]; }
        const double cnt = (double)HW * cpg;
        const double mean = s / cnt;
        double var = q / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[tid] = mean;
        s_rstd[tid] = 1.0 / sqrt(var + (double)eps);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int gg = c / cpg;
        const double a = s_rstd[gg] * (double)gamma[c];
        scale[(long)n * C + c] = (float)a;
        shift[(long)n * C + c] = (float)((double)beta[c] - s_mean[gg] * a);
    }
}

// y = act(x*scale + shift); x,y: [N][HW][C] bf16. grid = (blocks, N): blockIdx.y is the image, the blocks of an image stride over
// its 16-byte vectors. The stride (blocks * 256) is a multiple of the vectors per pixel, so a thread always meets the same 8
// channels: its scale / shift live in registers and the loop has no index arithmetic beyond one add (the first version divided
// two 64-bit indices per vector and re-read scale / shift for every vector: 3.6 TB/s).
template <bool FP8>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       long HW, int C, long nvec_img, int do_silu, float out_mul, long span) {
    const int vpp = C >> 3;
    const int n = blockIdx.y;
    const int cv = threadIdx.x % vpp;  // 256 % vpp == 0 and span % 256 == 0 (launcher): a thread always meets the same 8 channels
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = scale[(long)n * C + cv * 8 + e];
        sh[e] = shift[(long)n * C + cv * 8 + e];
    }
    x += (long)n * nvec_img * 8;
    y += (long)n * nvec_img * (FP8 ? 4 : 8);  // FP8: 8 output bytes per vector
    auto one = [&](long i, const uint4& v) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[4];
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = bflo(w[e]) * sc[2 * e] + sh[2 * e];
            float b = bfhi(w[e]) * sc[2 * e + 1] + sh[2 * e + 1];
            if (do_silu) { a = silu(a); b = silu(b); }
            o[e] = pack2bf_valu(a, b);   // a, b come out of VALU arithmetic (one v_cvt_pk instead of four instructions)
            f[2 * e] = a; f[2 * e + 1] = b;
        }
        if constexpr (FP8) {  // OCP e4m3 (max 448), round to nearest even; the consumer conv's epilogue divides out_mul out again
            int lo = 0, hi = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(f[e] * out_mul, -448.f), 448.f);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
            *reinterpret_cast<uint2*>(y + i * 4) = make_uint2((uint32_t)lo, (uint32_t)hi);
        } else {
            *reinterpret_cast<uint4*>(y + i * 8) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    };
    // Workgroup b owns the CONTIGUOUS span [b * span, (b + 1) * span) of the image's vectors (span a multiple of 1024, so of the vectors per
    // pixel too): measured on 1-2 GB tensors (tools/copy_rate.hip, profiles/r04_copy_rate.txt) a read + write pass in this order moves
    // 5.2-5.5 TB/s against 4.5-4.7 for the grid-stride order (every workgroup touching four windows 16 MB apart), arithmetic or not.
    constexpr long stride = 256;
    long i = (long)blockIdx.x * span + threadIdx.x;
    const long end = min(nvec_img, (long)(blockIdx.x + 1) * span);
    for (; i + 3 * stride < end; i += 4 * stride) {  // 4 independent 16-byte loads in flight per lane
        uint4 v0 = *reinterpret_cast<const uint4*>(x + i * 8);
        uint4 v1 = *reinterpret_cast<const uint4*>(x + (i + stride) * 8);
        uint4 v2 = *reinterpret_cast<const uint4*>(x + (i + 2 * stride) * 8);
        uint4 v3 = *reinterpret_cast<const uint4*>(x + (i + 3 * stride) * 8);
        one(i, v0); one(i + stride, v1); one(i + 2 * stride, v2); one(i + 3 * stride, v3);
    }
    for (; i < end; i += stride) one(i, *reinterpret_cast<const uint4*>(x + i * 8));
}

static void launch_gn_apply(const bf16_t* x, bf16_t* y, const float* scale, const float* shift, int N, long HW, int C, int do_silu,
                            hipStream_t s, int out_fp8, float out_mul) {
    const long nvec_img = HW * C / 8;
    long blocks = (nvec_img + 1023) / 1024;
    const long cap = (256L * 16 + N - 1) / N;  // about 16 blocks per CU over the whole launch
    if (blocks > cap) blocks = cap;
    const long span = ((nvec_img + blocks - 1) / blocks + 1023) / 1024 * 1024;
    blocks = (nvec_img + span - 1) / span;
    if (out_fp8) hipLaunchKernelGGL(gn_apply_kernel<true>, dim3((unsigned)blocks, N), dim3(256), 0, s, x, y, scale, shift, HW, C, nvec_img, do_silu, out_mul, span);
    else hipLaunchKernelGGL(gn_apply_kernel<false>, dim3((unsigned)blocks, N), dim3(256), 0, s, x, y, scale, shift, HW, C, nvec_img, do_silu, 1.f, span);
}

int ir_launch_groupnorm(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, float* ws, int N, long HW, int C,
                        int G, float eps, int do_silu, hipStream_t s, int out_fp8, float out_mul) {
    if (C % 8 || C > 512 || G > 64 || C % G || 256 % (C / 8) || 256 % G) return -2;
    if (HW >= (1L << 31)) return -3;
    int chunks = ir_gn_chunks(HW);
    float* part = ws;                                  // [N][chunks][2][C]
    float* scale = ws + (long)N * chunks * 2 * C;      // [N][C]
    float* shift = scale + (long)N * C;                // [N][C]
    hipLaunchKernelGGL(gn_partial_kernel, dim3(chunks, N), dim3(256), 0, s, x, part, (int)HW, C, chunks);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(N), dim3(256), 0, s, part, gamma, beta, scale, shift, (int)HW, C, G, chunks, eps);
    launch_gn_apply(x, y, scale, shift, N, HW, C, do_silu, s, out_fp8, out_mul);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// First reduction stage for many tiles: block (r, n) adds the per-group partials of a contiguous range of chunks in a fixed order
// (4 interleaved slices per value, then the slices in order) -> part2 [N][R][2][G]. grid = (R, N), block = 256, 2*G <= 64.
__global__ __launch_bounds__(256) void gn_reduce_groups_kernel(const float* __restrict__ part, float* __restrict__ part2, int G, int chunks,
                                                               int R) {
    __shared__ float s_red[4][64];
    const int r = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int idx = tid & 63, sl = tid >> 6;
    const int per = (chunks + R - 1) / R;
    const int c0 = r * per, c1 = min(c0 + per, chunks);
    float a = 0.f;
    if (idx < 2 * G)
        for (int ch = c0 + sl; ch < c1; ch += 4) a += part[((long)n * chunks + ch) * 2 * G + idx];
    s_red[sl][idx] = a;
    __syncthreads();
    if (tid < 2 * G) part2[((long)n * R + r) * 2 * G + tid] = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
}

// One workgroup per (group, image): thread t adds the partials of chunks t, t + 256, ... in fp64, then a fixed-order tree over the 256
// threads (bit-identical run to run), then the group's channels get their scale / shift. Replaces the single-block finalise (12-14 us of
// dependent loads per call) and, for images with more than 512 tiles, the extra reduction launch in front of it. grid = (G, N).
__global__ __launch_bounds__(256) void gn_finalize_group_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ scale,
                                                                float* __restrict__ shift, long HW, int C, int G, int chunks, float eps) {
    __shared__ double s_s[256], s_q[256];
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / G;
    double s = 0.0, q = 0.0;
    const float* pp = part + (long)n * chunks * 2 * G + g;
    for (int ch = tid; ch < chunks; ch += 256) {
        s += (double)pp[(long)ch * 2 * G];
        q += (double)pp[(long)ch * 2 * G + G];
    }
    s_s[tid] = s;
    s_q[tid] = q;
    __syncthreads();
#pragma unroll
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) { s_s[tid] += s_s[tid + st]; s_q[tid] += s_q[tid + st]; }
        __syncthreads();
    }
    if (tid < cpg) {
        const double cnt = (double)HW * cpg;
        const double mean = s_s[0] / cnt;
        double var = s_q[0] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const int c = g * cpg + tid;
        const double a = (1.0 / sqrt(var + (double)eps)) * (double)gamma[c];
        scale[(long)n * C + c] = (float)a;
        shift[(long)n * C + c] = (float)((double)beta[c] - mean * a);
    }
}

int ir_launch_groupnorm_fused(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, const float* part, float* ws, int N,
                              long HW, int C, int G, int chunks, float eps, int do_silu, hipStream_t s, int out_fp8, float out_mul) {
    if (C % 8 || C > 512 || G > 64 || C % G || 256 % (C / 8) || 256 % G || chunks <= 0) return -2;
    if (HW >= (1L << 31)) return -3;
    float* scale = ws;                 // [N][C]
    float* shift = ws + (long)N * C;   // [N][C]
    static const bool old_finalize = getenv("IR_GN_FINALIZE_V1") != nullptr;   // experiment knob: round 2's single-block finalise
    if (old_finalize) {
        if (chunks > 512) {                // two-stage: a single finalise block per image would crawl through megabytes of partials
            if (2 * G > 64) return -2;
            const int R = 256;
            float* part2 = shift + (long)N * C;  // [N][R][2][G]: fits the stand-alone path's partial area of ws (R*2*G <= chunks*2*C)
            hipLaunchKernelGGL(gn_reduce_groups_kernel, dim3(R, N), dim3(256), 0, s, part, part2, G, chunks, R);
            part = part2;
            chunks = R;
        }
        hipLaunchKernelGGL(gn_finalize_groups_kernel, dim3(N), dim3(256), 0, s, part, gamma, beta, scale, shift, (int)HW, C, G, chunks, eps);
    } else {
        hipLaunchKernelGGL(gn_finalize_group_kernel, dim3(G, N), dim3(256), 0, s, part, gamma, beta, scale, shift, HW, C, G, chunks, eps);
    }
    if (y) launch_gn_apply(x, y, scale, shift, N, HW, C, do_silu, s, out_fp8, out_mul);   // y == nullptr: finalise only (scale = ws, shift = ws + N * C)
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------- LayerNorm
// x: [rows][ldx] fp32 (first C entries normalised), y: [rows][ldy] bf16, y[c] = xn*a[c] + b[c] for c < C,
// 0 for C <= c < ldy. a/b: fp32 [C] (+ batch*ab_stride). One wave per row; VPL = values per lane.
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, float* __restrict__ yf,
                                                        const float* __restrict__ a, const float* __restrict__ b, long rows,
                                                        int C, int ldx, int ldy, float eps, long rows_per_batch, int ab_stride) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ldx;
    float v[VPL];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        v[i] = c < C ? xr[c] : 0.f;
        sum += v[i];
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        const float d = c < C ? v[i] - mean : 0.f;
        sq += d * d;
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
    const long batch = row / rows_per_batch;
    const float* ap = a ? a + batch * ab_stride : nullptr;
    const float* bp = b ? b + batch * ab_stride : nullptr;
    bf16_t* yr = y ? y + row * ldy : nullptr;
    float* yfr = yf ? yf + row * ldy : nullptr;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        if (c < ldy) {
            float o = 0.f;
            if (c < C) {
                o = (v[i] - mean) * rstd;
                if (ap) o *= ap[c];
                if (bp) o += bp[c];
            }
            if (yr) yr[c] = f2bf(o);
            if (yfr) yfr[c] = o;
        }
    }
}

// Vectorised form (C, ldx, ldy multiples of 4, 16-byte aligned rows): a lane owns float4 columns j = i*64 + lane, so the row is read
// with 16-byte loads, a / b with 16-byte loads, and written as 8-byte bf16 (16-byte fp32) vectors. VPL4 = float4 per lane.
template <int VPL4>
__global__ __launch_bounds__(256) void layernorm_v4_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, float* __restrict__ yf,
                                                           const float* __restrict__ a, const float* __restrict__ b, long rows,
                                                           int C, int ldx, int ldy, float eps, long rows_per_batch, int ab_stride) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ldx;
    const int C4 = C >> 2, L4 = ldy >> 2;
    f32x4 v[VPL4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL4; ++i) {
        const int j = i * 64 + lane;
        v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * min(j, C4 - 1));  // unconditional load on a valid address
        if (j >= C4) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL4; ++i) {
        const int j = i * 64 + lane;
        if (j < C4) {
            const f32x4 d = v[i] - mean;
            sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
    const long batch = row / rows_per_batch;
    const float* ap = a ? a + batch * ab_stride : nullptr;
    const float* bp = b ? b + batch * ab_stride : nullptr;
#pragma unroll
    for (int i = 0; i < VPL4; ++i) {
        const int j = i * 64 + lane;
        if (j < L4) {
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < C4) {
                o = (v[i] - mean) * rstd;
                if (ap) o *= *reinterpret_cast<const f32x4*>(ap + 4 * j);
                if (bp) o += *reinterpret_cast<const f32x4*>(bp + 4 * j);
            }
            if (y) *reinterpret_cast<uint2*>(y + row * ldy + 4 * j) = make_uint2(pack2bf_valu(o[0], o[1]), pack2bf_valu(o[2], o[3]));
            if (yf) *reinterpret_cast<f32x4*>(yf + row * ldy + 4 * j) = o;
        }
    }
}

// Short rows (SwinIR: C = 180 in rows of 192): 16 lanes per row, 4 rows per wave, three float4 per lane; the row sums are four DPP
// steps inside the 16-lane row (quad permutes, half-row mirror, row mirror) instead of six LDS-routed shuffles over a wave that is a
// quarter empty (layernorm_v4_kernel<1>: 2.9 TB/s).
IR_DEVINL float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));   // row_mirror
    return v;
}
__global__ __launch_bounds__(256) void layernorm_r16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, float* __restrict__ yf,
                                                            const float* __restrict__ a, const float* __restrict__ b, long rows, int C, int ldx,
                                                            int ldy, float eps, long rows_per_batch, int ab_stride) {
    const int l16 = threadIdx.x & 15;
    const long row_raw = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = row_raw < rows;
    const long row = live ? row_raw : rows - 1;   // spare 16-lane rows mirror the last row (DPP needs every lane) and store nothing
    const float* xr = x + row * ldx;
    const int C4 = C >> 2, L4 = ldy >> 2;
    f32x4 v[3];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = i * 16 + l16;
        v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * min(j, C4 - 1));
        if (j >= C4) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = row16_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = i * 16 + l16;
        if (j < C4) {
            const f32x4 d = v[i] - mean;
            sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float rstd = rsqrtf(row16_sum(sq) / (float)C + eps);
    const long batch = row / rows_per_batch;
    const float* ap = a ? a + batch * ab_stride : nullptr;
    const float* bp = b ? b + batch * ab_stride : nullptr;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = i * 16 + l16;
        if (j < L4 && live) {
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < C4) {
                o = (v[i] - mean) * rstd;
                if (ap) o *= *reinterpret_cast<const f32x4*>(ap + 4 * j);
                if (bp) o += *reinterpret_cast<const f32x4*>(bp + 4 * j);
            }
            if (y) *reinterpret_cast<uint2*>(y + row * ldy + 4 * j) = make_uint2(pack2bf_valu(o[0], o[1]), pack2bf_valu(o[2], o[3]));
            if (yf) *reinterpret_cast<f32x4*>(yf + row * ldy + 4 * j) = o;
        }
    }
}

int ir_launch_layernorm(const float* x, bf16_t* y, float* yf, const float* a, const float* b, long rows, int C, int ldx, int ldy,
                        float eps, long rows_per_batch, int ab_stride, hipStream_t s) {
    if (rows <= 0) return 0;
    if (C > ldx || C > ldy || ldy > 1280) return -2;
    if (rows_per_batch <= 0) return -3;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    const bool v4 = !((C | ldx | ldy | ab_stride) & 3) && !(reinterpret_cast<uintptr_t>(x) & 15) && !(reinterpret_cast<uintptr_t>(y) & 7) &&
                    !(reinterpret_cast<uintptr_t>(yf) & 15) && !(reinterpret_cast<uintptr_t>(a) & 15) && !(reinterpret_cast<uintptr_t>(b) & 15);
    if (v4 && ldy <= 192 && rows >= 4096) {
        hipLaunchKernelGGL(layernorm_r16_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, x, y, yf, a, b, rows, C, ldx, ldy, eps, rows_per_batch,
                           ab_stride);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    if (v4 && ldy <= 256) {
        hipLaunchKernelGGL((layernorm_v4_kernel<1>), dim3(grid), dim3(256), 0, s, x, y, yf, a, b, rows, C, ldx, ldy, eps, rows_per_batch, ab_stride);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    if (v4 && ldy <= 1280) {
        hipLaunchKernelGGL((layernorm_v4_kernel<5>), dim3(grid), dim3(256), 0, s, x, y, yf, a, b, rows, C, ldx, ldy, eps, rows_per_batch, ab_stride);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    if (ldy <= 192)
        hipLaunchKernelGGL((layernorm_kernel<3>), dim3(grid), dim3(256), 0, s, x, y, yf, a, b, rows, C, ldx, ldy, eps, rows_per_batch, ab_stride);
    else if (ldy <= 1152)
        hipLaunchKernelGGL((layernorm_kernel<18>), dim3(grid), dim3(256), 0, s, x, y, yf, a, b, rows, C, ldx, ldy, eps, rows_per_batch, ab_stride);
    else
        return -2;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------- fp32 GEMV (conditioning path, run once per timestep)
// out[n] = act(b[n] + sum_k w[n][k] * x[k]); one wave per output row. Used for the timestep MLP and adaLN-single
// linear (PixArt_blocks.py:336-358, PixArtMS.py:134-137) where bf16 rounding of a [1,K] operand would bias every token.
__global__ __launch_bounds__(256) void gemv_f32_kernel(const float* __restrict__ w, const float* __restrict__ x,
                                                       const float* b, float* out, int N, int K, int act) {   // b may alias out (out[row] = w[row] . x + out[row]: the size embedders of the DiT)
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* wr = w + (long)row * K;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc += wr[k] * x[k];
    acc = wave_sum(acc);
    if (lane == 0) {
        float v = acc + (b ? b[row] : 0.f);
        if (act == IR_ACT_SILU) v = silu(v);
        out[row] = v;
    }
}
int ir_launch_gemv_f32(const float* w, const float* x, const float* b, float* out, int N, int K, int act, hipStream_t s) {
    hipLaunchKernelGGL(gemv_f32_kernel, dim3((N + 3) / 4), dim3(256), 0, s, w, x, b, out, N, K, act);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Token preparation of the DiT self-attention's optional branches (AttentionKVCompress, PixArt_blocks.py:60-158; round 6): one kernel for
//   * KV compression (downsample_2d, :97-121): output token (oy, ox) of the (gh / r) x (gw / r) grid = depthwise r x r / stride r convolution
//     over the token grid (`sr`; 'uniform' / 'ave' sampling = the same with a weight of 1 on the window's first token) + optional LayerNorm (`norm`);
//   * qk_norm (:136-137): r = 1, no weights, LayerNorm over all C channels of the token, in place on the q / k columns of the qkv rows.
// in: [B][gh * gw] bf16 rows (row stride in_rs, batch stride in_bs); out: [B][(gh / r) * (gw / r)] rows (out_rs, out_bs); w: [C][r * r] fp32 or null
// (weight 1 on tap 0), bias [C] or null, gamma / beta [C] or null (no LayerNorm); eps 1e-5 (nn.LayerNorm default). One 256-thread workgroup per
// output token, a thread's channels (at most 8: C <= 2048) in registers between the two passes, so in == out is allowed for r == 1.
__global__ __launch_bounds__(256) void dit_token_prep_kernel(const bf16_t* __restrict__ in, bf16_t* out, const float* __restrict__ w, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, int gh, int gw, int r, int C,
                                                             int in_rs, long in_bs, int out_rs, long out_bs, float eps) {
    __shared__ float red[2][4];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int ow = gw / r, j = blockIdx.x, oy = j / ow, ox = j - oy * ow;
    const bf16_t* src = in + (long)b * in_bs;
    float v[8];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + 256 * i;
        v[i] = 0.f;
        if (c < C) {
            float a = bias ? bias[c] : 0.f;
            for (int dy = 0; dy < r; ++dy)
                for (int dx = 0; dx < r; ++dx) {
                    const float wt = w ? w[(long)c * r * r + dy * r + dx] : ((dy | dx) == 0 ? 1.f : 0.f);
                    a += wt * bf2f(src[(long)((oy * r + dy) * gw + ox * r + dx) * in_rs + c]);
                }
            v[i] = a;
            s += a;
        }
    }
    if (gamma) {   // two-pass LayerNorm over the C channels of this token (mean first, then the centred second moment: as F.layer_norm)
        s = wave_sum(s);
        if ((tid & 63) == 0) red[0][tid >> 6] = s;
        __syncthreads();
        const float mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)C;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + 256 * i;
            if (c < C) { const float d = v[i] - mean; q += d * d; }
        }
        q = wave_sum(q);
        if ((tid & 63) == 0) red[1][tid >> 6] = q;
        __syncthreads();
        const float rstd = rsqrtf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)C + eps);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + 256 * i;
            if (c < C) v[i] = (v[i] - mean) * rstd * gamma[c] + beta[c];
        }
    }
    bf16_t* dst = out + (long)b * out_bs + (long)j * out_rs;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + 256 * i;
        if (c < C) dst[c] = f2bf(v[i]);
    }
}
int ir_launch_dit_token_prep(const bf16_t* in, bf16_t* out, const float* w, const float* bias, const float* gamma, const float* beta, int B, int gh, int gw, int r,
                             int C, int in_rs, long in_bs, int out_rs, long out_bs, hipStream_t s) {
    if (B <= 0 || gh <= 0 || gw <= 0 || r < 1 || gh % r || gw % r || C <= 0 || C > 2048 || (gamma == nullptr) != (beta == nullptr)) return -2;
    if (in == out && r != 1) return -3;
    hipLaunchKernelGGL(dit_token_prep_kernel, dim3((gh / r) * (gw / r), B), dim3(256), 0, s, in, out, w, bias, gamma, beta, gh, gw, r, C, in_rs, in_bs, out_rs, out_bs, 1e-5f);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
