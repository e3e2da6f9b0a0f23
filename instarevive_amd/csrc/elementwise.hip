// Layout, cast and glue kernels (all HBM-bound, gfx950). Each one cites the reference lines it restates.
#include "common.h"
#include "kernels.h"

#define GRID1D(n) dim3((unsigned)(((n) + 255) / 256 > 65535L * 16 ? 65535L * 16 : ((n) + 255) / 256))
#define FOR_GRID(i, n) for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (n); i += (long)gridDim.x * 256)
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : -1)

// uint8 HWC -> fp32 NCHW in [0,1]  (test_scripts/inference.py:92-93: np.stack(imgs)/255.0 in f64, cast to f32, n h w c -> n c h w)
__global__ void u8_to_nchw_kernel(const uint8_t* in, float* out, long HW, long total) {
    FOR_GRID(i, total) {
        long n = i / (3 * HW), rem = i - n * 3 * HW;
        long c = rem / HW, pix = rem - c * HW;
        out[i] = (float)((double)in[(n * HW + pix) * 3 + c] / 255.0);
    }
}
int ir_launch_u8_to_nchw(const uint8_t* in, float* out, int N, int H, int W, hipStream_t s) {
    long HW = (long)H * W, total = 3 * HW * N;
    hipLaunchKernelGGL(u8_to_nchw_kernel, GRID1D(total), dim3(256), 0, s, in, out, HW, total);
    return LAUNCH_OK();
}

// SwinIR head: (x - mean) * img_range then PixelUnshuffle(8) (swinir.py:871-872,707-708), emitted as NHWC bf16 with
// channel k = c*64 + i*8 + j  <-  x[n][c][8y+i][8x+j]. One thread per (pixel, c, i): 8 floats in, 16 bytes out.
__global__ void swin_prep_kernel(const float* x, bf16_t* out, int H, int W, float m0, float m1, float m2, float range, long total) {
    const int h8 = H >> 3, w8 = W >> 3;
    FOR_GRID(t, total) {
        int ci = (int)(t % 24);
        long pix = t / 24;
        int xx = (int)(pix % w8);
        long r2 = pix / w8;
        int yy = (int)(r2 % h8);
        long n = r2 / h8;
        int c = ci >> 3, i = ci & 7;
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
        const float* src = x + ((n * 3 + c) * H + (yy * 8 + i)) * (long)W + xx * 8;
        f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
        uint4 o = make_uint4(pack2bf((a[0] - mean) * range, (a[1] - mean) * range), pack2bf((a[2] - mean) * range, (a[3] - mean) * range),
                             pack2bf((b[0] - mean) * range, (b[1] - mean) * range), pack2bf((b[2] - mean) * range, (b[3] - mean) * range));
        *reinterpret_cast<uint4*>(out + pix * 192 + ci * 8) = o;
    }
}
int ir_launch_swin_prep(const float* x, bf16_t* out, int N, int H, int W, const float* mean3, float img_range, hipStream_t s) {
    if ((H & 7) || (W & 7)) return -2;
    long total = (long)N * (H >> 3) * (W >> 3) * 24;
    hipLaunchKernelGGL(swin_prep_kernel, GRID1D(total), dim3(256), 0, s, x, out, H, W, mean3[0], mean3[1], mean3[2], img_range, total);
    return LAUNCH_OK();
}

// fp32 NHWC (pixel stride in_cs, first C channels) -> fp32 NCHW, v*scale+shift, optional clamp to [0,1]
__global__ void nhwc_to_nchw_kernel(const float* in, int in_cs, float* out, int C, long HW, float scale, float shift, int clamp01,
                                    long total) {
    FOR_GRID(i, total) {
        long n = i / (C * HW), rem = i - n * C * HW;
        long c = rem / HW, pix = rem - c * HW;
        float v = in[(n * HW + pix) * in_cs + c] * scale + shift;
        if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        out[i] = v;
    }
}
int ir_launch_nhwc_to_nchw(const float* in, int in_cs, float* out, int N, int C, long HW, float scale, float shift, int clamp01,
                           hipStream_t s) {
    long total = (long)N * C * HW;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, GRID1D(total), dim3(256), 0, s, in, in_cs, out, C, HW, scale, shift, clamp01, total);
    return LAUNCH_OK();
}

// fp32 NCHW -> bf16 NHWC zero-padded to Cpad channels, v*scale+shift (inference.py:104 control*2-1 feeds the VAE)
__global__ void nchw_to_nhwc_bf16_kernel(const float* in, bf16_t* out, int C, long HW, int Cpad, float scale, float shift, long total) {
    FOR_GRID(i, total) {  // one thread per (pixel, 8-channel chunk)
        const int chunks = Cpad >> 3;
        long pixg = i / chunks;
        int ch = (int)(i - pixg * chunks);
        long n = pixg / HW, pix = pixg - n * HW;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int c = ch * 8 + e;
            v[e] = c < C ? in[(n * C + c) * HW + pix] * scale + shift : 0.f;
        }
        *reinterpret_cast<uint4*>(out + pixg * Cpad + ch * 8) =
            make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    }
}
int ir_launch_nchw_to_nhwc_bf16(const float* in, bf16_t* out, int N, int C, long HW, int Cpad, float scale, float shift,
                                hipStream_t s) {
    if (Cpad & 7) return -2;
    long total = (long)N * HW * (Cpad >> 3);
    hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel, GRID1D(total), dim3(256), 0, s, in, out, C, HW, Cpad, scale, shift, total);
    return LAUNCH_OK();
}

// fp32 NHWC (stride in_cs, C real) -> bf16 NHWC padded to Cpad
__global__ void nhwc_f32_to_bf16pad_kernel(const float* in, int in_cs, bf16_t* out, int C, int Cpad, float scale, float shift,
                                           long total) {
    FOR_GRID(i, total) {
        const int chunks = Cpad >> 3;
        long pix = i / chunks;
        int ch = (int)(i - pix * chunks);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int c = ch * 8 + e;
            v[e] = c < C ? in[pix * in_cs + c] * scale + shift : 0.f;
        }
        *reinterpret_cast<uint4*>(out + pix * Cpad + ch * 8) =
            make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    }
}
int ir_launch_nhwc_f32_to_bf16pad(const float* in, int in_cs, bf16_t* out, long npix, int C, int Cpad, float scale, float shift,
                                  hipStream_t s) {
    if (Cpad & 7) return -2;
    long total = npix * (Cpad >> 3);
    hipLaunchKernelGGL(nhwc_f32_to_bf16pad_kernel, GRID1D(total), dim3(256), 0, s, in, in_cs, out, C, Cpad, scale, shift, total);
    return LAUNCH_OK();
}

// VAE encode tail: quant_conv (1x1, 8->8) and DiagonalGaussianDistribution.mode() == first 4 channels
// (ldm/models/autoencoder.py:82-86, ldm/modules/distributions/distributions.py:24-27,59-60) -> fp32 NCHW latents * scale.
__global__ void quant_mean_kernel(const float* h8, int h_cs, const float* wq, const float* bq, float* lat, long HW, float scale,
                                  long total) {
    FOR_GRID(i, total) {  // one thread per pixel
        long n = i / HW, pix = i - n * HW;
        float hv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) hv[k] = h8[i * h_cs + k];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = bq[c];
#pragma unroll
            for (int k = 0; k < 8; ++k) a += wq[c * 8 + k] * hv[k];
            lat[(n * 4 + c) * HW + pix] = a * scale;
        }
    }
}
int ir_launch_quant_mean(const float* h8, int h_cs, const float* wq, const float* bq, float* lat, int N, long HW, float scale,
                         hipStream_t s) {
    long total = (long)N * HW;
    hipLaunchKernelGGL(quant_mean_kernel, GRID1D(total), dim3(256), 0, s, h8, h_cs, wq, bq, lat, HW, scale, total);
    return LAUNCH_OK();
}

// VAE decode head: post_quant_conv (1x1, 4->4; autoencoder.py:88-90) on latents*in_scale -> bf16 NHWC padded to Cpad
__global__ void latent_prep_kernel(const float* lat, const float* w, const float* b, bf16_t* out, long HW, int Cpad, float in_scale,
                                   long total) {
    FOR_GRID(i, total) {
        long n = i / HW, pix = i - n * HW;
        float z[4], o[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) z[c] = lat[(n * 4 + c) * HW + pix] * in_scale;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = b[c] + w[c * 4] * z[0] + w[c * 4 + 1] * z[1] + w[c * 4 + 2] * z[2] + w[c * 4 + 3] * z[3];
#pragma unroll
        for (int c = 4; c < 8; ++c) o[c] = 0.f;
        bf16_t* op = out + i * Cpad;
        *reinterpret_cast<uint4*>(op) = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), 0u, 0u);
        for (int c = 8; c < Cpad; c += 8) *reinterpret_cast<uint4*>(op + c) = make_uint4(0, 0, 0, 0);
    }
}
int ir_launch_latent_prep(const float* lat, const float* w, const float* b, bf16_t* out, int N, long HW, int Cpad, float in_scale,
                          hipStream_t s) {
    if (Cpad & 7) return -2;
    long total = (long)N * HW;
    hipLaunchKernelGGL(latent_prep_kernel, GRID1D(total), dim3(256), 0, s, lat, w, b, out, HW, Cpad, in_scale, total);
    return LAUNCH_OK();
}

// DiT patch embedding input (PixArtMS.py:38-46: Conv2d(4,1152,k=2,s=2) == linear over k = c*4 + p*2 + q):
// latents fp32 NCHW [N][4][2h][2w] -> bf16 tokens [N*h*w][Cpad], entries >= 16 zero.
__global__ void patchify_kernel(const float* lat, bf16_t* out, int h, int w, int Cpad, long total) {
    FOR_GRID(i, total) {
        long n = i / ((long)h * w);
        int rem = (int)(i - n * (long)h * w);
        int hh = rem / w, ww = rem - hh * w;
        const int Hf = 2 * h, Wf = 2 * w;
        float v[16];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int pq = 0; pq < 4; ++pq)
                v[c * 4 + pq] = lat[((n * 4 + c) * Hf + 2 * hh + (pq >> 1)) * (long)Wf + 2 * ww + (pq & 1)];
        bf16_t* op = out + i * Cpad;
        *reinterpret_cast<uint4*>(op) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
        *reinterpret_cast<uint4*>(op + 8) =
            make_uint4(pack2bf(v[8], v[9]), pack2bf(v[10], v[11]), pack2bf(v[12], v[13]), pack2bf(v[14], v[15]));
        for (int c = 16; c < Cpad; c += 8) *reinterpret_cast<uint4*>(op + c) = make_uint4(0, 0, 0, 0);
    }
}
int ir_launch_patchify(const float* lat, bf16_t* out, int N, int h, int w, int Cpad, hipStream_t s) {
    if ((Cpad & 7) || Cpad < 16) return -2;
    long total = (long)N * h * w;
    hipLaunchKernelGGL(patchify_kernel, GRID1D(total), dim3(256), 0, s, lat, out, h, w, Cpad, total);
    return LAUNCH_OK();
}

// unpatchify (PixArtMS.py:236-248, einsum nhwpqc->nchpwq): tok fp32 [N*h*w][32], column (p*2+q)*8 + c -> out[N][8][2h][2w]
__global__ void unpatchify_kernel(const float* tok, float* out, int h, int w, long total) {
    FOR_GRID(i, total) {  // one thread per output element
        const int Hf = 2 * h, Wf = 2 * w;
        long n = i / (8L * Hf * Wf);
        long rem = i - n * 8L * Hf * Wf;
        int c = (int)(rem / ((long)Hf * Wf));
        int r2 = (int)(rem - (long)c * Hf * Wf);
        int y = r2 / Wf, x = r2 - y * Wf;
        long t = (n * h + (y >> 1)) * w + (x >> 1);
        out[i] = tok[t * 32 + ((y & 1) * 2 + (x & 1)) * 8 + c];
    }
}
int ir_launch_unpatchify(const float* tok, float* out, int N, int h, int w, hipStream_t s) {
    long total = (long)N * 8 * 4 * h * w;
    hipLaunchKernelGGL(unpatchify_kernel, GRID1D(total), dim3(256), 0, s, tok, out, h, w, total);
    return LAUNCH_OK();
}

// fused unpatchify + keep eps half (generate.py:84-85) + eps_to_mu (generate.py:44-51) (+ optional 1/scaling_factor)
__global__ void eps_to_x0_kernel(const float* tok, const float* lat_in, float* lat_out, int h, int w, float s0, float s1,
                                 float out_scale, long total) {
    FOR_GRID(i, total) {
        const int Hf = 2 * h, Wf = 2 * w;
        long n = i / (4L * Hf * Wf);
        long rem = i - n * 4L * Hf * Wf;
        int c = (int)(rem / ((long)Hf * Wf));
        int r2 = (int)(rem - (long)c * Hf * Wf);
        int y = r2 / Wf, x = r2 - y * Wf;
        long t = (n * h + (y >> 1)) * w + (x >> 1);
        float eps = tok[t * 32 + ((y & 1) * 2 + (x & 1)) * 8 + c];
        lat_out[i] = ((lat_in[i] - s1 * eps) / s0) * out_scale;
    }
}
int ir_launch_eps_to_x0(const float* tok, const float* lat_in, float* lat_out, int N, int h, int w, float s0, float s1,
                        float out_scale, hipStream_t s) {
    long total = (long)N * 4 * 4 * h * w;
    hipLaunchKernelGGL(eps_to_x0_kernel, GRID1D(total), dim3(256), 0, s, tok, lat_in, lat_out, h, w, s0, s1, out_scale, total);
    return LAUNCH_OK();
}

// fp32 NHWC (3 channels, stride in_cs) -> uint8 HWC: clamp(v*scale+shift,0,1)*255, truncating cast (inference.py:159-160)
__global__ void nhwc_to_u8_kernel(const float* in, int in_cs, uint8_t* out, float scale, float shift, long total) {
    FOR_GRID(i, total) {
        long pix = i / 3;
        int c = (int)(i - pix * 3);
        float v = fminf(fmaxf(in[pix * in_cs + c] * scale + shift, 0.f), 1.f) * 255.f;
        out[i] = (uint8_t)fminf(fmaxf(v, 0.f), 255.f);
    }
}
int ir_launch_nhwc_to_u8(const float* in, int in_cs, uint8_t* out, long npix, float scale, float shift, hipStream_t s) {
    long total = npix * 3;
    hipLaunchKernelGGL(nhwc_to_u8_kernel, GRID1D(total), dim3(256), 0, s, in, in_cs, out, scale, shift, total);
    return LAUNCH_OK();
}
__global__ void nchw_to_u8_kernel(const float* in, uint8_t* out, long HW, long total) {
    FOR_GRID(i, total) {
        long pixg = i / 3;
        int c = (int)(i - pixg * 3);
        long n = pixg / HW, pix = pixg - n * HW;
        float v = fminf(fmaxf(in[(n * 3 + c) * HW + pix], 0.f), 1.f) * 255.f;
        out[i] = (uint8_t)fminf(fmaxf(v, 0.f), 255.f);
    }
}
int ir_launch_nchw_to_u8(const float* in, uint8_t* out, int N, long HW, hipStream_t s) {
    long total = (long)N * HW * 3;
    hipLaunchKernelGGL(nchw_to_u8_kernel, GRID1D(total), dim3(256), 0, s, in, out, HW, total);
    return LAUNCH_OK();
}

// ---- tiled glue (inference.py:119-153): accumulate tiles, divide by the data-independent overlap count
__global__ void tile_add_kernel(float* dst, const float* src, int C, int H, int W, int th, int tw, int y0, int x0, long total) {
    FOR_GRID(i, total) {
        int x = (int)(i % tw);
        long r = i / tw;
        int y = (int)(r % th);
        long nc = r / th;
        dst[(nc * H + y0 + y) * W + x0 + x] += src[i];
    }
}
// Zero fill as a kernel: memset NODES of a recorded hipGraph did not re-run reliably on replay (tiled pipeline, ROCm 7.2), kernels do.
__global__ void zero_f32_kernel(float* p, long n) {
    FOR_GRID(i, n) p[i] = 0.f;
}
int ir_launch_zero_f32(float* p, long n, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(zero_f32_kernel, GRID1D(n), dim3(256), 0, s, p, n);
    return LAUNCH_OK();
}
__global__ void fill_u32_kernel(uint32_t* p, long n, uint32_t v) {
    FOR_GRID(i, n) p[i] = v;
}
int ir_launch_fill_u32(uint32_t* p, long n, uint32_t v, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(fill_u32_kernel, GRID1D(n), dim3(256), 0, s, p, n, v);
    return LAUNCH_OK();
}
__global__ void count_flag_kernel(const int* flag, int* counter) {
    if (threadIdx.x == 0 && *flag != 0) atomicAdd(counter, 1);
}
int ir_launch_count_flag(const int* flag, int* counter, hipStream_t s) {
    if (!flag || !counter) return -2;
    hipLaunchKernelGGL(count_flag_kernel, dim3(1), dim3(64), 0, s, flag, counter);
    return LAUNCH_OK();
}
// V^T [heads_total][DV][Tpad] of the DiT self-attention when the qkv projection's epilogue writes rows d < D itself (IGemmParams::vt_out): the
// parts it never writes - row D = ones over the real keys (the softmax denominator), rows above zero, and the columns t >= T of every row.
__global__ void vt_pad_init_kernel(bf16_t* vt, int D, int DV, int T, int Tpad, long total) {
    FOR_GRID(i, total) {
        const int t = (int)(i % Tpad);
        const int d = (int)((i / Tpad) % DV);
        if (d >= D || t >= T) vt[i] = (d == D && t < T) ? (bf16_t)0x3f80 : (bf16_t)0;
    }
}
int ir_launch_vt_pad_init(bf16_t* vt, int heads_total, int D, int DV, int T, int Tpad, hipStream_t s) {
    const long total = (long)heads_total * DV * Tpad;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(vt_pad_init_kernel, GRID1D(total), dim3(256), 0, s, vt, D, DV, T, Tpad, total);
    return LAUNCH_OK();
}
int ir_launch_tile_add(float* dst, const float* src, int N, int C, int H, int W, int th, int tw, int y0, int x0, hipStream_t s) {
    if (y0 < 0 || x0 < 0 || y0 + th > H || x0 + tw > W) return -2;
    long total = (long)N * C * th * tw;
    hipLaunchKernelGGL(tile_add_kernel, GRID1D(total), dim3(256), 0, s, dst, src, C, H, W, th, tw, y0, x0, total);
    return LAUNCH_OK();
}
IR_DEVINL int window_count(int pos, int size, int tile, int stride) {  // number of _sliding_windows starts covering pos
    int cnt = 0;
    for (int st = 0; st <= size - tile; st += stride) cnt += (pos >= st && pos < st + tile);
    if ((size - tile) % stride != 0) cnt += (pos >= size - tile);
    return cnt;
}
__global__ void tile_div_kernel(float* dst, int H, int W, int th, int tw, int sy, int sx, long total) {
    FOR_GRID(i, total) {
        int x = (int)(i % W);
        int y = (int)((i / W) % H);
        dst[i] /= (float)(window_count(y, H, th, sy) * window_count(x, W, tw, sx));
    }
}
int ir_launch_tile_div(float* dst, int N, int C, int H, int W, int th, int tw, int sy, int sx, hipStream_t s) {
    if (th > H || tw > W || sy <= 0 || sx <= 0) return -2;
    long total = (long)N * C * H * W;
    hipLaunchKernelGGL(tile_div_kernel, GRID1D(total), dim3(256), 0, s, dst, H, W, th, tw, sy, sx, total);
    return LAUNCH_OK();
}
__global__ void crop_nchw_kernel(const float* src, float* dst, int H, int W, int y0, int x0, int th, int tw, float scale, long total) {
    FOR_GRID(i, total) {
        int x = (int)(i % tw);
        long r = i / tw;
        int y = (int)(r % th);
        long nc = r / th;
        dst[i] = src[(nc * H + y0 + y) * W + x0 + x] * scale;
    }
}
int ir_launch_crop_nchw(const float* src, float* dst, int N, int C, int H, int W, int y0, int x0, int th, int tw, float scale,
                        hipStream_t s) {
    if (y0 < 0 || x0 < 0 || y0 + th > H || x0 + tw > W) return -2;
    long total = (long)N * C * th * tw;
    hipLaunchKernelGGL(crop_nchw_kernel, GRID1D(total), dim3(256), 0, s, src, dst, H, W, y0, x0, th, tw, scale, total);
    return LAUNCH_OK();
}

// ---- colour fix (utils/image/align_color.py:73-119)
// one a-trous level: low = blur3x3(img, dilation=radius, replicate pad); if (high) high += img - low
__global__ void wavelet_level_kernel(const float* img, float* low, float* high, int H, int W, int radius, long total) {
    FOR_GRID(i, total) {
        int x = (int)(i % W);
        long r = i / W;
        int y = (int)(r % H);
        const float* pl = img + (r / H) * (long)H * W;
        float acc = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            int yy = min(max(y + dy * radius, 0), H - 1);
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                int xx = min(max(x + dx * radius, 0), W - 1);
                const float wgt = (dy == 0 ? 0.5f : 0.25f) * (dx == 0 ? 0.5f : 0.25f);
                acc += pl[(long)yy * W + xx] * wgt;
            }
        }
        low[i] = acc;
        if (high) high[i] += img[i] - acc;
    }
}
__global__ void add2_kernel(const float* a, const float* b, float* o, long total) {
    FOR_GRID(i, total) o[i] = a[i] + b[i];
}
// out = high_freq(content) + low_freq(style); tmp: 3 planes of N*3*H*W floats
int ir_launch_wavelet_fix(const float* content, const float* style, float* out, float* tmp, int N, int H, int W, hipStream_t s) {
    const long total = (long)N * 3 * H * W;
    float *ping = tmp, *pong = tmp + total, *high = tmp + 2 * total;
    if (ir_launch_zero_f32(high, total, s)) return -1;
    const float* cur = content;
    for (int lvl = 0; lvl < 5; ++lvl) {
        float* dst = (lvl & 1) ? pong : ping;
        hipLaunchKernelGGL(wavelet_level_kernel, GRID1D(total), dim3(256), 0, s, cur, dst, high, H, W, 1 << lvl, total);
        cur = dst;
    }
    cur = style;
    for (int lvl = 0; lvl < 5; ++lvl) {
        float* dst = (lvl & 1) ? pong : ping;
        hipLaunchKernelGGL(wavelet_level_kernel, GRID1D(total), dim3(256), 0, s, cur, dst, (float*)nullptr, H, W, 1 << lvl, total);
        cur = dst;
    }
    hipLaunchKernelGGL(add2_kernel, GRID1D(total), dim3(256), 0, s, high, cur, out, total);
    return LAUNCH_OK();
}

// adaptive_instance_normalization (align_color.py:44-70): per (n,c) mean / unbiased var + 1e-5
__global__ __launch_bounds__(256) void plane_stats_kernel(const float* x, float* stats, long HW) {
    __shared__ double sh[2][4];
    const float* p = x + (long)blockIdx.x * HW;
    double s = 0.0, q = 0.0;
    for (long i = threadIdx.x; i < HW; i += 256) { double v = p[i]; s += v; q += v * v; }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        q = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        double mean = s / (double)HW;
        double var = (q - s * mean) / (double)(HW - 1);
        stats[blockIdx.x * 2] = (float)mean;
        stats[blockIdx.x * 2 + 1] = (float)sqrt(var + 1e-5);
    }
}
__global__ void adain_apply_kernel(const float* c, const float* cs, const float* ss, float* out, long HW, long total) {
    FOR_GRID(i, total) {
        long pl = i / HW;
        out[i] = (c[i] - cs[pl * 2]) / cs[pl * 2 + 1] * ss[pl * 2 + 1] + ss[pl * 2];
    }
}
int ir_launch_adain_fix(const float* content, const float* style, float* out, float* ws, int N, int H, int W, hipStream_t s) {
    const long HW = (long)H * W, total = (long)N * 3 * HW;
    if (HW < 2) return -2;
    float *cs = ws, *ss = ws + N * 3 * 2;
    hipLaunchKernelGGL(plane_stats_kernel, dim3(N * 3), dim3(256), 0, s, content, cs, HW);
    hipLaunchKernelGGL(plane_stats_kernel, dim3(N * 3), dim3(256), 0, s, style, ss, HW);
    hipLaunchKernelGGL(adain_apply_kernel, GRID1D(total), dim3(256), 0, s, content, cs, ss, out, HW, total);
    return LAUNCH_OK();
}

// ---- conditioning helpers (constant per timestep / prompt; run once, cached by the context)
__global__ void silu_f32_kernel(const float* in, float* out, long n) { FOR_GRID(i, n) out[i] = silu(in[i]); }
int ir_launch_silu_f32(const float* in, float* out, long n, hipStream_t s) {
    hipLaunchKernelGGL(silu_f32_kernel, GRID1D(n), dim3(256), 0, s, in, out, n);
    return LAUNCH_OK();
}
// sinusoidal timestep embedding, cos || sin (PixArt_blocks.py:336-351 == diffusers Timesteps(flip_sin_to_cos=True, shift 0))
__global__ void timestep_embed_kernel(float* out, float t, int dim) {
    int i = blockIdx.x * 256 + threadIdx.x;
    const int half = dim / 2;
    if (i < half) {
        float freq = expf(-logf(10000.f) * (float)i / (float)half);
        float a = t * freq;
        out[i] = cosf(a);
        out[half + i] = sinf(a);
    }
}
int ir_launch_timestep_embed(float* out, float t, int dim, hipStream_t s) {
    hipLaunchKernelGGL(timestep_embed_kernel, dim3((dim / 2 + 255) / 256), dim3(256), 0, s, out, t, dim);
    return LAUNCH_OK();
}
__global__ void f32_to_bf16_kernel(const float* in, bf16_t* out, long n) { FOR_GRID(i, n) out[i] = f2bf(in[i]); }
int ir_launch_f32_to_bf16(const float* in, bf16_t* out, long n, hipStream_t s) {
    hipLaunchKernelGGL(f32_to_bf16_kernel, GRID1D(n), dim3(256), 0, s, in, out, n);
    return LAUNCH_OK();
}
// adaLN-single tables (PixArtMS.py:74: scale_shift_table[None] + t.reshape(B,6,-1); PixArt_blocks.py:272 for the final
// layer): out[l][i][c] = sst[l][i][c] + t[i*t_stride + c], with +1 folded into the scale rows (bit i of scale_mask)
// so that the LayerNorm kernel applies xn*a + b directly.
__global__ void modtab_kernel(const float* t, const float* sst, float* out, int R, int C, int t_stride, int scale_mask, long total) {
    FOR_GRID(i, total) {
        int c = (int)(i % C);
        int row = (int)((i / C) % R);
        float v = sst[i] + t[row * t_stride + c];
        if ((scale_mask >> row) & 1) v += 1.0f;
        out[i] = v;
    }
}
int ir_launch_modtab(const float* t, const float* sst, float* out, int L, int R, int C, int t_stride, int scale_mask, hipStream_t s) {
    long total = (long)L * R * C;
    hipLaunchKernelGGL(modtab_kernel, GRID1D(total), dim3(256), 0, s, t, sst, out, R, C, t_stride, scale_mask, total);
    return LAUNCH_OK();
}
__global__ void add_bias_rows_kernel(float* x, const float* b, int C, long n) { FOR_GRID(i, n) x[i] += b[i % C]; }
int ir_launch_add_bias_rows(float* x, const float* b, long n, int C, hipStream_t s) {
    hipLaunchKernelGGL(add_bias_rows_kernel, GRID1D(n), dim3(256), 0, s, x, b, C, n);
    return LAUNCH_OK();
}
