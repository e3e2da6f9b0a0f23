// The two ends of the VAE at full resolution, where a 3x3 convolution has 3 channels on one side and the tensor on the other side is a
// gigabyte (2048 x 2048 x 128 bf16): both are HBM-bound, and as generic implicit GEMMs (igemm_kernel with a 32-channel zero-padded
// input / a 32-wide output tile) they ran at a fifth of what the bytes allow.
//
// vae_conv_in_kernel        Encoder.conv_in (ldm/modules/diffusionmodules/model.py:384-388, 3 -> 128): reads the fp32 NCHW image planes
//                           directly (x * in_scale + in_shift, the `*2 - 1` of test_scripts/inference.py:106), k = 27 (tap, channel) pairs
//                           zero-padded to ONE 16x16x32 MFMA k-step, writes bf16 NHWC and the GroupNorm partial statistics of what it
//                           stored. Replaces nchw_to_nhwc_bf16 (a 268 MB round trip through a 32-channel image) + igemm_kernel<..,9,32>.
//                           Bound: the 1 GB store.
// vae_norm_conv_out_kernel  Decoder.norm_out + nonlinearity + conv_out (model.py:650-655, 128 -> 3): reads the bf16 tensor ONCE, applies the
//                           finalised GroupNorm scale / shift and SiLU on the way into LDS (rounded to bf16 there, exactly what the
//                           stand-alone gn_apply pass stored), 36 k-steps of 16x16x32 MFMAs against 3 (of 16) weight rows, writes fp32
//                           [pixel][4]. Replaces gn_apply (1 GB read + 1 GB write) + igemm_kernel<128,32,..,9> (1 GB read). Bound: the 1 GB read.
// Both take exactly the shapes of the released VAE (ch = 128, 3 image channels); anything else stays on the generic kernels.
#include "common.h"
#include "kernels.h"

typedef __attribute__((ext_vector_type(4))) float f32x4_v;

IR_DEVINL float vio_row16_sum(float v) {   // sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), result on every lane of the row
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));   // row_mirror
    return v;
}

// ================================================================================================ conv_in
namespace vci {
constexpr int TH = 8, TW = 64, LW = 68;              // pixel tile, LDS row stride (bf16 elements)
constexpr int PLANE = (TH + 2) * LW;                 // 680 elements per channel plane
constexpr int TILE_BYTES = 3 * PLANE * 2;            // 4080
constexpr int ZERO_OFF = 4096, ZERO_BYTES = 1152;    // zeros for the k >= 27 slots: any fragment immediate (<= 1048) stays inside
constexpr int RED_OFF = ZERO_OFF + ZERO_BYTES;       // [4 waves][32 groups][2] floats
constexpr int LDS_BYTES = RED_OFF + 4 * 32 * 2 * 4;
}  // namespace vci

// Output channel of accumulator tile f, row m: a lane of the 16x16 result (rows 4 q' .. 4 q' + 3 of pixel column n) then holds, over the tile
// pairs (2 j, 2 j + 1), the 8 CONSECUTIVE channels 32 j + 8 q' .. + 7, so one 16-byte store per pair and the four q' lanes of a pixel write
// 64 contiguous bytes per instruction. Pure bookkeeping on the weight side (the rows of A are gathered in this order).
IR_DEVINL int vci_channel(int f, int m) { return 32 * (f >> 1) + 8 * (m >> 2) + 4 * (f & 1) + (m & 3); }

__global__ __launch_bounds__(256) void vae_conv_in_kernel(const float* __restrict__ in, const bf16_t* __restrict__ wgt, const float* __restrict__ bias,
                                                          bf16_t* __restrict__ out, float* __restrict__ gn_part, int N, int H, int W, float in_scale,
                                                          float in_shift, int tiles_x, int tiles_per_img, int total_tiles) {
    using namespace vci;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n16 = lane & 15, q = lane >> 4;
    uint16_t* tile = reinterpret_cast<uint16_t*>(smem);
    // A fragments: tile f, row m = n16, k = 8 q + j = (tap, channel) pair 3 * tap + c (27 of 32 used). wgt: [128][9][32] bf16.
    bf16x8 aw[8];
    f32x4_v binit[8];
    int koff[8];   // element offset of k's input value relative to the fragment's first pixel (+ lane's pixel), or the zero area
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * q + j, tap = k / 3, c = k - 3 * tap, ky = tap / 3, kx = tap - 3 * ky;
        koff[j] = k < 27 ? c * PLANE + ky * LW + kx + n16 : ZERO_OFF / 2;
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        uint16_t w8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * q + j, tap = k / 3, c = k - 3 * tap;
            w8[j] = k < 27 ? wgt[(vci_channel(f, n16) * 9 + tap) * 32 + c] : (uint16_t)0;
        }
        aw[f] = __builtin_bit_cast(bf16x8, w8);
#pragma unroll
        for (int i = 0; i < 4; ++i) binit[f][i] = bias ? bias[vci_channel(f, 4 * q + i)] : 0.f;
    }
    for (int i = tid; i < ZERO_BYTES / 4; i += 256) reinterpret_cast<uint32_t*>(smem + ZERO_OFF)[i] = 0u;
    float* red = reinterpret_cast<float*>(smem + RED_OFF);

    for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
        const int img = t / tiles_per_img, trem = t - img * tiles_per_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        __syncthreads();   // the previous tile's reads of `tile` / `red` are done
        // ---- stage the (TH + 2) x (TW + 2) x 3 input window: scaled, rounded to bf16 (what the old 32-channel image held), zero outside the image
        const float* ip = in + (long)img * 3 * H * W;
        for (int e = tid; e < 3 * (TH + 2) * (TW + 2); e += 256) {
            const int c = e / ((TH + 2) * (TW + 2)), r = e - c * (TH + 2) * (TW + 2);
            const int hy = r / (TW + 2), hx = r - hy * (TW + 2);
            const int y = oy0 + hy - 1, x = ox0 + hx - 1;
            float v = 0.f;
            if (y >= 0 && y < H && x >= 0 && x < W) v = ip[((long)c * H + y) * W + x] * in_scale + in_shift;
            tile[c * PLANE + hy * LW + hx] = (y >= 0 && y < H && x >= 0 && x < W) ? f2bf(v) : (uint16_t)0;
        }
        __syncthreads();
        float gs[8], gq[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) gs[f] = gq[f] = 0.f;
#pragma unroll
        for (int fr = 0; fr < 8; ++fr) {   // wave w: rows 2 w, 2 w + 1; four 16-pixel fragments per row
            const int r = 2 * wid + (fr >> 2), xf = fr & 3;
            const int base = r * LW + 16 * xf;
            uint16_t p8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) p8[j] = tile[koff[j] + base];
            const bf16x8 px = __builtin_bit_cast(bf16x8, p8);
            const int y = oy0 + r, x = ox0 + 16 * xf + n16;
            const bool ok = y < H && x < W;
            bf16_t* op = out + (((long)img * H + y) * W + x) * 128 + 8 * q;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4_v a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[2 * jj], px, binit[2 * jj], 0, 0, 0);
                const f32x4_v b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[2 * jj + 1], px, binit[2 * jj + 1], 0, 0, 0);
                const uint4 pk = make_uint4(pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3]));
                if (ok) {
                    *reinterpret_cast<uint4*>(op + 32 * jj) = pk;
                    const float a0 = bflo(pk.x), a1 = bfhi(pk.x), a2 = bflo(pk.y), a3 = bfhi(pk.y);
                    const float b0 = bflo(pk.z), b1 = bfhi(pk.z), b2 = bflo(pk.w), b3 = bfhi(pk.w);
                    gs[2 * jj] += (a0 + a1) + (a2 + a3);
                    gq[2 * jj] += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                    gs[2 * jj + 1] += (b0 + b1) + (b2 + b3);
                    gq[2 * jj + 1] += (b0 * b0 + b1 * b1) + (b2 * b2 + b3 * b3);
                }
            }
        }
        if (gn_part) {   // fixed-order reduction: the 16 pixel lanes of a DPP row, then the four waves in order (bit-identical run to run)
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                const float s = vio_row16_sum(gs[f]), qq = vio_row16_sum(gq[f]);
                if (n16 == 0) {
                    const int g = 8 * (f >> 1) + 2 * q + (f & 1);   // group of channels vci_channel(f, 4 q .. 4 q + 3)
                    red[(wid * 32 + g) * 2] = s;
                    red[(wid * 32 + g) * 2 + 1] = qq;
                }
            }
            __syncthreads();
            if (tid < 64) {
                const int g = tid & 31, which = tid >> 5;
                const float v = ((red[(0 * 32 + g) * 2 + which] + red[(1 * 32 + g) * 2 + which]) + red[(2 * 32 + g) * 2 + which]) + red[(3 * 32 + g) * 2 + which];
                gn_part[((long)img * tiles_per_img + trem) * 64 + which * 32 + g] = v;
            }
        }
    }
}

int ir_vae_conv_in_tiles(int H, int W) { return ((H + vci::TH - 1) / vci::TH) * ((W + vci::TW - 1) / vci::TW); }

int ir_launch_vae_conv_in(const float* in, const bf16_t* wgt, const float* bias, bf16_t* out, float* gn_part, int N, int H, int W, float in_scale,
                          float in_shift, hipStream_t s) {
    if (N <= 0 || H <= 0 || W <= 0 || (reinterpret_cast<uintptr_t>(out) & 15)) return -2;
    const int tiles_x = (W + vci::TW - 1) / vci::TW, per = ir_vae_conv_in_tiles(H, W);
    const long total = (long)N * per;
    if (total > 0x7fffffffL) return -12;
    const long grid = total < 2048 ? total : 2048;   // 8 workgroups per CU; a workgroup gathers its weight fragments once and walks its tiles
    hipLaunchKernelGGL(vae_conv_in_kernel, dim3((unsigned)grid), dim3(256), 0, s, in, wgt, bias, out, gn_part, N, H, W, in_scale, in_shift, tiles_x, per, (int)total);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ================================================================================================ norm_out + SiLU + conv_out
namespace vco {
constexpr int TH = 8, TW = 32, HWD = TW + 2, HP = (TH + 2) * HWD;   // 340 halo pixels
constexpr int ROWB = 128;                                            // 64 channels of a half per halo pixel
constexpr int HALO_BYTES = HP * ROWB;                                // 43 520
constexpr int W_OFF = HALO_BYTES;                                    // weights [KS][4 rows: out channel 0, 1, 2, zeros][64 B]
constexpr int NV = (HP * 8 + 255) / 256;                             // 11 16-byte vectors per thread and half
}  // namespace vco

// HALVES x 64 input channels, KS = HALVES x 9 taps x 2 k-steps of 32 channels. <2, true>: the VAE's norm_out + SiLU + conv_out (52 736 B of LDS: three
// workgroups per CU). <1, false> (round 6): SwinIR's conv_last (swinir.py:896, 64 -> 3 at full resolution, no normalisation; `x / img_range + mean` is
// folded into its weights and bias by weights.py) - the generic implicit GEMM spent 0.49 ms on the 0.5 GB it reads at 2048 x 2048.
template <int HALVES, bool NORM>
__global__ __launch_bounds__(256) void vae_norm_conv_out_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const bf16_t* __restrict__ wgt, const float* __restrict__ bias, float* __restrict__ out,
                                                                int N, int H, int W, int tiles_x, int tiles_per_img, int total_tiles) {
    using namespace vco;
    constexpr int C = HALVES * 64, KS = HALVES * 18, LDS_BYTES = W_OFF + KS * 256;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n16 = lane & 15, q = lane >> 4;
    // ---- weights: wgt [32][9][C] bf16 (rows 0..2 real). k-step ks = (half * 9 + tap) * 2 + s covers channels half * 64 + s * 32 .. + 31.
    for (int e = tid; e < KS * 16; e += 256) {   // 16-byte pieces: [ks][row 0..3][4 pieces]
        const int ks = e >> 4, row = (e >> 2) & 3, pc = e & 3;
        const int half = ks / 18, tap = (ks % 18) >> 1, s = ks & 1;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row < 3) v = *reinterpret_cast<const uint4*>(wgt + ((long)row * 9 + tap) * C + half * 64 + s * 32 + pc * 8);
        *reinterpret_cast<uint4*>(smem + W_OFF + ks * 256 + row * 64 + pc * 16) = v;
    }
    const uint32_t lds0 = lds_addr(smem);
    const uint32_t wrd = lds0 + W_OFF + (uint32_t)min(n16, 3) * 64 + q * 16;   // A fragment: row m = n16 (rows >= 3: the zero row), chunk q
    // B fragment of (kx, x half xf, s): halo column hx = 16 xf + kx + n16, 16-byte chunk 4 s + q, swizzled by the column
    uint32_t prd[3][2][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int xf = 0; xf < 2; ++xf)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int hx = 16 * xf + kx + n16;
                prd[kx][xf][s] = lds0 + hx * ROWB + (((4 * s + q) ^ ((hx >> 1) & 7)) << 4);
            }
    const int cv = tid & 7;                      // this thread's 16-byte chunk (8 channels) of every halo pixel it stages
    f32x4_v bsel = {0.f, 0.f, 0.f, 0.f};
    if (q == 0) bsel = f32x4_v{bias ? bias[0] : 0.f, bias ? bias[1] : 0.f, bias ? bias[2] : 0.f, 0.f};

    for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
        const int img = t / tiles_per_img, trem = t - img * tiles_per_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        f32x4_v acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = bsel;
#pragma unroll 1
        for (int half = 0; half < HALVES; ++half) {
            // ---- the half's halo: load, GroupNorm scale / shift + SiLU, round to bf16, into LDS (zeros outside the image: the conv pads the
            // ACTIVATED tensor)
            float sc[8], sh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sc[e] = NORM ? scale[(long)img * C + half * 64 + cv * 8 + e] : 1.f;
                sh[e] = NORM ? shift[(long)img * C + half * 64 + cv * 8 + e] : 0.f;
            }
            uint4 v[NV];
            bool ok[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int hp = (tid >> 3) + 32 * i;
                const int hy = hp / HWD, hx = hp - hy * HWD;
                const int y = oy0 + hy - 1, xx = ox0 + hx - 1;
                ok[i] = hp < HP && y >= 0 && y < H && xx >= 0 && xx < W;
                const long pix = ((long)img * H + min(max(y, 0), H - 1)) * W + min(max(xx, 0), W - 1);
                v[i] = *reinterpret_cast<const uint4*>(x + pix * C + half * 64 + cv * 8);
            }
            __syncthreads();   // every wave has finished the MFMAs of the previous half / tile (and the weight fill, first time)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int hp = (tid >> 3) + 32 * i;
                const int hx = hp % HWD;
                const uint32_t w4[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
                uint32_t o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (NORM) {
                        const float a = silu(bflo(w4[e]) * sc[2 * e] + sh[2 * e]);
                        const float b = silu(bfhi(w4[e]) * sc[2 * e + 1] + sh[2 * e + 1]);
                        o4[e] = ok[i] ? pack2bf(a, b) : 0u;
                    } else {
                        o4[e] = ok[i] ? w4[e] : 0u;
                    }
                }
                if (hp < HP) *reinterpret_cast<uint4*>(smem + hp * ROWB + ((cv ^ ((hx >> 1) & 7)) << 4)) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
            }
            __syncthreads();
            // ---- 18 k-steps: wave w owns patch rows 2 w, 2 w + 1, two 16-pixel fragments each
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int ks = (half * 9 + tap) * 2 + s;
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(smem + (wrd - lds0) + ks * 256);
#pragma unroll
                    for (int fr = 0; fr < 4; ++fr) {
                        const int r = 2 * wid + (fr >> 1), xf = fr & 1;
                        const bf16x8 b = *reinterpret_cast<const bf16x8*>(smem + (prd[kx][xf][s] - lds0) + (r + ky) * HWD * ROWB);
                        acc[fr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[fr], 0, 0, 0);
                    }
                }
            }
        }
        if (q == 0) {   // rows 0..3 of the result = out channels 0, 1, 2 and the unused fourth lane of the [pixel][4] tensor
#pragma unroll
            for (int fr = 0; fr < 4; ++fr) {
                const int y = oy0 + 2 * wid + (fr >> 1), xx = ox0 + 16 * (fr & 1) + n16;
                if (y < H && xx < W) *reinterpret_cast<f32x4_v*>(out + (((long)img * H + y) * W + xx) * 4) = acc[fr];
            }
        }
    }
}

int ir_launch_vae_norm_conv_out(const bf16_t* x, const float* scale, const float* shift, const bf16_t* wgt, const float* bias, float* out, int N, int H,
                                int W, hipStream_t s) {
    if (N <= 0 || H <= 0 || W <= 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(wgt) & 15)) return -2;
    const int tiles_x = (W + vco::TW - 1) / vco::TW, per = ((H + vco::TH - 1) / vco::TH) * tiles_x;
    const long total = (long)N * per;
    if (total > 0x7fffffffL) return -12;
    const long grid = total < 768 * 8 ? total : 768 * 8;   // three workgroups per CU resident, eight rounds of them
    hipLaunchKernelGGL((vae_norm_conv_out_kernel<2, true>), dim3((unsigned)grid), dim3(256), 0, s, x, scale, shift, wgt, bias, out, N, H, W, tiles_x, per, (int)total);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// 3 x 3 conv 64 -> 3 (output [pixel][4] fp32, like the VAE's): x [N][H][W][64] bf16, wgt [32][9][64] bf16 (rows 0..2 real), bias[3]
int ir_launch_conv64_to3(const bf16_t* x, const bf16_t* wgt, const float* bias, float* out, int N, int H, int W, hipStream_t s) {
    if (N <= 0 || H <= 0 || W <= 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(wgt) & 15)) return -2;
    const int tiles_x = (W + vco::TW - 1) / vco::TW, per = ((H + vco::TH - 1) / vco::TH) * tiles_x;
    const long total = (long)N * per;
    if (total > 0x7fffffffL) return -12;
    const long grid = total < 768 * 8 ? total : 768 * 8;
    hipLaunchKernelGGL((vae_norm_conv_out_kernel<1, false>), dim3((unsigned)grid), dim3(256), 0, s, x, nullptr, nullptr, wgt, bias, out, N, H, W, tiles_x, per, (int)total);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ================================================================================================ 64 -> 64 at full resolution (SwinIR conv_hr)
// SwinIR's conv_hr (swinir.py:895, 64 -> 64 + LeakyReLU at the OUTPUT resolution: 2048 x 2048 on the headline path) moves 0.54 GB in and 0.54 GB out
// for 0.31 TFLOP, and as conv_halo_kernel<64> (8 x 16 patches, nine k-steps per workgroup, prologue and epilogue of 32 768 workgroups) it took
// 0.56 ms inside the pipeline (0.67 alone). Here: vae_norm_conv_out_kernel's read side (8 x 32 patch, its 10 x 34 x 64-channel halo in LDS, each wave
// two patch rows) with all 64 output channels (four 16-row weight tiles per k-step, the whole 72 KB weight image resident in LDS: one persistent
// workgroup per CU) and vae_conv_in_kernel's store side (weight rows ordered so that a lane's accumulators of tiles 2j, 2j+1 are 8 consecutive
// channels: 16-byte stores, 64 contiguous bytes per pixel and instruction). The halo travels through registers two tiles ahead.
// 0.38 ms inside the pipeline, 0.45 alone (2.4 TB/s). Knock-outs (-DIR_C64_KO, alone): without MFMAs 0.23 ms, without stores 0.36, without the halo
// loads / staging 0.34 - the parts add up instead of overlapping: per k-step a wave reads 8 KB of fragments for 16 MFMAs, so the four waves keep
// the LDS pipe as busy as the matrix pipe (576 KB per tile at 128 B per clock = the 4 600 cycles of its 288 MFMAs) and one wave per SIMD has
// nothing to hide the rest behind. What did NOT matter (each built and timed): pinning the fragment reads a k-step ahead, the halo two tiles
// instead of one ahead, precomputed per-lane offsets, inline-asm loads with hand-counted waits (hipcc's own waits did sit in front of every
// tile's loads). The next step would be the weights in registers (half the LDS traffic) - not built.
#ifndef IR_C64_KO
#define IR_C64_KO 0   // knock-out builds, timing only (results wrong by design; never set in the library): 1 no MFMAs, 2 no stores, 3 halo fetched once, 4 no LDS staging
#endif
typedef unsigned int c64_u32x4 __attribute__((ext_vector_type(4)));
namespace c64 {
using namespace vco;                                   // TH, TW, HWD, HP, ROWB, HALO_BYTES, NV
constexpr int KS = 18;                                  // 9 taps x 2 k-steps of 32 channels
constexpr int WT_OFF = HALO_BYTES;                      // weights [KS][4 tiles][16 rows][64 B], chunk-swizzled by the row
constexpr int LDS_BYTES = WT_OFF + KS * 4096;           // 117 248
}  // namespace c64

__global__ __launch_bounds__(256) void conv64_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wgt, const float* __restrict__ bias,
                                                      bf16_t* __restrict__ out, int N, int H, int W, int act, float slope, int tiles_x,
                                                      int tiles_per_img, int total_tiles) {
    using namespace c64;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n16 = lane & 15, q = lane >> 4;
    // ---- weights: wgt [64][9][64] bf16. LDS row (tile f, row m) holds output channel vci_channel(f, m); k-step ks = tap * 2 + s covers input channels
    // 32 s .. 32 s + 31; the 16-byte chunk c of a row sits at slot c ^ ((m >> 2) & 3) (16 rows x 64 B: rows m and m + 4 would share banks)
    for (int e = tid; e < KS * 64 * 4; e += 256) {
        const int pc = e & 3, m = (e >> 2) & 15, f = (e >> 6) & 3, ks = e >> 8;
        const int tap = ks >> 1, sh = ks & 1;
        const uint4 v = *reinterpret_cast<const uint4*>(wgt + ((long)vci_channel(f, m) * 9 + tap) * 64 + sh * 32 + pc * 8);
        *reinterpret_cast<uint4*>(smem + WT_OFF + ks * 4096 + f * 1024 + m * 64 + ((pc ^ ((m >> 2) & 3)) << 4)) = v;
    }
    const uint32_t lds0 = lds_addr(smem);
    const uint32_t wrd = WT_OFF + n16 * 64 + ((q ^ ((n16 >> 2) & 3)) << 4);   // A fragment of tile f, k-step ks: + ks * 4096 + f * 1024
    uint32_t prd[3][2][2];                                                     // B fragments: as vae_norm_conv_out_kernel
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int xf = 0; xf < 2; ++xf)
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                const int hx = 16 * xf + kx + n16;
                prd[kx][xf][sh] = hx * ROWB + (((4 * sh + q) ^ ((hx >> 1) & 7)) << 4);
            }
    const int cv = tid & 7;
    // bias of this lane's 4 channels of tile f: channels vci_channel(f, 4 q .. 4 q + 3) = 32 (f >> 1) + 8 q + 4 (f & 1) + 0..3
    f32x4_v bsel[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int c0 = 32 * (f >> 1) + 8 * q + 4 * (f & 1);
        bsel[f] = bias ? f32x4_v{bias[c0], bias[c0 + 1], bias[c0 + 2], bias[c0 + 3]} : f32x4_v{0.f, 0.f, 0.f, 0.f};
    }
    // the bias loads must have LANDED before the tile loop: hipcc's wait-count pass carries "a load into these registers is pending" around the
    // loop and put an s_waitcnt vmcnt(0) in front of the first MFMA of every other tile - behind the halo loads just issued. An empty asm that
    // reads them forces the wait here.
#pragma unroll
    for (int f = 0; f < 4; ++f) asm volatile("" ::"v"(bsel[f][0]), "v"(bsel[f][1]), "v"(bsel[f][2]), "v"(bsel[f][3]));
    // per-thread constants of the halo staging (tile-independent): halo pixel hp_i = (tid >> 3) + 32 i -> its element offset from the halo's first
    // pixel, its LDS address, whether it exists; an INTERIOR tile (the whole halo inside the image) then needs one add per load and no clamps
    // (unsigned 32-bit BYTE offsets from a wave-uniform base: the loads take the `saddr + voffset` form, no per-lane 64-bit address arithmetic)
    uint32_t soff[NV];
    int ldst[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int hp = (tid >> 3) + 32 * i;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        soff[i] = (uint32_t)((hy * W + hx) * 64 + cv * 8) * 2u;
        ldst[i] = hp < HP ? hp * ROWB + ((cv ^ ((hx >> 1) & 7)) << 4) : -1;
    }
    // the halo travels through registers TWO tiles ahead (sets A and B, alternating)
    c64_u32x4 vA[NV], vB[NV];
    bool okA[NV], okB[NV];
    auto fetch = [&](c64_u32x4 (&v)[NV], bool (&ok)[NV], int t) {
        const int img = t / tiles_per_img, trem = t - img * tiles_per_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
        if (y0 >= 0 && x0 >= 0 && y0 + TH + 2 <= H && x0 + TW + 2 <= W) {   // (uniform)
            const unsigned char* base = reinterpret_cast<const unsigned char*>(x + (((long)img * H + y0) * W + x0) * 64);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                ok[i] = true;
                v[i] = *reinterpret_cast<const c64_u32x4*>(base + (ldst[i] >= 0 ? soff[i] : soff[0]));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int hp = (tid >> 3) + 32 * i;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int y = y0 + hy, xx = x0 + hx;
            ok[i] = hp < HP && y >= 0 && y < H && xx >= 0 && xx < W;
            const long pix = ((long)img * H + min(max(y, 0), H - 1)) * W + min(max(xx, 0), W - 1);
            v[i] = *reinterpret_cast<const c64_u32x4*>(x + pix * 64 + cv * 8);
        }
    };
    auto stage = [&](c64_u32x4 (&v)[NV], bool (&ok)[NV]) {   // the fetched halo (registers) into LDS, zeros outside the image
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (ldst[i] >= 0) *reinterpret_cast<c64_u32x4*>(smem + ldst[i]) = ok[i] ? v[i] : c64_u32x4{0u, 0u, 0u, 0u};
    };
    int sto[4];   // element offset of this lane's pixel of fragment fr from the tile's first pixel
#pragma unroll
    for (int fr = 0; fr < 4; ++fr) sto[fr] = ((2 * wid + (fr >> 1)) * W + 16 * (fr & 1) + n16) * 64 + 8 * q;
    const int G = gridDim.x;
    int t = blockIdx.x;
    if (t < total_tiles) fetch(vA, okA, t);
    if (t + G < total_tiles) fetch(vB, okB, t + G);
    __syncthreads();   // the weight fill
    if (t < total_tiles) stage(vA, okA);
    // Per tile: [barrier: halo visible] request the halo two tiles ahead into the register set just staged - 288 MFMAs - [barrier: every wave is done
    // with the halo] the NEXT tile's halo (requested one tile ago) into LDS - this tile's stores (last: vmcnt counts loads and stores in issue order).
    auto tile = [&](c64_u32x4 (&vcur)[NV], bool (&okcur)[NV], c64_u32x4 (&vnxt)[NV], bool (&oknxt)[NV], int tt) __attribute__((always_inline)) {
        const int img = tt / tiles_per_img, trem = tt - img * tiles_per_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        __syncthreads();
        if (tt + 2 * G < total_tiles && IR_C64_KO != 3) fetch(vcur, okcur, tt + 2 * G);   // (its previous content is in LDS)
        f32x4_v acc[4][4];   // [weight tile f][pixel fragment fr = 2 * (row of the wave) + x half]
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int fr = 0; fr < 4; ++fr) acc[f][fr] = bsel[f];
        // fragments double-buffered in registers, the order pinned (hipcc otherwise reads a k-step's eight fragments right in front of its MFMAs
        // and waits for them - with one wave per SIMD the LDS latency was exposed 18 times per tile): first two MFMAs of step ks, then the reads of
        // step ks + 1, then the other fourteen MFMAs
        bf16x8 a[2][4], b[2][4];
        auto load = [&](auto ksc, int set) __attribute__((always_inline)) {
            constexpr int ks = decltype(ksc)::value, tap = ks >> 1, sh = ks & 1, ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int f = 0; f < 4; ++f) a[set][f] = *reinterpret_cast<const bf16x8*>(smem + wrd + ks * 4096 + f * 1024);
#pragma unroll
            for (int fr = 0; fr < 4; ++fr) {
                const int r = 2 * wid + (fr >> 1), xf = fr & 1;
                b[set][fr] = *reinterpret_cast<const bf16x8*>(smem + prd[kx][xf][sh] + (r + ky) * HWD * ROWB);
            }
        };
        auto mfmas = [&](int set, int first, int last) __attribute__((always_inline)) {
#pragma unroll
            for (int e = first; e < last; ++e)
                if (IR_C64_KO != 1 || e == first) acc[e >> 2][e & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[set][e >> 2], b[set][e & 3], acc[e >> 2][e & 3], 0, 0, 0);
        };
        load(std::integral_constant<int, 0>{}, 0);
        [&]<int... KSI>(std::integer_sequence<int, KSI...>) {
            ([&] {
                __builtin_amdgcn_sched_barrier(0);
                mfmas(KSI & 1, 0, 2);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (KSI + 1 < KS) load(std::integral_constant<int, KSI + 1>{}, (KSI + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(KSI & 1, 2, 16);
            }(), ...);
        }(std::make_integer_sequence<int, KS>{});
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (tt + G < total_tiles && IR_C64_KO != 4) stage(vnxt, oknxt);
        // ---- store: per pixel fragment and tile pair j the lane's 8 consecutive channels 32 j + 8 q .. + 7
#pragma unroll
        for (int fr = 0; fr < 4; ++fr) {
            const int y = oy0 + 2 * wid + (fr >> 1), xx = ox0 + 16 * (fr & 1) + n16;
            if (y < H && xx < W && (IR_C64_KO != 2 || acc[0][fr][0] == 1234.5f)) {
                bf16_t* dst = out + (((long)img * H + oy0) * W + ox0) * 64 + sto[fr];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = acc[2 * j][fr][e]; o[4 + e] = acc[2 * j + 1][fr][e]; }
                    if (act == IR_ACT_LRELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = o[e] > 0.f ? o[e] : o[e] * slope;
                    }
                    *reinterpret_cast<uint4*>(dst + 32 * j) = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
                }
            }
        }
    };
    for (; t < total_tiles; t += 2 * G) {
        tile(vA, okA, vB, okB, t);
        if (t + G < total_tiles) tile(vB, okB, vA, okA, t + G);
    }
}

// Opt-in (IR_CONV64=1): on the stress fixture at 2048 x 2048 the bf16 path's PSNR against the oracle moves by +- 0.7 dB with the summation ORDER of
// this one conv (four combinations of this kernel and conv_last's: 44.30 / 44.46 / 45.10 / 45.71 dB, profiles/r06_stress_sensitivity.txt) and this
// kernel's order lands at the low end; for 0.18 ms per image the default keeps conv_halo_kernel. ir_launch_conv64(p, s, true) (the op test) skips the switch.
static bool conv64_shape(const IGemmParams& p);
bool ir_conv64_takes(const IGemmParams& p) {
    static const bool on = getenv("IR_CONV64") != nullptr;
    return on && conv64_shape(p);
}
static bool conv64_shape(const IGemmParams& p) {
    if (g_ir_plain_kernels || p.fp8 || p.force_generic || p.up || p.up2x2 || p.gn_part || p.nrm_scale) return false;
    if (p.taps != 9 || p.stride != 1 || p.pad != 1 || p.Cin != 64 || p.in_cs != 64 || p.Cout != 64 || p.Cout_pad != 64 || p.out_cs != 64) return false;
    if (p.res || p.out_f32 || p.out2 || p.gate || p.out_scale != 1.f || p.wgt_rs != 9 * 64) return false;
    if (p.act != IR_ACT_NONE && p.act != IR_ACT_LRELU) return false;
    if ((reinterpret_cast<uintptr_t>(p.in) & 15) || (reinterpret_cast<uintptr_t>(p.out) & 15) || (reinterpret_cast<uintptr_t>(p.wgt) & 15)) return false;
    return (long)p.Ho * p.Wo >= 256L * 256;   // per IMAGE: a tile (8 x 32 pixels) per CU at least; smaller maps stay on conv_halo_kernel
}
int ir_launch_conv64(const IGemmParams& p, hipStream_t s, bool force) {
    if (!(force ? conv64_shape(p) : ir_conv64_takes(p))) return -2;
    const int tiles_x = (p.Wo + vco::TW - 1) / vco::TW, per = ((p.Ho + vco::TH - 1) / vco::TH) * tiles_x;
    const long total = (long)p.NB * per;
    if (total > 0x7fffffffL) return -12;
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        cus = n;
    }
    const long grid = total < cus ? total : cus;   // persistent: one workgroup per CU (117 KB of LDS) fills its weight image once and walks its tiles
    hipLaunchKernelGGL(conv64_kernel, dim3((unsigned)grid), dim3(256), 0, s, p.in, p.wgt, p.bias, reinterpret_cast<bf16_t*>(p.out), p.NB, p.Ho, p.Wo, p.act, p.slope,
                       tiles_x, per, (int)total);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
