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
constexpr int KS = 36;                                               // k-steps: 2 halves x 9 taps x 2 (32 channels each)
constexpr int W_OFF = HALO_BYTES;                                    // weights [KS][4 rows: out channel 0, 1, 2, zeros][64 B]
constexpr int LDS_BYTES = W_OFF + KS * 256;                          // 52 736: three workgroups per CU
constexpr int NV = (HP * 8 + 255) / 256;                             // 11 16-byte vectors per thread and half
}  // namespace vco

__global__ __launch_bounds__(256) void vae_norm_conv_out_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const bf16_t* __restrict__ wgt, const float* __restrict__ bias, float* __restrict__ out,
                                                                int N, int H, int W, int tiles_x, int tiles_per_img, int total_tiles) {
    using namespace vco;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n16 = lane & 15, q = lane >> 4;
    // ---- weights: wgt [32][9][128] bf16 (rows 0..2 real). k-step ks = (half * 9 + tap) * 2 + s covers channels half * 64 + s * 32 .. + 31.
    for (int e = tid; e < KS * 16; e += 256) {   // 16-byte pieces: [ks][row 0..3][4 pieces]
        const int ks = e >> 4, row = (e >> 2) & 3, pc = e & 3;
        const int half = ks / 18, tap = (ks % 18) >> 1, s = ks & 1;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row < 3) v = *reinterpret_cast<const uint4*>(wgt + ((long)row * 9 + tap) * 128 + half * 64 + s * 32 + pc * 8);
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
        for (int half = 0; half < 2; ++half) {
            // ---- the half's halo: load, GroupNorm scale / shift + SiLU, round to bf16, into LDS (zeros outside the image: the conv pads the
            // ACTIVATED tensor)
            float sc[8], sh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sc[e] = scale[(long)img * 128 + half * 64 + cv * 8 + e];
                sh[e] = shift[(long)img * 128 + half * 64 + cv * 8 + e];
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
                v[i] = *reinterpret_cast<const uint4*>(x + pix * 128 + half * 64 + cv * 8);
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
                    const float a = silu(bflo(w4[e]) * sc[2 * e] + sh[2 * e]);
                    const float b = silu(bfhi(w4[e]) * sc[2 * e + 1] + sh[2 * e + 1]);
                    o4[e] = ok[i] ? pack2bf(a, b) : 0u;
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
    hipLaunchKernelGGL(vae_norm_conv_out_kernel, dim3((unsigned)grid), dim3(256), 0, s, x, scale, shift, wgt, bias, out, N, H, W, tiles_x, per, (int)total);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
