// Implicit-GEMM convolution / linear kernel for gfx950 (bf16 MFMA 32x32x16, fp32 accumulate).
//
//   out[m][n] = epilogue( sum_{tap,c} in[pixel(m,tap)][c] * wgt[n][tap][c] )
//
// One kernel family covers every dense contraction of the hot path:
//   * TAPS = 9 : 3x3 convolutions over NHWC activations — VAE ResnetBlock/Downsample/Upsample convs
//                (reference ldm/modules/diffusionmodules/model.py:57-61,76-86,102-116), SwinIR convs
//                (reference diffusion/model/swinir.py:476,709,773,803-813). Stride-2 with the
//                asymmetric (0,1,0,1) pad and the nearest-2x upsample are folded into the A-tile
//                addressing, so neither a padded nor an upsampled tensor is ever materialised.
//   * TAPS = 1 : linears / 1x1 convs — DiT qkv/proj/MLP (PixArt_blocks.py:123-158, PixArtMS.py:67-77),
//                SwinIR qkv/proj/MLP (swinir.py:35-41,132,154), VAE nin_shortcut/q/k/v/proj_out.
// Tiling: 256 threads = 4 waves; block tile BM x BN x 32; A/B tiles are register-staged into LDS
// rows of 80 B (64 B data + 16 B pad => conflict-free ds_read_b128 fragment reads), double-buffered,
// one barrier per k-tile. The epilogue transposes each wave's accumulators through LDS so that all
// global traffic (bias, residual, gate, stores) is row-contiguous 8/16-byte vectors.
#include "common.h"
#include "kernels.h"

#define LDS_STRIDE 40  // bf16 elements per LDS row (32 data + 8 pad)

template <int BM, int BN, int WM, int WN, int TAPS>
__global__ __launch_bounds__(256) void igemm_kernel(IGemmParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_CH = BM * 4 / 256;  // 16-byte chunks per thread per A tile
    constexpr int B_CH = (BN * 4 + 255) / 256;
    constexpr int COLS = TN * 32;
    constexpr int LDS_AB = 2 * (BM + BN) * LDS_STRIDE * 2;
    constexpr int LDS_EP = 4 * 32 * COLS * 4;
    constexpr int LDS_BYTES = LDS_AB > LDS_EP ? LDS_AB : LDS_EP;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    bf16_t(*As)[BM][LDS_STRIDE] = reinterpret_cast<bf16_t(*)[BM][LDS_STRIDE]>(smem);
    bf16_t(*Bs)[BN][LDS_STRIDE] =
        reinterpret_cast<bf16_t(*)[BN][LDS_STRIDE]>(smem + 2 * BM * LDS_STRIDE * 2);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give one XCD the n-tiles of the
    // same m-tile back to back so the A tile is re-read from that L2. Speed only, never correctness.
    const int NT = p.Cout_pad / BN;
    const int MT = (p.M + BM - 1) / BM;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int mt = (j / NT) * 8 + xcd, nt = j % NT;
    if (mt >= MT) return;
    const int m0 = mt * BM, n0 = nt * BN;

    const int cchunks = p.Cin >> 5;
    const int KT = TAPS * cchunks;
    const long Kw = p.wgt_rs;

    // ---- per-thread A rows
    const int seg = tid & 3;
    int a_n[A_CH], a_oy[A_CH], a_ox[A_CH];
    bool a_ok[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        int m = m0 + (tid >> 2) + i * 64;
        a_ok[i] = m < p.M;
        if (TAPS == 1) {
            a_n[i] = m; a_oy[i] = 0; a_ox[i] = 0;
        } else {
            int hw = p.Ho * p.Wo;
            int n = m / hw, rem = m - n * hw;
            a_n[i] = n; a_oy[i] = rem / p.Wo; a_ox[i] = rem - a_oy[i] * p.Wo;
        }
    }
    const int Hc = p.up ? 2 * p.H : p.H, Wc = p.up ? 2 * p.W : p.W;

    uint4 a_reg[A_CH], b_reg[B_CH];  // initialised: an uninitialised array written under a condition stays in scratch
#pragma unroll
    for (int i = 0; i < A_CH; ++i) a_reg[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < B_CH; ++i) b_reg[i] = make_uint4(0, 0, 0, 0);
    auto load_tile = [&](int kt) {
        int tap = 0, cc = kt;
        if (TAPS > 1) { tap = kt / cchunks; cc = kt - tap * cchunks; }
        const int c0 = cc * 32 + seg * 8;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const bf16_t* src = nullptr;
            if (TAPS == 1) {
                if (a_ok[i]) src = p.in + (long)a_n[i] * p.in_cs + c0;
            } else {
                int ky = tap / 3, kx = tap - ky * 3;
                int cy = a_oy[i] * p.stride + ky - p.pad, cx = a_ox[i] * p.stride + kx - p.pad;
                if (a_ok[i] && cy >= 0 && cy < Hc && cx >= 0 && cx < Wc) {
                    int iy = cy >> p.up, ix = cx >> p.up;
                    src = p.in + (((long)a_n[i] * p.H + iy) * p.W + ix) * p.in_cs + c0;
                }
            }
            a_reg[i] = src ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            int row = (tid >> 2) + i * 64;
            if (BN >= 64 || row < BN)
                b_reg[i] = *reinterpret_cast<const uint4*>(p.wgt + (long)(n0 + row) * Kw + (long)kt * 32 + seg * 8);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i)
            *reinterpret_cast<uint4*>(&As[buf][(tid >> 2) + i * 64][seg * 8]) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            int row = (tid >> 2) + i * 64;
            if (BN >= 64 || row < BN) *reinterpret_cast<uint4*>(&Bs[buf][row][seg * 8]) = b_reg[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) load_tile(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(&As[cur][wm * (BM / WM) + i * 32 + r][ks * 16 + h * 8]);
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
                bfr[jn] = *reinterpret_cast<const bf16x8*>(&Bs[cur][wn * (BN / WN) + jn * 32 + r][ks * 16 + h * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mfma32(af[i], bfr[jn], acc[i][jn]);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: per wave, 32 x COLS fp32 slab through LDS, then row-contiguous vector I/O
    float* slab = reinterpret_cast<float*>(smem) + wid * 32 * COLS;
    constexpr int LPR = COLS / 4;       // lanes per row
    constexpr int RPI = 64 / LPR;       // rows per iteration
    const int ecol = (lane % LPR) * 4;  // column (within the wave tile) of this lane's 4-vector
    const int nbase = n0 + wn * (BN / WN) + ecol;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bias4[e] = p.bias[nbase + e];
    }
    const bool vec_ok = p.vec && (nbase + 3 < p.Cout);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        __syncthreads();
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int g = 0; g < 16; ++g) slab[mfma_row(g, lane) * COLS + jn * 32 + r] = acc[i][jn][g];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int row = it * RPI + lane / LPR;
            const int m = m0 + wm * (BM / WM) + i * 32 + row;
            if (m >= p.M || nbase >= p.Cout) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(&slab[row * COLS + ecol]);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = v[e] + bias4[e];
                switch (p.act) {
                    case IR_ACT_GELU_ERF: x = gelu_erf(x); break;
                    case IR_ACT_GELU_TANH: x = gelu_tanh(x); break;
                    case IR_ACT_LRELU: x = x > 0.f ? x : x * p.slope; break;
                    case IR_ACT_SILU: x = silu(x); break;
                    default: break;
                }
                o[e] = x * p.out_scale;
            }
            if (p.gate) {
                const float* g = p.gate + (long)(m / p.rows_per_batch) * p.gate_stride + nbase;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (nbase + e < p.Cout) o[e] *= g[e];
            }
            if (p.res) {
                const long rm = p.res_mod > 0 ? (long)(m % p.res_mod) : (long)m;
                if (p.res_f32) {
                    const float* rp = reinterpret_cast<const float*>(p.res) + rm * p.res_cs + nbase;
                    if (vec_ok) {
                        f32x4 rv = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] += rv[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (nbase + e < p.Cout) o[e] += rp[e];
                    }
                } else {
                    const bf16_t* rp = reinterpret_cast<const bf16_t*>(p.res) + rm * p.res_cs + nbase;
                    if (vec_ok) {
                        uint2 rv = *reinterpret_cast<const uint2*>(rp);
                        o[0] += bflo(rv.x); o[1] += bfhi(rv.x); o[2] += bflo(rv.y); o[3] += bfhi(rv.y);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (nbase + e < p.Cout) o[e] += bf2f(rp[e]);
                    }
                }
            }
            if (p.out_f32) {
                float* op = reinterpret_cast<float*>(p.out) + (long)m * p.out_cs + nbase;
                if (vec_ok) {
                    f32x4 ov = {o[0], o[1], o[2], o[3]};
                    *reinterpret_cast<f32x4*>(op) = ov;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (nbase + e < p.Cout) op[e] = o[e];
                }
            } else {
                bf16_t* op = reinterpret_cast<bf16_t*>(p.out) + (long)m * p.out_cs + nbase;
                if (vec_ok) {
                    *reinterpret_cast<uint2*>(op) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (nbase + e < p.Cout) op[e] = f2bf(o[e]);
                }
            }
            if (p.out2) {
                bf16_t* op = p.out2 + (long)m * p.out2_cs + nbase;
                if (vec_ok) {
                    *reinterpret_cast<uint2*>(op) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (nbase + e < p.Cout) op[e] = f2bf(o[e]);
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const IGemmParams& p, hipStream_t s) {
    const int MT = (p.M + BM - 1) / BM, NT = p.Cout_pad / BN;
    const int grid = ((MT + 7) / 8) * 8 * NT;
    if (p.taps == 9)
        hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 9>), dim3(grid), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 1>), dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Host launcher. Returns 0 or a negative error code; validates every shape assumption the kernel makes.
int ir_launch_igemm(const IGemmParams& pin, hipStream_t s) {
    IGemmParams p = pin;
    if (p.M <= 0) return 0;
    if (p.taps != 1 && p.taps != 9) return -2;
    if (p.Cin <= 0 || (p.Cin & 31) || (p.in_cs & 7) || p.in_cs < p.Cin) return -3;
    if (p.Cout <= 0 || p.Cout > p.Cout_pad || (p.Cout_pad & 31)) return -4;
    if (p.wgt_rs < (long)p.taps * p.Cin || (p.wgt_rs & 7)) return -10;
    p.vec = !((p.out_cs & 3) || (p.res && (p.res_cs & 3)) || (p.out2 && (p.out2_cs & 3)) ||
              (reinterpret_cast<uintptr_t>(p.res) & 15) || (reinterpret_cast<uintptr_t>(p.out2) & 7) ||
              (p.gate && (p.gate_stride & 3)));
    if (!p.out) return -5;
    if (p.gate && p.rows_per_batch <= 0) return -6;
    if ((reinterpret_cast<uintptr_t>(p.in) & 15) || (reinterpret_cast<uintptr_t>(p.wgt) & 15) ||
        (reinterpret_cast<uintptr_t>(p.out) & 15))
        return -7;
    if (p.taps == 9) {
        if (p.stride != 1 && p.stride != 2) return -8;
        if ((long)p.NB * p.Ho * p.Wo != p.M) return -9;
    }
    if (p.Cout_pad % 128 == 0) return launch_cfg<128, 128, 2, 2>(p, s);
    if (p.Cout_pad % 64 == 0) return launch_cfg<128, 64, 2, 2>(p, s);
    return launch_cfg<128, 32, 4, 1>(p, s);
}
