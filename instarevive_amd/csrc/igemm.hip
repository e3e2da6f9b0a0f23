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
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <utility>
#include "common.h"
#include "kernels.h"


// 16-byte LDS-DMA: LDS[wave-uniform base + lane*16] <- *per-lane global address (asynchronous, counted by vmcnt).
IR_DEVINL void glds16(const void* g, lds_ptr_t l) { __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0); }

// 64 KB of zeros: zero-padding taps of a convolution read from here, so the LDS-DMA never needs a mask or a branch.
__device__ uint4 g_zero_page[4096];

// Diagnostic build only (-DIR_STAMPS, tools/conv_stamp.hip): per-block phase time stamps; never compiled into the library.
#ifdef IR_STAMPS
__device__ unsigned long long g_stamps[65536 * 8];  // [0..3] s_memrealtime (100 MHz) per phase edge, [4..7] s_memtime (core clock)
#define IR_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 65536) { g_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
                                                                        g_stamps[blockIdx.x * 8 + 4 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#ifndef IR_KO
#define IR_KO 0  // knock-out experiments of tools/conv_stamp.hip (results are wrong by design): 1 no per-step barrier, 2 no weight
#endif           // re-staging, 3 no LDS fragment reads, 4 no MFMAs
#else
#define IR_STAMP(k) do { } while (0)
#define IR_KO 0
#endif

// Epilogue shared by the igemm and halo-conv kernels: each wave transposes its accumulators through a private 32 x COLS fp32
// LDS slab so that bias / activation / gate / residual / stores are row-contiguous 8-16-byte vectors.
// map_row(i, row) gives the global output row (pixel / token index) of row `row` of the wave's i-th 32-row tile, or -1.
template <int ACT>
IR_DEVINL float apply_act(float x, float slope) {
    if (ACT == IR_ACT_GELU_ERF) return gelu_erf(x);
    if (ACT == IR_ACT_GELU_TANH) return gelu_tanh(x);
    if (ACT == IR_ACT_LRELU) return x > 0.f ? x : x * slope;
    if (ACT == IR_ACT_SILU) return silu(x);
    return x;
}

// Position of accumulator register g of a 32 x 32 output tile. M16 = false: one v_mfma_f32_32x32x16 tile. M16 = true: the tile is four
// v_mfma_f32_16x16x32 tiles, g = (a*2 + b)*4 + q for sub-tile (a, b) and register q.
template <bool M16>
IR_DEVINL int acc_row(int g, int lane) { return M16 ? ((g >> 3) & 1) * 16 + 4 * (lane >> 4) + (g & 3) : mfma_row(g, lane); }
template <bool M16>
IR_DEVINL int acc_col(int g, int lane) { return M16 ? ((g >> 2) & 1) * 16 + (lane & 15) : (lane & 31); }

template <int TM, int TN, int WN, int NW = 4, bool M16 = false, class MapRow>
IR_DEVINL void igemm_epilogue(const IGemmParams& p, f32x16 (&acc)[TM][TN], unsigned char* smem, int wid, int lane, int n_wave, int n0, int gn_img,
                               int gn_chunk, MapRow map_row) {
    constexpr int COLS = TN * 32;
    constexpr int LPR = COLS / 4;       // lanes per row
    constexpr int ERPI = 64 / LPR;      // rows per iteration
    constexpr int IT = 32 / ERPI;       // iterations per 32-row tile
    float* slab0 = reinterpret_cast<float*>(smem) + wid * 32 * COLS;  // tile i uses slab0 + i * SLAB_I (wave-private)
    constexpr int SLAB_I = NW * 32 * COLS;  // NW waves per workgroup
    const int r = lane & 31;
    const int ecol = (lane % LPR) * 4;  // column (within the wave tile) of this lane's 4-vector
    const int nbase = n_wave + ecol;
    const bool vec_ok = p.vec && (nbase + 3 < p.Cout);  // the fast, fully vectorised path (every layer of the network but Cout = 3)
    int mrow[TM][IT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int it = 0; it < IT; ++it) mrow[i][it] = map_row(i, it * ERPI + lane / LPR);
    // residual prefetch: all loads of the tile are issued back to back on clamped (always valid) rows BEFORE the LDS
    // transposes, so their latency overlaps the transposes instead of being paid once per row group
    f32x4 rres[TM][IT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int it = 0; it < IT; ++it) rres[i][it] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.res && vec_ok) {
        if (p.res_f32) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int m = max(mrow[i][it], 0);
                    const long rm = p.res_mod > 0 ? (long)(m % p.res_mod) : (long)m;
                    rres[i][it] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + rm * p.res_cs + nbase);
                }
        } else {
            uint2 rb[TM][IT];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int m = max(mrow[i][it], 0);
                    const long rm = p.res_mod > 0 ? (long)(m % p.res_mod) : (long)m;
                    rb[i][it] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.res) + rm * p.res_cs + nbase);
                }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int it = 0; it < IT; ++it)
                    rres[i][it] = f32x4{bflo(rb[i][it].x), bfhi(rb[i][it].x), bflo(rb[i][it].y), bfhi(rb[i][it].y)};
        }
    }
    // Bias / activation / scale / gate are per-column operations, so they are applied in the accumulator layout while the
    // tile is written to the slab (column = lane & 31: one bias and one multiplier per lane and jn); the activation is chosen
    // by a uniform switch OUTSIDE the unrolled writes. The row loop after the transpose then only adds the residual and stores
    // and exists once (with the switch inside it, the epilogue was 18 k instructions and instruction-fetch bound).
    constexpr int NCB = M16 ? 2 : 1;  // distinct columns a lane holds per tile
    float cbias[TN][NCB], cmul[TN][NCB];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
        for (int b = 0; b < NCB; ++b) {
            const int n = n_wave + jn * 32 + acc_col<M16>(b * 4, lane);
            cbias[jn][b] = (p.bias && n < p.Cout_pad) ? p.bias[n] : 0.f;
            cmul[jn][b] = p.out_scale * ((p.gate && n < p.Cout) ? p.gate[n] : 1.f);
        }
    auto write_slab = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    slab0[i * SLAB_I + acc_row<M16>(g, lane) * COLS + jn * 32 + acc_col<M16>(g, lane)] =
                        apply_act<ACT>(acc[i][jn][g] + cbias[jn][M16 ? (g >> 2) & 1 : 0], p.slope) * cmul[jn][M16 ? (g >> 2) & 1 : 0];
    };
    float gsum = 0.f, gsq = 0.f;  // fused GroupNorm statistics of this lane's 4 channels (one group) over its rows
    auto rows_fast = [&](auto gn_tag) {
        constexpr bool GN = decltype(gn_tag)::value;
        // all LDS reads of the wave's tiles are issued back to back (one latency instead of one per row group), then the stores
        f32x4 v[TM][IT];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int it = 0; it < IT; ++it)
                v[i][it] = *reinterpret_cast<const f32x4*>(&slab0[i * SLAB_I + (it * ERPI + lane / LPR) * COLS + ecol]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int m = mrow[i][it];
                if (m < 0) continue;
                const f32x4 o = v[i][it] + rres[i][it];
                const uint2 pk = make_uint2(pack2bf_valu(o[0], o[1]), pack2bf_valu(o[2], o[3]));   // o is a VALU result (LDS read + add): no MFMA hazard
                if (p.out_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (long)m * p.out_cs + nbase) = o;
                else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (long)m * p.out_cs + nbase) = pk;
                if (p.out2) *reinterpret_cast<uint2*>(p.out2 + (long)m * p.out2_cs + nbase) = pk;
                if (GN) {  // statistics of the values as stored (bf16-rounded), like the stand-alone gn_partial pass reads them
                    const float a = bflo(pk.x), b = bfhi(pk.x), c = bflo(pk.y), d = bfhi(pk.y);
                    gsum += (a + b) + (c + d);
                    gsq += (a * a + b * b) + (c * c + d * d);
                }
            }
    };
    auto rows_slow = [&](int i) {  // scalar fallback (Cout not a multiple of 4 or unaligned strides): rolled, rare, tiny tensors
        for (int it = 0; it < IT; ++it) {
            const int row = it * ERPI + lane / LPR;
            const int m = map_row(i, row);
            if (m < 0) continue;
            const long rm = p.res_mod > 0 ? (long)(m % p.res_mod) : (long)m;
            for (int e = 0; e < 4; ++e) {
                const int n = nbase + e;
                if (n >= p.Cout) break;
                float x = slab0[i * SLAB_I + row * COLS + ecol + e];
                if (p.res) x += p.res_f32 ? reinterpret_cast<const float*>(p.res)[rm * p.res_cs + n] : bf2f(reinterpret_cast<const bf16_t*>(p.res)[rm * p.res_cs + n]);
                if (p.out_f32) reinterpret_cast<float*>(p.out)[(long)m * p.out_cs + n] = x;
                else reinterpret_cast<bf16_t*>(p.out)[(long)m * p.out_cs + n] = f2bf(x);
                if (p.out2) p.out2[(long)m * p.out2_cs + n] = f2bf(x);
            }
        }
    };
    __syncthreads();  // every wave has finished reading the main-loop buffers the slabs overlay
    switch (p.act) {
        case IR_ACT_GELU_ERF: write_slab(std::integral_constant<int, IR_ACT_GELU_ERF>{}); break;
        case IR_ACT_GELU_TANH: write_slab(std::integral_constant<int, IR_ACT_GELU_TANH>{}); break;
        case IR_ACT_LRELU: write_slab(std::integral_constant<int, IR_ACT_LRELU>{}); break;
        case IR_ACT_SILU: write_slab(std::integral_constant<int, IR_ACT_SILU>{}); break;
        default: write_slab(std::integral_constant<int, IR_ACT_NONE>{}); break;
    }
    // The slabs are wave-private and a wave's LDS operations complete in issue order, so no workgroup barrier is needed between
    // the transposing writes and the reads; the fences only keep the compiler from reordering them.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (vec_ok) {
        if (p.gn_part) rows_fast(std::true_type{});
        else rows_fast(std::false_type{});
    } else if (nbase < p.Cout) {
#pragma unroll
        for (int i = 0; i < TM; ++i) rows_slow(i);
    }
    if (p.gn_part) {
        // Fixed-order block reduction (bit-identical run to run, no atomics): every lane parks its two partials at the head of its
        // wave's slab (its own slab reads completed above), then thread gl adds, in a fixed order, the lanes of every wave that
        // hold group gl: the WM waves of the column half, the ERPI row lanes and the gn_cpg/4 adjacent column lanes.
        constexpr int WM = NW / WN;
        float* red = reinterpret_cast<float*>(smem);  // [wave][lane][2] at the head of each wave's slab 0
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        slab0[lane * 2] = vec_ok ? gsum : 0.f;
        slab0[lane * 2 + 1] = vec_ok ? gsq : 0.f;
        __syncthreads();
        const int tid = wid * 64 + lane;
        const int groups = (WN * COLS) / p.gn_cpg;     // groups covered by this workgroup's BN channels
        if (tid < groups) {
            const int col0 = tid * p.gn_cpg;           // first channel of the group, relative to n0
            const int wn = col0 / COLS, cl0 = (col0 - wn * COLS) / 4;
            float a = 0.f, b = 0.f;
            for (int wmi = 0; wmi < WM; ++wmi) {
                const float* rw = red + (wmi * WN + wn) * 32 * COLS;
                for (int rl = 0; rl < ERPI; ++rl)
                    for (int cl = 0; cl < p.gn_cpg / 4; ++cl) {
                        a += rw[(rl * LPR + cl0 + cl) * 2];
                        b += rw[(rl * LPR + cl0 + cl) * 2 + 1];
                    }
            }
            const int G = p.Cout / p.gn_cpg, g = n0 / p.gn_cpg + tid;
            if (g < G) {
                float* dst = p.gn_part + ((long)gn_img * p.gn_chunks + gn_chunk) * 2 * G;
                dst[g] = a;
                dst[G + g] = b;
            }
        }
    }
}

// Block -> (row tile, column tile). Large launches: XCD-aware order - blocks b and b+8 share an XCD (and its L2); one XCD gets the n-tiles of the
// same m-tile back to back so the A tile is re-read from that L2 (the grid is padded to a multiple of 8 row tiles; padding blocks exit). Speed
// only, never correctness. SMALL launches (tile_grid: fewer than 64 row tiles, not a multiple of 8) get one block per tile in plain order: under
// the XCD order a launch with MT row tiles uses min(MT, 8) of the 8 XCDs - the ControlLDM path's 8 x 8 and 16 x 16 levels (MT = 1, 2) ran on 32
// and 64 of the 256 CUs.
IR_DEVINL bool tile_of_block(int bid, int MT, int NT, int& mt, int& nt) {
    if ((MT & 7) && (int)gridDim.x == MT * NT) {
        mt = bid / NT;
        nt = bid - mt * NT;
        return true;
    }
    const int xcd = bid & 7, j = bid >> 3;
    mt = (j / NT) * 8 + xcd;
    nt = j % NT;
    return mt < MT;
}
// NST: k-tiles resident in LDS (a ring of NST A | NST B slots). 2 is the form every large launch runs (two workgroups per CU cover each other's
// waits). NST = 4 is for the SMALL launches of the ControlLDM path (a handful of row tiles, the reduction split over blockIdx.y: <= 2 workgroups per
// CU): there every k-tile exposed the latency of its weight fetch (issued one 0.25 us tile earlier) - about 1 us per k-tile; with four tiles
// resident the fetch of tile kt+3 is issued when tile kt retires. Same accumulation order, bit-identical results.
template <int BM, int BN, int WM, int WN, int TAPS, int BK, bool M16 = false, int NST = 2>
__global__ __launch_bounds__(64 * WM * WN, NST == 2 ? 2 : 1) void igemm_kernel(IGemmParams p) {
    constexpr int NW = WM * WN;   // waves per workgroup: 4, or 8 for the small-launch form (two waves per SIMD, see launch_cfg)
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int SP = BK / 8;            // 16-byte slots per tile row
    constexpr int ROWB = BK * 2;          // bytes per tile row in LDS (unpadded: LDS-DMA writes base + lane*16)
    constexpr int RB = 256 / ROWB;        // tile rows per 256-byte LDS bank row
    constexpr int RPI = 64 / SP;          // tile rows written by one wave-wide LDS-DMA instruction
    constexpr int A_Q = BM / RPI, B_Q = BN / RPI;           // DMA instructions per A / B tile
    constexpr int A_I = (A_Q + NW - 1) / NW, B_I = (B_Q + NW - 1) / NW;  // per wave (instruction q = wave + NW*i)
    constexpr int COLS = TN * 32;
    constexpr int LDS_AB = NST * (BM + BN) * ROWB;
    static_assert(NST == 2 || (A_Q % NW == 0 && B_Q % NW == 0), "the ring's partial waits count A_I + B_I pieces per wave and tile");
    constexpr int LDS_EP = TM * NW * 32 * COLS * 4;  // one wave-private fp32 slab per 32-row tile of every wave
    constexpr int LDS_BYTES = LDS_AB > LDS_EP ? LDS_AB : LDS_EP;
    __shared__ __attribute__((aligned(256))) unsigned char smem[LDS_BYTES];  // the ONLY LDS object of the kernel
    // layout: A[0] | A[1] | B[0] | B[1]; element (row, 16-byte chunk c) of a tile lives at row*ROWB + ((c ^ swz(row)) * 16),
    // swz(row) = (row / RB) % SP: 16 consecutive rows then cover all 16 slots of a 256-byte bank row => the ds_read_b128
    // fragment reads are conflict-free although rows are unpadded. The swizzle is applied to the per-lane SOURCE address of
    // the DMA and again on the read side (never to the LDS destination, which is lane-linear by construction).

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int wm = wid / WN, wn = wid % WN;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give one XCD the n-tiles of the
    // same m-tile back to back so the A tile is re-read from that L2. Speed only, never correctness.
    const int NT = p.Cout_pad / BN;
    const int MT = (p.M + BM - 1) / BM;
    int mt, nt;
    if (!tile_of_block(blockIdx.x, MT, NT, mt, nt)) return;
    const int m0 = mt * BM, n0 = nt * BN;

    const int cchunks = p.Cin / BK;
    // split-K (p.ksplit > 1, blockIdx.y = split): this workgroup reduces k-tiles [kbeg, kbeg + KT) and stores its partial tile into slice
    // blockIdx.y of the fp32 workspace p.out ([ksplit][M][Cout_pad]; splitk_finish_kernel adds the slices in order: deterministic). Small-M
    // launches (the UNet's 8 x 8 and 16 x 16 levels) otherwise leave most CUs idle behind 10-40 workgroups that stream 30-60 MB of weights
    int kbeg = 0, KT = TAPS * cchunks;
    if (p.ksplit > 1) {
        const int per = (KT + p.ksplit - 1) / p.ksplit;   // the launcher chose ksplit so that no split is empty
        kbeg = (int)blockIdx.y * per;
        KT = min(per, KT - kbeg);
        p.out = reinterpret_cast<float*>(p.out) + (long)blockIdx.y * p.M * p.Cout_pad;
    }

    // ---- per-lane DMA sources. Every lane always issues its loads on a valid address: rows beyond M re-read row M-1 (never
    // stored) and zero-padding taps read the zero page.
    const int lrow = lane / SP, lslot = lane % SP;
    const bf16_t* a_ptr[A_I];
    int a_n[A_I], a_oy[A_I], a_ox[A_I], a_sw[A_I];
#pragma unroll
    for (int i = 0; i < A_I; ++i) {
        const int row = min((wid + NW * i) * RPI + lrow, BM - 1);
        const int m = min(m0 + row, p.M - 1);
        a_sw[i] = (lslot ^ ((row / RB) % SP)) * 8;
        if (TAPS == 1) {
            a_n[i] = a_oy[i] = a_ox[i] = 0;
            a_ptr[i] = p.in + (long)m * p.in_cs + a_sw[i] + (long)kbeg * BK;
        } else {
            const int hw = p.Ho * p.Wo;
            const int n = m / hw, rem = m - n * hw;
            a_n[i] = n; a_oy[i] = rem / p.Wo; a_ox[i] = rem - a_oy[i] * p.Wo;
            a_ptr[i] = p.in;
        }
    }
    const int Hc = p.up ? 2 * p.H : p.H, Wc = p.up ? 2 * p.W : p.W;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
    auto set_tap = [&](int tap) {  // TAPS == 9 only: source pixel (or the zero page) of every owned row for this tap
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int i = 0; i < A_I; ++i) {
            const int cy = a_oy[i] * p.stride + ky - p.pad, cx = a_ox[i] * p.stride + kx - p.pad;
            const bool ok = cy >= 0 && cy < Hc && cx >= 0 && cx < Wc;
            const int iy = min(max(cy, 0), Hc - 1) >> p.up, ix = min(max(cx, 0), Wc - 1) >> p.up;
            const bf16_t* src = p.in + (((long)a_n[i] * p.H + iy) * p.W + ix) * p.in_cs;
            a_ptr[i] = (ok ? src : zero) + a_sw[i];
        }
    };
    const bf16_t* b_ptr[B_I];
#pragma unroll
    for (int i = 0; i < B_I; ++i) {
        const int row = min((wid + NW * i) * RPI + lrow, BN - 1);
        b_ptr[i] = p.wgt + (long)(n0 + row) * p.wgt_rs + (lslot ^ ((row / RB) % SP)) * 8 + (long)kbeg * BK;
    }
    int cc = TAPS > 1 ? kbeg % cchunks : 0, tap = TAPS > 1 ? kbeg / cchunks : 0;
    auto stage = [&](int buf) {  // asynchronous global -> LDS copy of the k-tile the pointers address; then advance them
#pragma unroll
        for (int i = 0; i < A_I; ++i) {
            const int q = wu + NW * i;
            if (A_Q % NW == 0 || q < A_Q)
                glds16(a_ptr[i], (lds_ptr_t)(smem + buf * BM * ROWB + q * RPI * ROWB));
            a_ptr[i] += BK;
        }
#pragma unroll
        for (int i = 0; i < B_I; ++i) {
            const int q = wu + NW * i;
            if (B_Q % NW == 0 || q < B_Q)
                glds16(b_ptr[i], (lds_ptr_t)(smem + NST * BM * ROWB + buf * BN * ROWB + q * RPI * ROWB));
            b_ptr[i] += BK;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    // fragment read addresses (bytes): row base and the row's swizzle
    int fa_base[TM], fa_sw[TM], fb_base[TN], fb_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int R = wm * (BM / WM) + i * 32 + r;
        fa_base[i] = R * ROWB; fa_sw[i] = (R / RB) % SP;
    }
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int R = wn * (BN / WN) + jn * 32 + r;
        fb_base[jn] = NST * BM * ROWB + R * ROWB; fb_sw[jn] = (R / RB) % SP;
    }

    // Main loop. Two k-tiles are resident (double-buffered LDS) and the fragments are double-buffered in registers. Per k-step the
    // order is pinned with sched_barriers (hipcc otherwise sinks every LDS read below the MFMAs and waits lgkmcnt(0) four times
    // per k-tile with nothing in flight): first MFMA of the step, then the reads of the NEXT step, then the remaining MFMAs, so a
    // read always has >= 3 MFMAs (96 cycles) of cover. The workgroup barrier sits inside the last k-step of a tile, after its first
    // MFMA: behind it the DMA of tile kt+2 is issued into the buffer just released and the first fragments of tile kt+1 are read
    // while the last MFMAs of tile kt still issue.
    constexpr int NK = BK / 16;
    auto advance = [&]() { if (TAPS > 1 && ++cc == cchunks) { cc = 0; set_tap(++tap); } };
    if (TAPS > 1) {
        set_tap(tap);
#pragma unroll
        for (int i = 0; i < A_I; ++i) a_ptr[i] += cc * BK;
    }
    stage(0);
#pragma unroll
    for (int st = 1; st < NST; ++st)
        if (KT > st) { advance(); stage(st); }
    // ring form: wait until tile kt + 1 has landed while the younger tiles (at most NST - 2 of them, A_I + B_I pieces per wave each, in order) stay in flight
    auto wait_younger = [&](int after) {   // returns when at most `after` of this wave's youngest tiles are still in flight
        constexpr int PW = A_I + B_I;
        if (NST == 2 || after <= 0) wait_dma();
        else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
        else if (after == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PW) : "memory");
    };
    // at the end of tile kt the tiles up to kt + NST - 1 have been issued; tile kt + 1 is needed, the younger ones may fly
    auto wait_next = [&](int kt) { wait_younger(min(KT - 1, kt + NST - 1) - (kt + 1)); };
    static_assert(NST == 2 || NST == 4, "wait_younger covers up to three younger tiles");
    static_assert(3 * (A_I + B_I) < 64, "vmcnt is a 6-bit counter");
    wait_younger(min(KT, NST) - 1);
    __syncthreads();  // tile 0 (NST == 2: both tiles) landed and published to all waves
    if constexpr (M16) {
        // v_mfma_f32_16x16x32_bf16 form of the same loop (BK = 64 = two k-steps of 32): a 32-row tile is two 16-row fragments, lane l
        // holds row (l & 15) and the 8 k-values of 16-byte chunk 4*ks + (l >> 4). Same LDS image, same bytes read per FLOP, same
        // MFMA cycles; the CDNA4 notes report a higher sustained clock for this shape under load.
        static_assert(BK == 64, "two k-steps of 32");
        typedef __attribute__((ext_vector_type(4))) float f32x4_t;
        const int r16 = lane & 15, kq = lane >> 4;
        int a_off[TM][2], a_swz[TM][2], b_off[TN][2], b_swz[TN][2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int R = wm * (BM / WM) + i * 32 + a * 16 + r16;
                a_off[i][a] = R * ROWB; a_swz[i][a] = (R / RB) % SP;
            }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int R = wn * (BN / WN) + jn * 32 + b * 16 + r16;
                b_off[jn][b] = NST * BM * ROWB + R * ROWB; b_swz[jn][b] = (R / RB) % SP;
            }
        f32x4_t c16[TM][TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int t = 0; t < 4; ++t) c16[i][jn][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        bf16x8 fa[2][TM][2], fb[2][TN][2];
        auto load16 = [&](int buf, int ks, int set) {
            const unsigned char* Ab = smem + buf * BM * ROWB;
            const unsigned char* Bb = smem + buf * BN * ROWB;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int a = 0; a < 2; ++a) fa[set][i][a] = *reinterpret_cast<const bf16x8*>(Ab + a_off[i][a] + (((4 * ks + kq) ^ a_swz[i][a]) << 4));
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int b = 0; b < 2; ++b) fb[set][jn][b] = *reinterpret_cast<const bf16x8*>(Bb + b_off[jn][b] + (((4 * ks + kq) ^ b_swz[jn][b]) << 4));
        };
        auto mfma16s = [&](int set, int first, int last) {  // e = ((i*TN + jn)*2 + a)*2 + b
#pragma unroll
            for (int e = first; e < last; ++e) {
                const int b = e & 1, a = (e >> 1) & 1, jn = (e >> 2) % TN, i = (e >> 2) / TN;
                c16[i][jn][a * 2 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[set][i][a], fb[set][jn][b], c16[i][jn][a * 2 + b], 0, 0, 0);
            }
        };
        load16(0, 0, 0);
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & (NST - 1), nxt = (kt + 1) & (NST - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                __builtin_amdgcn_sched_barrier(0);
                mfma16s(ks, 0, 2);
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 0) {
                    load16(cur, 1, 1);
                } else {
                    wait_next(kt);
                    __syncthreads();
                    if (kt + NST < KT) { advance(); stage(cur); }
                    if (kt + 1 < KT) load16(nxt, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma16s(ks, 2, TM * TN * 4);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[i][jn][g] = c16[i][jn][g >> 2][g & 3];
        const int gn_hw16 = !p.gn_part ? 1 : (p.taps == 9 ? p.Ho * p.Wo : p.M / p.NB);
        igemm_epilogue<TM, TN, WN, NW, true>(p, acc, smem, wid, lane, n0 + wn * (BN / WN), n0, m0 / gn_hw16, (m0 % gn_hw16) / BM, [&](int i, int row) {
            const int m = m0 + wm * (BM / WM) + i * 32 + row;
            return m < p.M ? m : -1;
        });
        return;
    }
    bf16x8 af[2][TM], bfr[2][TN];
    auto load_frags = [&](int buf, int ks, int set) {
        const unsigned char* Ab = smem + buf * BM * ROWB;
        const unsigned char* Bb = smem + buf * BN * ROWB;
#pragma unroll
        for (int i = 0; i < TM; ++i)
            af[set][i] = *reinterpret_cast<const bf16x8*>(Ab + fa_base[i] + (((2 * ks + h) ^ fa_sw[i]) << 4));
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
            bfr[set][jn] = *reinterpret_cast<const bf16x8*>(Bb + fb_base[jn] + (((2 * ks + h) ^ fb_sw[jn]) << 4));
    };
    auto mfmas = [&](int set, int first, int last) {
#pragma unroll
        for (int e = first; e < last; ++e) acc[e / TN][e % TN] = mfma32(af[set][e / TN], bfr[set][e % TN], acc[e / TN][e % TN]);
    };
    load_frags(0, 0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & (NST - 1), nxt = (kt + 1) & (NST - 1);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            __builtin_amdgcn_sched_barrier(0);
            mfmas(ks & 1, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < NK) {
                load_frags(cur, ks + 1, (ks + 1) & 1);
            } else {
                wait_next(kt);
                __syncthreads();  // tile kt+1 has landed (issued a whole tile ago); every wave has read the last fragment of tile kt
                if (kt + NST < KT) { advance(); stage(cur); }
                if (kt + 1 < KT) load_frags(nxt, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas(ks & 1, 1, TM * TN);
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    const int gn_hw = !p.gn_part ? 1 : (p.taps == 9 ? p.Ho * p.Wo : p.M / p.NB);  // rows per image (fused GroupNorm statistics only)
    igemm_epilogue<TM, TN, WN, NW>(p, acc, smem, wid, lane, n0 + wn * (BN / WN), n0, m0 / gn_hw, (m0 % gn_hw) / BM, [&](int i, int row) {
        const int m = m0 + wm * (BM / WM) + i * 32 + row;
        return m < p.M ? m : -1;
    });
}

// ---------------------------------------------------------------------------------------------------------------------
// Halo-tile 3x3 convolution (stride 1, optional nearest-2x upsample): one block computes an 8 x 16 patch of output pixels
// for BN output channels. Per 64-channel chunk the 10 x 18 input halo of the patch is brought into LDS ONCE by LDS-DMA and
// all nine taps read their A fragments from it with a tap offset on the LDS address; only the weight tile is re-staged per
// tap. Compared with the generic implicit GEMM above this cuts the activation traffic from L2/HBM ~6x (180 instead of
// 9 x 128 pixel rows per chunk) and removes all per-tap global address arithmetic.
// Swizzle key of a halo pixel (a 128-byte LDS row of 8 chunks; chunk c of the pixel in halo column hx is stored at slot
// c ^ halo_key(hx), on the LDS-DMA source side and on the read side). The key follows the fragment read pattern:
//  * 32x32x16 MFMAs: the 16 lanes of a ds_read_b128 lane group read the SAME chunk of 16 pixels -> 8 distinct keys per column pair.
//  * 16x16x32 MFMAs: a lane group reads all 16 columns of one patch row once, columns 4-11 with chunk 4ks + kq and columns 0-3, 12-15
//    with 4ks + (kq ^ 1) (or the other way round), shifted by the tap's kx. With key = 2 * ((hx >> 1) & 3) the two pixels that share
//    key bits (columns hx and hx + 8) always sit in opposite chunk sets, whose chunks differ in bit 0, for every kx: conflict-free.
//    The 32x32 key under this read pattern conflicts for kx = 1, 2 (SQ_LDS_BANK_CONFLICT was 27 % of SQ_LDS_IDX_ACTIVE).
template <bool M16>
IR_DEVINL int halo_key(int hx) { return M16 ? ((hx >> 1) & 3) << 1 : (hx >> 1) & 7; }

// PH (16x16x32 form only): the sub-pixel phase form of "nearest-2x upsample + 3x3" (see conv_s1.hip): four 2x2 convs on the low-resolution tensor,
// p.wgt = [phase][Cout_pad][2x2][Cin] (weights.pack_conv_up2x2), tile index = 4 * patch + phase, output pixel (2y + dy, 2x + dx); the halo is the
// 9-tap one shifted by (dy, dx) (one row / column of it unused). SwinIR's three 64-channel upsampler convs (swinir.py:880-886).
template <int BN, int UP, bool M16 = false, bool FP8 = false, bool PH = false>
__global__ __launch_bounds__(256, 2) void conv_halo_kernel(IGemmParams p, int tiles_y, int tiles_x) {
    static_assert(!PH || (M16 && !FP8 && UP == 0), "phase form: 16x16x32 bf16 path on the low-resolution grid");
    constexpr int NTAP = PH ? 4 : 9;
    constexpr int BK = 64, ROWB = 128, SP = 8;
    constexpr int TH = 8, TW = 16, HW = TW + 2, HP = (TH + 2) * HW;  // 180 halo pixels
    constexpr int H_Q = (HP + 7) / 8;                                 // 23 DMA instructions (8 pixels x 8 slots each)
    constexpr int HPAD = H_Q * 8;                                     // 184 rows in LDS
    constexpr int H_I = (H_Q + 3) / 4;
    constexpr int B_Q = BN / 8, B_I = B_Q / 4;
    constexpr int TM = 2, TN = BN / 64;
    constexpr int HALO_BYTES = HPAD * ROWB, BT_BYTES = BN * ROWB;
    constexpr int LDS_MAIN = 2 * HALO_BYTES + 2 * BT_BYTES;
    constexpr int LDS_EP = TM * 4 * 32 * TN * 32 * 4;
    constexpr int LDS_BYTES = LDS_MAIN > LDS_EP ? LDS_MAIN : LDS_EP;
    __shared__ __attribute__((aligned(256))) unsigned char smem[LDS_BYTES];  // halo[0] | halo[1] | B[0] | B[1]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int wm = wid >> 1, wn = wid & 1;
    const int r = lane & 31, h = lane >> 5;

    const int NT = p.Cout_pad / BN;
    const int MT = p.NB * tiles_y * tiles_x * (PH ? 4 : 1);
    int mt, nt;
    if (!tile_of_block(blockIdx.x, MT, NT, mt, nt)) return;
    const int n0 = nt * BN;
    const int per_img = tiles_y * tiles_x * (PH ? 4 : 1);
    const int img = mt / per_img, trem_ph = mt - img * per_img;
    const int phase = PH ? (trem_ph & 3) : 0, trem = PH ? (trem_ph >> 2) : trem_ph;
    const int dy = phase >> 1, dx = phase & 1;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int Hc = UP ? 2 * p.H : p.H, Wc = UP ? 2 * p.W : p.W;  // conv-input (== output; PH: low-resolution) extent

    const int chunks = p.Cin / BK;
    const int lrow = lane >> 3, lslot = lane & 7;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
    // halo DMA sources: instruction q covers halo pixels 8q .. 8q+7
    const bf16_t* h_ptr[H_I];
#pragma unroll
    for (int i = 0; i < H_I; ++i) {
        const int hp = (wid + 4 * i) * 8 + lrow;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int cy = oy0 + hy - 1 + dy, cx = ox0 + hx - 1 + dx;
        const bool ok = hp < HP && cy >= 0 && cy < Hc && cx >= 0 && cx < Wc;
        const int iy = min(max(cy, 0), Hc - 1) >> UP, ix = min(max(cx, 0), Wc - 1) >> UP;
        const bf16_t* src = p.in + (((long)img * p.H + iy) * p.W + ix) * p.in_cs;
        h_ptr[i] = (ok ? src : zero) + ((lslot ^ halo_key<M16>(hx)) << 3);  // swizzle key from the halo COLUMN: see halo_key
    }
    const bf16_t* b_ptr[B_I];
#pragma unroll
    for (int i = 0; i < B_I; ++i) {
        const int row = (wid + 4 * i) * 8 + lrow;
        b_ptr[i] = p.wgt + (long)(phase * p.Cout_pad + n0 + row) * p.wgt_rs + ((lslot ^ ((row >> 1) & 7)) << 3);
    }
    auto stage_halo = [&](int buf) {  // the chunk the pointers address; then advance to the next chunk
#pragma unroll
        for (int i = 0; i < H_I; ++i) {
            const int q = wu + 4 * i;
            if (q < H_Q) glds16(h_ptr[i], (lds_ptr_t)(smem + buf * HALO_BYTES + q * 8 * ROWB));
            h_ptr[i] += BK;
        }
    };
    auto stage_b = [&](int buf, int koff) {  // weight tile of (tap, chunk): k offset koff = tap*Cin + chunk*64
#pragma unroll
        for (int i = 0; i < B_I; ++i)
            glds16(b_ptr[i] + koff, (lds_ptr_t)(smem + 2 * HALO_BYTES + buf * BT_BYTES + (wu + 4 * i) * 8 * ROWB));
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    // this lane's output pixels: tile i covers patch rows wm*4 + 2i, +1 (16 pixels each)
    int hid0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) hid0[i] = (wm * 4 + i * 2 + (r >> 4)) * HW + (r & 15);
    int fb_base[TN], fb_sw[TN];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int R = wn * (BN / 2) + jn * 32 + r;
        fb_base[jn] = R * ROWB; fb_sw[jn] = (R >> 1) & 7;
    }

    // Main loop over steps s = (chunk c, tap t); same pinned schedule as igemm_kernel: per k-step first MFMA, reads of the next
    // k-step, remaining MFMAs; the workgroup barrier sits inside the last k-step, and behind it the weight tile of step s+2 (and,
    // at the first tap of a chunk, the whole halo of the next chunk) is issued and the first fragments of step s+1 are read.
    const int steps = chunks * NTAP;
    IR_STAMP(0);
    stage_halo(0);
    stage_b(0, 0);
    stage_b(1, p.Cin);  // step 1 = (chunk 0, tap 1); steps >= 4 always
    wait_dma();
    __syncthreads();
    IR_STAMP(1);
    if constexpr (FP8) {
        // fp8 (OCP e4m3) operands through the MX-scaled MFMA with unit block scales, v_mfma_scale_f32_32x32x64_f8f6f4: twice the
        // FLOPs per byte staged and per MFMA cycle of the bf16 forms (BASELINE.json configs[4]). Everything above addresses memory in
        // 2-byte units, so a "channel" up there is a PAIR of fp8 channels: the launcher passes Cin = Cin_fp8 / 2 (and the strides
        // likewise), a 64-unit chunk is 128 fp8 channels in the same 128-byte LDS rows, and staging, swizzles and the bank-conflict
        // analysis of the 32x32 read pattern carry over unchanged. Lane (r, h) holds row r and k = 32h .. 32h+31 of a 64-wide k-step
        // (tools/mfma_fp8_probe.hip): the 16-byte chunks 4ks + 2h and 4ks + 2h + 1 of the row. Dequantisation (per-output-channel
        // weight scale x the activation scale) is the epilogue's per-column multiplier p.gate, the bias arrives pre-divided by it.
        typedef __attribute__((ext_vector_type(8))) int i32x8_t;
        typedef __attribute__((ext_vector_type(4))) int i32x4_t;
        i32x8_t fa[2][TM], fb[2][TN];
        auto load8 = [&](int cbuf, int bbuf, int tap, int ks, int set) {
            const unsigned char* Hb = smem + cbuf * HALO_BYTES;
            const unsigned char* Bb = smem + 2 * HALO_BYTES + bbuf * BT_BYTES;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int toff = ky * HW + kx;
            const int sw = halo_key<false>((r & 15) + kx);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned char* row = Hb + (hid0[i] + toff) * ROWB;
                const i32x4_t lo = *reinterpret_cast<const i32x4_t*>(row + (((4 * ks + 2 * h) ^ sw) << 4));
                const i32x4_t hi = *reinterpret_cast<const i32x4_t*>(row + (((4 * ks + 2 * h + 1) ^ sw) << 4));
                fa[set][i] = i32x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                const unsigned char* row = Bb + fb_base[jn];
                const i32x4_t lo = *reinterpret_cast<const i32x4_t*>(row + (((4 * ks + 2 * h) ^ fb_sw[jn]) << 4));
                const i32x4_t hi = *reinterpret_cast<const i32x4_t*>(row + (((4 * ks + 2 * h + 1) ^ fb_sw[jn]) << 4));
                fb[set][jn] = i32x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        };
        auto mfma8 = [&](int set, int first, int last) {
#pragma unroll
            for (int e = first; e < last; ++e)
                acc[e / TN][e % TN] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[set][e / TN], fb[set][e % TN], acc[e / TN][e % TN], 0, 0, 0,
                                                                                      0x7F7F7F7F, 0, 0x7F7F7F7F);
        };
        int c = 0, t = 0, c2 = 0, t2 = 2;
        load8(0, 0, 0, 0, 0);
        for (int s = 0; s < steps; ++s) {
            int tn = t + 1, cn = c;
            if (tn == 9) { tn = 0; cn = c + 1; }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                __builtin_amdgcn_sched_barrier(0);
                mfma8(ks, 0, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 0) {
                    load8(c & 1, s & 1, t, 1, 1);
                } else {
                    wait_dma();
                    __syncthreads();
                    if (s + 2 < steps) stage_b(s & 1, t2 * p.Cin + c2 * BK);
                    if (t == 0 && c + 1 < chunks) stage_halo((c + 1) & 1);
                    if (s + 1 < steps) load8(cn & 1, (s + 1) & 1, tn, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma8(ks, 1, TM * TN);
            }
            t = tn; c = cn;
            if (++t2 == 9) { t2 = 0; ++c2; }
        }
        __builtin_amdgcn_sched_barrier(0);
        igemm_epilogue<TM, TN, 2>(p, acc, smem, wid, lane, n0 + wn * (BN / 2), n0, img, trem, [&](int i, int row) {
            const int oy = oy0 + wm * 4 + i * 2 + (row >> 4), ox = ox0 + (row & 15);
            return (oy < p.Ho && ox < p.Wo) ? (img * p.Ho + oy) * p.Wo + ox : -1;
        });
        return;
    }
    if constexpr (M16) {
        // v_mfma_f32_16x16x32_bf16 form (see igemm_kernel): a 16-row fragment is ONE patch row (16 pixels), lane l holds pixel (l & 15)
        // and 16-byte chunk 4*ks + (l >> 4) of its 64 channels; two k-steps of 32 per (tap, chunk) step.
        typedef __attribute__((ext_vector_type(4))) float f32x4_t;
        const int c16 = lane & 15, kq = lane >> 4;
        int b_off[TN][2], b_swz[TN][2];
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int R = wn * (BN / 2) + jn * 32 + b * 16 + c16;
                b_off[jn][b] = R * ROWB; b_swz[jn][b] = (R >> 1) & 7;
            }
        f32x4_t c4[TM][TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) c4[i][jn][t4] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        bf16x8 fa[2][TM][2], fb[2][TN][2];
        auto load16 = [&](int cbuf, int bbuf, int tap, int ks, int set) {
            const unsigned char* Hb = smem + cbuf * HALO_BYTES;
            const unsigned char* Bb = smem + 2 * HALO_BYTES + bbuf * BT_BYTES;
            const int ky = PH ? tap >> 1 : tap / 3, kx = PH ? tap & 1 : tap - ky * 3;
            const int sw = halo_key<true>(c16 + kx);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int hid = (wm * 4 + i * 2 + a + ky) * HW + c16 + kx;
                    fa[set][i][a] = *reinterpret_cast<const bf16x8*>(Hb + hid * ROWB + (((4 * ks + kq) ^ sw) << 4));
                }
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int b = 0; b < 2; ++b) fb[set][jn][b] = *reinterpret_cast<const bf16x8*>(Bb + b_off[jn][b] + (((4 * ks + kq) ^ b_swz[jn][b]) << 4));
        };
        auto mfma16s = [&](int set, int first, int last) {  // e = ((i*TN + jn)*2 + a)*2 + b
#pragma unroll
            for (int e = first; e < last; ++e) {
                const int b = e & 1, a = (e >> 1) & 1, jn = (e >> 2) % TN, i = (e >> 2) / TN;
                c4[i][jn][a * 2 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[set][i][a], fb[set][jn][b], c4[i][jn][a * 2 + b], 0, 0, 0);
            }
        };
        int c = 0, t = 0, c2 = 0, t2 = 2;
        load16(0, 0, 0, 0, 0);
        for (int s = 0; s < steps; ++s) {
            int tn = t + 1, cn = c;
            if (tn == NTAP) { tn = 0; cn = c + 1; }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                __builtin_amdgcn_sched_barrier(0);
                mfma16s(ks, 0, 2);
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 0) {
                    load16(c & 1, s & 1, t, 1, 1);
                } else {
                    wait_dma();
                    __syncthreads();
                    if (s + 2 < steps) stage_b(s & 1, t2 * p.Cin + c2 * BK);
                    if (t == 0 && c + 1 < chunks) stage_halo((c + 1) & 1);
                    if (s + 1 < steps) load16(cn & 1, (s + 1) & 1, tn, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma16s(ks, 2, TM * TN * 4);
            }
            t = tn; c = cn;
            if (++t2 == NTAP) { t2 = 0; ++c2; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[i][jn][g] = c4[i][jn][g >> 2][g & 3];
        igemm_epilogue<TM, TN, 2, 4, true>(p, acc, smem, wid, lane, n0 + wn * (BN / 2), n0, img, trem, [&](int i, int row) {
            const int oy = oy0 + wm * 4 + i * 2 + (row >> 4), ox = ox0 + (row & 15);
            if constexpr (PH) return (oy < p.H && ox < p.W) ? (img * p.Ho + 2 * oy + dy) * p.Wo + 2 * ox + dx : -1;
            return (oy < p.Ho && ox < p.Wo) ? (img * p.Ho + oy) * p.Wo + ox : -1;
        });
        return;
    }
    bf16x8 af[2][TM], bfr[2][TN];
    auto load_frags = [&](int cbuf, int bbuf, int tap, int ks, int set) {
        const unsigned char* Hb = smem + cbuf * HALO_BYTES;
        const unsigned char* Bb = smem + 2 * HALO_BYTES + bbuf * BT_BYTES;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int toff = ky * HW + kx;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // A ds_read_b128 is served in groups of 16 lanes that span two patch rows (8 pixels of row y, 8 of row y+1): keyed on
            // the halo column alone, the 16 slots are distinct; keyed on the pixel index (18 per row) they collided 2-way
            // (SQ_LDS_BANK_CONFLICT was 30 % of the LDS cycles).
            const int hid = hid0[i] + toff;
            const int sw = halo_key<false>((r & 15) + kx);
            af[set][i] = *reinterpret_cast<const bf16x8*>(Hb + hid * ROWB + (((2 * ks + h) ^ sw) << 4));
        }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) bfr[set][jn] = *reinterpret_cast<const bf16x8*>(Bb + fb_base[jn] + (((2 * ks + h) ^ fb_sw[jn]) << 4));
    };
    auto mfmas = [&](int set, int first, int last) {
#pragma unroll
        for (int e = first; e < last; ++e) acc[e / TN][e % TN] = mfma32(af[set][e / TN], bfr[set][e % TN], acc[e / TN][e % TN]);
    };
    int c = 0, t = 0;      // (chunk, tap) of step s
    int c2 = 0, t2 = 2;    // (chunk, tap) of step s + 2
    load_frags(0, 0, 0, 0, 0);
    for (int s = 0; s < steps; ++s) {
        int tn = t + 1, cn = c;  // step s + 1
        if (tn == 9) { tn = 0; cn = c + 1; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            __builtin_amdgcn_sched_barrier(0);
            if (IR_KO != 4) mfmas(ks & 1, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < 4) {
                if (IR_KO != 3) load_frags(c & 1, s & 1, t, ks + 1, (ks + 1) & 1);
            } else {
                if (IR_KO != 1) {
                    wait_dma();
                    __syncthreads();  // step s+1's weights have landed; every wave has read the last fragments of step s
                }
                if (s + 2 < steps && IR_KO != 2) stage_b(s & 1, t2 * p.Cin + c2 * BK);
                if (t == 0 && c + 1 < chunks) stage_halo((c + 1) & 1);  // whole next chunk's halo, eight steps ahead of its use
                if (s + 1 < steps && IR_KO != 3) load_frags(cn & 1, (s + 1) & 1, tn, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (IR_KO != 4) mfmas(ks & 1, 1, TM * TN);
        }
        t = tn; c = cn;
        if (++t2 == 9) { t2 = 0; ++c2; }
    }
    __builtin_amdgcn_sched_barrier(0);
    IR_STAMP(2);
    igemm_epilogue<TM, TN, 2>(p, acc, smem, wid, lane, n0 + wn * (BN / 2), n0, img, trem, [&](int i, int row) {
        const int oy = oy0 + wm * 4 + i * 2 + (row >> 4), ox = ox0 + (row & 15);
        return (oy < p.Ho && ox < p.Wo) ? (img * p.Ho + oy) * p.Wo + ox : -1;
    });
    IR_STAMP(3);
}

// ---------------------------------------------------------------------------------------------------------------------
// Ping-pong halo-tile convolution: the same mathematics, LDS images and epilogue as conv_halo_kernel, as ONE 512-thread
// workgroup per CU that computes a 16 x 16 patch x 128 output channels. Two independent 4-wave workgroups per CU run in lockstep
// (they stage together and multiply together), so their MFMA time and their LDS-DMA issue + LDS time add up (knock-outs: the data
// movement alone is 8.7 us of a 14.3 us main loop). Here waves w and w+4 share a SIMD, waves 0-3 own patch rows 0-7 and waves
// 4-7 rows 8-15, and workgroup barriers keep the halves in complementary segments:
//     segment:    2s            2s+1            2s+2           2s+3
//     waves 0-3   matrix(s)     vector(s)       matrix(s+1)    vector(s+1)
//     waves 4-7   vector(s-1)   matrix(s)       vector(s)      matrix(s+1)
// matrix(s) = the 16 MFMAs of step s = (chunk, tap) with their fragment reads; vector(s) = this wave's LDS-DMA pieces (2 of a weight
// tile, at most 1 of the next chunk's halo) and the first fragment reads of step s+1, issued above the barrier. Both halves share
// every weight tile (half the weight traffic per MFMA of the 4-wave kernel) and the 18 x 18 halo. Weight tiles live in a ring of 3:
// waves 0-3 issue their pieces of step s+2 in vector(s), waves 4-7 (a segment later) theirs of step s+3, each into the slot whose
// previous tile both halves finished reading at an earlier barrier; every wave waits for its own pieces (vmcnt) at the end of its
// next matrix segment, and the barrier there publishes them at least a segment before their first read.
template <int UP, bool M16 = false>
__global__ __launch_bounds__(512, 1) void conv_halo_pp_kernel(IGemmParams p, int tiles_y, int tiles_x) {
    constexpr int BN = 128, BK = 64, ROWB = 128;
    constexpr int TH = 16, TW = 16, HW = TW + 2, HP = (TH + 2) * HW;  // 324 halo pixels
    constexpr int H_Q = (HP + 7) / 8;                                 // 41 DMA instructions (8 pixels x 8 slots each)
    constexpr int HPAD = H_Q * 8;
    constexpr int H_I = (H_Q + 3) / 4;                                // halo pieces per wave of waves 4-7 (11)
    constexpr int B_I = 3;                                            // weight pieces per tile: waves 0-3 three each, waves 4-7 one each
    constexpr int NSB = 3;                                            // weight-tile ring
    constexpr int TM = 2, TN = 2;
    constexpr int HALO_BYTES = HPAD * ROWB, BT_BYTES = BN * ROWB;
    constexpr int LDS_MAIN = 2 * HALO_BYTES + NSB * BT_BYTES;
    constexpr int LDS_EP = TM * 8 * 32 * TN * 32 * 4;
    constexpr int LDS_BYTES = LDS_MAIN > LDS_EP ? LDS_MAIN : LDS_EP;
    __shared__ __attribute__((aligned(256))) unsigned char smem[LDS_BYTES];  // halo[0] | halo[1] | B[0] | B[1] | B[2]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int grp = wu >> 2;                 // waves 4-7 run one segment behind waves 0-3
    const int wm = (wid >> 1) & 1, wn = wid & 1;
    const int r = lane & 31, h = lane >> 5;

    const int NT = p.Cout_pad / BN;
    const int MT = p.NB * tiles_y * tiles_x;
    int mt, nt;
    if (!tile_of_block(blockIdx.x, MT, NT, mt, nt)) return;
    const int n0 = nt * BN;
    const int img = mt / (tiles_y * tiles_x), trem = mt - img * tiles_y * tiles_x;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int Hc = UP ? 2 * p.H : p.H, Wc = UP ? 2 * p.W : p.W;  // conv-input (== output) extent
    const int chunks = p.Cin / BK;
    const int steps = chunks * 9;
    const int lrow = lane >> 3, lslot = lane & 7;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
    // Division of the LDS-DMA work (20.5 one-KB pieces per step and workgroup, about 100 cycles of issue each): waves 0-3 bring 12 of
    // the 16 weight pieces (3 per wave and step, L2 hits); waves 4-7 the other 4 (1 per wave) and ALL halo pieces (11 per wave and
    // chunk, HBM misses; 2,2,2,1,1,1,1,1,0 over the nine steps). vmcnt completes in order, so in a vector segment waves 4-7 issue
    // their weight piece FIRST and wait for it a segment later with vmcnt(number of halo pieces issued behind it): an HBM-latency
    // halo piece then has two segments before anything waits for it, and waves 0-3 only ever wait for L2 hits.
    const int wq = wu & 3;
    // halo DMA sources: piece q = (wave & 3) + 4*i covers halo pixels 8q .. 8q+7
    const bf16_t* h_ptr[H_I];
#pragma unroll
    for (int i = 0; i < H_I; ++i) {
        const int hp = ((wid & 3) + 4 * i) * 8 + lrow;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int cy = oy0 + hy - 1, cx = ox0 + hx - 1;
        const bool ok = hp < HP && cy >= 0 && cy < Hc && cx >= 0 && cx < Wc;
        const int iy = min(max(cy, 0), Hc - 1) >> UP, ix = min(max(cx, 0), Wc - 1) >> UP;
        const bf16_t* src = p.in + (((long)img * p.H + iy) * p.W + ix) * p.in_cs;
        h_ptr[i] = (ok ? src : zero) + ((lslot ^ halo_key<M16>(hx)) << 3);  // swizzle key from the halo COLUMN (see halo_key)
    }
    const bf16_t* b_ptr[B_I];
#pragma unroll
    for (int i = 0; i < B_I; ++i) {
        const int q = wid < 4 ? (wid & 3) + 4 * i : 12 + (wid & 3);  // waves 4-7 use only i = 0
        const int row = q * 8 + lrow;
        b_ptr[i] = p.wgt + (long)(n0 + row) * p.wgt_rs + ((lslot ^ ((row >> 1) & 7)) << 3);
    }
    auto halo_piece = [&](int i, int chunk) {  // piece i of this wave, channels chunk*64.. into halo buffer chunk & 1
        const int q = wq + 4 * i;
        if (q < H_Q) glds16(h_ptr[i] + chunk * BK, (lds_ptr_t)(smem + (chunk & 1) * HALO_BYTES + q * 8 * ROWB));
    };
    auto stage_b = [&](int step) {  // this wave's pieces of the weight tile of `step` into ring slot step % 3
        const int c = step / 9, t = step - c * 9;
        const int koff = t * p.Cin + c * BK;
        unsigned char* dst = smem + 2 * HALO_BYTES + (step % NSB) * BT_BYTES;
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < B_I; ++i) glds16(b_ptr[i] + koff, (lds_ptr_t)(dst + (wq + 4 * i) * 8 * ROWB));
        } else {
            glds16(b_ptr[0] + koff, (lds_ptr_t)(dst + (12 + wq) * 8 * ROWB));
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;
    // this lane's output pixels: tile i covers patch rows grp*8 + wm*4 + 2i, +1 (16 pixels each)
    int hid0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) hid0[i] = (grp * 8 + wm * 4 + i * 2 + (r >> 4)) * HW + (r & 15);
    int fb_base[TN], fb_sw[TN];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int R = wn * (BN / 2) + jn * 32 + r;
        fb_base[jn] = R * ROWB; fb_sw[jn] = (R >> 1) & 7;
    }
    // Fragments rotate through THREE register sets, indexed by the running k-step g = 4*step + ks (g % 3): the reads of k-step g+1
    // are issued before the MFMAs of g, into the set the MFMAs of g-2 used - never into registers an MFMA issued one instruction
    // earlier may still be reading. Only one wave per SIMD is in its matrix segment, so nobody else hides its LDS latency. The
    // step loop is unrolled over the 9 taps of a chunk, which makes the set indices (and the tap) compile-time constants.
    // M16: v_mfma_f32_16x16x32_bf16 form (see igemm_kernel): a 16-row fragment is one patch row, lane l holds pixel (l & 15) and chunk
    // 4*ks + (l >> 4); two k-steps of 32 per step, two register sets (set = k-step), 8 reads and 16 MFMAs per k-step.
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
    bf16x8 fa16[2][TM][2], fb16[2][TN][2];
    f32x4_t c4[TM][TN][4];
    const int c16 = lane & 15, kq = lane >> 4;
    uint32_t b16_off[TN][2];
    int b16_swz[TN][2];
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) c4[i][jn][t4] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int R = wn * (BN / 2) + jn * 32 + b * 16 + c16;
                b16_off[jn][b] = R * ROWB; b16_swz[jn][b] = (R >> 1) & 7;
            }
    }
    bf16x8 af[3][TM], bfr[3][TN];
    const uint32_t lds0 = lds_addr(smem);
    auto load16 = [&](int step, int ks, int set) {  // 8 asm reads (4 A, 4 B); the caller counts the waits
        const int c = step / 9, tap = step - c * 9;
        const uint32_t Hb = lds0 + (c & 1) * HALO_BYTES;
        const uint32_t Bb = lds0 + 2 * HALO_BYTES + (step % NSB) * BT_BYTES;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int sw = halo_key<true>(c16 + kx);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int hid = (grp * 8 + wm * 4 + i * 2 + a + ky) * HW + c16 + kx;
                fa16[set][i][a] = lds_read16<0>(Hb + hid * ROWB + (((4 * ks + kq) ^ sw) << 4));
            }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int b = 0; b < 2; ++b) fb16[set][jn][b] = lds_read16<0>(Bb + b16_off[jn][b] + (((4 * ks + kq) ^ b16_swz[jn][b]) << 4));
    };
    auto mfma16s = [&](int set) {  // e = ((i*TN + jn)*2 + a)*2 + b
#pragma unroll
        for (int e = 0; e < TM * TN * 4; ++e) {
            const int b = e & 1, a = (e >> 1) & 1, jn = (e >> 2) % TN, i = (e >> 2) / TN;
            c4[i][jn][a * 2 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa16[set][i][a], fb16[set][jn][b], c4[i][jn][a * 2 + b], 0, 0, 0);
        }
    };
    auto load_frags = [&](int step, int ks, int set) {  // 4 asm reads (2 A, 2 B); the caller counts the waits
        const int c = step / 9, tap = step - c * 9;
        const uint32_t Hb = lds0 + (c & 1) * HALO_BYTES;
        const uint32_t Bb = lds0 + 2 * HALO_BYTES + (step % NSB) * BT_BYTES;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int toff = ky * HW + kx;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int hid = hid0[i] + toff;
            const int sw = halo_key<false>((r & 15) + kx);
            af[set][i] = lds_read16<0>(Hb + hid * ROWB + (((2 * ks + h) ^ sw) << 4));
        }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) bfr[set][jn] = lds_read16<0>(Bb + fb_base[jn] + (((2 * ks + h) ^ fb_sw[jn]) << 4));
    };
    auto mfmas = [&](int set) {
#pragma unroll
        for (int e = 0; e < TM * TN; ++e) acc[e / TN][e % TN] = mfma32(af[set][e / TN], bfr[set][e % TN], acc[e / TN][e % TN]);
    };
    auto seg_barrier = [&]() {  // bare s_barrier: a __syncthreads() fence would drain the LDS-DMA pieces left in flight on purpose
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue: weight tiles of steps 0 and 1 (and waves 4-7's piece of step 2: their loop starts at step 3), halo of chunk 0
    stage_b(0);
    stage_b(1);
    if (grp == 1) {
        stage_b(2);
#pragma unroll
        for (int i = 0; i < H_I; ++i) halo_piece(i, 0);
    }
    wait_dma();
    __syncthreads();
    if constexpr (M16) load16(0, 0, 0);
    else load_frags(0, 0, 0);
    wait_lds<0>();
    if (grp == 1) seg_barrier();  // waves 4-7 start one segment late
#ifdef IR_STAMPS
    unsigned long long st_acc[4] = {0, 0, 0, 0};
#define IR_PP_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define IR_PP_ACC(k, a, b) st_acc[k] += (b) - (a)
#else
#define IR_PP_T(v) do { } while (0)
#define IR_PP_ACC(k, a, b) do { } while (0)
#endif
    auto tap_step = [&](auto uc, int c) {  // one step (matrix + vector segment); u = tap is a compile-time constant
            constexpr int u = decltype(uc)::value;
            const int s = c * 9 + u;
            constexpr int t = u;
            IR_PP_T(ta);
            // ---- matrix segment of step s (the fragments of its first k-step are already in flight)
            if constexpr (M16) {
                __builtin_amdgcn_sched_barrier(0);
                load16(s, 1, 1);
                wait_lds<2 * (TM + TN)>();  // only the eight reads just issued may still be in flight
                __builtin_amdgcn_sched_barrier(0);
                mfma16s(0);
                __builtin_amdgcn_sched_barrier(0);
                wait_lds<0>();
                __builtin_amdgcn_sched_barrier(0);
                mfma16s(1);
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + 1 < 4) {
                        load_frags(s, ks + 1, (u + ks + 1) % 3);
                        wait_lds<TM + TN>();  // only the four reads just issued may still be in flight
                    } else {
                        wait_lds<0>();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas((u + ks) % 3);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            IR_PP_T(tb);
            // this wave's weight pieces issued in its previous vector segment must land; waves 4-7 leave the halo pieces they
            // issued behind theirs in flight (NH of the previous step's tap), except in the last chunk, where there are none
            constexpr int NHP = ((u + 8) % 9) < 3 ? 2 : (((u + 8) % 9) < 8 ? 1 : 0);
            if (grp == 0 || c + 1 >= chunks) wait_dma();
            else wait_vm<NHP>();
            seg_barrier();
            IR_PP_T(tc);
            // ---- vector segment: DMA pieces, then the first fragment reads of step s+1 (its tiles were published a segment ago)
            const int sb = s + 2 + grp;  // waves 0-3 fetch weights two steps ahead, waves 4-7 (a segment later) three
            if (sb < steps) stage_b(sb);
            if (grp == 1 && c + 1 < chunks) {  // next chunk's halo over the first eight vector segments of this chunk: 2,2,2,1,1,1,1,1
                if (t < 3) { halo_piece(2 * t, c + 1); halo_piece(2 * t + 1, c + 1); }
                else if (t < 8) halo_piece(t + 3, c + 1);
                if (t == 7) wait_dma();  // ... and published by the barrier below, a segment before waves 0-3 first read it
            }
            if (s + 1 < steps) {
                if constexpr (M16) load16(s + 1, 0, 0);
                else load_frags(s + 1, 0, (u + 4) % 3);
            }
            IR_PP_T(td);
            seg_barrier();
            IR_PP_T(te);
            IR_PP_ACC(0, ta, tb); IR_PP_ACC(1, tb, tc); IR_PP_ACC(2, tc, td); IR_PP_ACC(3, td, te);
    };
    for (int c = 0; c < chunks; ++c)
        [&]<int... U>(std::integer_sequence<int, U...>) { (tap_step(std::integral_constant<int, U>{}, c), ...); }(std::make_integer_sequence<int, 9>{});
#ifdef IR_STAMPS
    if (lane == 0 && (wid & 3) == 0 && blockIdx.x < 65536 / 2)
        for (int k = 0; k < 4; ++k) g_stamps[(blockIdx.x * 2 + grp) * 8 + k] = st_acc[k];
#endif
    if (grp == 0) seg_barrier();  // pairs the late start of waves 4-7
    wait_dma();
    auto map_row = [&](int i, int row) {
        const int oy = oy0 + grp * 8 + wm * 4 + i * 2 + (row >> 4), ox = ox0 + (row & 15);
        return (oy < p.Ho && ox < p.Wo) ? (img * p.Ho + oy) * p.Wo + ox : -1;
    };
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[i][jn][g] = c4[i][jn][g >> 2][g & 3];
        igemm_epilogue<TM, TN, 2, 8, true>(p, acc, smem, wid, lane, n0 + wn * (BN / 2), n0, img, trem, map_row);
    } else {
        igemm_epilogue<TM, TN, 2, 8>(p, acc, smem, wid, lane, n0 + wn * (BN / 2), n0, img, trem, map_row);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Ping-pong big-tile GEMM for the DiT linears (TAPS = 1, Cout % 288 == 0: 1152 = 4, 3456 = 12, 4608 = 16 column tiles, so a
// 16384-token launch is exactly 1, 3 or 4 rounds of 256 workgroups): one 512-thread workgroup per CU computes a 256 x 288 tile, half
// the staged bytes per MFMA of the 128 x 128 kernel. Structure (that of conv_halo_pp_kernel / flash_attn_pp_kernel): waves w and
// w+4 share a SIMD; waves 0-3 own columns 0-143, waves 4-7 columns 144-287, each wave a 64 x 144 sub-tile (4 x 9 MFMA tiles of
// 16 x 16, 144 accumulator registers). k-tiles are 32 wide (one v_mfma_f32_16x16x32_bf16 step, 36 MFMAs per wave), staged by LDS-DMA
// into a ring of FIVE 16 KB A slots (256 rows x 64 B) and FOUR 18 KB B slots (288 x 64 B). Workgroup barriers keep the halves in
// complementary segments:
//     segment:    2k              2k+1                        2k+2
//     waves 0-3   matrix(k)       B(k+3) DMA, frags(k+1)      matrix(k+1)
//     waves 4-7   A(k+4) DMA,     matrix(k)                   A(k+5) DMA, frags(k+1)
//                 frags(k)
// matrix(k) = 36 MFMAs on the 13 fragments read in the wave's preceding vector segment (no LDS read inside a matrix segment; LDS-DMA
// pieces issued between the MFMAs instead were measured slower: the issuing wave blocks ~64 cycles per piece). The slots tile k+4
// (A) and k+3 (B) go to held tile k-1, whose last fragment reads (waves 4-7, segment 2k-2) were consumed by matrix(k-1) in segment
// 2k-1. The loop is LATENCY bound before it is anything else (activations come from HBM: with a cache-resident A operand the same
// loop ran 10-22 % faster), hence the deep A ring: waves 4-7 wait vmcnt(12) at the end of every vector segment - A(k+1) has landed,
// three younger 4-piece batches may fly (3+ k-tiles of latency cover); waves 0-3 wait vmcnt(5) at the end of every matrix segment
// - B(k+1) has landed, B(k+2) may fly (weights are L2 hits; every wave issues exactly 5 B pieces per batch: 18 pieces over 4 waves,
// the two spare slots re-issue a piece, which is idempotent). Every wait is followed by a workgroup barrier before the first read.
// LDS image: row r of a 16-row fragment tile is 64 B (4 chunks of 16 B); chunk c is stored at slot c ^ f(r >> 2), f = {0,2,3,1},
// which makes the 16 lanes of every ds_read_b128 lane group cover all 64 banks (applied to the DMA source and to the read).
struct GemmPP {
    static constexpr int BM = 256, BN = 288, BK = 32, NSA = 5, NSB = 4;
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;   // 16384, 18432
    static constexpr int B_BASE = NSA * A_BYTES;                          // 81920
    static constexpr int RING = B_BASE + NSB * B_BYTES;                   // 155648
    static constexpr int TM = 4, TN = 9;                                  // 16 x 16 tiles per wave
    static constexpr int SLAB = 32 * 144 * 4;                             // epilogue: two 16-row tiles of a wave at a time (fp32)
    static constexpr int PF_OFF = RING;                                   // (-DIR_GPP_PF builds only) 1 KB nobody reads: landing area of the A operand's prefetch touches
    static constexpr int LDS_PF = RING + 1024 > 8 * SLAB ? RING + 1024 : 8 * SLAB;
    static constexpr int LDS = RING > 8 * SLAB ? RING : 8 * SLAB;
};

// Round 6: the epilogue forms the DiT uses are compiled as their own instantiations (ACT_T / KIND_T >= 0: the activation and the row phase
// are compile-time constants; -1: decided at run time, the generic form every other caller gets). Before, ONE kernel carried every activation in
// every row phase behind run-time switches inside its unrolled loops: the two halves of the accumulator tile were selected per ELEMENT by a
// v_cndmask (the half index stayed a loop variable) and each half branched over the activation codes. UNIT: no per-column gate and out_scale == 1
// (the launcher checks), so the multiply behind the activation disappears. The tanh-GELU is common.h's five-instruction form. (The bias stays an
// add in the epilogue: started in the accumulators it saved one instruction per element, but bias + sum rounds differently from sum + bias, and the
// 128 x 128 kernel that takes the same linear at smaller row counts adds it last - a tile-sharded frame then differed from the unsharded one by
// up to 4 grey levels, tests/test_cli_gpu.py::test_tile_sharding_two_ranks_on_one_gpu.)
// Experiment knob (-DIR_GPP_PF=d, d > 0; never set in the library): every A-issuing wave also touches, per k-tile, the rows of its four pieces d
// k-tiles AHEAD of the tile it stages (one LDS-DMA piece of 64 rows x 16 B into a landing area nobody reads), to turn the A fetch - a miss for all
// four column-tile workgroups of a row tile, which run in lockstep on one XCD - into an L2 hit. Measured SLOWER whatever d (2, 4, 8): the kernel
// 17.4 -> 19.6 ms per image, every shape -12 to -16 % (profiles/r06_ab_gemm_l2_prefetch.txt). One more piece per four is +12 % of the bytes this
// loop moves through LDS-DMA and it costs +13 %: the loop's time follows the LDS-DMA volume (34 KB per k-tile and CU), not the A operand's latency.
#ifndef IR_GPP_PF
#define IR_GPP_PF 0
#endif
#ifndef IR_GPP_KO
#define IR_GPP_KO 0   // knock-outs of the fp32-residual row phase, timing only (results wrong by design; never set in the library): 1 no residual
#endif                // read, 2 no bf16 copy, 3 no fp32 store
template <int ACT_T, int KIND_T, bool UNIT>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(IGemmParams p) {
    typedef GemmPP G;
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[IR_GPP_PF > 0 ? G::LDS_PF : G::LDS];  // the ONLY LDS object of the kernel
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int grp = wu >> 2, wq = wu & 3;   // half (column group) and row quarter of this wave
    const int NT = p.Cout_pad / G::BN, MT = (p.M + G::BM - 1) / G::BM;
    const int bid = blockIdx.x, xcd = bid & 7, jb = bid >> 3;
    const int mt = (jb / NT) * 8 + xcd, nt = jb % NT;   // an XCD runs the column tiles of one row tile back to back (A re-read from its L2)
    if (mt >= MT) return;
    const int m0 = mt * G::BM, n0 = nt * G::BN;
    const int KT = p.Cin / G::BK;

    // ---- DMA sources: a piece = 16 tile rows x 64 B; lane l covers row l >> 2, LDS slot l & 3 <- global chunk (l & 3) ^ f(row >> 2)
    const int prow = lane >> 2;
    const int fsw = (0x1320 >> (4 * ((prow >> 2) & 3))) & 3;   // f = {0, 2, 3, 1}
    const int pchunk = ((lane & 3) ^ fsw) * 8;                 // element offset of the source chunk within the k-tile
    // waves 4-7: A pieces 4*wq .. 4*wq+3 (rows beyond M re-read row M-1 and are never stored); waves 0-3: B pieces wq + 4*i, i < 5
    const bf16_t* src[5];
    int dst[5];
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = wq * 4 + i;
            src[i] = p.in + (long)min(m0 + q * 16 + prow, p.M - 1) * p.in_cs + pchunk;
            dst[i] = q * 1024;
        }
        src[4] = src[3]; dst[4] = dst[3];
    } else {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int q = (wq + 4 * i) < 18 ? wq + 4 * i : wq + 12;   // spare slots repeat a piece (same bytes, same destination)
            src[i] = p.wgt + (long)(n0 + q * 16 + prow) * p.wgt_rs + pchunk;
            dst[i] = G::B_BASE + q * 1024;
        }
    }
    // L2 prefetch touch (IR_GPP_PF): lane l of A-issuing wave wq touches row 64 * wq + l (16 bytes of the k-tile's half line)
    const bf16_t* pf_src = p.in + (long)min(m0 + wq * 64 + lane, p.M - 1) * p.in_cs;
    // prologue: A tiles 0..3 and B tiles 0..2 by all 8 waves (A piece q = wu + 8*i, i < 2; B piece q = wu + 8*i, i < 3, q < 18)
    auto issue_prologue = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int kt = 0; kt < G::NSA - 1; ++kt)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = wu + 8 * i;
                glds16(p.in + (long)min(m0 + q * 16 + prow, p.M - 1) * p.in_cs + pchunk + kt * G::BK, (lds_ptr_t)(smem + kt * G::A_BYTES + q * 1024));
            }
#pragma unroll
        for (int kt = 0; kt < G::NSB - 1; ++kt)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = wu + 8 * i;
                if (q < 18)
                    glds16(p.wgt + (long)(n0 + q * 16 + prow) * p.wgt_rs + pchunk + kt * G::BK, (lds_ptr_t)(smem + G::B_BASE + kt * G::B_BYTES + q * 1024));
            }
    };

    // ---- fragments: lane (r16 = l & 15, kq = l >> 4) reads chunk kq of row r16 of a 16-row tile
    const int r16 = lane & 15, kq = lane >> 4;
    const int frag_off = r16 * 64 + ((kq ^ ((0x1320 >> (4 * ((r16 >> 2) & 3))) & 3)) << 4);
    const unsigned char* fbase = smem + frag_off + wq * 4 * 1024;                     // A tiles 4*wq + i
    const unsigned char* gbase = smem + frag_off + G::B_BASE + grp * 9 * 1024;        // B tiles 9*grp + j
    bf16x8 fa[G::TM], fb[G::TN];
    auto read_frags = [&](int sa, int sb) __attribute__((always_inline)) {   // A slot sa, B slot sb
#pragma unroll
        for (int i = 0; i < G::TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(fbase + sa * G::A_BYTES + i * 1024);
#pragma unroll
        for (int j = 0; j < G::TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(gbase + sb * G::B_BYTES + j * 1024);
    };
    f32x4_t acc[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // (Round 5 tried the GELU as packed fp32 pairs - v_pk_add / v_pk_mul / v_pk_fma - on the slab-write side: the fc1 launch alone 221 -> 210 us, but
    // at the kernel's 256-VGPR limit the pair form spilled 23 registers and every launch paid for the scratch set-up: profiles/r05_ab_gemm_pk_gelu.txt.)
#ifndef IR_GKO
#define IR_GKO 0  // knock-out builds for timing only (-DIR_GKO=n, results wrong by design; never set in the library): 1 no MFMAs, 2 no
#endif            // LDS-DMA in the K loop, 3 no fragment reads - DESIGN.md section 7 quotes the three timings
    auto matrix = [&]() __attribute__((always_inline)) {
        if (IR_GKO == 1) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
#pragma unroll
            for (int i = 0; i < G::TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    issue_prologue();
    wait_dma();
    __builtin_amdgcn_s_barrier();
    if (grp == 0) {
        read_frags(0, 0);
        int sa1 = 1;   // A slot of tile kt+1
        for (int kt = 0; kt < KT; ++kt) {
            __builtin_amdgcn_sched_barrier(0);
            matrix();
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 3 < KT) wait_vm<5>(); else wait_dma();       // B(kt+1) has landed (only B(kt+2) may be in flight)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 3 < KT) {
                unsigned char* base = smem + ((kt + 3) & 3) * G::B_BYTES;
                const int ko = (kt + 3) * G::BK;
#pragma unroll
                for (int i = 0; i < 5; ++i) if (IR_GKO != 2 && IR_GKO != 4) glds16(src[i] + ko, (lds_ptr_t)(base + dst[i]));   // (4: no B pieces only)
            }
            if (kt + 1 < KT && IR_GKO != 3) read_frags(sa1, (kt + 1) & 3);
            sa1 = sa1 == G::NSA - 1 ? 0 : sa1 + 1;
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
    } else {
        int sa = 0, sa4 = G::NSA - 1;   // A slots of tiles kt and kt+4
        for (int kt = 0; kt < KT; ++kt) {
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 4 < KT) {
                unsigned char* base = smem + sa4 * G::A_BYTES;
                const int ko = (kt + 4) * G::BK;
#pragma unroll
                for (int i = 0; i < 4; ++i) if (IR_GKO != 2 && IR_GKO != 5) glds16(src[i] + ko, (lds_ptr_t)(base + dst[i]));   // (5: no A pieces only)
                if (IR_GPP_PF > 0) glds16(pf_src + min(kt + 4 + IR_GPP_PF, KT - 1) * G::BK, (lds_ptr_t)(smem + G::PF_OFF));   // part of the batch: BATCH pieces
            }
            if (IR_GKO != 3 || kt == 0) read_frags(sa, kt & 3);
            // A(kt+1) has landed; the batches of A(kt+2) .. A(kt+4) may be in flight (fewer at the end of the K loop)
            const int rem = KT - 2 - kt;
            constexpr int BATCH = IR_GPP_PF > 0 ? 5 : 4;
            if (rem >= 3) wait_vm<3 * BATCH>(); else if (rem == 2) wait_vm<2 * BATCH>(); else if (rem == 1) wait_vm<BATCH>(); else wait_dma();
            sa = sa == G::NSA - 1 ? 0 : sa + 1;
            sa4 = sa4 == G::NSA - 1 ? 0 : sa4 + 1;
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            matrix();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: v = act(acc + bias) * out_scale * gate + res, through a wave-private fp32 slab (two 16-row tiles =
    // 32 x 144 at a time) so that residual reads and stores are row-contiguous 16-byte vectors. All loop reads were consumed before the last
    // barrier, so the slabs may overlay the ring.
    float* slab = reinterpret_cast<float*>(smem) + wu * (G::SLAB / 4);
    const int col = lane & 15, rq = lane >> 4;
    const int nw = n0 + grp * 144;          // first column of this wave
    const int act = ACT_T >= 0 ? ACT_T : p.act;
    constexpr bool CB_REGS = ACT_T >= 0;   // the instantiated forms keep the nine bias values in registers; the run-time form (at its register limit) re-reads them
    float cb[CB_REGS ? G::TN : 1], cm[UNIT ? 1 : G::TN];
    if constexpr (CB_REGS) {
#pragma unroll
        for (int j = 0; j < G::TN; ++j) cb[j] = p.bias ? p.bias[nw + j * 16 + col] : 0.f;
    }
    if constexpr (!UNIT) {
#pragma unroll
        for (int j = 0; j < G::TN; ++j) cm[j] = p.out_scale * (p.gate ? p.gate[nw + j * 16 + col] : 1.f);
    }
    auto half = [&](auto act_tag, auto hc) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value, HH = decltype(hc)::value;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < G::TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float bj;
                    if constexpr (CB_REGS) bj = cb[j];
                    else bj = p.bias ? p.bias[nw + j * 16 + col] : 0.f;
                    float y = apply_act<ACT>(acc[2 * HH + ii][j][q] + bj, p.slope);
                    if constexpr (!UNIT) y *= cm[j];
                    slab[(ii * 16 + rq * 4 + q) * 144 + j * 16 + col] = y;
                }
    };
    // Row phase of one half. KIND 0 = no residual, bf16 out; KIND 1 = fp32 residual, fp32 out (+ optional bf16 copy), whose 18 residual vectors
    // are requested six at a time (one exposed latency per batch); KIND 2 = anything else.
    auto write_half = [&](auto hc) __attribute__((always_inline)) {
        switch (act) {
            case IR_ACT_GELU_TANH: half(std::integral_constant<int, IR_ACT_GELU_TANH>{}, hc); break;
            case IR_ACT_GELU_ERF: half(std::integral_constant<int, IR_ACT_GELU_ERF>{}, hc); break;
            case IR_ACT_SILU: half(std::integral_constant<int, IR_ACT_SILU>{}, hc); break;
            case IR_ACT_LRELU: half(std::integral_constant<int, IR_ACT_LRELU>{}, hc); break;
            default: half(std::integral_constant<int, IR_ACT_NONE>{}, hc); break;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto done_half = [&]() __attribute__((always_inline)) {  // the next half's slab writes must not pass this half's slab reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const int kind = KIND_T >= 0 ? KIND_T : ((!p.res && !p.out_f32 && !p.out2) ? 0 : (p.res && p.res_f32 && p.out_f32 && p.res_mod == 0) ? 1 : 2);
    const int mw = m0 + wq * 64;
    // fast forms address with 32-bit element offsets from the uniform base pointers (the launcher checks that they fit): one
    // address register per vector instead of two
    auto rows_kind0 = [&](auto hc) __attribute__((always_inline)) {
        constexpr int hh = decltype(hc)::value;
        bf16_t* outb = reinterpret_cast<bf16_t*>(p.out);
        write_half(hc);
        if (KIND_T == 0 && (p.out_cs & 7) == 0) {   // 16-byte stores: 18 lanes x 8 columns per row, 9 vectors per lane and half (the instantiated bf16 forms;
                                                    // the launcher has checked the 16-byte alignment of p.out, nw is a multiple of 8)
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                const int v = it * 64 + lane, row = v / 18, c8 = (v - row * 18) * 8;
                const int m = mw + hh * 32 + row;
                const f32x4_t o0 = *reinterpret_cast<const f32x4_t*>(&slab[row * 144 + c8]), o1 = *reinterpret_cast<const f32x4_t*>(&slab[row * 144 + c8 + 4]);
                if (m < p.M) *reinterpret_cast<uint4*>(outb + (unsigned)(m * p.out_cs + nw + c8)) =
                    make_uint4(pack2bf(o0[0], o0[1]), pack2bf(o0[2], o0[3]), pack2bf(o1[0], o1[1]), pack2bf(o1[2], o1[3]));
            }
        } else {
#pragma unroll
            for (int it = 0; it < 18; ++it) {
                const int v = it * 64 + lane, row = v / 36, c4 = (v - row * 36) * 4;
                const int m = mw + hh * 32 + row;
                const f32x4_t o = *reinterpret_cast<const f32x4_t*>(&slab[row * 144 + c4]);
                if (m < p.M) *reinterpret_cast<uint2*>(outb + (unsigned)(m * p.out_cs + nw + c4)) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
            }
        }
        if (p.vt_out && nw >= p.vt_col0) {   // wave-uniform: this wave's 144 columns are two heads of V
            // The slab holds 32 consecutive tokens x 144 columns: a lane takes a column (three passes over the 144) and writes its 32
            // tokens as 64 contiguous bytes of the V^T row of that (head, d). Launcher: M % 256 == 0 and vt_T % 64 == 0, so the 32 tokens
            // never straddle an image and the destination is 64-byte aligned.
            const int tok0 = mw + hh * 32, bidx = tok0 / p.vt_T, t0 = tok0 - bidx * p.vt_T;
#pragma unroll
            for (int pass = 0; pass < 3; ++pass) {
                const int c = pass * 64 + lane;
                if (c < 144) {
                    const int nn = nw + c - p.vt_col0, head = nn / p.vt_hd, d = nn - head * p.vt_hd;
                    bf16_t* dst = p.vt_out + (long)bidx * p.vt_bs + ((long)head * p.vt_dv + d) * p.vt_ld + t0;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        uint32_t w[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = pack2bf(slab[(g4 * 8 + 2 * e) * 144 + c], slab[(g4 * 8 + 2 * e + 1) * 144 + c]);
                        *reinterpret_cast<uint4*>(dst + g4 * 8) = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                }
            }
        }
        done_half();
    };
    // KIND 1 (the DiT's residual stream: fp32 in, fp32 out in place, optional bf16 copy). The residual vectors come in FOUR batches of nine (two per
    // half), each requested one batch ahead of its use - the first before the slab of half 0 is written, the third (half 1's first) while half 0's
    // second is processed - so that one memory latency is exposed per launch instead of six (round 5: three batches of six per half, each requested
    // after the previous one's stores; the knock-out without the reads was 13 us of a 68 us launch shorter, profiles/r06_gemm_ab_ops.txt). A lane
    // reads and writes the same elements and the batches are disjoint, so the in-place form (res == out) stays exact.
    auto k1_load = [&](auto hc, auto i0c, auto nc, f32x4_t* rr) __attribute__((always_inline)) {   // vectors it0 .. it0 + n - 1 of half hh
        constexpr int hh = decltype(hc)::value, it0 = decltype(i0c)::value, n = decltype(nc)::value;
        const float* resf = reinterpret_cast<const float*>(p.res);
#pragma unroll
        for (int it = 0; it < n; ++it) {
            const int v = (it0 + it) * 64 + lane, row = v / 36, c4 = (v - row * 36) * 4;
            const int m = min(mw + hh * 32 + row, p.M - 1);
            if (IR_GPP_KO == 1) rr[it] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            else rr[it] = *reinterpret_cast<const f32x4_t*>(resf + (unsigned)(m * p.res_cs + nw + c4));
        }
    };
    auto k1_rows = [&](auto hc, auto i0c, auto nc, const f32x4_t* rr) __attribute__((always_inline)) {
        constexpr int hh = decltype(hc)::value, it0 = decltype(i0c)::value, n = decltype(nc)::value;
        float* outf = reinterpret_cast<float*>(p.out);
#pragma unroll
        for (int it = 0; it < n; ++it) {
            const int v = (it0 + it) * 64 + lane, row = v / 36, c4 = (v - row * 36) * 4;
            const int m = mw + hh * 32 + row;
            const f32x4_t o = *reinterpret_cast<const f32x4_t*>(&slab[row * 144 + c4]) + rr[it];
            if (m < p.M) {
                if (IR_GPP_KO != 3) *reinterpret_cast<f32x4_t*>(outf + (unsigned)(m * p.out_cs + nw + c4)) = o;
                if (p.out2 && IR_GPP_KO != 2) *reinterpret_cast<uint2*>(p.out2 + (unsigned)(m * p.out2_cs + nw + c4)) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
            }
        }
    };
#ifndef IR_GPP_K1
#define IR_GPP_K1 0   // residual request schedule of the instantiated KIND 1 form: 0 = three batches of six per half, each after the previous one's stores;
#endif                // 1 = IR_GPP_K1_NA vectors of a half requested BEFORE its slab is written, the rest behind it. Measured in round 6 (6 + 12, 9 + 9, 12 + 6
#ifndef IR_GPP_K1_NA  // against 0, profiles/r06_gemm_k1_schedules.txt): 55 us per 1152 -> 1152 launch and 150 us per 4608 -> 1152 launch whatever the schedule -
#define IR_GPP_K1_NA 9   // the row phase is bound by the memory system's burst (189 MB in and out per launch while every CU is in its epilogue), not by the
#endif                   // latency of the requests. 0 stays.
    auto run_kind1 = [&](auto hc) __attribute__((always_inline)) {
        constexpr int hh = decltype(hc)::value;
        using I0 = std::integral_constant<int, 0>;
        if constexpr (KIND_T == 1 && IR_GPP_K1 == 1) {
            constexpr int NA = IR_GPP_K1_NA;
            f32x4_t ra[NA], rb[18 - NA > 0 ? 18 - NA : 1];
            k1_load(hc, I0{}, std::integral_constant<int, NA>{}, ra);
            write_half(hc);   // this half's 72 accumulator registers die here
            if constexpr (NA < 18) k1_load(hc, std::integral_constant<int, NA>{}, std::integral_constant<int, 18 - NA>{}, rb);
            k1_rows(hc, I0{}, std::integral_constant<int, NA>{}, ra);
            if constexpr (NA < 18) k1_rows(hc, std::integral_constant<int, NA>{}, std::integral_constant<int, 18 - NA>{}, rb);
        } else {
            write_half(hc);
            f32x4_t rr[6];
            [&]<int... BT>(std::integer_sequence<int, BT...>) {
                ((k1_load(hc, std::integral_constant<int, 6 * BT>{}, std::integral_constant<int, 6>{}, rr),
                  k1_rows(hc, std::integral_constant<int, 6 * BT>{}, std::integral_constant<int, 6>{}, rr)), ...);
            }(std::make_integer_sequence<int, 3>{});
        }
        done_half();
    };
    auto rows_kind2 = [&](auto hc) __attribute__((always_inline)) {
        constexpr int hh = decltype(hc)::value;
        write_half(hc);
        for (int it = 0; it < 18; ++it) {
            const int v = it * 64 + lane, row = v / 36, c4 = (v - row * 36) * 4;
            const int m = mw + hh * 32 + row;
            f32x4_t o = *reinterpret_cast<const f32x4_t*>(&slab[row * 144 + c4]);
            if (m < p.M) {
                const int n = nw + c4;
                if (p.res) {
                    const long rm = p.res_mod > 0 ? (long)(m % p.res_mod) : (long)m;
                    if (p.res_f32) o += *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(p.res) + rm * p.res_cs + n);
                    else {
                        const uint2 rb = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.res) + rm * p.res_cs + n);
                        o += f32x4_t{bflo(rb.x), bfhi(rb.x), bflo(rb.y), bfhi(rb.y)};
                    }
                }
                const uint2 pk = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
                if (p.out_f32) *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(p.out) + (long)m * p.out_cs + n) = o;
                else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (long)m * p.out_cs + n) = pk;
                if (p.out2) *reinterpret_cast<uint2*>(p.out2 + (long)m * p.out2_cs + n) = pk;
            }
        }
        done_half();
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    if (kind == 0) { rows_kind0(H0{}); rows_kind0(H1{}); }
    else if (kind == 1) { run_kind1(H0{}); run_kind1(H1{}); }
    else { rows_kind2(H0{}); rows_kind2(H1{}); }
}

int g_ir_plain_kernels = 0;

// Grid of a tiled launch (see tile_of_block): one block per tile for small launches, the XCD-padded order otherwise. IR_NO_TILE_LIN: experiment knob.
static long tile_grid(long MT, long NT) {
    static const bool no_lin = getenv("IR_NO_TILE_LIN") != nullptr;
    return ((MT & 7) && MT < 64 && !no_lin) ? MT * NT : ((MT + 7) / 8) * 8 * NT;
}

// Shortest reduction gemm_pp_kernel takes: 8 k-tiles. (Its prologue needs 4; with the limit at 6 SwinIR's qkv projection - K = 192, N = 576 =
// 2 x 288 - runs here: measured in round 4 at 36 us per launch, exactly what igemm_kernel<128, 64> needs for it: no gain, limit left at 8.
// IR_GEMM_PP_MIN_K: experiment knob.)
static int gemm_pp_min_k() {
    static const int v = [] { const char* e = getenv("IR_GEMM_PP_MIN_K"); const int k = e ? atoi(e) : 8 * GemmPP::BK; return k < 4 * GemmPP::BK ? 4 * GemmPP::BK : k; }();
    return v;
}
static bool takes_gemm_pp(const IGemmParams& p) {
    static const bool off = getenv("IR_NO_GEMM_PP") != nullptr;  // experiment knob
    if (off || g_ir_plain_kernels || p.fp8 || p.taps != 1 || p.force_generic || !p.vec || p.gn_part) return false;
    if (p.Cout != p.Cout_pad || p.Cout % GemmPP::BN || p.Cin % GemmPP::BK || p.Cin < gemm_pp_min_k()) return false;
    const long span = (long)p.M * std::max(std::max(p.out_cs, p.res ? p.res_cs : 0), p.out2 ? p.out2_cs : 0);
    if (span >= (1L << 31)) return false;  // the epilogue's 32-bit element offsets
    const long blocks = (long)((p.M + GemmPP::BM - 1) / GemmPP::BM) * (p.Cout / GemmPP::BN);
    return blocks >= 192;  // below that the 128 x 128 kernel fills the chip better
}
// The transposed second output (IGemmParams::vt_out) is honoured by gemm_pp_kernel's bf16 / no-residual form on whole tiles only
static bool igemm_vec(const IGemmParams& p);
int ir_igemm_writes_vt(const IGemmParams& pin) {
    static const bool off = getenv("IR_NO_VT_FUSE") != nullptr;   // experiment knob
    IGemmParams p = pin;
    p.vec = igemm_vec(p);
    if (p.ks_ws && ir_igemm_splitk(p) > 1) return 0;
    return !off && p.vt_out && takes_gemm_pp(p) && !p.res && !p.out_f32 && !p.out2 && p.M % GemmPP::BM == 0 && p.vt_T > 0 && p.vt_T % 64 == 0 &&
           p.vt_col0 % GemmPP::BN == 0 && p.vt_hd * 2 == 144 && (p.Cout - p.vt_col0) % 144 == 0 && (p.vt_ld & 7) == 0 && (p.vt_bs & 7) == 0 &&
           !(reinterpret_cast<uintptr_t>(p.vt_out) & 15);
}
static int launch_gemm_pp(const IGemmParams& p, hipStream_t s) {
    const int MT = (p.M + GemmPP::BM - 1) / GemmPP::BM, NT = p.Cout / GemmPP::BN;
    const dim3 grid(((MT + 7) / 8) * 8 * NT);
    static const bool generic = getenv("IR_GEMM_PP_GENERIC") != nullptr;   // experiment knob: the run-time form for every launch
    const bool unit = !p.gate && p.out_scale == 1.f;
    const int kind = (!p.res && !p.out_f32 && !p.out2) ? 0 : (p.res && p.res_f32 && p.out_f32 && p.res_mod == 0) ? 1 : 2;
    if (generic) hipLaunchKernelGGL((gemm_pp_kernel<-1, -1, false>), grid, dim3(512), 0, s, p);
    else if (kind == 0 && unit && p.act == IR_ACT_NONE) hipLaunchKernelGGL((gemm_pp_kernel<IR_ACT_NONE, 0, true>), grid, dim3(512), 0, s, p);
    else if (kind == 0 && unit && p.act == IR_ACT_GELU_TANH) hipLaunchKernelGGL((gemm_pp_kernel<IR_ACT_GELU_TANH, 0, true>), grid, dim3(512), 0, s, p);
    else if (kind == 1 && p.act == IR_ACT_NONE) hipLaunchKernelGGL((gemm_pp_kernel<IR_ACT_NONE, 1, false>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((gemm_pp_kernel<-1, -1, false>), grid, dim3(512), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

static int launch_halo_pp(const IGemmParams& p, hipStream_t s) {
    const int tiles_y = (p.Ho + 15) / 16, tiles_x = (p.Wo + 15) / 16;
    const long MT = (long)p.NB * tiles_y * tiles_x, NT = p.Cout_pad / 128;
    const long grid = tile_grid(MT, NT);
    if (grid > 0x7fffffffL) return -12;
    static const bool m16 = getenv("IR_NO_MFMA16") == nullptr;  // 16x16x32 MFMA form by default (knob: A/B against 32x32x16)
    if (m16 && p.up) hipLaunchKernelGGL((conv_halo_pp_kernel<1, true>), dim3((unsigned)grid), dim3(512), 0, s, p, tiles_y, tiles_x);
    else if (m16) hipLaunchKernelGGL((conv_halo_pp_kernel<0, true>), dim3((unsigned)grid), dim3(512), 0, s, p, tiles_y, tiles_x);
    else if (p.up) hipLaunchKernelGGL((conv_halo_pp_kernel<1>), dim3((unsigned)grid), dim3(512), 0, s, p, tiles_y, tiles_x);
    else hipLaunchKernelGGL((conv_halo_pp_kernel<0>), dim3((unsigned)grid), dim3(512), 0, s, p, tiles_y, tiles_x);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int BN>
static int launch_halo(const IGemmParams& p, hipStream_t s) {
    const int tiles_y = (p.Ho + 7) / 8, tiles_x = (p.Wo + 15) / 16;
    const long MT = (long)p.NB * tiles_y * tiles_x, NT = p.Cout_pad / BN;
    const long grid = tile_grid(MT, NT);
    if (grid > 0x7fffffffL) return -12;
    if (p.fp8) {
        if (p.up) hipLaunchKernelGGL((conv_halo_kernel<BN, 1, false, true>), dim3((unsigned)grid), dim3(256), 0, s, p, tiles_y, tiles_x);
        else hipLaunchKernelGGL((conv_halo_kernel<BN, 0, false, true>), dim3((unsigned)grid), dim3(256), 0, s, p, tiles_y, tiles_x);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    static const bool m16 = getenv("IR_NO_MFMA16") == nullptr;  // 16x16x32 MFMA form by default (knob: A/B against 32x32x16)
    if (m16 && p.up) hipLaunchKernelGGL((conv_halo_kernel<BN, 1, true>), dim3((unsigned)grid), dim3(256), 0, s, p, tiles_y, tiles_x);
    else if (m16) hipLaunchKernelGGL((conv_halo_kernel<BN, 0, true>), dim3((unsigned)grid), dim3(256), 0, s, p, tiles_y, tiles_x);
    else if (p.up) hipLaunchKernelGGL((conv_halo_kernel<BN, 1>), dim3((unsigned)grid), dim3(256), 0, s, p, tiles_y, tiles_x);
    else hipLaunchKernelGGL((conv_halo_kernel<BN, 0>), dim3((unsigned)grid), dim3(256), 0, s, p, tiles_y, tiles_x);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Sub-pixel phase form through the 4-wave halo kernel (64- or 128-channel tiles; what conv_halo_s1_kernel<0, 4> does not take)
static bool takes_halo_up2x2(const IGemmParams& p) {
    static const bool off = getenv("IR_NO_UP2X2") != nullptr;   // experiment knob (shared with conv_s1.hip)
    return !off && !g_ir_plain_kernels && p.up && p.taps == 9 && p.stride == 1 && !p.fp8 && !p.res && !p.gn_part && !p.gate && !p.out2 && !p.out_f32 &&
           p.Cin % 64 == 0 && p.Cout_pad % 64 == 0 && p.Ho == 2 * p.H && p.Wo == 2 * p.W && igemm_vec(p);
}
bool ir_igemm_up2x2_takes(const IGemmParams& p) { return ir_conv_s1_up2x2_takes(p) || takes_halo_up2x2(p); }
static int launch_halo_up2x2(const IGemmParams& pin, hipStream_t s) {
    IGemmParams p = pin;
    p.vec = igemm_vec(p);
    if (p.wgt_rs != 4L * p.Cin) return -3;
    const int tiles_y = (p.H + 7) / 8, tiles_x = (p.W + 15) / 16;
    const long MT = (long)p.NB * 4 * tiles_y * tiles_x, NT = p.Cout_pad / (p.Cout_pad % 128 == 0 ? 128 : 64);
    const long grid = tile_grid(MT, NT);
    if (grid > 0x7fffffffL) return -12;
    if (p.Cout_pad % 128 == 0) hipLaunchKernelGGL((conv_halo_kernel<128, 0, true, false, true>), dim3((unsigned)grid), dim3(256), 0, s, p, tiles_y, tiles_x);
    else hipLaunchKernelGGL((conv_halo_kernel<64, 0, true, false, true>), dim3((unsigned)grid), dim3(256), 0, s, p, tiles_y, tiles_x);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Second half of a split-K launch: out = act(sum_k ws[k] + bias) * out_scale * gate + res, the epilogue order of igemm_epilogue, slices added in
// a fixed order. ws: [nsplit][M][Cout_pad] fp32.
__global__ __launch_bounds__(256) void splitk_finish_kernel(IGemmParams p, const float* __restrict__ ws, int nsplit) {
    const int vpr = p.Cout_pad >> 2;
    const long nv = (long)p.M * vpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        const long m = i / vpr;
        const int n0 = (int)(i - m * vpr) * 4;
        f32x4 a = *reinterpret_cast<const f32x4*>(ws + m * p.Cout_pad + n0);
        for (int k = 1; k < nsplit; ++k) a += *reinterpret_cast<const f32x4*>(ws + ((long)k * p.M + m) * p.Cout_pad + n0);
        const long rm = p.res_mod > 0 ? m % p.res_mod : m;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + e;
            if (n >= p.Cout) break;
            float x = a[e] + (p.bias ? p.bias[n] : 0.f);
            switch (p.act) {
                case IR_ACT_GELU_ERF: x = gelu_erf(x); break;
                case IR_ACT_GELU_TANH: x = gelu_tanh(x); break;
                case IR_ACT_LRELU: x = x > 0.f ? x : x * p.slope; break;
                case IR_ACT_SILU: x = silu(x); break;
                default: break;
            }
            x *= p.out_scale * (p.gate ? p.gate[n] : 1.f);
            if (p.res) x += p.res_f32 ? reinterpret_cast<const float*>(p.res)[rm * p.res_cs + n] : bf2f(reinterpret_cast<const bf16_t*>(p.res)[rm * p.res_cs + n]);
            if (p.out_f32) reinterpret_cast<float*>(p.out)[m * p.out_cs + n] = x;
            else reinterpret_cast<bf16_t*>(p.out)[m * p.out_cs + n] = f2bf(x);
            if (p.out2) p.out2[m * p.out2_cs + n] = f2bf(x);
        }
    }
}

// Split count ir_launch_igemm uses for p (0: no split): only launches whose caller allows it (p.allow_splitk) and provides the workspace
// (p.ks_ws, ir_igemm_splitk(p) * p.M * p.Cout_pad floats), with at most 48 output tiles and at least 24 k-tiles.
int ir_igemm_splitk(const IGemmParams& p) {
    static const bool off = getenv("IR_NO_SPLITK") != nullptr;   // experiment knob
    if (!p.allow_splitk || off || p.fp8 || p.gn_part || (p.Cin & 63) || (p.Cout_pad & 3)) return 0;
    const int BN = p.Cout_pad % 128 == 0 ? 128 : (p.Cout_pad % 64 == 0 ? 64 : 32);
    const int tiles = ((p.M + 127) / 128) * (p.Cout_pad / BN), KT = p.taps * (p.Cin / 64);
    // experiment knobs (defaults = the shipped heuristic): most tiles a split launch may have, fewest k-tiles it must have, workgroups aimed at,
    // fewest k-tiles per split
    static const int max_tiles = getenv("IR_SPLITK_TILES") ? atoi(getenv("IR_SPLITK_TILES")) : 48, min_kt = getenv("IR_SPLITK_KT") ? atoi(getenv("IR_SPLITK_KT")) : 24;
    static const int target = getenv("IR_SPLITK_TARGET") ? atoi(getenv("IR_SPLITK_TARGET")) : 256, per_min = getenv("IR_SPLITK_PER") ? atoi(getenv("IR_SPLITK_PER")) : 8;
    if (tiles > max_tiles || KT < min_kt) return 0;
    int ks = std::min(target / tiles, KT / std::max(per_min, 1));
    if (ks < 2) return 0;
    const int per = (KT + ks - 1) / ks;
    return (KT + per - 1) / per;   // no empty split
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const IGemmParams& pin, hipStream_t s) {
    IGemmParams p = pin;
    const int MT = (p.M + BM - 1) / BM, NT = p.Cout_pad / BN;
    const int tiles = (int)tile_grid(MT, NT);
    const int ks = p.ks_ws ? ir_igemm_splitk(pin) : 0;
    const dim3 grid(tiles, ks > 1 ? ks : 1);
    if (ks > 1) {   // partial tiles into the workspace; bias / activation / residual in splitk_finish_kernel
        p.ksplit = ks;
        p.bias = nullptr; p.act = IR_ACT_NONE; p.out_scale = 1.f; p.gate = nullptr; p.res = nullptr; p.out2 = nullptr; p.gn_part = nullptr;
        p.out = p.ks_ws; p.out_f32 = 1; p.out_cs = p.Cout_pad;
        p.vec = 1;
    } else {
        p.ksplit = 0;
    }
    static const bool force32 = getenv("IR_IGEMM_BK32") != nullptr;  // experiment knob
    const bool k64 = (p.Cin & 63) == 0 && !force32;
    // the four-tile ring (igemm_kernel's NST): launches of at most one workgroup per CU with at least eight k-tiles per workgroup (with more
    // workgroups the two-tile form's second workgroup per CU covers the waits better: 16384 x 640 -> 5120 at M = 1024 ran 17 against 25 us)
    static const int ring_max = getenv("IR_IGEMM_RING_MAX") ? atoi(getenv("IR_IGEMM_RING_MAX")) : 256;   // experiment knob (0: never)
    static const bool m16r = getenv("IR_NO_MFMA16") == nullptr;
    const int kt_all = p.taps * (p.Cin / 64), kt_per = ks > 1 ? (kt_all + ks - 1) / ks : kt_all;
    if (k64 && m16r && !g_ir_plain_kernels && (long)tiles * (ks > 1 ? ks : 1) <= ring_max && kt_per >= 8) {
        // ... and with EIGHT waves (two per SIMD, each a 32-row slice of the tile) where the tile is 2 x 2 waves: alone on its SIMD a wave pays the
        // issue of its LDS-DMA pieces (about 64 cycles each, 8 per k-tile = as long as its 32 MFMAs) with the matrix pipe idle; the second wave's
        // MFMAs run under them. Not with fused GroupNorm statistics (their block reduction sums per wave row: another order).
        static const bool no_w8 = getenv("IR_IGEMM_NO_W8") != nullptr;   // experiment knob
        if constexpr (WM == 2 && WN == 2) {
            if (!no_w8 && !p.gn_part) {
                if (p.taps == 9) hipLaunchKernelGGL((igemm_kernel<BM, BN, 4, 2, 9, 64, true, 4>), dim3(grid), dim3(512), 0, s, p);
                else hipLaunchKernelGGL((igemm_kernel<BM, BN, 4, 2, 1, 64, true, 4>), dim3(grid), dim3(512), 0, s, p);
                goto launched;
            }
        }
        if (p.taps == 9) hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 9, 64, true, 4>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 1, 64, true, 4>), dim3(grid), dim3(256), 0, s, p);
    } else if (p.taps == 9) {
        static const bool m16t = getenv("IR_NO_MFMA16") == nullptr;
        if (k64 && m16t) hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 9, 64, true>), dim3(grid), dim3(256), 0, s, p);
        else if (k64) hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 9, 64>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 9, 32>), dim3(grid), dim3(256), 0, s, p);
    } else {
        static const bool m16 = getenv("IR_NO_MFMA16") == nullptr;  // 16x16x32 MFMA form of the 128x128 GEMM (knob: A/B against 32x32x16)
        if (k64 && m16) hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 1, 64, true>), dim3(grid), dim3(256), 0, s, p);
        else if (k64) hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 1, 64>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 1, 32>), dim3(grid), dim3(256), 0, s, p);
    }
launched:
    if (ks > 1) {
        const long nv = (long)pin.M * (pin.Cout_pad / 4);
        const unsigned fg = (unsigned)std::min<long>((nv + 255) / 256, 4096);
        hipLaunchKernelGGL(splitk_finish_kernel, dim3(fg), dim3(256), 0, s, pin, pin.ks_ws, ks);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Fused GroupNorm statistics: which kernel would run and how many pixel tiles per image it has (0: cannot fuse).
static bool takes_halo(const IGemmParams& p) {
    static const bool no_halo = getenv("IR_NO_HALO") != nullptr;  // experiment knob
    return p.taps == 9 && p.stride == 1 && p.pad == 1 && (p.Cin & 63) == 0 && !p.force_generic && !no_halo && p.Cout_pad % 64 == 0;
}
static bool takes_halo_pp(const IGemmParams& p) {  // the 8-wave ping-pong variant: 16 x 16 patches x 128 channels
    static const bool no_pp = getenv("IR_NO_CONV_PP") != nullptr;  // experiment knob
    return takes_halo(p) && !p.fp8 && p.Cout_pad % 128 == 0 && p.Cin >= 128 && !no_pp && !g_ir_plain_kernels;  // measured with the 16x16x32 MFMAs: +8 % at 512 channels,
                                                                               // +7 % at 256, +2 % at 128 over the 4-wave kernel
}
int ir_igemm_gn_chunks(const IGemmParams& p) {
    if (p.gn_cpg < 4 || (p.gn_cpg & 3) || p.Cout % p.gn_cpg || p.Cout_pad % 64 || (p.Cout & 3) || p.NB <= 0) return 0;
    if ((p.Cout_pad % 128 == 0 ? 128 : 64) % p.gn_cpg) return 0;
    if (p.up2x2) return (ir_conv_s1_up2x2_takes(p) && p.gn_cpg <= 32 && !(p.gn_cpg & (p.gn_cpg - 1))) ? ir_conv_s1_up2x2_tiles(p) : 0;
    if (ir_conv_s1_takes(p) || ir_conv_s1_fp8_takes(p)) return (p.gn_cpg <= 32 && !(p.gn_cpg & (p.gn_cpg - 1))) ? ir_conv_s1_tiles(p) : 0;   // its reduction wants 4, 8, 16 or 32 channels per group
    if (takes_halo_pp(p)) return ((p.Ho + 15) / 16) * ((p.Wo + 15) / 16);
    if (takes_halo(p)) return ((p.Ho + 7) / 8) * ((p.Wo + 15) / 16);
    if (p.M % p.NB) return 0;
    const long hw = p.taps == 9 ? (long)p.Ho * p.Wo : p.M / p.NB;
    if (hw % 128) return 0;  // a 128-row tile must not straddle two images
    return (int)(hw / 128);
}

static bool igemm_vec(const IGemmParams& p) {
    return !((p.out_cs & 3) || (p.res && (p.res_cs & 3)) || (p.out2 && (p.out2_cs & 3)) || (reinterpret_cast<uintptr_t>(p.res) & 15) ||
             (reinterpret_cast<uintptr_t>(p.out2) & 7) || (p.gate && (p.gate_stride & 3)));
}
int ir_igemm_kernel_id(const IGemmParams& pin) {
    IGemmParams p = pin;
    p.vec = igemm_vec(p);
    if (p.up2x2) return ir_conv_s1_up2x2_takes(p) ? 0 : 3;
    if (p.ks_ws && ir_igemm_splitk(p) > 1) return 4;
    if (ir_conv64_takes(p)) return 5;
    if (ir_conv_s1_takes(p) || ir_conv_s1_fp8_takes(p)) return 0;
    if (takes_halo_pp(p)) return 1;
    if (takes_gemm_pp(p)) return 2;
    if (takes_halo(p)) return 3;
    return 4;
}

// Host launcher. Returns 0 or a negative error code; validates every shape assumption the kernel makes.
int ir_launch_igemm(const IGemmParams& pin, hipStream_t s) {
    IGemmParams p = pin;
    if (p.M <= 0) return 0;
    if (p.up2x2) {   // phase weights are no 9-tap weights: only the two phase kernels may run them
        if ((reinterpret_cast<uintptr_t>(p.in) & 15) || (reinterpret_cast<uintptr_t>(p.wgt) & 15) || (reinterpret_cast<uintptr_t>(p.out) & 15) || !p.out) return -7;
        if (ir_conv_s1_up2x2_takes(p)) return ir_launch_conv_s1_up2x2(p, s);
        return takes_halo_up2x2(p) ? launch_halo_up2x2(p, s) : -15;
    }
    if ((p.nrm_scale || p.nrm_shift) && !ir_conv_s1_norm_takes(p)) return -16;   // only conv_halo_s1_kernel<0, 9, NORM> normalises its input: the caller asks first
    if (p.taps != 1 && p.taps != 9) return -2;
    if (p.Cin <= 0 || (p.Cin & 31) || (p.in_cs & 7) || p.in_cs < p.Cin) return -3;
    if (p.Cout <= 0 || p.Cout > p.Cout_pad || (p.Cout_pad & 31)) return -4;
    if (p.wgt_rs < (long)p.taps * p.Cin || (p.wgt_rs & 7)) return -10;
    if (p.taps == 9 && (long)p.Cin * 2 + 64 > (long)sizeof(uint4) * 4096) return -11;  // zero page must cover one tap's channels
    p.vec = igemm_vec(p);
    if (!p.out) return -5;
    if (p.gate && p.gate_stride != 0) return -6;  // one gate row per launch (single timestep per batch)
    if ((reinterpret_cast<uintptr_t>(p.in) & 15) || (reinterpret_cast<uintptr_t>(p.wgt) & 15) ||
        (reinterpret_cast<uintptr_t>(p.out) & 15))
        return -7;
    if (p.taps == 9) {
        if (p.stride != 1 && p.stride != 2) return -8;
        if ((long)p.NB * p.Ho * p.Wo != p.M) return -9;
        if (p.H <= 0 || p.W <= 0) return -9;
    }
    if (p.gn_part && (!p.vec || p.gn_chunks <= 0 || p.gn_chunks != ir_igemm_gn_chunks(p))) return -13;
    if (p.fp8 && (!takes_halo(p) || !p.gate || p.act != IR_ACT_NONE)) return -14;  // fp8 operands: stride-1 3x3 halo kernel only
    if (p.ks_ws && ir_igemm_splitk(p) > 1) {   // small-M launch of a caller that allows split-K: generic kernel, K split over blockIdx.y
        if (p.Cout_pad % 128 == 0) return launch_cfg<128, 128, 2, 2>(p, s);
        if (p.Cout_pad % 64 == 0) return launch_cfg<128, 64, 2, 2>(p, s);
        return launch_cfg<128, 32, 4, 1>(p, s);
    }
    if (ir_conv64_takes(p)) return ir_launch_conv64(p, s);
    if (ir_conv_s1_takes(p)) return ir_launch_conv_s1(p, s);
    if (ir_conv_s1_fp8_takes(p)) return ir_launch_conv_s1_fp8(p, s);
    if (takes_halo_pp(p)) return launch_halo_pp(p, s);
    if (p.vt_out && !ir_igemm_writes_vt(p)) p.vt_out = nullptr;   // only gemm_pp_kernel's bf16 form writes the transposed copy: the caller asked first
    if (takes_gemm_pp(p)) return launch_gemm_pp(p, s);
    if (takes_halo(p)) {
        if (p.Cout_pad % 128 == 0) return launch_halo<128>(p, s);
        return launch_halo<64>(p, s);
    }
    if (p.Cout_pad % 128 == 0) return launch_cfg<128, 128, 2, 2>(p, s);
    if (p.Cout_pad % 64 == 0) return launch_cfg<128, 64, 2, 2>(p, s);
    return launch_cfg<128, 32, 4, 1>(p, s);
}
