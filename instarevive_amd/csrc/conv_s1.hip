// conv_halo_s1_kernel: the stride-1 3x3 convolution of the VAE (reference ldm/modules/diffusionmodules/model.py:57-61,102-116) as ONE wave
// per SIMD with the whole register file - the structure that took the attention kernels (attn_d512.hip) from 0.3 to 0.5+ of the MFMA
// peak - instead of the two-waves-per-SIMD ping-pong of conv_halo_pp_kernel (igemm.hip).
//
// A 256-thread workgroup (4 waves, one per SIMD, one workgroup per CU) computes a 16 x 32 patch of output pixels for 128 output
// channels. Wave w owns patch rows 4w .. 4w+3: 8 pixel fragments of 16 pixels x 8 channel fragments of 16 = 64 accumulator tiles of
// v_mfma_f32_16x16x32_bf16 = all 256 AGPRs, addressed literally from inline asm. Against the ping-pong kernel (64 x 64 per wave) that
// is a quarter of the LDS fragment bytes per MFMA (16 ds_read_b128 per 64 MFMAs), a 1.20x instead of 1.27x halo, and no second
// workgroup barrier per step.
// The reduction runs over 32-channel chunks: the 18 x 34 halo of a chunk (612 pixels x 64 B = 39 KB) is brought in ONCE by LDS-DMA
// (double-buffered, issued during the previous chunk) and all nine taps read their pixel fragments from it with a tap offset on the LDS
// address; the weight tile of a (chunk, tap) step (128 rows x 64 B) rides a ring of four. Per step every wave issues ONE pinned stream
// of 64 MFMAs (1024 matrix cycles); the 16 fragment reads of the NEXT step and this wave's 2-4 LDS-DMA pieces sit in the MFMA
// shadows, so after the single workgroup barrier of a step the next stream starts from registers.
// Ordering: tile s+4 (ring slot s % 4) and the next chunk's halo pieces are issued after the barrier that opens step s - every wave
// has finished reading tile s and the other halo buffer by then - halo pieces first, weight pieces second, so that one counted vmcnt
// before the next barrier (everything but the pieces of the last two steps) covers the weight tile of step s+1, and before the barrier
// that opens tap 8 the whole next halo. Past the end the last tile / chunk is re-read into the free slot: the counts stay uniform and
// the stream branch-free.
// Operand roles: A = weights (rows m = output channels), B = pixels (columns n = 16 consecutive pixels of a patch row), so a lane holds
// 4 consecutive channels of one pixel per tile and the epilogue writes float4s into a wave-private [32 pixels][128 channels] fp32 slab,
// reads rows back as 8-channel vectors, adds the residual, rounds to bf16, stores 16 bytes per lane and accumulates the fused GroupNorm
// statistics of the values as stored (same contract as igemm_epilogue).
// LDS images: pixel / weight rows are 64 B (4 chunks of 16 B); chunk c of halo column hx is stored at slot c ^ 2*((hx >> 2) & 1) and
// chunk c of weight row r at c ^ 2*((r >> 2) & 1), applied on the LDS-DMA source side and on the read side: the 16 lanes of every
// ds_read_b128 lane group then cover all 64 banks for each tap offset kx = 0, 1, 2 (searched exhaustively over the 4^4 key tables).
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "agpr256.h"
#include "common.h"
#include "kernels.h"

namespace cs1 {
constexpr int TH = 16, TW = 32, HWD = TW + 2, HP = (TH + 2) * HWD;   // 612 halo pixels
constexpr int BK = 32, ROWB = 64;
constexpr int H_Q = (HP + 15) / 16;            // 39 LDS-DMA pieces of 16 pixels x 64 B
constexpr int HALO_BYTES = H_Q * 1024;         // 39 936
constexpr int H_I = 10;                        // halo pieces per wave and chunk (piece q = wave + 4 i, clamped to the last)
constexpr int BN = 128, WT_BYTES = BN * ROWB;  // 8 192: 8 pieces of 16 rows
constexpr int NSB = 4;
constexpr int W_OFF = 2 * HALO_BYTES;          // 79 872
constexpr int LDS_MAIN = W_OFF + NSB * WT_BYTES;   // 112 640
constexpr int SROW = 132;                      // slab row stride in floats (128 + 4)
constexpr int SLAB = 32 * SROW * 4;            // 16 896 B per wave
constexpr int RED_OFF = 4 * SLAB;
constexpr int LDS_EP = RED_OFF + 4 * 64 * 16;
constexpr int LDS_BYTES = LDS_MAIN > LDS_EP ? LDS_MAIN : LDS_EP;
// halo pieces issued in the step of tap t (10 per chunk, all before tap 6)
constexpr int nh(int t) { return t < 4 ? 2 : (t < 6 ? 1 : 0); }
constexpr int nh_first(int t) { return t < 4 ? 2 * t : (t < 6 ? 4 + t : 0); }
constexpr int hkey(int hx) { return ((hx >> 2) & 1) << 1; }
}  // namespace cs1

__device__ uint4 g_zero_page_s1[4096];   // 64 KB of zeros: padding taps read from here, the LDS-DMA never needs a mask

typedef __attribute__((address_space(3))) void* cs1_lds_t;
IR_DEVINL void cs1_glds16(const void* g, cs1_lds_t l) { __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0); }
template <int LO>
IR_DEVINL void cs1_mfma(bf16x8 w, bf16x8 px) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(px), "n"(LO), "n"(LO + 3));
}
template <int I>
IR_DEVINL float cs1_acc_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "n"(I));
    return x;
}

template <int UP>
__global__ __launch_bounds__(256, 1) void conv_halo_s1_kernel(IGemmParams p, int tiles_y, int tiles_x) {
    using namespace cs1;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];   // halo[0] | halo[1] | W ring of 4 ; epilogue: slabs | red
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int c16 = lane & 15, kq = lane >> 4;

    const int NT = p.Cout_pad / BN;
    const int MT = p.NB * tiles_y * tiles_x;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, jb = bid >> 3;
    const int mt = (jb / NT) * 8 + xcd, nt = jb % NT;   // an XCD runs the channel tiles of one patch back to back (halo re-read from its L2)
    if (mt >= MT) return;
    const int n0 = nt * BN;
    const int img = mt / (tiles_y * tiles_x), trem = mt - img * tiles_y * tiles_x;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int Hc = UP ? 2 * p.H : p.H, Wc = UP ? 2 * p.W : p.W;   // conv-input (== output) extent
    const int chunks = p.Cin / BK;                                  // even (launcher)

    asm volatile(".set ir_cs1_i, 0\n\t.rept 256\n\tv_accvgpr_write_b32 a[ir_cs1_i], 0\n\t.set ir_cs1_i, ir_cs1_i + 1\n\t.endr" ::: IR_AGPR256_CLOBBERS);

    // ---- LDS-DMA sources. Halo piece q covers halo pixels 16q .. 16q+15: lane l -> pixel 16q + (l >> 2), LDS slot l & 3.
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_s1);
    const bf16_t* h_ptr[H_I];
#pragma unroll
    for (int i = 0; i < H_I; ++i) {
        const int q = min(wu + 4 * i, H_Q - 1);
        const int hp = q * 16 + (lane >> 2);
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const int cy = oy0 + hy - 1, cx = ox0 + hx - 1;
        const bool ok = hp < HP && cy >= 0 && cy < Hc && cx >= 0 && cx < Wc;
        const int iy = min(max(cy, 0), Hc - 1) >> UP, ix = min(max(cx, 0), Wc - 1) >> UP;
        const bf16_t* src = p.in + (((long)img * p.H + iy) * p.W + ix) * p.in_cs;
        h_ptr[i] = (ok ? src : zero) + (((lane & 3) ^ hkey(hx)) << 3);
    }
    const bf16_t* w_ptr[2];   // weight pieces wave, wave + 4: rows 16 j + (l >> 2)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wu + 4 * i) * 16 + (lane >> 2);
        w_ptr[i] = p.wgt + (long)(n0 + row) * p.wgt_rs + (((lane & 3) ^ hkey(row)) << 3);
    }
    auto halo_issue = [&](auto ic, int chunk, int buf) {   // piece i of this wave: channels chunk * 32 .. into halo buffer buf
        constexpr int i = decltype(ic)::value;
        const int q = min(wu + 4 * i, H_Q - 1);
        cs1_glds16(h_ptr[i] + chunk * BK, (cs1_lds_t)(smem + buf * HALO_BYTES + q * 1024));
    };
    auto w_issue = [&](int chunk, int tap, int slot) {
        const int koff = tap * p.Cin + chunk * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) cs1_glds16(w_ptr[i] + koff, (cs1_lds_t)(smem + W_OFF + slot * WT_BYTES + (wu + 4 * i) * 1024));
    };

    // ---- fragment read addresses. Pixel fragment (patch row 4w + a, half mx) of tap (ky, kx): halo pixel (4w + a + ky, 16 mx + kx + c16),
    // chunk kq; the row term is an immediate. Weight fragment ct: row 16 ct + c16, chunk kq; ct * 1024 is an immediate.
    const uint32_t lds0 = lds_addr(smem);
    uint32_t hrd[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int mx = 0; mx < 2; ++mx) {
            const int hx = 16 * mx + kx + c16;
            hrd[kx][mx] = lds0 + ((4 * wid) * HWD + hx) * ROWB + ((kq ^ hkey(hx)) << 4);
        }
    const uint32_t wrd = lds0 + W_OFF + c16 * ROWB + ((kq ^ hkey(c16)) << 4);

    bf16x8 fw[2][8], fp[2][8];   // [set][channel fragment] / [set][pixel fragment = a * 2 + mx]

    // ---- prologue: halo of chunk 0, weight tiles of steps 0..3, fragments of step 0
    [&]<int... I>(std::integer_sequence<int, I...>) { (halo_issue(std::integral_constant<int, I>{}, 0, 0), ...); }(std::make_integer_sequence<int, H_I>{});
#pragma unroll
    for (int t = 0; t < NSB; ++t) w_issue(0, t, t);
    wait_dma();
    __syncthreads();
    [&]<int... R>(std::integer_sequence<int, R...>) {
        ([&] {
            if constexpr (R < 8) fw[0][R] = lds_read16<R * 1024>(wrd);
            else fp[0][R - 8] = lds_read16<((R - 8) >> 1) * HWD * ROWB>(hrd[0][(R - 8) & 1]);
        }(), ...);
    }(std::make_integer_sequence<int, 16>{});
    wait_lds<0>();

    auto step = [&](auto tc, auto setc, int c) {
        constexpr int T = decltype(tc)::value, SET = decltype(setc)::value;
        constexpr int TNX = (T + 1) % 9, KXN = TNX % 3, KYN = TNX / 3;
        const int s = c * 9 + T;
        const uint32_t hb = (uint32_t)((T == 8 ? (c + 1) : c) & 1) * HALO_BYTES;
        const uint32_t ha0 = hrd[KXN][0] + hb, ha1 = hrd[KXN][1] + hb;
        const uint32_t wa = wrd + (uint32_t)((s + 1) & 3) * WT_BYTES;
        // weight tile s + 4 = (chunk cw, tap tw), clamped to the last tile
        constexpr int TW4 = (T + 4) % 9;
        int cw = c + (T + 4 >= 9 ? 1 : 0), tw = TW4;
        if (cw >= chunks) { cw = chunks - 1; tw = 8; }
        const int ch_next = min(c + 1, chunks - 1);
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                constexpr int PT = I >> 3, CT = I & 7;
                if constexpr ((I & 3) == 0) {   // one fragment of the next step per four MFMAs, into the other set
                    constexpr int R = I >> 2;
                    if constexpr (R < 8) fw[SET ^ 1][R] = lds_read16<R * 1024>(wa);
                    else fp[SET ^ 1][R - 8] = lds_read16<(((R - 8) >> 1) + KYN) * HWD * ROWB>(((R - 8) & 1) ? ha1 : ha0);
                }
                __builtin_amdgcn_sched_barrier(0);
                cs1_mfma<4 * I>(fw[SET][CT], fp[SET][PT]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (I == 1 && nh(T) > 0) {
                    halo_issue(std::integral_constant<int, nh_first(T)>{}, ch_next, (c + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (I == 5 && nh(T) > 1) {
                    halo_issue(std::integral_constant<int, (nh(T) > 1 ? nh_first(T) + 1 : 0)>{}, ch_next, (c + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (I == 9) {
                    w_issue(cw, tw, s & 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }(), ...);
        }(std::make_integer_sequence<int, 64>{});
        wait_lds<0>();
        // everything but the pieces of this step and the previous one has landed: the weight tile of step s + 2 (read during step s + 1)
        // and, before tap 8, the next chunk's halo (its last piece is issued at tap 5)
        wait_vm<4 + nh((T + 8) % 9) + nh(T)>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int c = 0; c < chunks; c += 2) {
        [&]<int... U>(std::integer_sequence<int, U...>) { (step(std::integral_constant<int, U>{}, std::integral_constant<int, (U & 1)>{}, c), ...); }(std::make_integer_sequence<int, 9>{});
        [&]<int... U>(std::integer_sequence<int, U...>) { (step(std::integral_constant<int, U>{}, std::integral_constant<int, ((U + 1) & 1)>{}, c + 1), ...); }(std::make_integer_sequence<int, 9>{});
    }

    // ---- epilogue
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // the last MFMA results -> v_accvgpr_read
    wait_dma();        // the re-read pieces past the end must have landed before the slabs overlay the ring
    __syncthreads();
    float* slab = reinterpret_cast<float*>(smem + wid * SLAB);
    const int co8 = (lane & 15) * 8, xq = lane >> 4;
    f32x4 bias4[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
        bias4[ct] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + 16 * ct + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float osc = p.out_scale;
    float sA = 0.f, qA = 0.f, sB = 0.f, qB = 0.f;   // GroupNorm partials of channels co8 .. +3 and co8+4 .. +7 over this lane's pixels
    const bf16_t* resp = reinterpret_cast<const bf16_t*>(p.res);
    bf16_t* outp = reinterpret_cast<bf16_t*>(p.out);
    auto pass = [&](auto ac) {
        constexpr int A = decltype(ac)::value;
        const int oy = oy0 + 4 * wid + A;
        long pix[8];
        bool ok[8];
        uint4 rr[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int ox = ox0 + 4 * it + xq;
            ok[it] = oy < p.Ho && ox < p.Wo;
            pix[it] = ((long)img * p.Ho + min(oy, p.Ho - 1)) * p.Wo + min(ox, p.Wo - 1);
            rr[it] = make_uint4(0, 0, 0, 0);
        }
        if (resp) {   // all residual loads of the pass in flight before the transposes
#pragma unroll
            for (int it = 0; it < 8; ++it) rr[it] = *reinterpret_cast<const uint4*>(resp + pix[it] * p.res_cs + n0 + co8);
        }
        [&]<int... J>(std::integer_sequence<int, J...>) {
            ([&] {
                constexpr int MX = J >> 3, CT = J & 7, LO = 4 * ((A * 2 + MX) * 8 + CT);
                f32x4 v = f32x4{cs1_acc_read<LO>(), cs1_acc_read<LO + 1>(), cs1_acc_read<LO + 2>(), cs1_acc_read<LO + 3>()};
                v = (v + bias4[CT]) * osc;
                *reinterpret_cast<f32x4*>(&slab[(16 * MX + c16) * SROW + 16 * CT + 4 * kq]) = v;
            }(), ...);
        }(std::make_integer_sequence<int, 16>{});
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        f32x4 lo[8], hi[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            lo[it] = *reinterpret_cast<const f32x4*>(&slab[(4 * it + xq) * SROW + co8]);
            hi[it] = *reinterpret_cast<const f32x4*>(&slab[(4 * it + xq) * SROW + co8 + 4]);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const f32x4 a = lo[it] + f32x4{bflo(rr[it].x), bfhi(rr[it].x), bflo(rr[it].y), bfhi(rr[it].y)};
            const f32x4 b = hi[it] + f32x4{bflo(rr[it].z), bfhi(rr[it].z), bflo(rr[it].w), bfhi(rr[it].w)};
            const uint4 pk = make_uint4(pack2bf_valu(a[0], a[1]), pack2bf_valu(a[2], a[3]), pack2bf_valu(b[0], b[1]), pack2bf_valu(b[2], b[3]));
            if (ok[it]) {
                *reinterpret_cast<uint4*>(outp + pix[it] * p.out_cs + n0 + co8) = pk;
                if (p.gn_part) {   // statistics of the values as stored (bf16-rounded)
                    const float a0 = bflo(pk.x), a1 = bfhi(pk.x), a2 = bflo(pk.y), a3 = bfhi(pk.y);
                    const float b0 = bflo(pk.z), b1 = bfhi(pk.z), b2 = bflo(pk.w), b3 = bfhi(pk.w);
                    sA += (a0 + a1) + (a2 + a3);
                    qA += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                    sB += (b0 + b1) + (b2 + b3);
                    qB += (b0 * b0 + b1 * b1) + (b2 * b2 + b3 * b3);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    pass(std::integral_constant<int, 0>{});
    pass(std::integral_constant<int, 1>{});
    pass(std::integral_constant<int, 2>{});
    pass(std::integral_constant<int, 3>{});
    if (p.gn_part) {
        // Fixed-order workgroup reduction (no atomics, bit-identical run to run): unit u = 4 channels; lane (L = lane & 15) holds units 2L, 2L+1
        float* red = reinterpret_cast<float*>(smem + RED_OFF);   // [wave][lane][4]
        *reinterpret_cast<f32x4*>(&red[(wid * 64 + lane) * 4]) = f32x4{sA, qA, sB, qB};
        __syncthreads();
        const int upg = p.gn_cpg >> 2;          // units per group
        const int groups = BN / p.gn_cpg;
        if (tid < groups) {
            float a = 0.f, b = 0.f;
            for (int w = 0; w < 4; ++w)
                for (int xr = 0; xr < 4; ++xr)
                    for (int k = 0; k < upg; ++k) {
                        const int u = tid * upg + k;
                        const float* e = &red[((w * 64) + xr * 16 + (u >> 1)) * 4 + (u & 1) * 2];
                        a += e[0];
                        b += e[1];
                    }
            const int G = p.Cout / p.gn_cpg, g = n0 / p.gn_cpg + tid;
            float* dst = p.gn_part + ((long)img * p.gn_chunks + trem) * 2 * G;
            dst[g] = a;
            dst[G + g] = b;
        }
    }
}

// Which launches take this kernel (everything else of the halo family stays with conv_halo_pp_kernel / conv_halo_kernel): plain bf16
// NHWC in and out, 128-channel output tiles, no activation / gate / second output, a bf16 residual at most, and enough patches to fill
// the chip (one 512-pixel patch per CU and round).
bool ir_conv_s1_takes(const IGemmParams& p) {
    static const bool off = getenv("IR_NO_CONV_S1") != nullptr;   // experiment knob
    if (off || g_ir_plain_kernels || p.fp8 || p.force_generic) return false;
    if (p.taps != 9 || p.stride != 1 || p.pad != 1 || (p.Cin & 63) || p.Cin < 128) return false;
    if (p.Cout != p.Cout_pad || p.Cout_pad % 128) return false;
    if (p.act != IR_ACT_NONE || p.gate || p.out2 || p.out_f32) return false;
    if (p.res && (p.res_f32 || p.res_mod > 0 || (p.res_cs & 7) || (reinterpret_cast<uintptr_t>(p.res) & 15))) return false;
    if ((p.out_cs & 7) || (reinterpret_cast<uintptr_t>(p.out) & 15)) return false;
    if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15)) return false;
    // per IMAGE, not per launch: which kernel runs (and with it the summation order) must not depend on an image's batch neighbours
    const long tiles = (long)((p.Ho + 15) / 16) * ((p.Wo + 31) / 32) * (p.Cout_pad / 128);
    return tiles >= 192;
}
int ir_conv_s1_tiles(const IGemmParams& p) { return ((p.Ho + 15) / 16) * ((p.Wo + 31) / 32); }

int ir_launch_conv_s1(const IGemmParams& p, hipStream_t s) {
    if (!ir_conv_s1_takes(p)) return -2;
    if (p.gn_part && (p.gn_cpg < 4 || (p.gn_cpg & 3) || 128 % p.gn_cpg || p.gn_chunks != ir_conv_s1_tiles(p))) return -13;
    const int tiles_y = (p.Ho + 15) / 16, tiles_x = (p.Wo + 31) / 32;
    const long MT = (long)p.NB * tiles_y * tiles_x, NT = p.Cout_pad / 128;
    const long grid = ((MT + 7) / 8) * 8 * NT;
    if (grid > 0x7fffffffL) return -12;
    if (p.up) hipLaunchKernelGGL((conv_halo_s1_kernel<1>), dim3((unsigned)grid), dim3(256), 0, s, p, tiles_y, tiles_x);
    else hipLaunchKernelGGL((conv_halo_s1_kernel<0>), dim3((unsigned)grid), dim3(256), 0, s, p, tiles_y, tiles_x);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
