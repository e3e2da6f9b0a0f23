// VAE mid-block attention (single head, head dim 512; reference ldm/modules/diffusionmodules/model.py:181-205) on fp8 (OCP e4m3) MFMA
// operands for gfx950 - the third fp8 kernel of BASELINE.json configs[4] next to conv_s1_fp8.hip and attn_fp8.hip, same number formats
// and block-scale mechanics as attn_fp8.hip (see there and attn_fp8_common.h), same structure as flash_attn_d512_v2_kernel
// (attn_d512.hip): ONE wave per SIMD with the whole register file, a wave owns 32 queries and ALL 512 output dims.
//
//   S^T = K Q^T     16 x v_mfma_scale_f32_32x32x64_f8f6f4 per 64-key tile (2 key halves x 8 k-steps of 64 d): A = K8 rows from LDS (one
//                   exponent per tile, attn_d512_fp8_prep_kernel), B = Q8 (the lane's own query row: 8 k-steps x 8 registers, one exponent
//                   per query, pre-multiplied by scale * log2 e); -m (the fixed softmax reference) is the C operand of the first MFMA of a chain
//   O^T += V^T P^T  16 MFMAs (one per 32 output dims, k = the tile's 64 keys): A = V8^T rows from LDS (per-tile exponent, keys stored in
//                   accumulator order so that the lane's 2 x 16 score registers ARE its B operand), B = P8 with one exponent per (query, tile)
//                   from the maximum over the lane pair; O^T in all 256 AGPRs, addressed literally.
// 32 MFMAs x 64 cycles per tile against 128 x 32 for the bf16 kernel (64 keys): half the matrix time. The softmax of tile t + 1 (32 v_exp,
// the block maximum, 16 scaled converts, the row sum - the denominator is summed on the VALU: there is no spare V^T row for a ones row) rides
// behind the PV MFMAs of tile t, the 65 LDS-DMA pieces of K(t+2) / V^T(t+1) behind the QK^T MFMAs; one barrier per tile, two ring slots of
// 65 KB (a tile image = its LDS image). Fixed softmax reference (first tile's maximum + headroom) with an overflow flag; the bf16
// rescaling kernel behind it recomputes when the flag is raised.
#include "common.h"
#include "kernels.h"
#include "agpr256.h"
#include "attn_fp8_common.h"
#include <type_traits>
#include <utility>

namespace f8d {
constexpr int D = 512, TK = 64, NKS = D / 64, NDT = D / 32;
constexpr int KROW = 528;                       // 512 e4m3 bytes + 16: rows 132 dwords apart -> 16 lanes of a ds_read_b128 cover all banks
constexpr int K_BYTES = TK * KROW;              // 33 792 = 33 DMA pieces
constexpr int V_BYTES = D * TK;                 // 32 768 = 32 pieces; chunk c (16 keys) of row d at slot c ^ ((d >> 2) & 3)
constexpr int TILE_BYTES = K_BYTES + V_BYTES;   // 66 560
constexpr int PIECES = TILE_BYTES / 1024;       // 65
constexpr int SCALE_OFF = 512;                  // pad bytes of K row 0: dword {K exponent byte | V exponent byte << 8}
constexpr int LDS_BYTES = 2 * TILE_BYTES;       // 133 120 (the epilogue stages 128 query rows of 1040 B in the same memory)
constexpr int OS = 512 + 8;                     // staging row stride (elements)
constexpr float MARGIN = 24.0f;
}  // namespace f8d

// K, V [B][T][512] bf16 (token stride rs) -> tile images [B][T/64][66 560 B]: K8 rows (528 B) | V8^T rows (64 B, keys in accumulator
// order: logical byte 32 h + j of a row <-> key 32 (j >> 4) + (j & 3) + 8 ((j & 15) >> 2) + 4 h, chunks XOR-swizzled)
__global__ __launch_bounds__(256) void attn_d512_fp8_prep_kernel(const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, uint8_t* __restrict__ tiles,
                                                                 long kv_bs, int rs, int NT) {
    using namespace f8d;
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    bf16_t (*Ts)[D + 8] = reinterpret_cast<bf16_t (*)[D + 8]>(psm);   // one 64 x 520 staging tile, K first, then V
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tile = blockIdx.x, b = blockIdx.y;
    uint8_t* dst = tiles + ((long)b * NT + tile) * TILE_BYTES;
    int bytes[2];
    for (int which = 0; which < 2; ++which) {
        const bf16_t* src = (which ? v : k) + (long)b * kv_bs + (long)tile * 64 * rs;
        float mx = 0.f;
        for (int c = tid; c < 64 * 64; c += 256) {
            const int row = c >> 6, ch = c & 63;
            const uint4 a = *reinterpret_cast<const uint4*>(src + (long)row * rs + ch * 8);
            *reinterpret_cast<uint4*>(&Ts[row][ch * 8]) = a;
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) mx = fmaxf(mx, fmaxf(fabsf(bflo(aw[i])), fabsf(bfhi(aw[i]))));
        }
        mx = wave_max(mx);
        if (lane == 0) red[wid] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const int bb = f8_block_byte(mx);
        bytes[which] = bb;
        const float sc = __builtin_bit_cast(float, (uint32_t)bb << 23);
        if (which == 0) {
            // K8: 64 rows x 33 chunks of 16 B (chunk 32 = the row's padding; row 0's carries the exponents, written below)
            for (int c = tid; c < 64 * 32; c += 256) {
                const int row = c >> 5, ch = c & 31;
                uint32_t w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16_t* s = &Ts[row][ch * 16 + 4 * i];
                    w[i] = f8_cvt2<false>(0u, bf2f(s[0]), bf2f(s[1]), sc);
                    w[i] = f8_cvt2<true>(w[i], bf2f(s[2]), bf2f(s[3]), sc);
                }
                *reinterpret_cast<uint4*>(dst + row * KROW + ch * 16) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        } else {
            for (int c = tid; c < D * 4; c += 256) {
                const int d = c >> 2, pc = c & 3, lc = pc ^ ((d >> 2) & 3);
                uint32_t w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int L = 16 * lc + 4 * i + e, h = L >> 5, j = L & 31;
                        const int key = 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * h;
                        x[e] = bf2f(Ts[key][d]);
                    }
                    w[i] = f8_cvt2<false>(0u, x[0], x[1], sc);
                    w[i] = f8_cvt2<true>(w[i], x[2], x[3], sc);
                }
                *reinterpret_cast<uint4*>(dst + K_BYTES + d * 64 + pc * 16) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        __syncthreads();   // the staging tile is free again
    }
    if (tid < 64) {   // the rows' padding chunks; row 0's first dword = the two exponent bytes
        const uint32_t e = tid == 0 ? ((uint32_t)bytes[0] | ((uint32_t)bytes[1] << 8)) : 0u;
        *reinterpret_cast<uint4*>(dst + tid * KROW + SCALE_OFF) = make_uint4(e, 0u, 0u, 0u);
    }
}

struct AttnD512F8Params {
    const bf16_t* q;
    const uint8_t* tiles;
    bf16_t* o;
    long q_bs, o_bs, tiles_bs;
    int T, rs, o_rs;
    float scale_log2;
    int* ovf_flag;
};

// The MFMAs are inline asm (S^T in arch VGPRs the VALU reads, O^T in literal AGPRs). Hazards hipcc cannot see are covered by construction:
// MFMA results are read by the VALU at least three MFMAs (about 200 cycles) behind their producer or behind explicit s_nops; VALU-written
// operands (P8, exponent bytes) are half a tile old when an MFMA reads them; all MFMAs are of ONE opcode, whose back-to-back accumulation
// the hardware interlocks; operand registers are pinned with keep() past the VALU work that follows their MFMA.
// knock-out builds (diagnostic, results wrong by design): -DIR_KO_D8=1 no end-of-tile DMA wait, 2 no softmax items, 4 no LDS-DMA in the loop, 8 no MFMAs
#ifndef IR_KO_D8
#define IR_KO_D8 0
#endif
IR_DEVINL void d8_mfma_c(f32x16& s, i32x8 a, i32x8 b, const f32x16& c, int sa, int sb) {   // s = A8 B8 + c (fresh destination)
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel_hi:[0,0,0]" : "=&v"(s) : "v"(a), "v"(b), "v"(c), "v"(sa), "v"(sb));
}
IR_DEVINL void d8_mfma_acc(f32x16& s, i32x8 a, i32x8 b, int sa, int sb) {                  // s += A8 B8
    if constexpr (IR_KO_D8 & 8) { asm volatile("" : "+v"(s) : "v"(a), "v"(b), "v"(sa), "v"(sb)); return; }
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(s) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}
template <int LO>
IR_DEVINL void d8_mfma_o(i32x8 a, i32x8 b, int sa, int sb) {                                // a[LO : LO + 15] += A8 B8
    if constexpr (IR_KO_D8 & 8) { asm volatile("" ::"v"(a), "v"(b), "v"(sa), "v"(sb)); return; }
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 a[%c4:%c5], %0, %1, a[%c4:%c5], %2, %3 op_sel_hi:[0,0,0]" ::"v"(a), "v"(b), "v"(sa), "v"(sb), "n"(LO), "n"(LO + 15));
}
template <int I>
IR_DEVINL float d8_acc_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "n"(I));
    return x;
}

__global__ __launch_bounds__(256, 1) void flash_attn_d512_fp8_kernel(AttnD512F8Params p) {
    using namespace f8d;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 128 + wid * 32;
    const long b = blockIdx.y;
    const int NT = p.T >> 6;
    const uint8_t* tp = p.tiles + b * p.tiles_bs + lane * 16;

    asm volatile(".set ir_d8_i, 0\n\t.rept 256\n\tv_accvgpr_write_b32 a[ir_d8_i], 0\n\t.set ir_d8_i, ir_d8_i + 1\n\t.endr" ::: IR_AGPR256_CLOBBERS);

    // LDS-DMA pieces of 1 KB; a wave issues pieces wu + 4 i. K part = pieces 0..32, V^T part = 33..64.
    auto piece = [&](int tile, int idx, int slot) {
        f8_glds16(tp + (long)min(tile, NT - 1) * TILE_BYTES + idx * 1024, (f8_lds_t)(smem + slot * TILE_BYTES + idx * 1024));
    };
    // prologue: all of tile 0, the K part of tile 1 (past the end the last tile is fetched again: finite bytes nobody consumes)
#pragma unroll
    for (int i = 0; i < 17; ++i) piece(0, min(wu + 4 * i, PIECES - 1), 0);
#pragma unroll
    for (int i = 0; i < 9; ++i) piece(1, min(wu + 4 * i, 32), 1);

    // ---- Q8: the lane's query row, d = 64 ks + 32 h .. + 31 per k-step, times scale * log2(e); ONE exponent per query (maximum over the lane pair)
    i32x8 q8[NKS];
    int eq;
    {
        const bf16_t* qrow = p.q + b * p.q_bs + (long)min(q0 + r, p.T - 1) * p.rs + 32 * h;
        float mx = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 v = *reinterpret_cast<const uint4*>(qrow + 64 * ks + 8 * i);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fmaxf(fabsf(bflo(w[e])), fabsf(bfhi(w[e]))));
            }
        const float sq = f8_pair_scale(mx * p.scale_log2, eq);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 v = *reinterpret_cast<const uint4*>(qrow + 64 * ks + 8 * i);   // (again: 256 values do not fit in registers next to everything else)
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    uint32_t u = f8_cvt2<false>(0u, bflo(w[2 * e]) * p.scale_log2, bfhi(w[2 * e]) * p.scale_log2, sq);
                    u = f8_cvt2<true>(u, bflo(w[2 * e + 1]) * p.scale_log2, bfhi(w[2 * e + 1]) * p.scale_log2, sq);
                    q8[ks][2 * i + e] = (int)u;
                }
            }
    }

    const uint32_t lds0 = lds_addr(smem);
    // K8 fragment of (key half kt, k-step ks): row 32 kt + r, bytes 64 ks + 32 h .. + 31; V8^T fragment of d-tile dt: row 32 dt + r, logical chunks 2h, 2h + 1
    const uint32_t k_lane[2] = {lds0 + r * KROW + 32 * h, lds0 + TILE_BYTES + r * KROW + 32 * h};
    uint32_t v_lane[2][2];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int s = 0; s < 2; ++s) v_lane[sl][s] = lds0 + sl * TILE_BYTES + K_BYTES + r * 64 + (((2 * h + s) ^ ((r >> 2) & 3)) << 4);

    struct Frag { bf16x8 a0, a1; };
    auto read_k = [&](Frag& f, int slot, auto ktc, auto ksc) {
        constexpr int off = decltype(ktc)::value * 32 * KROW + decltype(ksc)::value * 64;
        f.a0 = lds_read16<off>(k_lane[slot]);
        f.a1 = lds_read16<off + 16>(k_lane[slot]);
    };
    auto read_v = [&](Frag& f, int slot, auto dtc) {
        constexpr int off = decltype(dtc)::value * 32 * 64;
        f.a0 = lds_read16<off>(v_lane[slot][0]);
        f.a1 = lds_read16<off>(v_lane[slot][1]);
    };
    auto read_scale = [&](int& e, int slot) {
        const uint32_t a = lds0 + slot * TILE_BYTES;
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(e) : "v"(a), "n"(SCALE_OFF));
    };

    // ---- the softmax of one tile, cut into items that ride in the MFMA shadows (S = scores - m of the lane's 2 x 16 keys):
    // 0..31 exponentials in place; 32..47 block maximum and row sum (two values per item); 48 pair exchange -> exponent; 49..64 scaled converts
    float lsum = 0.f, smx = 0.f, ssc = 0.f;
    auto sm_item = [&](auto ic, f32x16 (&S)[2], i32x8& P, int& ebyte) {
        constexpr int I = decltype(ic)::value;
        if constexpr (I < 32) S[I >> 4][I & 15] = __builtin_amdgcn_exp2f(S[I >> 4][I & 15]);
        else if constexpr (I < 48) {
            constexpr int kt = (I - 32) >> 3, e = 2 * ((I - 32) & 7);
            smx = __builtin_fmaxf(__builtin_fmaxf(I == 32 ? 0.f : smx, S[kt][e]), S[kt][e + 1]);
            lsum += S[kt][e] + S[kt][e + 1];
        } else if constexpr (I == 48) ssc = f8_pair_scale(smx, ebyte);
        else {
            constexpr int c = I - 49, w = c >> 1, hi = c & 1, kt = w >> 2, e0 = 4 * (w & 3) + 2 * hi;
            if constexpr (hi) P[w] = (int)f8_cvt2<true>((uint32_t)P[w], S[kt][e0], S[kt][e0 + 1], ssc);
            else P[w] = (int)f8_cvt2<false>((uint32_t)P[w], S[kt][e0], S[kt][e0 + 1], ssc);
        }
    };
    constexpr int SM_ITEMS = 65;
    auto sm_range = [&](auto lo_c, auto hi_c, f32x16 (&S)[2], i32x8& P, int& ebyte) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        [&]<int... I>(std::integer_sequence<int, I...>) { (sm_item(std::integral_constant<int, LO + I>{}, S, P, ebyte), ...); }(std::make_integer_sequence<int, (HI > LO ? HI - LO : 0)>{});
    };

    f32x16 S[2], negm;
    i32x8 Pa = {0, 0, 0, 0, 0, 0, 0, 0}, Pb = Pa;
    int epa = 127, epb = 127, ek = 127, esc = 0;
    Frag fr[4];   // four fragment sets, reads two MFMAs ahead: a set is refilled two MFMAs after the one that read it
#pragma unroll
    for (int e = 0; e < 16; ++e) negm[e] = 0.f;
    asm volatile("s_nop 7" : "+v"(negm));   // pinned here: asm MFMAs are invisible to hipcc's hazard recogniser, which otherwise materialises these zeros directly in front of the MFMA that reads them as its C operand (tools/mfma_hazard_scan.py)

    // ---- the score product of one tile: 16 MFMAs, the two key halves alternating (consecutive MFMAs never share an accumulator),
    // fragments two MFMAs ahead in three register sets; `behind(i)` = what else issues behind MFMA i (LDS-DMA pieces)
    auto qk = [&](int slot, auto behind) {
        read_scale(ek, slot);
        read_k(fr[0], slot, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        read_k(fr[1], slot, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                constexpr int KT = I & 1, KS = I >> 1, N2 = I + 2;
                Frag& cur = fr[I & 3];
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (N2 < 16) read_k(fr[N2 & 3], slot, std::integral_constant<int, (N2 & 1)>{}, std::integral_constant<int, (N2 >> 1)>{});
                if constexpr (N2 < 16) wait_lds<2>(); else wait_lds<0>();
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (KS == 0) d8_mfma_c(S[KT], f8_join(cur.a0, cur.a1), q8[0], negm, ek, eq);
                else d8_mfma_acc(S[KT], f8_join(cur.a0, cur.a1), q8[KS], ek, eq);
                __builtin_amdgcn_sched_barrier(0);
                behind(std::integral_constant<int, I>{});
                keep(cur.a0); keep(cur.a1);
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 16>{});
        keep(ek);
    };
    // ---- O^T += V8^T(slot) P: 16 MFMAs, fragments two ahead; `behind(i)` = the softmax items of the next tile
    auto pv = [&](int slot, const i32x8& P, int ep, int ev, auto behind) {
        read_v(fr[0], slot, std::integral_constant<int, 0>{});
        read_v(fr[1], slot, std::integral_constant<int, 1>{});
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                constexpr int N2 = I + 2;
                Frag& cur = fr[I & 3];
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (N2 < 16) read_v(fr[N2 & 3], slot, std::integral_constant<int, N2>{});
                if constexpr (N2 < 16) wait_lds<2>(); else wait_lds<0>();
                __builtin_amdgcn_sched_barrier(0);
                d8_mfma_o<16 * I>(f8_join(cur.a0, cur.a1), P, ev, ep);
                __builtin_amdgcn_sched_barrier(0);
                behind(std::integral_constant<int, I>{});
                keep(cur.a0); keep(cur.a1);
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 16>{});
        keep(P); keep(ep); keep(ev);
    };
    auto nothing = [](auto) {};

    // ---- tile 0 in the open: scores with C = 0; the softmax reference is fixed here (row maximum + headroom)
    wait_vm<9>();   // everything but the nine K(1) pieces of this wave
    __syncthreads();
    qk(0, nothing);
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(S[0]), "+v"(S[1]));   // MFMA results -> VALU
    {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, S[kt][e]);
        const float m = xhalf_max(mx) + MARGIN;
#pragma unroll
        for (int e = 0; e < 16; ++e) negm[e] = -m;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) S[kt][e] -= m;
        sm_range(std::integral_constant<int, 0>{}, std::integral_constant<int, SM_ITEMS>{}, S, Pa, epa);
    }
    read_scale(esc, 0);
    wait_dma();       // K(1)
    wait_lds<0>();
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();

    // ---- main loop, two tiles per trip (ring slots and P sets static). Iteration t:
    //   QK^T(t+1) from slot (t+1) & 1, behind its MFMAs the DMA pieces of V^T(t+1) -> slot (t+1) & 1 and K(t+2) -> slot t & 1 (both free:
    //   V^T(t-1) was consumed by the previous iteration, K(t) by the iteration before that);
    //   PV(t) from slot t & 1, behind MFMAs 3..15 the softmax of tile t+1 -> the other P set; then all DMA has landed + one barrier.
    constexpr int SM_LO[17] = {0, 0, 0, 0, 5, 10, 15, 20, 25, 30, 35, 40, 45, 49, 54, 59, 65};
    auto step = [&](auto par, int t, const i32x8& Pcur, int epcur, i32x8& Pnext, int& epnext) {
        constexpr int SL = decltype(par)::value;        // slot of tile t; tile t + 1 sits in SL ^ 1
        const int ev = (esc >> 8) & 255;
        if (t + 1 >= NT) {   // the last tile: nothing left to score (and no phantom tile may reach the row sum)
            pv(SL, Pcur, epcur, ev, nothing);
            return;
        }
        qk(SL ^ 1, [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if constexpr (IR_KO_D8 & 4) return;
            if constexpr (I < 8) piece(t + 1, 33 + min(wu + 4 * I, 31), SL ^ 1);                  // V^T(t+1): pieces 33 + (wu + 4 i), i < 8
            else piece(t + 2, min(wu + 4 * (I - 8), 32), SL);                                  // K(t+2): pieces wu + 4 i, i < 9 (the ninth below)
            if constexpr (I == 15) piece(t + 2, min(wu + 32, 32), SL);
        });
        pv(SL, Pcur, epcur, ev, [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if constexpr (IR_KO_D8 & 2) { asm volatile("" : "+v"(S[0]), "+v"(S[1]), "+v"(Pnext), "+v"(epnext)); return; }
            sm_range(std::integral_constant<int, SM_LO[I]>{}, std::integral_constant<int, SM_LO[I + 1]>{}, S, Pnext, epnext);
        });
        read_scale(esc, SL ^ 1);   // exponents of tile t + 1 (its K part has been there since the previous iteration)
        if constexpr (!(IR_KO_D8 & 1)) wait_dma();
        wait_lds<0>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    for (int t = 0; t < NT; t += 2) {
        step(std::integral_constant<int, 0>{}, t, Pa, epa, Pb, epb);
        if (t + 1 >= NT) break;
        step(std::integral_constant<int, 1>{}, t + 1, Pb, epb, Pa, epa);
    }

    // ---- finalise: O^T[d][q] / l -> LDS [q][d] (the wave's own 32 rows) -> 16-byte row stores
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __syncthreads();   // every wave has finished with the ring (the staging rows overlay it)
    const float l = lsum + __shfl_xor(lsum, 32);
    const bool bad = !(l < 1e30f) || !(l > 0.f);
    const float inv = 1.0f / l;
    bf16_t* ow = reinterpret_cast<bf16_t*>(smem) + wid * 32 * OS;
    [&]<int... DT>(std::integer_sequence<int, DT...>) {
        ([&] {
            constexpr int A0 = 16 * DT;
            const float x[16] = {d8_acc_read<A0 + 0>(), d8_acc_read<A0 + 1>(), d8_acc_read<A0 + 2>(), d8_acc_read<A0 + 3>(),
                                 d8_acc_read<A0 + 4>(), d8_acc_read<A0 + 5>(), d8_acc_read<A0 + 6>(), d8_acc_read<A0 + 7>(),
                                 d8_acc_read<A0 + 8>(), d8_acc_read<A0 + 9>(), d8_acc_read<A0 + 10>(), d8_acc_read<A0 + 11>(),
                                 d8_acc_read<A0 + 12>(), d8_acc_read<A0 + 13>(), d8_acc_read<A0 + 14>(), d8_acc_read<A0 + 15>()};
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
                *reinterpret_cast<uint2*>(&ow[r * OS + DT * 32 + 8 * gg + 4 * h]) =
                    make_uint2(pack2bf(x[4 * gg] * inv, x[4 * gg + 1] * inv), pack2bf(x[4 * gg + 2] * inv, x[4 * gg + 3] * inv));
        }(), ...);
    }(std::make_integer_sequence<int, NDT>{});
    if (IR_KO_D8 == 0 && __any(bad) && lane == 0) atomicOr(p.ovf_flag, 1);   // (knock-out builds compute garbage: keep their timing free of the fallback)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bf16_t* op = p.o + b * p.o_bs;
    for (int c = lane; c < 32 * 64; c += 64) {   // the wave's own 32 rows of 64 chunks
        const int row = c >> 6, ch = c & 63;
        const int q = q0 + row;
        if (q < p.T) *reinterpret_cast<uint4*>(op + (long)q * p.o_rs + ch * 8) = *reinterpret_cast<const uint4*>(&ow[row * OS + ch * 8]);
    }
}

size_t ir_attn_d512_fp8_tile_bytes(int B, int T) { return (size_t)B * (T / 64) * f8d::TILE_BYTES; }
bool ir_attn_d512_fp8_takes(int T) {
    static const bool off = getenv("IR_NO_ATTN_D512_FP8") != nullptr;   // experiment knob
    return !off && T >= 256 && (T & 127) == 0;
}

// q, k, v: [B][T][512] bf16 rows (token stride rs, batch stride qk_bs); o: [B][T][512] (o_rs, o_bs); tiles: ir_attn_d512_fp8_tile_bytes(B, T)
// bytes of scratch; a set *ovf_flag afterwards means the result must be recomputed by the bf16 rescaling kernel (ir_launch_flash_attn_d512).
int ir_launch_flash_attn_d512_fp8(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* o, uint8_t* tiles, int B, int T, int rs, int o_rs,
                                  long qk_bs, long o_bs, float scale, int* ovf_flag, hipStream_t s) {
    using namespace f8d;
    if (B <= 0 || !ir_attn_d512_fp8_takes(T) || (rs & 7) || (o_rs & 7) || (qk_bs & 7) || (o_bs & 7) || !ovf_flag || !tiles) return -2;
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(o) |
         reinterpret_cast<uintptr_t>(tiles)) & 15)
        return -3;
    const int NT = T / 64;
    const size_t psm_bytes = 64 * (D + 8) * 2;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_d512_fp8_prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)psm_bytes) != hipSuccess) return -1;
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_d512_fp8_prep_kernel, dim3(NT, B), dim3(256), psm_bytes, s, k, v, tiles, qk_bs, rs, NT);
    AttnD512F8Params p;
    p.q = q; p.tiles = tiles; p.o = o; p.q_bs = qk_bs; p.o_bs = o_bs; p.tiles_bs = (long)NT * TILE_BYTES; p.T = T; p.rs = rs; p.o_rs = o_rs;
    p.scale_log2 = scale * 1.44269504088896340736f; p.ovf_flag = ovf_flag;
    hipLaunchKernelGGL(flash_attn_d512_fp8_kernel, dim3(T / 128, B), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
