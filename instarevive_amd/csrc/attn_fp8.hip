// DiT self-attention (16 heads x 72; reference diffusion/model/nets/PixArt_blocks.py:123-158) on fp8 (OCP e4m3) MFMA operands for gfx950:
// BASELINE.json configs[4] ("fp8 MFMA for DiT attention"). Both products run on v_mfma_scale_f32_32x32x64_f8f6f4 (k = 64 per
// instruction at twice the bf16 rate), with the MX block scales (one E8M0 exponent per row / column and 32-k block; a block is bytes
// 16b .. 16b+15 of the operand registers of the lane pair (l, l ^ 32), its exponent sits in lane (l & 31) + 32 b) doing the range work
// that fp8 cannot:
//
//   S^T = K Q^T     A = K8 (64 keys x d 0..63, e4m3, one exponent per (tile, head) from attn_fp8_prep_kernel), B = Q8 (the lane's own
//                   query row, quantised when it is loaded, one exponent per (query, scale block)); d 64..71 ride in ONE bf16
//                   v_mfma_f32_32x32x16_bf16 on the raw bf16 values (k = 16, upper half zero) - 96 matrix cycles per 32 keys
//                   instead of 160 for the bf16 kernel's five k-steps. -m (the fixed softmax reference) is the C operand.
//   O^T += V^T P^T  A = V8^T (d x 64 keys, e4m3, per (tile, head) exponent; row 72 = ones with exponent 0: the softmax denominator),
//                   B = P8: the lane's 32 probabilities of a tile (16 accumulator registers of each of the two 32-key score tiles)
//                   ARE one k = 64 operand - no data moves between lanes - and score tile kt is scale block kt: its exponent comes
//                   from the maximum over the lane pair's 2 x 16 registers (v_max3 chain + two v_permlane32_swap), so e4m3 only ever
//                   sees values relative to its own block: exp2(score - m) may be 2^-60 or 2^+60 against the fixed reference, the
//                   exponent byte carries that.
//                   The key order inside a tile is whatever the accumulator layout gives (lane half h, register g of score tile kt
//                   <-> key 32 kt + (g & 3) + 8 (g >> 2) + 4 h); V8^T is stored in that order by the prep kernel, so no operand is permuted
//                   at run time. 64 matrix cycles per 32 d-rows and 64 keys instead of 128.
//
// Per 64-key tile and 32-query group: 192 + 192 matrix cycles (bf16 kernel: 320 + 384). What is left is the softmax VALU work
// (32 v_exp + 16 v_max3 + 16 scaled converts per lane, group and tile), which now weighs as much as the matrix work.
// One wave per SIMD, two groups of 32 queries per wave (as flash_attn_pp2_kernel); K8 / V8^T of a tile are one 10 KB image that is
// its own LDS image (LDS-DMA, lane-linear), ring of PF + 1 slots, fixed softmax reference with overflow flag + rescaling fallback.
#include "common.h"
#include "kernels.h"
#include "attn_fp8_common.h"
#include <type_traits>
#include <utility>

namespace f8a {
constexpr int D = 72;
constexpr int KROW = 80;                       // K8 row: 64 e4m3 bytes (d 0..63) + 8 bf16 (d 64..71); 20 dwords: conflict-free for ds_read_b128
constexpr int K_BYTES = 64 * KROW;             // 5120
constexpr int VROWS = 80;                      // V8^T rows: d 0..71, ones row 72, zeros, row 79 = the tile's exponent bytes
constexpr int V_BYTES = VROWS * 64;            // 5120
constexpr int TILE_BYTES = K_BYTES + V_BYTES;  // 10 DMA pieces of 1 KB
constexpr int SCALE_OFF = K_BYTES + 79 * 64;   // dword {K exponent byte, V exponent byte, 0, 0}
constexpr int ONE_OFF = K_BYTES + 78 * 64;     // 16 bytes: bf16 {1, 0, 0, 0, 0, 0, 0, 0}
constexpr int PF = 5, NSLOT = PF + 1;          // tile t + PF is issued in iteration t, two iterations before its first use
constexpr int LDS_MAIN = NSLOT * TILE_BYTES;   // 61 440 B
constexpr int OS = 96 + 8;                     // O staging row stride (elements)
constexpr int LDS_O = 8 * 32 * OS * 2;         // 53 248 B
constexpr int LDS_BYTES = LDS_MAIN > LDS_O ? LDS_MAIN : LDS_O;
constexpr float MARGIN = 24.0f;                // headroom below the first tile's maximum
}  // namespace f8a
// ---------------------------------------------------------------------------------------------------------------------------------
// K, V [B][T][..] bf16 (token stride rs, head stride hs) -> tile images [B][Hh][T/64][10240 B] (see the header of this file)
__global__ __launch_bounds__(256) void attn_fp8_prep_kernel(const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, uint8_t* __restrict__ tiles,
                                                            long kv_bs, int rs, int hs, int NT) {
    using namespace f8a;
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64][72 + 8], Vs[64][72 + 8];
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tile = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const bf16_t* ksrc = k + (long)b * kv_bs + (long)head * hs + (long)tile * 64 * rs;
    const bf16_t* vsrc = v + (long)b * kv_bs + (long)head * hs + (long)tile * 64 * rs;
    float mk = 0.f, mv = 0.f;
    for (int c = tid; c < 64 * 9; c += 256) {
        const int row = c / 9, ch = c - row * 9;
        const uint4 a = *reinterpret_cast<const uint4*>(ksrc + (long)row * rs + ch * 8);
        const uint4 w = *reinterpret_cast<const uint4*>(vsrc + (long)row * rs + ch * 8);
        *reinterpret_cast<uint4*>(&Ks[row][ch * 8]) = a;
        *reinterpret_cast<uint4*>(&Vs[row][ch * 8]) = w;
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (ch < 8) mk = fmaxf(mk, fmaxf(fabsf(bflo(aw[i])), fabsf(bfhi(aw[i]))));   // d 64..71 of K stay bf16: not part of the fp8 block
            mv = fmaxf(mv, fmaxf(fabsf(bflo(ww[i])), fabsf(bfhi(ww[i]))));
        }
    }
    mk = wave_max(mk); mv = wave_max(mv);
    if (lane == 0) { red[0][wid] = mk; red[1][wid] = mv; }
    __syncthreads();
    mk = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    mv = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    const int bk = f8_block_byte(mk), bv = f8_block_byte(mv);
    const float sk = __builtin_bit_cast(float, (uint32_t)bk << 23), sv = __builtin_bit_cast(float, (uint32_t)bv << 23);
    uint8_t* dst = tiles + (((long)(b * gridDim.y + head) * NT) + tile) * TILE_BYTES;
    // K8: 64 rows x 5 chunks of 16 B
    for (int c = tid; c < 64 * 5; c += 256) {
        const int row = c / 5, ch = c - row * 5;
        uint4 o;
        if (ch == 4) {
            o = *reinterpret_cast<const uint4*>(&Ks[row][64]);
        } else {
            uint32_t w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16_t* s = &Ks[row][ch * 16 + 4 * i];
                w[i] = f8_cvt2<false>(0u, bf2f(s[0]), bf2f(s[1]), sk);
                w[i] = f8_cvt2<true>(w[i], bf2f(s[2]), bf2f(s[3]), sk);
            }
            o = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *reinterpret_cast<uint4*>(dst + row * KROW + ch * 16) = o;
    }
    // V8^T: 80 rows x 4 chunks of 16 B; logical byte L = 32 h + j of a row <-> key 32 (j >> 4) + (j & 3) + 8 ((j & 15) >> 2) + 4 h;
    // logical chunk c (16 bytes) sits at physical chunk c ^ ((d >> 2) & 3)
    for (int c = tid; c < VROWS * 4; c += 256) {
        const int d = c >> 2, pc = c & 3, lc = pc ^ ((d >> 2) & 3);
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (d < D) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int L = 16 * lc + 4 * i + e, h = L >> 5, j = L & 31;
                    const int key = 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * h;
                    x[e] = bf2f(Vs[key][d]);
                }
                w[i] = f8_cvt2<false>(0u, x[0], x[1], sv);
                w[i] = f8_cvt2<true>(w[i], x[2], x[3], sv);
            }
        } else if (d == D) {
            w[0] = w[1] = w[2] = w[3] = 0x38383838u;   // 1.0 in e4m3; the kernel multiplies this row with exponent byte 127
        } else if (d == VROWS - 1 && pc == 0) {
            w[0] = (uint32_t)bk | ((uint32_t)bv << 8);
        } else if (d == VROWS - 2 && pc == 0) {
            w[0] = 0x3F80u;   // bf16 {1, 0, 0, ...}: the K-side partner of -m in the remainder MFMA (see flash_attn_fp8_kernel)
        }
        *reinterpret_cast<uint4*>(dst + K_BYTES + d * 64 + pc * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

#ifdef IR_STAMPS_F8   // diagnostic build (tools/dbg/attn8_stamps.py): per-wave cycle sums of [0] phase A, [1] mid (waits + barrier + DMA issue), [2] phase B.
// Cycles, not time: this kernel is power-limited (1.70 GHz with every MFMA in place, 2.14 GHz with the score MFMAs knocked out), so
// knock-out TIMES mostly show the clock moving; only cycle counts say what an instruction group costs (about 95 cycles per tile for the
// eight score MFMAs, the same for the six PV MFMAs: they do hide behind the softmax).
__device__ unsigned long long g_f8_stamps[16];
#define F8_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define F8_ACC(k, a, b) st_acc[k] += (b) - (a)
#else
#define F8_T(v) do { } while (0)
#define F8_ACC(k, a, b) do { } while (0)
#endif
struct AttnF8Params {
    const bf16_t* q;
    const uint8_t* tiles;
    bf16_t* o;
    long q_bs, o_bs;
    int q_rs, o_rs, q_hs, o_hs;
    int Hh, Tq, Tk;
    float scale_log2;
    int* ovf_flag;
};

// The MFMAs are inline asm: S^T tiles land in arch VGPRs the VALU reads directly, O^T tiles in literal AGPRs. Hazards hipcc cannot see
// are covered by construction: an MFMA result is read by the VALU a phase (hundreds of cycles) later or behind explicit s_nops; VALU-written
// operands (P8, exponent bytes) are a phase old; LDS fragments are behind counted lgkmcnt waits; the bf16 remainder MFMA, which reads the
// e4m3 MFMA's result as its C operand, issues at least 18 wait states behind it (another MFMA plus softmax items sit in between).
// diagnostic builds: wait states behind the MFMAs of one kind (-DIR_F8_NOP=1: e4m3 score MFMAs, 2: bf16 remainder MFMAs, 4: O^T MFMAs; sums combine)
#ifndef IR_F8_NOP
#define IR_F8_NOP 0
#endif
// knock-out builds (diagnostic, results wrong by design): -DIR_KO_F8=1 no MFMAs, 2 no softmax items, 4 no per-tile barrier + DMA wait, 8 no fragment reads
#ifndef IR_KO_F8
#define IR_KO_F8 0
#endif
IR_DEVINL void f8_mfma_s(f32x16& s, i32x8 a, i32x8 b, int sa, int sb) {   // s = A8 B8 (block scales sa / sb, byte 0), fresh destination
    if constexpr (IR_KO_F8 & (1 | 32)) { asm volatile("" : "=v"(s) : "v"(a), "v"(b), "v"(sa), "v"(sb)); return; }
    if constexpr (IR_F8_NOP & 1) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel_hi:[0,0,0]\n\ts_nop 15" : "=&v"(s) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    else asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel_hi:[0,0,0]" : "=&v"(s) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}
IR_DEVINL void bf_mfma_acc(f32x16& s, bf16x8 a, bf16x8 b) {
    if constexpr (IR_KO_F8 & (1 | 16)) { asm volatile("" : "+v"(s) : "v"(a), "v"(b)); return; }
    if constexpr (IR_F8_NOP & 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15" : "+v"(s) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(a), "v"(b));
}
template <int LO>
IR_DEVINL void f8_mfma_o(i32x8 a, i32x8 b, int sa, int sb) {   // a[LO : LO + 15] += A8 B8
    if constexpr (IR_KO_F8 & (1 | 64)) { asm volatile("" ::"v"(a), "v"(b), "v"(sa), "v"(sb)); return; }
    if constexpr (IR_F8_NOP & 4) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 a[%c4:%c5], %0, %1, a[%c4:%c5], %2, %3 op_sel_hi:[0,0,0]\n\ts_nop 15" ::"v"(a), "v"(b), "v"(sa), "v"(sb), "n"(LO), "n"(LO + 15));
    else asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 a[%c4:%c5], %0, %1, a[%c4:%c5], %2, %3 op_sel_hi:[0,0,0]" ::"v"(a), "v"(b), "v"(sa), "v"(sb), "n"(LO), "n"(LO + 15));
}
// (keep(): attn_fp8_common.h - pins an MFMA's operand registers past the softmax items that follow it: with the fragment registers handed
// back to hipcc right behind the MFMA, an exponential landed in them and the product came out wrong - intermittently.)
template <int I>
IR_DEVINL float f8_acc_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "n"(I));
    return x;
}
#define IR_AGPR96_CLOBBERS "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95"

__global__ __launch_bounds__(256, 1) void flash_attn_fp8_kernel(AttnF8Params p) {
    using namespace f8a;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 256 + wid * 64, head = blockIdx.y, b = blockIdx.z;
    const int NT = p.Tk >> 6;
    const bf16_t* qp = p.q + (long)b * p.q_bs + (long)head * p.q_hs;
    const uint8_t* tp = p.tiles + ((long)(b * p.Hh + head) * NT) * TILE_BYTES + lane * 16;

    // LDS-DMA: the 10 pieces of a tile image, pieces wu, wu + 4, wu + 8 (clamped to 9: a repeated piece writes the same bytes) per wave
    auto issue_tile = [&](int tile, auto slot_c) {   // tile (clamped: past the end the last tile is fetched again, which keeps the vmcnt bookkeeping constant) -> ring slot
        constexpr int slot = decltype(slot_c)::value;
        const uint8_t* src = tp + (long)min(tile, NT - 1) * TILE_BYTES;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = min(wu + 4 * k, 9);
            f8_glds16(src + idx * 1024, (f8_lds_t)(smem + slot * TILE_BYTES + idx * 1024));
        }
    };
    [&]<int... T>(std::integer_sequence<int, T...>) { (issue_tile(T, std::integral_constant<int, T>{}), ...); }(std::make_integer_sequence<int, PF>{});

    // ---- Q: the lane's own query rows. fp8 part: d 32h .. 32h + 31 scaled by scale * log2(e), one exponent per (query, 32-d block);
    // bf16 part: d 64..71 (lanes of the upper half hold the zero padding k = 8..15 of that MFMA)
    i32x8 q8[2];
    bf16x8 qr[2];
    int eq[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const bf16_t* qrow = qp + (long)min(q0 + g * 32 + r, p.Tq - 1) * p.q_rs;
        uint4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const uint4*>(qrow + 32 * h + 8 * i);
        const uint4 vr = *reinterpret_cast<const uint4*>(qrow + 64);
        float x[32];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[8 * i + 2 * e] = bflo(w[e]) * p.scale_log2; x[8 * i + 2 * e + 1] = bfhi(w[e]) * p.scale_log2; }
        }
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) mx = fmaxf(mx, fabsf(x[i]));
        const float sq = f8_pair_scale(mx, eq[g]);   // one exponent for the query's 64 e4m3 values (both scale blocks)
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            uint32_t u = f8_cvt2<false>(0u, x[4 * w], x[4 * w + 1], sq);
            u = f8_cvt2<true>(u, x[4 * w + 2], x[4 * w + 3], sq);
            q8[g][w] = (int)u;
        }
        const float sc = h ? 0.f : p.scale_log2;
        const uint4 qq = make_uint4(pack2bf(bflo(vr.x) * sc, bfhi(vr.x) * sc), pack2bf(bflo(vr.y) * sc, bfhi(vr.y) * sc),
                                    pack2bf(bflo(vr.z) * sc, bfhi(vr.z) * sc), pack2bf(bflo(vr.w) * sc, bfhi(vr.w) * sc));
        qr[g] = __builtin_bit_cast(bf16x8, qq);
    }

    const uint32_t lds0 = lds_addr(smem);
    const uint32_t k_lane = lds0 + r * KROW + 32 * h;        // + slot * TILE_BYTES + kt * 32 * KROW (+16 second half); remainder chunk at r * KROW + 64
    // remainder MFMA (k = 16): lower lanes read d 64..71 of their key row; upper lanes (k = 8..15) read the constant chunk {1, 0, ..}: with
    // -m in element 0 of the query side's upper lanes that MFMA also subtracts the softmax reference (per query constant, so its bf16
    // rounding cancels in the normalisation), and the e4m3 MFMA before it can take the inline constant 0 as its C operand
    const uint32_t kr_lane[2] = {h ? lds0 + ONE_OFF : lds0 + r * KROW + 64, h ? lds0 + ONE_OFF : lds0 + (32 + r) * KROW + 64};
    // V8^T: row 32 dt + r (dt = 2: rows 64 + (r & 15): rows 80..95 do not exist; their results are never read), logical chunks 2h, 2h + 1
    uint32_t v_lane[3][2];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
        const int d = dt < 2 ? 32 * dt + r : 64 + (r & 15);
#pragma unroll
        for (int s = 0; s < 2; ++s) v_lane[dt][s] = lds0 + K_BYTES + d * 64 + (((2 * h + s) ^ ((d >> 2) & 3)) << 4);
    }
    const bool ones_row = (r & 15) == 8;   // d-tile 2, row 72: exponent byte 127 instead of the tile's V exponent

    // O^T accumulators: 2 groups x 3 d-tiles x 16 = a[0:95], addressed literally by the inline-asm MFMAs (hipcc moves accumulators it can
    // see between the two register files around every MFMA: 1000 v_accvgpr moves per tile in the first version of this kernel). Everything
    // the compiler sees lives in arch VGPRs (< 256, so it has no reason to touch the AGPR half; the resource-usage remark must say AGPRs: 96).
    asm volatile(".set ir_f8_i, 0\n\t.rept 96\n\tv_accvgpr_write_b32 a[ir_f8_i], 0\n\t.set ir_f8_i, ir_f8_i + 1\n\t.endr" ::: IR_AGPR96_CLOBBERS);

    // fragment reads (inline asm: invisible to hipcc's waitcnt insertion, so every use is preceded by a counted wait_lds + sched_barrier).
    // The ring slot of every read is a compile-time constant (the loop body is six tiles long), so an address is a per-lane register
    // plus an immediate: no address arithmetic in the stream.
    struct KF { bf16x8 a0, a1, ar; };
    struct VF { bf16x8 a0, a1; };
    auto read_k = [&](KF& f, auto slot_c, auto ktc) {
        constexpr int base = decltype(slot_c)::value * TILE_BYTES, kt = decltype(ktc)::value;
        if constexpr (IR_KO_F8 & 8) { asm volatile("" : "+v"(f.a0), "+v"(f.a1), "+v"(f.ar)); return; }
        f.a0 = lds_read16<base + kt * 32 * KROW>(k_lane);
        f.a1 = lds_read16<base + kt * 32 * KROW + 16>(k_lane);
        f.ar = lds_read16<base>(kr_lane[kt]);
    };
    auto read_v = [&](VF& f, auto slot_c, auto dtc) {
        constexpr int base = decltype(slot_c)::value * TILE_BYTES, dt = decltype(dtc)::value;
        if constexpr (IR_KO_F8 & 8) { asm volatile("" : "+v"(f.a0), "+v"(f.a1)); return; }
        f.a0 = lds_read16<base>(v_lane[dt][0]);
        f.a1 = lds_read16<base>(v_lane[dt][1]);
    };
    auto read_scale = [&](int& e, auto slot_c) {
        const uint32_t a = lds0;   // (named outside the asm: clang does not capture a variable that only an asm operand of a generic lambda names)
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(e) : "v"(a), "n"(decltype(slot_c)::value * TILE_BYTES + SCALE_OFF));
    };

    // the softmax work of one query group and tile, cut into items that ride in the MFMA shadows: S (scores - m of two 32-key tiles)
    // -> exponentials in place -> maximum -> exponent -> P8. Measured issue costs (tools/valu_rate_probe.hip): v_exp_f32 and the scaled
    // convert 8.8 cycles each, v_max3_f32 5, the pair exchange about 35.
    struct Sm { float mx, sp; };
    auto sm_item = [&](auto ic, f32x16 (&S)[2], Sm& st, i32x8& P, int& ebyte) {
        constexpr int I = decltype(ic)::value;
        if constexpr (I < 16) S[0][I] = __builtin_amdgcn_exp2f(S[0][I]);
        else if constexpr (I < 24) {
            constexpr int e = 2 * (I - 16);
            st.mx = __builtin_fmaxf(__builtin_fmaxf(I == 16 ? 0.f : st.mx, S[0][e]), S[0][e + 1]);
        } else if constexpr (I < 40) S[1][I - 24] = __builtin_amdgcn_exp2f(S[1][I - 24]);
        else if constexpr (I < 48) {
            constexpr int e = 2 * (I - 40);
            st.mx = __builtin_fmaxf(__builtin_fmaxf(st.mx, S[1][e]), S[1][e + 1]);
        } else if constexpr (I == 48) {
            st.sp = f8_pair_scale(st.mx, ebyte);
        } else {
            constexpr int c = I - 49, w = c >> 1, hi = c & 1, kt = w >> 2, e0 = 4 * (w & 3) + 2 * hi;
            if constexpr (hi) P[w] = (int)f8_cvt2<true>((uint32_t)P[w], S[kt][e0], S[kt][e0 + 1], st.sp);
            else P[w] = (int)f8_cvt2<false>((uint32_t)P[w], S[kt][e0], S[kt][e0 + 1], st.sp);   // (the other half is rewritten by the next item: no zeroing)
        }
    };
    constexpr int SM_ITEMS = 65;
    auto sm_range = [&](auto lo_c, auto hi_c, f32x16 (&S)[2], Sm& st, i32x8& P, int& ebyte) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (sm_item(std::integral_constant<int, LO + I>{}, S, st, P, ebyte), ...);
        }(std::make_integer_sequence<int, (HI > LO ? HI - LO : 0)>{});
    };

    f32x16 S0[2], S1a[2], S1b[2];     // scores of group 0 (consumed in the iteration that produced them) / of group 1 (consumed one iteration later: two sets)
    i32x8 P0a = {0, 0, 0, 0, 0, 0, 0, 0}, P0b = P0a, P1 = P0a;   // P8 of group 0 (two sets: PV(t) reads one while the softmax of tile t + 1 writes the other) / of group 1
    int e0a = 127, e0b = 127, e1 = 127;
    Sm st0, st1;
    KF kf0, kf1;
    VF vfa, vfb;
    int ek, esc;
#ifdef IR_STAMPS_F8
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>; using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>;

    // ---- tile 0 in the open: scores with C = 0; the softmax reference is fixed here (row maximum + headroom)
    wait_vm<3 * (PF - 1)>();
    __syncthreads();
    read_scale(ek, I0{});
    read_k(kf0, I0{}, I0{});
    read_k(kf1, I0{}, I1{});
    wait_lds<0>();
    __builtin_amdgcn_sched_barrier(0);
    f8_mfma_s(S0[0], f8_join(kf0.a0, kf0.a1), q8[0], ek, eq[0]);
    f8_mfma_s(S1a[0], f8_join(kf0.a0, kf0.a1), q8[1], ek, eq[1]);
    f8_mfma_s(S0[1], f8_join(kf1.a0, kf1.a1), q8[0], ek, eq[0]);
    f8_mfma_s(S1a[1], f8_join(kf1.a0, kf1.a1), q8[1], ek, eq[1]);
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // the remainder MFMAs read these results as C: 18 wait states behind the last producer
    bf_mfma_acc(S0[0], kf0.ar, qr[0]);
    bf_mfma_acc(S1a[0], kf0.ar, qr[1]);
    bf_mfma_acc(S0[1], kf1.ar, qr[0]);
    bf_mfma_acc(S1a[1], kf1.ar, qr[1]);
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(S0[0]), "+v"(S0[1]), "+v"(S1a[0]), "+v"(S1a[1]));   // MFMA results -> VALU: the wait states hipcc cannot see
    {
        float mx0 = -INFINITY, mx1 = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { mx0 = fmaxf(mx0, S0[kt][e]); mx1 = fmaxf(mx1, S1a[kt][e]); }
        // rounded to bf16 once: later tiles get it through the bf16 operand of the remainder MFMA, tile 0 through the subtraction below
        const float m0 = bf2f(f2bf(xhalf_max(mx0) + MARGIN)), m1 = bf2f(f2bf(xhalf_max(mx1) + MARGIN));
        if (h) {   // -m into element 0 (k = 8) of the remainder MFMA's query operand on the upper lanes
            qr[0] = __builtin_bit_cast(bf16x8, make_uint4((uint32_t)f2bf(-m0), 0u, 0u, 0u));
            qr[1] = __builtin_bit_cast(bf16x8, make_uint4((uint32_t)f2bf(-m1), 0u, 0u, 0u));
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { S0[kt][e] -= m0; S1a[kt][e] -= m1; }   // from here on the scores arrive as score - m
        sm_range(I0{}, std::integral_constant<int, SM_ITEMS>{}, S0, st0, P0a, e0a);
    }
    wait_vm<3 * (PF - 2)>();   // tile 1
    __syncthreads();
    read_scale(ek, I1{});
    read_k(kf0, I1{}, I0{});

    // ---- main loop. Iteration t (ring slot K = t % 6 is a template argument):
    //   phase A   8 MFMAs: S(t+1) per 32 keys and group = one e4m3 k = 64 MFMA + the bf16 remainder (which also subtracts m);
    //             in their shadows the softmax of (tile t, group 1), whose scores the previous iteration left in S1old -> P1
    //   mid       tile t + 2 has landed for every wave (counted vmcnt + the one barrier of the iteration); tile t + PF is issued
    //   phase B   6 MFMAs: O^T += V8^T(t) P8(t), d-tile by d-tile, both groups; in their shadows the softmax of (tile t + 1, group 0) -> P0next
    // Fragments are read one MFMA pair ahead. The last iteration runs the same stream: its "tile t + 1" is whatever the next slot holds
    // (finite bytes of an older tile), and nothing consumes those scores.
    // Item ranges per MFMA shadow, cut by issue cost: phase A slot weights 2 2 1 1 2 2 1 1 (e4m3 / bf16 MFMA), phase B equal.
    constexpr int A_LO[9] = {0, 9, 19, 25, 30, 38, 49, 56, 65}, B_LO[7] = {0, 9, 19, 30, 39, 50, 65};
    auto tile_step = [&](auto kc, int t, f32x16 (&S1old)[2], f32x16 (&S1new)[2], i32x8& P0cur, int& e0cur, i32x8& P0next, int& e0next) {
        constexpr int K = decltype(kc)::value;
        using SV = std::integral_constant<int, K>;                 // slot of tile t
        using SK = std::integral_constant<int, (K + 1) % NSLOT>;   // of tile t + 1
        using SK2 = std::integral_constant<int, (K + 2) % NSLOT>;  // of tile t + 2
        using SF = std::integral_constant<int, (K + PF) % NSLOT>;  // free: tile t - 1 was its last user
        auto sm_a = [&](auto jc) {
            constexpr int J = decltype(jc)::value;
            if constexpr (IR_KO_F8 & 2) { asm volatile("" : "+v"(S1old[0]), "+v"(S1old[1]), "+v"(P1), "+v"(e1)); return; }
            sm_range(std::integral_constant<int, A_LO[J]>{}, std::integral_constant<int, A_LO[J + 1]>{}, S1old, st1, P1, e1);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto sm_b = [&](auto jc) {
            constexpr int J = decltype(jc)::value;
            if constexpr (IR_KO_F8 & 2) { asm volatile("" : "+v"(S0[0]), "+v"(S0[1]), "+v"(P0next), "+v"(e0next)); return; }
            sm_range(std::integral_constant<int, B_LO[J]>{}, std::integral_constant<int, B_LO[J + 1]>{}, S0, st0, P0next, e0next);
            __builtin_amdgcn_sched_barrier(0);
        };
        F8_T(ta);
        // ---- phase A (in flight from the previous phase: ek, kf0 of tile t + 1). Order: the two e4m3 MFMAs of a 32-key half (groups 0, 1), then
        // their two bf16 remainders. A remainder MFMA reads the e4m3 MFMA's result as its C operand, and between MFMAs of DIFFERENT opcodes
        // nothing forwards or interlocks: the consumer must issue >= 16 passes + 2 wait states behind the producer (hipcc's hazard table).
        // Here another 16- / 8-pass MFMA and >= 15 softmax items always sit in between.
        __builtin_amdgcn_sched_barrier(0);
        read_k(kf1, SK{}, I1{});
        read_scale(esc, SV{});
        wait_lds<4>();
        __builtin_amdgcn_sched_barrier(0);
        f8_mfma_s(S0[0], f8_join(kf0.a0, kf0.a1), q8[0], ek, eq[0]); __builtin_amdgcn_sched_barrier(0);
        sm_a(I0{});
        f8_mfma_s(S1new[0], f8_join(kf0.a0, kf0.a1), q8[1], ek, eq[1]); __builtin_amdgcn_sched_barrier(0);
        sm_a(I1{});
        keep(kf0.a0); keep(kf0.a1);
        bf_mfma_acc(S0[0], kf0.ar, qr[0]); __builtin_amdgcn_sched_barrier(0);
        sm_a(I2{});
        bf_mfma_acc(S1new[0], kf0.ar, qr[1]); __builtin_amdgcn_sched_barrier(0);
        sm_a(I3{});
        keep(kf0.ar);
        wait_lds<0>();    // kf1, esc
        __builtin_amdgcn_sched_barrier(0);
        read_v(vfa, SV{}, I0{});
        __builtin_amdgcn_sched_barrier(0);
        f8_mfma_s(S0[1], f8_join(kf1.a0, kf1.a1), q8[0], ek, eq[0]); __builtin_amdgcn_sched_barrier(0);
        sm_a(I4{});
        f8_mfma_s(S1new[1], f8_join(kf1.a0, kf1.a1), q8[1], ek, eq[1]); __builtin_amdgcn_sched_barrier(0);
        sm_a(I5{});
        keep(kf1.a0); keep(kf1.a1);
        bf_mfma_acc(S0[1], kf1.ar, qr[0]); __builtin_amdgcn_sched_barrier(0);
        sm_a(I6{});
        bf_mfma_acc(S1new[1], kf1.ar, qr[1]); __builtin_amdgcn_sched_barrier(0);
        sm_a(I7{});
        keep(kf1.ar); keep(ek);
        const int ev = esc >> 8, ev2 = ones_row ? 127 : ev;
        F8_T(tb);
        // ---- mid
        wait_lds<0>();            // V(t) d-tile 0
        if constexpr (!(IR_KO_F8 & 4)) {
        wait_vm<3 * (PF - 3)>();  // tile t + 2
        __builtin_amdgcn_s_barrier();   // bare: __syncthreads() makes hipcc drain vmcnt(0) in front of it, i.e. wait for the tiles still in flight
        }
        issue_tile(t + PF, SF{});
        __builtin_amdgcn_sched_barrier(0);
        F8_T(tc);
        // ---- phase B
        read_v(vfb, SV{}, I1{});
        __builtin_amdgcn_sched_barrier(0);
        f8_mfma_o<0>(f8_join(vfa.a0, vfa.a1), P0cur, ev, e0cur);
        __builtin_amdgcn_sched_barrier(0);
        sm_b(I0{});
        f8_mfma_o<48>(f8_join(vfa.a0, vfa.a1), P1, ev, e1);
        __builtin_amdgcn_sched_barrier(0);
        sm_b(I1{});
        keep(vfa.a0); keep(vfa.a1);
        wait_lds<0>();
        __builtin_amdgcn_sched_barrier(0);
        read_v(vfa, SV{}, I2{});
        __builtin_amdgcn_sched_barrier(0);
        f8_mfma_o<16>(f8_join(vfb.a0, vfb.a1), P0cur, ev, e0cur);
        __builtin_amdgcn_sched_barrier(0);
        sm_b(I2{});
        f8_mfma_o<64>(f8_join(vfb.a0, vfb.a1), P1, ev, e1);
        __builtin_amdgcn_sched_barrier(0);
        sm_b(I3{});
        keep(vfb.a0); keep(vfb.a1);
        wait_lds<0>();
        __builtin_amdgcn_sched_barrier(0);
        read_scale(ek, SK2{});   // tile t + 2 (landed: the mid barrier): next iteration's first fragments
        read_k(kf0, SK2{}, I0{});
        __builtin_amdgcn_sched_barrier(0);
        f8_mfma_o<32>(f8_join(vfa.a0, vfa.a1), P0cur, ev2, e0cur);
        __builtin_amdgcn_sched_barrier(0);
        sm_b(I4{});
        f8_mfma_o<80>(f8_join(vfa.a0, vfa.a1), P1, ev2, e1);
        __builtin_amdgcn_sched_barrier(0);
        sm_b(I5{});
        keep(vfa.a0); keep(vfa.a1); keep(P0cur); keep(P1); keep(ev); keep(ev2); keep(e0cur); keep(e1);
        F8_T(td);
        F8_ACC(0, ta, tb); F8_ACC(1, tb, tc); F8_ACC(2, tc, td);
    };
    for (int t = 0; t < NT; t += NSLOT) {   // six tiles per trip: ring slots and buffer parities are static; a short last trip leaves early
        tile_step(I0{}, t, S1a, S1b, P0a, e0a, P0b, e0b);
        if (t + 1 >= NT) break;
        tile_step(I1{}, t + 1, S1b, S1a, P0b, e0b, P0a, e0a);
        if (t + 2 >= NT) break;
        tile_step(I2{}, t + 2, S1a, S1b, P0a, e0a, P0b, e0b);
        if (t + 3 >= NT) break;
        tile_step(I3{}, t + 3, S1b, S1a, P0b, e0b, P0a, e0a);
        if (t + 4 >= NT) break;
        tile_step(I4{}, t + 4, S1a, S1b, P0a, e0a, P0b, e0b);
        if (t + 5 >= NT) break;
        tile_step(I5{}, t + 5, S1b, S1a, P0b, e0b, P0a, e0a);
    }

#ifdef IR_STAMPS_F8
    if (lane == 0) for (int i = 0; i < 3; ++i) atomicAdd(&g_f8_stamps[i], st_acc[i]);
#endif
    // ---- finalise: O^T[d][q] / l -> LDS [q][d] -> 16-byte row stores; l = O^T row 72 (the ones row): d-tile 2, register 4, lane half 0
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    wait_dma();
    __syncthreads();   // every wave has finished with the tile ring (the staging rows overlay it)
    bf16_t* ow = reinterpret_cast<bf16_t*>(smem) + wid * 64 * OS;
    const float l0 = __shfl(f8_acc_read<32 + 4>(), r), l1 = __shfl(f8_acc_read<80 + 4>(), r);
    const bool bad = !(l0 < 1e30f) || !(l1 < 1e30f) || !(l0 > 0.f) || !(l1 > 0.f);   // also catches inf / NaN: the fixed reference was outgrown
    const float inv0 = 1.0f / l0, inv1 = 1.0f / l1;
    [&]<int... I>(std::integer_sequence<int, I...>) {
        ([&] {
            constexpr int G = I / 3, DT = I % 3, A0 = 16 * I;
            const float inv = G ? inv1 : inv0;
            const float x[16] = {f8_acc_read<A0 + 0>(), f8_acc_read<A0 + 1>(), f8_acc_read<A0 + 2>(), f8_acc_read<A0 + 3>(),
                                 f8_acc_read<A0 + 4>(), f8_acc_read<A0 + 5>(), f8_acc_read<A0 + 6>(), f8_acc_read<A0 + 7>(),
                                 f8_acc_read<A0 + 8>(), f8_acc_read<A0 + 9>(), f8_acc_read<A0 + 10>(), f8_acc_read<A0 + 11>(),
                                 f8_acc_read<A0 + 12>(), f8_acc_read<A0 + 13>(), f8_acc_read<A0 + 14>(), f8_acc_read<A0 + 15>()};
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const uint2 w = make_uint2(pack2bf(x[4 * gg] * inv, x[4 * gg + 1] * inv), pack2bf(x[4 * gg + 2] * inv, x[4 * gg + 3] * inv));
                *reinterpret_cast<uint2*>(&ow[(G * 32 + r) * OS + DT * 32 + 8 * gg + 4 * h]) = w;
            }
        }(), ...);
    }(std::make_integer_sequence<int, 6>{});
    if (IR_KO_F8 == 0 && __any(bad) && lane == 0) atomicOr(p.ovf_flag, 1);   // (knock-out builds compute garbage: keep their timing free of the fallback)
    __syncthreads();
    bf16_t* op = p.o + (long)b * p.o_bs + (long)head * p.o_hs;
    for (int c = lane; c < 64 * 9; c += 64) {
        const int row = c / 9, ch = c - row * 9;
        const int q = q0 + row;
        if (q < p.Tq) *reinterpret_cast<uint4*>(op + (long)q * p.o_rs + ch * 8) = *reinterpret_cast<const uint4*>(&ow[row * OS + ch * 8]);
    }
}

#ifdef IR_STAMPS_F8
extern "C" void ir_f8_stamps(unsigned long long* out, int reset) {
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_f8_stamps), z, sizeof z); return; }
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f8_stamps), 16 * sizeof(unsigned long long));
}
#endif
size_t ir_attn_fp8_tile_bytes(int B, int Hh, int Tk) { return (size_t)B * Hh * (Tk / 64) * f8a::TILE_BYTES; }

// p as for ir_launch_flash_attn's DiT self-attention form (D = 72, Tk % 64 == 0, no key bias, ovf_flag set, p.vt = the bf16 V^T buffer the
// rescaling fallback uses); v = the V rows (same strides as p.k); tiles = ir_attn_fp8_tile_bytes(...) bytes of scratch.
int ir_launch_flash_attn_fp8(const AttnParams& p, const bf16_t* v, uint8_t* tiles, hipStream_t s) {
    if (p.D != 72 || p.Tq <= 0 || p.Tk < 64 || (p.Tk & 63) || !p.ovf_flag || p.key_bias || !v || !tiles) return -2;
    if ((p.q_rs & 7) || (p.k_rs & 7) || (p.q_hs & 7) || (p.k_hs & 7) || (p.q_bs & 7) || (p.k_bs & 7) || (p.o_rs & 7) || (p.o_hs & 7) || (p.o_bs & 7)) return -3;
    if ((reinterpret_cast<uintptr_t>(p.q) | reinterpret_cast<uintptr_t>(p.k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(p.o) |
         reinterpret_cast<uintptr_t>(tiles)) & 15)
        return -3;
    const int NT = p.Tk / 64;
    if (ir_launch_zero_f32(reinterpret_cast<float*>(p.ovf_flag), 1, s)) return -1;
    hipLaunchKernelGGL(attn_fp8_prep_kernel, dim3(NT, p.Hh, p.B), dim3(256), 0, s, p.k, v, tiles, p.k_bs, p.k_rs, p.k_hs, NT);
    AttnF8Params f;
    f.q = p.q; f.tiles = tiles; f.o = p.o; f.q_bs = p.q_bs; f.o_bs = p.o_bs; f.q_rs = p.q_rs; f.o_rs = p.o_rs; f.q_hs = p.q_hs; f.o_hs = p.o_hs;
    f.Hh = p.Hh; f.Tq = p.Tq; f.Tk = p.Tk; f.scale_log2 = p.scale_log2; f.ovf_flag = p.ovf_flag;
    hipLaunchKernelGGL(flash_attn_fp8_kernel, dim3((p.Tq + 255) / 256, p.Hh, p.B), dim3(256), 0, s, f);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
