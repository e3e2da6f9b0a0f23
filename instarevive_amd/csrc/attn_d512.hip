// Single-head attention with head dim 512 for gfx950: the VAE mid-block AttnBlock at full image size (reference
// ldm/modules/diffusionmodules/model.py:181-205; 65 536 tokens at 2048 x 2048, 4 T^2 512 = 8.8 TFLOP per call), without the redundant
// score product of flash_attn_d512_kernel<2> (attention.hip), which splits the 512 output dims over two workgroups and computes
// S^T twice (1.5x the MFMA work).
//
// One wave per SIMD with the whole 512-register file: a wave owns 32 queries and ALL 512 output dims.
//   O^T accumulators  16 tiles x 16 = 256 registers = the whole AGPR file, addressed literally (a[16*dt : 16*dt+15]) by the inline-asm
//                     MFMAs; hipcc never sees them as values, so it can neither spill nor shuffle them (left to it, 256 "+a"
//                     operands plus 128 Q registers did not allocate without scratch);
//   Q^T fragments     32 k-steps x 4 = 128 arch VGPRs (B operand), pre-multiplied by scale * log2(e);
//   everything the VALU touches (scores, probabilities, LDS fragments, addresses) in the remaining arch VGPRs.
// Same transposed formulation as the other attention kernels: S^T = K Q^T (query on the lane, its keys in the lane's registers),
// O^T = V^T P^T with the bf16-packed S^T accumulators directly as the B operand (K rows fetched swap23-permuted).
// Tiles of 32 keys; K (32 x 1 KB rows, padded to 1040 B in LDS) and V^T (512 x 64 B rows, 16-byte chunks XOR-swizzled) each
// double-buffered in LDS (129 KB) and filled by LDS-DMA. Per tile ONE pinned stream of 64 MFMAs:
//     32 x QK^T(t+1) | 32 x PV(t)
// with everything else riding in the MFMA shadows: the 16 DMA pieces of K(t+2) / V(t+1) behind every other QK^T MFMA (so they
// have the whole PV half to land), the fragment reads four MFMAs ahead (also across the QK^T -> PV seam), and the softmax of tile
// t+1 (one element per MFMA: subtract, v_exp, add, pack) behind the PV MFMAs of tile t, starting three MFMAs after the last QK^T
// MFMA has issued. One barrier per tile.
// The softmax reference is FIXED after the first tile (its row maximum + 2^24 headroom): later probabilities are exp2(score - m)
// whatever they are (fp32 and bf16 have the exponent range), so O^T is never rescaled and the 256 accumulators are only ever touched
// by MFMAs. A query whose scores outgrow the reference by 2^80 raises ovf_flag; the launcher runs the rescaling kernel of
// attention.hip behind this one, which returns at once unless the flag is set.
// V^T comes tile-major ([T/32][512][32 keys], swizzle baked in) from transpose_v_tiles_kernel below, so a tile is 32 contiguous KB
// and its LDS image is its memory image.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"
#include "agpr256.h"
#include <type_traits>
#include <utility>

typedef __attribute__((address_space(3))) void* a5_lds_t;
IR_DEVINL void a5_glds16(const void* g, a5_lds_t l) { __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0); }
IR_DEVINL int a5_swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

namespace a5 {
constexpr int D = 512, NKS = D / 16, NDT = D / 32, TK = 32;
constexpr int KROW = 1024 + 16;          // one DMA instruction per K row, so rows can be padded: (key * 65 + chunk) % 16 is conflict-free
constexpr int KSLOT = TK * KROW;         // 33 280 B
constexpr int VSLOT = D * TK * 2;        // 32 768 B
constexpr int V_OFF = 2 * KSLOT;
constexpr int LDS_BYTES = V_OFF + 2 * VSLOT;   // 132 096 B (the epilogue stages 128 queries x 1 KB of O in the same memory)
constexpr float MARGIN = 24.0f;          // headroom below the first tile's maximum: probabilities of that tile are <= 2^-24
constexpr float OVF_LIMIT = 80.0f;       // a later score may exceed the reference by 2^80 before the fallback is needed
constexpr float RETRIG = 48.0f;          // flash_attn_d512_v2_kernel moves its reference in place once a score lies 2^48 above it (probabilities <= 2^48:
                                         // far inside fp32 / bf16 range, so nothing has overflowed when the check at the end of a tile sees it ...
                                         // unless ONE tile jumps by more than 2^128, which the in-place path handles by recomputing that tile)
#ifndef IR_D512_LA
#define IR_D512_LA 4
#endif
constexpr int LA = IR_D512_LA, NB = LA + 3;   // fragment reads in flight ahead of their MFMA; fragment register sets (see flash_attn_pp_kernel)
#ifndef IR_D512_DMA_EVERY
#define IR_D512_DMA_EVERY 2
#endif
constexpr int DMA_EVERY = IR_D512_DMA_EVERY;
constexpr int SM0 = 35;                  // first stream step that carries a softmax slice: three PV MFMAs behind the last QK^T MFMA
}  // namespace a5

// V [B][T][512] (token stride rs) -> V^T tiles [B][T/32][512][32] bf16; chunk c (keys 8c .. 8c+7) of row d at 16-byte slot c ^ ((d >> 2) & 3)
__global__ __launch_bounds__(256) void transpose_v_tiles_kernel(const bf16_t* __restrict__ v, bf16_t* __restrict__ vt, int rs, long v_bs,
                                                                long vt_bs, const int* __restrict__ only_if) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[32][512 + 8];
    if (only_if && *reinterpret_cast<volatile const int*>(only_if) == 0) return;   // preparation of a fallback that is not needed
    const int tid = threadIdx.x, b = blockIdx.y;
    const bf16_t* src = v + (long)b * v_bs + (long)blockIdx.x * 32 * rs;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + 256 * i, row = c >> 6, ch = c & 63;
        *reinterpret_cast<uint4*>(&tile[row][ch * 8]) = *reinterpret_cast<const uint4*>(src + (long)row * rs + ch * 8);
    }
    __syncthreads();
    bf16_t* dst = vt + (long)b * vt_bs + (long)blockIdx.x * (512 * 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i, d = idx >> 2, slot = idx & 3, c = slot ^ ((d >> 2) & 3);
        uint4 w;
        w.x = (uint32_t)tile[8 * c + 0][d] | ((uint32_t)tile[8 * c + 1][d] << 16);
        w.y = (uint32_t)tile[8 * c + 2][d] | ((uint32_t)tile[8 * c + 3][d] << 16);
        w.z = (uint32_t)tile[8 * c + 4][d] | ((uint32_t)tile[8 * c + 5][d] << 16);
        w.w = (uint32_t)tile[8 * c + 6][d] | ((uint32_t)tile[8 * c + 7][d] << 16);
        *reinterpret_cast<uint4*>(dst + d * 32 + slot * 8) = w;
    }
}

// O^T tile DT += A (VGPR) x B (VGPR), accumulating in place in a[16*DT : 16*DT+15]
template <int DT>
IR_DEVINL void a5_mfma_pv(bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(16 * DT), "n"(16 * DT + 15));
}
template <int W>
IR_DEVINL void a5_set_word(uint4& v, uint32_t x) {
    if constexpr (W == 0) v.x = x;
    else if constexpr (W == 1) v.y = x;
    else if constexpr (W == 2) v.z = x;
    else v.w = x;
}
template <int I>
IR_DEVINL float a5_acc_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "n"(I));
    return x;
}

__global__ __launch_bounds__(256, 1) void flash_attn_d512_v2_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                    const bf16_t* __restrict__ vt, bf16_t* __restrict__ o, int T, int rs,
                                                                    int o_rs, long qk_bs, long vt_bs, long o_bs, float scale_log2,
                                                                    int* __restrict__ ovf_flag, const int* __restrict__ only_if) {
    using namespace a5;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    // launched behind flash_attn_d512_fp8_kernel as ITS fallback (round 5): nothing to do unless that kernel flagged a query it could not handle
    if (only_if && *reinterpret_cast<volatile const int*>(only_if) == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 128 + wid * 32;
    const long b = blockIdx.y;
    q += b * qk_bs; k += b * qk_bs; vt += b * vt_bs; o += b * o_bs;
    const int NT = T >> 5;

    // the 256 accumulators: zeroed here, by the one statement that tells hipcc (and the kernel descriptor) that they are in use
    asm volatile(".set ir_a5_i, 0\n\t.rept 256\n\tv_accvgpr_write_b32 a[ir_a5_i], 0\n\t.set ir_a5_i, ir_a5_i + 1\n\t.endr" ::: IR_AGPR256_CLOBBERS);

    // LDS-DMA pieces (1 KB per wave instruction): K rows wu + 4i and V^T pieces wu + 4i of a tile, i = 0..7
    const bf16_t* k_lane = k + (long)wu * rs + lane * 8;
    const bf16_t* v_lane = vt + wu * 512 + lane * 8;
    const long k_tile = 32L * rs, k_step = 4L * rs;
    // A piece = wave-uniform 64-bit base (scalar registers) + the lane's 32-bit byte offset (lane * 16, the same for every piece) in the
    // `global_load_lds_dwordx4 v_off, s[base]` form, written out: through the builtin hipcc keeps a 64-bit per-lane address and adds the
    // tile / piece term on the VALU in front of every piece (a v_lshl_add_u64 per piece in the busiest gaps of an issue-bound stream).
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const uint32_t lds0_dma = lds_addr(smem);
    auto dma_piece = [&](const bf16_t* base, uint32_t lds_byte) {
        const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0_dma + lds_byte));
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(base), "s"(m0v) : "memory");
    };
    const bf16_t* k_wave = k + (long)wu * rs;
    const bf16_t* v_wave = vt + wu * 512;
    auto k_piece = [&](int tile, int i, int slot) {
#ifdef IR_D512_BUILTIN_DMA
        a5_glds16(k_lane + tile * k_tile + i * k_step, (a5_lds_t)(smem + slot * KSLOT + (wu + 4 * i) * KROW));
#else
        dma_piece(k_wave + tile * k_tile + i * k_step, (uint32_t)(slot * KSLOT + (wu + 4 * i) * KROW));
#endif
    };
    auto v_piece = [&](int tile, int i, int slot) {
#ifdef IR_D512_BUILTIN_DMA
        a5_glds16(v_lane + (long)tile * (512 * 32) + i * 2048, (a5_lds_t)(smem + V_OFF + slot * VSLOT + (wu + 4 * i) * 1024));
#else
        dma_piece(v_wave + (long)tile * (512 * 32) + i * 2048, (uint32_t)(V_OFF + slot * VSLOT + (wu + 4 * i) * 1024));
#endif
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        k_piece(0, i, 0);
        v_piece(0, i, 0);
    }
    if (NT > 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) k_piece(1, i, 1);
    }
    // Q^T fragments straight from HBM in the MFMA B-operand layout (lane = query, 8 consecutive d per k-step half), scaled
    bf16x8 qf[NKS];
    {
        const bf16_t* qrow = q + (long)min(q0 + r, T - 1) * rs + h * 8;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const uint4 v = *reinterpret_cast<const uint4*>(qrow + ks * 16);
            uint4 w;
            w.x = pack2bf(bflo(v.x) * scale_log2, bfhi(v.x) * scale_log2);
            w.y = pack2bf(bflo(v.y) * scale_log2, bfhi(v.y) * scale_log2);
            w.z = pack2bf(bflo(v.z) * scale_log2, bfhi(v.z) * scale_log2);
            w.w = pack2bf(bflo(v.w) * scale_log2, bfhi(v.w) * scale_log2);
            qf[ks] = __builtin_bit_cast(bf16x8, w);
        }
    }
    // LDS fragment addresses (bytes): one base per operand plus compile-time immediates
    const uint32_t lds0 = lds_addr(smem);
    const uint32_t k_addr = lds0 + a5_swap23(r) * KROW + h * 16;                                   // + slot*KSLOT + ks*32
    const uint32_t v_addr0 = lds0 + V_OFF + r * 64 + (((0 + h) ^ ((r >> 2) & 3)) << 4);            // + slot*VSLOT + dt*2048 ; s2 = 0
    const uint32_t v_addr1 = lds0 + V_OFF + r * 64 + (((2 + h) ^ ((r >> 2) & 3)) << 4);            // s2 = 1

    f32x16 sacc;
    uint4 pbA[2], pbB[2];   // P^T fragments (8 bf16 each) of the tile being multiplied / the tile being exponentiated
    bf16x8 fr[NB];
    float p_hold = 0.f;
    float m_ref = 0.f, l_i = 0.f, ovf = -INFINITY;
    uint32_t ka = k_addr, va0 = v_addr0, va1 = v_addr1;

    // One stream step. j in [0, 32): QK^T k-step j of the tile in K slot `ka`; j in [32, 64): PV MFMA (dt = (j-32) >> 1, s2 = j & 1)
    // of the tile in V slot `va*` with P^T fragments pc[]; pn[] receives the probabilities of the scores in sacc (SOFTMAX).
    auto frag_read = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < 32) fr[j % NB] = lds_read16<j * 32>(ka);
        else if constexpr ((j & 1) == 0) fr[j % NB] = lds_read16<((j - 32) >> 1) * 2048>(va0);
        else fr[j % NB] = lds_read16<((j - 32) >> 1) * 2048>(va1);
    };
    auto stream = [&](auto j0c, auto j1c, auto smc, auto dmac, uint4 (&pc)[2], uint4 (&pn)[2], int t, int kslot_n, int vslot_n) {
        constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;
        constexpr bool SOFTMAX = decltype(smc)::value, DMA = decltype(dmac)::value;
        auto step = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + LA < J1) frag_read(std::integral_constant<int, j + LA>{});
            // one counted wait per TWO MFMAs (at the even one, covering the next fragment as well): a satisfied s_waitcnt still takes an issue
            // slot, and the gaps that carry a DMA piece or a softmax slice are over their 24 free cycles already
            constexpr int rem = J1 - 1 - j;
            if constexpr (((j - J0) & 1) == 0) wait_lds<(rem < LA ? (rem > 0 ? rem - 1 : 0) : LA - 1)>();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(sacc) : "v"(fr[j % NB]), "v"(qf[0]));
            else if constexpr (j < 32) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sacc) : "v"(fr[j % NB]), "v"(qf[j]));
            else a5_mfma_pv<((j - 32) >> 1)>(fr[j % NB], __builtin_bit_cast(bf16x8, pc[j & 1]));
            if constexpr (j - 2 >= J0) asm volatile("" ::"v"(fr[(j - 2) % NB]));  // keep the fragment of MFMA j-2 allocated until here
            __builtin_amdgcn_sched_barrier(0);
            // The next tiles' 16 pieces of this wave, one behind every other QK^T MFMA (DMA_EVERY = 2). Measured in round 4: spreading them over
            // the tile (every 3rd / 4th MFMA) is SLOWER (6.53 -> 6.70 / 6.84 ms at 65536 tokens) - the gaps of the QK^T half carry nothing but a
            // fragment read, the later ones carry the softmax slices and are over their 24 free issue cycles already.
            if constexpr (DMA && (j % DMA_EVERY) == 1 && j / DMA_EVERY < 16) {
                constexpr int pi = j / DMA_EVERY;
                // a full stream runs only while t + 1 < NT, so V^T(t+1) exists; past the end K(t+2) re-reads the last tile into the free
                // slot (never used) instead of branching around the DMA
                if constexpr (pi < 8) k_piece(min(t + 2, NT - 1), pi, kslot_n);
                else v_piece(t + 1, pi - 8, vslot_n);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (SOFTMAX && j >= SM0 && j < SM0 + 16) {  // probability of score e (keys of tile t+1), one per MFMA shadow
                constexpr int e = j - SM0;
                const float sv = sacc[e];
                ovf = fmaxf(ovf, sv);
                const float p = __builtin_amdgcn_exp2f(sv - m_ref);
                l_i += p;
                if constexpr (e & 1) a5_set_word<((e & 7) >> 1)>(pn[e >> 3], pack2bf_trans(p_hold, p));   // one v_cvt_pk behind the transcendental-use wait state (p is fresh from v_exp)
                else p_hold = p;
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) { (frag_read(std::integral_constant<int, J0 + I>{}), ...); }(std::make_integer_sequence<int, LA>{});
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) { (step(std::integral_constant<int, J0 + I>{}), ...); }(std::make_integer_sequence<int, J1 - J0>{});
    };
    using I0 = std::integral_constant<int, 0>;
    using I32 = std::integral_constant<int, 32>;
    using I64 = std::integral_constant<int, 64>;

    // ---- tile 0: S^T(0), then its softmax in the open (this is where the reference is fixed)
    wait_dma();
    __syncthreads();
    stream(I0{}, I32{}, std::false_type{}, std::false_type{}, pbA, pbA, 0, 0, 0);
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(sacc));  // MFMA results -> VALU
    {
        float mx = -INFINITY;
#pragma unroll
        for (int g = 0; g < 16; ++g) mx = fmaxf(mx, sacc[g]);
        m_ref = xhalf_max(mx) + MARGIN;
        float pv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            pv[e] = __builtin_amdgcn_exp2f(sacc[e] - m_ref);
            l_i += pv[e];
        }
        pbA[0] = make_uint4(pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3]), pack2bf(pv[4], pv[5]), pack2bf(pv[6], pv[7]));
        pbA[1] = make_uint4(pack2bf(pv[8], pv[9]), pack2bf(pv[10], pv[11]), pack2bf(pv[12], pv[13]), pack2bf(pv[14], pv[15]));
        pbB[0] = pbB[1] = make_uint4(0, 0, 0, 0);
    }
    // ---- main loop, two tiles per trip so that the P^T buffers swap roles statically
    auto tile_step = [&](int t, uint4 (&pc)[2], uint4 (&pn)[2]) {
        wait_dma();        // this wave's pieces of K(t+1) / V(t), issued a whole PV half (or more) ago
        __syncthreads();   // ... published; and every wave is done with the slots the pieces of this trip overwrite
        const int kcur = (t + 1) & 1, vcur = t & 1;
        ka = k_addr + kcur * KSLOT;
        va0 = v_addr0 + vcur * VSLOT;
        va1 = v_addr1 + vcur * VSLOT;
        const float l_snap = l_i;   // the denominator before this trip's softmax slices (the re-referencing path below starts from it)
        if (t + 1 < NT) stream(I0{}, I64{}, std::true_type{}, std::true_type{}, pc, pn, t, t & 1, (t + 1) & 1);
        else stream(I32{}, I64{}, std::false_type{}, std::false_type{}, pc, pn, t, 0, 0);
        // ---- Round 5: the reference moves IN PLACE when a later tile outgrows it (rare, wave-uniform branch; 3 instructions per tile otherwise).
        // Before, a score 2^80 above the reference raised ovf_flag and the rescaling 4-wave kernel recomputed the WHOLE launch: 11.8 ms at 65536
        // tokens, already with the encoder's logits times 4 on the bench image (profiles/r05_stress_parity.txt) - and released SD-VAE weights are
        // the ones known for extreme logits. Here: once a score of tile t + 1 lies RETRIG above the reference, the wave waits for PV(t), multiplies
        // its 256 accumulators and the denominator by 2^-K (K a whole number per query: every rescaling is EXACT, results do not depend on
        // whether or when it happens beyond the usual fp32 rounding of later sums), sets the reference to the row maximum so far + MARGIN and
        // recomputes the probabilities of tile t + 1 from the scores still in sacc (the ones the stream made may have overflowed).
        if (t + 1 < NT && __any(ovf - m_ref > RETRIG)) {
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(sacc) : : "memory");   // PV(t) has written the accumulators; the scores are readable
            const float top = xhalf_max(ovf);                                  // both key halves of a query agree
            const float K = fmaxf(ceilf(top + MARGIN - m_ref), 0.f);
            const int nk = -(int)fminf(K, 1.0e6f);
            float tmp;
            asm volatile(".set ir_a5_r, 0\n\t.rept 256\n\tv_accvgpr_read_b32 %0, a[ir_a5_r]\n\ts_nop 0\n\tv_ldexp_f32 %0, %0, %1\n\ts_nop 0\n\t"
                         "v_accvgpr_write_b32 a[ir_a5_r], %0\n\t.set ir_a5_r, ir_a5_r + 1\n\t.endr\n\ts_nop 7"
                         : "=&v"(tmp) : "v"(nk) : "memory");
            m_ref += K;
            float l_new = __builtin_ldexpf(l_snap, nk);
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const float p0 = __builtin_amdgcn_exp2f(sacc[e] - m_ref), p1 = __builtin_amdgcn_exp2f(sacc[e + 1] - m_ref);
                l_new += p0 + p1;
                const uint32_t w = pack2bf(p0, p1);
                if (e == 0) pn[0].x = w; else if (e == 2) pn[0].y = w; else if (e == 4) pn[0].z = w; else if (e == 6) pn[0].w = w;
                else if (e == 8) pn[1].x = w; else if (e == 10) pn[1].y = w; else if (e == 12) pn[1].z = w; else pn[1].w = w;
            }
            l_i = l_new;
        }
    };
    for (int t = 0; t < NT; t += 2) {
        tile_step(t, pbA, pbB);
        if (t + 1 < NT) tile_step(t + 1, pbB, pbA);
    }
    // ---- finalise: O^T / l -> LDS [q][512] bf16 (32 KB per wave) -> 16-byte row stores
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");  // the last MFMA results -> v_accvgpr_read
    l_i += __shfl_xor(l_i, 32);
    if (__any(ovf - m_ref > OVF_LIMIT) && lane == 0 && ovf_flag) atomicOr(ovf_flag, 1);
    const float inv = 1.0f / l_i;
    __syncthreads();  // every wave has finished reading the K / V^T ring
    constexpr int OROW = D * 2;
    unsigned char* ow = smem + wid * 32 * OROW;
    [&]<int... DT>(std::integer_sequence<int, DT...>) {
        ([&] {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                float x0, x1, x2, x3;
                if (gg == 0) { x0 = a5_acc_read<16 * DT + 0>(); x1 = a5_acc_read<16 * DT + 1>(); x2 = a5_acc_read<16 * DT + 2>(); x3 = a5_acc_read<16 * DT + 3>(); }
                else if (gg == 1) { x0 = a5_acc_read<16 * DT + 4>(); x1 = a5_acc_read<16 * DT + 5>(); x2 = a5_acc_read<16 * DT + 6>(); x3 = a5_acc_read<16 * DT + 7>(); }
                else if (gg == 2) { x0 = a5_acc_read<16 * DT + 8>(); x1 = a5_acc_read<16 * DT + 9>(); x2 = a5_acc_read<16 * DT + 10>(); x3 = a5_acc_read<16 * DT + 11>(); }
                else { x0 = a5_acc_read<16 * DT + 12>(); x1 = a5_acc_read<16 * DT + 13>(); x2 = a5_acc_read<16 * DT + 14>(); x3 = a5_acc_read<16 * DT + 15>(); }
                const uint2 w = make_uint2(pack2bf(x0 * inv, x1 * inv), pack2bf(x2 * inv, x3 * inv));
                // row r (query), 8-byte chunk (dt*32 + 8gg + 4h) / 4; XOR with the row spreads the 32 rows over the banks
                const int c8 = (DT * 8 + 2 * gg + h) ^ (r & 31);
                *reinterpret_cast<uint2*>(ow + r * OROW + c8 * 8) = w;
            }
        }(), ...);
    }(std::make_integer_sequence<int, NDT>{});
    __syncthreads();
    constexpr int OCH = D / 8;  // 16-byte chunks per staged row
    for (int c = lane; c < 32 * OCH; c += 64) {
        const int row = c / OCH, ch = c % OCH;
        const int qq = q0 + row;
        const uint2 lo = *reinterpret_cast<const uint2*>(ow + row * OROW + (((2 * ch) ^ (row & 31)) * 8));
        const uint2 hi = *reinterpret_cast<const uint2*>(ow + row * OROW + (((2 * ch + 1) ^ (row & 31)) * 8));
        if (qq < T) *reinterpret_cast<uint4*>(o + (long)qq * o_rs + ch * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}

int ir_launch_transpose_v_tiles(const bf16_t* v, bf16_t* vt, int B, int T, int rs, long v_bs, long vt_bs, hipStream_t s, const int* only_if) {
    if (T <= 0 || (T & 31) || (rs & 7) || rs < 512 || B <= 0) return -2;
    hipLaunchKernelGGL(transpose_v_tiles_kernel, dim3(T / 32, B), dim3(256), 0, s, v, vt, rs, v_bs, vt_bs, only_if);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int ir_launch_flash_attn_d512_v2(const bf16_t* q, const bf16_t* k, const bf16_t* vt_tiles, bf16_t* o, int B, int T, int rs, int o_rs, long qk_bs,
                                 long vt_bs, long o_bs, float scale, int* ovf_flag, hipStream_t s, const int* only_if) {
    if (T <= 0 || (T & 31) || (rs & 7) || (o_rs & 7) || rs < 512 || o_rs < 512 || B <= 0 || !ovf_flag) return -2;
    hipLaunchKernelGGL(flash_attn_d512_v2_kernel, dim3((T + 127) / 128, B), dim3(256), 0, s, q, k, vt_tiles, o, T, rs, o_rs, qk_bs, vt_bs, o_bs,
                       scale * 1.44269504088896340736f, ovf_flag, only_if);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
// The same for `rows` query rows starting at q / o (both already offset by the caller) against all T keys of ONE image: the query-row shard
// of a rank when the encoder's mid-block attention of a large frame is split over several GPUs. rows % 128 == 0; every row's result is the
// one the full launch computes (a workgroup owns 128 queries and shares nothing with the others).
int ir_launch_flash_attn_d512_v2_rows(const bf16_t* q, const bf16_t* k, const bf16_t* vt_tiles, bf16_t* o, int T, int rows, int rs, int o_rs,
                                      float scale, int* ovf_flag, hipStream_t s) {
    if (T <= 0 || (T & 31) || rows <= 0 || (rows & 127) || rows > T || (rs & 7) || (o_rs & 7) || rs < 512 || o_rs < 512 || !ovf_flag) return -2;
    hipLaunchKernelGGL(flash_attn_d512_v2_kernel, dim3(rows / 128, 1), dim3(256), 0, s, q, k, vt_tiles, o, T, rs, o_rs, 0L, 0L, 0L,
                       scale * 1.44269504088896340736f, ovf_flag, (const int*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// =====================================================================================================================
// flash_attn_pp2_kernel<72>: the DiT self-attention (16 heads x 72, T % 64 == 0, no key bias; reference PixArt_blocks.py:123-158) in the
// structure of flash_attn_d512_v2_kernel above - ONE wave per SIMD with the whole register file, one pinned MFMA stream per tile,
// everything else in the MFMA shadows - instead of the two-waves-per-SIMD ping-pong of flash_attn_pp_kernel (attention.hip), whose
// matrix and vector segments are kept complementary by two workgroup barriers per tile (phase stamps: the waves wait at them for
// about a third of a tile).
// A wave owns TWO groups of 32 queries: O^T (2 x 3 tiles) and the Q^T fragments (2 x 5 k-steps) live in AGPRs, addressed literally.
// Per 64-key tile one stream of 44 MFMAs: S^T(t+1) of group 0, of group 1 (10 each: -m rides in as the C operand of the first),
// then O^T += V^T(t) P^T(t) for both groups (24, the V^T fragments shared by the two groups). The exponentials of tile t+1 (no
// subtraction, no row sum - the denominator comes from the ones row of V^T - no running maximum) follow three MFMAs behind the
// last score MFMA of their group, two per MFMA shadow; the 19 LDS-DMA pieces of K(t+2) / V^T(t+1) ride behind the score MFMAs. One
// barrier per tile, K and V^T double-buffered (42 KB of LDS). Same math as the other kernels: S^T = K Q^T with swap23-permuted K
// rows, P^T registers as the B operand of the second product, reference fixed after the first tile (row maximum + 2^24 headroom).
// Overflow is detected at the end: a denominator that is not a moderate finite number raises ovf_flag and the rescaling 4-wave
// kernel, launched behind, recomputes everything.
// Knock-out builds of flash_attn_pp2_kernel (diagnostic, -DIR_KO_PP2=n, results wrong by design): 1 no per-tile wait + barrier, 2 no LDS-DMA in the
// stream, 3 no exponentials, 4 no fragment reads, 5 no MFMAs
#ifndef IR_KO_PP2
#define IR_KO_PP2 0
#endif
// Experiment knob (-DIR_PP2_TRUNC=1): the probabilities go to bf16 by truncation (one full-rate v_perm_b32 per pair instead of the quarter-rate
// v_cvt_pk_bf16_f32: 128 issue cycles less per tile). Correct and as accurate as rounding (see pp2_pack), 6 % faster when the op runs alone
// (1.349 -> 1.264 ms at 16384 tokens) - and NO faster inside the pipeline (30.15 against 30.16 ms for the 28 layers, A/B twice on one box): there
// the kernel sits at the power-limited clock (about 1.8 GHz with the matrix pipe 75 % busy), where the time follows the energy of the MFMAs, not
// the issue slots beside them. Default off: rounding is what every other kernel does.
#ifndef IR_PP2_TRUNC
#define IR_PP2_TRUNC 0
#endif
namespace pp2 {
constexpr int D = 72, NKS = 5, NDT = 3, RCH = 9;
constexpr int KROW = RCH * 16;              // 144-byte K rows, unpadded (9 chunks: conflict-free)
constexpr int KSLOT = 64 * KROW;            // 9216 B = 9 DMA pieces
constexpr int VROWS = 96, VSLOT = VROWS * 128;   // V^T tile: 96 rows x 64 keys (rows 73.. are never loaded and never stored)
constexpr int K_Q = 9, V_Q = (D + 1 + 7) / 8;    // DMA pieces per tile: 9 + 10
constexpr int V_OFF = 2 * KSLOT;
constexpr int LDS_MAIN = V_OFF + 2 * VSLOT;      // 43 008 B
constexpr int OS = 96 + 8;                       // O staging row stride (elements)
constexpr int LDS_O = 8 * 32 * OS * 2;           // 53 248 B
constexpr int LDS_BYTES = LDS_MAIN > LDS_O ? LDS_MAIN : LDS_O;
constexpr int NPC = (K_Q + V_Q + 3) / 4;         // pieces per wave and tile (at most)
constexpr float MARGIN = 24.0f;
#ifndef IR_PP2_DMA_EVERY
#define IR_PP2_DMA_EVERY 2
#endif
constexpr int DMA_EVERY = IR_PP2_DMA_EVERY;      // one LDS-DMA piece behind every DMA_EVERY-th MFMA of the stream: 2 = behind the first score MFMAs, whose gaps
                                                 // carry no exponentials (6 / 8: 1.29 -> 1.30 / 1.32 ms per layer, measured in round 4)
constexpr int LA = 6, NB = LA + 3;
constexpr int NQK = 20, NPV = 24, NSTEP = NQK + NPV;
constexpr int O_BASE = 0, Q_BASE = 96;           // AGPR map: O^T a[0:95] (group g, tile dt at 16*(3g+dt)), Q^T a[96:135] (4*(5g+ks))
}  // namespace pp2

// Softmax work list of a tile: item n (0..63) -> (query group, score element). The first 20 items are group 0's elements 0..19 (group 1's
// scores are not finished yet); then three blocks of 12 = 4 of group 0 + 8 of group 1, then the last 8 of group 1. Chunks are even-sized and
// start at even elements, so the two halves of a bf16 pair are always consecutive items (n even, n odd).
constexpr int pp2_item_g(int n) { return n < 20 ? 0 : (((n - 20) / 12 < 3 && (n - 20) % 12 < 4) ? 0 : 1); }
constexpr int pp2_item_e(int n) {
    if (n < 20) return n;
    const int m = n - 20, blk = m / 12, pos = m % 12;
    if (blk < 3 && pos < 4) return 20 + 4 * blk + pos;
    return blk < 3 ? 8 * blk + pos - 4 : 24 + (m - 36);
}
template <int LO>
IR_DEVINL void pp2_mfma_qk_first(f32x16& s, bf16x8 a, const f32x16& c) {  // s = a x Q^T(AGPR) + c, s and c in different registers
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %4" : "=&v"(s) : "v"(a), "n"(LO), "n"(LO + 3), "v"(c));
}
template <int LO>
IR_DEVINL void pp2_mfma_qk(f32x16& s, bf16x8 a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(s) : "v"(a), "n"(LO), "n"(LO + 3));
}
template <int LO>
IR_DEVINL void pp2_mfma_pv(bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(LO), "n"(LO + 15));
}
template <int I>
IR_DEVINL void pp2_acc_write(uint32_t x) {
    asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(x), "n"(I));
}

__global__ __launch_bounds__(256, 1) void flash_attn_pp2_kernel(AttnParams p) {
    using namespace pp2;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int r = lane & 31, h = lane >> 5;
    // Workgroup -> (head, query tile). gridDim.y == 1 with several heads: XCD-aware (launcher: Hh % 8 == 0). Workgroups are dealt round-robin over the 8 XCDs, so
    // b % 8 names the XCD; all query tiles of a head go to ONE XCD, whose 32 CUs then stream that head's K / V^T (4.7 MB at 16384 tokens)
    // through its 4 MB L2 together instead of every XCD streaming every head (round 3: 751 MB of fabric reads per launch for 151 MB of operands).
    int qt = blockIdx.x, head = blockIdx.y;
    if (gridDim.y == 1 && p.Hh > 1) {   // (a one-head launch has gridDim.y == 1 as well: the plain map)
        const int QT = (p.Tq + 255) >> 8, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        head = (j / QT) * 8 + xcd;
        qt = j - (j / QT) * QT;
    }
    const int q0 = qt * 256 + wid * 64, b = blockIdx.z;
    const bf16_t* qp = p.q + (long)b * p.q_bs + (long)head * p.q_hs;
    const bf16_t* kp = p.k + (long)b * p.k_bs + (long)head * p.k_hs;
    const bf16_t* vtp = p.vt + (long)b * p.vt_bs + (long)head * 96 * p.Tk_pad;
    const int NT = p.Tk >> 6;

    asm volatile(".set ir_pp2_i, 0\n\t.rept 96\n\tv_accvgpr_write_b32 a[ir_pp2_i], 0\n\t.set ir_pp2_i, ir_pp2_i + 1\n\t.endr" ::: IR_AGPR256_CLOBBERS);

    // LDS-DMA pieces of a tile pair {K, V^T}: piece idx = wave + 4k belongs to this wave (19 pieces: the fifth piece of wave 3 repeats
    // its fourth - same bytes to the same place - so that the stream issues its pieces without a branch). A piece's source is a wave-uniform
    // 64-bit base (the tile's first K row / V^T column: two scalars per TILE, set by tile_bases) plus the lane's 32-bit byte offset, which
    // never changes - the `global_load_lds ... v_off, s[base]` form: no 64-bit vector add and no per-piece tile arithmetic in the stream
    // (round 3 recomputed tile * stride per piece: 13 scalar + 1 vector instruction in ONE MFMA gap, about 40 idle matrix cycles per piece).
    // Pieces k = 0, 1 are always K pieces, k = 3, 4 always V^T pieces, k = 2 is a K piece on wave 0 only.
    uint32_t pc_off[NPC];
    int pc_dst[NPC];
    const bool k2_is_k = wu + 8 < K_Q;
    const uint32_t lds0_early = lds_addr(smem);
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
        const int idx = min(wu + 4 * k, K_Q + V_Q - 1);
        pc_dst[k] = idx < K_Q ? idx * 1024 : V_OFF + (idx - K_Q) * 1024;   // wave-uniform (kept out of the lane-dependent branches: a scalar)
        if (idx < K_Q) {
            const int ci = 64 * idx + lane, row = ci / RCH, ch = ci - row * RCH;
            pc_off[k] = (uint32_t)(row * p.k_rs + ch * 8) * 2u;
        } else {
            const int j = idx - K_Q;
            const int d = 8 * j + (lane >> 3), c = (lane & 7) ^ ((d >> 1) & 7);
            pc_off[k] = (uint32_t)(d * p.Tk_pad + c * 8) * 2u;
        }
    }
    const unsigned char* kbase = reinterpret_cast<const unsigned char*>(kp);    // K rows of the tile being fetched
    const unsigned char* vbase = reinterpret_cast<const unsigned char*>(vtp);   // V^T columns of the tile being fetched
    int kslot = 0, vslot = 0;                                                    // LDS slot byte offsets of those tiles
    auto tile_bases = [&](int t) {   // fetches that ride behind PV tile t: K(t + 2), V^T(t + 1); past the end the last tile again (never used)
        const int tk = min(t + 2, NT - 1), tv = min(t + 1, NT - 1);
        kbase = reinterpret_cast<const unsigned char*>(kp + (long)tk * 64 * p.k_rs);
        vbase = reinterpret_cast<const unsigned char*>(vtp + (long)tv * 64);
        kslot = (tk & 1) * KSLOT;
        vslot = (tv & 1) * VSLOT;
    };
    auto issue = [&](auto kc) {  // this wave's k-th piece of the tiles tile_bases() selected
        constexpr int k = decltype(kc)::value;
        const bool isk = k < 2 || (k == 2 && k2_is_k);
        const unsigned char* base = isk ? kbase : vbase;
#ifndef IR_PP2_BUILTIN_DMA
        // scalar base + 32-bit lane offset, written out: hipcc widens pc_off to a register pair and adds the base on the VALU
        const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0_early + (uint32_t)((isk ? kslot : vslot) + pc_dst[k])));
        const uint32_t off = pc_off[k];
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(m0v) : "memory");
#else
        a5_glds16(base + pc_off[k], (a5_lds_t)(smem + (isk ? kslot : vslot) + pc_dst[k]));
#endif
    };
#ifndef IR_PP2_NO_PADZERO
    // Rows 80..95 of the two V^T slots are never loaded (the 10 pieces of a tile end at row 79) but ARE multiplied: the third 32-row tile of O^T = V^T P^T
    // spans rows 64..95. Left as whatever the previous kernel had in LDS they cost the multiplier array as much as real data; as zeros they toggle
    // nothing - and this kernel runs at the power-limited clock (section 3, round 4). 2 x 2 KB, once per workgroup.
    for (int c = tid; c < 2 * 16 * 8; c += 256) {
        const int slot = c >> 7, rem = c & 127;
        *reinterpret_cast<uint4*>(smem + V_OFF + slot * VSLOT + (80 + (rem >> 3)) * 128 + (rem & 7) * 16) = make_uint4(0, 0, 0, 0);
    }
#endif
    // prologue: K(0) -> slot 0 and V^T(0) -> slot 0, then K(1) -> slot 1
    [&]<int... K>(std::integer_sequence<int, K...>) { ((issue(std::integral_constant<int, K>{})), ...); }(std::make_integer_sequence<int, NPC>{});
    if (NT > 1) {
        kbase = reinterpret_cast<const unsigned char*>(kp + 64L * p.k_rs);
        kslot = KSLOT;
        issue(std::integral_constant<int, 0>{});
        issue(std::integral_constant<int, 1>{});
        if (k2_is_k) issue(std::integral_constant<int, 2>{});
    }
    // Q^T fragments -> AGPRs (B operand layout: lane = query, 8 consecutive d per k-step half), scaled; d >= 72 is zero
    [&]<int... I>(std::integer_sequence<int, I...>) {
        ([&] {
            constexpr int g = I / NKS, ks = I % NKS;
            const bf16_t* qrow = qp + (long)min(q0 + g * 32 + r, p.Tq - 1) * p.q_rs;
            const int d0 = ks * 16 + h * 8;
            const uint4 v = *reinterpret_cast<const uint4*>(qrow + (d0 < D ? d0 : 0));
            const float sc = d0 < D ? p.scale_log2 : 0.f;
            pp2_acc_write<Q_BASE + 4 * I + 0>(pack2bf(bflo(v.x) * sc, bfhi(v.x) * sc));
            pp2_acc_write<Q_BASE + 4 * I + 1>(pack2bf(bflo(v.y) * sc, bfhi(v.y) * sc));
            pp2_acc_write<Q_BASE + 4 * I + 2>(pack2bf(bflo(v.z) * sc, bfhi(v.z) * sc));
            pp2_acc_write<Q_BASE + 4 * I + 3>(pack2bf(bflo(v.w) * sc, bfhi(v.w) * sc));
        }(), ...);
    }(std::make_integer_sequence<int, 2 * NKS>{});

    const uint32_t lds0 = lds_addr(smem);
    const uint32_t k_addr = lds0 + a5_swap23(r) * KROW + h * 16;           // + slot*KSLOT + kt*32*KROW + ks*32
    const int vsw = (r >> 1) & 7;
    uint32_t v_addr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v_addr[j] = lds0 + V_OFF + r * 128 + ((((2 * j) | h) ^ vsw) << 4);   // + slot*VSLOT + dt*4096 ; j = kt*2 + s2

    f32x16 sacc[2][2], negm[2];
    uint4 pbA[2][4], pbB[2][4];   // P^T fragments [group][kk] of the tile being multiplied / being exponentiated
    bf16x8 fr[NB];
    float p_hold2[2] = {0.f, 0.f}, pend0[2] = {0.f, 0.f}, pend1[2] = {0.f, 0.f};
    // P -> bf16: 32 packs per tile in an issue-bound stream. Truncation is one full-rate v_perm_b32 instead of the quarter-rate v_cvt_pk; its mean
    // bias (-0.27 %) is common to numerator and denominator (the ones row of V^T sums the SAME operand), what is left has round-to-nearest's variance
    // (16384 tokens, random operands: rms error against fp32 3.88e-5 for 3.80e-5 rounded). Tile 0, computed in the open, keeps the rounding pack: its
    // 64 keys weigh 0.27 % more than the others', 1e-5 of the result at 16384 tokens and 2e-4 at 1024.
    const uint32_t psel = __builtin_amdgcn_readfirstlane(0x07060302u);
    auto pp2_pack0 = [&](float lo, float hi) -> uint32_t { return pack2bf(lo, hi); };   // tile 0 (outside the pinned stream, compiler-scheduled): round to nearest
    auto pp2_pack_last = [&](float lo, float hi) -> uint32_t {
        if constexpr (IR_PP2_TRUNC) return pack2bf_trunc_trans(lo, hi, psel);
        else return pack2bf_valu(lo, hi);
    };
    auto pp2_pack = [&](float lo, float hi) -> uint32_t {
        if constexpr (IR_PP2_TRUNC) {
            return pack2bf_trunc(lo, hi, psel);
        } else return pack2bf_valu(lo, hi);
    };
    if (IR_KO_PP2 == 4) {
#pragma unroll
        for (int i = 0; i < NB; ++i) fr[i] = __builtin_bit_cast(bf16x8, make_uint4(lane, 0x3c003c00, 0x3c003c00, 0x3c003c00));
    }
    uint32_t ka = k_addr, va[4] = {v_addr[0], v_addr[1], v_addr[2], v_addr[3]};
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 16; ++e) negm[g][e] = 0.f;
    asm volatile("s_nop 7" : "+v"(negm[0]), "+v"(negm[1]));   // pinned here: asm MFMAs are invisible to hipcc's hazard recogniser, which otherwise materialises these zeros directly in front of the MFMA that reads them as its C operand (tools/mfma_hazard_scan.py)

    // stream step j: [0, 20) S^T MFMA (group j / 10, key sub-tile (j % 10) / 5, k-step j % 5); [20, 44) PV MFMA (pair (j - 20) >> 1 =
    // dt*4 + kk, group j & 1)
    auto frag_read = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (IR_KO_PP2 == 4) return;
        if constexpr (j < NQK) fr[j % NB] = lds_read16<(((j % 10) / 5) * 32 * KROW + (j % 5) * 32)>(ka);
        else fr[(NQK + ((j - NQK) >> 1)) % NB] = lds_read16<(((j - NQK) >> 3) * 4096)>(va[((j - NQK) >> 1) & 3]);
    };
    // fragment slots: score MFMA j uses slot j; PV pair q uses slot NQK + q (one read for the two groups); reads run LA slots ahead
    auto stream = [&](auto j0c, auto smc, auto dmac, uint4 (&pc)[2][4], uint4 (&pn)[2][4], int t) {
        constexpr int J0 = decltype(j0c)::value;
        constexpr bool SOFTMAX = decltype(smc)::value, DMA = decltype(dmac)::value;
        constexpr int S0 = J0 < NQK ? J0 : NQK + ((J0 - NQK) >> 1), S1 = NQK + NPV / 2;   // fragment slots [S0, S1)
        auto slot_read = [&](auto sc) {
            constexpr int sl = decltype(sc)::value;
            if constexpr (sl < NQK) frag_read(std::integral_constant<int, sl>{});
            else frag_read(std::integral_constant<int, NQK + 2 * (sl - NQK)>{});
        };
        auto step = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int sl = j < NQK ? j : NQK + ((j - NQK) >> 1);     // fragment slot of this MFMA
            constexpr bool first_use = j < NQK || ((j - NQK) & 1) == 0;  // a PV pair's fragment is read once, for its first MFMA
            if constexpr (first_use) {
                if constexpr (sl + LA < S1) slot_read(std::integral_constant<int, sl + LA>{});
                // one counted wait per TWO fragment slots (at the even slot, for the odd one behind it as well): the stream is issue-bound -
                // every instruction between two MFMAs beyond 24 cycles' worth idles the matrix pipe - and a wait that is already
                // satisfied still takes its issue slot
                constexpr int rem = S1 - 1 - sl;   // reads issued behind slot sl by now (capped by the look-ahead)
                if constexpr (((sl - S0) & 1) == 0) wait_lds<(rem < LA ? (rem > 0 ? rem - 1 : 0) : LA - 1)>();
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (IR_KO_PP2 == 5) {
                asm volatile("" ::"v"(fr[sl % NB]));
            } else if constexpr (j < NQK) {
                constexpr int g = j / 10, kt = (j % 10) / 5, ks = j % 5;
                if constexpr (ks == 0) pp2_mfma_qk_first<Q_BASE + 4 * (5 * g + ks)>(sacc[g][kt], fr[sl % NB], negm[g]);
                else pp2_mfma_qk<Q_BASE + 4 * (5 * g + ks)>(sacc[g][kt], fr[sl % NB]);
            } else {
                constexpr int q = (j - NQK) >> 1, g = j & 1, dt = q >> 2, kk = q & 3;
                pp2_mfma_pv<O_BASE + 16 * (3 * g + dt)>(fr[sl % NB], __builtin_bit_cast(bf16x8, pc[g][kk]));
            }
            if constexpr (sl - 2 >= S0) asm volatile("" ::"v"(fr[(sl - 2) % NB]));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DMA && IR_KO_PP2 != 2 && (j % DMA_EVERY) == 1 && j / DMA_EVERY < NPC) {   // K(t+2) / V^T(t+1) pieces, spread over the tile (see a5::DMA_EVERY)
                issue(std::integral_constant<int, (j / DMA_EVERY)>{});
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (SOFTMAX && IR_KO_PP2 != 3 && j >= 13) {   // two exponentials per MFMA shadow: group 0 from step 13 (three MFMAs behind its
                                                                     // last score MFMA, step 9), group 1 from step 23
                constexpr int n0 = j < 23 ? 2 * (j - 13) : 20 + (j - 23) * 44 / 21, n1 = j < 23 ? n0 + 2 : 20 + (j - 22) * 44 / 21;
                [&]<int... E>(std::integer_sequence<int, E...>) {
                    ([&] {
                        constexpr int n = n0 + E;
                        if constexpr (n < n1 && n < 64) {
                            constexpr int g = pp2_item_g(n), e = pp2_item_e(n);
                            const float pv = __builtin_amdgcn_exp2f(sacc[g][e >> 4][e & 15]);
                            if constexpr (e & 1) {
                                // The pair (e-1, e) is packed one pair LATER, behind the next two exponentials: the v_cvt_pk then never waits
                                // for the transcendental pipe. Two pending slots alternate.
                                if constexpr (((n >> 1) & 1) == 0) { pend0[0] = p_hold2[g]; pend0[1] = pv; } else { pend1[0] = p_hold2[g]; pend1[1] = pv; }
                                if constexpr (n >= 3) {
                                    constexpr int gp = pp2_item_g(n - 2), ep = pp2_item_e(n - 2);
                                    if constexpr (((n >> 1) & 1) == 0) a5_set_word<((ep & 7) >> 1)>(pn[gp][ep >> 3], pp2_pack(pend1[0], pend1[1]));
                                    else a5_set_word<((ep & 7) >> 1)>(pn[gp][ep >> 3], pp2_pack(pend0[0], pend0[1]));
                                }
                            } else {
                                p_hold2[g] = pv;
                            }
                        }
                    }(), ...);
                }(std::make_integer_sequence<int, 4>{});
                if constexpr (j == NSTEP - 1) {   // the last pair (item 63, pending slot 1)
                    constexpr int gp = pp2_item_g(63), ep = pp2_item_e(63);
                    a5_set_word<((ep & 7) >> 1)>(pn[gp][ep >> 3], pp2_pack_last(pend1[0], pend1[1]));   // its exponentials are the two just issued
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) { (slot_read(std::integral_constant<int, S0 + I>{}), ...); }(std::make_integer_sequence<int, LA>{});
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) { (step(std::integral_constant<int, J0 + I>{}), ...); }(std::make_integer_sequence<int, NSTEP - J0>{});
    };
    using J0 = std::integral_constant<int, 0>;
    using JPV = std::integral_constant<int, NQK>;

    // ---- tile 0: scores of both groups (C = 0), softmax in the open: the reference is fixed here
    wait_dma();
    __syncthreads();
    {
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                constexpr int g = I / 10, kt = (I % 10) / 5, ks = I % 5;
                const bf16x8 a = lds_read16<(kt * 32 * KROW + ks * 32)>(ka);
                wait_lds<0>();
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ks == 0) pp2_mfma_qk_first<Q_BASE + 4 * (5 * g + ks)>(sacc[g][kt], a, negm[g]);
                else pp2_mfma_qk<Q_BASE + 4 * (5 * g + ks)>(sacc[g][kt], a);
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, NQK>{});
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(sacc[0][0]), "+v"(sacc[0][1]), "+v"(sacc[1][0]), "+v"(sacc[1][1]));
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[g][kt][e]);
        const float m_ref = xhalf_max(mx) + MARGIN;
#pragma unroll
        for (int e = 0; e < 16; ++e) negm[g][e] = -m_ref;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float pv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = __builtin_amdgcn_exp2f(sacc[g][kk >> 1][(kk & 1) * 8 + e] - m_ref);
            pbA[g][kk] = make_uint4(pp2_pack0(pv[0], pv[1]), pp2_pack0(pv[2], pv[3]), pp2_pack0(pv[4], pv[5]), pp2_pack0(pv[6], pv[7]));
            pbB[g][kk] = make_uint4(0, 0, 0, 0);
        }
    }
    // ---- main loop (two tiles per trip: the P^T buffers swap roles statically)
    auto tile_step = [&](int t, uint4 (&pc)[2][4], uint4 (&pn)[2][4]) {
        if (IR_KO_PP2 != 1) {
            wait_dma();
            __syncthreads();
        }
        ka = k_addr + ((t + 1) & 1) * KSLOT;
#pragma unroll
        for (int j = 0; j < 4; ++j) va[j] = v_addr[j] + (t & 1) * VSLOT;
        tile_bases(t);
        if (t + 1 < NT) stream(J0{}, std::true_type{}, std::true_type{}, pc, pn, t);
        else stream(JPV{}, std::false_type{}, std::false_type{}, pc, pn, t);
    };
    for (int t = 0; t < NT; t += 2) {
        tile_step(t, pbA, pbB);
        if (t + 1 < NT) tile_step(t + 1, pbB, pbA);
    }
    // ---- finalise: O^T[d][q] / l -> LDS [q][d] -> 16-byte row stores; l = O^T row 72 (the ones row of V^T): tile 2, register 4, half 0
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    __syncthreads();  // every wave has finished reading the K / V^T ring
    bf16_t* ow = reinterpret_cast<bf16_t*>(smem) + wid * 64 * OS;
    const float l0 = __shfl(a5_acc_read<O_BASE + 16 * 2 + 4>(), r), l1 = __shfl(a5_acc_read<O_BASE + 16 * 5 + 4>(), r);
    const bool bad = !(l0 < 1e30f) || !(l1 < 1e30f);   // also catches inf / NaN: the fixed reference was outgrown by about 2^100
    const float inv0 = 1.0f / l0, inv1 = 1.0f / l1;
    [&]<int... I>(std::integer_sequence<int, I...>) {
        ([&] {
            constexpr int G = I / NDT, DT = I % NDT, A0 = O_BASE + 16 * I;
            const float inv = G ? inv1 : inv0;
            const float x[16] = {a5_acc_read<A0 + 0>(), a5_acc_read<A0 + 1>(), a5_acc_read<A0 + 2>(), a5_acc_read<A0 + 3>(),
                                 a5_acc_read<A0 + 4>(), a5_acc_read<A0 + 5>(), a5_acc_read<A0 + 6>(), a5_acc_read<A0 + 7>(),
                                 a5_acc_read<A0 + 8>(), a5_acc_read<A0 + 9>(), a5_acc_read<A0 + 10>(), a5_acc_read<A0 + 11>(),
                                 a5_acc_read<A0 + 12>(), a5_acc_read<A0 + 13>(), a5_acc_read<A0 + 14>(), a5_acc_read<A0 + 15>()};
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const uint2 w = make_uint2(pack2bf(x[4 * gg] * inv, x[4 * gg + 1] * inv), pack2bf(x[4 * gg + 2] * inv, x[4 * gg + 3] * inv));
                *reinterpret_cast<uint2*>(&ow[(G * 32 + r) * OS + DT * 32 + 8 * gg + 4 * h]) = w;
            }
        }(), ...);
    }(std::make_integer_sequence<int, 2 * NDT>{});
    if (__any(bad) && lane == 0) {
        atomicOr(p.ovf_flag, 1);
        if (p.ovf_map > 0) p.ovf_flag[1 + ((long)b * p.Hh + head) * ((p.Tq + 255) >> 8) + qt] = 1;   // this workgroup's 256 queries: what the rescaling kernel recomputes
    }
    __syncthreads();
    bf16_t* op = p.o + (long)b * p.o_bs + (long)head * p.o_hs;
    for (int c = lane; c < 64 * RCH; c += 64) {
        const int row = c / RCH, ch = c - row * RCH;
        const int q = q0 + row;
        if (q < p.Tq) *reinterpret_cast<uint4*>(op + (long)q * p.o_rs + ch * 8) = *reinterpret_cast<const uint4*>(&ow[row * OS + ch * 8]);
    }
}

int ir_launch_flash_attn_pp2(const AttnParams& p, hipStream_t s) {
    if (p.D != 72 || p.Tq <= 0 || p.Tk < 64 || (p.Tk & 63) || !p.ovf_flag || p.key_bias) return -2;
    static const bool no_xcd = getenv("IR_PP2_NO_XCD_MAP") != nullptr;   // experiment knob
    // (from 32 query tiles per head on - an XCD's 32 CUs then share one head; measured 1.253 -> 1.243 ms at 16384 tokens, 0.093 -> 0.094 at 4096)
    if ((p.Hh & 7) == 0 && (p.Tq + 255) / 256 >= 32 && !no_xcd) hipLaunchKernelGGL(flash_attn_pp2_kernel, dim3(((p.Tq + 255) / 256) * p.Hh, 1, p.B), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(flash_attn_pp2_kernel, dim3((p.Tq + 255) / 256, p.Hh, p.B), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------------------------------
// DiT cross-attention (PixArt_blocks.py MultiHeadCrossAttention: 16 heads x 72, a few hundred prompt keys, an additive bias per key) as a
// PERSISTENT form of flash_attn_pp2_kernel. The 4-wave kernel spends 66 us per launch at 16384 queries, of which the keys account for 23
// (tools/xattn_pp2_probe.py: 4.6 us per 64-key tile and launch): the rest is what every workgroup pays around its few tiles - K / V^T of the
// head into LDS, Q^T into registers, the output transposition - paid serially, four rounds of workgroups deep. Here one workgroup per CU
// walks a contiguous range of (batch, head, 256-query block) items: K and V^T of a head (<= 5 tiles = 320 keys: 55 + 60 KB) stay resident in
// LDS while the head does not change, so the tile loop has no barrier, no DMA and no wait; the next block's Q rows are fetched into registers
// while the current block computes; the output leaves through a wave-private staging area.
// Same arithmetic as flash_attn_pp2_kernel (S^T = K Q^T - m via the accumulator preset, fixed reference after the first tile, P^T registers as
// the B operand of O^T = V^T P^T, denominator from the ones row of V^T, overflow -> ovf_flag -> the rescaling kernel behind). New: the additive
// key bias. K rows are 176 bytes in LDS (11 chunks: odd, conflict-free): 72 head dims + {bias_hi, bias_lo} (bias * log2 e split into two bf16,
// error 2^-17) + padding, against Q^T = 1 in dims 72 / 73 - the bias enters through the score MFMA of k-step 4, whose upper half was zero
// padding anyway. Keys beyond Tk are excluded by V^T (transpose_v_kernel: zero columns, zero ones-row entries); their K rows are zeroed here.
namespace x72 {
constexpr int D = 72, NKS = 5, NDT = 3, RCH = 9, MAXT = 5;
constexpr int KROW = 176, KSLOT = 64 * KROW;      // 11 264 B per 64-key tile
constexpr int VSLOT = 96 * 128;                   // as pp2: 96 rows x 64 keys, chunk-swizzled (rows 80.. never loaded)
constexpr int V_OFF = MAXT * KSLOT;               // 56 320
constexpr int O_OFF = V_OFF + MAXT * VSLOT;       // 117 760
constexpr int OS = 96 + 8;                        // O staging row stride (elements); 32 rows per wave: one query group at a time
constexpr int LDS_BYTES = O_OFF + 4 * 32 * OS * 2;   // 144 384 B
constexpr float MARGIN = 24.0f;
constexpr int LA = 6, NB = LA + 3;
constexpr int NQK = 20, NPV = 24, NSTEP = NQK + NPV;
constexpr int O_BASE = 0, Q_BASE = 96, QN_BASE = 136;   // AGPRs: O^T a[0:95], Q^T a[96:135], the next item's raw Q rows a[136:175]
}  // namespace x72
__device__ __attribute__((aligned(16))) const uint32_t x72_bias_ones[4] = {0x3f803f80u, 0u, 0u, 0u};   // bf16 {1, 1, 0 x 6}: Q^T dims 72..79

__global__ __launch_bounds__(256, 1) void flash_attn_x72_kernel(AttnParams p, int n_items, int per_wg) {
    using namespace x72;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int NT = (p.Tk + 63) >> 6, QT = (p.Tq + 255) >> 8;
    const int it0 = blockIdx.x * per_wg, it1 = min(it0 + per_wg, n_items);
    if (it0 >= it1) return;

    const uint32_t lds0 = lds_addr(smem);
    const uint32_t k_addr = lds0 + a5_swap23(r) * KROW + h * 16;           // + tile*KSLOT + kt*32*KROW + ks*32
    const int vsw = (r >> 1) & 7;
    uint32_t v_addr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v_addr[j] = lds0 + V_OFF + r * 128 + ((((2 * j) | h) ^ vsw) << 4);   // + tile*VSLOT + dt*4096
    bf16_t* ow = reinterpret_cast<bf16_t*>(smem + O_OFF) + wid * 32 * OS;

    // item -> (batch, head, query block); Q rows of a block: lane (r, h) holds dims 16 ks + 8 h .. + 7 of query q0 + 32 g + r, (g, ks) = I / 5, I % 5
    auto item = [&](int it, int& b, int& head, int& qt) { qt = it % QT; const int hb = it / QT; head = hb % p.Hh; b = hb / p.Hh; };
    // The next item's Q rows wait in AGPRs a[136:175] (global loads may target the accumulation file directly): 40 arch VGPRs held through the
    // whole tile loop do not fit beside the stream's own (the arch file ends at v255 whatever the AGPR count).
    auto q_fetch = [&](int it) {
        int b, head, qt;
        item(it, b, head, qt);
        const bf16_t* qp = p.q + (long)b * p.q_bs + (long)head * p.q_hs;   // wave-uniform: a scalar base
        const int q0 = qt * 256 + wid * 64;
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                constexpr int g = I / NKS, ks = I % NKS;
                const uint32_t row = (uint32_t)min(q0 + g * 32 + r, p.Tq - 1) * (uint32_t)p.q_rs;
                const bf16_t* base = qp;   // (a captured variable is not accepted as an asm operand inside a generic lambda)
                if constexpr (ks < NKS - 1) {
                    const uint32_t off = (row + (uint32_t)(ks * 16 + h * 8)) * 2u;
                    asm volatile("global_load_dwordx4 a[%c0:%c1], %2, %3" ::"n"(QN_BASE + 4 * I), "n"(QN_BASE + 4 * I + 3), "v"(off), "s"(base) : "memory");
                } else {   // k-step 4: dims 64..71 of the row (h = 0) | {1, 1, 0 x 6} against the bias pair in K dims 72 / 73 (h = 1)
                    const void* a4 = h ? static_cast<const void*>(x72_bias_ones) : static_cast<const void*>(base + row + 64);
                    asm volatile("global_load_dwordx4 a[%c0:%c1], %2, off" ::"n"(QN_BASE + 4 * I), "n"(QN_BASE + 4 * I + 3), "v"(a4) : "memory");
                }
            }(), ...);
        }(std::make_integer_sequence<int, 2 * NKS>{});
    };
    asm volatile(".set ir_x72_i, 0\n\t.rept 96\n\tv_accvgpr_write_b32 a[ir_x72_i], 0\n\t.set ir_x72_i, ir_x72_i + 1\n\t.endr" ::: IR_AGPR176_CLOBBERS);

    q_fetch(it0);
    wait_dma();   // (vmcnt(0): the first item's rows; later items' rows are waited for before the previous item's output stores)
    int cur_b = -1, cur_head = -1;
    for (int it = it0; it < it1; ++it) {
        int b, head, qt;
        item(it, b, head, qt);
        const int q0 = qt * 256 + wid * 64;
        // ---- K / V^T of the head -> LDS (when the head, or a batch with its own K / V, changes)
        if (head != cur_head || (b != cur_b && (p.k_bs != 0 || p.vt_bs != 0 || p.kb_bs != 0))) {
            __syncthreads();   // every wave is done with the previous head's tiles
            const bf16_t* kp = p.k + (long)b * p.k_bs + (long)head * p.k_hs;
            const bf16_t* vtp = p.vt + (long)b * p.vt_bs + (long)head * 96 * p.Tk_pad;
            const float* kb = p.key_bias ? p.key_bias + (long)b * p.kb_bs : nullptr;
            // every load of a thread is issued before its first LDS store (fixed trip counts, unrolled: 13 + 13 x 16 bytes in flight per thread):
            // a load -> store loop costs one memory latency per trip, 25 trips per head
            constexpr int KTRIP = (MAXT * 64 * 10 + 255) / 256, VTRIP = (MAXT * 80 * 8 + 255) / 256;
            int tid_o = tid;
            asm volatile("" : "+v"(tid_o));   // opaque: left transparent, hipcc hoists the 26 trips' index arithmetic out of the item loop and keeps
                                               // ~100 values alive through the tile streams (AGPR copies, scratch, v_writelane'd scalars)
            uint4 kx[KTRIP];
#pragma unroll
            for (int i = 0; i < KTRIP; ++i) {     // K rows: 9 chunks of the cache row + the bias chunk
                const int c = tid_o + 256 * i, key = c / 10, ch = c - key * 10;
                kx[i] = make_uint4(0, 0, 0, 0);
                if (c < NT * 640 && key < p.Tk) {
                    if (ch < RCH) {   // softmax scale * log2 e folded into K (once per head and workgroup) instead of into every block's Q rows
                        const uint4 x = *reinterpret_cast<const uint4*>(kp + (long)key * p.k_rs + ch * 8);
                        const float sc = p.scale_log2;
                        kx[i] = make_uint4(pack2bf(bflo(x.x) * sc, bfhi(x.x) * sc), pack2bf(bflo(x.y) * sc, bfhi(x.y) * sc),
                                           pack2bf(bflo(x.z) * sc, bfhi(x.z) * sc), pack2bf(bflo(x.w) * sc, bfhi(x.w) * sc));
                    } else if (kb) {
                        const float v = kb[key] * 1.44269504088896340736f;
                        const uint32_t hi = pack2bf(v, 0.f) & 0xffffu;
                        const uint32_t lo = pack2bf(v - bflo(hi), 0.f) & 0xffffu;
                        kx[i].x = hi | (lo << 16);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < KTRIP; ++i) {
                const int c = tid_o + 256 * i, key = c / 10, ch = c - key * 10;
                if (c < NT * 640) *reinterpret_cast<uint4*>(smem + (key >> 6) * KSLOT + (key & 63) * KROW + ch * 16) = kx[i];
            }
            __builtin_amdgcn_sched_barrier(0);   // the K batch's registers are free before the V^T batch takes its own
            uint4 vx[VTRIP];
#pragma unroll
            for (int i = 0; i < VTRIP; ++i) {     // V^T rows 0..79 (72 dims, the ones row, zero rows), 8 chunks of 8 keys, chunk-swizzled
                const int c = tid_o + 256 * i, t = c / 640, rem = c - t * 640, d = rem >> 3, pos = rem & 7;
                vx[i] = make_uint4(0, 0, 0, 0);
                if (c < NT * 640) vx[i] = *reinterpret_cast<const uint4*>(vtp + (long)d * p.Tk_pad + t * 64 + ((pos ^ ((d >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int i = 0; i < VTRIP; ++i) {
                const int c = tid_o + 256 * i, t = c / 640, rem = c - t * 640, d = rem >> 3, pos = rem & 7;
                if (c < NT * 640) *reinterpret_cast<uint4*>(smem + V_OFF + t * VSLOT + d * 128 + pos * 16) = vx[i];
            }
            if (cur_head < 0)   // rows 80..95 of every tile: multiplied (third O^T tile) but never loaded - zeros toggle nothing (see flash_attn_pp2_kernel)
                for (int c = tid_o; c < MAXT * 16 * 8; c += 256)
                    *reinterpret_cast<uint4*>(smem + V_OFF + (c >> 7) * VSLOT + (80 + ((c & 127) >> 3)) * 128 + (c & 7) * 16) = make_uint4(0, 0, 0, 0);
            cur_b = b; cur_head = head;
            __syncthreads();
        }

        // ---- Q^T fragments: the staged rows ARE the B operands (bf16 rows, the scale lives in K): 40 accumulator-file moves, then the next item's rows on their way
        asm volatile(".set ir_x72_i, 0\n\t.rept 40\n\tv_accvgpr_mov_b32 a[96 + ir_x72_i], a[136 + ir_x72_i]\n\t.set ir_x72_i, ir_x72_i + 1\n\t.endr" ::: "memory");
        asm volatile("s_nop 1" ::: "memory");
        if (it + 1 < it1) q_fetch(it + 1);

        f32x16 sacc[2][2], negm[2];
        uint4 pbA[2][4], pbB[2][4];
        bf16x8 fr[NB];
        float p_hold2[2] = {0.f, 0.f}, pend0[2] = {0.f, 0.f}, pend1[2] = {0.f, 0.f};
        uint32_t ka = k_addr, va[4] = {v_addr[0], v_addr[1], v_addr[2], v_addr[3]};

        auto frag_read = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j < NQK) fr[j % NB] = lds_read16<(((j % 10) / 5) * 32 * KROW + (j % 5) * 32)>(ka);
            else fr[(NQK + ((j - NQK) >> 1)) % NB] = lds_read16<(((j - NQK) >> 3) * 4096)>(va[((j - NQK) >> 1) & 3]);
        };
        // one pinned stream per tile pair, as in flash_attn_pp2_kernel without its LDS-DMA pieces: [0, 20) S^T MFMAs of tile t+1, [20, 44) PV MFMAs of tile t
        auto stream = [&](auto j0c, auto j1c, auto smc, uint4 (&pc)[2][4], uint4 (&pn)[2][4]) {
            constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;   // steps [J0, J1)
            constexpr bool SOFTMAX = decltype(smc)::value;
            constexpr int S0 = J0 < NQK ? J0 : NQK + ((J0 - NQK) >> 1), S1 = J1 <= NQK ? J1 : NQK + ((J1 - NQK) >> 1);
            auto slot_read = [&](auto sc) {
                constexpr int sl = decltype(sc)::value;
                if constexpr (sl < NQK) frag_read(std::integral_constant<int, sl>{});
                else frag_read(std::integral_constant<int, NQK + 2 * (sl - NQK)>{});
            };
            auto step = [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int sl = j < NQK ? j : NQK + ((j - NQK) >> 1);
                constexpr bool first_use = j < NQK || ((j - NQK) & 1) == 0;
                if constexpr (first_use) {
                    if constexpr (sl + LA < S1) slot_read(std::integral_constant<int, sl + LA>{});
                    constexpr int rem = S1 - 1 - sl;
                    if constexpr (((sl - S0) & 1) == 0) wait_lds<(rem < LA ? (rem > 0 ? rem - 1 : 0) : LA - 1)>();
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (j < NQK) {
                    constexpr int g = j / 10, kt = (j % 10) / 5, ks = j % 5;
                    if constexpr (ks == 0) pp2_mfma_qk_first<Q_BASE + 4 * (5 * g + ks)>(sacc[g][kt], fr[sl % NB], negm[g]);
                    else pp2_mfma_qk<Q_BASE + 4 * (5 * g + ks)>(sacc[g][kt], fr[sl % NB]);
                } else {
                    constexpr int q = (j - NQK) >> 1, g = j & 1, dt = q >> 2, kk = q & 3;
                    pp2_mfma_pv<O_BASE + 16 * (3 * g + dt)>(fr[sl % NB], __builtin_bit_cast(bf16x8, pc[g][kk]));
                }
                if constexpr (sl - 2 >= S0) asm volatile("" ::"v"(fr[(sl - 2) % NB]));
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (SOFTMAX && j >= 13) {
                    constexpr int n0 = j < 23 ? 2 * (j - 13) : 20 + (j - 23) * 44 / 21, n1 = j < 23 ? n0 + 2 : 20 + (j - 22) * 44 / 21;
                    [&]<int... E>(std::integer_sequence<int, E...>) {
                        ([&] {
                            constexpr int n = n0 + E;
                            if constexpr (n < n1 && n < 64) {
                                constexpr int g = pp2_item_g(n), e = pp2_item_e(n);
                                const float pv = __builtin_amdgcn_exp2f(sacc[g][e >> 4][e & 15]);
                                if constexpr (e & 1) {
                                    if constexpr (((n >> 1) & 1) == 0) { pend0[0] = p_hold2[g]; pend0[1] = pv; } else { pend1[0] = p_hold2[g]; pend1[1] = pv; }
                                    if constexpr (n >= 3) {
                                        constexpr int gp = pp2_item_g(n - 2), ep = pp2_item_e(n - 2);
                                        if constexpr (((n >> 1) & 1) == 0) a5_set_word<((ep & 7) >> 1)>(pn[gp][ep >> 3], pack2bf_valu(pend1[0], pend1[1]));
                                        else a5_set_word<((ep & 7) >> 1)>(pn[gp][ep >> 3], pack2bf_valu(pend0[0], pend0[1]));
                                    }
                                } else {
                                    p_hold2[g] = pv;
                                }
                            }
                        }(), ...);
                    }(std::make_integer_sequence<int, 4>{});
                    if constexpr (j == NSTEP - 1) {
                        constexpr int gp = pp2_item_g(63), ep = pp2_item_e(63);
                        a5_set_word<((ep & 7) >> 1)>(pn[gp][ep >> 3], pack2bf_valu(pend1[0], pend1[1]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            __builtin_amdgcn_sched_barrier(0);
            [&]<int... I>(std::integer_sequence<int, I...>) { (slot_read(std::integral_constant<int, S0 + I>{}), ...); }(std::make_integer_sequence<int, LA>{});
            __builtin_amdgcn_sched_barrier(0);
            [&]<int... I>(std::integer_sequence<int, I...>) { (step(std::integral_constant<int, J0 + I>{}), ...); }(std::make_integer_sequence<int, J1 - J0>{});
        };
        using J0 = std::integral_constant<int, 0>;
        using JPV = std::integral_constant<int, NQK>;
        using JEND = std::integral_constant<int, NSTEP>;

        // ---- the reference. First attempt: the maximum of tile 0 + 2^24 headroom, as in flash_attn_pp2_kernel. If a later key outgrows that by about
        // 2^100 (a denominator that is not a moderate finite number), the wave repeats ITS item with the exact row maxima, found by a scores-only
        // pass over the resident tiles - no flag, no second kernel behind this one (4.9 us per launch even when it returns at once).
        bool retry = false;
        float l0, l1;
        for (;;) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int e = 0; e < 16; ++e) negm[g][e] = 0.f;
        // The zeros are materialised HERE, a few wait states ahead of the MFMA that reads them as its C operand: the MFMAs are asm statements, so hipcc's
        // hazard recogniser does not see a VALU write -> MFMA SrcC read and may place the v_mov's directly in front of the MFMA. Without this, second
        // and later items of a workgroup came out wrong in query group 0, differently from run to run (first items have a barrier in between).
        asm volatile("s_nop 7" : "+v"(negm[0]), "+v"(negm[1]));
        float mxr[2] = {-INFINITY, -INFINITY};
        if (retry) {
            for (int t = 0; t < NT; ++t) {
                ka = k_addr + t * KSLOT;
                stream(J0{}, JPV{}, std::false_type{}, pbA, pbB);
                asm volatile("s_nop 15\n\ts_nop 7" : "+v"(sacc[0][0]), "+v"(sacc[0][1]), "+v"(sacc[1][0]), "+v"(sacc[1][1]));
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) mxr[g] = fmaxf(mxr[g], sacc[g][kt][e]);
            }
        }
        // ---- tile 0: scores of both groups (the stream's first 20 steps: fragment reads LA ahead), softmax in the open
        ka = k_addr;
        stream(J0{}, JPV{}, std::false_type{}, pbA, pbB);
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(sacc[0][0]), "+v"(sacc[0][1]), "+v"(sacc[1][0]), "+v"(sacc[1][1]));
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[g][kt][e]);
            const float m_ref = retry ? xhalf_max(mxr[g]) : xhalf_max(mx) + MARGIN;
#pragma unroll
            for (int e = 0; e < 16; ++e) negm[g][e] = -m_ref;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float pv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[e] = __builtin_amdgcn_exp2f(sacc[g][kk >> 1][(kk & 1) * 8 + e] - m_ref);
                pbA[g][kk] = make_uint4(pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3]), pack2bf(pv[4], pv[5]), pack2bf(pv[6], pv[7]));
                pbB[g][kk] = make_uint4(0, 0, 0, 0);
            }
        }
        // ---- tiles: no barrier, no DMA - K / V^T are resident
        auto tile_step = [&](int t, uint4 (&pc)[2][4], uint4 (&pn)[2][4]) {
            ka = k_addr + (t + 1) * KSLOT;
#pragma unroll
            for (int j = 0; j < 4; ++j) va[j] = v_addr[j] + t * VSLOT;
            if (t + 1 < NT) stream(J0{}, JEND{}, std::true_type{}, pc, pn);
            else stream(JPV{}, JEND{}, std::false_type{}, pc, pn);
        };
        for (int t = 0; t < NT; t += 2) {
            tile_step(t, pbA, pbB);
            if (t + 1 < NT) tile_step(t + 1, pbB, pbA);
        }
        // ---- finalise: O^T[d][q] / l -> wave-private staging [q][d] -> 16-byte row stores, one query group at a time; then O^T = 0 again
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        l0 = __shfl(a5_acc_read<O_BASE + 16 * 2 + 4>(), r);
        l1 = __shfl(a5_acc_read<O_BASE + 16 * 5 + 4>(), r);
        if (retry || !__any(!(l0 < 1e30f) || !(l1 < 1e30f))) break;
        retry = true;
        asm volatile(".set ir_x72_i, 0\n\t.rept 96\n\tv_accvgpr_write_b32 a[ir_x72_i], 0\n\t.set ir_x72_i, ir_x72_i + 1\n\t.endr" ::: "memory");
        }
        const float inv0 = 1.0f / l0, inv1 = 1.0f / l1;
        bf16_t* op = p.o + (long)b * p.o_bs + (long)head * p.o_hs;
        wait_dma();   // the next item's Q rows have landed (issued a whole tile loop ago); nothing else is outstanding, and the stores below are not waited for
        [&]<int... G>(std::integer_sequence<int, G...>) {
            ([&] {
                constexpr int GG = G;   // (a pack of the outer fold must not appear inside the inner one)
                const float inv = GG ? inv1 : inv0;
                [&]<int... DT>(std::integer_sequence<int, DT...>) {
                    ([&] {
                        constexpr int A0 = O_BASE + 16 * (NDT * GG + DT);
                        const float x[16] = {a5_acc_read<A0 + 0>(), a5_acc_read<A0 + 1>(), a5_acc_read<A0 + 2>(), a5_acc_read<A0 + 3>(),
                                             a5_acc_read<A0 + 4>(), a5_acc_read<A0 + 5>(), a5_acc_read<A0 + 6>(), a5_acc_read<A0 + 7>(),
                                             a5_acc_read<A0 + 8>(), a5_acc_read<A0 + 9>(), a5_acc_read<A0 + 10>(), a5_acc_read<A0 + 11>(),
                                             a5_acc_read<A0 + 12>(), a5_acc_read<A0 + 13>(), a5_acc_read<A0 + 14>(), a5_acc_read<A0 + 15>()};
#pragma unroll
                        for (int gg = 0; gg < 4; ++gg) {
                            const uint2 w = make_uint2(pack2bf(x[4 * gg] * inv, x[4 * gg + 1] * inv), pack2bf(x[4 * gg + 2] * inv, x[4 * gg + 3] * inv));
                            *reinterpret_cast<uint2*>(&ow[r * OS + DT * 32 + 8 * gg + 4 * h]) = w;
                        }
                    }(), ...);
                }(std::make_integer_sequence<int, NDT>{});
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (int c = lane; c < 32 * RCH; c += 64) {
                    const int row = c / RCH, ch = c - row * RCH;
                    const int q = q0 + GG * 32 + row;
                    const uint4 v = *reinterpret_cast<const uint4*>(&ow[row * OS + ch * 8]);
                    if (q < p.Tq) *reinterpret_cast<uint4*>(op + (long)q * p.o_rs + ch * 8) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the rows are in registers before the other group overwrites the staging area
                __builtin_amdgcn_wave_barrier();
            }(), ...);
        }(std::integer_sequence<int, 0, 1>{});
        asm volatile(".set ir_x72_i, 0\n\t.rept 96\n\tv_accvgpr_write_b32 a[ir_x72_i], 0\n\t.set ir_x72_i, ir_x72_i + 1\n\t.endr" ::: IR_AGPR176_CLOBBERS);
    }
}

bool ir_flash_attn_x72_takes(const AttnParams& p) {
    static const bool off = getenv("IR_NO_X72") != nullptr;   // experiment knob: the 4-wave kernel for the cross-attention again
    return !off && !g_ir_plain_kernels && p.D == 72 && p.Tk > 0 && p.Tk <= 64 * x72::MAXT && p.Tk_pad >= ((p.Tk + 63) & ~63) && p.Tq >= 256 &&
           (long)p.B * p.Hh * ((p.Tq + 255) / 256) >= 64;
}
int ir_launch_flash_attn_x72(const AttnParams& p, hipStream_t s) {
    if (!ir_flash_attn_x72_takes(p)) return -2;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int n_items = p.B * p.Hh * ((p.Tq + 255) / 256);
    const int per_wg = (n_items + cus - 1) / cus, grid = (n_items + per_wg - 1) / per_wg;
    hipLaunchKernelGGL(flash_attn_x72_kernel, dim3(grid), dim3(256), 0, s, p, n_items, per_wg);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
