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
constexpr int LA = 4, NB = LA + 3;       // fragment reads in flight ahead of their MFMA; fragment register sets (see flash_attn_pp_kernel)
constexpr int SM0 = 35;                  // first stream step that carries a softmax slice: three PV MFMAs behind the last QK^T MFMA
}  // namespace a5

// V [B][T][512] (token stride rs) -> V^T tiles [B][T/32][512][32] bf16; chunk c (keys 8c .. 8c+7) of row d at 16-byte slot c ^ ((d >> 2) & 3)
__global__ __launch_bounds__(256) void transpose_v_tiles_kernel(const bf16_t* __restrict__ v, bf16_t* __restrict__ vt, int rs, long v_bs,
                                                                long vt_bs) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[32][512 + 8];
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
                                                                    int* __restrict__ ovf_flag) {
    using namespace a5;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
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
    auto k_piece = [&](int tile, int i, int slot) {
        a5_glds16(k_lane + tile * k_tile + i * k_step, (a5_lds_t)(smem + slot * KSLOT + (wu + 4 * i) * KROW));
    };
    auto v_piece = [&](int tile, int i, int slot) {
        a5_glds16(v_lane + (long)tile * (512 * 32) + i * 2048, (a5_lds_t)(smem + V_OFF + slot * VSLOT + (wu + 4 * i) * 1024));
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
            wait_lds<(J1 - 1 - j < LA ? J1 - 1 - j : LA)>();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(sacc) : "v"(fr[j % NB]), "v"(qf[0]));
            else if constexpr (j < 32) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sacc) : "v"(fr[j % NB]), "v"(qf[j]));
            else a5_mfma_pv<((j - 32) >> 1)>(fr[j % NB], __builtin_bit_cast(bf16x8, pc[j & 1]));
            if constexpr (j - 2 >= J0) asm volatile("" ::"v"(fr[(j - 2) % NB]));  // keep the fragment of MFMA j-2 allocated until here
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DMA && j < 32 && (j & 1)) {  // the next tiles' pieces, one behind every other QK^T MFMA
                constexpr int pi = j >> 1;
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
                if constexpr (e & 1) a5_set_word<((e & 7) >> 1)>(pn[e >> 3], pack2bf(p_hold, p));
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
        if (t + 1 < NT) stream(I0{}, I64{}, std::true_type{}, std::true_type{}, pc, pn, t, t & 1, (t + 1) & 1);
        else stream(I32{}, I64{}, std::false_type{}, std::false_type{}, pc, pn, t, 0, 0);
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

int ir_launch_transpose_v_tiles(const bf16_t* v, bf16_t* vt, int B, int T, int rs, long v_bs, long vt_bs, hipStream_t s) {
    if (T <= 0 || (T & 31) || (rs & 7) || rs < 512 || B <= 0) return -2;
    hipLaunchKernelGGL(transpose_v_tiles_kernel, dim3(T / 32, B), dim3(256), 0, s, v, vt, rs, v_bs, vt_bs);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int ir_launch_flash_attn_d512_v2(const bf16_t* q, const bf16_t* k, const bf16_t* vt_tiles, bf16_t* o, int B, int T, int rs, int o_rs, long qk_bs,
                                 long vt_bs, long o_bs, float scale, int* ovf_flag, hipStream_t s) {
    if (T <= 0 || (T & 31) || (rs & 7) || (o_rs & 7) || rs < 512 || o_rs < 512 || B <= 0 || !ovf_flag) return -2;
    hipLaunchKernelGGL(flash_attn_d512_v2_kernel, dim3((T + 127) / 128, B), dim3(256), 0, s, q, k, vt_tiles, o, T, rs, o_rs, qk_bs, vt_bs, o_bs,
                       scale * 1.44269504088896340736f, ovf_flag);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
