// EXPERIMENT, not built into the library (round 2): measured against gemm_pp_kernel on the same box it is SLOWER on every DiT shape but
// K = 4608 (16384 x 1152 -> 1152: 64 us vs 49 us; -> 3456: 134 vs 124; -> 4608: 198 vs 171; 4608 -> 1152: 155 vs 161): a 256 x 288 tile moves
// 34 KB per k-tile and CU, so the loop is bound by what LDS-DMA has in flight (three stages = 100 KB per CU), not by barriers, and with one
// wave per SIMD the 136 KB prologue and the 128 x 144 epilogue of every tile are exposed. Kept for the record; correct (passed the
// linear tests of tests/test_ops_gpu.py when routed in). To try it: copy to instarevive_amd/csrc, add to build.py, route in ir_launch_igemm.
// gemm_s1_kernel: the big-tile GEMM of the DiT linears (out = epilogue(in [M][K] x wgt [N][K]^T), PixArt_blocks.py:123-158, PixArtMS.py:67-77)
// in the one-wave-per-SIMD structure of conv_halo_s1_kernel (conv_s1.hip) instead of the two-waves-per-SIMD ping-pong of gemm_pp_kernel
// (igemm.hip), whose two workgroup barriers per k-tile leave the matrix pipe idle half of the time (SQ_VALU_MFMA_BUSY 0.48).
//
// Same 256 x 288 workgroup tile (16384 tokens x 1152 / 3456 / 4608 columns = exactly 1 / 3 / 4 rounds of 256 workgroups) on 4 waves, one
// per SIMD: wave (wm, wn) owns rows 128 wm .. +127 and columns 144 wn .. +143 = 8 x 9 tiles of v_mfma_f32_16x16x32_bf16 = 288
// accumulators: 64 tiles in the 256 AGPRs (named literally from inline asm), the last column fragment's 8 tiles in VGPRs.
// A = weights (rows m of the MFMA = output columns), B = tokens: a lane ends up with 4 consecutive output columns of one token, so the
// epilogue writes float4s into its fp32 slab.
// k-tiles are 32 wide: 16 KB of tokens + 18 KB of weights = 34 one-KB LDS-DMA pieces (16 rows x 64 B; chunk c of row r at slot
// c ^ 2 ((r >> 2) & 1): conflict-free ds_read_b128) in a ring of FOUR 34 KB stages. Per k-tile every wave issues one pinned stream of 72
// MFMAs (1152 matrix cycles) with the 17 fragment reads of the NEXT k-tile and its 9 DMA pieces of k-tile s + 4 (ring slot s & 3: its
// fragments are in registers already) in the gaps; one workgroup barrier per k-tile, before it a counted vmcnt(18) = everything but the
// pieces of the last two k-tiles has landed (k-tile s + 2, read during s + 1). Past the end the last k-tile is re-read into a free slot.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <utility>
#include "agpr256.h"
#include "common.h"
#include "kernels.h"

namespace gs1 {
constexpr int BM = 256, BN = 288, BK = 32;
constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;   // 16384 + 18432 = 34816
constexpr int NS = 4;
constexpr int LDS_MAIN = NS * STAGE;           // 139 264
constexpr int NPIECE = STAGE / 1024;           // 34
constexpr int PW = 9;                          // pieces per wave and k-tile (piece q = wave + 4 i, clamped to the last)
constexpr int SROW = 148;                      // slab row stride in floats (144 + 4)
constexpr int SLAB = 32 * SROW * 4;            // 18 944 B per wave: two 16-token fragments x 144 columns
constexpr int LDS_EP = 4 * SLAB;
constexpr int LDS_BYTES = LDS_MAIN > LDS_EP ? LDS_MAIN : LDS_EP;
constexpr int hkey(int r) { return ((r >> 2) & 1) << 1; }
}  // namespace gs1

typedef __attribute__((address_space(3))) void* gs1_lds_t;
typedef __attribute__((ext_vector_type(4))) float gs1_f4;
IR_DEVINL void gs1_glds16(const void* g, gs1_lds_t l) { __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0); }
template <int LO>
IR_DEVINL void gs1_mfma_a(bf16x8 w, bf16x8 x) {   // accumulator tile in a[LO : LO+3]
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(x), "n"(LO), "n"(LO + 3));
}
IR_DEVINL void gs1_mfma_v(gs1_f4& c, bf16x8 w, bf16x8 x) {   // accumulator tile in VGPRs
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(x));
}
template <int I>
IR_DEVINL float gs1_acc_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "n"(I));
    return x;
}
template <int ACT>
IR_DEVINL float gs1_act(float x, float slope) {
    if (ACT == IR_ACT_GELU_ERF) return gelu_erf(x);
    if (ACT == IR_ACT_GELU_TANH) return gelu_tanh(x);
    if (ACT == IR_ACT_LRELU) return x > 0.f ? x : x * slope;
    if (ACT == IR_ACT_SILU) return silu(x);
    return x;
}

__global__ __launch_bounds__(256, 1) void gemm_s1_kernel(IGemmParams p) {
    using namespace gs1;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];   // 4 stages of {tokens 256 x 64 B | weights 288 x 64 B}; epilogue: slabs
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int wm = wu >> 1, wn = wu & 1;
    const int c16 = lane & 15, kq = lane >> 4;
    const int NT = p.Cout_pad / BN, MT = (p.M + BM - 1) / BM;
    const int bid = blockIdx.x, xcd = bid & 7, jb = bid >> 3;
    const int mt = (jb / NT) * 8 + xcd, nt = jb % NT;   // an XCD runs the column tiles of one row tile back to back (tokens re-read from its L2)
    if (mt >= MT) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int KT = p.Cin / BK;   // even, >= 8 (launcher)

    asm volatile(".set ir_gs1_i, 0\n\t.rept 256\n\tv_accvgpr_write_b32 a[ir_gs1_i], 0\n\t.set ir_gs1_i, ir_gs1_i + 1\n\t.endr" ::: IR_AGPR256_CLOBBERS);
    gs1_f4 accv[8];   // tiles (i, 8): the ninth column fragment
#pragma unroll
    for (int i = 0; i < 8; ++i) accv[i] = gs1_f4{0.f, 0.f, 0.f, 0.f};

    // ---- LDS-DMA sources: piece q covers stage rows 16 q .. 16 q + 15 (q < 16: token rows; else weight rows 16 (q - 16) ..); lane l ->
    // row l >> 2, LDS slot l & 3 <- global chunk (l & 3) ^ hkey(row). Rows beyond M re-read row M - 1 (never stored).
    const bf16_t* src[PW];
    int dstq[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int q = min(wu + 4 * i, NPIECE - 1);
        const int r = lane >> 2;
        const int sw = ((lane & 3) ^ hkey(r)) << 3;
        if (q < 16) src[i] = p.in + (long)min(m0 + q * 16 + r, p.M - 1) * p.in_cs + sw;
        else src[i] = p.wgt + (long)(n0 + (q - 16) * 16 + r) * p.wgt_rs + sw;
        dstq[i] = q * 1024;
    }
    auto issue = [&](auto ic, int kt, int slot) __attribute__((always_inline)) {   // this wave's piece i of k-tile kt into ring slot `slot`
        constexpr int i = decltype(ic)::value;
        gs1_glds16(src[i] + kt * BK, (gs1_lds_t)(smem + slot * STAGE + dstq[i]));
    };
    // ---- fragment read addresses: token fragment i = rows 128 wm + 16 i + c16, weight fragment j = rows 144 wn + 16 j + c16, chunk kq
    const uint32_t lds0 = lds_addr(smem);
    const uint32_t fsw = (uint32_t)((kq ^ hkey(c16)) << 4);
    const uint32_t xrd = lds0 + (128 * wm + c16) * 64 + fsw;              // + slot * STAGE + i * 1024
    const uint32_t wrd = lds0 + A_BYTES + (144 * wn + c16) * 64 + fsw;    // + slot * STAGE + j * 1024
    bf16x8 fx[2][8], fw[2][9];

    // ---- prologue: k-tiles 0..3, fragments of k-tile 0
#pragma unroll
    for (int kt = 0; kt < NS; ++kt)
        [&]<int... I>(std::integer_sequence<int, I...>) { (issue(std::integral_constant<int, I>{}, kt, kt), ...); }(std::make_integer_sequence<int, PW>{});
    wait_dma();
    __syncthreads();
    [&]<int... R>(std::integer_sequence<int, R...>) {
        ([&] {
            if constexpr (R < 9) fw[0][R] = lds_read16<R * 1024>(wrd);
            else fx[0][R - 9] = lds_read16<(R - 9) * 1024>(xrd);
        }(), ...);
    }(std::make_integer_sequence<int, 17>{});
    wait_lds<0>();

    auto step = [&](auto setc, int s) __attribute__((always_inline)) {
        constexpr int SET = decltype(setc)::value;
        const uint32_t so = (uint32_t)((s + 1) & 3) * STAGE;
        const uint32_t xa = xrd + so, wa = wrd + so;
        const int kt4 = min(s + 4, KT - 1);   // past the end: the last k-tile again, into the free slot
        const int slot4 = s & 3;
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                constexpr int TI = I / 9, TJ = I % 9;   // token fragment, column fragment
                if constexpr ((I & 3) == 0 && (I >> 2) < 17) {   // one fragment of the next k-tile per four MFMAs, into the other set
                    constexpr int R = I >> 2;
                    if constexpr (R < 9) fw[SET ^ 1][R] = lds_read16<R * 1024>(wa);
                    else fx[SET ^ 1][R - 9] = lds_read16<(R - 9) * 1024>(xa);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (TJ < 8) gs1_mfma_a<4 * (TI * 8 + TJ)>(fw[SET][TJ], fx[SET][TI]);
                else gs1_mfma_v(accv[TI], fw[SET][TJ], fx[SET][TI]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (I % 7 == 2 && I / 7 < PW) {   // the 9 pieces of k-tile s + 4, one per seven MFMAs
                    issue(std::integral_constant<int, I / 7>{}, kt4, slot4);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }(), ...);
        }(std::make_integer_sequence<int, 72>{});
        wait_lds<0>();
        wait_vm<2 * PW>();   // everything but the pieces of k-tiles s + 3 and s + 4: k-tile s + 2 has landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < KT; s += 2) {
        step(std::integral_constant<int, 0>{}, s);
        step(std::integral_constant<int, 1>{}, s + 1);
    }

    // ---- epilogue: v = act(acc + bias) * out_scale * gate + res, 32 tokens x 144 columns per wave at a time through an fp32 slab
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(accv[0]), "+v"(accv[1]), "+v"(accv[2]), "+v"(accv[3]), "+v"(accv[4]), "+v"(accv[5]), "+v"(accv[6]), "+v"(accv[7])
                 :: "memory");   // the last MFMA results -> v_accvgpr_read / VALU
    wait_dma();      // the re-read pieces past the end must have landed before the slabs overlay the ring
    __syncthreads();
    float* slab = reinterpret_cast<float*>(smem + wid * SLAB);
    const int nw = n0 + wn * 144;          // first column of this wave
    const int mw = m0 + wm * 128;          // first token of this wave
    gs1_f4 cb[9], cm[9];                   // bias and out_scale * gate of this lane's 4 columns of every column fragment
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int n = nw + 16 * j + 4 * kq;
        cb[j] = p.bias ? *reinterpret_cast<const gs1_f4*>(p.bias + n) : gs1_f4{0.f, 0.f, 0.f, 0.f};
        cm[j] = (p.gate ? *reinterpret_cast<const gs1_f4*>(p.gate + n) : gs1_f4{1.f, 1.f, 1.f, 1.f}) * p.out_scale;
    }
    auto write_pass = [&](auto hc, auto act_tag) __attribute__((always_inline)) {   // token fragments 2 hh, 2 hh + 1 -> slab rows 0..31
        constexpr int HH = decltype(hc)::value, ACT = decltype(act_tag)::value;
        [&]<int... J>(std::integer_sequence<int, J...>) {
            ([&] {
                constexpr int II = J / 9, TJ = J % 9, TI = 2 * HH + II;
                gs1_f4 v;
                if constexpr (TJ < 8) {
                    constexpr int LO = 4 * (TI * 8 + TJ);
                    v = gs1_f4{gs1_acc_read<LO>(), gs1_acc_read<LO + 1>(), gs1_acc_read<LO + 2>(), gs1_acc_read<LO + 3>()};
                } else {
                    v = accv[TI];
                }
                v += cb[TJ];
                if constexpr (ACT != IR_ACT_NONE) v = gs1_f4{gs1_act<ACT>(v[0], p.slope), gs1_act<ACT>(v[1], p.slope), gs1_act<ACT>(v[2], p.slope), gs1_act<ACT>(v[3], p.slope)};
                v *= cm[TJ];
                *reinterpret_cast<gs1_f4*>(&slab[(II * 16 + c16) * SROW + 16 * TJ + 4 * kq]) = v;
            }(), ...);
        }(std::make_integer_sequence<int, 18>{});
    };
    auto wave_sync = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // Row phase of a pass. The forms the DiT uses are specialised so that no uniform condition sits inside the unrolled loops: KIND 0 =
    // no residual, bf16 out; KIND 1 = fp32 residual, fp32 out (+ optional bf16 copy); KIND 2 = anything else.
    const int kind = (!p.res && !p.out_f32 && !p.out2) ? 0 : (p.res && p.res_f32 && p.out_f32 && p.res_mod == 0) ? 1 : 2;
    auto rows = [&](int hh) __attribute__((always_inline)) {
        const int mh = mw + hh * 32;
        if (kind == 0) {
            bf16_t* outb = reinterpret_cast<bf16_t*>(p.out);
#pragma unroll
            for (int it = 0; it < 18; ++it) {
                const int v = it * 64 + lane, row = v / 36, c4 = (v - row * 36) * 4;
                const int m = mh + row;
                const gs1_f4 o = *reinterpret_cast<const gs1_f4*>(&slab[row * SROW + c4]);
                if (m < p.M) *reinterpret_cast<uint2*>(outb + (unsigned)(m * p.out_cs + nw + c4)) = make_uint2(pack2bf_valu(o[0], o[1]), pack2bf_valu(o[2], o[3]));
            }
        } else if (kind == 1) {
            const float* resf = reinterpret_cast<const float*>(p.res);
            float* outf = reinterpret_cast<float*>(p.out);
#pragma unroll
            for (int bt = 0; bt < 3; ++bt) {   // three batches of 6 vectors: 6 residual requests back to back, then 6 add + stores
                gs1_f4 rr[6];
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const int v = (bt * 6 + it) * 64 + lane, row = v / 36, c4 = (v - row * 36) * 4;
                    const int m = min(mh + row, p.M - 1);
                    rr[it] = *reinterpret_cast<const gs1_f4*>(resf + (unsigned)(m * p.res_cs + nw + c4));
                }
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const int v = (bt * 6 + it) * 64 + lane, row = v / 36, c4 = (v - row * 36) * 4;
                    const int m = mh + row;
                    const gs1_f4 o = *reinterpret_cast<const gs1_f4*>(&slab[row * SROW + c4]) + rr[it];
                    if (m < p.M) {
                        *reinterpret_cast<gs1_f4*>(outf + (unsigned)(m * p.out_cs + nw + c4)) = o;
                        if (p.out2) *reinterpret_cast<uint2*>(p.out2 + (unsigned)(m * p.out2_cs + nw + c4)) = make_uint2(pack2bf_valu(o[0], o[1]), pack2bf_valu(o[2], o[3]));
                    }
                }
            }
        } else {
            for (int it = 0; it < 18; ++it) {
                const int v = it * 64 + lane, row = v / 36, c4 = (v - row * 36) * 4;
                const int m = mh + row;
                gs1_f4 o = *reinterpret_cast<const gs1_f4*>(&slab[row * SROW + c4]);
                if (m < p.M) {
                    const int n = nw + c4;
                    if (p.res) {
                        const long rm = p.res_mod > 0 ? (long)(m % p.res_mod) : (long)m;
                        if (p.res_f32) o += *reinterpret_cast<const gs1_f4*>(reinterpret_cast<const float*>(p.res) + rm * p.res_cs + n);
                        else {
                            const uint2 rb = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.res) + rm * p.res_cs + n);
                            o += gs1_f4{bflo(rb.x), bfhi(rb.x), bflo(rb.y), bfhi(rb.y)};
                        }
                    }
                    const uint2 pk = make_uint2(pack2bf_valu(o[0], o[1]), pack2bf_valu(o[2], o[3]));
                    if (p.out_f32) *reinterpret_cast<gs1_f4*>(reinterpret_cast<float*>(p.out) + (long)m * p.out_cs + n) = o;
                    else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (long)m * p.out_cs + n) = pk;
                    if (p.out2) *reinterpret_cast<uint2*>(p.out2 + (long)m * p.out2_cs + n) = pk;
                }
            }
        }
    };
    auto pass = [&](auto hc) __attribute__((always_inline)) {
        switch (p.act) {   // ONE uniform switch around the unrolled slab writes
            case IR_ACT_GELU_TANH: write_pass(hc, std::integral_constant<int, IR_ACT_GELU_TANH>{}); break;
            case IR_ACT_GELU_ERF: write_pass(hc, std::integral_constant<int, IR_ACT_GELU_ERF>{}); break;
            case IR_ACT_SILU: write_pass(hc, std::integral_constant<int, IR_ACT_SILU>{}); break;
            case IR_ACT_LRELU: write_pass(hc, std::integral_constant<int, IR_ACT_LRELU>{}); break;
            default: write_pass(hc, std::integral_constant<int, IR_ACT_NONE>{}); break;
        }
        wave_sync();
        rows(decltype(hc)::value);
        wave_sync();   // the next pass's slab writes must not pass this pass's slab reads
    };
    pass(std::integral_constant<int, 0>{});
    pass(std::integral_constant<int, 1>{});
    pass(std::integral_constant<int, 2>{});
    pass(std::integral_constant<int, 3>{});
}

// Which launches take this kernel: what gemm_pp_kernel takes (Cout % 288 == 0, no fused statistics, 32-bit epilogue offsets, enough
// workgroups), with K % 64 == 0 and 16-byte aligned bias / gate rows.
bool ir_gemm_s1_takes(const IGemmParams& p) {
    static const bool off = getenv("IR_NO_GEMM_S1") != nullptr;   // experiment knob
    if (off || g_ir_plain_kernels || p.fp8 || p.taps != 1 || p.force_generic || !p.vec || p.gn_part) return false;
    if (p.Cout != p.Cout_pad || p.Cout % gs1::BN || (p.Cin & 63) || p.Cin < 8 * gs1::BK) return false;
    if ((p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15)) || (p.gate && (reinterpret_cast<uintptr_t>(p.gate) & 15))) return false;
    const long span = (long)p.M * std::max(std::max(p.out_cs, p.res ? p.res_cs : 0), p.out2 ? p.out2_cs : 0);
    if (span >= (1L << 31)) return false;   // the epilogue's 32-bit element offsets
    const long blocks = (long)((p.M + gs1::BM - 1) / gs1::BM) * (p.Cout / gs1::BN);
    return blocks >= 192;   // below that the 128 x 128 kernel fills the chip better
}
int ir_launch_gemm_s1(const IGemmParams& p, hipStream_t s) {
    if (!ir_gemm_s1_takes(p)) return -2;
    const int MT = (p.M + gs1::BM - 1) / gs1::BM, NT = p.Cout / gs1::BN;
    hipLaunchKernelGGL(gemm_s1_kernel, dim3(((MT + 7) / 8) * 8 * NT), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
