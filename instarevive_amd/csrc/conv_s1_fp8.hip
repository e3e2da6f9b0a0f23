// conv_halo_s1_fp8_kernel: the one-wave-per-SIMD 3x3 convolution (conv_s1.hip) on fp8 (OCP e4m3) operands - BASELINE.json configs[4],
// the VAE ResnetBlock convolutions (reference ldm/modules/diffusionmodules/model.py:102-116) with GroupNorm+SiLU outputs stored as e4m3
// and weights quantised per output channel. Same patch (16 x 32 output pixels x 128 channels per workgroup, wave w = patch rows 4w..4w+3,
// 64 accumulator tiles of 16 x 16 in all 256 AGPRs), same persistent workgroups with the LDS-DMA stream running through the tile
// boundary, same epilogue (conv_s1_epi.h, here with the per-channel dequantisation gate). What changes is the reduction:
//
//   * the MFMA is v_mfma_scale_f32_16x16x128_f8f6f4 (k = 128 per instruction, 32 cycles: twice the bf16 rate), unit block scales;
//   * LDS rows stay 64 bytes = 64 e4m3 channels (a "chunk"), so halo buffers (612 pixels x 64 B), weight tiles (128 rows x 64 B) and
//     their swizzles are the bf16 kernel's. A HALF-STEP is one (chunk, tap) = 64 k-values; an MFMA step is two consecutive half-steps:
//     the lane quads q = 0, 1 of an operand (k = 32q .. 32q+31) read the two 32-byte halves of the rows of half-step A, quads 2, 3 those
//     of half-step B - two taps (or a tap of each of two chunks) in one instruction, chosen per lane by nothing but its read address.
//     Everything about a step is static: the 18 half-steps of two chunks form 9 step types, and every per-lane read address of every
//     type is computed once per kernel (9 registers);
//   * per MFMA step (64 MFMAs, 2048 matrix cycles) a wave reads 16 fragments of 32 B per lane: the 8 weight fragments stay for the whole
//     step and are refreshed in place behind their last use in the last pixel pass; the 8 pixel fragments go through a ring of three;
//   * two halo buffers (chunk c in buffer c & 1, chunk c + 1 fetched during the first four taps of chunk c) and a ring of six weight
//     tiles (the tiles of step S + 3 are issued during step S into the slots of step S): 129 KB of LDS. One counted vmcnt + one barrier
//     per step: at the end of step S everything issued before step S has landed.
// Dequantisation: out = (acc + bias / gate) * gate per output channel (gate = weight scale / activation scale), p.bias holds bias / gate.
// IGemmParams carries fp8 tensors in 2-byte units (Cin, in_cs, wgt_rs count PAIRS of channels), so byte offsets are "units * 2" as in
// the bf16 kernel.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "agpr256.h"
#include "common.h"
#include "kernels.h"
#include "conv_s1_epi.h"

namespace c8 {
constexpr int TH = 16, TW = 32, HWD = TW + 2, HP = (TH + 2) * HWD;   // 612 halo pixels
constexpr int BK = 32, ROWB = 64;              // chunk = 32 two-byte units = 64 e4m3 channels = one 64-byte row
constexpr int H_Q = (HP + 15) / 16;            // 39 LDS-DMA pieces of 16 pixels x 64 B
constexpr int HALO_BYTES = H_Q * 1024;         // 39 936
constexpr int H_I = 10;                        // halo pieces per wave and chunk (piece q = wave + 4 i, clamped to the last)
constexpr int BN = 128, WT_BYTES = BN * ROWB;  // 8 192: 8 pieces of 16 rows
constexpr int NHB = 2, NSB = 6;
constexpr int W_OFF = NHB * HALO_BYTES;        // 79 872
constexpr int LDS_BYTES = W_OFF + NSB * WT_BYTES;   // 129 024
static_assert(cs1e::BYTES <= HALO_BYTES, "epilogue slabs + reduction area must fit one halo buffer");
constexpr int hkey(int hx) { return ((hx >> 2) & 1) << 1; }
// Step type P = 0..8 of a two-chunk period (chunks c, c + 1; c even): half-steps 2P and 2P + 1 of the 18; half-step h = (chunk h / 9 of
// the period, tap h % 9). Halo fetches by step type: the ten pieces a wave owes chunk c + 1 (buffer 1) in steps 0, 1; those of chunk c + 2
// (buffer 0) in steps 5, 6 - not earlier: step 4 still reads chunk c's last tap out of buffer 0 (the pixel fragments of a step's passes
// 2..7 are read during the step itself), and not later: the chunk's first fragments are read one step before its first step.
constexpr int n_halo(int P) { return (P == 0 || P == 5) ? 6 : ((P == 1 || P == 6) ? 4 : 0); }
constexpr int halo_first(int P) { return (P == 1 || P == 6) ? 6 : 0; }
constexpr int halo_chunk(int P) { return P < 5 ? 1 : 2; }   // relative to c
}  // namespace c8

__device__ uint4 g_zero_page_s1f8[4096];   // 64 KB of zeros: padding taps read from here, the LDS-DMA never needs a mask

typedef __attribute__((address_space(3))) void* c8_lds_t;
typedef __attribute__((ext_vector_type(8))) int c8_i32x8;
IR_DEVINL void c8_glds16(const void* g, c8_lds_t l) { __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0); }
template <int LO>
IR_DEVINL void c8_mfma(c8_i32x8 w, c8_i32x8 px, int unit) {   // a[LO : LO + 3] += W8 (16 channels x 128 k) x PX8 (128 k x 16 pixels), block scales 1
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 a[%c3:%c4], %0, %1, a[%c3:%c4], %2, %2 op_sel_hi:[0,0,0]" ::"v"(w), "v"(px), "v"(unit), "n"(LO), "n"(LO + 3));
}
IR_DEVINL c8_i32x8 c8_join(bf16x8 lo, bf16x8 hi) {
    const uint4 a = __builtin_bit_cast(uint4, lo), b = __builtin_bit_cast(uint4, hi);
    c8_i32x8 r;
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    return r;
}
// pins a value's registers up to this point: an MFMA keeps reading its 8-register operands after it has issued and nothing stalls a
// write into them (attn_fp8.hip found that the hard way); the fragment registers must not be handed back to hipcc right behind the MFMA
template <class T>
IR_DEVINL void c8_keep(const T& x) { asm volatile("" ::"v"(x)); }

// UP = 1: the conv runs on the nearest-2x upsampled input (2H x 2W), folded into the halo's source addresses (as conv_halo_s1_kernel<1>)
template <int UP, int EFULL = 0>   // EFULL: whole-patch launch with statistics (conv_s1_epi.h, round 6)
__global__ __launch_bounds__(256, 1) void conv_halo_s1_fp8_kernel(IGemmParams p, int tiles_y, int tiles_x, int total_vb) {
    using namespace c8;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];   // halo[0..1] | W ring of 6 ; epilogue: slabs + red in halo[1]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int c16 = lane & 15, kq = lane >> 4;
    const int NT = p.Cout_pad / BN;
    const int MT = p.NB * tiles_y * tiles_x;
    const int Hc = UP ? 2 * p.H : p.H, Wc = UP ? 2 * p.W : p.W;   // conv-input (== output) extent
    const int chunks = p.Cin / BK;   // 64-channel chunks; even (launcher): a tile is chunks / 2 periods of 9 MFMA steps
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_zero_page_s1f8);
    const int unit = 0x7F7F7F7F;     // E8M0 exponent 127 = 1.0 in every block-scale byte

    struct Tile { int img, trem, oy0, ox0, n0; };
    auto decode = [&](int bid, Tile& t) -> bool {
        const int xcd = bid & 7, jb = bid >> 3;
        const int mt = (jb / NT) * 8 + xcd, nt = jb % NT;   // an XCD runs the channel tiles of one patch back to back (halo re-read from its L2)
        if (bid >= total_vb || mt >= MT) return false;
        t.n0 = nt * BN;
        t.img = mt / (tiles_y * tiles_x);
        t.trem = mt - t.img * tiles_y * tiles_x;
        const int ty = t.trem / tiles_x, tx = t.trem - ty * tiles_x;
        t.oy0 = ty * TH; t.ox0 = tx * TW;
        return true;
    };
    // LDS-DMA sources of a tile, as 32-bit offsets in 16-byte units (bit 31: padding pixel -> zero page), exactly as in conv_s1.hip
    auto describe = [&](const Tile& t, uint32_t (&hp)[H_I], uint32_t (&wp)[2]) {
#pragma unroll
        for (int i = 0; i < H_I; ++i) {
            const int q = min(wu + 4 * i, H_Q - 1);
            const int hpix = q * 16 + (lane >> 2);
            const int hy = hpix / HWD, hx = hpix - hy * HWD;
            const int cy = t.oy0 + hy - 1, cx = t.ox0 + hx - 1;
            const bool ok = hpix < HP && cy >= 0 && cy < Hc && cx >= 0 && cx < Wc;
            const int iy = min(max(cy, 0), Hc - 1) >> UP, ix = min(max(cx, 0), Wc - 1) >> UP;
            const long pix = ((long)t.img * p.H + iy) * p.W + ix;
            const uint32_t sw = (uint32_t)((lane & 3) ^ hkey(hx));
            hp[i] = ok ? (uint32_t)((pix * p.in_cs) >> 3) + sw : (0x80000000u | sw);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {   // weight pieces wave, wave + 4: rows 16 j + (l >> 2)
            const int row = (wu + 4 * i) * 16 + (lane >> 2);
            wp[i] = (uint32_t)(((long)(t.n0 + row) * p.wgt_rs) >> 3) + (uint32_t)((lane & 3) ^ hkey(row));
        }
    };

    // ---- per-lane fragment read addresses, one per step type. Lane quad kq reads the 32-byte half (kq & 1) of the rows of half-step
    // (kq >> 1) of the step. Pixel fragment (patch row 4w + a, half mx): halo pixel (4w + a + ky, 16 mx + kx + c16) of the half-step's tap
    // in the half-step's halo buffer; a * HWD * 64 + mx * 1024 are immediates. Weight fragment ct: row 16 ct + c16 of the half-step's
    // tile = slot pair (S % 3) * 2 + (kq >> 1); ct * 1024 and the pair are immediates.
    const uint32_t lds0 = lds_addr(smem);
    uint32_t prd[9];
    [&]<int... PS>(std::integer_sequence<int, PS...>) {
        ([&] {
            constexpr int P = PS;
            const int h = 2 * P + (kq >> 1);
            const int tap = h % 9, ky = tap / 3, kx = tap - 3 * ky, buf = (h / 9) & 1;
            const int hx = kx + c16;
            prd[P] = lds0 + buf * HALO_BYTES + ((4 * wid + ky) * HWD + hx) * ROWB + (((2 * (kq & 1)) ^ hkey(hx)) << 4);
        }(), ...);
    }(std::make_integer_sequence<int, 9>{});
    const uint32_t wrd = lds0 + W_OFF + (kq >> 1) * WT_BYTES + c16 * ROWB + (((2 * (kq & 1)) ^ hkey(c16)) << 4);
    bf16x8 fwl[8], fwh[8], fpl[3], fph[3];   // weight fragments [channel tile] / pixel fragment ring; l / h = the two 16-byte reads of a 32-byte operand

    Tile cur, nxt;
    int bid = blockIdx.x;
    while (bid < total_vb && !decode(bid, cur)) bid += gridDim.x;
    if (bid >= total_vb) return;
    uint32_t h_ptr[H_I], h_nxt[H_I], w_ptr[2], w_nxt[2];
    describe(cur, h_ptr, w_ptr);

    // source of halo piece i of this wave of chunk ci of the current tile (ci >= chunks: of the next tile); it lands in buffer ci & 1
    auto halo_src = [&](int i, int ci) -> const unsigned char* {
        const bool mine = ci < chunks;
        const uint32_t d = mine ? h_ptr[i] : h_nxt[i];
        const unsigned char* base = (d >> 31) ? zero : reinterpret_cast<const unsigned char*>(p.in);
        return base + ((unsigned long long)(d & 0x7fffffffu) << 4) + (mine ? ci : ci - chunks) * (BK * 2);
    };
    auto halo_issue = [&](auto ic, int ci) {
        constexpr int i = decltype(ic)::value;
        const int q = min(wu + 4 * i, H_Q - 1);
        c8_glds16(halo_src(i, ci), (c8_lds_t)(smem + (ci & 1) * HALO_BYTES + q * 1024));
    };
    // the weight tile of half-step hh (absolute index within the current tile; >= 9 * chunks: of the next tile) into ring slot hh % 6
    auto w_issue = [&](int hh, int slot) {
        const int steps = 9 * chunks;
        const bool mine = hh < steps;
        const int hr = mine ? hh : hh - steps;
        const int chunk = hr / 9, tap = hr - 9 * chunk;
        const int koff = tap * p.Cin + chunk * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            c8_glds16(reinterpret_cast<const unsigned char*>(p.wgt) + ((unsigned long long)(mine ? w_ptr[i] : w_nxt[i]) << 4) + koff * 2,
                      (c8_lds_t)(smem + W_OFF + slot * WT_BYTES + (wu + 4 * i) * 1024));
    };

    // ---- prologue of the FIRST tile only: halo of chunk 0, weight tiles of half-steps 0..5 (steps 0..2)
    [&]<int... I>(std::integer_sequence<int, I...>) { (halo_issue(std::integral_constant<int, I>{}, 0), ...); }(std::make_integer_sequence<int, H_I>{});
#pragma unroll
    for (int t = 0; t < NSB; ++t) w_issue(t, t);

    // fragment reads of a step of type P: weight fragment ct / pixel fragment f (= a * 2 + mx)
    auto read_w = [&](auto pc, auto ctc) {
        constexpr int P = decltype(pc)::value, CT = decltype(ctc)::value, OFF = (P % 3) * 2 * WT_BYTES + CT * 1024;
        fwl[CT] = lds_read16<OFF>(wrd);
        fwh[CT] = lds_read16<OFF + 16>(wrd);
    };
    auto read_p = [&](auto pc, auto fc, auto slotc) {
        constexpr int P = decltype(pc)::value, F = decltype(fc)::value, SL = decltype(slotc)::value, OFF = (F >> 1) * HWD * ROWB + (F & 1) * 1024;
        const uint32_t a = prd[P];
        fpl[SL] = lds_read16<OFF>(a);
        fph[SL] = lds_read16<OFF + 16>(a);
    };

    // One MFMA step of type P of the period that starts at (even) chunk c. Pixel pass PT (8 MFMAs: the channel tiles) uses ring slot
    // (2P + PT) % 3 (global pass number 8 (9k + P) + PT mod 3). DMA issued here: the weight tiles of step S + 3 (half-steps +6, +7, into
    // this step's own slots, dead since the barrier that opened the step) and the halo pieces the two half-steps owe the next chunk.
    auto step = [&](auto pc, int c) {
        constexpr int P = decltype(pc)::value, PN = (P + 1) % 9;
        constexpr int hA = 2 * P, hB = 2 * P + 1;
        constexpr int NH = n_halo(P);
        const int s0 = c * 9 + hA;   // absolute half-step index of hA in this tile
        const unsigned char* hsrc[NH > 0 ? NH : 1];
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                constexpr int I_ = I;
                if constexpr (I_ / 4 < NH && (I_ & 3) == 0)   // MFMA gaps 0, 4, 8, ...: the source address of halo piece I / 4 of this step
                    hsrc[I_ / 4] = halo_src(halo_first(P) + I_ / 4, c + halo_chunk(P));
                if constexpr (I_ / 4 < NH && (I_ & 3) == 2) {   // two gaps later: issue it
                    const int q = min(wu + 4 * (halo_first(P) + I_ / 4), H_Q - 1);
                    c8_glds16(hsrc[I_ / 4], (c8_lds_t)(smem + (halo_chunk(P) & 1) * HALO_BYTES + q * 1024));
                }
                if constexpr (I_ == 26) w_issue(s0 + 6, (hA + 6) % 6);
                if constexpr (I_ == 34) w_issue(s0 + 7, (hB + 6) % 6);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int PT = I_ >> 3, CT = I_ & 7, SL = (2 * P + PT) % 3;
                if constexpr (CT == 0 && PT >= 2) {   // this pass's pixel fragment was read two passes ago; the next pass's (2 reads) may still fly
                    wait_lds<2>();
                    __builtin_amdgcn_sched_barrier(0);
                }
                c8_mfma<4 * I_>(c8_join(fwl[CT], fwh[CT]), c8_join(fpl[SL], fph[SL]), unit);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (CT == 1) {   // behind the second MFMA of pass PT: the fragment of pass PT + 2 into the slot pass PT - 1 used
                    constexpr int SLN = (2 * P + PT + 2) % 3;
                    c8_keep(fpl[SLN]); c8_keep(fph[SLN]);
                    if constexpr (PT + 2 < 8) read_p(std::integral_constant<int, P>{}, std::integral_constant<int, PT + 2>{}, std::integral_constant<int, SLN>{});
                    else read_p(std::integral_constant<int, PN>{}, std::integral_constant<int, PT + 2 - 8>{}, std::integral_constant<int, SLN>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (PT == 7 && CT >= 2) {   // last pass: weight fragment CT - 2 has had its last use two MFMAs ago -> next step's
                    c8_keep(fwl[CT - 2]); c8_keep(fwh[CT - 2]);
                    read_w(std::integral_constant<int, PN>{}, std::integral_constant<int, CT - 2>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            }(), ...);
        }(std::make_integer_sequence<int, 64>{});
        c8_keep(fwl[6]); c8_keep(fwh[6]); c8_keep(fwl[7]); c8_keep(fwh[7]);
        read_w(std::integral_constant<int, PN>{}, std::integral_constant<int, 6>{});
        read_w(std::integral_constant<int, PN>{}, std::integral_constant<int, 7>{});
        wait_lds<0>();
        // everything issued BEFORE this step has landed: the weight tiles of step S + 2 (issued in S - 1; their fragments are read during
        // S + 1) and the halo pieces of steps <= S - 1. Outstanding at most: this step's own 4 weight + NH halo pieces.
        wait_vm<4 + NH>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    for (;;) {
        int nbid = bid + gridDim.x;
        while (nbid < total_vb && !decode(nbid, nxt)) nbid += gridDim.x;
        const bool more = nbid < total_vb;
        if (!more) nxt = cur;   // the stream re-reads the current tile into buffers nobody reads again
        describe(nxt, h_nxt, w_nxt);
        asm volatile(".set ir_c8_i, 0\n\t.rept 256\n\tv_accvgpr_write_b32 a[ir_c8_i], 0\n\t.set ir_c8_i, ir_c8_i + 1\n\t.endr" ::: IR_AGPR256_CLOBBERS);
        // everything in flight has landed (first tile: the prologue; later: the pieces fetched through the tile boundary and the previous
        // epilogue's stores)
        wait_dma();
        __syncthreads();
        {   // step 0's fragments: all weight fragments, the pixel fragments of passes 0 and 1 (ring slots 0, 1)
            using P0 = std::integral_constant<int, 0>;
            [&]<int... R>(std::integer_sequence<int, R...>) { (read_w(P0{}, std::integral_constant<int, R>{}), ...); }(std::make_integer_sequence<int, 8>{});
            read_p(P0{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            read_p(P0{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            wait_lds<0>();
        }
        for (int c = 0; c < chunks; c += 2)
            [&]<int... U>(std::integer_sequence<int, U...>) { (step(std::integral_constant<int, U>{}, c), ...); }(std::make_integer_sequence<int, 9>{});

        // ---- epilogue through slabs in halo buffer 1 (the last chunk's: every wave passed the last barrier after its last read of it;
        // the next tile's chunk 0 is landing in buffer 0, its chunk 1 is fetched during its own first steps)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMA results -> v_accvgpr_read
        cs1_epilogue<true, EFULL>(p, smem + HALO_BYTES, tid, lane, wid, c16, kq, cur.n0, cur.img, cur.oy0, cur.ox0, cur.trem, true, true, nullptr, 1, 0, 0, p.Ho, p.Wo);
        if (!more) break;
        bid = nbid;
        cur = nxt;
#pragma unroll
        for (int i = 0; i < H_I; ++i) h_ptr[i] = h_nxt[i];
        w_ptr[0] = w_nxt[0]; w_ptr[1] = w_nxt[1];
    }
    wait_dma();   // the stream's last fetches (a re-read of this tile) must not outlive the workgroup's LDS allocation
}

// Which launches take this kernel: the fp8 form of ir_conv_s1_takes (stride-1 3x3, e4m3 NHWC in, bf16 out, per-channel gate, a bf16
// residual at most, Cin a multiple of 128 channels - an even number of 64-channel chunks -, 128-channel output tiles, >= 32 patch tiles
// per image). Everything else with fp8 operands stays with conv_halo_kernel<.., FP8> (igemm.hip), which is also the plain-kernel reference.
bool ir_conv_s1_fp8_takes(const IGemmParams& p) {
    static const bool off = getenv("IR_NO_CONV_S1_FP8") != nullptr;   // experiment knob
    if (off || g_ir_plain_kernels || !p.fp8 || p.force_generic) return false;
    if (p.taps != 9 || p.stride != 1 || p.pad != 1 || (p.Cin & 63)) return false;   // Cin counts pairs: 64 pairs = 128 channels
    if (p.Cout != p.Cout_pad || p.Cout_pad % 128) return false;
    if (p.act != IR_ACT_NONE || !p.gate || p.gate_stride != 0 || p.out2 || p.out_f32) return false;
    if ((reinterpret_cast<uintptr_t>(p.gate) & 15) || (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15))) return false;
    if (p.res && (p.res_f32 || p.res_mod > 0 || (p.res_cs & 7) || (reinterpret_cast<uintptr_t>(p.res) & 15))) return false;
    if ((p.out_cs & 7) || (reinterpret_cast<uintptr_t>(p.out) & 15) || (p.in_cs & 7) || (p.wgt_rs & 7)) return false;
    const long tiles = (long)((p.Ho + 15) / 16) * ((p.Wo + 31) / 32) * (p.Cout_pad / 128);
    return tiles >= 32;
}

int ir_launch_conv_s1_fp8(const IGemmParams& p, hipStream_t s) {
    if (!ir_conv_s1_fp8_takes(p)) return -2;
    if (p.gn_part && (p.gn_cpg < 4 || p.gn_cpg > 32 || (p.gn_cpg & (p.gn_cpg - 1)) || p.gn_chunks != ir_conv_s1_tiles(p))) return -13;
    const int tiles_y = (p.Ho + 15) / 16, tiles_x = (p.Wo + 31) / 32;
    const long MT = (long)p.NB * tiles_y * tiles_x, NT = p.Cout_pad / 128;
    const long total = ((MT + 7) / 8) * 8 * NT;
    if (total > 0x7fffffffL) return -12;
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        return n & ~7;
    }();
    const long grid = total < cus ? total : cus;
    static const bool no_full = getenv("IR_S1_NO_EFULL") != nullptr;   // experiment knob (shared with conv_s1.hip)
    if (p.up) hipLaunchKernelGGL(conv_halo_s1_fp8_kernel<1>, dim3((unsigned)grid), dim3(256), 0, s, p, tiles_y, tiles_x, (int)total);
    else if (!no_full && p.gn_part && p.Ho % 16 == 0 && p.Wo % 32 == 0) hipLaunchKernelGGL((conv_halo_s1_fp8_kernel<0, 1>), dim3((unsigned)grid), dim3(256), 0, s, p, tiles_y, tiles_x, (int)total);
    else hipLaunchKernelGGL(conv_halo_s1_fp8_kernel<0>, dim3((unsigned)grid), dim3(256), 0, s, p, tiles_y, tiles_x, (int)total);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
