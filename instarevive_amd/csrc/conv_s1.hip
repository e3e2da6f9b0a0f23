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
// (three buffers: chunk c+2 is fetched during chunk c, so an HBM miss has nine steps to land - with two buffers the
// waves stalled about 1 us at every chunk boundary) and all nine taps read their pixel fragments from it with a tap offset on the LDS
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
#include "conv_s1_epi.h"

namespace cs1 {
constexpr int TH = 16, TW = 32, HWD = TW + 2, HP = (TH + 2) * HWD;   // 612 halo pixels
constexpr int BK = 32, ROWB = 64;
constexpr int H_Q = (HP + 15) / 16;            // 39 LDS-DMA pieces of 16 pixels x 64 B
constexpr int HALO_BYTES = H_Q * 1024;         // 39 936
constexpr int H_I = 10;                        // halo pieces per wave and chunk (piece q = wave + 4 i, clamped to the last)
constexpr int BN = 128, WT_BYTES = BN * ROWB;  // 8 192: 8 pieces of 16 rows
constexpr int NSB = 4;
constexpr int NHB = 3;                         // halo buffers: chunk c lives in buffer c % 3, chunk c + 2 is fetched during chunk c
constexpr int W_OFF = NHB * HALO_BYTES;        // 119 808
constexpr int LDS_MAIN = W_OFF + NSB * WT_BYTES;   // 152 576
static_assert(cs1e::BYTES <= HALO_BYTES, "epilogue slabs + reduction area must fit one halo buffer");
constexpr int LDS_BYTES = LDS_MAIN;
// The same machine runs a 3 x 3 window (NTAP = 9) or a 2 x 2 window (NTAP = 4, the sub-pixel phases of a nearest-2x upsample + 3 x 3 conv, see
// conv_halo_s1_kernel): K taps per side, a (TH + K - 1) x (TW + K - 1) halo per 32-channel chunk, H_Q pieces of 16 halo pixels, H_I per wave.
template <int NTAP>
struct Geo {
    static constexpr int K = NTAP == 9 ? 3 : 2, HWD = TW + K - 1, HP = (TH + K - 1) * HWD, H_Q = (HP + 15) / 16, H_I = (H_Q + 3) / 4;
};
static_assert(Geo<9>::H_Q == H_Q && Geo<9>::H_I == H_I && Geo<4>::H_Q == 36 && Geo<4>::H_I == 9, "halo piece counts");
// halo pieces issued in the step of tap t: 3 x 3: 10 per chunk and wave, all before tap 6; 2 x 2: 9 per chunk and wave (3, 2, 2, 2)
constexpr int nh(int ntap, int t) { return ntap == 9 ? (t < 4 ? 2 : (t < 6 ? 1 : 0)) : (t == 0 ? 3 : 2); }
constexpr int nh_first(int ntap, int t) { return ntap == 9 ? (t < 4 ? 2 * t : (t < 6 ? 4 + t : 0)) : (t == 0 ? 0 : 1 + 2 * t); }
constexpr int hkey(int hx) { return ((hx >> 2) & 1) << 1; }
}  // namespace cs1

#ifndef IR_KO_S1
#define IR_KO_S1 0   // knock-out builds for timing only (results wrong by design): 1 no epilogue, 2 no main loop, 3 neither,
                     // 4 no halo DMA in the loop, 5 no weight DMA in the loop, 6 no fragment reads, 7 no per-step barrier
#endif
#ifdef IR_S1_STAMPS   // diagnostic build only (tools/conv_s1_stamp.hip): per-workgroup phase sums, never compiled into the library
__device__ unsigned long long g_s1_stamps[1024 * 8];   // [wg][0..4] s_memrealtime sums (100 MHz): prologue, main, drain, passes, gn; [5] s_memtime over main; [6] tiles
#define IR_S1_T(v) const unsigned long long v = __builtin_amdgcn_s_memrealtime()
#define IR_S1_ACC(k, a, b) do { if (tid == 0) g_s1_stamps[blockIdx.x * 8 + (k)] += (b) - (a); } while (0)
#else
#define IR_S1_T(v) do { } while (0)
#define IR_S1_ACC(k, a, b) do { } while (0)
#endif

typedef __attribute__((address_space(3))) void* cs1_lds_t;
template <int LO>
IR_DEVINL void cs1_mfma(bf16x8 w, bf16x8 px) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(px), "n"(LO), "n"(LO + 3));
}

// NTAP = 4 (with UP = 0): a nearest-2x upsample followed by a 3 x 3 convolution (model.py:63-67, the decoder's three Upsample convs: 5.6 of the
// path's 57 conv TFLOP) as FOUR 2 x 2 convolutions on the LOW-resolution tensor, one per output phase (dy, dx) = (row, column parity): output
// pixel (2y + dy, 2x + dx) sees the low-resolution pixels (y - 1 + dy + sy, x - 1 + dx + sx), sy, sx in {0, 1}, through the SUMS of the 3 x 3
// taps that land on the same source pixel (host: weights.pack_conv_up2x2, summed in fp32, rounded to bf16 once) - 16 instead of 36 tap products
// per low-resolution pixel and channel pair, the same zero padding (a source pixel outside the low-resolution image is exactly a padding tap of
// the upsampled one). A tile is 16 x 32 LOW-resolution positions of one phase; its outputs are every other pixel of 32 x 64 high-resolution ones.
// NORM (UP = 0, NTAP = 9): GroupNorm + SiLU of the input applied to every halo IN LDS, one chunk ahead of its use: during the nine steps of chunk c
// each lane takes the 16-byte vectors of "its" LDS-DMA pieces of chunk c + 1 (the same pixel the lane fetched; the channel group is lane & 3 whatever
// the slot swizzle) through scale / shift / SiLU and writes them back, rounded to bf16 exactly as gn_apply_kernel would have stored them. Padding
// pixels (out-of-range buffer loads: zeros) are left alone - the zero padding applies to the NORMALISED tensor. The per-image scale / shift tables
// of the current and the next tile live behind the weight ring. To make room in the arch register file (the 9-tap stream holds two full fragment
// sets, 128 VGPRs) this form keeps two weight-fragment sets but streams the pixel fragments through a ring of three.
// EFULL (round 6; 1: with GroupNorm statistics, 2: without): every tile of the launch is a whole 16 x 32 patch inside the image - the epilogue then
// carries no validity tests and no per-store exec-mask juggling, and the bias rides in the accumulators (conv_s1_epi.h). The launcher decides.
template <int UP, int NTAP, bool NORM = false, int EFULL = 0>
__global__ __launch_bounds__(256, 1) void conv_halo_s1_kernel(IGemmParams p, int tiles_y, int tiles_x, int total_vb) {
    static_assert(!NORM || (UP == 0 && NTAP == 9), "the in-kernel GroupNorm form exists for the plain 9-tap conv");
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub: the buffer-resource type of the body does not exist there, and with it in
                                      // sight hipcc (ROCm 7.2) silently drops the stub of a kernel TEMPLATE (undefined __device_stub__ at load time)
    using namespace cs1;
    using G = Geo<NTAP>;
    constexpr int K = G::K, HWD = G::HWD, HP = G::HP, H_Q = G::H_Q, H_I = G::H_I;   // (shadow the 3 x 3 constants of the namespace)
    constexpr bool PH = NTAP == 4;
    constexpr int NP_OFF = LDS_MAIN, NP_SLOT = 2 * 512 * 4;   // NORM: two tables {scale[512], shift[512]} fp32 (this tile's image, the next tile's)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NORM ? LDS_MAIN + 2 * NP_SLOT : LDS_BYTES];   // halo[0..2] | W ring of 4 ; epilogue: slabs + red in ONE halo buffer
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int c16 = lane & 15, kq = lane >> 4;
    const int NT = p.Cout_pad / BN;
    const int MT = p.NB * tiles_y * tiles_x * (PH ? 4 : 1);
    const int Hc = UP ? 2 * p.H : p.H, Wc = UP ? 2 * p.W : p.W;   // conv-input extent (PH: the low-resolution tensor; the tiles walk it, too)
    const int chunks = p.Cin / BK;                                  // a multiple of 4 (launcher): NTAP * chunks steps, ring slot = step & 3

    // Persistent workgroups (one per CU) walk the virtual block ids bid, bid + gridDim.x, ... (gridDim.x is a multiple of 8: the XCD of a
    // virtual block is the XCD of the workgroup that runs it), and the LDS-DMA stream runs THROUGH the tile boundary: during the last two
    // chunks of a tile the "chunk c + 2" fetches bring the first two chunks of the next tile's halo, the last four steps its first four
    // weight tiles, so the epilogue (whose slabs live in the halo buffer of the tile's last chunk) runs with the next tile's data landing
    // and there is no prologue between tiles.
    struct Tile { int img, trem, oy0, ox0, n0, dy, dx; };   // trem: the tile's slot among the image's GroupNorm partials; dy, dx: its phase (PH)
    auto decode = [&](int bid, Tile& t) -> bool {
        const int xcd = bid & 7, jb = bid >> 3;
        const int mt = (jb / NT) * 8 + xcd, nt = jb % NT;   // an XCD runs the channel tiles of one patch back to back (halo re-read from its L2)
        if (bid >= total_vb || mt >= MT) return false;
        t.n0 = nt * BN;
        const int per_img = tiles_y * tiles_x * (PH ? 4 : 1);
        t.img = mt / per_img;
        t.trem = mt - t.img * per_img;
        // PH: the four phases of a patch back to back (they read the same low-resolution halo: L2 hits)
        const int ph = PH ? (t.trem & 3) : 0, tile = PH ? (t.trem >> 2) : t.trem;
        t.dy = ph >> 1; t.dx = ph & 1;
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        t.oy0 = ty * TH; t.ox0 = tx * TW;
        return true;
    };
    // ---- LDS-DMA sources of a tile, as BUFFER loads (buffer_load_dwordx4 ... lds): a wave-uniform descriptor {tile base, 2 GB range} in
    // scalar registers + the lane's 32-bit byte offset + a scalar chunk / tap offset. A padding pixel's offset is out of the descriptor's
    // range, for which the hardware returns zeros: no zero page, no per-lane base select, and no 64-bit vector arithmetic in the MFMA gaps
    // (round 3 spent 16 vector instructions per step on the addresses of its four pieces, in gaps that have 8 free issue cycles each).
    // Halo piece q covers halo pixels 16q .. 16q+15: lane l -> pixel 16q + (l >> 2), LDS slot l & 3. The base is the first input row the
    // tile's halo touches, so offsets stay below 18 rows x W x in_cs x 2 bytes whatever the tensor's size (17 GB at batch 8).
    constexpr uint32_t OOB = 0xfffffff0u;
    auto tile_base = [&](const Tile& t) -> const bf16_t* {
        const int by = max(t.oy0 - 1 + t.dy, 0) >> UP;
        return p.in + ((long)t.img * p.H + by) * p.W * p.in_cs;
    };
    auto describe = [&](const Tile& t, uint32_t (&hp)[H_I], uint32_t (&wp)[2]) {
        const int by = max(t.oy0 - 1 + t.dy, 0) >> UP;
#pragma unroll
        for (int i = 0; i < H_I; ++i) {
            const int q = min(wu + 4 * i, H_Q - 1);
            const int hpix = q * 16 + (lane >> 2);
            const int hy = hpix / HWD, hx = hpix - hy * HWD;
            const int cy = t.oy0 + hy - 1 + t.dy, cx = t.ox0 + hx - 1 + t.dx;   // (dy = dx = 0 outside the phase form)
            const bool ok = hpix < HP && cy >= 0 && cy < Hc && cx >= 0 && cx < Wc;
            const int iy = min(max(cy, 0), Hc - 1) >> UP, ix = min(max(cx, 0), Wc - 1) >> UP;
            const uint32_t sw = (uint32_t)((lane & 3) ^ hkey(hx));
            hp[i] = ok ? (uint32_t)(((iy - by) * p.W + ix) * p.in_cs) * 2u + sw * 16u : OOB;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {   // weight pieces wave, wave + 4: rows 16 j + (l >> 2)
            const int row = (wu + 4 * i) * 16 + (lane >> 2);
            wp[i] = (uint32_t)(((PH ? (2 * t.dy + t.dx) * p.Cout_pad : 0) + t.n0 + row) * (int)p.wgt_rs) * 2u + (uint32_t)((lane & 3) ^ hkey(row)) * 16u;   // PH: [phase][Cout][4][Cin]
        }
    };
    auto rsrc_of = [&](const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000); };
    const __amdgpu_buffer_rsrc_t w_rsrc = rsrc_of(p.wgt);
    // ---- fragment read addresses. Pixel fragment (patch row 4w + a, half mx) of tap (ky, kx): halo pixel (4w + a + ky, 16 mx + kx + c16),
    // chunk kq; the row term is an immediate. Weight fragment ct: row 16 ct + c16, chunk kq; ct * 1024 is an immediate.
    const uint32_t lds0 = lds_addr(smem);
    uint32_t hrd[K];   // half mx = 1 is 16 pixels = 1024 bytes further (hkey has period 8 in hx): an immediate
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
        const int hx = kx + c16;
        hrd[kx] = lds0 + ((4 * wid) * HWD + hx) * ROWB + ((kq ^ hkey(hx)) << 4);
    }
    const uint32_t wrd = lds0 + W_OFF + c16 * ROWB + ((kq ^ hkey(c16)) << 4);
    bf16x8 fw[2][8], fp[NORM ? 1 : 2][NORM ? 1 : 8];   // [set][channel fragment] / [set][pixel fragment = a * 2 + mx]
    bf16x8 fq[3];                                        // NORM: the pixel fragments stream through a ring of three (fragment PT of tap T in slot (PT + 2 T) % 3)

    Tile cur, nxt;
    int bid = blockIdx.x;
    while (bid < total_vb && !decode(bid, cur)) bid += gridDim.x;
    if (bid >= total_vb) return;
    uint32_t h_ptr[H_I], h_nxt[H_I], w_ptr[2], w_nxt[2];
    describe(cur, h_ptr, w_ptr);
    const bf16_t* base_cur = tile_base(cur);
    const bf16_t* base_nxt = base_cur;

    auto halo_issue = [&](auto ic, int ci, int buf) {   // piece i of this wave of halo chunk ci (>= chunks: of the next tile) into halo buffer buf
        constexpr int i = decltype(ic)::value;
        const int q = min(wu + 4 * i, H_Q - 1);
        const bool mine = ci < chunks;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_of(mine ? base_cur : base_nxt), (cs1_lds_t)(smem + buf * HALO_BYTES + q * 1024), 16,
                                                 (int)(mine ? h_ptr[i] : h_nxt[i]), (mine ? ci : ci - chunks) * (BK * 2), 0, 0);
    };
    auto w_issue = [&](int chunk, int tap, bool mine, int slot) {
        const int koff = tap * p.Cin + chunk * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (cs1_lds_t)(smem + W_OFF + slot * WT_BYTES + (wu + 4 * i) * 1024), 16,
                                                     (int)(mine ? w_ptr[i] : w_nxt[i]), koff * 2, 0, 0);
    };

    // ---- prologue of the FIRST tile only: halo of chunk 0, weight tiles of steps 0..3, halo of chunk 1
    int hb3 = 0;   // halo buffer of the current chunk; advances by one per chunk, across tiles
    [&]<int... I>(std::integer_sequence<int, I...>) { (halo_issue(std::integral_constant<int, I>{}, 0, 0), ...); }(std::make_integer_sequence<int, H_I>{});
#pragma unroll
    for (int t = 0; t < NSB; ++t) w_issue(0, t, true, t);
    [&]<int... I>(std::integer_sequence<int, I...>) { (halo_issue(std::integral_constant<int, I>{}, 1, 1), ...); }(std::make_integer_sequence<int, H_I>{});

    auto step = [&](auto tc, auto setc, int c, int hbuf) {   // hbuf = halo buffer of chunk c
        constexpr int T = decltype(tc)::value, SET = decltype(setc)::value;
        constexpr int TNX = (T + 1) % NTAP, KXN = TNX % K, KYN = TNX / K;
        const int s = c * NTAP + T;
        const uint32_t hb = (uint32_t)(T == NTAP - 1 ? (hbuf == 2 ? 0 : hbuf + 1) : hbuf) * HALO_BYTES;
        const int hfill = hbuf == 0 ? 2 : hbuf - 1;   // buffer of chunk c + 2 = the one chunk c - 1 was read from
        const uint32_t ha = hrd[KXN] + hb;
        const uint32_t wa = wrd + (uint32_t)((s + 1) & 3) * WT_BYTES;
        // weight tile s + 4 = (chunk cw, tap tw); past the end: tile s + 4 - steps (chunk 0, taps 0..3) of the next tile
        constexpr int TW4 = (T + 4) % NTAP;
        int cw = c + (T + 4) / NTAP, tw = TW4;
        const bool wmine = cw < chunks;
        if (!wmine) cw = 0;
        // This step's LDS-DMA pieces: halo piece A behind MFMA 1, piece B behind MFMA 5, the weight pieces behind MFMAs 10 and 13. A piece is
        // one buffer load: descriptor and scalar offset are wave-uniform (selected between this tile and the next by scalar instructions),
        // the lane offset is a register that lives for the whole tile.
        const bool hmine = c + 2 < chunks;
        const int hoff = (hmine ? c + 2 : c + 2 - chunks) * (BK * 2);
        const int wkoff = (tw * p.Cin + cw * BK) * 2;
        auto halo_piece = [&](auto kc) {   // piece k (0 / 1 / 2) of this step
            constexpr int KP = decltype(kc)::value;
            constexpr int PI = nh(NTAP, T) > KP ? nh_first(NTAP, T) + KP : 0;
            const int q = min(wu + 4 * PI, H_Q - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_of(hmine ? base_cur : base_nxt), (cs1_lds_t)(smem + hfill * HALO_BYTES + q * 1024), 16,
                                                     (int)(hmine ? h_ptr[PI] : h_nxt[PI]), hoff, 0, 0);
        };
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                constexpr int PT = I >> 3, CT = I & 7;
                if constexpr ((I & 3) == 0 && IR_KO_S1 != 6) {   // one fragment of the next step per four MFMAs, into the other set
                    constexpr int R = I >> 2;
                    if constexpr (R < 8) fw[SET ^ 1][R] = lds_read16<R * 1024>(wa);
                    else fp[SET ^ 1][R - 8] = lds_read16<(((R - 8) >> 1) + KYN) * HWD * ROWB + ((R - 8) & 1) * 1024>(ha);
                }
                __builtin_amdgcn_sched_barrier(0);
                cs1_mfma<4 * I>(fw[SET][CT], fp[SET][PT]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (I == 1 && nh(NTAP, T) > 0 && IR_KO_S1 != 4) {
                    halo_piece(std::integral_constant<int, 0>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (I == 5 && nh(NTAP, T) > 1 && IR_KO_S1 != 4) {
                    halo_piece(std::integral_constant<int, 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (I == 17 && nh(NTAP, T) > 2 && IR_KO_S1 != 4) {
                    halo_piece(std::integral_constant<int, 2>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr ((I == 10 || I == 13) && IR_KO_S1 != 5) {
                    constexpr int K = I == 10 ? 0 : 1;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (cs1_lds_t)(smem + W_OFF + (s & 3) * WT_BYTES + (wu + 4 * K) * 1024), 16,
                                                             (int)(wmine ? w_ptr[K] : w_nxt[K]), wkoff, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }(), ...);
        }(std::make_integer_sequence<int, 64>{});
        wait_lds<0>();
        // everything but the pieces of this step and the previous one has landed: the weight tile of step s + 2 (read during step s + 1)
        // and, before tap 8, the next chunk's halo (its last piece was issued a chunk ago)
        wait_vm<(IR_KO_S1 == 5 ? 0 : 4) + (IR_KO_S1 == 4 ? 0 : nh(NTAP, (T + NTAP - 1) % NTAP) + nh(NTAP, T))>();
        __builtin_amdgcn_sched_barrier(0);
        if (IR_KO_S1 != 7) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- NORM form
    int tpar = 0;            // table slot of the current tile's image (the next tile's is tpar ^ 1)
    bool first_tile = true;
    uint32_t nkm = 0;        // bit i: hkey of the halo pixel of this lane in its piece i (slot = (lane & 3) ^ key, i.e. vector lane ^ key of the piece)
    if constexpr (NORM) {
#pragma unroll
        for (int i = 0; i < H_I; ++i) {
            const int hpix = min(wu + 4 * i, H_Q - 1) * 16 + (lane >> 2);
            nkm |= (uint32_t)(hkey(hpix % HWD) >> 1) << i;
        }
    }
    float nsc[NORM ? 8 : 1], nsh[NORM ? 8 : 1];
    auto load_tab = [&](int slot, int img) {   // plain loads / stores, outside the pinned stream
        float* tab = reinterpret_cast<float*>(smem + NP_OFF + slot * NP_SLOT);
        for (int i = tid; i < p.Cin; i += 256) {
            tab[i] = p.nrm_scale[(long)img * p.Cin + i];
            tab[512 + i] = p.nrm_shift[(long)img * p.Cin + i];
        }
    };
    // gn_apply_kernel's arithmetic, value for value (silu(a) = a * rcp(1 + __expf(-a)) with __expf(x) = v_exp_f32(x * log2 e), v_cvt_pk rounding), in six
    // stages that ride in six MFMA gaps. The full-rate part runs as packed fp32 pairs (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: IEEE per lane, the
    // same bits as the scalar forms, half the issue slots); the exponential and the reciprocal stay one instruction per value.
    typedef float nf32x2 __attribute__((ext_vector_type(2)));
    nf32x2 nfa[NORM ? 4 : 1], nfe[NORM ? 4 : 1];
#ifndef IR_S1_NORM_SCALAR
#define IR_S1_NORM_SCALAR 1   // 1 (default since round 6: 0.5-1.2 % faster on every shape, profiles/r06_norm_scalar_ab.txt): the full-rate part as scalar fp32 instructions (v_fma_f32 / v_mul_f32 / v_add_f32 through asm, which hipcc cannot re-pack) instead
#endif                        // of the packed pairs: MI355X_MICROARCH.md prices one v_pk_fma_f32 beside MFMAs at +22 cycles over two v_fma_f32
    auto sfma = [](float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; };
    auto smul = [](float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    auto sadd1 = [](float a) { float r; asm("v_add_f32 %0, 1.0, %1" : "=v"(r) : "v"(a)); return r; };
    auto norm_stage = [&](auto stc, const bf16x8& raw, uint4& out) {
        constexpr int ST = decltype(stc)::value;
        if constexpr (ST == 0) {
            // the vector's read: only the pixel and the weight fragment read issued behind it may still fly. The wait is TIED to the value: a bare
            // `asm volatile("s_waitcnt")` orders against other volatile asm only, and hipcc scheduled the unpacks of the vector IN FRONT of it
            // (the value of an asm ds_read looks ready to it) - sporadic stale pieces, different from run to run.
            bf16x8 rw = raw;
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(rw));
            const uint4 u = __builtin_bit_cast(uint4, rw);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (IR_S1_NORM_SCALAR) nfa[e] = nf32x2{sfma(bflo(w[e]), nsc[2 * e], nsh[2 * e]), sfma(bfhi(w[e]), nsc[2 * e + 1], nsh[2 * e + 1])};
                else nfa[e] = nf32x2{bflo(w[e]), bfhi(w[e])} * nf32x2{nsc[2 * e], nsc[2 * e + 1]} + nf32x2{nsh[2 * e], nsh[2 * e + 1]};
            }
        } else if constexpr (ST == 1 || ST == 2) {
#pragma unroll
            for (int e = 2 * (ST - 1); e < 2 * ST; ++e) {
                nf32x2 t;
                if constexpr (IR_S1_NORM_SCALAR) t = nf32x2{smul(nfa[e][0], -1.44269504088896340736f), smul(nfa[e][1], -1.44269504088896340736f)};
                else t = nfa[e] * -1.44269504088896340736f;   // __expf(-a)
                nfe[e] = nf32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
            }
        } else if constexpr (ST == 3 || ST == 4) {
#pragma unroll
            for (int e = 2 * (ST - 3); e < 2 * (ST - 2); ++e) {
                nf32x2 d;
                if constexpr (IR_S1_NORM_SCALAR) d = nf32x2{sadd1(nfe[e][0]), sadd1(nfe[e][1])};
                else d = nfe[e] + 1.0f;
                nfe[e] = nf32x2{fast_rcp(d[0]), fast_rcp(d[1])};
            }
        } else {
            nf32x2 y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (IR_S1_NORM_SCALAR) y[e] = nf32x2{smul(nfa[e][0], nfe[e][0]), smul(nfa[e][1], nfe[e][1])};
                else y[e] = nfa[e] * nfe[e];
            }
            out = make_uint4(pack2bf_valu(y[0][0], y[0][1]), pack2bf_valu(y[1][0], y[1][1]), pack2bf_valu(y[2][0], y[2][1]), pack2bf_valu(y[3][0], y[3][1]));
        }
    };
    auto step_n = [&](auto tc, auto setc, int c, int hbuf) {   // the 9-tap step with streamed pixel fragments and the in-LDS norm of chunk c + 1
        constexpr int T = decltype(tc)::value, SET = decltype(setc)::value;
        constexpr int KX = T % K, KY = T / K;
        constexpr int TNX = (T + 1) % NTAP, KXN = TNX % K, KYN = TNX / K;
        const int s = c * NTAP + T;
        const uint32_t hbc = (uint32_t)hbuf * HALO_BYTES;
        const uint32_t hbn = (uint32_t)(T == NTAP - 1 ? (hbuf == 2 ? 0 : hbuf + 1) : hbuf) * HALO_BYTES;
        const int hfill = hbuf == 0 ? 2 : hbuf - 1;
        const uint32_t ha_c = hrd[KX] + hbc, ha_n = hrd[KXN] + hbn;
        const uint32_t wa = wrd + (uint32_t)((s + 1) & 3) * WT_BYTES;
        constexpr int TW4 = (T + 4) % NTAP;
        int cw = c + (T + 4) / NTAP, tw = TW4;
        const bool wmine = cw < chunks;
        if (!wmine) cw = 0;
        const bool hmine = c + 2 < chunks;
        const int hoff = (hmine ? c + 2 : c + 2 - chunks) * (BK * 2);
        const int wkoff = (tw * p.Cin + cw * BK) * 2;
        auto halo_piece = [&](auto kc) {
            constexpr int KP = decltype(kc)::value;
            constexpr int PI = nh(NTAP, T) > KP ? nh_first(NTAP, T) + KP : 0;
            const int q = min(wu + 4 * PI, H_Q - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_of(hmine ? base_cur : base_nxt), (cs1_lds_t)(smem + hfill * HALO_BYTES + q * 1024), 16,
                                                     (int)(hmine ? h_ptr[PI] : h_nxt[PI]), hoff, 0, 0);
        };
        // ---- the norm of chunk c + 1 (buffer after hbuf): pieces i = T for T < 6, {6, 8} at T = 6, {7, 9} at T = 7, none at T = 8 - every wave must be done
        // one barrier before the first fragment of chunk c + 1 is read (during step 8)
        const bool nx = c + 1 >= chunks;                                         // chunk c + 1 belongs to the next tile
        const uint32_t nbuf = lds0 + (uint32_t)(hbuf == 2 ? 0 : hbuf + 1) * HALO_BYTES;
#ifdef IR_S1_NORM_NOUNITS   // timing experiment: the streamed-fragment form alone (results wrong: nothing is normalised)
        constexpr int NU = 0;
#else
        constexpr int NU = T < 6 ? 1 : (T < 8 ? 2 : 0);
#endif
        constexpr int UI[2] = {T < 8 ? T : 0, T == 6 ? 8 : 9};
        // gaps: one unit: read 22, arithmetic 30..45, write 48; two units: A read 6, arithmetic 14..29, write 31; B read 30, arithmetic 38..53, write 56
        constexpr int RD[2] = {NU == 2 ? 6 : 22, 30}, WR[2] = {NU == 2 ? 31 : 48, 56};
        bf16x8 nraw;
        uint4 nout;
        auto unit_addr = [&](int i) { return nbuf + (uint32_t)(wu + 4 * i) * 1024u + (uint32_t)((lane ^ (((nkm >> i) & 1u) << 1)) * 16); };
        auto unit_ok = [&](int i) { return wu + 4 * i < H_Q && (nx ? h_nxt[i] : h_ptr[i]) != OOB; };
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                constexpr int PT = I >> 3, CT = I & 7, G8 = I >> 3;
                if constexpr ((I & 7) == 0) {   // pixel fragment two groups ahead (the last two groups: fragments 0 / 1 of the NEXT step)
                    if constexpr (I > 0) wait_lds<3>();   // fragment PT of this group has landed: at most the three reads issued behind it are in flight
                    constexpr int P2 = G8 + 2;
                    if constexpr (P2 < 8) fq[(P2 + 2 * T) % 3] = lds_read16<((P2 >> 1) + KY) * HWD * ROWB + (P2 & 1) * 1024>(ha_c);
                    else fq[((P2 - 8) + 2 * TNX) % 3] = lds_read16<(((P2 - 8) >> 1) + KYN) * HWD * ROWB + ((P2 - 8) & 1) * 1024>(ha_n);
                }
                if constexpr ((I & 7) == 2) fw[SET ^ 1][G8] = lds_read16<G8 * 1024>(wa);
                __builtin_amdgcn_sched_barrier(0);
                cs1_mfma<4 * I>(fw[SET][CT], fq[(PT + 2 * T) % 3]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (I == 1 && nh(NTAP, T) > 0) { halo_piece(std::integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0); }
                if constexpr (I == 5 && nh(NTAP, T) > 1) { halo_piece(std::integral_constant<int, 1>{}); __builtin_amdgcn_sched_barrier(0); }
                if constexpr (I == 17 && nh(NTAP, T) > 2) { halo_piece(std::integral_constant<int, 2>{}); __builtin_amdgcn_sched_barrier(0); }
                if constexpr (I == 10 || I == 13) {
                    constexpr int KW = I == 10 ? 0 : 1;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (cs1_lds_t)(smem + W_OFF + (s & 3) * WT_BYTES + (wu + 4 * KW) * 1024), 16,
                                                             (int)(wmine ? w_ptr[KW] : w_nxt[KW]), wkoff, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- norm units
                constexpr int II = I;   // (a pack of the outer fold must not appear inside the inner one)
                [&]<int... UU>(std::integer_sequence<int, UU...>) {
                    ([&] {
                        constexpr int U = UU;
                        if constexpr (U < NU) {
                            if constexpr (II == RD[U]) {
                                nraw = lds_read16<0>(unit_addr(UI[U])); __builtin_amdgcn_sched_barrier(0); }
                            // (a group boundary lies between the read and the first stage: its wait_lds<3> has covered the read)
                            if constexpr (II >= RD[U] + 8 && II <= RD[U] + 23 && (II - RD[U] - 8) % 3 == 0) {
                                norm_stage(std::integral_constant<int, (II - RD[U] - 8) / 3>{}, nraw, nout);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if constexpr (II == WR[U]) {
                                typedef unsigned int nu32x4 __attribute__((ext_vector_type(4)));
                                const nu32x4 o = {nout.x, nout.y, nout.z, nout.w};
                                if (unit_ok(UI[U])) asm volatile("ds_write_b128 %0, %1" ::"v"(unit_addr(UI[U])), "v"(o) : "memory");
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }(), ...);
                }(std::integer_sequence<int, 0, 1>{});
            }(), ...);
        }(std::make_integer_sequence<int, 64>{});
        wait_lds<0>();
        wait_vm<4 + nh(NTAP, (T + NTAP - 1) % NTAP) + nh(NTAP, T)>();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (T == NTAP - 1) if (c + 1 < chunks) {   // the lane's scale / shift for the chunk normalised during the NEXT nine steps (chunk c + 2);
                                                              // behind a tile's last chunk the tile loop loads them (not kept alive across the epilogue)
            const int c2 = c + 2;
            const float* tab = reinterpret_cast<const float*>(smem + NP_OFF + ((c2 < chunks ? tpar : tpar ^ 1) * NP_SLOT)) + (c2 < chunks ? c2 : c2 - chunks) * BK + (lane & 3) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) { nsc[e] = tab[e]; nsh[e] = tab[512 + e]; }
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    for (;;) {
        IR_S1_T(st0);
        // the next tile of this workgroup (none: the stream re-reads the current one, into buffers nobody reads again)
        int nbid = bid + gridDim.x;
        while (nbid < total_vb && !decode(nbid, nxt)) nbid += gridDim.x;
        const bool more = nbid < total_vb;
        if (!more) nxt = cur;
        describe(nxt, h_nxt, w_nxt);
        base_nxt = tile_base(nxt);
        if constexpr (NORM) {
            if (first_tile) load_tab(tpar, cur.img);
            load_tab(tpar ^ 1, nxt.img);   // (the slot of the tile before this one: its last reader passed the barrier that ended that tile's stream)
        }
        if constexpr (EFULL != 0) {
            // the accumulators start at the bias of their channel (tile (pixel fragment PT, channel fragment CT) at a[4 (8 PT + CT) ..+3], a lane holding channels
            // 16 CT + 4 kq .. +3): the epilogue then stores the accumulators as they are (out_scale == 1: launcher)
            [&]<int... CT>(std::integer_sequence<int, CT...>) {
                ([&] {
                    const f32x4 b4 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + cur.n0 + 16 * CT + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (CT == 0)
                        asm volatile(".set ir_cs1_i, 0\n\t.rept 8\n\tv_accvgpr_write_b32 a[ir_cs1_i], %0\n\tv_accvgpr_write_b32 a[ir_cs1_i + 1], %1\n\t"
                                     "v_accvgpr_write_b32 a[ir_cs1_i + 2], %2\n\tv_accvgpr_write_b32 a[ir_cs1_i + 3], %3\n\t.set ir_cs1_i, ir_cs1_i + 32\n\t.endr"
                                     ::"v"(b4[0]), "v"(b4[1]), "v"(b4[2]), "v"(b4[3]) : IR_AGPR256_CLOBBERS);
                    else
                        asm volatile(".set ir_cs1_i, %c4\n\t.rept 8\n\tv_accvgpr_write_b32 a[ir_cs1_i], %0\n\tv_accvgpr_write_b32 a[ir_cs1_i + 1], %1\n\t"
                                     "v_accvgpr_write_b32 a[ir_cs1_i + 2], %2\n\tv_accvgpr_write_b32 a[ir_cs1_i + 3], %3\n\t.set ir_cs1_i, ir_cs1_i + 32\n\t.endr"
                                     ::"v"(b4[0]), "v"(b4[1]), "v"(b4[2]), "v"(b4[3]), "n"(4 * CT));
                }(), ...);
            }(std::make_integer_sequence<int, 8>{});
        } else {
            asm volatile(".set ir_cs1_i, 0\n\t.rept 256\n\tv_accvgpr_write_b32 a[ir_cs1_i], 0\n\t.set ir_cs1_i, ir_cs1_i + 1\n\t.endr" ::: IR_AGPR256_CLOBBERS);
        }
        // everything in flight has landed (first tile: the prologue; later: the pieces fetched through the tile boundary and the previous
        // epilogue's stores) - chunk 1's halo included, which costs nothing after an epilogue and ~1 us once per workgroup
        wait_dma();
        __syncthreads();
        if constexpr (NORM) {
            if (first_tile) {   // chunk 0 of a workgroup's FIRST tile has no chunk before it to ride under: normalise it here (later tiles: during the previous tile's last chunk)
                const float* tab = reinterpret_cast<const float*>(smem + NP_OFF + tpar * NP_SLOT) + (lane & 3) * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) { nsc[e] = tab[e]; nsh[e] = tab[512 + e]; }
                for (int i = 0; i < H_I; ++i) {
                    if (wu + 4 * i >= H_Q || h_ptr[i] == OOB) continue;
                    uint4* v = reinterpret_cast<uint4*>(smem + hb3 * HALO_BYTES + (wu + 4 * i) * 1024 + ((lane ^ (((nkm >> i) & 1u) << 1)) * 16));
                    const uint4 u = *v;
                    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
                    uint32_t o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack2bf_valu(silu(bflo(w[e]) * nsc[2 * e] + nsh[2 * e]), silu(bfhi(w[e]) * nsc[2 * e + 1] + nsh[2 * e + 1]));
                    *v = make_uint4(o[0], o[1], o[2], o[3]);
                }
                __syncthreads();
                first_tile = false;
            }
            const float* tab1 = reinterpret_cast<const float*>(smem + NP_OFF + tpar * NP_SLOT) + BK + (lane & 3) * 8;   // chunk 1 is normalised during chunk 0
#pragma unroll
            for (int e = 0; e < 8; ++e) { nsc[e] = tab1[e]; nsh[e] = tab1[512 + e]; }
        }
        if constexpr (NORM) {
            [&]<int... R>(std::integer_sequence<int, R...>) { ((fw[0][R] = lds_read16<R * 1024>(wrd)), ...); }(std::make_integer_sequence<int, 8>{});
            fq[0] = lds_read16<0>(hrd[0] + (uint32_t)hb3 * HALO_BYTES);
            fq[1] = lds_read16<1024>(hrd[0] + (uint32_t)hb3 * HALO_BYTES);
        } else {
        [&]<int... R>(std::integer_sequence<int, R...>) {
            ([&] {
                if constexpr (R < 8) fw[0][R] = lds_read16<R * 1024>(wrd);
                else fp[0][R - 8] = lds_read16<((R - 8) >> 1) * HWD * ROWB + ((R - 8) & 1) * 1024>(hrd[0] + (uint32_t)hb3 * HALO_BYTES);
            }(), ...);
        }(std::make_integer_sequence<int, 16>{});
        }
        wait_lds<0>();
        IR_S1_T(st1);
#ifdef IR_S1_STAMPS
        const unsigned long long sc0 = __builtin_amdgcn_s_memtime();
#endif
        if (IR_KO_S1 != 2 && IR_KO_S1 != 3)
        for (int c = 0; c < chunks; c += 2) {
            const int hb3b = hb3 == 2 ? 0 : hb3 + 1;
            if constexpr (NORM) {
                [&]<int... U>(std::integer_sequence<int, U...>) { (step_n(std::integral_constant<int, U>{}, std::integral_constant<int, (U & 1)>{}, c, hb3), ...); }(std::make_integer_sequence<int, NTAP>{});
                [&]<int... U>(std::integer_sequence<int, U...>) { (step_n(std::integral_constant<int, U>{}, std::integral_constant<int, ((U + NTAP) & 1)>{}, c + 1, hb3b), ...); }(std::make_integer_sequence<int, NTAP>{});
            } else {
            [&]<int... U>(std::integer_sequence<int, U...>) { (step(std::integral_constant<int, U>{}, std::integral_constant<int, (U & 1)>{}, c, hb3), ...); }(std::make_integer_sequence<int, NTAP>{});
            [&]<int... U>(std::integer_sequence<int, U...>) { (step(std::integral_constant<int, U>{}, std::integral_constant<int, ((U + NTAP) & 1)>{}, c + 1, hb3b), ...); }(std::make_integer_sequence<int, NTAP>{});
            }
            hb3 = hb3b == 2 ? 0 : hb3b + 1;
        }
        // hb3 is now the buffer of the next tile's chunk 0; the last chunk of this tile was read from the one before it

        // ---- epilogue: half a patch row (16 pixels x 128 channels) per wave at a time through a slab in the dead halo buffer
        IR_S1_T(st2);
#ifdef IR_S1_STAMPS
        if (tid == 0) g_s1_stamps[blockIdx.x * 8 + 5] += __builtin_amdgcn_s_memtime() - sc0;
#endif
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // the last MFMA results -> v_accvgpr_read
        IR_S1_T(st3);
        unsigned char* ebuf = smem + (hb3 == 0 ? 2 : hb3 - 1) * HALO_BYTES;   // every wave passed the last barrier after its last read of it
#ifdef IR_S1_STAMPS
        unsigned long long st4v = 0;
        cs1_epilogue<false, EFULL>(p, ebuf, tid, lane, wid, c16, kq, cur.n0, cur.img, cur.oy0, cur.ox0, cur.trem, IR_KO_S1 != 1 && IR_KO_S1 != 3, IR_KO_S1 == 0, &st4v,
                            PH ? 2 : 1, cur.dy, cur.dx, PH ? p.H : p.Ho, PH ? p.W : p.Wo);
        const unsigned long long st4 = st4v;
#else
        cs1_epilogue<false, EFULL>(p, ebuf, tid, lane, wid, c16, kq, cur.n0, cur.img, cur.oy0, cur.ox0, cur.trem, IR_KO_S1 != 1 && IR_KO_S1 != 3, IR_KO_S1 == 0, nullptr,
                            PH ? 2 : 1, cur.dy, cur.dx, PH ? p.H : p.Ho, PH ? p.W : p.Wo);
#endif
        IR_S1_T(st5);
        IR_S1_ACC(0, st0, st1); IR_S1_ACC(1, st1, st2); IR_S1_ACC(2, st2, st3); IR_S1_ACC(3, st3, st4); IR_S1_ACC(4, st4, st5);
#ifdef IR_S1_STAMPS
        if (tid == 0) g_s1_stamps[blockIdx.x * 8 + 6] += 1;
#endif
        if (!more) break;
        tpar ^= 1;
        bid = nbid;
        cur = nxt;
        base_cur = base_nxt;
#pragma unroll
        for (int i = 0; i < H_I; ++i) h_ptr[i] = h_nxt[i];
        w_ptr[0] = w_nxt[0]; w_ptr[1] = w_nxt[1];
    }
    wait_dma();   // the stream's last fetches (a re-read of this tile) must not outlive the workgroup's LDS allocation
#endif
}

// Which launches take this kernel (everything else of the halo family stays with conv_halo_pp_kernel / conv_halo_kernel): plain bf16
// NHWC in and out, 128-channel output tiles, no activation / gate / second output, a bf16 residual at most, and at least 32 patch
// tiles per image.
bool ir_conv_s1_takes(const IGemmParams& p) {
    static const bool off = getenv("IR_NO_CONV_S1") != nullptr;   // experiment knob
    if (off || g_ir_plain_kernels || p.fp8 || p.force_generic) return false;
    if (p.taps != 9 || p.stride != 1 || p.pad != 1 || (p.Cin & 127)) return false;   // chunks % 4 == 0: the weight ring runs through tile boundaries
    if (p.Cout != p.Cout_pad || p.Cout_pad % 128) return false;
    if (p.act != IR_ACT_NONE || p.gate || p.out2 || p.out_f32) return false;
    if (p.res && (p.res_f32 || p.res_mod > 0 || (p.res_cs & 7) || (reinterpret_cast<uintptr_t>(p.res) & 15))) return false;
    if ((p.out_cs & 7) || (reinterpret_cast<uintptr_t>(p.out) & 15)) return false;
    if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15)) return false;
    // per IMAGE, not per launch: which kernel runs (and with it the summation order) must not depend on an image's batch neighbours.
    // 32: a 64 x 64 map at 512 channels and up - batched tiles (--tiled) fill the chip. A caller whose launches are never batched that way (the
    // ControlLDM pipeline at 512 x 512) asks for one tile per CU instead (IGemmParams::s1_min_tiles = 256): below that the persistent 16 x 32 x 128
    // tiles leave CUs idle and the 16 x 16 ping-pong kernel, with four times the workgroups, is faster - 64 x 64 x 512 -> 512 runs 80 us here
    // against 54 us there, 128 x 128 x 512 -> 512 86 against 68 (256 x 256 x 256 -> 256, 256 tiles: 59 against 72); tools/bench_small.py vae.
    const long tiles = (long)((p.Ho + 15) / 16) * ((p.Wo + 31) / 32) * (p.Cout_pad / 128);
    return tiles >= (p.s1_min_tiles > 32 ? p.s1_min_tiles : 32);
}
int ir_conv_s1_tiles(const IGemmParams& p) { return ((p.Ho + 15) / 16) * ((p.Wo + 31) / 32); }

static int cs1_cus() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        return n & ~7;
    }();
    return cus;
}

bool ir_conv_s1_norm_takes(const IGemmParams& p) {
    static const bool off = getenv("IR_NO_S1_NORM") != nullptr;   // experiment knob: the stand-alone GroupNorm apply pass again
    return !off && p.nrm_scale && p.nrm_shift && !p.up && !p.up2x2 && p.Cin <= 512 && ir_conv_s1_takes(p);
}

int ir_launch_conv_s1(const IGemmParams& p, hipStream_t s) {
    if (!ir_conv_s1_takes(p)) return -2;
    if ((p.nrm_scale || p.nrm_shift) && !ir_conv_s1_norm_takes(p)) return -16;
    if (p.gn_part && (p.gn_cpg < 4 || p.gn_cpg > 32 || (p.gn_cpg & (p.gn_cpg - 1)) || p.gn_chunks != ir_conv_s1_tiles(p))) return -13;
    const int tiles_y = (p.Ho + 15) / 16, tiles_x = (p.Wo + 31) / 32;
    const long MT = (long)p.NB * tiles_y * tiles_x, NT = p.Cout_pad / 128;
    const long total = ((MT + 7) / 8) * 8 * NT;
    if (total > 0x7fffffffL) return -12;
    const long grid = total < cs1_cus() ? total : cs1_cus();
    static const bool no_full = getenv("IR_S1_NO_EFULL") != nullptr;   // experiment knob: the general epilogue for every launch
    const int full = (!no_full && p.Ho % 16 == 0 && p.Wo % 32 == 0 && p.out_scale == 1.f) ? (p.gn_part ? 1 : 2) : 0;
    const dim3 g((unsigned)grid), b(256);
    if (p.nrm_scale) {
        if (full == 1) hipLaunchKernelGGL((conv_halo_s1_kernel<0, 9, true, 1>), g, b, 0, s, p, tiles_y, tiles_x, (int)total);
        else if (full == 2) hipLaunchKernelGGL((conv_halo_s1_kernel<0, 9, true, 2>), g, b, 0, s, p, tiles_y, tiles_x, (int)total);
        else hipLaunchKernelGGL((conv_halo_s1_kernel<0, 9, true>), g, b, 0, s, p, tiles_y, tiles_x, (int)total);
    } else if (p.up) hipLaunchKernelGGL((conv_halo_s1_kernel<1, 9>), g, b, 0, s, p, tiles_y, tiles_x, (int)total);
    else if (full == 1) hipLaunchKernelGGL((conv_halo_s1_kernel<0, 9, false, 1>), g, b, 0, s, p, tiles_y, tiles_x, (int)total);
    else if (full == 2) hipLaunchKernelGGL((conv_halo_s1_kernel<0, 9, false, 2>), g, b, 0, s, p, tiles_y, tiles_x, (int)total);
    else hipLaunchKernelGGL((conv_halo_s1_kernel<0, 9>), g, b, 0, s, p, tiles_y, tiles_x, (int)total);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- the sub-pixel phase form of "nearest-2x upsample + 3 x 3 conv" (conv_halo_s1_kernel<0, 4>). p describes the conv as the generic launcher
// sees it (p.up = 1, p.H x p.W the LOW-resolution input, p.Ho = 2 H, p.Wo = 2 W, taps = 9), except that p.wgt holds the four phase matrices
// [phase = 2 dy + dx][Cout][tap = 2 sy + sx][Cin] (weights.pack_conv_up2x2) with p.wgt_rs = 4 * Cin.
bool ir_conv_s1_up2x2_takes(const IGemmParams& pin) {
    static const bool off = getenv("IR_NO_UP2X2") != nullptr;   // experiment knob: the 9-tap form on the upsampled grid
    if (off || !pin.up || pin.res) return false;
    IGemmParams p = pin;   // the low-resolution tile grid decides (4 phases per patch), everything else as for the 9-tap kernel
    p.up = 0; p.Ho = pin.H; p.Wo = pin.W;
    if (!ir_conv_s1_takes(p)) return false;
    return 4L * ((pin.H + 15) / 16) * ((pin.W + 31) / 32) * (pin.Cout_pad / 128) >= 32;
}
int ir_conv_s1_up2x2_tiles(const IGemmParams& p) { return 4 * ((p.H + 15) / 16) * ((p.W + 31) / 32); }   // GroupNorm partial tiles per image

int ir_launch_conv_s1_up2x2(const IGemmParams& p, hipStream_t s) {
    if (!ir_conv_s1_up2x2_takes(p)) return -2;
    if (p.wgt_rs != 4L * p.Cin || p.Ho != 2 * p.H || p.Wo != 2 * p.W) return -3;
    if (p.gn_part && (p.gn_cpg < 4 || p.gn_cpg > 32 || (p.gn_cpg & (p.gn_cpg - 1)) || p.gn_chunks != ir_conv_s1_up2x2_tiles(p))) return -13;
    const int tiles_y = (p.H + 15) / 16, tiles_x = (p.W + 31) / 32;
    const long MT = (long)p.NB * 4 * tiles_y * tiles_x, NT = p.Cout_pad / 128;
    const long total = ((MT + 7) / 8) * 8 * NT;
    if (total > 0x7fffffffL) return -12;
    const long grid = total < cs1_cus() ? total : cs1_cus();
    static const bool no_full = getenv("IR_S1_NO_EFULL") != nullptr;
    if (!no_full && p.gn_part && p.H % 16 == 0 && p.W % 32 == 0 && p.out_scale == 1.f) hipLaunchKernelGGL((conv_halo_s1_kernel<0, 4, false, 1>), dim3((unsigned)grid), dim3(256), 0, s, p, tiles_y, tiles_x, (int)total);
    else hipLaunchKernelGGL((conv_halo_s1_kernel<0, 4>), dim3((unsigned)grid), dim3(256), 0, s, p, tiles_y, tiles_x, (int)total);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
