// Fused halves of the SwinIR transformer block for gfx950 (reference diffusion/model/swinir.py:250-290: x = x + MLP(LN2(x)) is
// swin_mlp_kernel below). The stage is latency / memory bound as separate launches (K = 192 GEMMs at 15 % MFMA busy, a LayerNorm
// pass and a 50 MB hidden-state round trip per block); fused, a token's state stays in registers from the residual stream to the
// residual stream.
//
// Transposed-register formulation. Every product is computed transposed, Y^T[out][token] = W[out][k] . X^T[k][token], with the
// WEIGHTS as the MFMA A operand (from LDS, shared by the workgroup) and the ACTIVATIONS as the B operand. A 32x32 accumulator tile
// then has its token on the lane and its channels in the registers, which - packed to bf16 - is exactly the B operand of the next
// product (cdna_hip_programming.md "An accumulator tile as the next MFMA's operand"), so activations never touch LDS or HBM between
// the two linears. The k order this imposes (position p = 16s + 8h + j of k-step s holds channel 16s + 8(j >> 2) + 4h + (j & 3)) is
// baked into the weight tiles on the host (weights.py: pack_swin_mlp), for fc1's input channels and fc2's hidden units alike.
// The residual stream is read in the same "accumulator layout": lane (token r, half h) holds the 4-channel groups 32t + 8i + 4h.
//
// swin_mlp_kernel: one wave = 32 tokens, one workgroup = 8 waves = 256 tokens (two waves per SIMD, so one wave's GELU arithmetic
// overlaps the other's MFMAs). Per 32 hidden units jt: H^T = W1[jt] . LN(x)^T (12 MFMAs) -> + bias, exact-erf GELU -> bf16 ->
// Y^T += W2[:, jt] . H^T (12 MFMAs). The two weight tiles of a step (12.5 + 15 KB, rows padded on the host so that the LDS image is
// the memory image and conflict-free) arrive by LDS-DMA in a 2-slot ring, one barrier per step.
#include "common.h"
#include "kernels.h"

typedef __attribute__((address_space(3))) void* sw_lds_t;
IR_DEVINL void sw_glds16(const void* g, sw_lds_t l) { __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0); }

namespace swf {
constexpr int CP = 192;                      // padded width of the residual stream (channels C..191 are zero)
constexpr int W1_ROW = 400, W1_TILE = 13312; // 32 rows x (384 + 16) B, padded to 13 DMA pieces
constexpr int W2_ROW = 80, W2_TILE = 15360;  // 192 rows x (64 + 16) B = 15 DMA pieces
constexpr int SLOT = W1_TILE + W2_TILE;      // 28 672 B
constexpr int PIECES = SLOT / 1024;          // 28
constexpr int VEC_OFF = 2 * SLOT;            // fp32 vectors behind the ring: g[192] b[192] b1[hid_p <= 512] b2[192] | next block's LN1 g[192] b[192]
constexpr int NEXT_OFF = 1152;               // float offset of the next block's LayerNorm vectors inside the vector area
constexpr int STG_OFF = VEC_OFF + 6144;      // wave-private staging of token rows: 8 waves x 3 slices x 4 KB
constexpr int STG_WAVE = 3 * 4096;
constexpr int LDS_TOTAL = STG_OFF + 8 * STG_WAVE;   // 161 792 B
}  // namespace swf

// x, out: [T][192] fp32 (out may alias x); out2 (optional): [T][192] bf16 copy of the result; w: [NJ] tiles {W1 tile | W2 tile};
// vec: g[192] b[192] b1[32*NJ] b2[192] fp32; C: real channels (LayerNorm extent); NJ: hidden units / 32.
// Token rows travel between HBM and the accumulator layout through a wave-private LDS staging area: a lane needs 16-byte groups of
// ITS token (768-byte rows: 64 different cache lines per load instruction if read directly - measured 57 us per launch, most of it
// the memory path), so each 32-channel slice t of the wave's 32 tokens (one 128-byte line per token) is brought in by LDS-DMA
// pieces of 8 tokens x 128 B, chunk-swizzled on the source side, and read back as the lane's groups 32t + 8i + 4h; results leave
// the same way (LDS image -> coalesced 16-byte stores). Three slices (12 KB per wave) at a time.
// next_g / next_b (optional, C fp32 each): out2 then receives LayerNorm(result; next_g, next_b) instead of the plain bf16 copy - the norm1
// of the NEXT SwinTransformerBlock (swinir.py:263), whose qkv GEMM reads out2 directly; the stand-alone LayerNorm launch between two
// blocks of an RSTB disappears. The new token row is still in registers when the last product ends, so this costs one more pass through
// the wave-private staging area.
// QKV (with LN_NEXT): the next block's qkv projection too (swinir.py:263-271 up to the window partition: per token, so any tiling of the
// tokens serves). wq: NQ / 2 ring slots of two W1-format tiles each (32 output channels x 192 k positions, weights.pack_swin_qkv_tiles), bq: the
// 32 NQ biases; LN(new row) is packed straight into B fragments - as LN2(x) is at the top - and out2 receives [T][32 NQ] bf16 (q | k | v of the
// next block) instead of the normalised row: the qkv GEMM launch between two blocks of an RSTB and its input's round trip disappear.
template <bool LN_NEXT, bool QKV = false>
__global__ __launch_bounds__(512, 1) void swin_mlp_kernel(const float* __restrict__ x, float* __restrict__ out, bf16_t* __restrict__ out2,
                                                          const unsigned char* __restrict__ w, const float* __restrict__ vec, long T, int C,
                                                          int NJ, float eps, const float* __restrict__ next_g, const float* __restrict__ next_b,
                                                          const unsigned char* __restrict__ wq, const float* __restrict__ bq, int NQ) {
    using namespace swf;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int r = lane & 31, h = lane >> 5;
    const long tok0 = (long)blockIdx.x * 256 + wu * 32;                  // first token of this wave

    // weight ring: piece q of a step's 28 KB goes to wave q % 8 (3.5 pieces per wave: waves 0-3 take four)
    const int NS = NJ + (QKV ? NQ / 2 : 0);   // ring steps: the MLP's, then the qkv tile pairs
    auto stage = [&](int jt, int slot) {
        const unsigned char* src = (QKV && jt >= NJ ? wq + (long)(jt - NJ) * SLOT : w + (long)jt * SLOT) + lane * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (wu + 8 * i < PIECES) sw_glds16(src + (wu + 8 * i) * 1024, (sw_lds_t)(smem + slot * SLOT + (wu + 8 * i) * 1024));
    };
    stage(0, 0);
    float* vs = reinterpret_cast<float*>(smem + VEC_OFF);
    const int nvec = 3 * CP + 32 * NJ;
    for (int i = tid; i < nvec; i += 512) vs[i] = vec[i];
    constexpr bool ln_next = LN_NEXT;
    if (ln_next)
        for (int i = tid; i < 2 * CP; i += 512) {
            const int c = i < CP ? i : i - CP;
            vs[NEXT_OFF + i] = c < C ? (i < CP ? next_g[c] : next_b[c]) : 0.f;   // padded channels normalise to exactly 0
        }

    // wave-private staging: 3 slices x [32 tokens][128 B]; piece p of a slice = tokens 8p .. 8p+7, lane L -> token 8p + (L >> 3), slot L & 7
    unsigned char* stg = smem + STG_OFF + wu * STG_WAVE;
    const int ptok = lane >> 3, pslot = lane & 7;
    auto load_half = [&](const float* base, int t0) {   // slices t0, t0+1, t0+2 of the wave's tokens -> staging
#pragma unroll
        for (int ts = 0; ts < 3; ++ts)
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
                const int tk = 8 * pc + ptok;
                const long gt = min(tok0 + tk, T - 1);                  // clamped: a ragged last workgroup re-reads the last token
                sw_glds16(base + gt * CP + 32 * (t0 + ts) + 4 * (pslot ^ ((tk >> 1) & 7)), (sw_lds_t)(stg + ts * 4096 + pc * 1024));
            }
    };
    // chunk c of token k sits at slot c ^ ((k >> 1) & 7): the 16 lanes a ds_read_b128 serves together then hit 16 different bank groups
    auto frag_addr = [&](int ts, int i) { return stg + ts * 4096 + r * 128 + (((2 * i + h) ^ ((r >> 1) & 7)) << 4); };

    // ---- the token's row in the accumulator layout: xr[t][i] = channels 32t + 8i + 4h .. +3
    f32x4 xr[6][4];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        load_half(x, 3 * hf);
        wait_dma();   // wave-private data: the wave's own wait is all that is needed before its own reads
#pragma unroll
        for (int ts = 0; ts < 3; ++ts)
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[3 * hf + ts][i] = *reinterpret_cast<const f32x4*>(frag_addr(ts, i));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // reads of this half are done before the next half's DMA overwrites it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // LayerNorm over the C real channels (two passes over registers; the halves of a token sit on lanes r and r + 32)
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) s1 += (xr[t][i][0] + xr[t][i][1]) + (xr[t][i][2] + xr[t][i][3]);   // padded channels are zero
    s1 += __shfl_xor(s1, 32);
    const float mean = s1 / (float)C;
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = xr[t][i][e] - mean;
                s2 += (32 * t + 8 * i + 4 * h + e < C) ? d * d : 0.f;
            }
    s2 += __shfl_xor(s2, 32);
    const float rstd = rsqrtf(s2 / (float)C + eps);
    __syncthreads();  // the vectors are in LDS
    // B fragments of LN(x): k-step s = the two 4-channel groups (i = 2(s & 1), +1) of row tile s >> 1
    bf16x8 xn[12];
#pragma unroll
    for (int s = 0; s < 12; ++s) {
        uint32_t wv[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int t = s >> 1, i = 2 * (s & 1) + q, c0 = 32 * t + 8 * i + 4 * h;
            const f32x4 g = *reinterpret_cast<const f32x4*>(vs + c0), b = *reinterpret_cast<const f32x4*>(vs + CP + c0);
            const f32x4 v = (xr[t][i] - mean) * rstd * g + b;   // padded channels: g = b = 0
            wv[2 * q] = pack2bf(v[0], v[1]);
            wv[2 * q + 1] = pack2bf(v[2], v[3]);
        }
        xn[s] = __builtin_bit_cast(bf16x8, make_uint4(wv[0], wv[1], wv[2], wv[3]));
    }

    // The output accumulators start at x + b2 (the residual row is in registers in exactly their layout: register 4i + e of tile t is channel
    // 32t + 8i + 4h + e), so the second product lands on the residual stream directly and the row is not fetched a second time at the end.
    f32x16 y[6];
#pragma unroll
    for (int ot = 0; ot < 6; ++ot)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(vs + 2 * CP + 32 * NJ + 32 * ot + 8 * i + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[ot][4 * i + e] = xr[ot][i][e] + b2[e];
        }
    const int a1 = r * W1_ROW + h * 16;                 // + s * 32
    const int a2 = W1_TILE + r * W2_ROW + h * 16;       // + ot * 32 * W2_ROW + s2 * 32
    // (Waves w and w + 4 share a SIMD and run this loop in lockstep - both in their MFMAs, then both in their GELU arithmetic: the
    // wave spends about half its cycles waiting (SQ_WAIT_ANY). Running waves 4-7 one product late needs a third ring slot, which the
    // 160 KB do not have next to the row staging; tried with a second barrier per step instead: spills and no gain.)
    for (int jt = 0; jt < NJ; ++jt) {
        wait_dma();
        __syncthreads();           // tile jt has landed; every wave is done with tile jt - 1
        if (jt + 1 < NS) stage(jt + 1, (jt + 1) & 1);
        const unsigned char* sl = smem + (jt & 1) * SLOT;
        // H^T = W1[jt] . LN(x)^T
        f32x16 hacc;
#pragma unroll
        for (int g = 0; g < 16; ++g) hacc[g] = 0.f;
#pragma unroll
        for (int s = 0; s < 12; ++s) hacc = mfma32(*reinterpret_cast<const bf16x8*>(sl + a1 + s * 32), xn[s], hacc);
        // + bias, exact GELU (nn.GELU default), bf16: registers 8q .. 8q+7 are the B fragment of hidden k-step q
        bf16x8 hb[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint32_t wv[4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(vs + 2 * CP + 32 * jt + 16 * q + 8 * u + 4 * h);
                const int g0 = 8 * q + 4 * u;
                wv[2 * u] = pack2bf(gelu_erf(hacc[g0] + b1[0]), gelu_erf(hacc[g0 + 1] + b1[1]));
                wv[2 * u + 1] = pack2bf(gelu_erf(hacc[g0 + 2] + b1[2]), gelu_erf(hacc[g0 + 3] + b1[3]));
            }
            hb[q] = __builtin_bit_cast(bf16x8, make_uint4(wv[0], wv[1], wv[2], wv[3]));
        }
        // Y^T += W2[:, jt] . H^T
#pragma unroll
        for (int ot = 0; ot < 6; ++ot)
#pragma unroll
            for (int q = 0; q < 2; ++q) y[ot] = mfma32(*reinterpret_cast<const bf16x8*>(sl + a2 + ot * 32 * W2_ROW + q * 32), hb[q], y[ot]);
    }
    // ---- the new rows -> residual stream, three slices at a time through the staging area: the lane writes its groups (accumulator layout:
    // register 4i + e of tile t is channel 32t + 8i + 4h + e), then the image leaves with coalesced 16-byte stores (8 lanes per 128-byte
    // line), the bf16 copy as 8-byte stores
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int ts = 0; ts < 3; ++ts)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 3 * hf + ts;
                const f32x4 nx = f32x4{y[t][4 * i], y[t][4 * i + 1], y[t][4 * i + 2], y[t][4 * i + 3]};
                *reinterpret_cast<f32x4*>(frag_addr(ts, i)) = nx;
                if (ln_next) xr[t][i] = nx;   // the new row, kept for the next block's LayerNorm
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ts = 0; ts < 3; ++ts)
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
                const int tk = 8 * pc + ptok;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + ts * 4096 + pc * 1024 + lane * 16);
                const long gt = tok0 + tk;
                const int ch = 32 * (3 * hf + ts) + 4 * (pslot ^ ((tk >> 1) & 7));
                if (gt < T) {
                    *reinterpret_cast<f32x4*>(out + gt * CP + ch) = v;
                    if (out2 && !ln_next) *reinterpret_cast<uint2*>(out2 + gt * CP + ch) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image has been read before the next half's DMA lands on it
    }
    if (!ln_next || !out2) return;
    // ---- out2 = LayerNorm(new row) with the next block's norm1 parameters, same two-pass arithmetic as above
    float t1 = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) t1 += (xr[t][i][0] + xr[t][i][1]) + (xr[t][i][2] + xr[t][i][3]);   // padded channels are zero
    t1 += __shfl_xor(t1, 32);
    const float mean2 = t1 / (float)C;
    float t2 = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = xr[t][i][e] - mean2;
                t2 += (32 * t + 8 * i + 4 * h + e < C) ? d * d : 0.f;
            }
    t2 += __shfl_xor(t2, 32);
    const float rstd2 = rsqrtf(t2 / (float)C + eps);
    // straight from the accumulator layout: lanes r and r + 32 hold adjacent 8-byte pieces of token r's row (16 B per token and
    // instruction; a write needs no staging - nothing waits for it)
    const long gtok = tok0 + r;
    if constexpr (!QKV) {
        if (gtok < T) {
            bf16_t* orow = out2 + gtok * CP + 4 * h;
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c0 = 32 * t + 8 * i + 4 * h;
                    const f32x4 g = *reinterpret_cast<const f32x4*>(vs + NEXT_OFF + c0), b = *reinterpret_cast<const f32x4*>(vs + NEXT_OFF + CP + c0);
                    const f32x4 v = (xr[t][i] - mean2) * rstd2 * g + b;
                    *reinterpret_cast<uint2*>(orow + 32 * t + 8 * i) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                }
        }
    } else {
        // B fragments of LN(new row), as xn above
        bf16x8 xq[12];
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            uint32_t wv[4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int t = s >> 1, i = 2 * (s & 1) + q, c0 = 32 * t + 8 * i + 4 * h;
                const f32x4 g = *reinterpret_cast<const f32x4*>(vs + NEXT_OFF + c0), b = *reinterpret_cast<const f32x4*>(vs + NEXT_OFF + CP + c0);
                const f32x4 v = (xr[t][i] - mean2) * rstd2 * g + b;
                wv[2 * q] = pack2bf(v[0], v[1]);
                wv[2 * q + 1] = pack2bf(v[2], v[3]);
            }
            xq[s] = __builtin_bit_cast(bf16x8, make_uint4(wv[0], wv[1], wv[2], wv[3]));
        }
        const int LDQ = 32 * NQ;
        static_assert(STG_WAVE == 3 * 4096, "the qkv rows leave through the three staging slices in turn");
        for (int st = 0; st < NQ / 2; ++st) {
            const int jt = NJ + st;
            f32x4 bv[2][4];   // the two tiles' biases (L2-resident, the same for every token): in flight across the barrier
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[u][i] = *reinterpret_cast<const f32x4*>(bq + 32 * (2 * st + u) + 8 * i + 4 * h);
            wait_dma();
            __syncthreads();       // tile pair st has landed; every wave is done with the previous slot
            if (jt + 1 < NS) stage(jt + 1, (jt + 1) & 1);
            const unsigned char* sl = smem + (jt & 1) * SLOT;
            // the pair's 64 channels of the wave's 32 tokens = 128-byte row segments: through a staging slice (16-byte chunk c of token k at slot
            // c ^ ((k >> 1) & 7), as the token rows above) and out as full segments, 8 lanes per token. Written straight from the accumulator
            // layout (8 bytes per lane, 32 rows per instruction) the tail was bound by the addresser's line rate, as swin_attn_proj_kernel's was.
            unsigned char* sb = stg + (st % 3) * 4096;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[g] = 0.f;
#pragma unroll
                for (int s = 0; s < 12; ++s) acc = mfma32(*reinterpret_cast<const bf16x8*>(sl + u * W1_TILE + a1 + s * 32), xq[s], acc);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<uint2*>(sb + r * 128 + (((4 * u + i) ^ ((r >> 1) & 7)) << 4) + 8 * h) =
                        make_uint2(pack2bf(acc[4 * i] + bv[u][i][0], acc[4 * i + 1] + bv[u][i][1]), pack2bf(acc[4 * i + 2] + bv[u][i][2], acc[4 * i + 3] + bv[u][i][3]));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
                const int tk = 8 * pc + ptok;
                const uint4 v = *reinterpret_cast<const uint4*>(sb + pc * 1024 + lane * 16);
                const long gt = tok0 + tk;
                if (gt < T) *reinterpret_cast<uint4*>(out2 + gt * LDQ + 64 * st + 8 * (pslot ^ ((tk >> 1) & 7))) = v;
            }
        }
    }
}

int ir_launch_swin_mlp(const float* x, float* out, bf16_t* out2, const void* w_tiles, const float* vec, long T, int C, int hid_p, float eps,
                       hipStream_t s, const float* next_g, const float* next_b, const void* qkv_tiles, const float* qkv_b, int qkv_n) {
    if (T <= 0 || C <= 0 || C > swf::CP || hid_p <= 0 || (hid_p & 31) || hid_p > 512) return -2;
    if ((next_g != nullptr) != (next_b != nullptr) || (next_g && (!out2 || (C & 3)))) return -2;
    if (qkv_tiles && (!next_g || !qkv_b || qkv_n <= 0 || (qkv_n & 63))) return -2;   // pairs of 32-channel tiles behind the next block's norm1
    const int NJ = hid_p / 32, NQ = qkv_tiles ? qkv_n / 32 : 0;
    const size_t lds = swf::LDS_TOTAL;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(swin_mlp_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(swin_mlp_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(swin_mlp_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        attr_set = true;
    }
    const dim3 grid((unsigned)((T + 255) / 256));
    const unsigned char* wt = reinterpret_cast<const unsigned char*>(w_tiles);
    const unsigned char* wq = reinterpret_cast<const unsigned char*>(qkv_tiles);
    if (qkv_tiles)
        hipLaunchKernelGGL((swin_mlp_kernel<true, true>), grid, dim3(512), lds, s, x, out, out2, wt, vec, T, C, NJ, eps, next_g, next_b, wq, qkv_b, NQ);
    else if (next_g)
        hipLaunchKernelGGL((swin_mlp_kernel<true, false>), grid, dim3(512), lds, s, x, out, out2, wt, vec, T, C, NJ, eps, next_g, next_b, wq, qkv_b, NQ);
    else
        hipLaunchKernelGGL((swin_mlp_kernel<false, false>), grid, dim3(512), lds, s, x, out, out2, wt, vec, T, C, NJ, eps, next_g, next_b, wq, qkv_b, NQ);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// =====================================================================================================================
// swin_attn_proj_kernel: (S)W-MSA of one 8 x 8 window for ALL heads + the output projection + the residual add in one launch
// (reference diffusion/model/swinir.py:101-165 WindowAttention.forward, :258-290 SwinTransformerBlock.forward up to the first residual).
// Replaces swin_window_attn_kernel (one workgroup per (window, head), operands staged through LDS with a 2-byte V^T scatter: 53 % LDS
// bank conflicts, 2 % MFMA busy, bias / mask index arithmetic per element) and the proj GEMM launch behind it.
//
// One WAVE per (window, 32-query half); no LDS on the operand side: every MFMA operand is either a 16-byte row piece loaded straight from global
// memory (a lane's own token row: Q as B operand, K and V rows as A operands, the permuted proj weights as A operands) or a packed
// accumulator tile (transposed-register formulation, as in swin_mlp_kernel above):
//   S^T[key][query]   = K Q^T                      2 key tiles x 2 k-steps; lane = query, registers = keys
//   P^T               = softmax over keys (bias table in the log2 domain, shift mask as a per-lane bit set computed once per wave)
//   V'[key][d]        = V I                        V rows times the identity: the same values with lane = d, registers = keys - the
//                                                  A operand the second product needs, without a transpose through LDS (exact in bf16)
//   O^T[d][query]     = V'^T P^T                   lane = query, registers = d
//   Y^T[ch][query]   += Wp[:, head] O^T            6 channel tiles x 2 k-steps per head; Wp's columns are stored in accumulator order
//                                                  (weights.pack_swinir: proj_t), so a packed O^T tile IS the B operand, and in
//                                                  fragment order, so an A-operand load is one contiguous KB (by row it touched 32 lines)
// and finally Y + bias + x is written to the fp32 token rows (16-byte pieces: a lane holds 4 consecutive channels per 4 registers).
// The roll / window partition / reverse of the reference are the token gather tok(): shifted-frame pixel (Y, X) -> source pixel
// ((Y + s) % H, (X + s) % W), as in swin_window_attn_kernel.
IR_DEVINL uint4 swa_pack8(const f32x16& v, int lo, float mul) {
    return make_uint4(pack2bf(v[lo] * mul, v[lo + 1] * mul), pack2bf(v[lo + 2] * mul, v[lo + 3] * mul), pack2bf(v[lo + 4] * mul, v[lo + 5] * mul),
                      pack2bf(v[lo + 6] * mul, v[lo + 7] * mul));
}

IR_DEVINL uint4 swa_pack8_valu(const f32x16& v, int lo, float mul) {   // for values the VALU produced (exponentials, O * 1/sum): one instruction
    // per pair; with mul = 1 the operands may come straight from v_exp_f32, hence the form with the transcendental-use wait state
    return make_uint4(pack2bf_trans(v[lo] * mul, v[lo + 1] * mul), pack2bf_trans(v[lo + 2] * mul, v[lo + 3] * mul),
                      pack2bf_trans(v[lo + 4] * mul, v[lo + 5] * mul), pack2bf_trans(v[lo + 6] * mul, v[lo + 7] * mul));
}

__global__ __launch_bounds__(256, 2) void swin_attn_proj_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ xres, float* __restrict__ out,
                                                             const bf16_t* __restrict__ proj_t, const float* __restrict__ proj_b,
                                                             const float* __restrict__ biasT, int H, int W, int shift, float scale_log2,
                                                             long n_waves) {
    constexpr int HEADS = 6, CP = 192, LD = 3 * CP;
    __shared__ __attribute__((aligned(16))) float yslab[4 * 32 * 36];   // epilogue only: one 32 x 32 output tile per wave (rows padded to 36 floats)
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);   // wave = (window, query half)
    if (wv >= n_waves) return;
    const int g = (int)(wv & 1);
    long wlin = wv >> 1;
    const int nwx = W >> 3, nwy = H >> 3;
    const int wx = (int)(wlin % nwx); wlin /= nwx;
    const int wy = (int)(wlin % nwy);
    const long img = wlin / nwy;
    auto tok = [&](int i) {   // window token i (row-major in the 8 x 8 window) -> token index in the image
        int y = wy * 8 + (i >> 3) + shift, x = wx * 8 + (i & 7) + shift;
        if (y >= H) y -= H;
        if (x >= W) x -= W;
        return (long)y * W + x;
    };
    const long tbase = img * H * W;
    const long qtok = tbase + tok(32 * g + r);                       // this lane's query (n index of S^T / O^T / Y^T)
    const long ktok[2] = {tbase + tok(r), tbase + tok(32 + r)};      // this lane's key rows (m index) of the two key tiles
    // shifted-window mask (swinir.py:227-248): folded into the bias table on the host. A window of the shifted frame is one of four classes -
    // interior, last window column, last window row, corner - and biasT holds one [head][key][query] table per class for a shifted block
    // (weights.swin_masked_bias: class 0 is the plain relative-position bias, the others carry -100 log2(e) where key and query lie in different
    // regions), so the kernel only picks a table. As 32 per-lane mask bits tested in the head loop the mask cost 64 SGPRs of lane masks.
    const int cls = shift ? ((wy == nwy - 1 ? 2 : 0) | (wx == nwx - 1 ? 1 : 0)) : 0;
    // identity B operand of the V' product: lane = d = r, k position 16 ks + 8 h + e holds (that position == r)
    bf16x8 idf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        uint32_t w4[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            const int c0 = 16 * ks + 8 * h + 2 * e2;
            w4[e2] = (c0 == r ? 0x3f80u : 0u) | (c0 + 1 == r ? 0x3f800000u : 0u);
        }
        idf[ks] = __builtin_bit_cast(bf16x8, make_uint4(w4[0], w4[1], w4[2], w4[3]));
    }
    f32x16 yacc[6];
#pragma unroll
    for (int ct = 0; ct < 6; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) yacc[ct][e] = 0.f;
    const bf16_t* qrow = qkv + qtok * LD + 8 * h;
    const bf16_t* krow[2] = {qkv + ktok[0] * LD + CP + 8 * h, qkv + ktok[1] * LD + CP + 8 * h};
    const bf16_t* vrow[2] = {qkv + ktok[0] * LD + 2 * CP + 8 * h, qkv + ktok[1] * LD + 2 * CP + 8 * h};
    const float* brow = biasT + (long)cls * HEADS * 4096 + 32 * g + r;   // + (head * 64 + key) * 64

#pragma unroll 1
    for (int hd = 0; hd < HEADS; ++hd) {
        const int co = hd * 32;
        bf16x8 qf[2], kf[2][2], vf[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + co + 16 * ks);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                kf[kt][ks] = *reinterpret_cast<const bf16x8*>(krow[kt] + co + 16 * ks);
                vf[kt][ks] = *reinterpret_cast<const bf16x8*>(vrow[kt] + co + 16 * ks);
            }
        }
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kt][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) s[kt] = mfma32(kf[kt][ks], qf[ks], s[kt]);
        }
        // bias (+ mask), softmax over the 64 keys of this query (32 here, 32 in the partner lane)
        const float* bt = brow + (long)hd * 4096;
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = 32 * kt + (e & 3) + 8 * (e >> 2) + 4 * h;
                float v = s[kt][e] * scale_log2 + bt[key * 64];
                s[kt][e] = v;
                mx = fmaxf(mx, v);
            }
        mx = xhalf_max(mx);
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pv = __builtin_amdgcn_exp2f(s[kt][e] - mx);
                s[kt][e] = pv;
                rs += pv;
            }
        rs += __shfl_xor(rs, 32);
        // V' = V I (lane = d, registers = keys), then O^T = V'^T P^T
        f32x16 o;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 vt;
#pragma unroll
            for (int e = 0; e < 16; ++e) vt[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) vt = mfma32(vf[kt][ks], idf[ks], vt);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                o = mfma32(__builtin_bit_cast(bf16x8, swa_pack8(vt, 8 * s2, 1.0f)), __builtin_bit_cast(bf16x8, swa_pack8_valu(s[kt], 8 * s2, 1.0f)), o);
        }
        const float inv = 1.0f / rs;
        // Y^T += Wp[:, head] O^T: A = proj_t rows (channel 32 ct + r), 16-byte piece at column head * 32 + 16 s2 + 8 h
        const bf16x8 ob[2] = {__builtin_bit_cast(bf16x8, swa_pack8_valu(o, 0, inv)), __builtin_bit_cast(bf16x8, swa_pack8_valu(o, 8, inv))};
#pragma unroll
        for (int ct = 0; ct < 6; ++ct)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 wp = *reinterpret_cast<const bf16x8*>(proj_t + (((hd * 6 + ct) * 2 + s2) * 64 + lane) * 8);   // fragment order: one contiguous KB
                yacc[ct] = mfma32(wp, ob[s2], yacc[ct]);
            }
    }
    // out[token][ch] = Y + bias + x. In the accumulator layout a lane holds 16-byte pieces of ITS token row (32 different rows per store
    // instruction: the kernel is bound by the texture addresser's line rate, TA busy 83 %), so each 32-channel tile goes through a
    // wave-private LDS slab and leaves as 128-byte row segments: 8 lanes per token, 8 tokens per instruction.
    float* ys = &yslab[(threadIdx.x >> 6) * 32 * 36];
    long rtok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rtok[i] = (tbase + tok(32 * g + 8 * i + (lane >> 3))) * CP + (lane & 7) * 4;
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(&ys[r * 36 + 8 * j + 4 * h]) = f32x4{yacc[ct][4 * j], yacc[ct][4 * j + 1], yacc[ct][4 * j + 2], yacc[ct][4 * j + 3]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const f32x4 bv = *reinterpret_cast<const f32x4*>(proj_b + 32 * ct + (lane & 7) * 4);
        f32x4 xv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const f32x4*>(xres + rtok[i] + 32 * ct);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 y = *reinterpret_cast<const f32x4*>(&ys[(8 * i + (lane >> 3)) * 36 + (lane & 7) * 4]) + bv + xv[i];
            *reinterpret_cast<f32x4*>(out + rtok[i] + 32 * ct) = y;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

int ir_launch_swin_attn_proj(const bf16_t* qkv, const float* xres, float* out, const void* proj_t, const float* proj_b, const float* biasT, int B,
                             int H, int W, int shift, float scale, hipStream_t s) {
    if ((H & 7) || (W & 7) || shift < 0 || shift >= 8 || B <= 0) return -2;
    if ((long)B * H * W * 576 * 2 >= (1L << 32)) return -4;   // the kernels address the qkv tensor with 32-bit byte offsets (ir_swin_fused_fits)
    const long n_waves = 2L * B * (H >> 3) * (W >> 3);
    const long blocks = (n_waves + 3) / 4;
    if (blocks > 0x7fffffffL) return -4;
    hipLaunchKernelGGL(swin_attn_proj_kernel, dim3((unsigned)blocks), dim3(256), 0, s, qkv, xres, out, reinterpret_cast<const bf16_t*>(proj_t), proj_b, biasT, H, W,
                       shift, scale * 1.44269504088896340736f, n_waves);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// =====================================================================================================================
// swin_block_kernel: a WHOLE SwinTransformerBlock behind its qkv projection in one launch (swinir.py:258-290) - the attention half of
// swin_attn_proj_kernel and the MLP half of swin_mlp_kernel with nothing between them: a wave owns (window, 32-query half), the projected
// attention output stays in its 96 accumulator registers, + bias + x (its token rows, gathered through the staging slices) IS the row
// LayerNorm2 reads and the value the second product of the MLP accumulates onto (swin_mlp_kernel's x + b2 start), so the post-attention residual
// stream never goes to memory; the tail is swin_mlp_kernel's (the new rows, then optionally norm1 / norm1 + qkv of the NEXT block), with the
// wave's window tokens as the row addresses. One workgroup = 8 waves = 4 windows; HBM traffic of a block 250 MB (qkv in, x in, x out, qkv out)
// against 350 MB as two launches. x_in and x_out may alias (a wave touches only its own tokens' rows; qkv likewise: every wave of the workgroup
// has finished reading K / V of its window - several workgroup barriers ago - when the first q | k | v row of the next block is written).
template <bool LN_NEXT, bool QKV>
__global__ __launch_bounds__(512, 1) void swin_block_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ x_in, float* __restrict__ x_out,
                                                            bf16_t* __restrict__ out2, const bf16_t* __restrict__ proj_t,
                                                            const float* __restrict__ proj_b, const float* __restrict__ biasT, int H, int W, int shift,
                                                            float scale_log2, long n_waves, const unsigned char* __restrict__ w,
                                                            const float* __restrict__ vec, int C, int NJ, float eps, const float* __restrict__ next_g,
                                                            const float* __restrict__ next_b, const unsigned char* __restrict__ wq,
                                                            const float* __restrict__ bq, int NQ) {
    using namespace swf;
    constexpr int HEADS = 6, LD = 3 * CP;
#ifdef IR_SWIN_STAMPS
    unsigned long long stp[8];
    stp[0] = __builtin_amdgcn_s_memrealtime();
#define IR_STAMP(k) stp[k] = __builtin_amdgcn_s_memrealtime()
#else
#define IR_STAMP(k)
#endif
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int r = lane & 31, h = lane >> 5;
    long wv = (long)blockIdx.x * 8 + wu;               // wave = (window, query half)
    const bool active = wv < n_waves;                  // a ragged last workgroup: idle waves redo the last wave's work and store nothing
    if (!active) wv = n_waves - 1;

    // ---- weight ring + vectors, as swin_mlp_kernel (in flight under the whole attention phase)
    const int NS = NJ + (QKV ? NQ / 2 : 0);
    auto stage = [&](int jt, int slot) {
        const unsigned char* src = (QKV && jt >= NJ ? wq + (long)(jt - NJ) * SLOT : w + (long)jt * SLOT) + lane * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (wu + 8 * i < PIECES) sw_glds16(src + (wu + 8 * i) * 1024, (sw_lds_t)(smem + slot * SLOT + (wu + 8 * i) * 1024));
    };
    stage(0, 0);
    float* vs = reinterpret_cast<float*>(smem + VEC_OFF);
    const int nvec = 3 * CP + 32 * NJ;
    for (int i = tid; i < nvec; i += 512) vs[i] = vec[i];
    constexpr bool ln_next = LN_NEXT;
    if (ln_next)
        for (int i = tid; i < 2 * CP; i += 512) {
            const int c = i < CP ? i : i - CP;
            vs[NEXT_OFF + i] = c < C ? (i < CP ? next_g[c] : next_b[c]) : 0.f;
        }

    // ---- window geometry (swin_attn_proj_kernel)
    const int g = (int)(wv & 1);
    long wlin = wv >> 1;
    const int nwx = W >> 3, nwy = H >> 3;
    const int wx = (int)(wlin % nwx); wlin /= nwx;
    const int wy = (int)(wlin % nwy);
    const long img = wlin / nwy;
    auto tok = [&](int i) {
        int y = wy * 8 + (i >> 3) + shift, x = wx * 8 + (i & 7) + shift;
        if (y >= H) y -= H;
        if (x >= W) x -= W;
        return (long)y * W + x;
    };
    const long tbase = img * H * W;
    const long qtok = tbase + tok(32 * g + r);
    f32x16 yacc[6];
    {
        const long ktok[2] = {tbase + tok(r), tbase + tok(32 + r)};
        const int cls = shift ? ((wy == nwy - 1 ? 2 : 0) | (wx == nwx - 1 ? 1 : 0)) : 0;   // the window's mask class: its bias table (swin_attn_proj_kernel)
        bf16x8 idf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint32_t w4[4];
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const int c0 = 16 * ks + 8 * h + 2 * e2;
                w4[e2] = (c0 == r ? 0x3f80u : 0u) | (c0 + 1 == r ? 0x3f800000u : 0u);
            }
            idf[ks] = __builtin_bit_cast(bf16x8, make_uint4(w4[0], w4[1], w4[2], w4[3]));
        }
#pragma unroll
        for (int ct = 0; ct < 6; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) yacc[ct][e] = 0.f;
        // 32-bit byte offsets from the uniform bases (launcher: the qkv tensor is below 4 GB): one address register per row instead of two
        const unsigned char* qkvb = reinterpret_cast<const unsigned char*>(qkv);
        const uint32_t qoff = (uint32_t)(qtok * LD + 8 * h) * 2u;
        const uint32_t koff[2] = {(uint32_t)(ktok[0] * LD + CP + 8 * h) * 2u, (uint32_t)(ktok[1] * LD + CP + 8 * h) * 2u};
        const float* brow = biasT + (long)cls * HEADS * 4096 + 32 * g + r;
#pragma unroll 1
        for (int hd = 0; hd < HEADS; ++hd) {
            const int co = hd * 32;
            bf16x8 qf[2], kf[2][2], vf[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qf[ks] = *reinterpret_cast<const bf16x8*>(qkvb + qoff + 2 * (co + 16 * ks));
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    kf[kt][ks] = *reinterpret_cast<const bf16x8*>(qkvb + koff[kt] + 2 * (co + 16 * ks));
                    vf[kt][ks] = *reinterpret_cast<const bf16x8*>(qkvb + koff[kt] + 2 * (CP + co + 16 * ks));
                }
            }
            f32x16 s[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) s[kt][e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) s[kt] = mfma32(kf[kt][ks], qf[ks], s[kt]);
            }
            const float* bt = brow + (long)hd * 4096;
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = 32 * kt + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const float v = s[kt][e] * scale_log2 + bt[key * 64];
                    s[kt][e] = v;
                    mx = fmaxf(mx, v);
                }
            mx = xhalf_max(mx);
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(s[kt][e] - mx);
                    s[kt][e] = pv;
                    rs += pv;
                }
            rs += __shfl_xor(rs, 32);
            f32x16 o;
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x16 vt;
#pragma unroll
                for (int e = 0; e < 16; ++e) vt[e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) vt = mfma32(vf[kt][ks], idf[ks], vt);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    o = mfma32(__builtin_bit_cast(bf16x8, swa_pack8(vt, 8 * s2, 1.0f)), __builtin_bit_cast(bf16x8, swa_pack8_valu(s[kt], 8 * s2, 1.0f)), o);
            }
            const float inv = 1.0f / rs;
            const bf16x8 ob[2] = {__builtin_bit_cast(bf16x8, swa_pack8_valu(o, 0, inv)), __builtin_bit_cast(bf16x8, swa_pack8_valu(o, 8, inv))};
#pragma unroll
            for (int ct = 0; ct < 6; ++ct)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 wp = *reinterpret_cast<const bf16x8*>(proj_t + (((hd * 6 + ct) * 2 + s2) * 64 + lane) * 8);
                    yacc[ct] = mfma32(wp, ob[s2], yacc[ct]);
                }
        }
    }

    IR_STAMP(1);
    // ---- x rows of the wave's 32 window tokens through the staging slices; piece pc = window row 4g + pc (8 tokens), lane -> token ptok, slot pslot
    unsigned char* stg = smem + STG_OFF + wu * STG_WAVE;
    const int ptok = lane >> 3, pslot = lane & 7;
    long gtk[4];   // row index of the lane's token in piece pc
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) gtk[pc] = tbase + tok(32 * g + 8 * pc + ptok);
    auto load_half = [&](int t0) {
#pragma unroll
        for (int ts = 0; ts < 3; ++ts)
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
                const int tk = 8 * pc + ptok;
                sw_glds16(x_in + gtk[pc] * CP + 32 * (t0 + ts) + 4 * (pslot ^ ((tk >> 1) & 7)), (sw_lds_t)(stg + ts * 4096 + pc * 1024));
            }
    };
    auto frag_addr = [&](int ts, int i) { return stg + ts * 4096 + r * 128 + (((2 * i + h) ^ ((r >> 1) & 7)) << 4); };
    // the row after the attention residual, accumulator layout: xr[t][i] = channels 32t + 8i + 4h .. +3 = attention + proj bias + x
    f32x4 xr[6][4];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        load_half(3 * hf);
        wait_dma();
#pragma unroll
        for (int ts = 0; ts < 3; ++ts)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 3 * hf + ts;
                const f32x4 pb = *reinterpret_cast<const f32x4*>(proj_b + 32 * t + 8 * i + 4 * h);
                xr[t][i] = f32x4{yacc[t][4 * i], yacc[t][4 * i + 1], yacc[t][4 * i + 2], yacc[t][4 * i + 3]} + pb + *reinterpret_cast<const f32x4*>(frag_addr(ts, i));
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    IR_STAMP(2);
    // ---- from here on swin_mlp_kernel with gathered rows
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) s1 += (xr[t][i][0] + xr[t][i][1]) + (xr[t][i][2] + xr[t][i][3]);
    s1 += __shfl_xor(s1, 32);
    const float mean = s1 / (float)C;
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = xr[t][i][e] - mean;
                s2 += (32 * t + 8 * i + 4 * h + e < C) ? d * d : 0.f;
            }
    s2 += __shfl_xor(s2, 32);
    const float rstd = rsqrtf(s2 / (float)C + eps);
    __syncthreads();  // the vectors are in LDS
    bf16x8 xn[12];
#pragma unroll
    for (int s = 0; s < 12; ++s) {
        uint32_t wv4[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int t = s >> 1, i = 2 * (s & 1) + q, c0 = 32 * t + 8 * i + 4 * h;
            const f32x4 gg = *reinterpret_cast<const f32x4*>(vs + c0), bb = *reinterpret_cast<const f32x4*>(vs + CP + c0);
            const f32x4 v = (xr[t][i] - mean) * rstd * gg + bb;
            wv4[2 * q] = pack2bf(v[0], v[1]);
            wv4[2 * q + 1] = pack2bf(v[2], v[3]);
        }
        xn[s] = __builtin_bit_cast(bf16x8, make_uint4(wv4[0], wv4[1], wv4[2], wv4[3]));
    }
    f32x16 y[6];
#pragma unroll
    for (int ot = 0; ot < 6; ++ot)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(vs + 2 * CP + 32 * NJ + 32 * ot + 8 * i + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[ot][4 * i + e] = xr[ot][i][e] + b2[e];
        }
    const int a1 = r * W1_ROW + h * 16;
    const int a2 = W1_TILE + r * W2_ROW + h * 16;
    for (int jt = 0; jt < NJ; ++jt) {
        wait_dma();
        __syncthreads();
        if (jt + 1 < NS) stage(jt + 1, (jt + 1) & 1);
        const unsigned char* sl = smem + (jt & 1) * SLOT;
        f32x16 hacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) hacc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < 12; ++s) hacc = mfma32(*reinterpret_cast<const bf16x8*>(sl + a1 + s * 32), xn[s], hacc);
        bf16x8 hb[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint32_t wv4[4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(vs + 2 * CP + 32 * jt + 16 * q + 8 * u + 4 * h);
                const int g0 = 8 * q + 4 * u;
                wv4[2 * u] = pack2bf(gelu_erf(hacc[g0] + b1[0]), gelu_erf(hacc[g0 + 1] + b1[1]));
                wv4[2 * u + 1] = pack2bf(gelu_erf(hacc[g0 + 2] + b1[2]), gelu_erf(hacc[g0 + 3] + b1[3]));
            }
            hb[q] = __builtin_bit_cast(bf16x8, make_uint4(wv4[0], wv4[1], wv4[2], wv4[3]));
        }
#pragma unroll
        for (int ot = 0; ot < 6; ++ot)
#pragma unroll
            for (int q = 0; q < 2; ++q) y[ot] = mfma32(*reinterpret_cast<const bf16x8*>(sl + a2 + ot * 32 * W2_ROW + q * 32), hb[q], y[ot]);
    }
    IR_STAMP(3);
    // ---- the new rows -> residual stream through the staging slices (and the bf16 copy for the RSTB's conv when this is its last block)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int ts = 0; ts < 3; ++ts)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 3 * hf + ts;
                const f32x4 nx = f32x4{y[t][4 * i], y[t][4 * i + 1], y[t][4 * i + 2], y[t][4 * i + 3]};
                *reinterpret_cast<f32x4*>(frag_addr(ts, i)) = nx;
                if (ln_next) xr[t][i] = nx;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ts = 0; ts < 3; ++ts)
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
                const int tk = 8 * pc + ptok;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + ts * 4096 + pc * 1024 + lane * 16);
                const int ch = 32 * (3 * hf + ts) + 4 * (pslot ^ ((tk >> 1) & 7));
                if (active) {
                    *reinterpret_cast<f32x4*>(x_out + gtk[pc] * CP + ch) = v;
                    if (out2 && !ln_next) *reinterpret_cast<uint2*>(out2 + gtk[pc] * CP + ch) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    IR_STAMP(4);
    if (!ln_next || !out2) return;
    float t1 = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) t1 += (xr[t][i][0] + xr[t][i][1]) + (xr[t][i][2] + xr[t][i][3]);
    t1 += __shfl_xor(t1, 32);
    const float mean2 = t1 / (float)C;
    float t2 = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = xr[t][i][e] - mean2;
                t2 += (32 * t + 8 * i + 4 * h + e < C) ? d * d : 0.f;
            }
    t2 += __shfl_xor(t2, 32);
    const float rstd2 = rsqrtf(t2 / (float)C + eps);
    if constexpr (!QKV) {
        if (active) {
            bf16_t* orow = out2 + qtok * CP + 4 * h;
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c0 = 32 * t + 8 * i + 4 * h;
                    const f32x4 gg = *reinterpret_cast<const f32x4*>(vs + NEXT_OFF + c0), bb = *reinterpret_cast<const f32x4*>(vs + NEXT_OFF + CP + c0);
                    const f32x4 v = (xr[t][i] - mean2) * rstd2 * gg + bb;
                    *reinterpret_cast<uint2*>(orow + 32 * t + 8 * i) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                }
        }
    } else {
        bf16x8 xq[12];
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            uint32_t wv4[4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int t = s >> 1, i = 2 * (s & 1) + q, c0 = 32 * t + 8 * i + 4 * h;
                const f32x4 gg = *reinterpret_cast<const f32x4*>(vs + NEXT_OFF + c0), bb = *reinterpret_cast<const f32x4*>(vs + NEXT_OFF + CP + c0);
                const f32x4 v = (xr[t][i] - mean2) * rstd2 * gg + bb;
                wv4[2 * q] = pack2bf(v[0], v[1]);
                wv4[2 * q + 1] = pack2bf(v[2], v[3]);
            }
            xq[s] = __builtin_bit_cast(bf16x8, make_uint4(wv4[0], wv4[1], wv4[2], wv4[3]));
        }
        const int LDQ = 32 * NQ;
        for (int st = 0; st < NQ / 2; ++st) {
            const int jt = NJ + st;
            f32x4 bv[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[u][i] = *reinterpret_cast<const f32x4*>(bq + 32 * (2 * st + u) + 8 * i + 4 * h);
            wait_dma();
            __syncthreads();
            if (jt + 1 < NS) stage(jt + 1, (jt + 1) & 1);
            const unsigned char* sl = smem + (jt & 1) * SLOT;
            unsigned char* sb = stg + (st % 3) * 4096;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                for (int s = 0; s < 12; ++s) acc = mfma32(*reinterpret_cast<const bf16x8*>(sl + u * W1_TILE + a1 + s * 32), xq[s], acc);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<uint2*>(sb + r * 128 + (((4 * u + i) ^ ((r >> 1) & 7)) << 4) + 8 * h) =
                        make_uint2(pack2bf(acc[4 * i] + bv[u][i][0], acc[4 * i + 1] + bv[u][i][1]), pack2bf(acc[4 * i + 2] + bv[u][i][2], acc[4 * i + 3] + bv[u][i][3]));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
                const int tk = 8 * pc + ptok;
                const uint4 v = *reinterpret_cast<const uint4*>(sb + pc * 1024 + lane * 16);
                if (active) *reinterpret_cast<uint4*>(out2 + gtk[pc] * LDQ + 64 * st + 8 * (pslot ^ ((tk >> 1) & 7))) = v;
            }
        }
    }
#ifdef IR_SWIN_STAMPS
    IR_STAMP(5);
    if ((blockIdx.x == 3 || blockIdx.x == 131) && (tid == 0 || tid == 448))
        printf("STAMP wg %d wave %d attn %llu xload %llu ln+mlp %llu store %llu tail %llu (x10ns)\n", (int)blockIdx.x, wu, stp[1] - stp[0], stp[2] - stp[1], stp[3] - stp[2],
               stp[4] - stp[3], stp[5] - stp[4]);
#endif
}

int ir_launch_swin_block(const bf16_t* qkv, const float* x_in, float* x_out, bf16_t* out2, const void* proj_t, const float* proj_b, const float* biasT,
                         int B, int H, int W, int shift, float scale, const void* w_tiles, const float* vec, int C, int hid_p, float eps, hipStream_t s,
                         const float* next_g, const float* next_b, const void* qkv_tiles, const float* qkv_b, int qkv_n) {
    if ((H & 7) || (W & 7) || shift < 0 || shift >= 8 || B <= 0) return -2;
    if ((long)B * H * W * 576 * 2 >= (1L << 32)) return -4;   // 32-bit byte offsets into the qkv tensor (row stride 576 elements)
    if (C <= 0 || C > swf::CP || hid_p <= 0 || (hid_p & 31) || hid_p > 512) return -2;
    if ((next_g != nullptr) != (next_b != nullptr) || (next_g && (!out2 || (C & 3)))) return -2;
    if (qkv_tiles && (!next_g || !qkv_b || qkv_n <= 0 || (qkv_n & 63))) return -2;
    const long n_waves = 2L * B * (H >> 3) * (W >> 3);
    const long blocks = (n_waves + 7) / 8;
    if (blocks > 0x7fffffffL) return -4;
    const int NJ = hid_p / 32, NQ = qkv_tiles ? qkv_n / 32 : 0;
    const size_t lds = swf::LDS_TOTAL;
    const float sl2 = scale * 1.44269504088896340736f;
    const bf16_t* pt = reinterpret_cast<const bf16_t*>(proj_t);
    const unsigned char* wt = reinterpret_cast<const unsigned char*>(w_tiles);
    const unsigned char* wq = reinterpret_cast<const unsigned char*>(qkv_tiles);
    auto go = [&](auto kern) {
        static bool attr_set = false;   // one per instantiation (the lambda's body is instantiated per kernel type)
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, s, qkv, x_in, x_out, out2, pt, proj_b, biasT, H, W, shift, sl2, n_waves, wt, vec, C, NJ, eps,
                           next_g, next_b, wq, qkv_b, NQ);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    };
    if (qkv_tiles) return go(swin_block_kernel<true, true>);
    if (next_g) return go(swin_block_kernel<true, false>);
    return go(swin_block_kernel<false, false>);
}
