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
constexpr int VEC_OFF = 2 * SLOT;            // fp32 vectors behind the ring: g[192] b[192] b1[hid_p <= 512] b2[192]
constexpr int STG_OFF = VEC_OFF + 4608;      // wave-private staging of token rows: 8 waves x 3 slices x 4 KB
constexpr int STG_WAVE = 3 * 4096;
constexpr int LDS_TOTAL = STG_OFF + 8 * STG_WAVE;   // 160 256 B
}  // namespace swf

// x, out: [T][192] fp32 (out may alias x); out2 (optional): [T][192] bf16 copy of the result; w: [NJ] tiles {W1 tile | W2 tile};
// vec: g[192] b[192] b1[32*NJ] b2[192] fp32; C: real channels (LayerNorm extent); NJ: hidden units / 32.
// Token rows travel between HBM and the accumulator layout through a wave-private LDS staging area: a lane needs 16-byte groups of
// ITS token (768-byte rows: 64 different cache lines per load instruction if read directly - measured 57 us per launch, most of it
// the memory path), so each 32-channel slice t of the wave's 32 tokens (one 128-byte line per token) is brought in by LDS-DMA
// pieces of 8 tokens x 128 B, chunk-swizzled on the source side, and read back as the lane's groups 32t + 8i + 4h; results leave
// the same way (LDS image -> coalesced 16-byte stores). Three slices (12 KB per wave) at a time.
__global__ __launch_bounds__(512, 1) void swin_mlp_kernel(const float* __restrict__ x, float* __restrict__ out, bf16_t* __restrict__ out2,
                                                          const unsigned char* __restrict__ w, const float* __restrict__ vec, long T, int C,
                                                          int NJ, float eps) {
    using namespace swf;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int r = lane & 31, h = lane >> 5;
    const long tok0 = (long)blockIdx.x * 256 + wu * 32;                  // first token of this wave

    // weight ring: piece q of a step's 28 KB goes to wave q % 8 (3.5 pieces per wave: waves 0-3 take four)
    auto stage = [&](int jt, int slot) {
        const unsigned char* src = w + (long)jt * SLOT + lane * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (wu + 8 * i < PIECES) sw_glds16(src + (wu + 8 * i) * 1024, (sw_lds_t)(smem + slot * SLOT + (wu + 8 * i) * 1024));
    };
    stage(0, 0);
    float* vs = reinterpret_cast<float*>(smem + VEC_OFF);
    const int nvec = 3 * CP + 32 * NJ;
    for (int i = tid; i < nvec; i += 512) vs[i] = vec[i];

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

    f32x16 y[6];
#pragma unroll
    for (int ot = 0; ot < 6; ++ot)
#pragma unroll
        for (int g = 0; g < 16; ++g) y[ot][g] = 0.f;
    const int a1 = r * W1_ROW + h * 16;                 // + s * 32
    const int a2 = W1_TILE + r * W2_ROW + h * 16;       // + ot * 32 * W2_ROW + s2 * 32
    // (Waves w and w + 4 share a SIMD and run this loop in lockstep - both in their MFMAs, then both in their GELU arithmetic: the
    // wave spends about half its cycles waiting (SQ_WAIT_ANY). Running waves 4-7 one product late needs a third ring slot, which the
    // 160 KB do not have next to the row staging; tried with a second barrier per step instead: spills and no gain.)
    for (int jt = 0; jt < NJ; ++jt) {
        wait_dma();
        __syncthreads();           // tile jt has landed; every wave is done with tile jt - 1
        if (jt + 1 < NJ) stage(jt + 1, (jt + 1) & 1);
        if (jt == NJ - 1) load_half(x, 0);   // the residual rows again (first half), under the last step's arithmetic
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
    // ---- + b2 + x -> residual stream, three slices at a time through the staging area: the lane replaces its groups of the staged
    // input rows by the results (accumulator layout: register 4i + e of tile t is channel 32t + 8i + 4h + e), then the image leaves
    // with coalesced 16-byte stores (8 lanes per 128-byte line), the bf16 copy as 8-byte stores
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        if (hf == 1) load_half(x, 3);
        wait_dma();
#pragma unroll
        for (int ts = 0; ts < 3; ++ts)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 3 * hf + ts;
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(vs + 2 * CP + 32 * NJ + 32 * t + 8 * i + 4 * h);
                f32x4* pa = reinterpret_cast<f32x4*>(frag_addr(ts, i));
                *pa = f32x4{y[t][4 * i], y[t][4 * i + 1], y[t][4 * i + 2], y[t][4 * i + 3]} + b2 + *pa;
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
                    if (out2) *reinterpret_cast<uint2*>(out2 + gt * CP + ch) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image has been read before the next half's DMA lands on it
    }
}

int ir_launch_swin_mlp(const float* x, float* out, bf16_t* out2, const void* w_tiles, const float* vec, long T, int C, int hid_p, float eps,
                       hipStream_t s) {
    if (T <= 0 || C <= 0 || C > swf::CP || hid_p <= 0 || (hid_p & 31) || hid_p > 512) return -2;
    const int NJ = hid_p / 32;
    const size_t lds = swf::LDS_TOTAL;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(swin_mlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        attr_set = true;
    }
    hipLaunchKernelGGL(swin_mlp_kernel, dim3((unsigned)((T + 255) / 256)), dim3(512), lds, s, x, out, out2, reinterpret_cast<const unsigned char*>(w_tiles),
                       vec, T, C, NJ, eps);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
