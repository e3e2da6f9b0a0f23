// Attention kernels for gfx950 (bf16 MFMA 32x32x16, fp32 softmax state).
//
//  flash_attn_kernel<DQK,DV> : multi-head softmax(Q K^T * scale + key_bias) V, online softmax, never materialises
//      the score matrix. Used for the PixArt-DiT self-attention (16 heads x 72; reference
//      diffusion/model/nets/PixArt_blocks.py:123-158 == diffusers Attention/SDPA) and its cross-attention to the
//      300 caption tokens with an additive per-key bias (PixArt_blocks.py:43-58; diffusers semantics: the raw
//      0/1 mask is ADDED).
//      Layout trick (CDNA4): scores are computed transposed, S^T = K Q^T, so a query lives on a lane and its keys
//      in that lane's accumulator registers -> row max / row sum are in-register plus one cross-half shuffle, and
//      the S^T accumulators are directly the B operand of the second product O^T = V^T P^T. K rows are fetched
//      with bits 2/3 of the row index swapped so that the V^T operand is a plain 16-byte LDS read.
//  transpose_v_kernel        : V[token][head*D+d] -> V^T[batch][head][DV][Tpad] (zero padded), HBM-bound helper.
//  swin_window_attn_kernel   : SwinIR (S)W-MSA, 8x8 windows, 6 heads x 30 (padded 32), relative-position bias and
//      shifted-window mask generated in-kernel (reference diffusion/model/swinir.py:125-156,227-248,259-283).
//  softmax_rows_kernel       : fp32 row softmax -> bf16, for the materialised VAE mid-block attention
//      (reference ldm/modules/diffusionmodules/model.py:181-205).
#include "common.h"
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

// LDS-DMA: 16 bytes per lane from a per-lane global address to LDS base + lane*16 (wrapped in a device function: used directly
// inside a template kernel the builtin suppresses the host stub)
typedef __attribute__((address_space(3))) void* attn_lds_ptr_t;
IR_DEVINL void attn_glds16(const void* g, attn_lds_ptr_t l) { __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0); }

IR_DEVINL int swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// Knock-out experiments of tools/attn_stamp.hip (diagnostic builds only; results are wrong by design): 1 no per-tile barrier,
// 2 no K/V global loads + LDS stores, 3 no LDS fragment reads, 4 no MFMAs, 5 no softmax arithmetic.
#ifndef IR_KO_ATTN
#define IR_KO_ATTN 0
#endif
// Diagnostic build only (-DIR_STAMPS_ATTN, tools/attn_stamp.hip): per-wave cycle sums (s_memtime) of the segments of
// flash_attn_pp_kernel: [0] vector segment, [1] wait at the barrier after it, [2] matrix segment, [3] wait at the barrier after it.
#ifdef IR_STAMPS_ATTN
__device__ unsigned long long g_attn_stamps[4096 * 8 * 4];
#define IR_ATT_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define IR_ATT_ACC(k, a, b) st_acc[k] += (b) - (a)
#else
#define IR_ATT_T(v) do { } while (0)
#define IR_ATT_ACC(k, a, b) do { } while (0)
#endif

template <int D, bool GENERAL>
__global__ __launch_bounds__(256, 2) void flash_attn_kernel(AttnParams p) {
    constexpr int DQK = (D + 15) & ~15;  // head dim padded for the QK^T k-steps (16 per MFMA)
    constexpr int DV = (D + 31) & ~31;   // head dim padded for the 32-row O^T tiles
    constexpr int QS = DQK + 8;          // LDS row stride of Q/K tiles (bf16 elements); odd number of 16-B chunks
    constexpr int VS = 64 + 8;           // LDS row stride of V^T tiles
    constexpr int KCH = DQK / 8;         // 16-B chunks per LDS Q/K row
    constexpr int RCH = D / 8;           // real 16-B chunks per row in HBM
    constexpr int NKS = DQK / 16;
    constexpr int NDT = DV / 32;
    constexpr int OS = DV + 8;           // O staging row stride
    constexpr int K_ITEMS = (64 * RCH + 255) / 256;
    constexpr int V_ITEMS = (DV * 8 + 255) / 256;
    constexpr int Q_ITEMS = (128 * RCH + 255) / 256;
    constexpr int KV_ELEMS = 2 * 64 * QS + 2 * DV * VS;
    constexpr int O_ELEMS = 128 * OS;
    constexpr int TAIL = KV_ELEMS > O_ELEMS ? KV_ELEMS : O_ELEMS;
    constexpr float RESCALE_THR = 8.0f;  // log2 units: skip the O/l rescale while the running max grows by less than 2^8
    // When the head dim leaves padded rows in the O^T tiles (72 -> 96), transpose_v_kernel fills row D of V^T with ones, so the
    // second product accumulates the softmax denominator in O^T row D for free (no per-element row-sum adds on the VALU).
    constexpr bool ONES = DV > D;
    constexpr int L_DT = D / 32, L_G = ((D % 32) & 3) + 4 * ((D % 32) >> 3), L_H = ((D % 32) >> 2) & 1;
    __shared__ __attribute__((aligned(16))) bf16_t smem[128 * QS + TAIL];
    __shared__ float kbs[2][64];         // per-tile additive key bias (log2 domain; -inf beyond Tk)
    bf16_t* Qs = smem;
    bf16_t* Ks = smem + 128 * QS;
    bf16_t* Vs = Ks + 2 * 64 * QS;
    bf16_t* Os = Ks;  // reused after the KV loop

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 128, head = blockIdx.y, b = blockIdx.z;
    // launched behind flash_attn_pp_kernel as its fallback: nothing to do unless that kernel flagged a query it could not handle
    if (p.ovf_flag && *reinterpret_cast<volatile const int*>(p.ovf_flag) == 0) return;
    // ... and, when the flagging kernel kept a map of its 256-query workgroups, nothing unless THIS block's queries belong to one that overflowed
    if (p.ovf_flag && p.ovf_map > 0 &&
        reinterpret_cast<volatile const int*>(p.ovf_flag)[1 + ((long)b * p.Hh + head) * ((p.Tq + 255) >> 8) + (q0 >> 8)] == 0) return;

    const bf16_t* qp = p.q + (long)b * p.q_bs + (long)head * p.q_hs;
    const bf16_t* kp = p.k + (long)b * p.k_bs + (long)head * p.k_hs;
    const bf16_t* vtp = p.vt + (long)b * p.vt_bs + (long)head * DV * p.Tk_pad;
    const float* kb = p.key_bias ? p.key_bias + (long)b * p.kb_bs : nullptr;
    constexpr bool general = GENERAL;  // additive key bias and/or ragged last tile -> bias path through LDS (separate instantiation)

    // Every global load below is UNCONDITIONAL on a clamped, always-valid address and never feeds a select: hipcc turns
    // "load, then zero if out of range" into a branch around the load plus a vmcnt(0) per element, which serialises
    // the whole prefetch. Out-of-range keys are handled by the -inf key bias, out-of-range queries are never stored,
    // and the zero padding of the head dim lives in LDS columns that are written once here.
    if (KCH > RCH) {
        for (int c = tid; c < 256 * (KCH - RCH); c += 256) {
            const int row = c / (KCH - RCH), ch = RCH + c % (KCH - RCH);
            *reinterpret_cast<uint4*>(&smem[row * QS + ch * 8]) = make_uint4(0, 0, 0, 0);  // Qs rows 0..127, Ks rows 0..127
        }
    }
#pragma unroll
    for (int i = 0; i < Q_ITEMS; ++i) {
        const int c = tid + i * 256;
        const int row = min(c / RCH, 127), ch = c % RCH;
        uint4 v = *reinterpret_cast<const uint4*>(qp + (long)min(q0 + row, p.Tq - 1) * p.q_rs + ch * 8);
        // Q is pre-multiplied by scale * log2(e): the QK^T accumulators then are exp2-domain scores and need no per-element fma
        v.x = pack2bf(bflo(v.x) * p.scale_log2, bfhi(v.x) * p.scale_log2);
        v.y = pack2bf(bflo(v.y) * p.scale_log2, bfhi(v.y) * p.scale_log2);
        v.z = pack2bf(bflo(v.z) * p.scale_log2, bfhi(v.z) * p.scale_log2);
        v.w = pack2bf(bflo(v.w) * p.scale_log2, bfhi(v.w) * p.scale_log2);
        if (c < 128 * RCH) *reinterpret_cast<uint4*>(&Qs[row * QS + ch * 8]) = v;
    }
    uint4 kreg[K_ITEMS], vreg[V_ITEMS];  // initialised: an uninitialised array written under a condition stays in scratch
#pragma unroll
    for (int i = 0; i < K_ITEMS; ++i) kreg[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < V_ITEMS; ++i) vreg[i] = make_uint4(0, 0, 0, 0);
    float kbreg = 0.f;
    auto load_kv = [&](int t) {
        const int key0 = t * 64;
#pragma unroll
        for (int i = 0; i < K_ITEMS; ++i) {
            const int c = tid + i * 256;
            const int row = min(c / RCH, 63), ch = c % RCH;
            kreg[i] = *reinterpret_cast<const uint4*>(kp + (long)min(key0 + row, p.Tk - 1) * p.k_rs + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < V_ITEMS; ++i) {
            const int c = tid + i * 256;
            const int row = min(c >> 3, DV - 1), ch = c & 7;
            vreg[i] = *reinterpret_cast<const uint4*>(vtp + (long)row * p.Tk_pad + key0 + ch * 8);
        }
        if (general) {
            const int key = key0 + (tid & 63);
            float v = kb ? kb[min(key, p.Tk - 1)] * 1.44269504088896340736f : 0.f;
            kbreg = key < p.Tk ? v : -INFINITY;
        }
    };
    auto store_kv = [&](int buf) {
#pragma unroll
        for (int i = 0; i < K_ITEMS; ++i) {
            const int c = tid + i * 256;
            const int row = c / RCH, ch = c % RCH;
            if (c < 64 * RCH) *reinterpret_cast<uint4*>(&Ks[(buf * 64 + row) * QS + ch * 8]) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < V_ITEMS; ++i) {
            const int c = tid + i * 256;
            if (c < DV * 8) *reinterpret_cast<uint4*>(&Vs[(buf * DV + (c >> 3)) * VS + (c & 7) * 8]) = vreg[i];
        }
        if (general && tid < 64) kbs[buf][tid] = kbreg;
    };
    load_kv(0);
    store_kv(0);
    __syncthreads();

    bf16x8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(&Qs[(wid * 32 + r) * QS + ks * 16 + h * 8]);

    f32x16 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int g = 0; g < 16; ++g) o[dt][g] = 0.f;
    // Online softmax state. m_i is the running max (exp2 domain) the probabilities are taken against; -m_i is kept broadcast in
    // 16 registers and fed to the first QK^T MFMA as its C operand, so the accumulators come out as (score - m_i) directly.
    float m_i = 0.f, l_i = 0.f;
    f32x16 negm;
#pragma unroll
    for (int g = 0; g < 16; ++g) negm[g] = 0.f;
    asm volatile("s_nop 7" : "+v"(negm));   // pinned here: asm MFMAs are invisible to hipcc's hazard recogniser, which otherwise materialises these zeros directly in front of the MFMA that reads them as its C operand (tools/mfma_hazard_scan.py)
    const int NT = (p.Tk + 63) >> 6;
    const int kr = swap23(r);
    int cur = 0;
    for (int t = 0; t < NT; ++t) {
        const bool more = t + 1 < NT;
        if (more && IR_KO_ATTN != 2) load_kv(t + 1);
        // ---- S^T = K Q^T - m_i (two 32-key tiles); this lane's keys for (kt, g): t*64 + kt*32 + 16*(g>>3) + 8*h + (g&7)
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                bf16x8 a = qf[ks];
                if (IR_KO_ATTN != 3) a = *reinterpret_cast<const bf16x8*>(&Ks[(cur * 64 + kt * 32 + kr) * QS + ks * 16 + h * 8]);
                if (IR_KO_ATTN != 4) s[kt] = mfma32(a, qf[ks], ks == 0 ? negm : s[kt]);
                else if (ks == 0) s[kt] = negm;
            }
        }
        if (general) {  // additive key bias (log2 domain; -inf beyond Tk)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g8 = 0; g8 < 2; ++g8) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(&kbs[cur][kt * 32 + 16 * g8 + 8 * h]);
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(&kbs[cur][kt * 32 + 16 * g8 + 8 * h + 4]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) s[kt][g8 * 8 + e] += e < 4 ? b0[e & 3] : b1[e & 3];
                }
        }
        float mx = -INFINITY;  // max of this tile RELATIVE to m_i
        if (IR_KO_ATTN != 5) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g = 0; g < 16; ++g) mx = fmaxf(mx, s[kt][g]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
        }
        // deferred rescale: the running max moves only when some query's scores exceed it by more than RESCALE_THR (and on the
        // first tile); until then probabilities are taken against the slightly stale max and stay <= 2^THR (exact in fp32 / bf16
        // range). The decision precedes this tile's exponentials, so everything at the old max is scaled exactly once.
        if (t == 0 || __any(mx > RESCALE_THR)) {
            const float delta = t == 0 ? (mx > -INFINITY ? mx : 0.f) : fmaxf(mx, 0.f);
            const float alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);  // O and l are still zero on the first tile
            m_i += delta;
            l_i *= alpha;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int g = 0; g < 16; ++g) o[dt][g] *= alpha;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g = 0; g < 16; ++g) s[kt][g] -= delta;
#pragma unroll
            for (int g = 0; g < 16; ++g) negm[g] = -m_i;
        }
        // exponentials and bf16 P^T fragments (B operand: registers 8*s2 .. 8*s2+7 of tile kt)
        float rs = 0.f;
        bf16x8 pb[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = IR_KO_ATTN == 5 ? s[kt][s2 * 8 + e] : __builtin_amdgcn_exp2f(s[kt][s2 * 8 + e]);
                    if (!ONES) rs += pv;
                    pb[kt][s2][e] = (__bf16)pv;
                }
        if (!ONES) {
            rs += __shfl_xor(rs, 32);
            l_i += rs;
        }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 a = pb[kt][s2];
                    if (IR_KO_ATTN != 3) a = *reinterpret_cast<const bf16x8*>(&Vs[(cur * DV + dt * 32 + r) * VS + kt * 32 + s2 * 16 + h * 8]);
                    if (IR_KO_ATTN != 4) o[dt] = mfma32(a, pb[kt][s2], o[dt]);
                    else o[dt][0] += __builtin_bit_cast(float, (uint32_t)__builtin_bit_cast(uint16_t, pb[kt][s2][0]) + (uint32_t)__builtin_bit_cast(uint16_t, a[1]));
                }
        if (more && IR_KO_ATTN != 2) store_kv(cur ^ 1);
        if (IR_KO_ATTN != 1) __syncthreads();
        cur ^= 1;
    }
    // ---- finalise: O^T[d][q] / l -> LDS [q][d] -> 16-byte row stores
    if (ONES) l_i = __shfl(o[L_DT][L_G], r + 32 * L_H);  // O^T row D of this lane's query
    const float inv = 1.0f / l_i;
    bf16_t* ow = Os + wid * 32 * OS;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            uint2 w = make_uint2(pack2bf(o[dt][4 * gg] * inv, o[dt][4 * gg + 1] * inv),
                                 pack2bf(o[dt][4 * gg + 2] * inv, o[dt][4 * gg + 3] * inv));
            *reinterpret_cast<uint2*>(&ow[r * OS + dt * 32 + 8 * gg + 4 * h]) = w;
        }
    __syncthreads();
    bf16_t* op = p.o + (long)b * p.o_bs + (long)head * p.o_hs;
    for (int c = lane; c < 32 * RCH; c += 64) {
        int row = c / RCH, ch = c - row * RCH;
        int q = q0 + wid * 32 + row;
        if (q < p.Tq) *reinterpret_cast<uint4*>(op + (long)q * p.o_rs + ch * 8) = *reinterpret_cast<const uint4*>(&ow[row * OS + ch * 8]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Ping-pong flash attention for the DiT self-attention (16 heads x 72, T % 64 == 0, no key bias): the same mathematics and
// register layouts as flash_attn_kernel (S^T = K Q^T - m, O^T = V^T P^T, keys swap23-permuted, softmax denominator from the
// ones row of V^T), restructured so that the matrix pipe never waits for the softmax.
// Two independent 4-wave workgroups per CU run in lockstep (QK^T together, exponentials together): measured with knock-outs,
// their MFMA time and their VALU + data-movement time simply ADD (1.54 ms = 0.72 + 0.82). Here ONE 512-thread workgroup puts two
// waves on every SIMD, wave w and w+4, each owning 32 queries, and forces complementary phases with workgroup barriers:
//     segment:   sigma-1      sigma          sigma+1       sigma+2
//     waves 0-3  softmax(t)   PV(t)+QK(t+1)  softmax(t+1)  PV(t+1)+QK(t+2)
//     waves 4-7  PV+QK(t-1)   softmax(t)     PV(t)+QK(t+1) softmax(t+1)
// so on every SIMD one wave issues MFMAs (22 per segment, 704 cycles) while its partner does the VALU work. K/V tiles are shared
// by all 256 queries (half the L2->LDS traffic per query) and arrive by LDS-DMA into a ring of three stages, stage u = {V(u),
// K(u+1)} in slot u % 3. Every wave issues its 2-3 one-KB pieces per tile at the start of its vector segment, which has slack
// (about 500 cycles of softmax against 700+ of MFMAs): waves 0-3 those of stage t+1, waves 4-7 (a segment later) those of stage
// t+2, each into a slot whose previous stage both halves finished reading at an earlier barrier, and each wave waits for its
// own pieces (vmcnt) at the end of its following matrix segment; the barrier there publishes them a full segment before the
// first read of that stage.
// K rows are stored UNPADDED (144 B = 9 chunks, odd -> conflict-free): the fifth k-step reads elements 72..79 from the next
// row's first chunk, finite values that meet the zero padding of the Q fragment. V^T rows are 128 B with the XOR swizzle of
// flash_attn_d512_kernel. Q fragments come straight from HBM in the B-operand layout, pre-multiplied by scale * log2(e).
// The fragment reads of a matrix segment are inline-asm ds_read_b128 with hand-counted s_waitcnt lgkmcnt(N): left to hipcc, every
// fifth MFMA waited for lgkmcnt(0), i.e. for a read issued one instruction earlier (73 instead of 32 cycles per MFMA).
// MFMAs with the register file of each operand chosen by hand (flash_attn_d512_kernel): hipcc keeps the 128 registers of Q^T
// fragments and the 128 of O^T in arch VGPRs "spilled" to AGPRs and moves ~400 of them per tile (v_accvgpr_read/write); the matrix
// instruction can read B and accumulate C/D in AGPRs directly. Being asm, these are invisible to hipcc's hazard recogniser: the
// caller separates them from VALU writes of their inputs and from VALU reads of their results with mfma_fence().
IR_DEVINL void mfma32_b_agpr(f32x16& c, bf16x8 a, bf16x8 b) {  // c (VGPR) += a (VGPR) x b (AGPR)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
}
IR_DEVINL void mfma32_v(f32x16& c, bf16x8 a, bf16x8 b) {  // c (VGPR) += a (VGPR) x b (VGPR)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
IR_DEVINL void mfma32_c_agpr(f32x16& c, bf16x8 a, bf16x8 b) {  // c (AGPR) += a (VGPR) x b (VGPR)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// >= 18 wait states (16-pass MFMA result -> VALU read) and the few a VALU result needs before an MFMA reads it. The registers
// concerned are in/out operands: without that data dependence hipcc is free to schedule the VALU instructions that consume (or
// produce) them on the wrong side of the nops.
// LONG = MFMA results about to be read by the VALU (18 wait states for a 16-pass MFMA; 24 given); short = VALU results about to
// be read by an MFMA (4 given).
template <bool LONG>
IR_DEVINL void mfma_fence_v(f32x16& a, f32x16& b) {
    if (LONG) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a), "+v"(b));
    else asm volatile("s_nop 3" : "+v"(a), "+v"(b));
}
template <bool LONG, int N>
IR_DEVINL void mfma_fence_acc(f32x16 (&acc)[N], bf16x8 (&pb)[4]) {
    static_assert(N % 8 == 0, "operand lists below are written out for 8 accumulator tiles at a time");
#pragma unroll
    for (int b = 0; b < N; b += 8) {
        if (LONG && b == 0)
            asm volatile("s_nop 15\n\ts_nop 7"
                         : "+a"(acc[b]), "+a"(acc[b + 1]), "+a"(acc[b + 2]), "+a"(acc[b + 3]), "+a"(acc[b + 4]), "+a"(acc[b + 5]), "+a"(acc[b + 6]),
                           "+a"(acc[b + 7]), "+v"(pb[0]), "+v"(pb[1]), "+v"(pb[2]), "+v"(pb[3]));
        else
            asm volatile("s_nop 3"
                         : "+a"(acc[b]), "+a"(acc[b + 1]), "+a"(acc[b + 2]), "+a"(acc[b + 3]), "+a"(acc[b + 4]), "+a"(acc[b + 5]), "+a"(acc[b + 6]),
                           "+a"(acc[b + 7]), "+v"(pb[0]), "+v"(pb[1]), "+v"(pb[2]), "+v"(pb[3]));
    }
}


template <int D>
__global__ __launch_bounds__(512, 1) void flash_attn_pp_kernel(AttnParams p) {
    constexpr int DQK = (D + 15) & ~15, DV = (D + 31) & ~31;
    static_assert(DV > D, "needs a spare V^T row (ones) for the softmax denominator");
    constexpr int RCH = D / 8;           // 16-byte chunks per K row, in HBM and in LDS
    static_assert(RCH % 2 == 1, "unpadded K rows are conflict-free only with an odd chunk count");
    constexpr int KROW = RCH * 16;       // bytes per K row in LDS
    constexpr int NKS = DQK / 16, NDT = DV / 32;
    constexpr int K_BYTES = 64 * KROW, V_BYTES = DV * 128;
    static_assert(K_BYTES % 1024 == 0 && V_BYTES % 1024 == 0, "tiles are whole LDS-DMA instructions");
    // DMA instructions per tile. V^T rows beyond D (the ones row) are never transferred: their O^T rows are never stored, and a row
    // of O^T depends on no other row of V^T, so whatever those LDS rows hold is harmless.
    constexpr int K_Q = K_BYTES / 1024, V_Q = (D + 1 + 7) / 8;
    constexpr int NSLOT = 3;
    constexpr int V_OFF = NSLOT * K_BYTES;
    constexpr int OS = DV + 8;           // O staging row stride (elements)
    constexpr int LDS_MAIN = NSLOT * (K_BYTES + V_BYTES), LDS_O = 8 * 32 * OS * 2;
    constexpr int LDS_BYTES = LDS_MAIN > LDS_O ? LDS_MAIN : LDS_O;
    constexpr int L_DT = D / 32, L_G = ((D % 32) & 3) + 4 * ((D % 32) >> 3), L_H = ((D % 32) >> 2) & 1;
    constexpr float RESCALE_THR = 8.0f;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];  // K[0..2] | V^T[0..2]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int grp = wu >> 2;  // grp 1 (waves 4-7) runs one segment behind grp 0
    const int r = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 256 + wid * 32, head = blockIdx.y, b = blockIdx.z;
    const bf16_t* qp = p.q + (long)b * p.q_bs + (long)head * p.q_hs;
    const bf16_t* kp = p.k + (long)b * p.k_bs + (long)head * p.k_hs;
    const bf16_t* vtp = p.vt + (long)b * p.vt_bs + (long)head * DV * p.Tk_pad;
    const int NT = p.Tk >> 6;

    // LDS-DMA pieces (1 KB each): a tile pair is K_Q + V_Q pieces, piece idx = wave + 8*k belongs to this wave (at most NPC). Per
    // piece the lane's tile-0 source address is computed once; issuing it for a tile is one 64-bit multiply-add plus the DMA.
    constexpr int NPC = (K_Q + V_Q + 7) / 8;
    const bf16_t* pc_src[NPC];
    int pc_stride[NPC], pc_dst[NPC];
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
        const int idx = wu + 8 * k;
        if (idx < K_Q) {
            const int ci = 64 * idx + lane, row = ci / RCH, ch = ci - row * RCH;
            pc_src[k] = kp + (long)row * p.k_rs + ch * 8;
            pc_stride[k] = 64 * p.k_rs;
            pc_dst[k] = idx * 1024;
        } else {
            const int j = min(idx - K_Q, V_Q - 1);
            const int d = 8 * j + (lane >> 3), c = (lane & 7) ^ ((d >> 1) & 7);  // LDS slot (lane & 7) of row d holds chunk c
            pc_src[k] = vtp + (long)d * p.Tk_pad + c * 8;
            pc_stride[k] = 64;
            pc_dst[k] = V_OFF + j * 1024;
        }
    }
    auto issue_k = [&](int tile, int slot) {  // this wave's K pieces of `tile` into K slot `slot`
#pragma unroll
        for (int k = 0; k < NPC; ++k)
            if (wu + 8 * k < K_Q) attn_glds16(pc_src[k] + (long)tile * pc_stride[k], (attn_lds_ptr_t)(smem + slot * K_BYTES + pc_dst[k]));
    };
    auto issue_v = [&](int tile, int slot) {
#pragma unroll
        for (int k = 0; k < NPC; ++k)
            if (wu + 8 * k >= K_Q && wu + 8 * k < K_Q + V_Q)
                attn_glds16(pc_src[k] + (long)tile * pc_stride[k], (attn_lds_ptr_t)(smem + slot * V_BYTES + pc_dst[k]));
    };
    auto issue_stage = [&](int u) {  // stage u = {V(u), K(u+1)} -> ring slot u % 3
        const int slot = u % NSLOT;
        if (u < NT) issue_v(u, slot);
        if (u + 1 < NT) issue_k(u + 1, slot);
    };
    // prologue: K(0) -> K slot 2 (free until stage 2) and stage 0; in the loop, vector segment t of waves 0-3 issues their pieces
    // of stage t+1 and that of waves 4-7 (one segment later) their pieces of stage t+2, so waves 4-7 add stage 1 here
    issue_k(0, 2);
    issue_stage(0);
    if (grp == 1) issue_stage(1);

    // Q^T fragments (B operand: lane = query, 8 consecutive d per k-step half), scaled; d >= D is zero
    bf16x8 qf[NKS];
    {
        const bf16_t* qrow = qp + (long)min(q0 + r, p.Tq - 1) * p.q_rs;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d0 = ks * 16 + h * 8;
            const uint4 v = *reinterpret_cast<const uint4*>(qrow + (d0 < D ? d0 : 0));  // unconditional load on a valid address
            const float sc = d0 < D ? p.scale_log2 : 0.f;
            uint4 w;
            w.x = pack2bf(bflo(v.x) * sc, bfhi(v.x) * sc);
            w.y = pack2bf(bflo(v.y) * sc, bfhi(v.y) * sc);
            w.z = pack2bf(bflo(v.z) * sc, bfhi(v.z) * sc);
            w.w = pack2bf(bflo(v.w) * sc, bfhi(v.w) * sc);
            qf[ks] = __builtin_bit_cast(bf16x8, w);
        }
    }
    f32x16 o[NDT], sacc[2], negm;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int g = 0; g < 16; ++g) o[dt][g] = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) negm[g] = 0.f;
    asm volatile("s_nop 7" : "+v"(negm));   // pinned here: asm MFMAs are invisible to hipcc's hazard recogniser, which otherwise materialises these zeros directly in front of the MFMA that reads them as its C operand (tools/mfma_hazard_scan.py)
    float m_i = 0.f;
    bf16x8 pb[2][2];

    // LDS fragment addresses (bytes): one base per operand plus compile-time immediates
    const uint32_t lds0 = (uint32_t)(uintptr_t)(attn_lds_ptr_t)smem;
    const uint32_t k_addr = lds0 + swap23(r) * KROW + h * 16;                       // + slot*K_BYTES + kt*32*KROW + ks*32
    const int vsw = (r >> 1) & 7;                                                    // swizzle of rows d = dt*32 + r
    uint32_t v_addr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v_addr[j] = lds0 + V_OFF + r * 128 + ((((2 * j) | h) ^ vsw) << 4);  // + slot*V_BYTES + dt*4096

    // One matrix segment: PV(t) (NDT*4 MFMAs) and, if QK, S^T(t+1) (2*NKS MFMAs); fragment reads run LA MFMAs ahead of their use.
    // The first LA reads are issued by matrix_prefetch() at the END of the preceding vector segment, above the barrier (their
    // tiles were published at least a segment earlier), so the segment opens with an MFMA instead of an LDS round trip; the S^T
    // accumulators are preset to -m_i there as well, so every MFMA here accumulates in place.
    constexpr int NPV = NDT * 4;
    constexpr int LA = 4;   // only ONE wave per SIMD is in its matrix segment: nobody else hides its LDS latency
    constexpr int NB = LA + 3;  // fragment buffers (registers stay allocated two MFMAs beyond their use, see below)
    bf16x8 fr[NB];
    uint32_t ka = 0, va[4] = {0, 0, 0, 0};
    auto frag_read = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (IR_KO_ATTN == 12) { fr[j % NB] = qf[j % NKS]; return; }
        if constexpr (j < NPV) fr[j % NB] = lds_read16<(j >> 2) * 4096>(va[j & 3]);
        else fr[j % NB] = lds_read16<((j - NPV) / NKS) * 32 * KROW + ((j - NPV) % NKS) * 32>(ka);
    };
    auto matrix_prefetch = [&](auto pv_tag, auto qk_tag, int vslot, int kslot) {
        constexpr bool PV = decltype(pv_tag)::value, QK = decltype(qk_tag)::value;
        constexpr int J0 = PV ? 0 : NPV, J1 = QK ? NPV + 2 * NKS : NPV;
        ka = k_addr + kslot * K_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) va[j] = v_addr[j] + vslot * V_BYTES;
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) { (frag_read(std::integral_constant<int, J0 + I>{}), ...); }(std::make_integer_sequence<int, (J1 - J0 < LA ? J1 - J0 : LA)>{});
        __builtin_amdgcn_sched_barrier(0);
    };
    auto matrix_segment = [&](auto pv_tag, auto qk_tag) {
        constexpr bool PV = decltype(pv_tag)::value, QK = decltype(qk_tag)::value;
        constexpr int J0 = PV ? 0 : NPV, J1 = QK ? NPV + 2 * NKS : NPV;
        auto step = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + LA < J1) frag_read(std::integral_constant<int, j + LA>{});
            wait_lds<(J1 - 1 - j < LA ? J1 - 1 - j : LA)>();  // reads issued after the one MFMA j consumes
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j < NPV) o[j >> 2] = mfma32(fr[j % NB], pb[(j & 3) >> 1][j & 1], o[j >> 2]);
            else if constexpr ((j - NPV) % NKS == 0)  // first k-step of a score tile: D = A x B + (-m), D and C in different registers
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(sacc[(j - NPV) / NKS]) : "v"(fr[j % NB]), "v"(qf[0]), "v"(negm));
            else sacc[(j - NPV) / NKS] = mfma32(fr[j % NB], qf[(j - NPV) % NKS], sacc[(j - NPV) / NKS]);
            // keep the fragment of MFMA j-2 allocated until here: the register allocator otherwise hands its registers to the very
            // next read, which then waits (with everything behind it) until that MFMA has finished reading them
            if constexpr (j - 2 >= J0) asm volatile("" ::"v"(fr[(j - 2) % NB]));
            __builtin_amdgcn_sched_barrier(0);
        };
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) { (step(std::integral_constant<int, J0 + I>{}), ...); }(std::make_integer_sequence<int, J1 - J0>{});
    };
    // Softmax of the tile whose (score - m) sits in sacc. The reference m is FIXED after the first tile (its maximum): the
    // probabilities of later tiles are exp2(score - m) whatever they are, up to 2^OVF_LIMIT - bf16 and fp32 have the exponent range,
    // and relative precision does not depend on the reference. Without a rescale path the accumulators have one definition per tile
    // (hipcc had kept the two paths' O^T in different registers: 24 64-bit moves per tile on the common path) and -m can stay
    // broadcast in 16 registers that the first QK^T MFMA of a tile takes as its C operand (no per-tile preset of the S^T
    // accumulators). A query whose scores outgrow 2^OVF_LIMIT raises a flag; the launcher runs the rescaling 4-wave kernel after
    // this one, which returns at once unless the flag is set.
    constexpr float OVF_LIMIT = 64.0f;
    bool ovf = false;
    auto softmax_segment = [&](int t) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 16; ++g) mx = fmaxf(mx, sacc[kt][g]);
        mx = xhalf_max(mx);
        if (t == 0) {  // the accumulators of tile 0 were computed against m = 0
            m_i = mx;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g = 0; g < 16; ++g) sacc[kt][g] -= mx;
#pragma unroll
            for (int g = 0; g < 16; ++g) negm[g] = -mx;
        } else {
            ovf |= mx > OVF_LIMIT;
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 8; ++e) pb[kt][s2][e] = (__bf16)__builtin_amdgcn_exp2f(sacc[kt][s2 * 8 + e]);
    };

    wait_dma();
    __syncthreads();                   // K(0) and the prologue stages landed
    if (grp == 1) __syncthreads();     // waves 4-7 start one segment late
    if (grp == 1) __builtin_amdgcn_s_setprio(1);  // static priority for the younger half, which otherwise loses the VALU arbitration
                                                  // on every segment (CDNA4 notes; +1.3 % measured here)
    matrix_prefetch(std::false_type{}, std::true_type{}, 0, 2);
    matrix_segment(std::false_type{}, std::true_type{});  // S^T(0) from K slot 2
    __syncthreads();
#ifdef IR_STAMPS_ATTN
    unsigned long long st_acc[4] = {0, 0, 0, 0};
#endif
    int slot = 0;  // t % 3
    for (int t = 0; t < NT; ++t) {
        // ---- vector segment (the SIMD partner is in its matrix segment): DMA issue, softmax, first fragment reads
        IR_ATT_T(ta);
        if (IR_KO_ATTN != 11) issue_stage(t + 1 + grp);
        if (IR_KO_ATTN != 13) softmax_segment(t);
        if (t + 1 < NT) matrix_prefetch(std::true_type{}, std::true_type{}, slot, slot);
        else matrix_prefetch(std::true_type{}, std::false_type{}, slot, slot);
        IR_ATT_T(tb);
        __syncthreads();
        IR_ATT_T(tc);
        // ---- matrix segment: O^T += V^T(t) P^T(t); S^T(t+1) = K(t+1) Q^T - m
        if (t + 1 < NT) matrix_segment(std::true_type{}, std::true_type{});
        else matrix_segment(std::true_type{}, std::false_type{});
        IR_ATT_T(td);
        wait_dma();  // this wave's pieces, issued at the start of its vector segment: the barrier publishes them before any wave's
                     // matrix_prefetch (which runs ABOVE the next barrier) can touch their stage
        __syncthreads();
        IR_ATT_T(te);
        IR_ATT_ACC(0, ta, tb); IR_ATT_ACC(1, tb, tc); IR_ATT_ACC(2, tc, td); IR_ATT_ACC(3, td, te);
        slot = slot == NSLOT - 1 ? 0 : slot + 1;
    }
#ifdef IR_STAMPS_ATTN
    {
        const int bl = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (lane == 0 && bl < 4096)
            for (int k = 0; k < 4; ++k) g_attn_stamps[(bl * 8 + wid) * 4 + k] = st_acc[k];
    }
#endif
    if (grp == 0) __syncthreads();     // pairs the late start of waves 4-7; afterwards nobody reads the K/V ring any more
    wait_dma();

    if (__any(ovf) && lane == 0) atomicOr(p.ovf_flag, 1);
    // ---- finalise: O^T[d][q] / l -> LDS [q][d] -> 16-byte row stores
    const float l_i = __shfl(o[L_DT][L_G], r + 32 * L_H);  // O^T row D (ones row of V^T) of this lane's query
    const float inv = 1.0f / l_i;
    bf16_t* ow = reinterpret_cast<bf16_t*>(smem) + wid * 32 * OS;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            uint2 w = make_uint2(pack2bf(o[dt][4 * gg] * inv, o[dt][4 * gg + 1] * inv),
                                 pack2bf(o[dt][4 * gg + 2] * inv, o[dt][4 * gg + 3] * inv));
            *reinterpret_cast<uint2*>(&ow[r * OS + dt * 32 + 8 * gg + 4 * h]) = w;
        }
    __syncthreads();
    bf16_t* op = p.o + (long)b * p.o_bs + (long)head * p.o_hs;
    for (int c = lane; c < 32 * RCH; c += 64) {
        int row = c / RCH, ch = c - row * RCH;
        int q = q0 + row;
        if (q < p.Tq) *reinterpret_cast<uint4*>(op + (long)q * p.o_rs + ch * 8) = *reinterpret_cast<const uint4*>(&ow[row * OS + ch * 8]);
    }
}

int ir_launch_flash_attn_fallback(const AttnParams& p, hipStream_t s) {
    if (p.D != 72 || !p.ovf_flag || p.key_bias || (p.Tk & 63)) return -2;
    hipLaunchKernelGGL((flash_attn_kernel<72, false>), dim3((p.Tq + 127) / 128, p.Hh, p.B), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

bool ir_flash_attn_is_pp2(const AttnParams& p) {
    static const bool no_pp = getenv("IR_NO_PINGPONG") != nullptr, pp1 = getenv("IR_ATTN_PP1") != nullptr;
    const bool general = p.key_bias != nullptr || (p.Tk & 63);
    return p.D == 72 && !general && p.Tq >= 256 && p.ovf_flag && !no_pp && !pp1 && !g_ir_plain_kernels;
}

int ir_launch_flash_attn(const AttnParams& p, hipStream_t s) {
    if (p.Tq <= 0 || p.Tk <= 0 || p.B <= 0 || p.Hh <= 0) return -2;
    if ((p.D & 7) || (p.q_rs & 7) || (p.k_rs & 7) || (p.o_rs & 7) || (p.q_hs & 7) || (p.k_hs & 7) || (p.o_hs & 7) ||
        (p.q_bs & 7) || (p.k_bs & 7) || (p.o_bs & 7) || (p.vt_bs & 7))
        return -3;
    if ((p.Tk_pad & 63) || p.Tk_pad < ((p.Tk + 63) & ~63)) return -4;
    if (p.scale_log2 <= 0.f) return -6;
    dim3 grid((p.Tq + 127) / 128, p.Hh, p.B);
    const bool general = p.key_bias != nullptr || (p.Tk & 63);
    static const bool no_pp = getenv("IR_NO_PINGPONG") != nullptr;  // experiment knob
    if (p.D == 72 && !general && p.Tq >= 256 && p.ovf_flag && !no_pp && !g_ir_plain_kernels) {  // the DiT self-attention: ping-pong kernel, 256 queries per workgroup
        static const bool pp1e = getenv("IR_ATTN_PP1") != nullptr, no_map = getenv("IR_ATTN_NO_OVF_MAP") != nullptr;   // (no_map: experiment knob - the whole-launch fallback of rounds 2-5)
        const long need = (long)p.B * p.Hh * ((p.Tq + 255) / 256);
        AttnParams pm = p;
        if (pp1e || no_map || p.ovf_map < need) pm.ovf_map = 0;   // the two-waves-per-SIMD kernel keeps no map; a short scratch means the caller did not size one
        if (ir_launch_zero_f32(reinterpret_cast<float*>(p.ovf_flag), 1 + (pm.ovf_map ? need : 0), s)) return -1;
        static const bool pp1 = getenv("IR_ATTN_PP1") != nullptr;  // experiment knob: the two-waves-per-SIMD ping-pong kernel
        if (!pp1) {
            if (ir_launch_flash_attn_pp2(pm, s)) return -1;
        } else
        hipLaunchKernelGGL((flash_attn_pp_kernel<72>), dim3((p.Tq + 255) / 256, p.Hh, p.B), dim3(512), 0, s, pm);
        hipLaunchKernelGGL((flash_attn_kernel<72, false>), grid, dim3(256), 0, s, pm);  // fallback: returns at once unless flagged (per 256-query workgroup with the map)
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    AttnParams q = p;
    if (ir_flash_attn_x72_takes(p)) return ir_launch_flash_attn_x72(p, s);   // the DiT cross-attention: persistent kernel (handles its own overflow)
    q.ovf_flag = nullptr;  // stand-alone use of the 4-wave kernel below
#define IR_FA(DD)                                                                                     \
    do {                                                                                              \
        if (general) hipLaunchKernelGGL((flash_attn_kernel<DD, true>), grid, dim3(256), 0, s, q);     \
        else hipLaunchKernelGGL((flash_attn_kernel<DD, false>), grid, dim3(256), 0, s, q);            \
    } while (0)
    if (p.D == 72) IR_FA(72);
    else if (p.D == 32) IR_FA(32);
    else if (p.D == 64) IR_FA(64);
    else return -5;
#undef IR_FA
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Single-head attention with head dim 512 (the VAE mid-block, reference ldm/modules/diffusionmodules/model.py:181-205),
// flash style. One wave per SIMD with the whole 512-register budget: each wave owns 32 queries and keeps its Q^T fragments (32
// k-steps, pre-multiplied by scale * log2 e) in 128 VGPRs. K (64 keys x 512, 64 KB) and V^T (256 x 64 keys, 32 KB) tiles are
// single-buffered in LDS and filled by LDS-DMA: K(t+1) lands while softmax + PV of tile t run, V(t+1) while QK^T of tile t+1
// runs, two barriers per tile. Same transposed-score formulation as flash_attn_kernel (S^T = K Q^T - m via the accumulator
// preset, O^T = V^T P^T, keys swap23-permuted).
// q, k: [T][512] bf16 rows (row stride rs); vt: [512][vt_rs] bf16; o: [T][512]. T % 64 == 0.
// Register budget: O^T for all 512 output dims (256 accumulators) plus the Q fragments (128) plus S/P exceeds what hipcc
// allocates without spilling, so the output dims are split over blockIdx.y (DSPLIT = 2): each block recomputes S^T and owns
// 256 output dims (128 accumulators). That costs 1.5x the MFMA work of an ideal kernel but keeps everything in registers
// (DSPLIT = 1 with O^T in all 256 AGPRs and Q^T in VGPRs was tried again with the asm MFMAs below: hipcc still spills, 17 ms
// against 11.5 ms at T = 65536).
// The MFMAs are inline asm with the register file of every operand chosen by hand: Q^T fragments in AGPRs (read directly as the B
// operand), O^T accumulators in AGPRs, everything the VALU touches in VGPRs. Left to hipcc, Q^T and O^T sat in VGPRs "spilled" to
// AGPRs with about 400 v_accvgpr moves per tile.
// With one wave per SIMD nothing else hides what the wave does between MFMAs, so the two matrix loops are written as pinned
// instruction streams (like flash_attn_pp_kernel): fragment reads LA MFMAs ahead with counted lgkmcnt waits, and the 24 LDS-DMA
// pieces a wave issues per tile (16 of K, 8 of V^T) are interleaved one by one behind MFMAs instead of in two bursts after the
// barriers, where they cost the wave about 2000 of its 7000 cycles per tile with the matrix pipe idle.
template <int DSPLIT>
__global__ __launch_bounds__(256, 1) void flash_attn_d512_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                  const bf16_t* __restrict__ vt, bf16_t* __restrict__ o, int T, int rs,
                                                                  int o_rs, long vt_rs, float scale_log2, const int* __restrict__ only_if) {
    // launched behind flash_attn_d512_v2_kernel as its fallback: nothing to do unless that kernel flagged a query it could not handle
    if (only_if && *reinterpret_cast<volatile const int*>(only_if) == 0) return;
    constexpr int D = 512, NKS = D / 16, DVB = D / DSPLIT, NDT = DVB / 32;
    constexpr bool QF_AGPR = DSPLIT > 1;  // DSPLIT 1: the 256 AGPRs are all O^T, the Q^T fragments stay in VGPRs
    constexpr int KROW = 1024 + 16;  // K rows are one DMA instruction each, so they can be padded: (key*65 + c) % 16 is conflict-free
    constexpr int KBYTES = 64 * KROW;
    constexpr int NKP = 16, NVP = DVB / 32;  // DMA pieces per wave and tile: K rows wu + 4i, V^T row groups wu + 4i
    constexpr float RESCALE_THR = 8.0f;
    __shared__ __attribute__((aligned(256))) unsigned char smem[KBYTES + DVB * 128];  // K tile | V^T tile (reused for O at the end)
    const int dv0 = blockIdx.y * DVB;  // first output dim owned by this block
    vt += (long)dv0 * vt_rs;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 128;

    // DMA sources. K piece i of tile t: row t*64 + wu + 4i, 16 bytes per lane. V^T rows are 128 B (8 rows per DMA instruction,
    // unpaddable): chunk c of row d sits at slot c ^ ((d >> 1) & 7). For the rows d = 8*qi + (lane>>3) of instruction qi = wu + 4i:
    // (d >> 1) & 7 = ((lane >> 4) & 3) | ((qi & 1) << 2), and qi & 1 = wu & 1.
    const bf16_t* k_lane = k + (long)wu * rs + lane * 8;
    const bf16_t* v_lane = vt + (long)(wu * 8 + (lane >> 3)) * vt_rs + (((lane & 7) ^ ((lane >> 4) & 3) ^ (((wu * 8) >> 1) & 4)) << 3);
    const long k_step = 4L * rs, v_step = 32L * vt_rs;  // element strides between a wave's consecutive pieces
    auto k_piece = [&](int tile, int i) {
        attn_glds16(k_lane + (long)tile * 64 * rs + i * k_step, (attn_lds_ptr_t)(smem + (wu + 4 * i) * KROW));
    };
    auto v_piece = [&](int tile, int i) {
        attn_glds16(v_lane + i * v_step + tile * 64, (attn_lds_ptr_t)(smem + KBYTES + (wu + 4 * i) * 1024));
    };
#pragma unroll
    for (int i = 0; i < NKP; ++i) k_piece(0, i);
#pragma unroll
    for (int i = 0; i < NVP; ++i) v_piece(0, i);
    // Q^T fragments straight from HBM in the MFMA B-operand layout (lane = query, 8 consecutive d per k-step half), scaled
    bf16x8 qf[NKS];
    {
        const bf16_t* qrow = q + (long)min(q0 + wid * 32 + r, T - 1) * rs + h * 8;
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
    f32x16 acc[NDT], sacc[2];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[dt][g] = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) sacc[0][g] = sacc[1][g] = 0.f;
    float m_i = 0.f, l_i = 0.f;
    bf16x8 pb[4];
    // LDS fragment addresses (bytes): ONE base per operand plus compile-time immediates
    const uint32_t lds0 = (uint32_t)(uintptr_t)(attn_lds_ptr_t)smem;
    const uint32_t k_addr = lds0 + swap23(r) * KROW + h * 16;                       // + kt*32*KROW + ks*32
    const int vsw = (r >> 1) & 7;                                                    // swizzle of rows d = dt*32 + r
    uint32_t v_addr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v_addr[j] = lds0 + KBYTES + r * 128 + ((((2 * j) | h) ^ vsw) << 4);  // + dt*4096 ; j = kt*2 + s2

    constexpr int LA = 4, NB = LA + 3;  // reads in flight ahead of their MFMA; fragment buffers (see flash_attn_pp_kernel)
    bf16x8 fr[NB];
    // S^T(t) = K Q^T - m: 64 MFMAs; the V^T pieces of tile vtile (if >= 0) ride behind MFMAs 1, 9, 17, ...
    auto qk_loop = [&](int vtile) {
        auto read = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            fr[j % NB] = lds_read16<(j / NKS) * 32 * KROW + (j % NKS) * 32>(k_addr);
        };
        auto step = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + LA < 2 * NKS) read(std::integral_constant<int, j + LA>{});
            wait_lds<(2 * NKS - 1 - j < LA ? 2 * NKS - 1 - j : LA)>();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (QF_AGPR) mfma32_b_agpr(sacc[j / NKS], fr[j % NB], qf[j % NKS]);
            else mfma32_v(sacc[j / NKS], fr[j % NB], qf[j % NKS]);
            if constexpr (j >= 2) asm volatile("" ::"v"(fr[(j - 2) % NB]));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j % 8 == 1 && j / 8 < NVP) {
                if (vtile >= 0) v_piece(vtile, j / 8);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) { (read(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, LA>{});
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) { (step(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, 2 * NKS>{});
    };
    // O^T += V^T(t) P^T: 32 MFMAs; the K pieces of tile ktile (if >= 0) ride behind MFMAs 1, 3, 5, ...
    auto pv_loop = [&](int ktile) {
        auto read = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            fr[j % NB] = lds_read16<(j >> 2) * 4096>(v_addr[j & 3]);
        };
        auto step = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + LA < NDT * 4) read(std::integral_constant<int, j + LA>{});
            wait_lds<(NDT * 4 - 1 - j < LA ? NDT * 4 - 1 - j : LA)>();
            __builtin_amdgcn_sched_barrier(0);
            mfma32_c_agpr(acc[j >> 2], fr[j % NB], pb[j & 3]);
            if constexpr (j >= 2) asm volatile("" ::"v"(fr[(j - 2) % NB]));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j % 2 == 1 && j / 2 < NKP) {
                if (ktile >= 0) k_piece(ktile, j / 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) { (read(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, LA>{});
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) { (step(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, NDT * 4>{});
    };

    const int NT = T >> 6;
    wait_dma();
    __syncthreads();  // K(0), V(0) landed
    for (int t = 0; t < NT; ++t) {
        // ---- S^T = K Q^T - m (accumulators preset to -m). The V^T tile is free since the barrier that ended PV(t-1), so the pieces
        // of V^T(t) ride behind this loop's MFMAs (V^T(0) came with the prologue); they are needed only after the next barrier.
        mfma_fence_v<false>(sacc[0], sacc[1]);  // the accumulator preset (VALU) precedes
        qk_loop(t >= 1 ? t : -1);
        mfma_fence_v<true>(sacc[0], sacc[1]);   // the softmax (VALU) follows
        wait_dma();
        __syncthreads();                       // every wave is done with the K tile, and V^T(t) has landed
        // ---- online softmax (exp2 domain, deferred rescale); the scores are relative to the running max already
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 16; ++g) mx = fmaxf(mx, sacc[kt][g]);
        mx = xhalf_max(mx);
        if (t == 0 || __any(mx > RESCALE_THR)) {
            const float delta = t == 0 ? mx : fmaxf(mx, 0.f);
            const float alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);  // O and l are still zero on the first tile
            m_i += delta;
            l_i *= alpha;
            // the O^T accumulators live in AGPRs; pinning them there at both ends of this rare path keeps hipcc from hoisting the
            // 128 v_accvgpr_reads of the rescale above the branch, into every tile
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) asm volatile("" : "+a"(acc[dt]));
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[dt][g] *= alpha;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) asm volatile("" : "+a"(acc[dt]));
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g = 0; g < 16; ++g) sacc[kt][g] -= delta;
        }
        float rsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(sacc[kt][s2 * 8 + e]);
                    rsum += pv;
                    pb[kt * 2 + s2][e] = (__bf16)pv;
                }
        rsum += __shfl_xor(rsum, 32);
        l_i += rsum;
#pragma unroll
        for (int g = 0; g < 16; ++g) sacc[0][g] = sacc[1][g] = -m_i;  // the next S^T accumulates onto -m in place
        // ---- O^T += V^T P^T, with the K(t+1) pieces behind its MFMAs (every wave passed the barrier above: the K tile is free)
        mfma_fence_acc<false>(acc, pb);  // pb (VALU) precedes
        pv_loop(t + 1 < NT ? t + 1 : -1);
        // The O^T accumulators are next touched by the following tile's PV MFMAs (in place: no hazard) or, rarely, by its rescale,
        // which comes after a whole QK^T loop; the epilogue below fences for itself.
        wait_dma();
        __syncthreads();                       // every wave is done with the V^T tile, and K(t+1) has landed
    }
    // ---- finalise: O^T / l -> LDS [q][DVB] bf16 (16 KB per wave) -> 16-byte row stores
    mfma_fence_acc<true>(acc, pb);
    const float inv = 1.0f / l_i;
    constexpr int OROW = DVB * 2;  // bytes per staged row
    unsigned char* ow = smem + wid * 32 * OROW;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            uint2 w = make_uint2(pack2bf(acc[dt][4 * gg] * inv, acc[dt][4 * gg + 1] * inv),
                                 pack2bf(acc[dt][4 * gg + 2] * inv, acc[dt][4 * gg + 3] * inv));
            // row r (query), 8-byte chunk index (dt*32 + 8gg + 4h)/4; XOR with the row spreads the 32 rows over banks
            const int c8 = (dt * 8 + 2 * gg + h) ^ (r & 31);
            *reinterpret_cast<uint2*>(ow + r * OROW + c8 * 8) = w;
        }
    __syncthreads();
    constexpr int OCH = DVB / 8;  // 16-byte chunks per staged row
    for (int c = lane; c < 32 * OCH; c += 64) {
        const int row = c / OCH, ch = c % OCH;
        const int qq = q0 + wid * 32 + row;
        // undo the 8-byte XOR swizzle: 16-byte chunk ch = 8-byte chunks 2ch, 2ch+1 -> stored at (2ch)^row, (2ch+1)^row
        const uint2 lo = *reinterpret_cast<const uint2*>(ow + row * OROW + (((2 * ch) ^ (row & 31)) * 8));
        const uint2 hi = *reinterpret_cast<const uint2*>(ow + row * OROW + (((2 * ch + 1) ^ (row & 31)) * 8));
        if (qq < T) *reinterpret_cast<uint4*>(o + (long)qq * o_rs + dv0 + ch * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}

int ir_launch_flash_attn_d512(const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* o, int T, int rs, int o_rs, long vt_rs,
                              float scale, hipStream_t s, const int* only_if) {
    if (T <= 0 || (T & 63) || (rs & 7) || (o_rs & 7) || (vt_rs & 7) || vt_rs < T) return -2;
    hipLaunchKernelGGL(flash_attn_d512_kernel<2>, dim3((T + 127) / 128, 2), dim3(256), 0, s, q, k, vt, o, T, rs, o_rs, vt_rs,
                       scale * 1.44269504088896340736f, only_if);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// V[b][t][head*D + d] (row stride v_rs) -> Vt[b][head][DV][Tpad]; rows d >= D and columns t >= T are zero.
// grid = (Tpad/64, Hh, B), block 256. LDS tile [64 tokens][DV].
__global__ __launch_bounds__(256) void transpose_v_kernel(const bf16_t* __restrict__ v, bf16_t* __restrict__ vt, long v_bs, int v_rs,
                                                          int v_hs, int T, int Tpad, int D, int DV, int Hh, const int* __restrict__ only_if) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][128 + 8];
    if (only_if && *reinterpret_cast<volatile const int*>(only_if) == 0) return;  // fallback preparation: see ir_launch_flash_attn_d512_v2
    const int t0 = blockIdx.x * 64, head = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const bf16_t* src = v + (long)b * v_bs + (long)head * v_hs;
    if (!(D & 7) && !(v_rs & 7) && !(v_hs & 7) && !(v_bs & 7) && !(reinterpret_cast<uintptr_t>(v) & 15) && !(reinterpret_cast<uintptr_t>(vt) & 15)) {
        // vector form (every shape of the network): 16-byte row pieces in, 16 bytes = 8 tokens of one d out (the scalar form below moved
        // 2 bytes per instruction: 42 us per DiT layer)
        const int dch = D >> 3;
        for (int c = tid; c < 64 * dch; c += 256) {
            const int tok = c / dch, ch = c - tok * dch;
            uint4 x = make_uint4(0, 0, 0, 0);
            if (t0 + tok < T) x = *reinterpret_cast<const uint4*>(src + (long)(t0 + tok) * v_rs + ch * 8);
            *reinterpret_cast<uint4*>(&tile[tok][ch * 8]) = x;
        }
        for (int c = tid; c < 64 * (DV - D); c += 256) {
            const int tok = c / (DV - D), j = c - tok * (DV - D);
            tile[tok][D + j] = (j == 0 && t0 + tok < T) ? 0x3f80 : 0;   // row D: ones over the real keys (the softmax denominator), zeros above
        }
        __syncthreads();
        bf16_t* dstv = vt + ((long)b * Hh + head) * (long)DV * Tpad + t0;
        for (int c = tid; c < DV * 8; c += 256) {
            const int d = c >> 3, tc = c & 7;
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[tc * 8 + 2 * e][d] | ((uint32_t)tile[tc * 8 + 2 * e + 1][d] << 16);
            *reinterpret_cast<uint4*>(dstv + (long)d * Tpad + tc * 8) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        return;
    }
    for (int i = tid; i < 64 * DV; i += 256) {
        int tok = i / DV, d = i - tok * DV;
        bf16_t x = 0;
        if (d < D && t0 + tok < T) x = src[(long)(t0 + tok) * v_rs + d];
        if (d == D && t0 + tok < T) x = 0x3f80;  // row D (if DV > D): ones over the real keys -> flash_attn_kernel's softmax denominator
        tile[tok][d] = x;
    }
    __syncthreads();
    bf16_t* dst = vt + ((long)b * Hh + head) * (long)DV * Tpad + t0;
    for (int i = tid; i < DV * 64; i += 256) {
        int d = i >> 6, tok = i & 63;
        dst[(long)d * Tpad + tok] = tile[tok][d];
    }
}

int ir_launch_transpose_v(const bf16_t* v, bf16_t* vt, long v_bs, int v_rs, int v_hs, int B, int Hh, int T, int Tpad, int D,
                          int DV, hipStream_t s, const int* only_if) {
    if (DV > 128 || D > DV || (Tpad & 63) || Tpad < T) return -2;
    hipLaunchKernelGGL(transpose_v_kernel, dim3(Tpad / 64, Hh, B), dim3(256), 0, s, v, vt, v_bs, v_rs, v_hs, T, Tpad, D, DV, Hh, only_if);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------- SwinIR window attention
// qkv: [B][H*W][ld] bf16 with columns [q: heads*32 | k: heads*32 | v: heads*32] (head dim 30 zero-padded to 32).
// out: [B][H*W][ldo] bf16, columns head*32 + d. One block (2 waves) per (window, head); wave = 32-query half.
// biasT: [heads][64 keys][64 queries] fp32, relative-position bias already multiplied by log2(e).
__global__ __launch_bounds__(128) void swin_window_attn_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                               const float* __restrict__ biasT, int H, int W, int heads, int ld,
                                                               int ldo, int shift, float scale_log2) {
    __shared__ __attribute__((aligned(16))) bf16_t Qs[64][40];
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64][40];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[32][72];
    __shared__ int tok_idx[64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nwx = W >> 3, nwy = H >> 3;
    int wlin = blockIdx.x;
    const int head = wlin % heads; wlin /= heads;
    const int wx = wlin % nwx; wlin /= nwx;
    const int wy = wlin % nwy;
    const int b = wlin / nwy;
    if (tid < 64) {
        int iy = tid >> 3, ix = tid & 7;
        int y = wy * 8 + iy + shift, x = wx * 8 + ix + shift;  // shifted-frame (Y,X) -> source pixel (Y+s)%H
        if (y >= H) y -= H;
        if (x >= W) x -= W;
        tok_idx[tid] = y * W + x;
    }
    __syncthreads();
    const bf16_t* base = qkv + (long)b * H * W * ld;
    // 64 tokens x 3 tensors x 4 chunks (16 B) = 768 chunks
    for (int c = tid; c < 768; c += 128) {
        int tok = c / 12, rem = c - tok * 12;
        int which = rem >> 2, ch = rem & 3;
        uint4 v = *reinterpret_cast<const uint4*>(base + (long)tok_idx[tok] * ld + which * heads * 32 + head * 32 + ch * 8);
        if (which == 0)
            *reinterpret_cast<uint4*>(&Qs[tok][ch * 8]) = v;
        else if (which == 1)
            *reinterpret_cast<uint4*>(&Ks[tok][ch * 8]) = v;
        else {
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Vt[ch * 8 + 2 * e][tok] = (bf16_t)(w[e] & 0xffff);
                Vt[ch * 8 + 2 * e + 1][tok] = (bf16_t)(w[e] >> 16);
            }
        }
    }
    __syncthreads();
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(&Qs[wid * 32 + r][ks * 16 + h * 8]);
    const int kr = swap23(r);
    f32x16 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int g = 0; g < 16; ++g) s[kt][g] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a = *reinterpret_cast<const bf16x8*>(&Ks[kt * 32 + kr][ks * 16 + h * 8]);
            s[kt] = mfma32(a, qf[ks], s[kt]);
        }
    }
    // bias + shifted-window mask (regions in the shifted frame, swinir.py:227-248)
    const int qi = wid * 32 + r;
    auto region = [&](int idx) {
        int Y = wy * 8 + (idx >> 3), X = wx * 8 + (idx & 7);
        int rh = Y < H - 8 ? 0 : (Y < H - shift ? 1 : 2);
        int rw = X < W - 8 ? 0 : (X < W - shift ? 1 : 2);
        return rh * 3 + rw;
    };
    const int rq = shift ? region(qi) : 0;
    const float* bt = biasT + (long)head * 4096 + qi;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int key = kt * 32 + 16 * (g >> 3) + 8 * h + (g & 7);
            float v = s[kt][g] * scale_log2 + bt[key * 64];
            if (shift && region(key) != rq) v += -100.0f * 1.44269504088896340736f;
            s[kt][g] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float rs = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float pv = __builtin_amdgcn_exp2f(s[kt][g] - mx);
            s[kt][g] = pv;
            rs += pv;
        }
    rs += __shfl_xor(rs, 32);
    f32x16 o;
#pragma unroll
    for (int g = 0; g < 16; ++g) o[g] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 pb;
#pragma unroll
            for (int e = 0; e < 8; ++e) pb[e] = (__bf16)s[kt][s2 * 8 + e];
            bf16x8 a = *reinterpret_cast<const bf16x8*>(&Vt[r][kt * 32 + s2 * 16 + h * 8]);
            o = mfma32(a, pb, o);
        }
    const float inv = 1.0f / rs;
    bf16_t* op = out + ((long)b * H * W + tok_idx[qi]) * ldo + head * 32;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
        uint2 w = make_uint2(pack2bf(o[4 * gg] * inv, o[4 * gg + 1] * inv), pack2bf(o[4 * gg + 2] * inv, o[4 * gg + 3] * inv));
        *reinterpret_cast<uint2*>(op + 8 * gg + 4 * h) = w;
    }
}

int ir_launch_swin_attn(const bf16_t* qkv, bf16_t* out, const float* biasT, int B, int H, int W, int heads, int ld, int ldo,
                        int shift, float scale, hipStream_t s) {
    if ((H & 7) || (W & 7) || (ld & 7) || (ldo & 3) || shift < 0 || shift >= 8) return -2;
    if (ld < 3 * heads * 32 || ldo < heads * 32) return -3;
    const long blocks = (long)B * (H >> 3) * (W >> 3) * heads;
    if (blocks <= 0 || blocks > 0x7fffffffL) return -4;
    hipLaunchKernelGGL(swin_window_attn_kernel, dim3((unsigned)blocks), dim3(128), 0, s, qkv, out, biasT, H, W, heads, ld, ldo,
                       shift, scale * 1.44269504088896340736f);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------- fp32 row softmax -> bf16 (one block per row)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int cols, long ldx,
                                                           long ldy) {
    __shared__ float red[8];
    const long row = blockIdx.x;
    const float* xr = x + row * ldx;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float L2E = 1.44269504088896340736f;
    float m = -INFINITY, l = 0.f;  // online max / sum over this thread's elements (float4 steps)
    for (int c = tid * 4; c < cols; c += 1024) {
        f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        float vm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        float mn = fmaxf(m, vm);
        l = l * __builtin_amdgcn_exp2f((m - mn) * L2E);
#pragma unroll
        for (int e = 0; e < 4; ++e) l += __builtin_amdgcn_exp2f((v[e] - mn) * L2E);
        m = mn;
    }
    float wm = wave_max(m);
    // threads (or whole waves) without elements hold m = -inf, l = 0: keep them at exactly 0 (avoid -inf - -inf)
    l = (m == -INFINITY) ? 0.f : l * __builtin_amdgcn_exp2f((m - wm) * L2E);
    l = wave_sum(l);
    if (lane == 0) { red[wid] = wm; red[4 + wid] = l; }
    __syncthreads();
    float gm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float gl = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) gl += (red[i] == -INFINITY) ? 0.f : red[4 + i] * __builtin_amdgcn_exp2f((red[i] - gm) * L2E);
    const float inv = 1.0f / gl;
    bf16_t* yr = y + row * ldy;
    for (int c = tid * 4; c < cols; c += 1024) {
        f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        float pz[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) pz[e] = __builtin_amdgcn_exp2f((v[e] - gm) * L2E) * inv;
        *reinterpret_cast<uint2*>(yr + c) = make_uint2(pack2bf(pz[0], pz[1]), pack2bf(pz[2], pz[3]));
    }
}

int ir_launch_softmax_rows(const float* x, bf16_t* y, long rows, int cols, long ldx, long ldy, hipStream_t s) {
    if ((cols & 3) || (ldx & 3) || (ldy & 3) || rows <= 0 || rows > 0x7fffffffL) return -2;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, x, y, cols, ldx, ldy);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
