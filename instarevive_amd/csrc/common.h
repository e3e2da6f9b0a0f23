// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the InstaRevive one-step path.
// Wave = 64 lanes; MFMA = v_mfma_f32_32x32x16_bf16 (A/B: 8 bf16 per lane, C/D: 16 f32 per lane).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits; storage type for every activation / weight tensor
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define IR_DEVINL __device__ __forceinline__

// f32 -> bf16 round-to-nearest-even through the hardware convert (keeps NaN a NaN).
IR_DEVINL bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
IR_DEVINL float bf2f(bf16_t b) {
    uint32_t u = ((uint32_t)b) << 16;
    return __builtin_bit_cast(float, u);
}
IR_DEVINL uint32_t pack2bf(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
// The same as ONE instruction. hipcc (ROCm 7.2) turns pack2bf into two v_cvt_pk_bf16_f32 (each with a zero second operand), a shift and
// an or. ONLY for operands produced by ordinary full-rate VALU instructions at least one instruction earlier: an asm statement is invisible
// to hipcc's hazard recogniser, so on operands fresh from an MFMA it reads them before the required wait states (replacing pack2bf
// globally by this broke every GEMM epilogue: 16 dB), and directly behind a transcendental (v_exp_f32 / v_rcp_f32 need one wait state
// before a VALU reads their result) it packs garbage into one half (seen in a cross-attention experiment: pack2bf_valu(exp2(a), exp2(b))).
// The attention kernels that pack exponentials keep at least one other instruction between the v_exp and this (attn_d512.hip: the row-sum
// add / the deferred pair); where that distance is not under control use pack2bf_trans.
IR_DEVINL uint32_t pack2bf_valu(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
IR_DEVINL uint32_t pack2bf_trans(float lo, float hi) {   // operands may come straight from v_exp_f32 / v_rcp_f32: one wait state first
    uint32_t r;
    asm("s_nop 0\n\tv_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// Two fp32 -> packed bf16 by TRUNCATION: one full-rate v_perm_b32 (bytes 2,3 of either source) instead of the quarter-rate v_cvt_pk_bf16_f32
// (measured 8.8 issue cycles). Only where the consumer normalises by a sum of the SAME truncated values (the softmax probabilities of
// flash_attn_pp2_kernel: the denominator is the ones row of V^T under the same P operand), so that the mean truncation bias cancels and what is
// left has the variance of round-to-nearest. sel = 0x07060302 lives in a scalar register (VOP3 takes no literal on gfx9).
IR_DEVINL uint32_t pack2bf_trunc(float lo, float hi, uint32_t sel) {
    uint32_t r;
    asm("v_perm_b32 %0, %1, %2, %3" : "=v"(r) : "v"(hi), "v"(lo), "s"(sel));
    return r;
}
// The same with operands that may be fresh from v_exp_f32. The transcendental unit retires a wave in four passes of 16 lanes, and a full-rate
// consumer two issue slots behind it (s_nop 0 + v_perm_b32, measured in flash_attn_pp2_kernel's first tile) still read stale values in lanes
// 16..31 of some waves - results changed from run to run. v_cvt_pk_bf16_f32 is slow enough to hide that behind one wait state; v_perm_b32 is not.
IR_DEVINL uint32_t pack2bf_trunc_trans(float lo, float hi, uint32_t sel) {
    uint32_t r;
    asm("s_nop 4\n\tv_perm_b32 %0, %1, %2, %3" : "=v"(r) : "v"(hi), "v"(lo), "s"(sel));
    return r;
}
IR_DEVINL float bflo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
IR_DEVINL float bfhi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

IR_DEVINL f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// C/D fragment of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
IR_DEVINL int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Compact activations (the epilogue is unrolled 16x per lane: a libm erff/tanhf there costs more instruction-cache misses
// than arithmetic). fast_rcp: v_rcp_f32 (1 ulp); __expf: v_exp_f32 on x*log2(e).
IR_DEVINL float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
IR_DEVINL float fast_sigmoid(float x) { return fast_rcp(1.0f + __expf(-x)); }
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, below fp32 rounding of the surrounding arithmetic)
IR_DEVINL float fast_erf(float z) {
    const float az = fabsf(z);
    const float t = fast_rcp(1.0f + 0.3275911f * az);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.0f - poly * __expf(-az * az);
    return copysignf(e, z);
}
IR_DEVINL float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f)); }
// tanh-GELU as x * rcp(1 + exp2(x * (c0 + c1 x^2))), c0 = -2 sqrt(2 / pi) log2(e), c1 = c0 * 0.044715: 0.5 (1 + tanh(u)) == sigmoid(2u), five full-rate
// instructions and two transcendentals (round 6; before: nine and two). ONE definition for every kernel: which GEMM kernel runs a linear depends on its row
// count (gemm_pp_kernel from 192 workgroups on), and a tile-sharded frame must not differ from the unsharded one in the last bit.
// exp2 overflow (x << 0) gives rcp(inf) = 0 -> -0, underflow gives x.
IR_DEVINL float gelu_tanh(float x) {
    const float c0 = -2.0f * 0.7978845608028654f * 1.44269504088896340736f, c1 = c0 * 0.044715f;
    const float t = __builtin_fmaf(x * x, c1, c0);
    return x * fast_rcp(1.0f + __builtin_amdgcn_exp2f(x * t));
}
IR_DEVINL float silu(float x) { return x * fast_sigmoid(x); }

// LDS-DMA (global_load ... lds) completion counts on vmcnt. hipcc does NOT order a later ds_read behind a pending DMA on its own
// (a workgroup-scope fence only waits lgkmcnt on gfx950), so every barrier that publishes DMA-written LDS is preceded by this.
IR_DEVINL void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Hand-scheduled LDS fragment reads: an inline-asm ds_read_b128 is invisible to hipcc's waitcnt insertion, so the caller counts
// s_waitcnt lgkmcnt(N) itself (N = reads issued after the one about to be consumed). Left to hipcc, a read issued for a LATER MFMA
// is often waited for together with the current one (lgkmcnt(0)), which exposes the full LDS latency when only one wave per SIMD is
// in its matrix phase. Pin the order around these with __builtin_amdgcn_sched_barrier(0).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int OFF>
IR_DEVINL bf16x8 lds_read16(uint32_t addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
IR_DEVINL void wait_lds() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N)); }
template <int N>
IR_DEVINL void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }  // all but the N youngest VMEM / LDS-DMA ops
IR_DEVINL uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_ptr_t)p; }

// max of a value with its partner lane in the other half of the wave (lane ^ 32): one v_permlane32_swap instead of an LDS round
// trip (ds_bpermute). The instruction swaps the upper half of its first operand with the lower half of its second; with both
// holding v, afterwards a = {v[0..31], v[0..31]} and b = {v[32..63], v[32..63]}, so max(a, b) is the cross-half max on every lane.
// Inline asm on purpose: written with __builtin_amdgcn_permlane32_swap, hipcc (ROCm 7.2) folds max(r[0], r[1]) to r[0] - it never
// emits the second extract - and the exchange silently disappears. The s_nop covers the VALU-write -> permlane read hazard.
IR_DEVINL float xhalf_max(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

IR_DEVINL float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
IR_DEVINL float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// activation codes shared by host and device
enum { IR_ACT_NONE = 0, IR_ACT_GELU_ERF = 1, IR_ACT_GELU_TANH = 2, IR_ACT_LRELU = 3, IR_ACT_SILU = 4 };
