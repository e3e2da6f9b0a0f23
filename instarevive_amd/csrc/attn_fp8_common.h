// Helpers shared by the fp8 (OCP e4m3, MX-scaled MFMA) attention kernels: attn_fp8.hip (DiT self-attention, head dim 72) and
// attn_d512_fp8.hip (VAE mid-block attention, head dim 512).
#pragma once
#include "common.h"

typedef __attribute__((address_space(3))) void* f8_lds_t;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) short s16x2;
IR_DEVINL void f8_glds16(const void* g, f8_lds_t l) { __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0); }

namespace f8c {
constexpr int E_BIAS = 7;   // block exponent = exponent of the block maximum - 7: the maximum lands in [128, 256) <= 448
}

// E8M0 byte of a block whose largest magnitude is mx (>= 0): 2^(byte - 127) = 2^(floor(log2 mx) - 7), clamped to a valid byte
IR_DEVINL int f8_block_byte(float mx) {
    const int b = (int)(__builtin_bit_cast(uint32_t, mx) >> 23) - f8c::E_BIAS;
    return b < 1 ? 1 : (b > 254 ? 254 : b);
}
// The MFMA's two 32-k scale blocks are BYTE RANGES of the operand registers, not lane halves (tools/fp8_cvt_probe.hip): block b = bytes
// 16b .. 16b+15 of BOTH lanes (l, l ^ 32) of a row / column, and its exponent is read from lane (l & 31) + 32 b. This kernel gives both
// blocks of a row / column ONE exponent (e4m3 is a floating format: the shared exponent only has to keep the largest of the 64 values
// in range, values 2^17 below it do not matter to any sum), so an operand's exponent is the maximum over the lane PAIR's 2 x 32 values.
// mx: the lane's own maximum (>= 0). Returns 2^(byte - 127) as a float (the convert's scale operand) and the byte for the MFMA.
IR_DEVINL float f8_pair_scale(float mx, int& byte) {
    float a = mx, b = mx;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a = {lower half's mx} on both halves, b = {upper half's}
    const int t = max((int)(__builtin_bit_cast(uint32_t, __builtin_fmaxf(a, b)) >> 23), f8c::E_BIAS + 1);   // biased exponent of the pair maximum, >= 8
    byte = t - f8c::E_BIAS;
    return __builtin_bit_cast(float, (uint32_t)byte << 23);
}
// two fp32 / 2^(byte - 127) -> two e4m3 bytes in the low or the high half of `old`
template <bool HI>
IR_DEVINL uint32_t f8_cvt2(uint32_t old, float a, float b, float scale_f) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(s16x2, old), a, b, scale_f, HI));
}

// An MFMA reads its A / B registers for a while after it has issued (the 8-register e4m3 operands longest), and nothing stalls a VALU
// instruction that overwrites them meanwhile. keep() pins a value's registers up to the point where it stands (an empty asm that "reads" it).
template <class T>
IR_DEVINL void keep(const T& x) { asm volatile("" ::"v"(x)); }
IR_DEVINL i32x8 f8_join(bf16x8 lo, bf16x8 hi) {
    const uint4 a = __builtin_bit_cast(uint4, lo), b = __builtin_bit_cast(uint4, hi);
    i32x8 r;
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    return r;
}
