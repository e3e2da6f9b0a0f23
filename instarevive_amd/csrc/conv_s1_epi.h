// Epilogue of the one-wave-per-SIMD 3x3 convolution kernels (conv_s1.hip bf16, conv_s1_fp8.hip e4m3): the 64 accumulator tiles of a wave
// (a[0:255]; tile (patch row A, half MX, channel tile CT) at 4 * ((2A + MX) * 8 + CT), a lane holding 4 consecutive channels of one pixel)
// -> bias / per-channel gate -> wave-private fp32 slab in LDS -> rows read back as 8-channel vectors -> + bf16 residual -> bf16 -> 16-byte
// stores, plus the fused GroupNorm statistics of the values as stored (fixed-order reduction: bit-identical run to run). Same contract as
// igemm_epilogue (igemm.hip). GATE: out = (acc + bias) * gate[channel] (the fp8 kernel's dequantisation; bias arrives divided by gate),
// otherwise out = acc * out_scale + bias * out_scale.
#pragma once
#include "common.h"
#include "kernels.h"

namespace cs1e {
constexpr int SROW = 132;                      // slab row stride in floats (128 + 4)
constexpr int SLAB = 16 * SROW * 4;            // 8 448 B per wave: 16 pixels x 128 channels fp32
constexpr int RED_OFF = 4 * SLAB;              // 33 792; the GroupNorm reduction area (1 KB) follows the slabs
constexpr int BYTES = RED_OFF + 4 * 16 * 16;   // what the epilogue needs of the LDS buffer it is given
}  // namespace cs1e

template <int I>
IR_DEVINL float cs1_acc_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "n"(I));
    return x;
}

// ebuf: an LDS buffer of cs1e::BYTES bytes that no wave reads any more; t_*: the tile (first channel, image, patch origin, patch index in
// the image). do_passes / do_stats / mid_stamp: diagnostics of the callers' knock-out and stamp builds (true, true, nullptr in the product).
// Output map (the sub-pixel phase form of conv_s1.hip): tile pixel (y, x) is stored at output pixel (omul * y + oyoff, omul * x + oxoff); tile
// pixels are valid below (hlim, wlim). Identity map: omul = 1, offsets 0, limits = p.Ho, p.Wo. The residual (if any) uses the same map.
// FULL (round 6): the whole 16 x 32 patch lies inside the image - every tile of a 2048 x 2048 map and all but the last row / column of tiles
// otherwise - so the row / column validity tests, the safe residual pixel and the exec-mask save / restore around every 16-byte store (a
// saveexec + branch pair per store, plus the spilled scalars they dragged in) disappear, and store / residual addresses are one wave-uniform
// 64-bit base plus a 32-bit lane offset. GN: the statistics are compiled in or out instead of branched over per store. Same arithmetic in the
// same order as the general form: results are bit-identical.
template <bool GATE, bool FULL, bool FGN>
IR_DEVINL void cs1_epilogue_impl(const IGemmParams& p, unsigned char* ebuf, int tid, int lane, int wid, int c16, int kq, int t_n0, int t_img, int t_oy0,
                                 int t_ox0, int t_trem, bool do_passes, bool do_stats, unsigned long long* mid_stamp, int omul, int oyoff, int oxoff, int hlim,
                                 int wlim) {
    using namespace cs1e;
    float* slab = reinterpret_cast<float*>(ebuf + wid * SLAB);
    const int co8 = (lane & 15) * 8, xq = lane >> 4;
    const int n0 = t_n0, img = t_img, oy0 = t_oy0, ox0 = t_ox0;
    const float osc = p.out_scale;
    // plain: bias * out_scale per accumulator tile, the slab write is one fused multiply-add per value. GATE: the accumulators go to the
    // slab as they are and gate / bias are applied on the read-back side, where a lane keeps the same 8 channels for the whole tile
    // (16 registers instead of 64)
    // FULL && !GATE: the caller started the accumulators at the bias (out_scale == 1, launcher) - the slab write is the accumulator itself, and the 32
    // bias registers do not sit in the epilogue's register budget (with them the NORM kernel spilled 39 registers that every tile's prologue reloaded)
    constexpr bool BIAS_IN_ACC = FULL && !GATE;
    f32x4 bias4[(GATE || BIAS_IN_ACC) ? 1 : 8], g_lo, g_hi, b_lo, b_hi;
    if constexpr (BIAS_IN_ACC) {
    } else if constexpr (GATE) {
        g_lo = *reinterpret_cast<const f32x4*>(p.gate + n0 + co8);
        g_hi = *reinterpret_cast<const f32x4*>(p.gate + n0 + co8 + 4);
        b_lo = (p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + co8) : f32x4{0.f, 0.f, 0.f, 0.f}) * g_lo;
        b_hi = (p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + co8 + 4) : f32x4{0.f, 0.f, 0.f, 0.f}) * g_hi;
    } else {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
            bias4[ct] = (p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + 16 * ct + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f}) * osc;
    }
    f32x4 sA4 = {0.f, 0.f, 0.f, 0.f}, qA4 = sA4, sB4 = sA4, qB4 = sA4;   // GroupNorm partials, channels co8 .. +3 and co8+4 .. +7
    // Row it of a pass is pixel (oyw + A, oxl + 4 it): per-lane base pointers once per tile, uniform offsets per pass and row
    const int oyw = oy0 + 4 * wid, oxl = ox0 + xq;
    unsigned xm = 0;   // bit it: column oxl + 4 it lies inside the image
    if constexpr (!FULL) {
#pragma unroll
        for (int it = 0; it < 8; ++it) xm |= (oxl + 4 * it < wlim ? 1u : 0u) << it;
    }
    const long pix0 = ((long)img * p.Ho + (long)omul * oyw + oyoff) * p.Wo + (long)omul * oxl + oxoff;
    bf16_t* obase = reinterpret_cast<bf16_t*>(p.out) + pix0 * p.out_cs + n0 + co8;
    const bf16_t* rbase = reinterpret_cast<const bf16_t*>(p.res) + pix0 * p.res_cs + n0 + co8;
    const bf16_t* rsafe = reinterpret_cast<const bf16_t*>(p.res) + (((long)img * p.Ho + (long)omul * oy0 + oyoff) * p.Wo + (long)omul * ox0 + oxoff) * p.res_cs + n0 + co8;   // always inside
    const long o_row = (long)omul * p.Wo * p.out_cs, r_row = (long)omul * p.Wo * p.res_cs;
    const long o_col = (long)omul * p.out_cs, r_col = (long)omul * p.res_cs;   // element step between tile columns
    // FULL: wave-uniform bases (the wave's first pixel, the tile's first channel) + 32-bit lane offsets; four rows of a patch span less than
    // 2^31 elements for every tensor the launcher admits (4 x omul x Wo x cs)
    const long upix = ((long)img * p.Ho + (long)omul * oyw + oyoff) * p.Wo + (long)omul * ox0 + oxoff;
    bf16_t* const uo = reinterpret_cast<bf16_t*>(p.out) + upix * p.out_cs + n0;
    const bf16_t* const ur = reinterpret_cast<const bf16_t*>(p.res) + upix * p.res_cs + n0;
    // byte offsets, so that the address is "uniform 64-bit base + zero-extended 32-bit lane offset" - the saddr form of global_load / global_store.
    // The empty asm makes the lane terms opaque per call: left visible, hipcc hoists all 64 (row, column) offsets of a tile out of the persistent
    // tile loop as 64-bit values - 128 registers it then spills, each reloaded in front of the load or store that uses it
    unsigned lo_o = (unsigned)(xq * (int)o_col + co8) * 2u, lo_r = (unsigned)(xq * (int)r_col + co8) * 2u;
    if constexpr (FULL) asm volatile("" : "+v"(lo_o), "+v"(lo_r));
    const unsigned uo_row = (unsigned)o_row * 2u, ur_row = (unsigned)r_row * 2u, uo_c4 = 8u * (unsigned)o_col, ur_c4 = 8u * (unsigned)r_col;
    unsigned char* const uob = reinterpret_cast<unsigned char*>(uo);
    const unsigned char* const urb = reinterpret_cast<const unsigned char*>(ur);
    constexpr bool GN = FULL && FGN;                     // FULL: the statistics are compiled in (FGN) or out
    const bool do_gn = FULL ? FGN : p.gn_part != nullptr;
    uint4 rrb[2][8];   // residual rows of the pass being finished / of the next pass
    auto res_fetch = [&](int a) {   // rows outside the image read a safe pixel
        if constexpr (FULL) {
#pragma unroll
            for (int it = 0; it < 8; ++it) rrb[a & 1][it] = *reinterpret_cast<const uint4*>(urb + (lo_r + (unsigned)a * ur_row + (unsigned)it * ur_c4));
        } else {
            const bool yok = oyw + a < hlim;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const bool v = yok && ((xm >> it) & 1);
                rrb[a & 1][it] = *reinterpret_cast<const uint4*>(v ? rbase + a * r_row + (long)(4 * it) * r_col : rsafe);
            }
        }
    };
    auto pass = [&](auto ac, auto resc) {
        constexpr int A = decltype(ac)::value;
        constexpr bool RES = decltype(resc)::value;
        const bool yok = FULL || oyw + A < hlim;
        uint4 (&rr)[8] = rrb[A & 1];
        if constexpr (RES && A < 3) res_fetch(A + 1);   // the next pass's residual rows fly during this pass
        [&]<int... MXS>(std::integer_sequence<int, MXS...>) {
            ([&] {
                constexpr int MX = MXS;
                [&]<int... CTS>(std::integer_sequence<int, CTS...>) {
                    ([&] {
                        constexpr int CT = CTS, LO = 4 * ((A * 2 + MX) * 8 + CT);
                        f32x4 v = f32x4{cs1_acc_read<LO>(), cs1_acc_read<LO + 1>(), cs1_acc_read<LO + 2>(), cs1_acc_read<LO + 3>()};
                        if constexpr (!GATE && !BIAS_IN_ACC) v = v * osc + bias4[CT];
                        *reinterpret_cast<f32x4*>(&slab[c16 * SROW + 16 * CT + 4 * kq]) = v;
                    }(), ...);
                }(std::make_integer_sequence<int, 8>{});
                if constexpr (FULL) __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                f32x4 lo[4], hi[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lo[j] = *reinterpret_cast<const f32x4*>(&slab[(4 * j + xq) * SROW + co8]);
                    hi[j] = *reinterpret_cast<const f32x4*>(&slab[(4 * j + xq) * SROW + co8 + 4]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int it = 4 * MX + j;   // pixel column oxl + 4 it
                    f32x4 a = lo[j], b = hi[j];
                    if constexpr (GATE) { a = a * g_lo + b_lo; b = b * g_hi + b_hi; }
                    if constexpr (RES) {
                        a += f32x4{bflo(rr[it].x), bfhi(rr[it].x), bflo(rr[it].y), bfhi(rr[it].y)};
                        b += f32x4{bflo(rr[it].z), bfhi(rr[it].z), bflo(rr[it].w), bfhi(rr[it].w)};
                    }
                    const uint4 pk = make_uint4(pack2bf_valu(a[0], a[1]), pack2bf_valu(a[2], a[3]), pack2bf_valu(b[0], b[1]), pack2bf_valu(b[2], b[3]));
                    if constexpr (FULL) {
                        *reinterpret_cast<uint4*>(uob + (lo_o + (unsigned)A * uo_row + (unsigned)it * uo_c4)) = pk;
                        if constexpr (GN) {   // statistics of the values as stored (bf16-rounded)
                            const f32x4 ar = {bflo(pk.x), bfhi(pk.x), bflo(pk.y), bfhi(pk.y)}, br = {bflo(pk.z), bfhi(pk.z), bflo(pk.w), bfhi(pk.w)};
                            sA4 += ar; qA4 += ar * ar;
                            sB4 += br; qB4 += br * br;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (yok && ((xm >> it) & 1)) {
                        *reinterpret_cast<uint4*>(obase + A * o_row + (long)(4 * it) * o_col) = pk;
                        if (do_gn) {   // statistics of the values as stored (bf16-rounded)
                            const f32x4 ar = {bflo(pk.x), bfhi(pk.x), bflo(pk.y), bfhi(pk.y)}, br = {bflo(pk.z), bfhi(pk.z), bflo(pk.w), bfhi(pk.w)};
                            sA4 += ar; qA4 += ar * ar;
                            sB4 += br; qB4 += br * br;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if constexpr (FULL) __builtin_amdgcn_sched_barrier(0);   // straight-line code: without it hipcc interleaves the half passes until it spills
            }(), ...);
        }(std::make_integer_sequence<int, 2>{});
    };
    if (do_passes) {
        if (p.res) {
            res_fetch(0);
            pass(std::integral_constant<int, 0>{}, std::true_type{});
            pass(std::integral_constant<int, 1>{}, std::true_type{});
            pass(std::integral_constant<int, 2>{}, std::true_type{});
            pass(std::integral_constant<int, 3>{}, std::true_type{});
        } else {
            pass(std::integral_constant<int, 0>{}, std::false_type{});
            pass(std::integral_constant<int, 1>{}, std::false_type{});
            pass(std::integral_constant<int, 2>{}, std::false_type{});
            pass(std::integral_constant<int, 3>{}, std::false_type{});
        }
    }
    if (mid_stamp) *mid_stamp = __builtin_amdgcn_s_memrealtime();
    if (do_gn && do_stats) {
        // Fixed-order workgroup reduction (no atomics, bit-identical run to run). Unit u = 4 channels; lane (L = lane & 15) holds units 2L and
        // 2L+1 over the pixel columns xq, xq + 4, ...: first the four column classes of a wave (lanes L, L+16, L+32, L+48), then the four
        // waves through LDS, then the units of a group.
        const float sA = (sA4[0] + sA4[1]) + (sA4[2] + sA4[3]), qA = (qA4[0] + qA4[1]) + (qA4[2] + qA4[3]);
        const float sB = (sB4[0] + sB4[1]) + (sB4[2] + sB4[3]), qB = (qB4[0] + qB4[1]) + (qB4[2] + qB4[3]);
        f32x4 v = {sA, qA, sB, qB};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] += __shfl_xor(v[e], 16);
            v[e] += __shfl_xor(v[e], 32);
        }
        float* red = reinterpret_cast<float*>(ebuf + RED_OFF);   // [wave][L][4]
        if (lane < 16) *reinterpret_cast<f32x4*>(&red[(wid * 16 + lane) * 4]) = v;
        __syncthreads();
        if (tid < 32) {   // thread u: unit u of the 32 units of this channel tile
            const int L = tid >> 1, hf = (tid & 1) * 2;
            float a = (red[(0 * 16 + L) * 4 + hf] + red[(1 * 16 + L) * 4 + hf]) + (red[(2 * 16 + L) * 4 + hf] + red[(3 * 16 + L) * 4 + hf]);
            float b = (red[(0 * 16 + L) * 4 + hf + 1] + red[(1 * 16 + L) * 4 + hf + 1]) + (red[(2 * 16 + L) * 4 + hf + 1] + red[(3 * 16 + L) * 4 + hf + 1]);
            const int upg = p.gn_cpg >> 2;   // units per group: 1, 2, 4 or 8 (launcher)
            for (int m = 1; m < upg; m <<= 1) {
                a += __shfl_xor(a, m);
                b += __shfl_xor(b, m);
            }
            if ((tid & (upg - 1)) == 0) {
                const int G = p.Cout / p.gn_cpg, g = n0 / p.gn_cpg + tid / upg;
                float* dst = p.gn_part + ((long)img * p.gn_chunks + t_trem) * 2 * G;
                dst[g] = a;
                dst[G + g] = b;
            }
        }
    }
}

// FULL is the caller's promise (a kernel instantiation of its own, chosen by the launcher): EVERY tile of the launch is a whole patch inside the
// image; whether the launch writes statistics is part of the instantiation too. (Chosen per tile inside one kernel - both forms inlined side by side - hipcc's register allocation
// of the 512-register kernels fell apart: 145 spilled VGPRs and conv_halo_s1_kernel 40.6 -> 46.3 ms per image, profiles/r06_ab_s1_epi_pertile.txt.)
template <bool GATE, int EMODE = 0>   // EMODE 0: general; 1: FULL with statistics; 2: FULL without (the launch has no gn_part)
IR_DEVINL void cs1_epilogue(const IGemmParams& p, unsigned char* ebuf, int tid, int lane, int wid, int c16, int kq, int t_n0, int t_img, int t_oy0,
                            int t_ox0, int t_trem, bool do_passes, bool do_stats, unsigned long long* mid_stamp, int omul, int oyoff, int oxoff, int hlim,
                            int wlim) {
    cs1_epilogue_impl<GATE, EMODE != 0, EMODE == 1>(p, ebuf, tid, lane, wid, c16, kq, t_n0, t_img, t_oy0, t_ox0, t_trem, do_passes, do_stats, mid_stamp, omul, oyoff, oxoff, hlim, wlim);
}
