// Internal launcher interface between the C-ABI layer (api.cpp) and the HIP kernels. Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;

// 1: launchers pick the older 4-wave kernels only (ir_set_plain_kernels; defined in igemm.hip)
extern int g_ir_plain_kernels;

// ---- implicit GEMM (igemm.hip)
struct IGemmParams {
    // A operand: NHWC bf16 activations. taps==1: row m at in + m*in_cs. taps==9: 3x3 window around output pixel.
    const bf16_t* in;
    int NB, H, W;   // input batch / height / width (before the optional nearest-2x upsample)
    int Cin;        // channels reduced over per tap (multiple of 32; zero-padded channels are fine)
    int in_cs;      // elements between consecutive pixels / rows of `in`
    int Ho, Wo;     // output height / width
    int taps;       // 1 or 9
    int stride;     // 1 or 2
    int pad;        // 1: symmetric "same" pad; 0: pad only bottom/right (the VAE's (0,1,0,1) stride-2 downsample)
    int up;         // 1: conv runs on the nearest-2x upsampled input (2H x 2W), folded into addressing
    // B operand
    const bf16_t* wgt;  // [Cout_pad][taps*Cin] bf16, K contiguous; rows wgt_rs elements apart
    long wgt_rs;
    int Cout, Cout_pad, M;
    // epilogue: v = act(acc + bias) * out_scale; v *= gate; v += res; store
    const float* bias;  // [Cout_pad] or null
    int act;
    float slope;
    float out_scale;
    const float* gate;  // gate[(m / rows_per_batch) * gate_stride + n] or null
    int gate_stride, rows_per_batch;
    const void* res;    // residual [.][res_cs], row index m (or m % res_mod when res_mod > 0)
    int res_f32, res_cs, res_mod;
    void* out;
    int out_f32, out_cs;
    bf16_t* out2;       // optional extra bf16 copy of the result
    int out2_cs;
    int fp8;            // 1: `in` and `wgt` hold OCP e4m3 bytes; Cin / in_cs / wgt_rs count PAIRS of fp8 channels (2-byte units), gate
                        // carries the dequantisation factor per output channel, bias is pre-divided by it (3x3 stride-1 convs only)
    int force_generic;  // 1: never take the halo-tile conv path (A/B testing)
    int vec;            // set by the launcher: all strides/pointers allow 4-element vector I/O
    // Optional fused GroupNorm statistics of the OUTPUT (the tensor a following GroupNorm normalises): every workgroup writes
    // the per-group sum and sum of squares of its (bf16-rounded) outputs to gn_part[((image*gn_chunks + chunk)*2 + {0,1})*G + g],
    // G = Cout / gn_cpg, chunk = the workgroup's pixel tile within the image. Set gn_chunks from ir_igemm_gn_chunks().
    float* gn_part;
    int gn_cpg, gn_chunks;
    // Split-K for small-M launches: the caller sets allow_splitk and, when ir_igemm_splitk(p) > 1, ks_ws (that many slices of p.M * Cout_pad
    // floats of scratch); ksplit is set by the launcher. Each split stores its partial tile in its own slice; a second kernel adds the
    // slices in order and applies the epilogue (deterministic).
    int allow_splitk, ksplit;
    int s1_min_tiles;   // > 32: conv_halo_s1_kernel only takes images with at least that many patch tiles (callers that never batch small images)
    float* ks_ws;
    // Optional second, TRANSPOSED copy of the output columns >= vt_col0 (gemm_pp_kernel's bf16 form only; M % 256 == 0, vt_T % 64 == 0): column
    // vt_col0 + head * vt_hd + d of row m = batch * vt_T + t goes to vt_out[batch * vt_bs + (head * vt_dv + d) * vt_ld + t] - the V^T operand of
    // the DiT self-attention straight from the qkv projection's epilogue (the rows d >= vt_hd of a head - the ones row and the padding - are
    // written once per run by ir_launch_vt_pad_init). ir_igemm_writes_vt(p) tells whether the launch ir_launch_igemm picks honours it.
    bf16_t* vt_out;
    int vt_col0, vt_hd, vt_dv, vt_ld, vt_T;
    long vt_bs;
    // 1: `wgt` holds the four sub-pixel phase matrices [2 dy + dx][Cout_pad][2 sy + sx][Cin] of an up = 1 conv (wgt_rs = 4 * Cin): only valid when
    // ir_conv_s1_up2x2_takes(p) - the caller asks first and passes the ordinary 9-tap weights otherwise
    int up2x2;
    // GroupNorm + SiLU of the INPUT applied inside conv_halo_s1_kernel (NORM form): per image and input channel scale / shift ([NB][Cin] fp32, what
    // ir_launch_groupnorm_fused(..., y = nullptr) leaves in its workspace); `in` is then the un-normalised tensor. Only when ir_conv_s1_norm_takes(p).
    const float* nrm_scale;
    const float* nrm_shift;
};
bool ir_conv_s1_norm_takes(const IGemmParams& p);
int ir_igemm_writes_vt(const IGemmParams& p);
int ir_launch_vt_pad_init(bf16_t* vt, int heads_total, int D, int DV, int T, int Tpad, hipStream_t s);
int ir_launch_igemm(const IGemmParams& p, hipStream_t s);
// which kernel ir_launch_igemm picks for p: 0 conv_halo_s1, 1 conv_halo_pp, 2 gemm_pp, 3 conv_halo, 4 igemm_kernel (profiler rows)
int ir_igemm_kernel_id(const IGemmParams& p);
// Pixel tiles per image the kernel ir_launch_igemm would pick for p writes statistics for, or 0 if this launch cannot fuse them.
int ir_igemm_gn_chunks(const IGemmParams& p);
// Split count ir_launch_igemm would use for p given a workspace (0: none); see IGemmParams::allow_splitk.
int ir_igemm_splitk(const IGemmParams& p);

// conv_s1.hip: the one-wave-per-SIMD 3x3 convolution (16 x 32 patches x 128 channels); ir_launch_igemm routes eligible launches to it
bool ir_conv_s1_takes(const IGemmParams& p);
bool ir_conv64_takes(const IGemmParams& p);   // vae_io.hip: 64 -> 64 at full resolution (SwinIR conv_hr)
int ir_launch_conv64(const IGemmParams& p, hipStream_t s, bool force = false);   // force: take the shape whatever IR_CONV64 says (ir_op_conv64)
// "nearest-2x upsample + 3 x 3 conv" as four 2 x 2 convs on the low-resolution tensor (conv_s1.hip); p.up2x2 = 1, p.wgt = the phase matrices
bool ir_conv_s1_up2x2_takes(const IGemmParams& p);
int ir_conv_s1_up2x2_tiles(const IGemmParams& p);
int ir_launch_conv_s1_up2x2(const IGemmParams& p, hipStream_t s);
int ir_conv_s1_tiles(const IGemmParams& p);   // pixel tiles per image (fused GroupNorm statistics: one partial per tile)
int ir_launch_conv_s1(const IGemmParams& p, hipStream_t s);
// conv_s1_fp8.hip: the same structure on e4m3 operands (p.fp8 launches with Cin a multiple of 128 channels)
bool ir_conv_s1_fp8_takes(const IGemmParams& p);
int ir_launch_conv_s1_fp8(const IGemmParams& p, hipStream_t s);

// ---- norms (norm.hip)
static inline int ir_gn_chunks(long HW) {
    long c = HW / 2048;
    if (c < 1) c = 1;
    if (c > 1024) c = 1024;
    return (int)c;
}
// workspace floats needed by ir_launch_groupnorm
static inline long ir_gn_ws_floats(int N, long HW, int C) {
    const long part = (long)N * ir_gn_chunks(HW) * 2 * C, part2 = (long)N * 256 * 64;  // stand-alone partials | fused second stage
    return (part > part2 ? part : part2) + 2L * N * C;
}
// out_fp8 != 0: y receives OCP e4m3 bytes of (result * out_mul) instead of bf16 (the input of an fp8 convolution)
int ir_launch_groupnorm(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, float* ws, int N, long HW, int C,
                        int G, float eps, int do_silu, hipStream_t s, int out_fp8 = 0, float out_mul = 1.f);
// Same, with the statistics already reduced to per-group partials by the producing conv (IGemmParams::gn_part, `chunks` tiles per
// image): only the finalise and apply launches run. ws needs 2*N*C + N*256*64 floats (scale, shift, second-stage partials).
int ir_launch_groupnorm_fused(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, const float* part, float* ws, int N,
                              long HW, int C, int G, int chunks, float eps, int do_silu, hipStream_t s, int out_fp8 = 0, float out_mul = 1.f);
// y (bf16, may be null) and yf (fp32, may be null) both receive xn*a + b; columns C..ldy-1 are written as zero.
// DiT self-attention token preparation (norm.hip): KV compression (depthwise r x r / stride r over the token grid + optional LayerNorm) and qk_norm (r = 1)
int ir_launch_dit_token_prep(const bf16_t* in, bf16_t* out, const float* w, const float* bias, const float* gamma, const float* beta, int B, int gh, int gw, int r,
                             int C, int in_rs, long in_bs, int out_rs, long out_bs, hipStream_t s);
int ir_launch_layernorm(const float* x, bf16_t* y, float* yf, const float* a, const float* b, long rows, int C, int ldx, int ldy,
                        float eps, long rows_per_batch, int ab_stride, hipStream_t s);
int ir_launch_gemv_f32(const float* w, const float* x, const float* b, float* out, int N, int K, int act, hipStream_t s);

// ---- attention (attention.hip)
struct AttnParams {
    const bf16_t *q, *k, *vt;  // q/k: [B][T][..] rows with head slices; vt: [B][Hh][DV][Tk_pad]
    bf16_t* o;
    long q_bs, k_bs, o_bs;     // batch strides (elements)
    long vt_bs;                // batch stride of vt (0: one K/V set shared by the whole batch); head stride = DV*Tk_pad
    int q_rs, k_rs, o_rs;      // token strides
    int q_hs, k_hs, o_hs;      // head strides
    int B, Hh, Tq, Tk, Tk_pad, D;
    float scale_log2;          // softmax scale * log2(e)
    const float* key_bias;     // optional additive bias per key [B][kb_bs] (natural-log domain)
    long kb_bs;
    int* ovf_flag;             // optional 4 bytes of device scratch: enables the fixed-reference ping-pong kernel (see attention.hip)
    int ovf_map;               // ints available BEHIND ovf_flag[0] (0: none). With B * Hh * ceil(Tq / 256) of them (round 6) flash_attn_pp2_kernel marks the
                               // 256-query workgroups whose fixed reference was outgrown, and the rescaling kernel behind it recomputes only those
                               // instead of the whole launch (one peaky row among 16384 x 16 used to cost 28 x 1.7 ms per image)
};
int ir_launch_flash_attn(const AttnParams& p, hipStream_t s);
bool ir_flash_attn_is_pp2(const AttnParams& p);   // ir_launch_flash_attn routes p to flash_attn_pp2_kernel (profiler rows)
// DiT self-attention (D = 72, Tk % 64 == 0, no key bias, ovf_flag set) as one wave per SIMD with two query groups (attn_d512.hip)
int ir_launch_flash_attn_pp2(const AttnParams& p, hipStream_t s);
// DiT cross-attention (D = 72, <= 320 keys, optional additive key bias, ovf_flag set) as a persistent one-wave-per-SIMD kernel with the head's
// K / V^T resident in LDS (attn_d512.hip); ir_launch_flash_attn routes to it and queues the rescaling kernel behind it
bool ir_igemm_up2x2_takes(const IGemmParams& p);   // a phase kernel (conv_halo_s1_kernel<0, 4> or conv_halo_kernel<.., PH>) takes this upsampling conv in its sub-pixel form
bool ir_flash_attn_x72_takes(const AttnParams& p);
int ir_launch_flash_attn_x72(const AttnParams& p, hipStream_t s);
// the rescaling 4-wave kernel alone, as the fallback behind a fixed-reference kernel: returns at once unless *p.ovf_flag is set
int ir_launch_flash_attn_fallback(const AttnParams& p, hipStream_t s);
// DiT self-attention on fp8 (e4m3) MFMA operands (attn_fp8.hip; BASELINE.json configs[4]). p as for ir_launch_flash_attn's self-attention form
// (p.vt: the bf16 V^T buffer, only written / read when the overflow fallback runs); v: the V rows (strides of p.k); tiles: scratch of
// ir_attn_fp8_tile_bytes(B, Hh, Tk) bytes for the quantised K / V^T tile images.
size_t ir_attn_fp8_tile_bytes(int B, int Hh, int Tk);
int ir_launch_flash_attn_fp8(const AttnParams& p, const bf16_t* v, uint8_t* tiles, hipStream_t s);
int ir_launch_flash_attn_d512(const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* o, int T, int rs, int o_rs, long vt_rs,
                              float scale, hipStream_t s, const int* only_if = nullptr);
// d = 512 without the redundant score product (attn_d512.hip): V^T in 32-key tiles [B][T/32][512][32]; a set *ovf_flag afterwards
// means the result must be recomputed by ir_launch_flash_attn_d512 (which is given the flag as `only_if` and returns at once otherwise)
int ir_launch_transpose_v_tiles(const bf16_t* v, bf16_t* vt, int B, int T, int rs, long v_bs, long vt_bs, hipStream_t s, const int* only_if = nullptr);
int ir_launch_flash_attn_d512_v2(const bf16_t* q, const bf16_t* k, const bf16_t* vt_tiles, bf16_t* o, int B, int T, int rs, int o_rs, long qk_bs,
                                 long vt_bs, long o_bs, float scale, int* ovf_flag, hipStream_t s, const int* only_if = nullptr);   // only_if: run only when *only_if != 0
int ir_launch_flash_attn_d512_v2_rows(const bf16_t* q, const bf16_t* k, const bf16_t* vt_tiles, bf16_t* o, int T, int rows, int rs, int o_rs,
                                      float scale, int* ovf_flag, hipStream_t s);   // a query-row shard (q / o offset by the caller), all T keys
int ir_launch_transpose_v(const bf16_t* v, bf16_t* vt, long v_bs, int v_rs, int v_hs, int B, int Hh, int T, int Tpad, int D,
                          int DV, hipStream_t s, const int* only_if = nullptr);
int ir_launch_swin_attn(const bf16_t* qkv, bf16_t* out, const float* biasT, int B, int H, int W, int heads, int ld, int ldo,
                        int shift, float scale, hipStream_t s);
int ir_launch_softmax_rows(const float* x, bf16_t* y, long rows, int cols, long ldx, long ldy, hipStream_t s);
static inline int ir_attn_dv(int D) { return (D + 31) & ~31; }

// ---- fused SwinIR block halves (swin_fused.hip)
// out = x + fc2(gelu(fc1(LayerNorm(x)))) on the fp32 residual stream [T][192] (in place allowed); out2: optional bf16 copy
int ir_launch_swin_mlp(const float* x, float* out, bf16_t* out2, const void* w_tiles, const float* vec, long T, int C, int hid_p, float eps,
                       hipStream_t s, const float* next_g = nullptr, const float* next_b = nullptr, const void* qkv_tiles = nullptr,
                       const float* qkv_b = nullptr, int qkv_n = 0);   // qkv_tiles: out2 = the next block's qkv rows [T][qkv_n] instead of its norm1
// window attention of all heads + output projection + residual in one launch (swin_fused.hip): qkv [tokens][576] bf16, xres / out [tokens][192]
// fp32 (may alias), proj_t = proj weights [192][192] bf16 with columns in accumulator order (weights.pack_swinir)
int ir_launch_swin_attn_proj(const bf16_t* qkv, const float* xres, float* out, const void* proj_t, const float* proj_b, const float* biasT, int B,
                             int H, int W, int shift, float scale, hipStream_t s);

// the whole block behind its qkv projection in one launch (swin_block_kernel): the two above back to back with the post-attention row kept in registers;
// x_in / x_out may alias, out2 as in ir_launch_swin_mlp (bf16 copy of the new rows, or the next block's norm1 rows, or - with qkv_tiles - its qkv rows)
int ir_launch_swin_block(const bf16_t* qkv, const float* x_in, float* x_out, bf16_t* out2, const void* proj_t, const float* proj_b, const float* biasT,
                         int B, int H, int W, int shift, float scale, const void* w_tiles, const float* vec, int C, int hid_p, float eps, hipStream_t s,
                         const float* next_g = nullptr, const float* next_b = nullptr, const void* qkv_tiles = nullptr, const float* qkv_b = nullptr,
                         int qkv_n = 0);

// ---- T5 encoder glue (t5.hip)
int ir_launch_t5_embed(const int* ids, const bf16_t* table, float* x, long rows, int D, int vocab, int* bad, hipStream_t s);
int ir_launch_t5_rmsnorm(const float* x, const float* w, bf16_t* yb, float* yf, long rows, int D, float eps, hipStream_t s);
int ir_launch_t5_attn(const bf16_t* qkv, const float* bias, const float* key_mask, bf16_t* out, int B, int T, int H, int dk, hipStream_t s);
int ir_launch_t5_gated_gelu(const bf16_t* ab, bf16_t* out, long rows, int F, hipStream_t s);

// ---- layout / elementwise (elementwise.hip)
int ir_launch_u8_to_nchw(const uint8_t* in, float* out, int N, int H, int W, hipStream_t s);
int ir_launch_swin_prep(const float* x_nchw, bf16_t* out, int N, int H, int W, const float* mean3, float img_range, hipStream_t s);
int ir_launch_nhwc_to_nchw(const float* in, int in_cs, float* out, int N, int C, long HW, float scale, float shift, int clamp01,
                           hipStream_t s);
int ir_launch_nchw_to_nhwc_bf16(const float* in, bf16_t* out, int N, int C, long HW, int Cpad, float scale, float shift,
                                hipStream_t s);
int ir_launch_nhwc_f32_to_bf16pad(const float* in, int in_cs, bf16_t* out, long npix, int C, int Cpad, float scale, float shift,
                                  hipStream_t s);
int ir_launch_quant_mean(const float* h8, int h_cs, const float* wq, const float* bq, float* lat_nchw, int N, long HW, float scale,
                         hipStream_t s);
int ir_launch_latent_prep(const float* lat_nchw, const float* wpq, const float* bpq, bf16_t* out, int N, long HW, int Cpad,
                          float in_scale, hipStream_t s);
int ir_launch_patchify(const float* lat_nchw, bf16_t* out, int N, int h, int w, int Cpad, hipStream_t s);
int ir_launch_unpatchify(const float* tok, float* out_nchw, int N, int h, int w, hipStream_t s);
int ir_launch_eps_to_x0(const float* tok, const float* lat_in, float* lat_out, int N, int h, int w, float sqrt_acp,
                        float sqrt_1macp, float out_scale, hipStream_t s);
int ir_launch_nhwc_to_u8(const float* in, int in_cs, uint8_t* out, long npix, float scale, float shift, hipStream_t s);
int ir_launch_nchw_to_u8(const float* in, uint8_t* out, int N, long HW, hipStream_t s);
int ir_launch_zero_f32(float* p, long n, hipStream_t s);
// vae_io.hip: the VAE's first / last convolution at full resolution as HBM-bound kernels of their own
int ir_vae_conv_in_tiles(int H, int W);   // GroupNorm partial tiles per image ir_launch_vae_conv_in writes (the gn_chunks of the following GroupNorm)
int ir_launch_vae_conv_in(const float* in, const bf16_t* wgt, const float* bias, bf16_t* out, float* gn_part, int N, int H, int W, float in_scale,
                          float in_shift, hipStream_t s);
int ir_launch_vae_norm_conv_out(const bf16_t* x, const float* scale, const float* shift, const bf16_t* wgt, const float* bias, float* out, int N, int H,
                                int W, hipStream_t s);
int ir_launch_conv64_to3(const bf16_t* x, const bf16_t* wgt, const float* bias, float* out, int N, int H, int W, hipStream_t s);   // SwinIR conv_last (64 -> 3), same kernel family
int ir_launch_fill_u32(uint32_t* p, long n, uint32_t v, hipStream_t s);
int ir_launch_count_flag(const int* flag, int* counter, hipStream_t s);   // *counter += 1 if *flag != 0 (diagnostic: ir_attn_fallback_count)
int ir_launch_tile_add(float* dst, const float* src, int N, int C, int H, int W, int th, int tw, int y0, int x0, hipStream_t s);
int ir_launch_tile_div(float* dst, int N, int C, int H, int W, int th, int tw, int sy, int sx, hipStream_t s);
int ir_launch_crop_nchw(const float* src, float* dst, int N, int C, int H, int W, int y0, int x0, int th, int tw, float scale,
                        hipStream_t s);
int ir_launch_wavelet_fix(const float* content, const float* style, float* out, float* tmp, int N, int H, int W, hipStream_t s);
int ir_launch_adain_fix(const float* content, const float* style, float* out, float* ws, int N, int H, int W, hipStream_t s);
int ir_launch_silu_f32(const float* in, float* out, long n, hipStream_t s);
int ir_launch_timestep_embed(float* out, float t, int dim, hipStream_t s);
int ir_launch_f32_to_bf16(const float* in, bf16_t* out, long n, hipStream_t s);
int ir_launch_modtab(const float* t, const float* sst, float* out, int L, int R, int C, int t_stride, int scale_mask, hipStream_t s);
int ir_launch_add_bias_rows(float* x, const float* b, long n, int C, hipStream_t s);

// attn_d512_fp8.hip: the VAE mid-block attention (d = 512) on fp8 (e4m3) MFMA operands (BASELINE.json configs[4]); T % 128 == 0, T >= 256
size_t ir_attn_d512_fp8_tile_bytes(int B, int T);
bool ir_attn_d512_fp8_takes(int T);
int ir_launch_flash_attn_d512_fp8(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* o, uint8_t* tiles, int B, int T, int rs, int o_rs,
                                  long qk_bs, long o_bs, float scale, int* ovf_flag, hipStream_t s);

// unet.hip: memory-bound kernels of the ControlLDM path (GroupNorm over any channel count, GEGLU, latent ends, skip rows)
int ir_gn_any_chunks(long HW);
long ir_gn_any_ws_floats(int N, long HW, int C);
int ir_launch_groupnorm_any(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, float* ws, int N, long HW, int C, int G, float eps,
                            int do_silu, hipStream_t s);
int ir_launch_geglu(const bf16_t* ag, bf16_t* out, long rows, int F, hipStream_t s);
int ir_launch_cldm_in(const float* x, const float* hint, bf16_t* out, int N, long HW, hipStream_t s);
int ir_launch_cldm_out(const float* zT, const float* v, int v_cs, float* out, int N, long HW, hipStream_t s);
int ir_launch_copy_rows(const bf16_t* src, int src_cs, const bf16_t* add, int add_cs, bf16_t* dst, int dst_cs, long rows, int C, hipStream_t s);
