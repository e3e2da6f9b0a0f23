// Diagnostic: flash_attn_kernel<72> timing with knock-outs (-DIR_KO_ATTN=n, see attention.hip) on the DiT self-attention shape.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=fast -DIR_KO_ATTN=0 -Iinstarevive_amd/csrc tools/attn_stamp.hip -o tools/attn_stamp_ko0
#include "../instarevive_amd/csrc/attention.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 16384, H = 16, D = 72, DV = 96;
    const size_t n = (size_t)T * H * D, nvt = (size_t)H * DV * T;
    std::vector<bf16_t> hq(n), hk(n), hv(nvt);
    unsigned s = 777;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (bf16_t)(0x3c00 + ((s >> 16) & 0x3ff) - ((s >> 9) & 0x8000 ? 0x8000 : 0)); };
    for (auto& v : hq) v = rnd();
    for (auto& v : hk) v = rnd();
    for (auto& v : hv) v = rnd();
    bf16_t *dq, *dk, *dvt, *dout;
    CK(hipMalloc(&dq, n * 2)); CK(hipMalloc(&dk, n * 2)); CK(hipMalloc(&dvt, nvt * 2)); CK(hipMalloc(&dout, n * 2));
    CK(hipMemcpy(dq, hq.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dk, hk.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dvt, hv.data(), nvt * 2, hipMemcpyHostToDevice));
    AttnParams p{};
    p.q = dq; p.k = dk; p.vt = dvt; p.o = dout;
    p.q_bs = p.k_bs = p.o_bs = (long)n; p.vt_bs = (long)nvt;
    p.q_rs = p.k_rs = p.o_rs = H * D; p.q_hs = p.k_hs = p.o_hs = D;
    p.B = 1; p.Hh = H; p.Tq = T; p.Tk = T; p.Tk_pad = T; p.D = D;
    p.scale_log2 = 0.117851f * 1.442695f;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 100; ++i) { int rc = ir_launch_flash_attn(p, st); if (rc) { printf("launch rc %d\n", rc); return 1; } }
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 10; ++i) ir_launch_flash_attn(p, st);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("KO=%d flash d72 T=%d: %.3f ms  %.1f TFLOP/s\n", IR_KO_ATTN, T, ms, 4.0 * H * (double)T * T * D / ms / 1e9);
#ifdef IR_STAMPS_ATTN
    {
        const int nblk = std::min(4096, (T / 256) * H);
        std::vector<unsigned long long> st((size_t)nblk * 32);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_attn_stamps), st.size() * 8));
        const double tiles = T / 64.0;
        for (int g = 0; g < 2; ++g) {
            double a[4] = {0, 0, 0, 0};
            for (int bl = 0; bl < nblk; ++bl)
                for (int w = 0; w < 4; ++w)
                    for (int k = 0; k < 4; ++k) a[k] += (double)st[((size_t)bl * 8 + g * 4 + w) * 4 + k];
            for (int k = 0; k < 4; ++k) a[k] /= (double)nblk * 4 * tiles;
            printf("  waves %d-%d per tile (s_memtime ticks): vector %.0f, barrier wait %.0f, matrix %.0f, barrier wait %.0f  (sum %.0f)\n", g * 4, g * 4 + 3,
                   a[0], a[1], a[2], a[3], a[0] + a[1] + a[2] + a[3]);
        }
    }
#endif
    return 0;
}
