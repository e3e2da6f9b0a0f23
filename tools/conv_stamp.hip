// Diagnostic: phase time stamps of conv_halo_kernel (prologue / main loop / epilogue per block) and the in-kernel clock.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=fast -DIR_STAMPS -Iinstarevive_amd/csrc tools/conv_stamp.hip -o tools/conv_stamp
// Run:    tools/conv_stamp [H W Cin Cout]
#include "../instarevive_amd/csrc/igemm.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 2048, W = argc > 2 ? atoi(argv[2]) : 2048;
    const int Cin = argc > 3 ? atoi(argv[3]) : 128, Cout = argc > 4 ? atoi(argv[4]) : 128;
    const size_t nin = (size_t)H * W * Cin, nout = (size_t)H * W * Cout, nw = (size_t)Cout * 9 * Cin;
    std::vector<bf16_t> hin(nin), hw(nw);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (bf16_t)(0x3c00 + ((s >> 16) & 0x3ff) - ((s >> 9) & 0x8000 ? 0x8000 : 0)); };
    for (auto& v : hin) v = rnd();
    for (auto& v : hw) v = (bf16_t)((rnd() & 0x83ff) | 0x3800);
    bf16_t *din, *dw, *dout;
    float* dbias;
    CK(hipMalloc(&din, nin * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dout, nout * 2)); CK(hipMalloc(&dbias, Cout * 4));
    CK(hipMemcpy(din, hin.data(), nin * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dbias, 0, Cout * 4));
    IGemmParams p{};
    p.in = din; p.NB = 1; p.H = H; p.W = W; p.Cin = Cin; p.in_cs = Cin; p.Ho = H; p.Wo = W; p.taps = 9; p.stride = 1; p.pad = 1;
    p.wgt = dw; p.wgt_rs = 9 * Cin; p.Cout = Cout; p.Cout_pad = Cout; p.M = H * W; p.bias = dbias; p.act = IR_ACT_NONE; p.out_scale = 1.f;
    p.out = dout; p.out_cs = Cout;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 200; ++i) { int rc = ir_launch_igemm(p, st); if (rc) { printf("launch rc %d\n", rc); return 1; } }  // warm the clocks
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 10; ++i) ir_launch_igemm(p, st);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("conv %dx%d %d->%d: %.3f ms  %.1f TFLOP/s\n", H, W, Cin, Cout, ms, 2.0 * H * W * Cout * 9 * Cin / ms / 1e9);
    if (!getenv("IR_NO_CONV_PP") && Cout % 128 == 0) {  // ping-pong kernel: per-segment cycle sums of waves 0 and 4 of every workgroup
        const long nb = std::min<long>(32768, (long)((H + 15) / 16) * ((W + 15) / 16) * (Cout / 128));
        std::vector<unsigned long long> st(nb * 16);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), nb * 16 * 8));
        const double nsteps = 9.0 * (Cin / 64);
        for (int g = 0; g < 2; ++g) {
            double a[4] = {0, 0, 0, 0};
            for (long b = 0; b < nb; ++b)
                for (int k = 0; k < 4; ++k) a[k] += (double)st[(b * 2 + g) * 8 + k];
            printf("  waves %d-%d per step (s_memtime ticks): matrix %.0f, wait+barrier %.0f, vector %.0f, barrier %.0f (sum %.0f)\n", g * 4, g * 4 + 3,
                   a[0] / nb / nsteps, a[1] / nb / nsteps, a[2] / nb / nsteps, a[3] / nb / nsteps, (a[0] + a[1] + a[2] + a[3]) / nb / nsteps);
        }
        return 0;
    }
    const long nblk = std::min<long>(65536, (long)((H + 7) / 8) * ((W + 15) / 16) * (Cout / (Cout % 128 == 0 ? 128 : 64)));
    std::vector<unsigned long long> st8(nblk * 8), st4(nblk * 4);
    CK(hipMemcpyFromSymbol(st8.data(), HIP_SYMBOL(g_stamps), nblk * 8 * 8));
    double clk = 0;
    for (long b = 0; b < nblk; ++b) {
        for (int k = 0; k < 4; ++k) st4[b * 4 + k] = st8[b * 8 + k];
        clk += (double)(st8[b * 8 + 6] - st8[b * 8 + 5]) / (double)(st8[b * 8 + 2] - st8[b * 8 + 1]) * 100.0;  // MHz over the main loop
    }
    printf("in-kernel clock over the main loop: %.0f MHz (KO=%d)\n", clk / nblk, IR_KO);
    double d[3] = {0, 0, 0};
    unsigned long long tmin = ~0ull, tmax = 0;
    for (long b = 0; b < nblk; ++b) {
        for (int k = 0; k < 3; ++k) d[k] += (double)(st4[b * 4 + k + 1] - st4[b * 4 + k]);
        tmin = std::min(tmin, st4[b * 4]); tmax = std::max(tmax, st4[b * 4 + 3]);
    }
    // s_memrealtime ticks at a constant 100 MHz; report microseconds
    printf("blocks %ld: prologue %.2f us, main loop %.2f us, epilogue %.2f us (s_memrealtime, 100 MHz); span %.1f us\n", nblk, d[0] / nblk / 100.0,
           d[1] / nblk / 100.0, d[2] / nblk / 100.0, (tmax - tmin) / 100.0);
    // distribution of the main-loop time
    std::vector<double> ml(nblk), ep(nblk), pr(nblk);
    for (long b = 0; b < nblk; ++b) { pr[b] = st4[b * 4 + 1] - st4[b * 4]; ml[b] = st4[b * 4 + 2] - st4[b * 4 + 1]; ep[b] = st4[b * 4 + 3] - st4[b * 4 + 2]; }
    auto pct = [&](std::vector<double>& v, const char* nm) {
        std::sort(v.begin(), v.end());
        printf("  %s: p10 %.2f p50 %.2f p90 %.2f p99 %.2f us\n", nm, v[nblk / 10] / 100, v[nblk / 2] / 100, v[nblk * 9 / 10] / 100, v[nblk * 99 / 100] / 100);
    };
    pct(pr, "prologue"); pct(ml, "main"); pct(ep, "epilogue");
    return 0;
}
