// tools/conv_bench.hip -- developer micro-benchmark: times the conv kernel at the shapes of the
// B=32 x 5 s workload (random data), one line per shape.  Build: see tools/build_conv_bench.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include "../quickvc-official_amd/csrc/qvc_kernels.h"
#include "../quickvc-official_amd/csrc/qvc_pack_util.h"
using namespace qvc;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int Cin, M, T, k, dil; int kind; /*0 std f32 in->y16, 1 op in -> res+y32, 2 gau, 3 rs */ };

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  // QVC_BENCH_ZEROS=1: all-zero activations and weights -- same instruction stream and cycles, but the chip holds a
  // higher clock on zero operands (DVFS), so the ratio to the random-data run is the clock given back under load
  const bool zeros = getenv("QVC_BENCH_ZEROS") != nullptr;
  // QVC_BENCH_PAIRS / QVC_BENCH_WN: only the ResBlock pairs / only the WaveNet stack launches (e.g. with many repetitions
  // under tools/power_probe.sh, to see what one kernel family draws)
  const bool only_pairs = getenv("QVC_BENCH_PAIRS") != nullptr, only_wn = getenv("QVC_BENCH_WN") != nullptr;
  std::vector<Shape> shapes = {
    {"s2 c1 k3 d1", 128, 128, 5000, 3, 1, 0}, {"s2 c1 k7 d3", 128, 128, 5000, 7, 3, 0}, {"s2 c1 k11 d5", 128, 128, 5000, 11, 5, 0},
    {"s2 c2 k3", 128, 128, 5000, 3, 1, 1}, {"s2 c2 k11", 128, 128, 5000, 11, 1, 1},
    {"s1 c1 k3 d1", 256, 256, 1250, 3, 1, 0}, {"s1 c1 k11 d5", 256, 256, 1250, 11, 5, 0}, {"s1 c2 k7", 256, 256, 1250, 7, 1, 1},
    {"wn in k5 gau", 192, 384, 250, 5, 1, 2}, {"wn rs 1x1", 192, 384, 250, 1, 1, 3},
    {"conv_pre k7", 192, 512, 250, 7, 1, 0}, {"post k7 M72", 128, 72, 5001, 7, 1, 1},
  };
  hipStream_t st; CK(hipStreamCreate(&st));
  size_t maxel = (size_t)B * 5001 * 512;
  float *x32, *y32, *res, *bb; void *x16, *y16;
  CK(hipMalloc(&x32, maxel * 4)); CK(hipMalloc(&y32, maxel * 4)); CK(hipMalloc(&res, maxel * 4));
  CK(hipMalloc(&x16, maxel * 2)); CK(hipMalloc(&y16, maxel * 2)); CK(hipMalloc(&bb, 1 << 20));
  {
    std::vector<float> h(maxel);
    for (size_t i = 0; i < maxel; ++i) h[i] = zeros ? 0.f : (float)((i * 2654435761u >> 8) & 0xffff) / 32768.f - 1.f;
    CK(hipMemcpy(x32, h.data(), maxel * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(res, h.data(), maxel * 4, hipMemcpyHostToDevice));
    std::vector<uint16_t> hh(maxel);
    for (size_t i = 0; i < maxel; ++i) hh[i] = zeros ? 0 : (uint16_t)(0x3000 + ((i * 40503u) & 0x7ff) + ((i & 1) << 15));   // small f16 values
    CK(hipMemcpy(x16, hh.data(), maxel * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bb, 0, 1 << 20));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (!only_pairs && !only_wn) for (const Shape& s : shapes) {
    ConvDesc d = make_conv(s.M, s.Cin, s.k, s.dil, s.kind == 2);
    d.w_off = 0; d.b_off = align_up(d.w_bytes(), 256);
    size_t wb = d.b_off + d.b_bytes();
    std::vector<char> hw(wb);
    std::vector<float> w((size_t)s.M * s.Cin * s.k), bias(s.M, 0.01f);
    for (size_t i = 0; i < w.size(); ++i) w[i] = zeros ? 0.f : ((float)((i * 1103515245u >> 10) & 0x3ff) / 512.f - 1.f) * 0.05f;
    pack_plain_conv(d, w.data(), bias.data(), QVC_F16, hw.data());
    void* dw; CK(hipMalloc(&dw, wb)); CK(hipMemcpy(dw, hw.data(), wb, hipMemcpyHostToDevice));
    ConvArgs a;
    a.w = dw; a.bias = (const float*)((char*)dw + d.b_off);
    a.T_in = s.T; a.Nq = s.T; a.T_out = s.T;
    int epi = EPI_STD;
    const int64_t bs_in = (int64_t)s.T * s.Cin, bs_out = (int64_t)s.T * s.M;
    if (s.kind == 0) { a.x = x32; a.x_kind = XK_F32_FM; a.x_bs = bs_in; a.x_ts = s.Cin; a.slope_in = 0.1f; a.y16 = y16; a.y16_bs = bs_out; a.y16_ts = s.M; a.slope_out = 0.1f; }
    if (s.kind == 1) { a.x = x16; a.x_kind = XK_OP_FM; a.x_bs = bs_in; a.x_ts = s.Cin; a.res = res; a.res_bs = bs_out; a.res_ts = s.M; a.y32 = y32; a.y32_bs = bs_out; a.y32_ts = s.M; }
    if (s.kind == 2) { a.x = x32; a.x_kind = XK_F32_FM; a.x_bs = bs_in; a.x_ts = s.Cin; a.bbias = bb; a.bbias_bs = 0; a.gau_H = s.M / 2; a.y16 = y16; a.y16_bs = (int64_t)s.T * s.M / 2; a.y16_ts = s.M / 2; epi = EPI_GAU; }
    if (s.kind == 3) { a.x = x16; a.x_kind = XK_OP_FM; a.x_bs = bs_in; a.x_ts = s.Cin; a.split = s.M / 2; a.res = res; a.res_bs = (int64_t)s.T * s.M / 2; a.res_ts = s.M / 2; a.y32 = res; a.y32b = y32; a.y32_bs = (int64_t)s.T * s.M / 2; a.y32_ts = s.M / 2; }
    int nf = 0;
    for (int i = 0; i < 3; ++i) if (launch_conv(d, a, B, epi, QVC_F16, st, &nf) != QVC_OK) { printf("%s: launch failed\n", s.name); break; }
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) launch_conv(d, a, B, epi, QVC_F16, st, &nf);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const double flops = 2.0 * B * s.T * (double)s.M * s.k * s.Cin;
    printf("%-14s MF%d WM%d NF%-2d chunks%d  %8.1f us  %7.1f TF  (%.1f%% of 2.5PF)\n", s.name, d.MF, d.WM, nf, d.nchunk, us, flops / us * 1e-6, flops / us * 1e-6 / 25.0);
    CK(hipFree(dw));
  }
  if (!only_pairs && !only_wn) {  // ---- polyphase up-samplers, conv_pre (weight-heavy, few frames)
    struct U { const char* name; int Cin, Cout, T, k, s, p; int kind; };
    std::vector<U> us = {{"ups0 512>256 s5", 512, 256, 250, 16, 5, 6, 0}, {"ups1 256>128 s4", 256, 128, 1250, 16, 4, 6, 1}};
    for (const U& u : us) {
      ConvDesc d = make_upconv(u.Cin, u.Cout, u.k, u.s, u.p);
      d.w_off = 0; d.b_off = align_up(d.w_bytes(), 256);
      size_t wb = d.b_off + d.b_bytes();
      std::vector<char> hw(wb);
      std::vector<float> w((size_t)d.M * d.Cin * d.taps), bias(d.M, 0.01f);
      for (size_t i = 0; i < w.size(); ++i) w[i] = ((float)((i * 1103515245u >> 10) & 0x3ff) / 512.f - 1.f) * 0.02f;
      pack_plain_conv(d, w.data(), bias.data(), QVC_F16, hw.data());
      void* dw; CK(hipMalloc(&dw, wb)); CK(hipMemcpy(dw, hw.data(), wb, hipMemcpyHostToDevice));
      ConvArgs a;
      a.w = dw; a.bias = (const float*)((char*)dw + d.b_off);
      const int t_out = u.T * u.s;
      a.T_in = u.T; a.Nq = (t_out - 1 + u.p) / u.s + 1; a.T_out = t_out;
      if (u.kind == 0) { a.x = x16; a.x_kind = XK_OP_FM; } else { a.x = x32; a.x_kind = XK_F32_FM; a.slope_in = 0.1f; }
      a.x_bs = (int64_t)u.T * u.Cin; a.x_ts = u.Cin;
      a.y16 = y16; a.y16_bs = (int64_t)t_out * u.Cout; a.y16_ts = u.Cout;
      int nf = 0;
      for (int i = 0; i < 3; ++i) if (launch_conv(d, a, B, EPI_STD, QVC_F16, st, &nf) != QVC_OK) { printf("%s: launch failed\n", u.name); break; }
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) launch_conv(d, a, B, EPI_STD, QVC_F16, st, &nf);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us_ = ms * 1e3 / reps;
      const double flops = 2.0 * B * u.T * (double)u.Cin * u.Cout * u.k;
      uint64_t sum = 0;   // checksum of the output: tile variants (QVC_WIDE_CONV=0/2, QVC_WIDE_NF, QVC_WIDE_CL) must agree bit for bit
      { std::vector<uint16_t> hy((size_t)B * t_out * u.Cout); CK(hipMemcpy(hy.data(), y16, hy.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < hy.size(); ++i) sum = sum * 1000003u + hy[i]; }
      printf("%-16s MF%d WM%d NF%-3d chunks%d %8.1f us  %7.1f TF  sum %016llx\n", u.name, d.MF, d.WM, nf, d.nchunk, us_, flops / us_ * 1e-6, (unsigned long long)sum);
      CK(hipFree(dw));
    }
  }
  if (!only_pairs && !only_wn) {  // ---- fused WaveNet layers: 16 distinct weight sets in sequence (cold weights, as in the real step)
    const int H = 192, T = 250, L = 16;
    ConvDesc din = make_conv(2 * H, H, 5, 1, true), drs = make_conv(2 * H, H, 1, 1, true);
    wn_layout(din, 2); wn_layout(drs, 2);
    din.w_off = 0; drs.w_off = align_up(din.w_bytes(), 256); drs.b_off = drs.w_off + align_up(drs.w_bytes(), 256);
    const size_t per = drs.b_off + align_up(drs.b_bytes(), 256);
    std::vector<char> hw(per * L, 0);
    for (size_t i = 0; i < per * L; i += 2) { hw[i] = (char)((i * 131) & 0x7f); hw[i + 1] = (char)(0x20 + ((i >> 3) & 7)); }   // small f16 values
    void* dw; CK(hipMalloc(&dw, per * L)); CK(hipMemcpy(dw, hw.data(), per * L, hipMemcpyHostToDevice));
    // evict caches between timed loops with a big memset
    WnArgs a; a.bs = (int64_t)T * H; a.T = T; a.H = H; a.HP = din.CinP; a.bbias = bb; a.bbias_bs = 0;
    a.taps = 5; a.KS = din.KS(); a.nIt1 = din.nIt(); a.last = 0; a.oacc = y32;
    int nf = 0;
    auto run = [&]() { for (int l = 0; l < L; ++l) { a.x_in = l % 2 ? res : x32; a.x_out = l % 2 ? x32 : res;
        a.w_in = (char*)dw + per * l; a.w_rs = (char*)dw + per * l + drs.w_off; a.b_rs = (const float*)((char*)dw + per * l + drs.b_off);
        launch_wn(din, a, B, QVC_F16, st, &nf); } };
    run(); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) run();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps / L;
    const double flops = 2.0 * B * T * (double)H * (2.0 * H * 5 + 2.0 * H);
    printf("%-14s W%-2d NF%-2d               %8.1f us/layer  %7.1f TF  (%.1f%% of 2.5PF)\n", "wn layer x16", din.WM, nf, us, flops / us * 1e-6, flops / us * 1e-6 / 25.0);
    CK(hipFree(dw));
  }
  if (!only_pairs) {  // ---- whole-stack WaveNet launches: 4 layers per launch, 8 distinct weight sets (cold weights)
    const int H = 192, T = 250, L = 4, SETS = 8;
    ConvDesc din = make_conv(2 * H, H, 5, 1, true), drs = make_conv(2 * H, H, 1, 1, true);
    wn_layout(din, 2); wn_layout(drs, 2);
    din.w_off = 0; drs.w_off = align_up(din.w_bytes(), 256); drs.b_off = drs.w_off + align_up(drs.w_bytes(), 256);
    const size_t per = drs.b_off + align_up(drs.b_bytes(), 256);
    std::vector<char> hw(per * L * SETS, 0);
    for (size_t i = 0; i < hw.size(); i += 2) { hw[i] = (char)((i * 131) & 0x7f); hw[i + 1] = (char)(0x20 + ((i >> 3) & 7)); }
    void* dw; CK(hipMalloc(&dw, hw.size())); CK(hipMemcpy(dw, hw.data(), hw.size(), hipMemcpyHostToDevice));
    for (int variant = 1; variant >= 0; --variant) {     // debug switch wn_kernel: 1 = the generic stack kernel, 0 = default (continuous-stream kernel where built)
    debug_table()[DBG_WN_KERNEL].store(variant);
    auto run = [&]() {
      for (int sset = 0; sset < SETS; ++sset) {
        WnStackArgs a; a.x0 = x32; a.out = y32; a.bs = (int64_t)T * H; a.T = T; a.H = H; a.HP = din.CinP;
        for (int l = 0; l < L; ++l) { char* base = (char*)dw + per * (sset * L + l);
          a.w_in[l] = base; a.w_rs[l] = base + drs.w_off; a.b_rs[l] = (const float*)(base + drs.b_off); }
        a.bbias = bb; a.bbias_bs = 0; a.layers = L; a.taps = 5; a.KS = din.KS(); a.nIt1 = din.nIt(); a.final_layer = 0; a.x_out = res;
        launch_wn_stack(din, a, B, QVC_F16, st);
      } };
    run(); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) run();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps / (SETS * L);
    const double flops = 2.0 * B * T * (double)H * (2.0 * H * 5 + 2.0 * H);
    printf("%-14s W%-2d L4                 %8.1f us/layer  %7.1f TF\n", variant ? "wn stack" : "wn stack2", din.WM, us, flops / us * 1e-6);
    {   // checksum of the outputs (skip sum + residual stream): the two kernels must agree bit for bit
      std::vector<float> hy((size_t)B * T * H), hx((size_t)B * T * H);
      CK(hipMemcpy(hy.data(), y32, hy.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx.data(), res, hx.size() * 4, hipMemcpyDeviceToHost));
      uint64_t sum = 0; for (size_t i = 0; i < hy.size(); ++i) { uint32_t u, v; memcpy(&u, &hy[i], 4); memcpy(&v, &hx[i], 4); sum = sum * 1000003u + u + 31u * v; }
      printf("               output checksum %016llx\n", (unsigned long long)sum);
    }
#ifdef QVC_STAMP
    {   // phase stamps of one launch (s_memtime cycles; clock from s_memrealtime at 100 MHz)
      const int nwg = ceil_div(T, kWnOutFrames) * B;
      unsigned long long* dst; CK(hipMalloc(&dst, (size_t)nwg * 16 * 32 * 8)); CK(hipMemset(dst, 0, (size_t)nwg * 16 * 32 * 8));
      WnStackArgs a; a.x0 = x32; a.out = y32; a.bs = (int64_t)T * H; a.T = T; a.H = H; a.HP = din.CinP;
      for (int l = 0; l < L; ++l) { char* base = (char*)dw + per * l;
        a.w_in[l] = base; a.w_rs[l] = base + drs.w_off; a.b_rs[l] = (const float*)(base + drs.b_off); }
      a.bbias = bb; a.bbias_bs = 0; a.layers = L; a.taps = 5; a.KS = din.KS(); a.nIt1 = din.nIt(); a.final_layer = 0; a.x_out = res;
      a.stamps = dst;
      for (int i = 0; i < 20; ++i) run();                  // clock at its loaded state
      launch_wn_stack(din, a, B, QVC_F16, st);
      CK(hipStreamSynchronize(st));
      std::vector<unsigned long long> hs((size_t)nwg * 16 * 32);
      CK(hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost));
      static const char* nm[6] = {"gemm1", "gate", "bar1", "gemm2", "epi", "bar2"};
      for (int wg : {0, nwg / 3, nwg - 1}) for (int wv : {0, 5, 11}) {
        const unsigned long long* t = &hs[((size_t)wg * 16 + wv) * 32];
        const double ghz = (double)(t[26] - t[0]) / ((double)(t[28] - t[27]) * 10.0);
        printf("stamps wg %3d wave %2d: total %6llu cyc (%.2f GHz, %.2f us)  prologue %5llu |", wg, wv, t[26] - t[0], ghz, (t[26] - t[0]) / ghz * 1e-3, t[1] - t[0]);
        for (int l = 0; l < 4; ++l) {
          unsigned long long prev = l == 0 ? t[1] : t[7 + 6 * (l - 1)];
          printf(" L%d:", l);
          for (int i = 0; i < 6; ++i) { printf(" %s %llu", nm[i], t[2 + 6 * l + i] - prev); prev = t[2 + 6 * l + i]; }
          printf(" |");
        }
        printf(" tail %llu\n", t[26] - t[25]);
      }
      {   // wave start times (s_memrealtime, 100 MHz) relative to the launch's earliest wave: launch skew inside a workgroup and over the grid
        unsigned long long tmin = ~0ull, tmaxend = 0;
        for (int wg = 0; wg < nwg; ++wg) for (int wv = 0; wv < 12; ++wv) { const unsigned long long* t = &hs[((size_t)wg * 16 + wv) * 32]; if (t[27] && t[27] < tmin) tmin = t[27]; if (t[28] > tmaxend) tmaxend = t[28]; }
        printf("launch: first wave start -> last wave end %.2f us\n", (double)(tmaxend - tmin) * 0.01);
        for (int wg : {0, 1, 8, nwg / 3, nwg - 1}) {
          printf("start offsets wg %3d (us):", wg);
          for (int wv = 0; wv < 12; ++wv) printf(" %.2f", (double)(hs[((size_t)wg * 16 + wv) * 32 + 27] - tmin) * 0.01);
          printf("  | end:");
          for (int wv = 0; wv < 12; wv += 11) printf(" %.2f", (double)(hs[((size_t)wg * 16 + wv) * 32 + 28] - tmin) * 0.01);
          printf("\n");
        }
      }
      // mean over all workgroups and waves of each phase
      double sum[26] = {0}; int cnt = 0;
      for (int wg = 0; wg < nwg; ++wg) for (int wv = 0; wv < 12; ++wv) {
        const unsigned long long* t = &hs[((size_t)wg * 16 + wv) * 32];
        if (!t[26]) continue;
        ++cnt;
        sum[0] += (double)(t[1] - t[0]);
        for (int l = 0; l < 4; ++l) { unsigned long long prev = l == 0 ? t[1] : t[7 + 6 * (l - 1)];
          for (int i = 0; i < 6; ++i) { sum[1 + 6 * l + i] += (double)(t[2 + 6 * l + i] - prev); prev = t[2 + 6 * l + i]; } }
        sum[25] += (double)(t[26] - t[25]);
      }
      printf("stamps mean over %d waves: prologue %.0f", cnt, sum[0] / cnt);
      for (int l = 0; l < 4; ++l) { printf(" | L%d", l); for (int i = 0; i < 6; ++i) printf(" %s %.0f", nm[i], sum[1 + 6 * l + i] / cnt); }
      printf(" | tail %.0f\n", sum[25] / cnt);
      CK(hipFree(dst));
    }
#endif
    }
    CK(hipFree(dw));
  }
  // ---- fused ResBlock pairs: one launch per chain, and pair q of the three chains (k 3 / 7 / 11) as ONE launch
  struct Stage { const char* name; int C, T; };
  if (!only_wn) for (const Stage& sg : {Stage{"s2", 128, 5000}, Stage{"s1", 256, 1250}}) {
    const int ks[3] = {11, 7, 3}, dils[3] = {1, 3, 5};
    for (int q = 0; q < 3; ++q) {
      ConvDesc d1s[3], d2s[3]; PairArgs3 a3; a3.n = 3;
      void* dws[3];
      double tot_us = 0, tot_fl = 0;
      for (int c = 0; c < 3; ++c) {
        ConvDesc d1 = make_conv(sg.C, sg.C, ks[c], dils[q]), d2 = make_conv(sg.C, sg.C, ks[c], 1);
        pair_layout(d1, d2);
        d1.w_off = 0; d1.b_off = align_up(d1.w_bytes(), 256);
        d2.w_off = 0; d2.b_off = d1.b_off;
        size_t wb = d1.b_off + d1.b_bytes();
        std::vector<char> hw(wb);
        std::vector<float> w((size_t)sg.C * sg.C * ks[c]), bias(sg.C, 0.01f);
        for (size_t i = 0; i < w.size(); ++i) w[i] = zeros ? 0.f : ((float)((i * 1103515245u >> 10) & 0x3ff) / 512.f - 1.f) * 0.05f;
        pack_plain_conv(d1, w.data(), bias.data(), QVC_F16, hw.data());
        CK(hipMalloc(&dws[c], wb)); CK(hipMemcpy(dws[c], hw.data(), wb, hipMemcpyHostToDevice));
        PairArgs a;
        // chain c reads / writes its own slice of the big buffers (as the three ResBlocks do)
        const size_t slice = (size_t)B * sg.T * sg.C;
        a.x = (char*)x16 + c * slice * 2; a.bs = (int64_t)sg.T * sg.C; a.T = sg.T; a.C = sg.C; a.CP = d1.CinP;
        a.w1 = dws[c]; a.b1 = (const float*)((char*)dws[c] + d1.b_off); a.w2 = dws[c]; a.b2 = a.b1;
        a.k = ks[c]; a.dil = dils[q]; a.KS = d1.KS(); a.nIt = d1.nIt(); a.y = (char*)y16 + c * slice * 2;
        d1s[c] = d1; d2s[c] = d2; a3.p[c] = a;
        int nf = 0;
        for (int i = 0; i < 3; ++i) if (launch_pair(d1, d2, a, B, QVC_F16, st, &nf) != QVC_OK) { printf("%s: launch failed\n", sg.name); break; }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) launch_pair(d1, d2, a, B, QVC_F16, st, &nf);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        const double flops = 2.0 * 2.0 * B * sg.T * (double)sg.C * ks[c] * sg.C;
        tot_us += us; tot_fl += flops;
        printf("pair %s k%-2d d%d  MF%d WM%d NF%-2d          %8.1f us  %7.1f TF  (%.1f%% of 2.5PF)\n", sg.name, ks[c], dils[q], d1.MF, d1.WM, nf, us, flops / us * 1e-6, flops / us * 1e-6 / 25.0);
      }
      // one launch for the three chains: chains interleaved on the CUs (x % n) vs chain-major grid (longest chain first)
      for (int cm = 0; cm < 2; ++cm) {
        a3.chain_major = cm;
        int nf = 0;
        for (int i = 0; i < 3; ++i) if (launch_pair3(d1s, d2s, a3, B, QVC_F16, st, &nf) != QVC_OK) { printf("%s: launch3 failed\n", sg.name); break; }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) launch_pair3(d1s, d2s, a3, B, QVC_F16, st, &nf);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("pair3 %s d%d  %-12s NF%-3d %8.1f us  %7.1f TF\n", sg.name, dils[q], cm ? "chain-major" : "interleaved", nf, ms * 1e3 / reps, tot_fl / (ms * 1e3 / reps) * 1e-6);
#ifdef QVC_STAMP
        if (cm == 1 && q == 0) {   // phase stamps of one chain-major launch (s_memtime cycles), wave 0 of every workgroup
          const int NT = nf * 16 * (d1s[0].WM == block_waves(d1s[0]) ? 1 : block_waves(d1s[0]) / d1s[0].WM);
          const int tiles = ceil_div(sg.T, NT);
          const size_t nwg = (size_t)tiles * B * 3;
          unsigned long long* dst; CK(hipMalloc(&dst, nwg * 64 * 8)); CK(hipMemset(dst, 0, nwg * 64 * 8));
          a3.stamps = dst;
          launch_pair3(d1s, d2s, a3, B, QVC_F16, st, &nf);
          CK(hipStreamSynchronize(st));
          a3.stamps = nullptr;
          std::vector<unsigned long long> hs(nwg * 64);
          CK(hipMemcpy(hs.data(), dst, nwg * 64 * 8, hipMemcpyDeviceToHost));
          CK(hipFree(dst));
          unsigned long long t0 = ~0ull, t1 = 0;
          for (size_t w = 0; w < nwg; ++w) if (hs[w * 64]) { t0 = std::min(t0, hs[w * 64]); t1 = std::max(t1, hs[w * 64 + 6]); }
          static const char* ph[6] = {"stage", "gemm1", "bar", "write", "gemm2", "epilogue"};
          for (int c = 0; c < 3; ++c) {
            double sum[6] = {0, 0, 0, 0, 0, 0}, tot = 0, start = 0; size_t cnt = 0;
            for (size_t w = (size_t)c * tiles * B; w < (size_t)(c + 1) * tiles * B; ++w) {
              const unsigned long long* s8 = &hs[w * 64];
              if (!s8[0] || !s8[6]) continue;
              for (int p = 0; p < 6; ++p) sum[p] += (double)(s8[p + 1] - s8[p]);
              tot += (double)(s8[6] - s8[0]); start += (double)(s8[0] - t0); ++cnt;
            }
            printf("stamps %s k%-2d: %zu workgroups, mean %.0f cycles |", sg.name, ks[c], cnt, tot / cnt);
            for (int p = 0; p < 6; ++p) printf(" %s %.0f", ph[p], sum[p] / cnt);
            double i8 = 0, i9 = 0, i10 = 0;
            for (size_t w = (size_t)c * tiles * B; w < (size_t)(c + 1) * tiles * B; ++w) {
              const unsigned long long* s8 = &hs[w * 64];
              if (!s8[0] || !s8[6] || !s8[8]) continue;
              i8 += (double)(s8[8] - s8[0]); i9 += (double)(s8[9] - s8[8]); i10 += (double)(s8[10] - s8[9]);
            }
            {
              double cyc = 0, rt = 0;
              for (size_t w = (size_t)c * tiles * B; w < (size_t)(c + 1) * tiles * B; ++w) {
                const unsigned long long* s8 = &hs[w * 64];
                if (!s8[0] || !s8[6] || !s8[12]) continue;
                cyc += (double)(s8[6] - s8[0]); rt += (double)(s8[12] - s8[11]);
              }
              printf(" | clock %.2f GHz", rt > 0 ? cyc / rt * 0.1 : 0.0);
            }
            printf(" | inside stage: loads issued %.0f, first data +%.0f, converted + written +%.0f, barrier +%.0f\n", i8 / cnt, i9 / cnt, i10 / cnt,
                   (sum[0] - i8 - i9 - i10) / cnt);
          }
          printf("stamps %s: launch span %llu cycles\n", sg.name, t1 - t0);
        }
#endif
      }
      printf("pair3 %s d%d  three launches %8.1f us  %7.1f TF\n", sg.name, dils[q], tot_us, tot_fl / tot_us * 1e-6);
      for (int c = 0; c < 3; ++c) CK(hipFree(dws[c]));
    }
  }
  return 0;
}
