// qvc_emu.cpp -- TEST INFRASTRUCTURE: a CPU replay of the product's launch sequence.
//
// Built by __graft_entry__.build() into oracle/_build/libqvc_emu.so and loaded only by tests.
// It instantiates the product's Path<> (quickvc-official_amd/csrc/qvc_path.h) with a backend whose
// "kernels" are plain scalar loops that consume the SAME packed weight blob through the SAME
// fragment index arithmetic as the gfx950 kernels, and that round operands to the MFMA operand
// type at the same places.  Purpose: check on a machine without a GPU that the packer
// (weight-norm fold, flip folding, polyphase rewrite, fragment order) and the orchestration
// (buffers, strides, fused epilogues) reproduce the oracle, so that what is left to verify on
// the MI355X is each kernel against this emulation's per-launch semantics.
// The product library never links or calls this file.
#include <cmath>
#include <cstring>
#include <vector>
#include "../quickvc-official_amd/csrc/qvc_path.h"
#include "../quickvc-official_amd/csrc/qvc_stream.h"

namespace {
using namespace qvc;

float from_bf16(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }
float from_f16(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  const int exp = (h >> 10) & 0x1f;
  const uint32_t man = h & 0x3ffu;
  float v;
  if (exp == 0) v = std::ldexp((float)man, -24);
  else if (exp == 31) v = man ? NAN : INFINITY;
  else v = std::ldexp((float)(man | 0x400u), exp - 25);
  return sign ? -v : v;
}
uint16_t to_bf16(float f) {
  uint32_t u; std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
uint16_t to_f16(float f) {   // round-to-nearest-even, saturating (as the device conversion does)
  if (std::isnan(f)) return 0x7e00;
  float a = std::fabs(f);
  uint16_t sign = std::signbit(f) ? 0x8000 : 0;
  if (a >= 65520.f) return sign | 0x7bff;
  if (a < 5.9604644775390625e-08f * 0.5f) return sign;
  int e; float m = std::frexp(a, &e);          // a = m * 2^e, m in [0.5,1)
  int exp = e - 1;                              // a = (2m) * 2^exp
  if (exp < -14) {
    float q = a / 5.9604644775390625e-08f;      // units of 2^-24
    float r = std::nearbyint(q);
    return sign | (uint16_t)r;
  }
  float q = (2.f * m - 1.f) * 1024.f;
  float r = std::nearbyint(q);
  uint32_t man = (uint32_t)r, ex = (uint32_t)(exp + 15);
  if (man == 1024) { man = 0; ++ex; }
  return sign | (uint16_t)((ex << 10) | man);
}
float round_op(float f, int dtype) {
  if (dtype == QVC_F16) { f = std::fmin(std::fmax(f, -65504.f), 65504.f); return from_f16(to_f16(f)); }
  return from_bf16(to_bf16(f));
}
float lrelu(float x, float s) { return x > 0.f ? x : x * s; }

struct EmuBackend {
  // accumulator row of packed position (chunk, wave, mf, i): the physical position, except for lane-packed
  // layouts (ConvDesc::lp), whose standard epilogue addresses rows by channel
  static int row_of(const ConvDesc& d, int chunk, int wave, int mf, int i) {
    return d.lp ? conv_row(d, chunk, wave, mf, i) : ((chunk * d.WM + wave) * d.MF + mf) * 16 + i;
  }
  void fork(int) {} void branch(int) {} void branch_done(int) {} void wait_branch_done(int) {} void join(int) {}
  // ---- conv: dense weights are recovered from the fragment stream with the kernel's index math
  int conv(const ConvDesc& d, const ConvArgs& a, int B, int epi, int dtype) {
    const int KS = d.KS(), nIt = d.nIt(), MP = d.MP();
    std::vector<float> W((size_t)MP * nIt * kKStep, 0.f);   // [packed row][it][32]
    const uint16_t* src = static_cast<const uint16_t*>(a.w);
    for (int chunk = 0; chunk < d.nchunk; ++chunk)
      for (int wave = 0; wave < d.WM; ++wave)
        for (int it = 0; it < nIt; ++it)
          for (int mf = 0; mf < d.MF; ++mf)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 8; ++j) {
                const size_t frag = ((size_t)(chunk * d.WM + wave) * nIt + it) * d.MF + mf;
                const uint16_t h = src[(frag * 64 + lane) * 8 + j];
                const int prow = row_of(d, chunk, wave, mf, lane & 15);
                if (prow < 0) continue;
                const int k = (lane >> 4) * 8 + j;
                W[((size_t)prow * nIt + it) * kKStep + k] = dtype == QVC_F16 ? from_f16(h) : from_bf16(h);
              }
    const int R_halo = (d.taps - 1) * d.dil;
    if (acc_store.size() < (size_t)MP * (size_t)a.Nq) acc_store.resize((size_t)MP * (size_t)a.Nq);
    acc_ = acc_store.data();
    for (int b = 0; b < B; ++b) {
      const int Tin = ragged_len(a.rg, b, a.T_in), Tlo = ragged_lo(a.rg, b);   // the conv zero-pads at the utterance's own ends
      // staged input, rounded like the LDS tile: rows t in [-left, Nq - left + halo)
      const int rows = a.Nq + R_halo;
      std::vector<float> X((size_t)rows * d.CinP, 0.f);
      for (int r = 0; r < rows; ++r) {
        const int ti = r - d.left;
        for (int c = 0; c < d.Cin; ++c) {
          float v = 0.f; bool ok;
          if (a.x_kind == XK_F32_FM) {
            int s;
            if (a.reflect) { ok = ti >= Tlo && ti <= Tin; s = ti == Tlo ? Tlo + 1 : ti - 1; } else { ok = ti >= Tlo && ti < Tin; s = ti; }
            if (ok) v = round_op(lrelu(static_cast<const float*>(a.x)[(size_t)b * a.x_bs + (size_t)s * a.x_ts + a.x_c0 + c], a.slope_in), dtype);
          } else if (a.x_kind == XK_OP_FM) {
            ok = ti >= Tlo && ti < Tin;
            if (a.x2) {   // MRF mean taken on the fly (reflect applies as in the fp32 path)
              int s2; bool ok2;
              if (a.reflect) { ok2 = ti >= Tlo && ti <= Tin; s2 = ti == Tlo ? Tlo + 1 : ti - 1; } else { ok2 = ok; s2 = ti; }
              if (ok2) {
                const size_t o = (size_t)b * a.x_bs + (size_t)s2 * a.x_ts + a.x_c0 + c;
                auto get = [&](const void* p) { const uint16_t h = static_cast<const uint16_t*>(p)[o]; return dtype == QVC_F16 ? from_f16(h) : from_bf16(h); };
                v = round_op(lrelu((get(a.x) + get(a.x2) + get(a.x3)) * (1.f / 3.f), a.slope_in), dtype);
              }
            } else if (ok) { uint16_t h = static_cast<const uint16_t*>(a.x)[(size_t)b * a.x_bs + (size_t)ti * a.x_ts + a.x_c0 + c]; v = dtype == QVC_F16 ? from_f16(h) : from_bf16(h); if (a.slope_in != 1.f) v = round_op(lrelu(v, a.slope_in), dtype); }
          } else {
            ok = ti >= Tlo && ti < Tin;
            if (ok) v = round_op(lrelu(static_cast<const float*>(a.x)[(size_t)b * a.x_bs + (size_t)c * a.x_ts + ti], a.slope_in), dtype);
          }
          X[(size_t)r * d.CinP + c] = v;
        }
      }
      for (int chunk = 0; chunk < d.nchunk; ++chunk)
        for (int wave = 0; wave < d.WM; ++wave)
          for (int mf = 0; mf < d.MF; ++mf)
            for (int i = 0; i < 16; ++i) {
              const int prow = row_of(d, chunk, wave, mf, i);
              if (prow < 0) continue;
              for (int q = 0; q < a.Nq; ++q) {
                double acc = 0.0;
                for (int it = 0; it < nIt; ++it) {
                  const int tap = it / KS, ks = it % KS;
                  const float* xr = &X[(size_t)(q + tap * d.dil) * d.CinP + ks * kKStep];
                  const float* wr = &W[((size_t)prow * nIt + it) * kKStep];
                  for (int k = 0; k < kKStep; ++k) acc += (double)wr[k] * xr[k];
                }
                acc_[idx(prow, q, a.Nq)] = (float)acc;
              }
            }
      epilogue(d, a, b, epi, dtype);
    }
    return QVC_OK;
  }
  std::vector<float> acc_store;
  float* acc_ = nullptr;
  static size_t idx(int prow, int q, int Nq) { return (size_t)prow * Nq + q; }

  void epilogue(const ConvDesc& d, const ConvArgs& a, int b, int epi, int dtype) {
    auto store16 = [&](void* base, size_t off, float v) {
      static_cast<uint16_t*>(base)[off] = dtype == QVC_F16 ? to_f16(std::fmin(std::fmax(v, -65504.f), 65504.f)) : to_bf16(v);
    };
    if (epi == EPI_GAU) {
      const int H = a.gau_H, hf = d.MF / 2;
      for (int chunk = 0; chunk < d.nchunk; ++chunk)
        for (int wave = 0; wave < d.WM; ++wave)
         for (int f = 0; f < hf; ++f)
          for (int i = 0; i < 16; ++i) {
            const int ch = ((chunk * d.WM + wave) * hf + f) * 16 + i;
            if (ch >= H) continue;
            const int pt = ((chunk * d.WM + wave) * d.MF + f) * 16 + i, ps = ((chunk * d.WM + wave) * d.MF + hf + f) * 16 + i;
            const float* bb = a.bbias + (size_t)b * a.bbias_bs;
            for (int q = 0; q < a.Nq; ++q) {
              const float t = acc_[idx(pt, q, a.Nq)] + bb[ch], s = acc_[idx(ps, q, a.Nq)] + bb[H + ch];
              const float act = std::tanh(t) * (1.f / (1.f + std::exp(-s)));
              store16(a.y16, (size_t)b * a.y16_bs + (size_t)q * a.y16_ts + ch, act);
            }
          }
      return;
    }
    if (epi == EPI_SAMPLE) {   // rows [mu | log sigma] paired like the gate rows; z = mu + noise * exp(log sigma) (models.py:93-94)
      const int C = a.gau_H, hf = d.MF / 2;
      for (int chunk = 0; chunk < d.nchunk; ++chunk)
        for (int wave = 0; wave < d.WM; ++wave)
         for (int f = 0; f < hf; ++f)
          for (int i = 0; i < 16; ++i) {
            const int ch = ((chunk * d.WM + wave) * hf + f) * 16 + i;
            if (ch >= C) continue;
            const int pm = ((chunk * d.WM + wave) * d.MF + f) * 16 + i, pl = ((chunk * d.WM + wave) * d.MF + hf + f) * 16 + i;
            for (int q = 0; q < a.Nq; ++q) {
              const float mu = acc_[idx(pm, q, a.Nq)] + a.bias[ch], ls = acc_[idx(pl, q, a.Nq)] + a.bias[C + ch];
              a.y32[(size_t)b * a.y32_bs + (size_t)q * a.y32_ts + ch] = mu + a.noise[(size_t)b * a.noise_bs + (size_t)ch * a.noise_ts + q] * std::exp(ls);
            }
          }
      return;
    }
    for (int v = 0; v < d.M; ++v) {
      const int ph = d.up_s > 1 ? v / d.Cout : 0, co = d.up_s > 1 ? v % d.Cout : v;
      float bias = a.bias ? a.bias[v] : 0.f;
      if (a.bbias) bias += a.bbias[(size_t)b * a.bbias_bs + v];
      for (int q = 0; q < a.Nq; ++q) {
        const int o = q * d.up_s + ph - d.up_p;
        if (o < 0 || o >= a.T_out) continue;
        float val = acc_[idx(v, q, a.Nq)] + bias;
        if (a.y32b && v >= a.split) { a.y32b[(size_t)b * a.y32_bs + (size_t)o * a.y32_ts + (v - a.split)] += val; continue; }
        if (a.res) val = a.res[(size_t)b * a.res_bs + (size_t)o * a.res_ts + a.res_c0 + co] + a.res_sign * val;
        if (a.res16) { const uint16_t h = static_cast<const uint16_t*>(a.res16)[(size_t)b * a.res_bs + (size_t)o * a.res_ts + co]; val = (dtype == QVC_F16 ? from_f16(h) : from_bf16(h)) + a.res_sign * val; }
        if (a.y32) {
          float* p = a.y32 + (size_t)b * a.y32_bs + (size_t)o * a.y32_ts + a.y32_c0 + co;
          *p = (a.y_accum ? *p : 0.f) + val * a.y_scale;
        }
        if (a.y16) store16(a.y16, (size_t)b * a.y16_bs + (size_t)o * a.y16_ts + co, lrelu(val, a.slope_out));
      }
    }
  }

  // dense natural-order weights [M][nIt][32] recovered from a fragment stream (any row permutation)
  static std::vector<float> dense(const ConvDesc& d, const void* w, int dtype) {
    const int nIt = d.nIt();
    std::vector<float> W((size_t)d.M * nIt * kKStep, 0.f);
    const uint16_t* src = static_cast<const uint16_t*>(w);
    for (int chunk = 0; chunk < d.nchunk; ++chunk)
      for (int wave = 0; wave < d.WM; ++wave)
        for (int it = 0; it < nIt; ++it)
          for (int mf = 0; mf < d.MF; ++mf)
            for (int lane = 0; lane < 64; ++lane) {
              const int row = conv_row(d, chunk, wave, mf, lane & 15);
              if (row < 0) continue;
              for (int j = 0; j < 8; ++j) {
                const size_t frag = ((size_t)(chunk * d.WM + wave) * nIt + it) * d.MF + mf;
                const uint16_t h = src[(frag * 64 + lane) * 8 + j];
                W[((size_t)row * nIt + it) * kKStep + (lane >> 4) * 8 + j] = dtype == QVC_F16 ? from_f16(h) : from_bf16(h);
              }
            }
    return W;
  }
  // fused WaveNet layer, evaluated frame by frame in natural channel order
  int wn(const ConvDesc& din, const ConvDesc& drs, const WnArgs& a, int B, int dtype) {
    const std::vector<float> W1 = dense(din, a.w_in, dtype), W2 = dense(drs, a.w_rs, dtype);
    const int H = a.H, HP = din.CinP, KS = din.KS(), nIt1 = din.nIt(), left = (din.taps - 1) / 2;
    std::vector<float> xr((size_t)(a.T + din.taps) * HP), acts(HP), pre(2 * H);
    for (int b = 0; b < B; ++b) {
      const int Tb = ragged_len(a.rg, b, a.T), Tlo = ragged_lo(a.rg, b);
      std::fill(xr.begin(), xr.end(), 0.f);
      for (int t = Tlo; t < Tb; ++t)
        for (int c = 0; c < H; ++c) xr[(size_t)(t + left) * HP + c] = round_op(a.x_in[(size_t)b * a.bs + (size_t)t * H + c], dtype);
      for (int t = Tlo; t < Tb; ++t) {
        for (int v = 0; v < 2 * H; ++v) {
          double acc = 0;
          for (int it = 0; it < nIt1; ++it) {
            const int tap = it / KS, ks = it % KS;
            const float* xrow = &xr[(size_t)(t + tap) * HP + ks * kKStep];
            const float* wr = &W1[((size_t)v * nIt1 + it) * kKStep];
            for (int k = 0; k < kKStep; ++k) acc += (double)wr[k] * xrow[k];
          }
          pre[v] = (float)acc + a.bbias[(size_t)b * a.bbias_bs + v];
        }
        std::fill(acts.begin(), acts.end(), 0.f);
        for (int c = 0; c < H; ++c) acts[c] = round_op(std::tanh(pre[c]) * (1.f / (1.f + std::exp(-pre[H + c]))), dtype);
        for (int v = 0; v < drs.M; ++v) {
          double acc = 0;
          for (int it = 0; it < KS; ++it)
            for (int k = 0; k < kKStep; ++k) acc += (double)W2[((size_t)v * KS + it) * kKStep + k] * acts[it * kKStep + k];
          const float val = (float)acc + a.b_rs[v];
          const size_t off = (size_t)b * a.bs + (size_t)t * H;
          if (!a.last && v < H) a.x_out[off + v] = a.x_in[off + v] + val;
          else a.oacc[off + (a.last ? v : v - H)] += val;
        }
      }
    }
    return QVC_OK;
  }
  bool use_wn_stack(int, int) const { return true; }
  int wn_stack_chunk(int layers) const { return layers % 4 == 0 ? 4 : layers; }
  // whole stack = the layers one after the other (x ping-pong in temporaries)
  int wn_stack(const ConvDesc& din, const ConvDesc& drs, const ConvDesc& drs_last, const WnStackArgs& s, int B, int dtype,
               const ConvDesc* dpre, const ConvDesc* dpost) {
    auto fill = [](ConvArgs& a, const ConvDesc& d) {
      a.Cin = d.Cin; a.CinP = d.CinP; a.taps = d.taps; a.dil = d.dil; a.left = d.left; a.KS = d.KS(); a.nIt = d.nIt();
      a.nchunk = d.nchunk; a.M = d.M; a.up_s = d.up_s; a.up_p = d.up_p; a.Cout = d.Cout; };
    std::vector<float> xa((size_t)B * s.bs, 0.f), xb((size_t)B * s.bs, 0.f), outbuf;
    float* out = s.out;
    if (s.w_pre) {   // fused pre 1x1: x0 = W_pre * z[in slice] + b
      ConvArgs a; fill(a, *dpre);
      a.w = s.w_pre; a.bias = s.b_pre; a.x = s.z; a.x_kind = XK_F32_FM; a.x_bs = s.z_bs; a.x_ts = s.z_ts; a.x_c0 = s.pre_c0;
      a.T_in = s.T; a.Nq = s.T; a.T_out = s.T; a.y32 = xa.data(); a.y32_bs = s.bs; a.y32_ts = s.H; a.rg = s.rg;
      conv(*dpre, a, B, EPI_STD, dtype);
    } else {
      std::memcpy(xa.data(), s.x0, (size_t)B * s.bs * 4);
    }
    if (s.w_post) { outbuf.assign((size_t)B * s.bs, 0.f); out = outbuf.data(); }
    else if (!s.accum) std::memset(out, 0, (size_t)B * s.bs * 4);
    for (int l = 0; l < s.layers; ++l) {
      WnArgs a;
      a.x_in = (l % 2 ? xb : xa).data(); a.x_out = (l % 2 ? xa : xb).data(); a.oacc = out;
      a.bs = s.bs; a.T = s.T; a.H = s.H; a.HP = s.HP;
      a.w_in = s.w_in[l]; a.w_rs = s.w_rs[l]; a.b_rs = s.b_rs[l];
      a.bbias = s.bbias + (size_t)l * 2 * s.H; a.bbias_bs = s.bbias_bs;
      a.taps = s.taps; a.KS = s.KS; a.nIt1 = s.nIt1; a.last = s.final_layer && l == s.layers - 1;
      a.rg = s.rg;
      wn(din, a.last ? drs_last : drs, a, B, dtype);
    }
    if (s.x_out) std::memcpy(s.x_out, (s.layers % 2 ? xb : xa).data(), (size_t)B * s.bs * 4);
    if (s.w_post) {  // fused post 1x1: z[out slice] -= W_post * out + b
      ConvArgs a; fill(a, *dpost);
      a.w = s.w_post; a.bias = s.b_post; a.x = out; a.x_kind = XK_F32_FM; a.x_bs = s.bs; a.x_ts = s.H;
      a.T_in = s.T; a.Nq = s.T; a.T_out = s.T; a.rg = s.rg;
      a.res = s.z; a.res_bs = s.z_bs; a.res_ts = s.z_ts; a.res_c0 = s.post_c0; a.res_sign = s.post_sign;
      a.y32 = s.z; a.y32_bs = s.z_bs; a.y32_ts = s.z_ts; a.y32_c0 = s.post_c0;
      conv(*dpost, a, B, EPI_STD, dtype);
    }
    return QVC_OK;
  }
  // fused pair = the two convs back to back with the intermediate rounded to the operand type
  int pair(const ConvDesc& d1, const ConvDesc& d2, const PairArgs& p, int B, int dtype, const Ragged& rg = Ragged()) {
    if (dtype == QVC_BF16X) return pair_mixed(d1, d2, p, B, rg);
    std::vector<uint16_t> xt((size_t)B * p.bs);
    auto fill = [](ConvArgs& a, const ConvDesc& d) {
      a.Cin = d.Cin; a.CinP = d.CinP; a.taps = d.taps; a.dil = d.dil; a.left = d.left; a.KS = d.KS(); a.nIt = d.nIt();
      a.nchunk = d.nchunk; a.M = d.M; a.up_s = d.up_s; a.up_p = d.up_p; a.Cout = d.Cout; };
    ConvArgs a1; fill(a1, d1);
    a1.w = p.w1; a1.bias = p.b1; a1.x = p.x; a1.x_kind = XK_OP_FM; a1.x_bs = p.bs; a1.x_ts = p.C; a1.T_in = p.T; a1.slope_in = p.slope;
    a1.rg = rg; a1.Nq = p.T; a1.T_out = p.T; a1.y16 = xt.data(); a1.y16_bs = p.bs; a1.y16_ts = p.C; a1.slope_out = p.slope;
    conv(d1, a1, B, EPI_STD, dtype);
    ConvArgs a2; fill(a2, d2);
    a2.w = p.w2; a2.bias = p.b2; a2.x = xt.data(); a2.x_kind = XK_OP_FM; a2.x_bs = p.bs; a2.x_ts = p.C; a2.T_in = p.T;
    a2.rg = rg; a2.Nq = p.T; a2.T_out = p.T; a2.res16 = p.x; a2.res_bs = p.bs; a2.res_ts = p.C;
    a2.y16 = p.y; a2.y16_bs = p.bs; a2.y16_ts = p.C; a2.slope_out = 1.f;
    conv(d2, a2, B, EPI_STD, dtype);
    return QVC_OK;
  }
  // QVC_BF16X: bf16 MFMA operands, f16 residual stream (x, y); the intermediate tile stays in the operand type
  int pair_mixed(const ConvDesc& d1, const ConvDesc& d2, const PairArgs& p, int B, const Ragged& rg) {
    const size_t n = (size_t)B * p.bs;
    std::vector<float> xf(n), yf(n, 0.f);
    const uint16_t* xs = static_cast<const uint16_t*>(p.x);
    for (size_t i = 0; i < n; ++i) xf[i] = from_f16(xs[i]);
    std::vector<uint16_t> xt(n);
    auto fill = [](ConvArgs& a, const ConvDesc& d) {
      a.Cin = d.Cin; a.CinP = d.CinP; a.taps = d.taps; a.dil = d.dil; a.left = d.left; a.KS = d.KS(); a.nIt = d.nIt();
      a.nchunk = d.nchunk; a.M = d.M; a.up_s = d.up_s; a.up_p = d.up_p; a.Cout = d.Cout; };
    ConvArgs a1; fill(a1, d1);
    a1.w = p.w1; a1.bias = p.b1; a1.x = xf.data(); a1.x_kind = XK_F32_FM; a1.x_bs = p.bs; a1.x_ts = p.C; a1.T_in = p.T; a1.slope_in = p.slope;
    a1.rg = rg; a1.Nq = p.T; a1.T_out = p.T; a1.y16 = xt.data(); a1.y16_bs = p.bs; a1.y16_ts = p.C; a1.slope_out = p.slope;
    conv(d1, a1, B, EPI_STD, QVC_BF16);
    ConvArgs a2; fill(a2, d2);
    a2.w = p.w2; a2.bias = p.b2; a2.x = xt.data(); a2.x_kind = XK_OP_FM; a2.x_bs = p.bs; a2.x_ts = p.C; a2.T_in = p.T;
    a2.rg = rg; a2.Nq = p.T; a2.T_out = p.T; a2.res = xf.data(); a2.res_bs = p.bs; a2.res_ts = p.C;
    a2.y32 = yf.data(); a2.y32_bs = p.bs; a2.y32_ts = p.C;
    conv(d2, a2, B, EPI_STD, QVC_BF16);
    uint16_t* ys = static_cast<uint16_t*>(p.y);
    for (size_t i = 0; i < n; ++i) ys[i] = to_f16(std::fmin(std::fmax(yf[i], -65504.f), 65504.f));
    return QVC_OK;
  }
  int pair3(const ConvDesc* d1, const ConvDesc* d2, const PairArgs3& a, int B, int dtype) {
    for (int i = 0; i < a.n; ++i) pair(d1[i], d2[i], a.p[i], B, dtype, a.rg);
    return QVC_OK;
  }
  // a whole ResBlock in one launch (qvc_chain_impl.h): replayed pair by pair through the buffers the arguments name
  bool chain_ok(const ConvDesc* d1, const ConvDesc* d2, int n) const { return chain_supported(d1, d2, n); }
  int chain(const ConvDesc* d1, const ConvDesc* d2, const ChainArgs& a, int B, int dtype) {
    for (int q = 0; q < a.n; ++q) pair(d1[q], d2[q], a.p[q], B, dtype, a.rg);
    return QVC_OK;
  }
  int gemv(const GemvArgs& a) {
    for (int b = 0; b < a.batch; ++b)
      for (int r = 0; r < a.rows; ++r) {
        double s = 0.0;
        for (int k = 0; k < a.gin; ++k) s += (double)a.w[(size_t)r * a.gin + k] * a.g[(size_t)b * a.gin + k];
        a.out[(size_t)b * a.rows + r] = (float)s + a.bias[r];
      }
    return QVC_OK;
  }
  int sample(const SampleArgs& a) {
    for (int b = 0; b < a.batch; ++b)
      for (int t = 0; t < a.frames; ++t)
        for (int c = 0; c < a.C; ++c) {
          const size_t bt = (size_t)b * a.frames + t;
          a.z[bt * a.C + c] = a.stats[bt * 2 * a.C + c] + a.noise[((size_t)b * a.C + c) * a.frames + t] * std::exp(a.stats[bt * 2 * a.C + a.C + c]);
        }
    return QVC_OK;
  }
  // conv_post + tail as one backend op (qvc_post_tail_impl.h): the same two steps through a host buffer
  bool post_tail_ok(const ConvDesc&) const { return true; }
  int post_tail(const ConvDesc& d, const PostTailArgs& a, int batch, int dtype) {
    std::vector<float> post((size_t)batch * a.F * 72, 0.f);
    ConvArgs c = a.c;
    c.y32 = post.data(); c.y32_bs = (int64_t)a.F * 72; c.y32_ts = 72;
    const int st = conv(d, c, batch, EPI_STD, dtype);
    if (st != QVC_OK) return st;
    TailArgs ta{post.data(), a.fir, a.out, nullptr, batch, a.F};
    ta.rg = a.rg;
    return tail(ta);
  }
  // Tail: same formulas as istft_synth_kernel, evaluated sample by sample.
  int tail(const TailArgs& a) {
    const int Fpad = a.F, Lpad = 4 * (Fpad - 1), NOpad = 4 * Lpad;
    const double pi = 3.14159265358979323846;
    for (int b = 0; b < a.batch; ++b) {
      const int F = ragged_len(a.rg, b, Fpad), L = F > 0 ? 4 * (F - 1) : 0, NO = 4 * L;   // this utterance's own frames [Flo, F)
      const int Flo = ragged_lo(a.rg, b);
      std::vector<double> xw((size_t)4 * std::max(F, 1) * 16), y((size_t)4 * std::max(L, 1));
      const float* pb = a.post + (size_t)b * Fpad * 72;
      for (int k = 0; k < 4; ++k)
        for (int t = Flo; t < F; ++t) {
          double re[9], im[9];
          for (int q = 0; q < 9; ++q) {
            const double mag = std::exp((double)pb[(size_t)t * 72 + k * 18 + q]);
            const double ph = pi * std::sin((double)pb[(size_t)t * 72 + k * 18 + 9 + q]);
            re[q] = mag * std::cos(ph); im[q] = mag * std::sin(ph);
          }
          for (int m = 0; m < 16; ++m) {
            double acc = re[0] + ((m & 1) ? -re[8] : re[8]);
            for (int q = 1; q < 8; ++q) acc += 2.0 * (re[q] * std::cos(2 * pi * q * m / 16) - im[q] * std::sin(2 * pi * q * m / 16));
            xw[((size_t)k * F + t) * 16 + m] = acc / 16.0 * (0.5 - 0.5 * std::cos(2 * pi * m / 16));
          }
        }
      for (int k = 0; k < 4; ++k)
        for (int n = 0; n < Lpad; ++n) {
          double v = 0;
          if (n >= 4 * Flo && n < L) {
            double num = 0, env = 0;
            for (int t = Flo; t < F; ++t) {
              const int m = n + 8 - 4 * t;
              if (m < 0 || m >= 16) continue;
              const double w = 0.5 - 0.5 * std::cos(2 * pi * m / 16);
              num += xw[((size_t)k * F + t) * 16 + m]; env += w * w;
            }
            v = num / env;
            y[(size_t)k * L + n] = v;
          }
          if (a.y_mb) a.y_mb[((size_t)b * 4 + k) * Lpad + n] = (float)v;
        }
      for (int o = 0; o < NOpad; ++o) {
        double s = 0;
        if (o < NO && o >= 16 * Flo)
          for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 63; ++j) {
              const int u = o + j - 31;
              if (u < 16 * Flo || u >= NO || (u & 3)) continue;
              s += (double)a.fir[k * 63 + j] * y[(size_t)k * L + (u >> 2)];
            }
        a.out[(size_t)b * NOpad + o] = (float)s;
      }
    }
    return QVC_OK;
  }
  int zero(void* p, size_t bytes) { std::memset(p, 0, bytes); return QVC_OK; }
  // csrc/qvc_small.hip copy_batch_kernel: the descriptors of a batch are independent, so their order does not matter
  int copy_batch(const CopyDesc* d, int n) {
    if (n < 0 || n > kCopyBatchMax) return QVC_ERR_BAD_ARG;
    for (int i = 0; i < n; ++i)
      for (size_t r = 0; r < d[i].rows; ++r) {
        char* dst = static_cast<char*>(d[i].dst) + r * d[i].dpitch;
        if (d[i].src) std::memcpy(dst, static_cast<const char*>(d[i].src) + r * d[i].spitch, d[i].width);
        else std::memset(dst, 0, d[i].width);
      }
    return QVC_OK;
  }

  // ---- speaker encoder launches (csrc/qvc_spk.hip): W_hh recovered from its fragment stream with the kernel's
  //      index math; h rounded to the operand type between steps as the LDS copy is; cell state in fp32
  int lstm(const LstmArgs& a, int KS, int dtype) {
    const int H = a.H, HP = KS * 32, NW = KS;
    std::vector<float> W((size_t)4 * H * HP, 0.f);
    const uint16_t* src = static_cast<const uint16_t*>(a.w_hh);
    SpkPlan S; S.H = H; S.HP = HP; S.NW = NW; S.KS = KS;
    for (int w = 0; w < NW; ++w)
      for (int ks = 0; ks < KS; ++ks)
        for (int f = 0; f < 8; ++f)
          for (int lane = 0; lane < 64; ++lane) {
            const int row = spk_hh_row(S, w, f, lane & 15);
            for (int j = 0; j < 8; ++j) {
              const uint16_t h = src[((((size_t)w * KS + ks) * 8 + f) * 64 + lane) * 8 + j];
              if (row >= 0) W[(size_t)row * HP + ks * 32 + (lane >> 4) * 8 + j] = dtype == QVC_F16 ? from_f16(h) : from_bf16(h);
            }
          }
    auto sig = [](float x) { return 1.f / (1.f + std::exp(-x)); };
    for (int p = 0; p < a.P; ++p) {
      const float* xb;
      if (a.shared) {
        const int u = p / a.n_part, i = p % a.n_part;
        xb = a.xp + ((size_t)u * a.F + spk_start(a.F, i)) * 4 * H;
      } else {
        xb = a.xp + (size_t)p * a.S * 4 * H;
      }
      std::vector<float> h((size_t)HP, 0.f), c((size_t)H, 0.f), hn((size_t)H, 0.f), gate((size_t)4 * H);
      for (int t = 0; t < a.S; ++t) {
        for (int r = 0; r < 4 * H; ++r) {
          double acc = xb[(size_t)t * 4 * H + r];
          for (int k = 0; k < H; ++k) acc += (double)W[(size_t)r * HP + k] * h[k];
          gate[r] = (float)acc;
        }
        for (int j = 0; j < H; ++j) {
          c[j] = sig(gate[H + j]) * c[j] + sig(gate[j]) * std::tanh(gate[2 * H + j]);
          hn[j] = sig(gate[3 * H + j]) * std::tanh(c[j]);
          h[j] = round_op(hn[j], dtype);
          if (a.hseq) static_cast<uint16_t*>(a.hseq)[((size_t)p * a.S + t) * HP + j] = dtype == QVC_F16 ? to_f16(h[j]) : to_bf16(h[j]);
        }
      }
      if (a.hfin) std::memcpy(a.hfin + (size_t)p * H, hn.data(), (size_t)H * 4);
    }
    return QVC_OK;
  }
  int spk_embed(const SpkEmbedArgs& a) {
    const int H = a.H;
    for (int u = 0; u < a.utterances; ++u) {
      std::vector<double> mean((size_t)H, 0.0);
      for (int i = 0; i < a.n_part; ++i) {
        const float* h = a.hfin + ((size_t)u * a.n_part + i) * H;
        std::vector<double> e((size_t)H);
        double ss = 0.0;
        for (int r = 0; r < H; ++r) {
          double s = a.lb[r];
          for (int k = 0; k < H; ++k) s += (double)a.lw[(size_t)r * H + k] * h[k];
          e[r] = s > 0.0 ? s : 0.0; ss += e[r] * e[r];
        }
        for (int r = 0; r < H; ++r) mean[r] += e[r] / std::sqrt(ss);
      }
      for (int r = 0; r < H; ++r) a.g[(size_t)u * H + r] = (float)(mean[r] / a.n_part);
    }
    return QVC_OK;
  }
};

struct Run {
  Plan P; Workspace W; EmuBackend be;
  int prepare(const qvc_config* cfg, int B, int T, int64_t ws_bytes) {
    P = build_plan(*cfg);
    if (P.status != QVC_OK) return P.status;
    W = carve_workspace(P, B, T);
    if (ws_bytes < W.bytes) return QVC_ERR_SMALL_BUFFER;
    return QVC_OK;
  }
};
}  // namespace

extern "C" {

// Same signature as qvc_infer_batch, host pointers everywhere, no stream.
int qvc_emu_infer_batch(const qvc_config* cfg, const void* blob, const float* unit, const float* g, const float* noise,
                        float* out, int32_t batch, int32_t frames, void* workspace, int64_t workspace_bytes) {
  Run r;
  int st = r.prepare(cfg, batch, frames, workspace_bytes);
  if (st != QVC_OK) return st;
  Path<EmuBackend> c{r.P, static_cast<const char*>(blob), static_cast<char*>(workspace), r.W, batch, frames, r.be};
  c.cond_table(g);
  c.enc_p(unit, noise, c.wsp<float>(r.W.z));
  c.flow(c.wsp<float>(r.W.z));
  c.dec_trunk_wave(c.wsp<float>(r.W.z), c.wsp<float>(r.W.post), out);
  return c.status;
}

// Same signature as qvc_infer_batch_ragged, host pointers (the lengths too), no stream.
int qvc_emu_infer_batch_ragged(const qvc_config* cfg, const void* blob, const float* unit, const float* g, const float* noise,
                               float* out, int32_t batch, int32_t max_frames, const int32_t* frames_host, void* workspace,
                               int64_t workspace_bytes) {
  Run r;
  int st = r.prepare(cfg, batch, max_frames, workspace_bytes);
  if (st != QVC_OK) return st;
  Path<EmuBackend> c{r.P, static_cast<const char*>(blob), static_cast<char*>(workspace), r.W, batch, max_frames, r.be};
  c.lens = frames_host;
  c.cond_table(g);
  c.enc_p(unit, noise, c.wsp<float>(r.W.z));
  c.flow(c.wsp<float>(r.W.z));
  c.dec_trunk_wave(c.wsp<float>(r.W.z), c.wsp<float>(r.W.post), out);
  return c.status;
}

// Same signature as qvc_stream_step, host pointers, no stream.
int qvc_emu_stream_step(const qvc_config* cfg, const void* blob, void* state, int64_t state_bytes, const float* unit_new,
                        const float* g, const float* noise_new, float* out, int32_t batch, int32_t hop, const int32_t* pos,
                        const int32_t* lens, void* workspace, int64_t workspace_bytes) {
  Plan P = build_plan(*cfg);
  const StreamGeom G = stream_geom(P, hop);
  if (G.status != QVC_OK) return G.status;
  if (state_bytes < carve_stream_state(P, G, batch).bytes || workspace_bytes < carve_stream_scratch(P, G, batch).bytes) return QVC_ERR_SMALL_BUFFER;
  EmuBackend be;
  return stream_step(P, static_cast<const char*>(blob), static_cast<char*>(state), static_cast<char*>(workspace), unit_new, g,
                     noise_new, out, batch, hop, pos, lens, be);
}

// Same signatures as qvc_enc_q / qvc_flow_forward, host pointers, no stream.
int qvc_emu_enc_q(const qvc_config* cfg, const void* encq_blob, const float* spec, const float* g, const float* noise,
                  float* z_fm, int32_t batch, int32_t frames, void* workspace, int64_t workspace_bytes) {
  Run r;
  int st = r.prepare(cfg, batch, frames, workspace_bytes);
  if (st != QVC_OK) return st;
  const EncQPlan Q = build_encq_plan(*cfg);
  if (Q.status != QVC_OK) return Q.status;
  Path<EmuBackend> c{r.P, static_cast<const char*>(encq_blob), static_cast<char*>(workspace), r.W, batch, frames, r.be};
  c.enc_q(Q, static_cast<const char*>(encq_blob), spec, g, noise, z_fm);
  return c.status;
}

int qvc_emu_flow_forward(const qvc_config* cfg, const void* blob, float* z_fm, const float* g, int32_t batch, int32_t frames,
                         void* workspace, int64_t workspace_bytes) {
  Run r;
  int st = r.prepare(cfg, batch, frames, workspace_bytes);
  if (st != QVC_OK) return st;
  Path<EmuBackend> c{r.P, static_cast<const char*>(blob), static_cast<char*>(workspace), r.W, batch, frames, r.be};
  c.cond_table(g);
  c.flow(z_fm, /*forward=*/true);
  return c.status;
}

// Same signature as qvc_speaker_embed, host pointers, no stream.
int qvc_emu_speaker_embed(const qvc_config* cfg, const void* spk_blob, const float* mel, float* g, int32_t utterances,
                          int32_t mel_frames, void* workspace, int64_t workspace_bytes) {
  SpkPlan S = build_spk_plan(*cfg);
  if (S.status != QVC_OK) return S.status;
  const SpkWorkspace W = carve_spk_workspace(S, utterances, mel_frames);
  if (workspace_bytes < W.bytes) return QVC_ERR_SMALL_BUFFER;
  EmuBackend be;
  return spk_path(S, dec_dtype(*cfg), static_cast<const char*>(spk_blob), static_cast<char*>(workspace), W, mel, g,
                  utterances, mel_frames, be);
}

// Stage taps for debugging: copies frame-major fp32 buffers out of the workspace after a run.
// which: 0 = z (after flow), 1 = post (conv_post output), 2 = stage-0 MRF mean, 3 = stage-1 MRF mean
// Layout decisions of the plan for `cfg` (test hook: the shipped configuration must actually take the fast paths):
// out[0] enc_p.proj rows paired [mu | log sigma] (sampling in the epilogue), out[1] its fragments per wave,
// out[2..3] up-samplers 0 / 1 lane-packed, out[4] conv_post + tail can run as one launch,
// out[5..6] waves per workgroup of the stage 0 / 1 ResBlock pairs, out[7] all pairs fusable
int qvc_emu_plan_flags(const qvc_config* cfg, int32_t* out) {
  if (!cfg || !out) return QVC_ERR_BAD_ARG;
  const Plan P = build_plan(*cfg);
  if (P.status != QVC_OK) return P.status;
  for (int i = 0; i < 8; ++i) out[i] = 0;
  out[0] = P.enc_proj.gau; out[1] = P.enc_proj.MF;
  for (size_t i = 0; i < P.stages.size() && i < 2; ++i) {
    out[2 + i] = P.stages[i].up.lp;
    out[5 + i] = block_waves(P.stages[i].c1[0]);
  }
  out[4] = post_tail_supported(P.conv_post) ? 1 : 0;
  int all = 1;
  for (const StagePlan& st : P.stages)
    for (size_t j = 0; j < st.c1.size(); ++j) all = all && pair_supported(st.c1[j], st.c2[j]) && st.c1[j].lp;
  out[7] = all;
  return QVC_OK;
}

int64_t qvc_emu_tap_offset(const qvc_config* cfg, int32_t batch, int32_t frames, int32_t which) {
  Plan P = build_plan(*cfg);
  if (P.status != QVC_OK) return P.status;
  Workspace W = carve_workspace(P, batch, frames);
  switch (which) {
    case 0: return W.z;
    case 1: return W.post;
    case 2: return W.ra[0][0];   // final tensor of ResBlock 0, stage 0 (operand type)
    case 3: return -1;
    case 4: return W.u[0];   // operand type since the fused-pair path
    case 5: return W.stats;
    default: return -1;
  }
}

}  // extern "C"
