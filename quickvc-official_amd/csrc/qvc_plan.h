// qvc_plan.h -- weight-blob layout ("plan") shared by the host packer, the device
// orchestration and the test-only host emulation.  Header-only, plain C++17, no HIP.
//
// The plan is a pure function of qvc_config: every conv of the hot path gets a
// descriptor (shapes, padded shapes, byte offsets into the blob).  Nothing here is
// stored in the blob itself, so packer and runtime can never disagree.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../include/qvc.h"

namespace qvc {

constexpr int kWaves = 4;          // waves per workgroup in the conv kernel (M split)
constexpr int kKStep = 32;         // MFMA 16x16x32: K elements per step
constexpr int kFragElems = 512;    // one A fragment = 64 lanes x 8 elements

// Operand types by part of the path.  QVC_BF16X = bf16 MFMA operands in the ResBlock pairs (80 % of the path's
// FLOPs) with their residual stream kept in f16, f16 everywhere else (WaveNets, conv_pre, up-samplers, conv_post).
inline int wn_dtype(const qvc_config& c) { return c.operand_dtype == QVC_BF16X ? QVC_F16 : c.operand_dtype; }    // enc_p / enc_q / flow
inline int dec_dtype(const qvc_config& c) { return c.operand_dtype == QVC_BF16X ? QVC_F16 : c.operand_dtype; }   // generator convs + streams
inline int pair_dtype(const qvc_config& c) { return c.operand_dtype; }                 // launch code of the pairs (QVC_BF16X itself)
inline int pair_weight_dtype(const qvc_config& c) { return c.operand_dtype == QVC_BF16X ? QVC_BF16 : c.operand_dtype; }

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

enum EpiKind : int32_t { EPI_STD = 0, EPI_GAU = 1, EPI_RESSKIP = 2,
                         EPI_SAMPLE = 3 };   // rows [mu | log sigma] paired like the gate rows: z = mu + noise * exp(log sigma)

// One conv expressed as an implicit GEMM  D[v][q] = sum_{tap,ci} Wv[v][tap][ci] * X[q + tap*dil - left][ci].
struct ConvDesc {
  int32_t M = 0;        // virtual output channels (s*Cout for a polyphase transposed conv)
  int32_t MF = 1;       // A fragments (16 rows) per wave
  int32_t WM = 4;       // waves of a workgroup along M (the other kWaves/WM split the frames)
  int32_t nchunk = 1;   // M chunks of WM*MF*16 rows
  int32_t Cin = 0;      // real input channels
  int32_t CinP = 0;     // padded to a multiple of 32
  int32_t taps = 1, dil = 1, left = 0;
  int32_t up_s = 1, up_p = 0;   // polyphase: stride / padding of the ConvTranspose1d (1/0: plain conv)
  int32_t Cout = 0;     // real output channels (M = up_s * Cout)
  int32_t ksize = 1;    // kernel size of the original (transposed) conv, for FLOP accounting
  int32_t gau = 0;      // 1: rows permuted so that a wave owns tanh and sigmoid rows of the same channels
  int32_t lp = 0;       // 1: "lane-packed" rows (fused ResBlock pairs): a lane's MF accumulator quads are 4*MF CONSECUTIVE
                        //    channels, so the epilogues move 16 bytes per lane instead of 8 (see conv_row)
  int64_t w_off = 0;    // byte offset of the A stream
  int64_t b_off = 0;    // byte offset of the fp32 bias [MP]
  int32_t MP() const { return nchunk * WM * MF * 16; }
  int32_t KS() const { return CinP / kKStep; }
  int32_t nIt() const { return taps * KS(); }
  int64_t w_bytes() const { return (int64_t)nchunk * WM * nIt() * MF * kFragElems * 2; }
  int64_t b_bytes() const { return (int64_t)MP() * 4; }
};

// Row permutation of the A stream: packed position -> original output channel (or -1 = zero padding).
// `wave` = the wave's index along M (0..WM-1).
inline int conv_row(const ConvDesc& d, int chunk, int wave, int mf, int i) {
  if (d.lp) {
    // MFMA 16x16 output: lane (col, lq) holds rows lq*4 .. lq*4+3 of each fragment.  Row i of fragment mf carries
    // channel base + (i/4)*4*MF + mf*4 + i%4, i.e. lane lq owns channels [base + lq*4*MF, base + (lq+1)*4*MF).
    const int v = (chunk * d.WM + wave) * d.MF * 16 + (i >> 2) * 4 * d.MF + mf * 4 + (i & 3);
    return v < d.M ? v : -1;
  }
  if (!d.gau) {
    int v = ((chunk * d.WM + wave) * d.MF + mf) * 16 + i;
    return v < d.M ? v : -1;
  }
  // GAU: a wave owns MF/2 tanh fragments followed by MF/2 sigmoid fragments of the same channels
  const int H = d.M / 2, hf = d.MF / 2;
  const int ch = ((chunk * d.WM + wave) * hf + (mf % hf)) * 16 + i;
  return ch < H ? (mf / hf) * H + ch : -1;
}

// Picks fragments-per-wave and the wave grid: least zero padding first, then the largest block
// (WM*MF rows), then the most waves along M: waves that split the frames instead would each fetch the
// same weight fragments, and weight delivery into the CU is the scarcer resource (measured on the
// stage-2 ResBlock pairs: MF2/WM4/NF10 is 5-7 % faster than MF4/WM2/NF5).
inline void choose_mf(ConvDesc& d) {
  if (d.gau) {
    const int H = d.M / 2;
    int best_hf = 1; long best_key = -1;
    for (int hf = 1; hf <= 4; ++hf) {
      const int nch = ceil_div(H, kWaves * hf * 16);
      const long key = -(long)(nch * kWaves * hf * 16 - H) * 1000 + hf;
      if (best_key == -1 || key > best_key) { best_key = key; best_hf = hf; }
    }
    d.MF = 2 * best_hf; d.WM = kWaves; d.nchunk = ceil_div(H, kWaves * best_hf * 16);
    return;
  }
  static const int cand[][2] = {{4, 4}, {3, 4}, {2, 4}, {1, 4}, {4, 2}, {3, 2}, {4, 1}};   // {MF, WM} built variants
  int best = 0; long best_key = 0; bool first = true;
  for (int c = 0; c < 7; ++c) {
    const int mf = cand[c][0], wm = cand[c][1];
    if (mf == 1 && d.M >= 32) continue;
    const int blockM = wm * mf * 16;
    const int waste = ceil_div(d.M, blockM) * blockM - d.M;
    const long key = -(long)waste * 100000 + blockM * 10 + wm;
    if (first || key > best_key) { best_key = key; best = c; first = false; }
  }
  d.MF = cand[best][0]; d.WM = cand[best][1];
  d.nchunk = ceil_div(d.M, d.WM * d.MF * 16);
}

// Waves of the workgroup that runs a conv: 4, or 8 (all along M) for the wide ResBlock-pair layout.
inline int block_waves(const ConvDesc& d) { return d.WM > kWaves ? d.WM : kWaves; }

// Can the two convs of a ResBlock1 pair run as one fused launch?  (one workgroup must own all channels)
inline bool pair_supported(const ConvDesc& d1, const ConvDesc& d2) {
  if (d1.nchunk != 1 || d2.nchunk != 1 || d1.MF != d2.MF || d1.WM != d2.WM) return false;
  if (d1.M != d2.M || d1.Cin != d1.M || d2.Cin != d2.M || d1.taps != d2.taps || d2.dil != 1 || d1.up_s != 1 || d2.up_s != 1) return false;
  if (d1.MF * 3 * 4 > 160) return false;
  const int64_t lds = (int64_t)((block_waves(d1) / d1.WM) * 3 * 16 + (d1.taps - 1) * d1.dil) * d1.CinP * 2;   // smallest tile
  return lds <= 160 * 1024;
}

// ResBlock convs with >= 256 channels: 8 waves, all along M, 2..4 fragments each.  Two 4-wave workgroups per CU
// would each stream the pair's full weights (2 x 1.44 MB at 256 channels, k 11), and weight delivery into the CU
// (~70 GB/s) is what bounds those launches; one 8-wave workgroup with twice the frames halves that traffic.
inline void wide_pair_layout(ConvDesc& d1, ConvDesc& d2) {
  if (d1.M < 256 || d1.M % 128 || d1.M / 128 > 4) return;
  ConvDesc a = d1, b = d2;
  a.WM = b.WM = 8; a.MF = b.MF = d1.M / 128; a.nchunk = b.nchunk = 1;
  if (pair_supported(a, b)) { d1 = a; d2 = b; }
}
// Layout of the two convs of a ResBlock1 pair: the wide (8-wave) layout where it applies, and lane-packed rows
// whenever the pair runs fused (the unfused fallback keeps natural row order for the generic conv kernel).
inline void pair_layout(ConvDesc& d1, ConvDesc& d2) {
  wide_pair_layout(d1, d2);
  if (pair_supported(d1, d2)) { d1.lp = 1; d2.lp = 1; }
}

// WaveNet kernels (fused layer / whole stack): ONE WAVE PER 16 CHANNELS -- a workgroup of CinP/16 waves (12 at
// h = 192) owns all channels of its frames.  A wave pulls global_load_dwordx4 data at only ~7.5 B/clk (measured:
// 4 waves stream 28-30 B/clk per CU, 8 waves ~60, the L1's 64 B/clk), and these layers are bound by exactly that
// weight stream (885 KB per layer into every CU), so the rows are spread over as many waves as there are row
// fragments instead of 4 waves x 3 fragments.
inline int wn_waves(int CinP) { return CinP / 16; }
inline void wn_layout(ConvDesc& d, int mf) { d.WM = wn_waves(d.CinP); d.MF = mf; d.nchunk = 1; }
// A plain conv packed one fragment per wave for W waves (a coupling layer's pre / post) is byte-identical to the
// generic conv kernel's 4 waves x W/4 chunks: the stream is ordered by (chunk*WM + wave) either way.
inline ConvDesc generic_layout(ConvDesc d) {
  if (!d.gau && d.MF == 1 && d.nchunk == 1 && d.WM > kWaves && d.WM % kWaves == 0) { d.nchunk = d.WM / kWaves; d.WM = kWaves; }
  return d;
}
inline bool wn_layout_ok(const ConvDesc& din) {
  return din.gau && din.MF == 2 && din.nchunk == 1 && din.WM == wn_waves(din.CinP) && din.WM >= 1 && din.WM <= 16 &&
         din.M == 2 * din.Cin;
}

// Whole-stack WaveNet kernel: window = 32 output frames + halo, built for 3 or 6 column fragments.
inline int wn_stack_nf(int taps, int layers) { return ceil_div(32 + (taps - 1) * layers, 16); }
inline bool wn_stack_ok(const ConvDesc& din, int layers) {
  const int nf = wn_stack_nf(din.taps, layers);
  const int halo = (din.taps - 1) / 2 * layers;                  // output frames = window columns [halo, halo+32)
  const bool cols_ok = nf == 3 || (nf == 6 && halo >= 16 && halo + 32 <= 64);   // fragments 1..3 hold the skip sum when nf == 6
  return wn_layout_ok(din) && layers <= 16 && cols_ok;
}

// Can a coupling layer's pre / post 1x1 convs ride inside its whole-stack launch?  (they must share the stack's
// wave <-> channel ownership: pre's rows = the residual-stream rows, post reads the skip sum)
inline bool wn_fuse_ok(const ConvDesc& din, const ConvDesc& pre, const ConvDesc& post, int layers) {
  return wn_stack_ok(din, layers) && pre.WM == din.WM && pre.MF == 1 && pre.nchunk == 1 && pre.CinP <= din.CinP &&
         post.WM == din.WM && post.nchunk == 1 && post.MF == 1 && post.CinP == din.CinP;
}

struct WNPlan {
  int32_t layers = 0;
  std::vector<ConvDesc> in_conv;   // k-tap h -> 2h, GAU row order; bias lives in the cond/bias table
  std::vector<ConvDesc> rs_conv;   // 1x1 h -> 2h (h on the last layer)
  int64_t inbias_off = -1;         // enc_p only: fp32 [layers][2h] in_layer biases (no conditioning)
};

struct FlowStepPlan {
  int32_t layer = 0;       // index into flow.flows (0,2,4,6 -> 0..3)
  int32_t flipped = 0;     // 1: logical channel c lives at physical channel C-1-c
  int32_t in_c0 = 0;       // physical first channel read by pre
  int32_t out_c0 = 0;      // physical first channel updated by post
  ConvDesc pre, post;
  WNPlan wn;
  int32_t cond_row0 = 0;   // first row of this step's block in the cond table ([layers][2h])
};

struct StagePlan {
  ConvDesc up;
  std::vector<ConvDesc> c1, c2;   // [resblock j][pair p] flattened: j*3+p
  int32_t ch = 0;                 // channels of this stage
  int32_t rate = 1;
};

struct Plan {
  qvc_config cfg{};
  int32_t status = QVC_OK;
  ConvDesc enc_pre, enc_proj;
  WNPlan enc_wn;
  std::vector<FlowStepPlan> flow;   // in execution order (reverse=True)
  ConvDesc conv_pre;
  int32_t dec_cond_row0 = 0;        // rows [dec_cond_row0, +init_ch) of the cond table
  std::vector<StagePlan> stages;
  ConvDesc conv_post;
  int32_t post_channels = 0;        // subbands * 2 * (n_fft/2+1)
  int32_t total_up = 1;             // prod(upsample_rates)
  // cond GEMV: fp32 weights [cond_rows][gin] + bias [cond_rows]
  int32_t cond_rows = 0;
  int64_t cond_w_off = 0, cond_b_off = 0;
  // synthesis FIR: fp32 [subbands][fir_taps], zero-stuffing gain folded in
  int64_t fir_off = 0;
  int64_t blob_bytes = 0;
};

inline int validate(const qvc_config& c) {
  auto bad = [](bool cond) { return cond; };
  if (bad(c.unit_channels <= 0 || c.inter_channels <= 0 || c.hidden_channels <= 0 || c.gin_channels <= 0)) return QVC_ERR_BAD_CONFIG;
  if (bad(c.inter_channels % 8 || c.hidden_channels % 8 || c.unit_channels % 8)) return QVC_ERR_BAD_CONFIG;
  if (bad(c.hidden_channels > 256 || c.gin_channels % 4 || c.gin_channels > 512)) return QVC_ERR_BAD_CONFIG;   // one workgroup owns all WN channels (up to 16 waves x 16 channels)
  if (bad(c.wn_kernel_size < 1 || c.wn_kernel_size % 2 == 0 || c.wn_kernel_size > 15)) return QVC_ERR_BAD_CONFIG;
  if (bad(c.enc_layers < 1 || c.enc_layers > 64 || c.flow_layers < 1 || c.flow_layers > 64)) return QVC_ERR_BAD_CONFIG;
  if (bad(c.n_flows < 1 || c.n_flows > 16 || c.n_flows % 2)) return QVC_ERR_BAD_CONFIG;   // flips must cancel
  if (bad(c.n_ups < 1 || c.n_ups > QVC_MAX_UPS || c.n_resblocks != 3)) return QVC_ERR_BAD_CONFIG;   // the MRF mean is taken over three ResBlocks
  int ch = c.upsample_initial_channel;
  if (bad(ch <= 0 || ch % 8)) return QVC_ERR_BAD_CONFIG;
  for (int i = 0; i < c.n_ups; ++i) {
    int s = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
    if (bad(s < 1 || s > 16 || k < s || k > 64)) return QVC_ERR_BAD_CONFIG;
    // models.py:335: padding (k-s+1-i)//2 with output_padding 1-i gives T_out = s*T_in only when k-s+1-i is even;
    // the workspace carve-up, the tail and the output shape all assume s*T_in (and PyTorch rejects output_padding < 0)
    if (bad((k - s + 1 - i) < 0 || (k - s + 1 - i) % 2 != 0 || i > 1)) return QVC_ERR_BAD_CONFIG;
    if (bad(ch % 2)) return QVC_ERR_BAD_CONFIG;
    ch /= 2;
    if (bad(ch % 8)) return QVC_ERR_BAD_CONFIG;
  }
  for (int j = 0; j < c.n_resblocks; ++j) {
    int k = c.resblock_kernel_sizes[j];
    if (bad(k < 1 || k % 2 == 0 || k > 15)) return QVC_ERR_BAD_CONFIG;
    for (int p = 0; p < 3; ++p) if (bad(c.resblock_dilations[j][p] < 1 || c.resblock_dilations[j][p] > 16)) return QVC_ERR_BAD_CONFIG;
  }
  if (bad(c.n_fft != 16 || c.hop != 4)) return QVC_ERR_BAD_CONFIG;           // the tail kernel is built for 16/4
  if (bad(c.decoder != QVC_DEC_MULTISTREAM && c.decoder != QVC_DEC_MULTIBAND)) return QVC_ERR_BAD_CONFIG;
  if (bad(c.subbands != 4 || c.fir_taps != 63)) return QVC_ERR_BAD_CONFIG;
  if (bad(c.operand_dtype != QVC_BF16 && c.operand_dtype != QVC_F16 && c.operand_dtype != QVC_BF16X)) return QVC_ERR_BAD_CONFIG;
  return QVC_OK;
}

inline ConvDesc make_conv(int M, int Cin, int taps, int dil, bool gau = false) {
  ConvDesc d;
  d.M = M; d.Cout = M; d.Cin = Cin; d.CinP = (int)align_up(Cin, kKStep);
  d.taps = taps; d.ksize = taps; d.dil = dil; d.left = (taps - 1) / 2 * dil; d.gau = gau ? 1 : 0;
  choose_mf(d);
  return d;
}

// proj of enc_p / enc_q (h -> 2*inter, 1x1; models.py:73,92): rows [mu | log sigma] in the gate-row pairing, so that
// a lane holds mu and log sigma of the same channels and the sampling z = mu + noise * exp(log sigma) (models.py:93-94)
// happens in the conv's epilogue -- the stats tensor is never written.  Falls back to natural rows (+ sample_kernel)
// for widths the paired-row kernels are not built for.
inline ConvDesc make_proj(int C, int H) {
  ConvDesc d = make_conv(2 * C, H, 1, 1, /*gau=*/true);
  if (d.WM == kWaves && (d.MF == 2 || d.MF == 4 || d.MF == 6)) return d;
  return make_conv(2 * C, H, 1, 1);
}

inline ConvDesc make_upconv(int Cin, int Cout, int k, int s, int p) {
  ConvDesc d;
  d.up_s = s; d.up_p = p; d.Cout = Cout; d.M = s * Cout;
  d.Cin = Cin; d.CinP = (int)align_up(Cin, kKStep);
  d.taps = ceil_div(k, s); d.ksize = k; d.dil = 1; d.left = d.taps - 1;
  choose_mf(d);
  // lane-packed rows (as in the fused pairs): a lane's MF quads are 4*MF consecutive channels of ONE phase, so the
  // polyphase scatter stores 8*MF bytes per lane and frame -- whole 128-byte lines per 4 lanes instead of 32-byte pieces
  if (d.MF % 2 == 0 && Cout % (4 * d.MF) == 0) d.lp = 1;
  return d;
}

inline Plan build_plan(const qvc_config& c) {
  Plan P;
  P.cfg = c;
  P.status = validate(c);
  if (P.status != QVC_OK) return P;
  int64_t off = 0;
  auto place = [&](ConvDesc& d, bool with_bias = true) {
    d.w_off = off; off = align_up(off + d.w_bytes(), 256);
    if (with_bias) { d.b_off = off; off = align_up(off + d.b_bytes(), 256); } else { d.b_off = -1; }
  };
  const int H = c.hidden_channels, C = c.inter_channels, K = c.wn_kernel_size;
  auto make_wn = [&](WNPlan& w, int layers) {
    w.layers = layers;
    for (int i = 0; i < layers; ++i) {
      ConvDesc a = make_conv(2 * H, H, K, 1, /*gau=*/true);
      wn_layout(a, 2);
      place(a, /*with_bias=*/false);
      w.in_conv.push_back(a);
      // 1x1 res/skip: rows paired like the gate rows ([res | skip] of the same channels per wave) so
      // that the fused layer kernel updates x and the skip accumulator for the channels it owns
      ConvDesc r = make_conv(i < layers - 1 ? 2 * H : H, H, 1, 1, /*gau=*/i < layers - 1);
      wn_layout(r, i < layers - 1 ? 2 : 1);
      place(r);
      w.rs_conv.push_back(r);
    }
  };
  // ---- enc_p
  P.enc_pre = make_conv(H, c.unit_channels, 1, 1); place(P.enc_pre);
  make_wn(P.enc_wn, c.enc_layers);
  P.enc_wn.inbias_off = off; off = align_up(off + (int64_t)c.enc_layers * 2 * H * 4, 256);
  P.enc_proj = make_proj(C, H); place(P.enc_proj);
  // ---- flow, execution order of reverse=True: Flip, L(n-1), Flip, L(n-2), ...
  int cond_rows = 0;
  bool flipped = false;
  for (int s = 0; s < c.n_flows; ++s) {
    FlowStepPlan f;
    flipped = !flipped;                       // the Flip that precedes the layer
    f.layer = c.n_flows - 1 - s;
    f.flipped = flipped ? 1 : 0;
    f.in_c0 = flipped ? C / 2 : 0;
    f.out_c0 = flipped ? 0 : C / 2;
    // pre / post share the WaveNet kernels' channel ownership (one wave per 16 channels) when they can run
    // inside the whole-stack launch: pre's rows = the stack's residual-stream rows, post reads the skip sum;
    // otherwise they keep the generic conv layout and run as launches of their own
    f.pre = make_conv(H, C / 2, 1, 1);
    f.post = make_conv(C / 2, H, 1, 1);
    {
      ConvDesc probe = make_conv(2 * H, H, K, 1, /*gau=*/true);
      wn_layout(probe, 2);
      const bool fuse = wn_stack_ok(probe, c.flow_layers) && f.pre.CinP <= probe.CinP && ceil_div(C / 2, 16) <= probe.WM &&
                        probe.WM % kWaves == 0;   // so that generic_layout() can still run them unfused
      if (fuse) { f.pre.WM = probe.WM; f.pre.MF = 1; f.pre.nchunk = 1; f.post.WM = probe.WM; f.post.MF = 1; f.post.nchunk = 1; }
    }
    place(f.pre);
    make_wn(f.wn, c.flow_layers);
    place(f.post);
    f.cond_row0 = cond_rows; cond_rows += c.flow_layers * 2 * H;
    P.flow.push_back(f);
  }
  // ---- decoder
  P.conv_pre = make_conv(c.upsample_initial_channel, C, 7, 1); place(P.conv_pre);
  P.dec_cond_row0 = cond_rows; cond_rows += c.upsample_initial_channel;
  int ch = c.upsample_initial_channel;
  for (int i = 0; i < c.n_ups; ++i) {
    StagePlan st;
    int s = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
    int p = (k - s + 1 - i) / 2;              // models.py:335 (output_padding 1-i only sets T_out)
    st.up = make_upconv(ch, ch / 2, k, s, p); place(st.up);
    ch /= 2; st.ch = ch; st.rate = s; P.total_up *= s;
    for (int j = 0; j < c.n_resblocks; ++j)
      for (int q = 0; q < 3; ++q) {
        ConvDesc a = make_conv(ch, ch, c.resblock_kernel_sizes[j], c.resblock_dilations[j][q]);
        ConvDesc b = make_conv(ch, ch, c.resblock_kernel_sizes[j], 1);
        pair_layout(a, b);
        place(a); place(b);
        st.c1.push_back(a); st.c2.push_back(b);
      }
    P.stages.push_back(st);
  }
  P.post_channels = c.subbands * 2 * (c.n_fft / 2 + 1);
  P.conv_post = make_conv(P.post_channels, ch, 7, 1); place(P.conv_post);
  // ---- cond GEMV table and FIR
  P.cond_rows = cond_rows;
  P.cond_w_off = off; off = align_up(off + (int64_t)cond_rows * c.gin_channels * 4, 256);
  P.cond_b_off = off; off = align_up(off + (int64_t)cond_rows * 4, 256);
  P.fir_off = off; off = align_up(off + (int64_t)c.subbands * c.fir_taps * 4, 256);
  P.blob_bytes = off;
  return P;
}

// ---------------------------------------------------------------- workspace carve-up
struct Workspace {
  int64_t bb = 0;        // fp32 [B][cond_rows]         cond GEMV output (+ folded biases)
  int64_t xw = 0;        // fp32 [B][T][H]              WN residual stream (ping)
  int64_t xw2 = 0;       // fp32 [B][T][H]              WN residual stream (pong)
  int64_t oacc = 0;      // fp32 [B][T][H]              WN skip accumulator
  int64_t acts = 0;      // op   [B][T][H]              gated activations
  int64_t stats = 0;     // fp32 [B][T][2C]             enc_p.proj output
  int64_t z = 0;         // fp32 [B][T][C]              latent (flow state)
  int64_t c0 = 0;        // op   [B][T][init_ch]        lrelu(conv_pre + cond)
  std::vector<int64_t> u, xt;              // per stage: up output, conv1 output (op type, unfused fallback)
  std::vector<std::vector<int64_t>> ra, rb;   // per stage, per ResBlock: stream ping/pong (op type) -- ResBlocks may run concurrently
  int64_t post = 0;      // fp32 [B][F][post_channels]
  int64_t bytes = 0;
};

inline Workspace carve_workspace(const Plan& P, int B, int T) {
  Workspace W;
  const qvc_config& c = P.cfg;
  int64_t off = 0;
  auto take = [&](int64_t n) { int64_t o = off; off = align_up(off + n, 256); return o; };
  const int64_t BT = (int64_t)B * T;
  W.bb = take((int64_t)B * std::max(P.cond_rows, c.enc_layers * 2 * c.hidden_channels) * 4);   // enc_q's table too
  W.xw = take(BT * c.hidden_channels * 4);
  W.xw2 = take(BT * c.hidden_channels * 4);
  W.oacc = take(BT * c.hidden_channels * 4);
  W.acts = take(BT * c.hidden_channels * 2);
  W.stats = take(BT * 2 * c.inter_channels * 4);
  W.z = take(BT * c.inter_channels * 4);
  W.c0 = take(BT * c.upsample_initial_channel * 2);
  int64_t t = T;
  for (size_t i = 0; i < P.stages.size(); ++i) {
    t *= P.stages[i].rate;
    int64_t n = (int64_t)B * t * P.stages[i].ch;
    W.u.push_back(take(n * 2));
    W.ra.emplace_back(); W.rb.emplace_back();
    for (int j = 0; j < c.n_resblocks; ++j) { W.ra.back().push_back(take(n * 2)); W.rb.back().push_back(take(n * 2)); }
    W.xt.push_back(take(n * 2 * c.n_resblocks));
  }
  W.post = take((int64_t)B * (t + 1) * P.post_channels * 4);
  W.bytes = off;
  return W;
}

// ---------------------------------------------------------------- posterior encoder enc_q (models.py:582,617)
// CondNormalWN(spec_channels -> inter, hidden, k5, 16 layers, gin): own blob, same kernels as enc_p; the
// conditioning rows (cond_layer on g + in_layer biases) form its own GEMV table.
struct EncQPlan {
  int32_t status = QVC_OK;
  int32_t spec_channels = 641;
  ConvDesc pre, proj;
  WNPlan wn;
  int32_t cond_rows = 0;
  int64_t cond_w_off = 0, cond_b_off = 0;
  int64_t blob_bytes = 0;
};

inline EncQPlan build_encq_plan(const qvc_config& c) {
  EncQPlan Q;
  Q.status = validate(c);
  if (Q.status != QVC_OK) return Q;
  Q.spec_channels = c.spec_channels > 0 ? c.spec_channels : 641;
  if (Q.spec_channels > 8192) { Q.status = QVC_ERR_BAD_CONFIG; return Q; }
  const int H = c.hidden_channels, C = c.inter_channels, K = c.wn_kernel_size;
  int64_t off = 0;
  auto place = [&](ConvDesc& d, bool with_bias = true) {
    d.w_off = off; off = align_up(off + d.w_bytes(), 256);
    if (with_bias) { d.b_off = off; off = align_up(off + d.b_bytes(), 256); } else { d.b_off = -1; }
  };
  Q.pre = make_conv(H, Q.spec_channels, 1, 1);
  if (Q.pre.WM < kWaves) {   // 641 input channels: a wave grid along the frames (WM < 4) would need a 4x wider LDS tile than fits
    Q.pre.WM = kWaves; Q.pre.MF = std::min(4, ceil_div(ceil_div(H, 16), kWaves)); Q.pre.nchunk = ceil_div(H, kWaves * Q.pre.MF * 16);
  }
  place(Q.pre);
  Q.wn.layers = c.enc_layers;
  for (int i = 0; i < c.enc_layers; ++i) {
    ConvDesc a = make_conv(2 * H, H, K, 1, /*gau=*/true);
    wn_layout(a, 2);
    place(a, /*with_bias=*/false);
    Q.wn.in_conv.push_back(a);
    ConvDesc r = make_conv(i < c.enc_layers - 1 ? 2 * H : H, H, 1, 1, /*gau=*/i < c.enc_layers - 1);
    wn_layout(r, i < c.enc_layers - 1 ? 2 : 1);
    place(r);
    Q.wn.rs_conv.push_back(r);
  }
  Q.proj = make_proj(C, H); place(Q.proj);
  Q.cond_rows = c.enc_layers * 2 * H;
  Q.cond_w_off = off; off = align_up(off + (int64_t)Q.cond_rows * c.gin_channels * 4, 256);
  Q.cond_b_off = off; off = align_up(off + (int64_t)Q.cond_rows * 4, 256);
  Q.blob_bytes = off;
  return Q;
}

// ---------------------------------------------------------------- speaker encoder (models.py:507-546)
constexpr int kSpkLayers = 3;      // nn.LSTM(mel, hidden, 3), models.py:508-510
constexpr int kSpkPartial = 128;   // embed_utterance(partial_frames=128, partial_hop=64), models.py:528
constexpr int kSpkHop = 64;
constexpr int kSpkCols = 16;       // partials per workgroup of the recurrence kernel (one MFMA column fragment)

// Partials of one utterance: range(0, F-128, 64) plus the last 128 frames; the whole mel when F <= 128.
inline int spk_partials(int mel_frames) {
  return mel_frames > kSpkPartial ? ceil_div(mel_frames - kSpkPartial, kSpkHop) + 1 : 1;
}
inline int spk_steps(int mel_frames) { return mel_frames > kSpkPartial ? kSpkPartial : mel_frames; }
inline int spk_start(int mel_frames, int i) {   // first mel frame of partial i
  return i + 1 < spk_partials(mel_frames) ? i * kSpkHop : (mel_frames > kSpkPartial ? mel_frames - kSpkPartial : 0);
}

// The LSTM as GEMMs: per layer one batched input projection  xp = W_ih x + b_ih + b_hh  over all frames (the
// conv kernel, M = 4H rows in PyTorch's gate order i,f,g,o) and one persistent launch that walks the steps with
// W_hh streamed as MFMA A fragments.  Wave w of that launch owns hidden units [32w, 32w+32) of all four gates:
// fragment f = 2*gate + half holds rows gate*H + 32w + 16*half + (0..15), so the gate math is lane-local.
struct SpkPlan {
  int32_t status = QVC_OK;
  int32_t n_mel = 80, H = 0, HP = 0, NW = 0, KS = 0;
  ConvDesc ih[kSpkLayers];
  int64_t hh_off[kSpkLayers] = {};
  int64_t hh_bytes = 0;
  int64_t lin_w_off = 0, lin_b_off = 0;   // fp32 [H][H], [H]  (model_embedding_size = hidden, models.py:585)
  int64_t blob_bytes = 0;
};

inline int spk_hh_row(const SpkPlan& S, int wave, int frag, int i) {   // packed position -> W_hh row or -1
  const int unit = wave * 32 + (frag & 1) * 16 + i;
  return unit < S.H ? (frag >> 1) * S.H + unit : -1;
}

inline SpkPlan build_spk_plan(const qvc_config& c) {
  SpkPlan S;
  S.n_mel = c.n_mel_channels > 0 ? c.n_mel_channels : 80;
  S.H = c.gin_channels;
  if (S.H <= 0 || S.H % 8 || S.H > 256 || S.n_mel > 1024 ||
      (c.operand_dtype != QVC_BF16 && c.operand_dtype != QVC_F16 && c.operand_dtype != QVC_BF16X)) { S.status = QVC_ERR_BAD_CONFIG; return S; }
  S.HP = (int)align_up(S.H, 32); S.NW = S.HP / 32; S.KS = S.HP / kKStep;
  int64_t off = 0;
  for (int l = 0; l < kSpkLayers; ++l) {
    S.ih[l] = make_conv(4 * S.H, l == 0 ? S.n_mel : S.H, 1, 1);
    S.ih[l].w_off = off; off = align_up(off + S.ih[l].w_bytes(), 256);
    S.ih[l].b_off = off; off = align_up(off + S.ih[l].b_bytes(), 256);
  }
  S.hh_bytes = (int64_t)S.NW * S.KS * 8 * kFragElems * 2;
  for (int l = 0; l < kSpkLayers; ++l) { S.hh_off[l] = off; off = align_up(off + S.hh_bytes, 256); }
  S.lin_w_off = off; off = align_up(off + (int64_t)S.H * S.H * 4, 256);
  S.lin_b_off = off; off = align_up(off + (int64_t)S.H * 4, 256);
  S.blob_bytes = off;
  return S;
}

struct SpkWorkspace {
  int64_t xp0 = 0;    // fp32 [U][F][4H]        layer-0 input projection, shared by overlapping partials
  int64_t xp = 0;     // fp32 [P][S][4H]        layer-1/2 input projections
  int64_t hseq = 0;   // op   [P16][S][HP]      hidden sequence of the previous layer (P, H padded: unconditional stores)
  int64_t hfin = 0;   // fp32 [P][H]            final hidden state of the last layer
  int64_t bytes = 0;
};

inline SpkWorkspace carve_spk_workspace(const SpkPlan& S, int U, int F) {
  SpkWorkspace W;
  int64_t off = 0;
  auto take = [&](int64_t n) { int64_t o = off; off = align_up(off + n, 256); return o; };
  const int64_t P = (int64_t)U * spk_partials(F), St = spk_steps(F);
  W.xp0 = take((int64_t)U * F * 4 * S.H * 4);
  W.xp = take(P * St * 4 * S.H * 4);
  W.hseq = take(align_up(P, kSpkCols) * St * S.HP * 2);
  W.hfin = take(P * S.H * 4);
  W.bytes = off;
  return W;
}

}  // namespace qvc
