// qvc_path.h -- the launch sequence of the hot path, written once and parameterised on a backend.
//
// Backend = what "launch" means: the product's HipBackend (qvc_api.hip) enqueues gfx950 kernels
// on a stream; the test-only host emulation under oracle/ replays the very same sequence on the
// CPU from the same packed blob, which lets the orchestration (buffers, strides, flip folding,
// fused epilogues) be checked against the oracle without a GPU.  The product never links the
// emulation.
//
// A backend provides:  int conv(const ConvDesc&, const ConvArgs&, int batch, int epi, int dtype);
//                      int pair(const ConvDesc&, const ConvDesc&, const PairArgs&, int batch, int dtype);
//                      int pair3(const ConvDesc* d1, const ConvDesc* d2, const PairArgs3&, int batch, int dtype);
//                      int wn(const ConvDesc& in, const ConvDesc& rs, const WnArgs&, int batch, int dtype);
//                      int gemv(const GemvArgs&); int sample(const SampleArgs&);
//                      int tail(const TailArgs&); int zero(void* ptr, size_t bytes);
//                      void fork(int n); void branch(int j); void branch_done(int j);
//                      void wait_branch_done(int j); void join(int n);   (stream fork/join; no-ops on one stream)
#pragma once
#include <algorithm>
#include <cstdlib>
#include "qvc_kernels.h"

namespace qvc {

template <class Backend>
struct Path {
  const Plan& P;
  const char* blob;
  char* ws;
  Workspace W;
  int B, T;
  Backend& be;
  int status = QVC_OK;
  // ragged batches / streaming windows (see Ragged): per-utterance sequence lengths, and -- for a window that is not
  // the whole sequence -- the absolute unit frame `pos[b]` that sits at buffer row `off`
  const int32_t* lens = nullptr;
  const int32_t* pos = nullptr;
  int32_t off = 0;
  Ragged rg(int mul, int add = 0) const { Ragged r; r.lens = lens; r.pos = pos; r.mul = mul; r.add = add; r.off = off; return r; }
  // unit frames as they are on disk (dataset/encode.py:38: (frames, 256) per utterance), i.e. [B][T][unit_channels],
  // instead of the reference's in-memory (B, 256, T) (data_utils_new_new.py:121-122 transposes after loading)
  bool unit_fm = false;
  // streaming: stage-0 ResBlock outputs handed over from a previous window instead of this call's own (dec_back)
  const void* s0_override[3] = {nullptr, nullptr, nullptr};

  template <typename U> U* wsp(int64_t off) const { return reinterpret_cast<U*>(ws + off); }
  int dtype_wn() const { return wn_dtype(P.cfg); }     // enc_p / enc_q / flow
  int dtype_dec() const { return dec_dtype(P.cfg); }   // generator convs
  int dtype_pair() const { return pair_dtype(P.cfg); } // fused ResBlock pairs (may differ: QVC_BF16X)

  ConvArgs args(const ConvDesc& d, const char* wb = nullptr) const {
    ConvArgs a;
    if (!wb) wb = blob;
    a.w = wb + d.w_off;
    a.bias = d.b_off >= 0 ? reinterpret_cast<const float*>(wb + d.b_off) : nullptr;
    a.Cin = d.Cin; a.CinP = d.CinP; a.taps = d.taps; a.dil = d.dil; a.left = d.left;
    a.KS = d.KS(); a.nIt = d.nIt(); a.nchunk = d.nchunk; a.M = d.M;
    a.up_s = d.up_s; a.up_p = d.up_p; a.Cout = d.Cout;
    return a;
  }
  void conv(const ConvDesc& d, const ConvArgs& a, int dt, int epi = EPI_STD) {
    if (status != QVC_OK) return;
    const ConvDesc g = generic_layout(d);
    ConvArgs ga = a; ga.nchunk = g.nchunk;
    status = be.conv(g, ga, B, epi, dt);
  }
  void zero(int64_t off, int64_t bytes) {
    if (status != QVC_OK) return;
    status = be.zero(ws + off, (size_t)bytes);
  }

  // ---- conditioning table: bb[b][row] for every cond row (flow WN layers + dec.cond)
  void cond_table(const float* g) {
    if (status != QVC_OK) return;
    GemvArgs ga{reinterpret_cast<const float*>(blob + P.cond_w_off), reinterpret_cast<const float*>(blob + P.cond_b_off),
                g, wsp<float>(W.bb), P.cond_rows, P.cfg.gin_channels, B};
    status = be.gemv(ga);
  }

  // ---- WN stack over xw (in place) accumulating into oacc (modules.py:69-114)
  // bb: conditioning rows for layer 0 (+ l*2H per layer), bb_bs: batch stride (0 = shared)
  void wn(const WNPlan& wn, const float* bb, int64_t bb_bs, const char* wb = nullptr) {
    if (!wb) wb = blob;
    const int H = P.cfg.hidden_channels;
    const int64_t bs = (int64_t)T * H;
    // Whole-stack kernel, in launches of `chunk` layers: fewer layers per launch = less halo to recompute
    // (4 layers: 48-frame window for 32 output frames; 16 layers: 96), more launches = more x round trips.
    const int chunk = be.wn_stack_chunk(wn.layers);
    if (chunk > 0 && wn.layers % chunk == 0 && wn_stack_ok(wn.in_conv[0], chunk) && be.use_wn_stack(B, T)) {
      for (int l0 = 0; l0 < wn.layers; l0 += chunk) {
        const int part = l0 / chunk;
        WnStackArgs a;
        a.x0 = wsp<float>(part % 2 == 0 ? W.xw : W.xw2);
        a.out = wsp<float>(W.oacc); a.bs = bs; a.T = T; a.H = H; a.HP = wn.in_conv[0].CinP;
        for (int l = 0; l < chunk; ++l) {
          a.w_in[l] = wb + wn.in_conv[l0 + l].w_off; a.w_rs[l] = wb + wn.rs_conv[l0 + l].w_off;
          a.b_rs[l] = reinterpret_cast<const float*>(wb + wn.rs_conv[l0 + l].b_off);
        }
        a.bbias = bb + (int64_t)l0 * 2 * H; a.bbias_bs = bb_bs;
        a.layers = chunk; a.taps = wn.in_conv[0].taps; a.KS = wn.in_conv[0].KS(); a.nIt1 = wn.in_conv[0].nIt();
        a.final_layer = l0 + chunk == wn.layers ? 1 : 0;
        a.accum = l0 > 0 ? 1 : 0;
        a.rg = rg(1);
        a.x_out = a.final_layer ? nullptr : wsp<float>(part % 2 == 0 ? W.xw2 : W.xw);
        if (status == QVC_OK) status = be.wn_stack(wn.in_conv[0], wn.rs_conv[0], wn.rs_conv[wn.layers - 1], a, B, dtype_wn(), nullptr, nullptr);
      }
      return;
    }
    zero(W.oacc, (int64_t)B * bs * 4);
    for (int l = 0; l < wn.layers; ++l) {
      const ConvDesc& din = wn.in_conv[l];
      const ConvDesc& drs = wn.rs_conv[l];
      WnArgs a;
      a.x_in = wsp<float>(l % 2 == 0 ? W.xw : W.xw2);
      a.x_out = wsp<float>(l % 2 == 0 ? W.xw2 : W.xw);
      a.oacc = wsp<float>(W.oacc);
      a.bs = bs; a.T = T; a.H = H; a.HP = din.CinP;
      a.w_in = wb + din.w_off; a.w_rs = wb + drs.w_off;
      a.b_rs = reinterpret_cast<const float*>(wb + drs.b_off);
      a.bbias = bb + (int64_t)l * 2 * H; a.bbias_bs = bb_bs;
      a.taps = din.taps; a.KS = din.KS(); a.nIt1 = din.nIt(); a.last = l == wn.layers - 1 ? 1 : 0;
      a.rg = rg(1);
      if (status == QVC_OK) status = be.wn(din, drs, a, B, dtype_wn());
    }
  }

  // ---- enc_p (models.py:75-95)
  void enc_p(const float* unit, const float* noise, float* z_out) {
    const qvc_config& c = P.cfg;
    const int H = c.hidden_channels, C = c.inter_channels;
    {
      ConvArgs a = args(P.enc_pre);
      a.x = unit; a.x_bs = (int64_t)c.unit_channels * T; a.T_in = T;
      if (unit_fm) { a.x_kind = XK_F32_FM; a.x_ts = c.unit_channels; } else { a.x_kind = XK_F32_CM; a.x_ts = T; }
      a.Nq = T; a.T_out = T; a.rg = rg(1);
      a.y32 = wsp<float>(W.xw); a.y32_bs = (int64_t)T * H; a.y32_ts = H;
      conv(P.enc_pre, a, dtype_wn());
    }
    wn(P.enc_wn, reinterpret_cast<const float*>(blob + P.enc_wn.inbias_off), 0);
    {
      ConvArgs a = args(P.enc_proj);
      a.x = wsp<float>(W.oacc); a.x_kind = XK_F32_FM; a.x_bs = (int64_t)T * H; a.x_ts = H; a.T_in = T;
      a.Nq = T; a.T_out = T; a.rg = rg(1);
      if (proj_and_sample(P.enc_proj, a, noise, z_out)) return;
      a.y32 = wsp<float>(W.stats); a.y32_bs = (int64_t)T * 2 * C; a.y32_ts = 2 * C;
      conv(P.enc_proj, a, dtype_wn());
    }
    if (status != QVC_OK) return;
    SampleArgs sa{wsp<float>(W.stats), noise, z_out, B, T, C};
    status = be.sample(sa);
  }
  // proj packed with paired [mu | log sigma] rows (make_proj): z = mu + noise * exp(log sigma) in the conv epilogue
  bool proj_and_sample(const ConvDesc& d, ConvArgs& a, const float* noise, float* z_out) {
    if (!d.gau) return false;
    const int C = P.cfg.inter_channels;
    a.gau_H = C; a.noise = noise; a.noise_bs = (int64_t)C * T; a.noise_ts = T;
    a.y32 = z_out; a.y32_bs = (int64_t)T * C; a.y32_ts = C;
    conv(d, a, dtype_wn(), EPI_SAMPLE);
    return true;
  }

  // ---- enc_q (models.py:75-95 with cond = g, :582,617): spec, g, noise -> z
  void enc_q(const EncQPlan& Q, const char* qblob, const float* spec, const float* g, const float* noise, float* z_out) {
    const qvc_config& c = P.cfg;
    const int H = c.hidden_channels, C = c.inter_channels;
    if (status != QVC_OK) return;
    GemvArgs ga{reinterpret_cast<const float*>(qblob + Q.cond_w_off), reinterpret_cast<const float*>(qblob + Q.cond_b_off),
                g, wsp<float>(W.bb), Q.cond_rows, c.gin_channels, B};
    status = be.gemv(ga);
    {
      ConvArgs a = args(Q.pre, qblob);
      a.x = spec; a.x_kind = XK_F32_CM; a.x_bs = (int64_t)Q.spec_channels * T; a.x_ts = T; a.T_in = T;
      a.Nq = T; a.T_out = T; a.rg = rg(1);
      a.y32 = wsp<float>(W.xw); a.y32_bs = (int64_t)T * H; a.y32_ts = H;
      conv(Q.pre, a, dtype_wn());
    }
    wn(Q.wn, wsp<float>(W.bb), Q.cond_rows, qblob);
    {
      ConvArgs a = args(Q.proj, qblob);
      a.x = wsp<float>(W.oacc); a.x_kind = XK_F32_FM; a.x_bs = (int64_t)T * H; a.x_ts = H; a.T_in = T;
      a.Nq = T; a.T_out = T; a.rg = rg(1);
      if (proj_and_sample(Q.proj, a, noise, z_out)) return;
      a.y32 = wsp<float>(W.stats); a.y32_bs = (int64_t)T * 2 * C; a.y32_ts = 2 * C;
      conv(Q.proj, a, dtype_wn());
    }
    if (status != QVC_OK) return;
    SampleArgs sa{wsp<float>(W.stats), noise, z_out, B, T, C};
    status = be.sample(sa);
  }

  // ---- flow (models.py:39-51, modules.py:199-224); z updated in place.  reverse (the conversion path): the plan's
  // steps in order, x1 <- x1 - m.  forward (the posterior direction, models.py:618): the same steps in the opposite
  // order, x1 <- m + x1 -- every coupling layer sees the same channel flip in both directions (layer l is preceded by
  // l flips going forward and by n-l going back, n even), so the flip-folded pre / post weights serve both.
  void flow(float* z, bool forward = false) {
    const float sign = forward ? 1.f : -1.f;
    const size_t n = P.flow.size();
    for (size_t s = 0; s < n; ++s) flow_step(P.flow[forward ? n - 1 - s : s], z, sign);
  }
  void flow_step(const FlowStepPlan& f, float* z, float sign) {
    const qvc_config& c = P.cfg;
    const int H = c.hidden_channels, C = c.inter_channels;
    const float* bb = wsp<float>(W.bb);
    {
      const ConvDesc& din = f.wn.in_conv[0];
      if (f.wn.layers == be.wn_stack_chunk(f.wn.layers) && wn_fuse_ok(din, f.pre, f.post, f.wn.layers) && be.use_wn_stack(B, T)) {
        // the whole coupling layer in ONE launch: pre 1x1 -> 4 WaveNet layers -> post 1x1 -> x1 -= m
        WnStackArgs a;
        a.bs = (int64_t)T * H; a.T = T; a.H = H; a.HP = din.CinP;
        for (int l = 0; l < f.wn.layers; ++l) {
          a.w_in[l] = blob + f.wn.in_conv[l].w_off; a.w_rs[l] = blob + f.wn.rs_conv[l].w_off;
          a.b_rs[l] = reinterpret_cast<const float*>(blob + f.wn.rs_conv[l].b_off);
        }
        a.bbias = bb + f.cond_row0; a.bbias_bs = P.cond_rows;
        a.layers = f.wn.layers; a.taps = din.taps; a.KS = din.KS(); a.nIt1 = din.nIt();
        a.w_pre = blob + f.pre.w_off; a.b_pre = reinterpret_cast<const float*>(blob + f.pre.b_off);
        a.pre_cin = f.pre.Cin; a.pre_c0 = f.in_c0; a.pre_KS = f.pre.KS();
        a.w_post = blob + f.post.w_off; a.b_post = reinterpret_cast<const float*>(blob + f.post.b_off);
        a.post_m = f.post.M; a.post_c0 = f.out_c0; a.post_mf = f.post.MF;
        a.z = z; a.z_bs = (int64_t)T * C; a.z_ts = C; a.post_sign = sign;
        a.rg = rg(1);
        if (status == QVC_OK)
          status = be.wn_stack(din, f.wn.rs_conv[0], f.wn.rs_conv[f.wn.layers - 1], a, B, dtype_wn(), &f.pre, &f.post);
        return;
      }
      {
        ConvArgs a = args(f.pre);
        a.x = z; a.x_kind = XK_F32_FM; a.x_bs = (int64_t)T * C; a.x_ts = C; a.x_c0 = f.in_c0; a.T_in = T;
        a.Nq = T; a.T_out = T; a.rg = rg(1);
        a.y32 = wsp<float>(W.xw); a.y32_bs = (int64_t)T * H; a.y32_ts = H;
        conv(f.pre, a, dtype_wn());
      }
      wn(f.wn, bb + f.cond_row0, P.cond_rows);
      {   // x1 <- x1 -/+ post(h)   (modules.py:214-217)
        ConvArgs a = args(f.post);
        a.x = wsp<float>(W.oacc); a.x_kind = XK_F32_FM; a.x_bs = (int64_t)T * H; a.x_ts = H; a.T_in = T;
        a.Nq = T; a.T_out = T; a.rg = rg(1);
        a.res = z; a.res_bs = (int64_t)T * C; a.res_ts = C; a.res_c0 = f.out_c0; a.res_sign = sign;
        a.y32 = z; a.y32_bs = (int64_t)T * C; a.y32_ts = C; a.y32_c0 = f.out_c0;
        conv(f.post, a, dtype_wn());
      }
    }
  }

  // input = MRF mean of stage i's ResBlock outputs (each ResBlock's final tensor sits in its `ra` buffer)
  void mean_input(ConvArgs& a, size_t i) {
    const int NB = P.cfg.n_resblocks;
    a.x_kind = XK_OP_FM;
    auto src = [&](int j) -> const void* { return (i == 0 && s0_override[j]) ? s0_override[j] : wsp<void>(W.ra[i][(size_t)j]); };
    a.x = src(0);
    a.x2 = src(NB > 1 ? 1 : 0);
    a.x3 = src(NB > 2 ? 2 : 0);
  }

  // ---- generator trunk (models.py:372-390).  dec_front = conv_pre + stage 0 (its three ResBlock outputs end up in
  // W.ra[0][j]); dec_back = the remaining stages + conv_post, reading stage 0's outputs (or s0_override: streaming).
  void dec_trunk(const float* z, float* post_out) {
    dec_part(z, post_out, 0, (int)P.stages.size(), true, true);
  }
  void dec_front(const float* z) { dec_part(z, nullptr, 0, 1, true, false); }
  void dec_back(float* post_out) { dec_part(nullptr, post_out, 1, (int)P.stages.size(), false, true); }
  // ... all the way to the waveform: conv_post and the tail as ONE launch where the backend has it (the post-conv
  // frames then never reach `post_buf`), else conv_post -> post_buf -> tail
  void dec_trunk_wave(const float* z, float* post_buf, float* wave) { dec_part(z, post_buf, 0, (int)P.stages.size(), true, true, wave); }
  void dec_back_wave(float* post_buf, float* wave) { dec_part(nullptr, post_buf, 1, (int)P.stages.size(), false, true, wave); }
  void dec_part(const float* z, float* post_out, int stage_lo, int stage_hi, bool with_pre, bool with_post, float* wave = nullptr) {
    const qvc_config& c = P.cfg;
    const int C = c.inter_channels, C0 = c.upsample_initial_channel;
    if (with_pre) {   // conv_pre(k7) + cond(g), then the first stage's leaky ReLU fused into the store
      ConvArgs a = args(P.conv_pre);
      a.x = z; a.x_kind = XK_F32_FM; a.x_bs = (int64_t)T * C; a.x_ts = C; a.T_in = T;
      a.Nq = T; a.T_out = T; a.rg = rg(1);
      a.bbias = wsp<float>(W.bb) + P.dec_cond_row0; a.bbias_bs = P.cond_rows;
      a.y16 = wsp<void>(W.c0); a.y16_bs = (int64_t)T * C0; a.y16_ts = C0; a.slope_out = 0.1f;
      conv(P.conv_pre, a, dtype_dec());
    }
    int t_in = T, ch_in = C0;
    int rate = 1;                                    // frames per unit frame at the current stage's INPUT
    for (size_t i = 0; i < (size_t)stage_hi; ++i) {
      if ((int)i < stage_lo) {                       // geometry of the stages this call skips
        const int s_ = P.stages[i].up.up_s, p_ = P.stages[i].up.up_p, k_ = c.upsample_kernel_sizes[i];
        t_in = (t_in - 1) * s_ - 2 * p_ + k_ + (1 - (int)i); ch_in = P.stages[i].ch; rate *= s_;
        continue;
      }
      const StagePlan& st = P.stages[i];
      const int s = st.up.up_s, p = st.up.up_p, k = c.upsample_kernel_sizes[i];
      const int t_out = (t_in - 1) * s - 2 * p + k + (1 - (int)i);         // models.py:335
      const int ch = st.ch;
      const int64_t bs = (int64_t)t_out * ch;
      {   // lrelu(0.1) -> ConvTranspose1d as `s` polyphase filters
        ConvArgs a = args(st.up);
        if (i == 0) { a.x = wsp<void>(W.c0); a.x_kind = XK_OP_FM; }
        else { mean_input(a, i - 1); a.slope_in = 0.1f; }
        a.x_bs = (int64_t)t_in * ch_in; a.x_ts = ch_in; a.T_in = t_in;
        a.Nq = (t_out - 1 + p) / s + 1; a.T_out = t_out; a.rg = rg(rate);
        a.y16 = wsp<void>(W.u[i]); a.y16_bs = bs; a.y16_ts = ch; a.slope_out = 1.f;   // raw, operand type
        conv(st.up, a, dtype_dec());
      }
      rate *= s;                                     // ... and at its output, where the ResBlocks work
      // The ResBlocks of a stage are independent until their mean.  Pair q of all three chains goes out as ONE
      // launch (workgroups of chains with different kernel sizes interleave on the CUs, see rbpair_kernel); a stage
      // whose pairs cannot run fused falls back to one launch per conv, optionally on parallel branches (streams).
      const int NB = c.n_resblocks;
      std::vector<const void*> src((size_t)NB, wsp<void>(W.u[i]));
      auto pair_args = [&](int j, int q, void* dst) {
        const ConvDesc& d1 = st.c1[(size_t)j * 3 + q];
        const ConvDesc& d2 = st.c2[(size_t)j * 3 + q];
        PairArgs pa;
        pa.x = src[(size_t)j]; pa.bs = bs; pa.T = t_out; pa.C = ch; pa.CP = d1.CinP;
        pa.w1 = blob + d1.w_off; pa.b1 = reinterpret_cast<const float*>(blob + d1.b_off);
        pa.w2 = blob + d2.w_off; pa.b2 = reinterpret_cast<const float*>(blob + d2.b_off);
        pa.k = d1.taps; pa.dil = d1.dil; pa.KS = d1.KS(); pa.nIt = d1.nIt(); pa.slope = 0.1f;
        pa.y = dst;
        return pa;
      };
      bool fused = NB <= 3;
      for (int j = 0; j < NB && fused; ++j)
        for (int q = 0; q < 3; ++q) {
          const ConvDesc& d1 = st.c1[(size_t)j * 3 + q];
          const ConvDesc& d2 = st.c2[(size_t)j * 3 + q];
          fused = fused && pair_supported(d1, d2) && d1.lp && d2.lp && d1.MF == st.c1[0].MF && d1.WM == st.c1[0].WM;
        }
      // One launch for the three chains, chain-major grid (blockIdx.z = chain, longest kernel first): the dispatcher
      // hands out all workgroups of the k 11 chain first and the shorter chains backfill the CUs as they free up --
      // longest-processing-time-first, so the launch ends on short workgroups.  Interleaving the chains on the CUs
      // (x % n) left long workgroups for the end: with one 8-wave workgroup per CU (stage 1) 261 us against 226 us for
      // three launches and 199 us chain-major; with two 4-wave workgroups per CU (stage 2) chain-major takes another
      // 45 us off the step (2.11 -> 2.065 ms, same box, alternating runs).  Tiny batches (40-80 workgroups per
      // chain) interleave: everything runs at once, in the time of the longest chain (batch 1: 3 x 43 us instead of
      // 3 x (19 + 31 + 43) us).
      const bool few_tiles = (int64_t)B * t_out * NB <= 256 * 32;
      const bool wide = block_waves(st.c1[0]) != kWaves;
      const int per_launch = (!wide || few_tiles || debug_get(DBG_PAIR_WIDE_LAUNCH)) ? NB : 1;   // 0: one chain per launch
      const int chain_major = ((wide || debug_get(DBG_PAIR_CM4)) && !few_tiles) ? 1 : 0;           // 0: 4-wave layouts interleave the chains (x % n)
      // A ResBlock with a short kernel (k 3: 12 frames of receptive field per side) runs as ONE launch, its three
      // pairs chained on chip (qvc_chain_impl.h): its stream is read once and written once instead of three times
      // each.  The other chains keep the pair-by-pair launches.  (Tiny batches: everything in the time of the longest
      // chain is better, see below.)
      bool chained[3] = {false, false, false};
      if (fused && !few_tiles) {
        for (int j = 0; j < NB; ++j) {
          ConvDesc d1s[3], d2s[3];
          for (int q = 0; q < 3; ++q) { d1s[q] = st.c1[(size_t)j * 3 + q]; d2s[q] = st.c2[(size_t)j * 3 + q]; }
          if (!be.chain_ok(d1s, d2s, 3)) continue;
          ChainArgs ca; ca.n = 3; ca.rg = rg(rate);
          const void* cur = wsp<void>(W.u[i]);
          for (int q = 0; q < 3; ++q) {
            void* dst = q == 1 ? wsp<void>(W.rb[i][(size_t)j]) : wsp<void>(W.ra[i][(size_t)j]);
            src[(size_t)j] = cur;
            ca.p[q] = pair_args(j, q, dst);
            cur = dst;
          }
          src[(size_t)j] = cur;
          if (status == QVC_OK) status = be.chain(d1s, d2s, ca, B, dtype_pair());
          chained[j] = true;
        }
      }
      if (fused) {
        int rest[3], n_rest = 0;
        for (int j = 0; j < NB; ++j) if (!chained[j]) rest[n_rest++] = j;
        for (int q = 0; q < 3 && n_rest > 0; ++q) for (int j0 = 0; j0 < n_rest; j0 += (per_launch < n_rest ? per_launch : n_rest)) {
          const int per_launch_q = per_launch < n_rest ? per_launch : n_rest;
          // stream of ResBlock j: u -> ra -> rb -> ra; the MRF mean of the three final tensors is taken by the
          // consumer (next up-sampler / conv_post) while it stages its input, so nothing is accumulated here
          PairArgs3 a3; a3.n = per_launch_q; a3.rg = rg(rate); a3.chain_major = per_launch_q > 1 ? chain_major : 0;
          ConvDesc d1s[3], d2s[3];
          // the chain with the largest kernel first: its workgroups are the longest, so they should start earliest
          int order[3] = {0, 0, 0};
          for (int s_ = 0; s_ < per_launch_q; ++s_) order[s_] = rest[j0 + s_];
          std::sort(order, order + per_launch_q, [&](int x, int y) { return st.c1[(size_t)x * 3 + q].nIt() > st.c1[(size_t)y * 3 + q].nIt(); });
          for (int s_ = 0; s_ < per_launch_q; ++s_) {
            const int j = order[s_];
            void* dst = q == 1 ? wsp<void>(W.rb[i][(size_t)j]) : wsp<void>(W.ra[i][(size_t)j]);
            a3.p[s_] = pair_args(j, q, dst);
            d1s[s_] = st.c1[(size_t)j * 3 + q]; d2s[s_] = st.c2[(size_t)j * 3 + q];
            src[(size_t)j] = dst;
          }
          if (status == QVC_OK) status = be.pair3(d1s, d2s, a3, B, dtype_pair());
        }
      } else {
      be.fork(NB);                                   // branches wait for everything enqueued so far
      for (int q = 0; q < 3; ++q)
        for (int j = 0; j < NB; ++j) {
          be.branch(j);
          const ConvDesc& d1 = st.c1[(size_t)j * 3 + q];
          const ConvDesc& d2 = st.c2[(size_t)j * 3 + q];
          void* dst = q == 1 ? wsp<void>(W.rb[i][(size_t)j]) : wsp<void>(W.ra[i][(size_t)j]);
          const bool last = q == 2;
          if (pair_supported(d1, d2) && d1.lp && d2.lp) {
            PairArgs3 a1; a1.n = 1; a1.rg = rg(rate);
            a1.p[0] = pair_args(j, q, dst);
            if (status == QVC_OK) status = be.pair3(&d1, &d2, a1, B, dtype_pair());
          } else {
            void* xt = wsp<char>(W.xt[i]) + (size_t)j * (size_t)B * (size_t)bs * 2;
            {   // lrelu -> dilated conv -> lrelu (stored already activated)
              ConvArgs a = args(d1);
              a.x = src[(size_t)j]; a.x_kind = XK_OP_FM; a.x_bs = bs; a.x_ts = ch; a.T_in = t_out; a.slope_in = 0.1f;
              a.Nq = t_out; a.T_out = t_out; a.rg = rg(rate);
              a.y16 = xt; a.y16_bs = bs; a.y16_ts = ch; a.slope_out = 0.1f;
              conv(d1, a, dtype_dec());
            }
            {   // conv -> + x
              ConvArgs a = args(d2);
              a.x = xt; a.x_kind = XK_OP_FM; a.x_bs = bs; a.x_ts = ch; a.T_in = t_out;
              a.Nq = t_out; a.T_out = t_out; a.rg = rg(rate);
              a.res16 = src[(size_t)j]; a.res_bs = bs; a.res_ts = ch;
              a.y16 = dst; a.y16_bs = bs; a.y16_ts = ch; a.slope_out = 1.f;
              conv(d2, a, dtype_dec());
            }
          }
          if (last) be.branch_done(j);
          src[(size_t)j] = dst;
        }
      be.join(NB);                                   // main stream continues when every branch is done
      }
      t_in = t_out; ch_in = ch;
    }
    if (with_post) {   // lrelu(0.01) -> ReflectionPad1d((1,0)) -> subband_conv_post(k7)
      ConvArgs a = args(P.conv_post);
      mean_input(a, P.stages.size() - 1);
      a.x_bs = (int64_t)t_in * ch_in; a.x_ts = ch_in;
      a.T_in = t_in; a.slope_in = 0.01f; a.reflect = 1;
      a.Nq = t_in + 1; a.T_out = t_in + 1; a.rg = rg(rate);
      if (wave && be.post_tail_ok(P.conv_post)) {
        if (status != QVC_OK) return;
        PostTailArgs pt;
        pt.c = a; pt.fir = reinterpret_cast<const float*>(blob + P.fir_off); pt.out = wave; pt.F = t_in + 1;
        pt.rg = rg(P.total_up, 1);
        status = be.post_tail(P.conv_post, pt, B, dtype_dec());
        return;
      }
      a.y32 = post_out; a.y32_bs = (int64_t)(t_in + 1) * P.post_channels; a.y32_ts = P.post_channels;
      conv(P.conv_post, a, dtype_dec());
      if (wave) tail(post_out, wave, nullptr, t_in + 1);
    }
  }

  void tail(const float* post, float* out, float* y_mb, int F) {
    if (status != QVC_OK) return;
    TailArgs ta{post, reinterpret_cast<const float*>(blob + P.fir_off), out, y_mb, B, F};
    ta.rg = rg(P.total_up, 1);
    status = be.tail(ta);
  }
};

// ---- speaker encoder (models.py:507-546): per layer one conv launch (input projection of all frames) and one
// persistent recurrence launch, then the embedding head.  Backend adds:  int lstm(const LstmArgs&, int KS, int dtype);
//                                                                        int spk_embed(const SpkEmbedArgs&);
template <class Backend>
int spk_path(const SpkPlan& S, int dtype, const char* blob, char* ws, const SpkWorkspace& W, const float* mel, float* g,
             int U, int F, Backend& be) {
  const int H = S.H, n_part = spk_partials(F), St = spk_steps(F), P = U * n_part;
  for (int l = 0; l < kSpkLayers; ++l) {
    const ConvDesc& d = S.ih[l];
    ConvArgs ca;
    ca.w = blob + d.w_off; ca.bias = reinterpret_cast<const float*>(blob + d.b_off);
    ca.Cin = d.Cin; ca.CinP = d.CinP; ca.taps = 1; ca.dil = 1; ca.left = 0;
    ca.KS = d.KS(); ca.nIt = d.nIt(); ca.nchunk = d.nchunk; ca.M = d.M; ca.up_s = 1; ca.up_p = 0; ca.Cout = d.Cout;
    int batch;
    if (l == 0) {     // mel (U, n_mel, F) as handed to infer() (convert.py:77); the transpose of models.py:635 is the staging
      ca.x = mel; ca.x_kind = XK_F32_CM; ca.x_bs = (int64_t)S.n_mel * F; ca.x_ts = F; ca.T_in = F;
      ca.Nq = F; ca.T_out = F;
      ca.y32 = reinterpret_cast<float*>(ws + W.xp0); ca.y32_bs = (int64_t)F * 4 * H; ca.y32_ts = 4 * H;
      batch = U;
    } else {
      ca.x = ws + W.hseq; ca.x_kind = XK_OP_FM; ca.x_bs = (int64_t)St * S.HP; ca.x_ts = S.HP; ca.T_in = St;
      ca.Nq = St; ca.T_out = St;
      ca.y32 = reinterpret_cast<float*>(ws + W.xp); ca.y32_bs = (int64_t)St * 4 * H; ca.y32_ts = 4 * H;
      batch = P;
    }
    int rc = be.conv(d, ca, batch, EPI_STD, dtype);
    if (rc != QVC_OK) return rc;
    LstmArgs la;
    la.xp = reinterpret_cast<const float*>(ws + (l == 0 ? W.xp0 : W.xp));
    la.shared = l == 0 ? 1 : 0;
    la.F = F; la.n_part = n_part; la.S = St; la.P = P; la.H = H;
    la.w_hh = blob + S.hh_off[l];
    la.hseq = l + 1 < kSpkLayers ? ws + W.hseq : nullptr;
    la.hfin = l + 1 < kSpkLayers ? nullptr : reinterpret_cast<float*>(ws + W.hfin);
    rc = be.lstm(la, S.KS, dtype);
    if (rc != QVC_OK) return rc;
  }
  SpkEmbedArgs ea{reinterpret_cast<const float*>(ws + W.hfin), reinterpret_cast<const float*>(blob + S.lin_w_off),
                  reinterpret_cast<const float*>(blob + S.lin_b_off), g, U, n_part, H};
  return be.spk_embed(ea);
}

}  // namespace qvc
