// qvc_stream.h -- incremental (streaming) conversion: one hop of new unit frames per call, state carried in a
// caller-owned buffer (BASELINE.json configs[4]).  Written once and parameterised on the backend like qvc_path.h.
//
// The reference converts a whole utterance in one shot (no chunking, no streaming).  Every op on
// the path is a convolution with a bounded, symmetric receptive field (modules.py:64, models.py:327-357) and g is
// global, so the path can run as a pipeline of SEGMENTS, each fed from a ring buffer that keeps the last 2*H frames
// of its input (H = the segment's one-sided receptive field):
//
//     E   enc_p                          ring: unit frames              H = enc_layers * (k-1)/2         (32)
//     F_k coupling layer k of the flow   ring: z before that layer      H = flow_layers * (k-1)/2        (8 each)
//     D1  conv_pre + ups[0] + MRF 0      ring: z after the flow         H = 3 + taps + ceil(mrf0 / r0)   (20)
//     D2  ups[1] + MRF 1 + conv_post + iSTFT / band synthesis           H in unit frames                 (6)
//         ring: the three ResBlock outputs of stage 0
//
// A step appends `hop` new frames to E's ring, runs every segment over its window of 2*H + hop frames, hands the
// CENTRAL hop frames of each result to the next ring (the H frames at either edge lack context and are dropped) and
// slides the rings.  A frame therefore costs (2*H + hop) / hop of its offline cost per segment -- 1.08x overall at
// hop 320 -- instead of (2*88 + hop) / hop for windows over the whole path, and the output lags the input by
// lag() = sum of the H's (90 frames at the shipped config): the look-ahead a non-causal network needs anyway.
// Inside a segment the fused kernels keep their layers' state on chip, so there are no per-layer caches to carry.
// Sequence starts and ends inside a window are exact: every kernel masks its input rows by the absolute position
// of the window (Ragged: pos / lens), i.e. each conv zero-pads at the sequence's own ends.
#pragma once
#include "qvc_path.h"

namespace qvc {

struct StreamGeom {
  int32_t status = QVC_OK;
  int32_t hop = 0, He = 0, Hf = 0, Hd1 = 0, Hd2 = 0, nf = 0, r0 = 1;
  int lag() const { return He + nf * Hf + Hd1 + Hd2; }
  int max_window() const { return hop + 2 * std::max(std::max(He, Hf), std::max(Hd1, Hd2)); }
};

inline StreamGeom stream_geom(const Plan& P, int hop) {
  StreamGeom G;
  const qvc_config& c = P.cfg;
  if (P.status != QVC_OK) { G.status = P.status; return G; }
  if (hop < 1 || c.n_ups != 2 || c.n_flows > 16) { G.status = hop < 1 ? QVC_ERR_BAD_ARG : QVC_ERR_BAD_CONFIG; return G; }
  const int K2 = (c.wn_kernel_size - 1) / 2;
  G.hop = hop; G.nf = c.n_flows; G.r0 = c.upsample_rates[0];
  G.He = c.enc_layers * K2;
  G.Hf = c.flow_layers * K2;
  auto mrf = [&]() {   // one-sided reach of a ResBlock stack in its own frames: sum over pairs of (k-1)/2 * (d + 1)
    int worst = 0;
    for (int j = 0; j < c.n_resblocks; ++j) {
      int s = 0;
      for (int q = 0; q < 3; ++q) s += (c.resblock_kernel_sizes[j] - 1) / 2 * (c.resblock_dilations[j][q] + 1);
      worst = std::max(worst, s);
    }
    return worst;
  }();
  const int taps0 = P.stages[0].up.taps, taps1 = P.stages[1].up.taps;
  const int r1 = c.upsample_rates[1];
  G.Hd1 = 3 + taps0 + ceil_div(mrf, G.r0) + 1;                               // conv_pre (k 7), ups[0], MRF 0, margin
  // ups[1] (in stage-0 frames) + [MRF 1 + conv_post (k 7 + reflect) + iSTFT overlap + 63-tap FIR] (stage-1 frames)
  G.Hd2 = ceil_div(taps1 + ceil_div(mrf + 4 + 2 + 2, r1) + 1, G.r0) + 1;
  return G;
}

struct StreamState {     // byte offsets into the caller-owned state buffer
  int64_t unit = 0;      // fp32 (B, unit_channels, 2*He + hop), the reference's (B, C, T) layout
  int64_t zf[16] = {};   // fp32 [B][2*Hf + hop][C]: z in front of coupling layer k
  int64_t zd = 0;        // fp32 [B][2*Hd1 + hop][C]: z after the flow
  int64_t s0[3] = {};    // operand type [B][(2*Hd2 + hop) * r0][ch0]: stage-0 ResBlock outputs
  int64_t bytes = 0;
};

inline StreamState carve_stream_state(const Plan& P, const StreamGeom& G, int B) {
  StreamState S;
  const qvc_config& c = P.cfg;
  int64_t off = 0;
  auto take = [&](int64_t n) { int64_t o = off; off = align_up(off + n, 256); return o; };
  S.unit = take((int64_t)B * c.unit_channels * (2 * G.He + G.hop) * 4);
  for (int k = 0; k < G.nf; ++k) S.zf[k] = take((int64_t)B * (2 * G.Hf + G.hop) * c.inter_channels * 4);
  S.zd = take((int64_t)B * (2 * G.Hd1 + G.hop) * c.inter_channels * 4);
  for (int j = 0; j < 3; ++j) S.s0[j] = take((int64_t)B * (2 * G.Hd2 + G.hop) * G.r0 * P.stages[0].ch * 2);
  S.bytes = off;
  return S;
}

struct StreamScratch {   // carved from the END of the step's workspace, after the path's own workspace
  int64_t path_bytes = 0;   // workspace of the widest segment
  int64_t noise = 0;        // fp32 (B, C, 2*He + hop): the step's noise placed at the frames enc_p really outputs
  int64_t zwork[2] = {};    // fp32 [B][2*Hf + hop][C]: the coupling layer updates z in place on a copy of its ring
                            // (two of them: layer k's result feeds the copy for layer k+1 in the same launch)
  int64_t wave = 0;         // fp32 (B, spf * (2*Hd2 + hop))
  int64_t tmp[21] = {};     // slide buffers, one per ring (unit, zf[0..nf), zd, s0[0..3)): that ring's kept part
  int64_t bytes = 0;
};

inline int samples_per_frame(const Plan& P) { return P.total_up * P.cfg.hop * P.cfg.subbands; }

inline StreamScratch carve_stream_scratch(const Plan& P, const StreamGeom& G, int B) {
  StreamScratch X;
  const qvc_config& c = P.cfg;
  X.path_bytes = carve_workspace(P, B, G.max_window()).bytes;
  int64_t off = X.path_bytes;
  auto take = [&](int64_t n) { int64_t o = off; off = align_up(off + n, 256); return o; };
  X.noise = take((int64_t)B * c.inter_channels * (2 * G.He + G.hop) * 4);
  for (int k = 0; k < 2; ++k) X.zwork[k] = take((int64_t)B * (2 * G.Hf + G.hop) * c.inter_channels * 4);
  X.wave = take((int64_t)B * samples_per_frame(P) * (2 * G.Hd2 + G.hop) * 4);
  int r = 0;
  X.tmp[r++] = take((int64_t)B * c.unit_channels * 2 * G.He * 4);
  for (int k = 0; k < G.nf; ++k) X.tmp[r++] = take((int64_t)B * 2 * G.Hf * c.inter_channels * 4);
  X.tmp[r++] = take((int64_t)B * 2 * G.Hd1 * c.inter_channels * 4);
  for (int j = 0; j < 3; ++j) X.tmp[r++] = take((int64_t)B * 2 * G.Hd2 * G.r0 * P.stages[0].ch * 2);
  X.bytes = off;
  return X;
}

// One step.  Backend adds:  int copy_batch(const CopyDesc* d, int n)  -- n independent strided copies / zero fills in
// one launch (qvc_kernels.h).  All data movement of a step goes through it: 10 launches where one memcpy per
// hand-off / slide would be ~35.
//   unit_new  (B, unit_channels, hop) fp32: unit frames [pos, pos + hop) of every stream
//   noise_new (B, inter, hop) fp32:         the N(0,1) draw of models.py:94 for frames [pos - He, pos - He + hop)
//                                           (the frames enc_p finishes in this step)
//   out       (B, hop * samples_per_frame): waveform of frames [pos - lag, pos - lag + hop); zeros where that is
//                                           outside [0, lens[b])
//   pos, lens (B,) int32 in the backend's memory: first new frame of this step / sequence length (large if unknown)
template <class Backend>
int stream_step(const Plan& P, const char* blob, char* state, char* ws, const float* unit_new, const float* g,
                const float* noise_new, float* out, int B, int hop, const int32_t* pos, const int32_t* lens, Backend& be) {
  const StreamGeom G = stream_geom(P, hop);
  if (G.status != QVC_OK) return G.status;
  const StreamState S = carve_stream_state(P, G, B);
  const StreamScratch X = carve_stream_scratch(P, G, B);
  const qvc_config& c = P.cfg;
  const int C = c.inter_channels, UC = c.unit_channels;
  const int h = hop, He = G.He, Hf = G.Hf, Hd1 = G.Hd1, Hd2 = G.Hd2;
  int st = QVC_OK;
  auto ok = [&](int rc) { if (st == QVC_OK && rc != QVC_OK) st = rc; };
  CopyDesc cd[kCopyBatchMax];
  int ncd = 0;
  auto flush = [&]() { if (ncd) ok(be.copy_batch(cd, ncd)); ncd = 0; };
  // queue a strided copy (src == nullptr: zero fill); the descriptors between two flushes run as ONE launch and must be
  // independent of each other
  auto add = [&](void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows) {
    if (ncd == kCopyBatchMax) flush();
    if ((dpitch | spitch | width | rows) >> 32) { ok(QVC_ERR_BAD_ARG); return; }
    cd[ncd++] = CopyDesc{dst, src, (uint32_t)dpitch, (uint32_t)spitch, (uint32_t)width, (uint32_t)rows};
  };
  // the rings slide at the END of the step, all of them in two launches (out to the scratch, back in: the kept part and
  // its destination overlap).  A ring is read by exactly one segment per step, so nothing needs it slid earlier.
  struct Slide { char* ring; size_t rows, keep, hopb; };
  Slide slides[21];
  int nslide = 0;
  auto slide = [&](char* ring, size_t rows, size_t keep, size_t hopb) { slides[nslide++] = Slide{ring, rows, keep, hopb}; };
  auto make = [&](int T, int lag_before) {
    Path<Backend> p{P, blob, ws, carve_workspace(P, B, T), B, T, be};
    p.lens = lens; p.pos = pos; p.off = lag_before + (T - h);      // buffer row `off` is absolute frame pos[b]
    return p;
  };
  const size_t zrow = (size_t)C * 4;                               // one frame of z
  const int Tf = 2 * Hf + h;
  // hand the central hop frames of a segment's result (`src`, frame pitch zrow, `srcT` frames per stream) to coupling
  // layer k: into its ring's tail AND -- together with the ring's kept part -- into the work copy the layer updates
  auto feed_flow = [&](int k, const char* src, int srcT) {
    char* ring = state + S.zf[k];
    char* zw = ws + X.zwork[k & 1];
    add(ring + (size_t)2 * Hf * zrow, (size_t)Tf * zrow, src, (size_t)srcT * zrow, (size_t)h * zrow, (size_t)B);
    add(zw, (size_t)Tf * zrow, ring, (size_t)Tf * zrow, (size_t)2 * Hf * zrow, (size_t)B);
    add(zw + (size_t)2 * Hf * zrow, (size_t)Tf * zrow, src, (size_t)srcT * zrow, (size_t)h * zrow, (size_t)B);
    flush();
  };

  {   // the conditioning table (cond rows x g) once per step: every segment's workspace keeps it at the same place
    Path<Backend> p = make(Tf, 0);
    p.cond_table(g);
    ok(p.status);
  }
  // ---- E: enc_p over the unit ring
  {
    const int T = 2 * He + h;
    char* ring = state + S.unit;
    char* nz = ws + X.noise;                                       // zeros either side of the step's noise
    add(ring + (size_t)2 * He * 4, (size_t)T * 4, unit_new, (size_t)h * 4, (size_t)h * 4, (size_t)B * UC);
    add(nz, (size_t)T * 4, nullptr, 0, (size_t)He * 4, (size_t)B * C);
    add(nz + (size_t)He * 4, (size_t)T * 4, noise_new, (size_t)h * 4, (size_t)h * 4, (size_t)B * C);
    add(nz + (size_t)(He + h) * 4, (size_t)T * 4, nullptr, 0, (size_t)He * 4, (size_t)B * C);
    flush();
    Path<Backend> p = make(T, 0);
    p.enc_p(reinterpret_cast<const float*>(ring), reinterpret_cast<const float*>(ws + X.noise), p.template wsp<float>(p.W.z));
    ok(p.status);
    if (G.nf > 0) feed_flow(0, ws + p.W.z + (size_t)He * zrow, T);
    else {
      add(state + S.zd + (size_t)2 * Hd1 * zrow, (size_t)(2 * Hd1 + h) * zrow, ws + p.W.z + (size_t)He * zrow, (size_t)T * zrow, (size_t)h * zrow, (size_t)B);
      flush();
    }
    slide(ring, (size_t)B * UC, (size_t)2 * He * 4, (size_t)h * 4);
  }
  // ---- F_k: one coupling layer per segment, in place on a copy of its ring (its edge rows come out wrong and
  //      must not be written back: the ring has to keep the untouched values for the next step's context)
  for (int k = 0; k < G.nf; ++k) {
    char* zw = ws + X.zwork[k & 1];
    Path<Backend> p = make(Tf, He + k * Hf);
    p.flow_step(P.flow[(size_t)k], reinterpret_cast<float*>(zw), -1.f);
    ok(p.status);
    if (k + 1 < G.nf) feed_flow(k + 1, zw + (size_t)Hf * zrow, Tf);
    else {
      add(state + S.zd + (size_t)2 * Hd1 * zrow, (size_t)(2 * Hd1 + h) * zrow, zw + (size_t)Hf * zrow, (size_t)Tf * zrow, (size_t)h * zrow, (size_t)B);
      flush();
    }
    slide(state + S.zf[k], (size_t)B, (size_t)2 * Hf * zrow, (size_t)h * zrow);
  }
  // ---- D1: conv_pre + stage 0; its three ResBlock outputs feed the stage-0 rings
  const int ch0 = P.stages[0].ch, r0 = G.r0;
  {
    const int T = 2 * Hd1 + h;
    char* ring = state + S.zd;
    Path<Backend> p = make(T, He + G.nf * Hf);
    p.dec_front(reinterpret_cast<const float*>(ring));
    ok(p.status);
    const size_t row = (size_t)ch0 * 2;
    for (int j = 0; j < 3; ++j) {
      const int jj = j < c.n_resblocks ? j : 0;
      add(state + S.s0[j] + (size_t)2 * Hd2 * r0 * row, (size_t)(2 * Hd2 + h) * r0 * row,
          ws + p.W.ra[0][(size_t)jj] + (size_t)Hd1 * r0 * row, (size_t)T * r0 * row, (size_t)h * r0 * row, (size_t)B);
    }
    flush();
    slide(ring, (size_t)B, (size_t)2 * Hd1 * zrow, (size_t)h * zrow);
  }
  // ---- D2: remaining stages + conv_post + iSTFT / band synthesis; the central hop frames are the step's output
  {
    const int T = 2 * Hd2 + h;
    const int spf = samples_per_frame(P);
    Path<Backend> p = make(T, He + G.nf * Hf + Hd1);
    for (int j = 0; j < 3; ++j) p.s0_override[j] = state + S.s0[j];
    p.dec_back_wave(p.template wsp<float>(p.W.post), reinterpret_cast<float*>(ws + X.wave));
    ok(p.status);
    add(out, (size_t)h * spf * 4, ws + X.wave + (size_t)Hd2 * spf * 4, (size_t)T * spf * 4, (size_t)h * spf * 4, (size_t)B);
    const size_t row = (size_t)ch0 * 2;
    for (int j = 0; j < 3; ++j) slide(state + S.s0[j], (size_t)B, (size_t)2 * Hd2 * r0 * row, (size_t)h * r0 * row);
  }
  // ---- slide every ring by one hop: kept parts out (in the launch that also delivers the output), then back in
  for (int r = 0; r < nslide; ++r) add(ws + X.tmp[r], slides[r].keep, slides[r].ring + slides[r].hopb, slides[r].keep + slides[r].hopb, slides[r].keep, slides[r].rows);
  flush();
  for (int r = 0; r < nslide; ++r) add(slides[r].ring, slides[r].keep + slides[r].hopb, ws + X.tmp[r], slides[r].keep, slides[r].keep, slides[r].rows);
  flush();
  return st;
}

}  // namespace qvc
