// qvc_kernels.h -- launch-side view of the gfx950 kernels (arguments + launcher prototypes).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include "qvc_plan.h"

namespace qvc {

enum XKind : int32_t {
  XK_OP_FM = 0,    // operand type (f16/bf16), frame-major [B][T][C]
  XK_F32_FM = 1,   // fp32, frame-major
  XK_F32_CM = 2    // fp32, channel-major (B, C, T) -- the reference's external layout
};

// Ragged batches and streaming windows: where does utterance b start and end inside the buffer a launch works on?
// Buffer row r (at `mul` rows per unit frame) of utterance b is absolute frame  (pos[b] - off) * mul + r  of its
// sequence, which is lens[b] * mul + add frames long.  Every conv zero-pads at the two ends of ITS sequence -- not of
// the padded buffer, not of a streaming window -- so the staging code masks rows outside [lo, hi):
//     lo = max(0, (off - pos[b]) * mul)          hi = min(T, (lens[b] - pos[b] + off) * mul + add)
// Rows outside are never read unmasked, which is why nothing has to be zeroed between launches.
//   lens == nullptr: the sequence ends with the buffer (hi = T);  pos == nullptr: it starts with it (pos = off = 0).
// The arrays live wherever the launch's other pointers live (device memory; host memory in the CPU emulation).
struct Ragged {
  const int32_t* lens = nullptr;
  const int32_t* pos = nullptr;
  int32_t mul = 1, add = 0, off = 0;
};
#if defined(__HIPCC__)
#define QVC_HD __host__ __device__ __forceinline__
#else
#define QVC_HD inline
#endif
// (by value and always inlined: a reference to a member of the by-value kernel arguments would make the compiler
//  copy the whole argument struct to scratch memory -- measured: 296 B of scratch and half the occupancy)
QVC_HD int ragged_len(const Ragged r, int b, int T) {          // hi
  if (!r.lens) return T;
  int d = r.lens[b] - (r.pos ? r.pos[b] - r.off : 0);          // unit frames of the sequence left from buffer row 0
  d = d < 0 ? 0 : (d < T ? d : T);                             // clamped BEFORE scaling: "unknown length" is 2^30
  const int n = d * r.mul + r.add;
  return n < 0 ? 0 : (n < T ? n : T);
}
QVC_HD int ragged_lo(const Ragged r, int b) {                  // lo
  if (!r.pos) return 0;
  const int d = r.off - r.pos[b];
  return d <= 0 ? 0 : d * r.mul;
}

// Arguments of one implicit-GEMM conv launch.  All strides in elements.
struct ConvArgs {
  // ---- input
  const void* x = nullptr;
  int32_t x_kind = XK_F32_FM;
  int64_t x_bs = 0;        // batch stride
  int32_t x_ts = 0;        // frame stride (frame-major) / channel stride (channel-major)
  int32_t x_c0 = 0;        // first channel of the slice (frame-major)
  int32_t Cin = 0, CinP = 0, T_in = 0;
  float slope_in = 1.f;    // leaky-ReLU slope applied while staging (1 = identity)
  // XK_OP_FM only: with x2/x3 set the staged value is act(mean(x, x2, x3)) -- the MRF average of the three
  // ResBlock outputs (models.py:378-384) is taken on the fly by its consumer, in fp32
  const void* x2 = nullptr; const void* x3 = nullptr;
  int32_t reflect = 0;     // 1: ReflectionPad1d((1,0)) in front of a 'same' conv (models.py:345,388)
  // ---- weights (A stream) and biases
  const void* w = nullptr;
  const float* bias = nullptr;     // [MP] or null
  const float* bbias = nullptr;    // per-utterance bias (conditioning), natural channel order
  int64_t bbias_bs = 0;
  int32_t taps = 1, dil = 1, left = 0, KS = 1, nIt = 1, nchunk = 1, M = 0;
  // ---- output mapping: column q, virtual row v -> frame o = q*up_s + v/Cout - up_p, channel v%Cout
  int32_t Nq = 0, up_s = 1, up_p = 0, Cout = 0, T_out = 0;
  // ---- epilogue
  const float* res = nullptr; int64_t res_bs = 0; int32_t res_ts = 0, res_c0 = 0; float res_sign = 1.f;
  const void* res16 = nullptr;     // residual in the operand type (same strides as res; res_c0 ignored)
  float* y32 = nullptr; int64_t y32_bs = 0; int32_t y32_ts = 0, y32_c0 = 0; float y_scale = 1.f; int32_t y_accum = 0;
  void* y16 = nullptr; int64_t y16_bs = 0; int32_t y16_ts = 0; float slope_out = 1.f;
  // res/skip split (WN 1x1, modules.py:104-112): rows >= split go to y32b[.. v-split] += val
  float* y32b = nullptr; int32_t split = 0;
  int32_t gau_H = 0;       // EPI_GAU: hidden size (rows are [tanh | sigmoid]); EPI_SAMPLE: inter channels (rows [mu | log sigma])
  // EPI_SAMPLE: y32[b][q][c] = (mu + bias) + noise[b][c][q] * exp(log sigma + bias), noise in the reference's (B, C, T)
  const float* noise = nullptr; int64_t noise_bs = 0; int32_t noise_ts = 0;
  int32_t ksize = 0;       // polyphase (up_s > 1): kernel size of the transposed conv -- phases with fewer real taps skip their zero ones
  int32_t lp = 0;          // ConvDesc::lp (lane-packed rows): EPI_STD with only a y16 output (the polyphase up-samplers)
  Ragged rg;               // per-utterance INPUT length (in T_in units); see Ragged
};

// Arguments of one fused ResBlock1 pair (modules.py:148-153):  y = x + conv2(lrelu(conv1(lrelu(x)))).
// x / y are frame-major tensors in the operand type (the MRF mean of the three ResBlocks, models.py:378-384,
// is taken by the consumer of the y tensors while it stages its input).
struct PairArgs {
  const void* x = nullptr; int64_t bs = 0; int32_t T = 0, C = 0, CP = 0;
  const void* w1 = nullptr; const float* b1 = nullptr;
  const void* w2 = nullptr; const float* b2 = nullptr;
  int32_t k = 1, dil = 1, KS = 1, nIt = 1;
  float slope = 0.1f;
  void* y = nullptr;
};

// Up to three INDEPENDENT pairs in one launch: pair q of the three ResBlocks of a stage (kernel sizes 3 / 7 / 11).
// Workgroup x serves chain x % n, tile x / n, so the workgroups that share a CU have different durations and drift
// out of phase: one's staging / epilogue (memory) runs under another's GEMMs (MFMA).  All chains share T, C, CP.
struct PairArgs3 {
  PairArgs p[3];
  int32_t n = 1;
  Ragged rg;               // per-utterance length (in T units), shared by the chains
  // 1: chain = blockIdx.z (grid tiles x batch x n): all workgroups of the first chain are dispatched before the second
  //    chain's -- the one-workgroup-per-CU layouts, where the launch then behaves like n launches back to back without
  //    the gaps between them; 0: chain = blockIdx.x % n (chains interleaved on the CUs)
  int32_t chain_major = 0;
#ifdef QVC_STAMP
  unsigned long long* stamps = nullptr;   // developer build only (tools/conv_bench): [workgroup][64] phase stamps of wave 0
#endif
};

// A whole ResBlock1 -- n chained pairs (dilations d_0 .. d_{n-1}) -- in ONE launch (qvc_chain_impl.h), for chains whose
// receptive field is short: p[0].x is read once (tile + halo), p[n-1].y written once, the stream stays on chip in
// between.  p[q].x / p[q].y of the inner pairs are the buffers the pair-by-pair path would use (the host emulation
// replays the chain that way).  halo / margin / NT are filled by the launcher (chain_geom).
struct ChainArgs {
  PairArgs p[3];
  int32_t n = 3;
  Ragged rg;
  int32_t halo = 0, margin = 0, NT = 0;
};

struct ChainGeom { int halo, margin, NT; size_t lds; };
// tile of NF fragments (NF * 16 rows): halo rows per side are recomputed, NT = rows - 2 * halo frames come out
inline ChainGeom chain_geom(const ConvDesc* d1, int n, int NF) {
  ChainGeom g{0, 0, 0, 0};
  for (int q = 0; q < n; ++q) {
    const int h = (d1[q].taps - 1) / 2;
    g.halo += h * (d1[q].dil + 1);
    g.margin = std::max(g.margin, h * d1[q].dil);
  }
  g.NT = NF * 16 - 2 * g.halo;
  g.lds = (size_t)(NF * 16 + 2 * g.margin + NF * 16) * d1[0].CinP * 2;
  return g;
}

// Tile choice: the largest tile (fewest recomputed halo frames per produced frame) whose LDS fits the occupancy the
// layout's pair kernel has -- two 4-wave workgroups or one 8-wave workgroup per CU.
inline int chain_pick_nf(const ConvDesc* d1, int n, ChainGeom* out) {
  const int nwv = block_waves(d1[0]);
  const size_t budget = nwv == 8 ? 160 * 1024 : 80 * 1024;
  static const int nfs[] = {9, 8, 6, 5, 4};
  for (int NF : nfs) {
    if (d1[0].MF * NF * 4 > (d1[0].MF > 2 ? 80 : 96)) continue;  // accumulator registers (+ as many for the B double buffer)
    const ChainGeom g = chain_geom(d1, n, NF);
    if (g.lds <= budget && g.NT >= 64 && 4 * 2 * g.halo <= NF * 16) { *out = g; return NF; }   // <= 25 % of the tile recomputed ...
  }
  return 0;
}

inline bool chain_supported(const ConvDesc* d1, const ConvDesc* d2, int n) {
  if (n < 2 || n > 3) return false;
  for (int q = 0; q < n; ++q) {
    if (!pair_supported(d1[q], d2[q]) || !d1[q].lp || !d2[q].lp) return false;
    if (d1[q].MF != d1[0].MF || d1[q].WM != d1[0].WM || d1[q].CinP != d1[0].CinP || d1[q].taps != d1[0].taps || d1[q].M != d1[0].M) return false;
  }
  const int nwv = block_waves(d1[0]);
  if (d1[0].MF % 2 || nwv != d1[0].WM) return false;             // all waves along the rows, 16-byte epilogue pieces
  ChainGeom g;
  return chain_pick_nf(d1, n, &g) != 0;
}

// One fused WaveNet layer (modules.py:87-112): k-tap conv h->2h + conditioning + tanh*sigmoid gate, then the
// 1x1 h->2h whose first half is added to the residual stream x and second half to the skip accumulator
// (all of it to the accumulator on the last layer).  x is ping-ponged (x_in -> x_out) because neighbouring
// workgroups read each other's halo frames; oacc is updated in place.
struct WnArgs {
  const float* x_in = nullptr; float* x_out = nullptr; float* oacc = nullptr;
  int64_t bs = 0; int32_t T = 0, H = 0, HP = 0;
  const void* w_in = nullptr; const void* w_rs = nullptr; const float* b_rs = nullptr;
  const float* bbias = nullptr; int64_t bbias_bs = 0;
  int32_t taps = 1, KS = 1, nIt1 = 1, last = 0;
  Ragged rg;
};

// A whole WaveNet stack (modules.py:69-114) in one launch: every workgroup carries a 32-frame output tile
// plus (taps-1)/2 * layers halo frames per side through all layers (overlap-tiled: the halo is recomputed,
// which is free because the layer is bound by weight delivery into the CU, not by the MFMA pipe).
constexpr int kWnMaxLayers = 16;
constexpr int kWnOutFrames = 32;
struct WnStackArgs {
  const float* x0 = nullptr;   // [B][T][H] stack input (pre conv output)
  float* out = nullptr;        // [B][T][H] sum of the skip paths (the stack's output)
  int64_t bs = 0; int32_t T = 0, H = 0, HP = 0;
  const void* w_in[kWnMaxLayers] = {}; const void* w_rs[kWnMaxLayers] = {}; const float* b_rs[kWnMaxLayers] = {};
  const float* bbias = nullptr; int64_t bbias_bs = 0;     // + l*2H per layer
  int32_t layers = 0, taps = 1, KS = 1, nIt1 = 1;
  // a stack may be split over several launches (less halo to recompute per launch): x_out receives the residual
  // stream for the next launch, accum continues the skip sum already in `out`, final_layer marks the launch that
  // holds the network's last layer (whose 1x1 has no residual half)
  float* x_out = nullptr; int32_t accum = 0, final_layer = 1;
  // Optional fused 1x1 convs of a coupling layer (modules.py:212-217): with w_pre the stack input is computed in the
  // kernel, x0 = W_pre * z[:, pre_c0 : pre_c0+pre_cin] + b_pre; with w_post the kernel ends with
  // z[:, post_c0 : post_c0+post_m] -= W_post * out + b_post (in place; the two channel slices are disjoint).
  const void* w_pre = nullptr; const float* b_pre = nullptr; int32_t pre_cin = 0, pre_c0 = 0, pre_KS = 0;
  const void* w_post = nullptr; const float* b_post = nullptr; int32_t post_m = 0, post_c0 = 0, post_mf = 0;
  float* z = nullptr; int64_t z_bs = 0; int32_t z_ts = 0;
  float post_sign = -1.f;   // -1: reverse flow, x1 - m (modules.py:217); +1: forward flow, m + x1
  Ragged rg;
#ifdef QVC_STAMP
  unsigned long long* stamps = nullptr;   // developer build only (tools/conv_bench): [workgroup][16 waves][32] phase stamps
#endif
};

struct GemvArgs {
  const float* w; const float* bias; const float* g; float* out;
  int32_t rows, gin, batch;
};

struct SampleArgs {   // z = mu + noise * exp(logs)   (models.py:93-94)
  const float* stats; const float* noise; float* z;
  int32_t batch, frames, C;
};

struct TailArgs {     // models.py:394-406 / pqmf.py:106-117
  const float* post;  // [B][F][subbands*18]
  const float* fir;   // [subbands][63], gain folded
  float* out;         // [B][subbands*hop*(F-1)]
  float* y_mb;        // optional [B][subbands][hop*(F-1)]
  int32_t batch, F;
  Ragged rg;          // per-utterance frame count min(F, lens*mul + add); samples past it are written as zeros
};

// subband_conv_post + tail in one launch (qvc_post_tail_impl.h): `c` = the conv_post launch without an output
// pointer, the rest = TailArgs.  The post-conv frames stay in the CU.
struct PostTailArgs {
  ConvArgs c;
  const float* fir = nullptr;   // [subbands][63], gain folded
  float* out = nullptr;         // [B][16*(F-1)]
  int32_t F = 0;                // post-conv frames (= c.T_in + 1)
  Ragged rg;                    // as TailArgs::rg
};
// the fused kernel walks conv_post's packed weights as 2 waves x 3 row fragments and a 128-frame tile
inline bool post_tail_supported(const ConvDesc& d) {
  return d.M == 72 && d.MF == 3 && d.WM == 2 && d.nchunk == 1 && d.up_s == 1 && d.dil == 1 && !d.gau && !d.lp &&
         (int64_t)(128 + d.taps - 1) * d.CinP * 2 <= 96 * 1024;
}

// One LSTM layer's recurrence over all partials (models.py:510,516): gates = xp[t] + W_hh h[t-1], PyTorch gate
// order i,f,g,o.  The input projection xp (with b_ih + b_hh) comes from a conv launch.
struct LstmArgs {
  const float* xp;      // fp32, 4H per frame
  int32_t shared;       // 1: xp is [U][F][4H], partial p reads frames spk_start(p)+t; 0: xp is [P][S][4H]
  int32_t F, n_part, S, P, H;
  const void* w_hh;     // [wave][k-step][8 fragments][64 lanes][8], see spk_hh_row
  void* hseq;           // [P16][S][HP] operand type (P, H padded to 16 / 32): input of the next layer (null on the last)
  float* hfin;          // [P][H] fp32: h after the last step (last layer only)
};

struct SpkEmbedArgs {   // relu(linear(h)) / ||.||, mean over an utterance's partials (models.py:514-518,539)
  const float* hfin; const float* lw; const float* lb; float* g;
  int32_t utterances, n_part, H;
};

// Developer / test switches.  Process-wide integers that start at their production value and change ONLY through
// qvc_debug_set() (include/qvc.h): the library never reads the environment, so nothing inherited by a deployment
// can alter launch shapes or kernel selection.  The GPU tests flip them in-process to prove the variants equal.
enum DebugSwitch : int32_t {
  DBG_POST_TAIL = 0,        // 1 (default): conv_post + iSTFT / band synthesis as one launch where post_tail_supported(); 0: two launches
  DBG_POST_TAIL_NF,         // column fragments per wave of post_tail_kernel: 4 (default, 128-frame tile) or 2
  DBG_PAIR_WIDE_LAUNCH,     // 1 (default): the three chains of an 8-wave pair layout share a launch; 0: one chain per launch
  DBG_PAIR_CM4,             // 1 (default): chain-major grid for three-chain launches of 4-wave layouts; 0: chains interleaved (x % n)
  DBG_CONV_CL,              // 1 (default): chunk-loop variant of the conv kernel where it applies; 0: one workgroup per row chunk
  DBG_WN_CHUNK,             // 0 (default): 4 WaveNet layers per stack launch; n > 0: n layers; -1: one launch per layer, pre / post as convs
  DBG_PAIR_CHAIN3,          // 0 (default): off; 1: the three pairs of a short-kernel ResBlock chained in one launch (qvc_chain_impl.h:
                            //    bit-identical, measured SLOWER -- the k 3 pairs cost less riding in the three-chain launches)
  DBG_WN_KERNEL,            // WaveNet stack kernel variant (0 = default)
  DBG_COUNT
};
inline std::atomic<int32_t>* debug_table() {
  static std::atomic<int32_t> t[DBG_COUNT] = {{1}, {4}, {1}, {1}, {1}, {0}, {0}, {0}};
  return t;
}
inline int debug_get(int which) { return debug_table()[which].load(std::memory_order_relaxed); }
inline const char* const* debug_names() {
  static const char* const n[DBG_COUNT] = {"post_tail", "post_tail_nf", "pair_wide_launch", "pair_cm4", "conv_cl", "wn_chunk", "pair_chain3", "wn_kernel"};
  return n;
}

// Launchers return a QVC_* status.  `stream` is a hipStream_t.
int launch_conv(const ConvDesc& d, ConvArgs a, int batch, int epi, int dtype, void* stream, int* nf_out = nullptr);
int launch_pair(const ConvDesc& d1, const ConvDesc& d2, PairArgs a, int batch, int dtype, void* stream, int* nf_out = nullptr);
// n pairs (1..3) of equal shape class in one launch; d1[i] / d2[i] are chain i's convs
int launch_pair3(const ConvDesc* d1, const ConvDesc* d2, const PairArgs3& a, int batch, int dtype, void* stream, int* nf_out = nullptr);
// the n pairs of one ResBlock chained in one launch (chain_supported() says when); d1[q] / d2[q] are pair q's convs
int launch_chain(const ConvDesc* d1, const ConvDesc* d2, const ChainArgs& a, int batch, int dtype, void* stream, int* nf_out = nullptr);
bool wn_stack_supported(const ConvDesc& din, int layers);
int launch_wn_stack(const ConvDesc& din, const WnStackArgs& a, int batch, int dtype, void* stream);
int wn_stack_variant(const ConvDesc& din, const WnStackArgs& a);   // 2: the continuous-stream kernel (qvc_wn2_impl.h), 1: the generic one
int launch_wn(const ConvDesc& din, WnArgs a, int batch, int dtype, void* stream, int* nf_out = nullptr);
int launch_gemv(const GemvArgs& a, void* stream);

// ---- several strided copies / zero fills in ONE launch (the streaming step's ring hand-offs and slides: ~35 strided
//      memcpys of a few hundred KB each cost ~5 us apiece as separate graph nodes).  All sizes in bytes, multiples of
//      4; src == nullptr fills with zeros.  The regions written by one batch must not overlap anything the same batch
//      reads or writes -- the descriptors run concurrently.
struct CopyDesc {
  void* dst;
  const void* src;
  uint32_t dpitch, spitch, width, rows;
};
constexpr int kCopyBatchMax = 12;
int launch_copy_batch(const CopyDesc* d, int n, void* stream);
int launch_sample(const SampleArgs& a, void* stream);
int launch_tail(const TailArgs& a, void* stream);
int launch_post_tail(const ConvDesc& d, PostTailArgs a, int batch, int dtype, void* stream);

// Instantiation entry (one translation unit per operand dtype).
template <typename T> int launch_conv_typed(const ConvDesc& d, const ConvArgs& a, int batch, int epi, void* stream, int* nf_out);
template <typename T> int launch_wn_stack_typed(const ConvDesc& din, const WnStackArgs& a, int batch, void* stream);
template <typename T> int launch_wn_typed(const ConvDesc& din, const WnArgs& a, int batch, void* stream, int* nf_out);
template <typename T> int launch_post_tail_typed(const ConvDesc& d, const PostTailArgs& a, int batch, void* stream);
template <typename T, typename TS> int launch_chain_typed(const ConvDesc* d1, const ConvDesc* d2, ChainArgs a, int batch, void* stream, int* nf_out);
template <typename T, typename TS> int launch_pair_typed(const ConvDesc* d1, const PairArgs3& a, int batch, void* stream, int* nf_out);   // TS: stream type

}  // namespace qvc
