// qvc_pack.cpp -- host-side weight packer (qvc_blob_bytes / qvc_pack_weights).
//
// Input: the reference checkpoint's state_dict as named fp32 tensors (utils.py:183-193).
// Output: one blob laid out by qvc_plan.h.  Work done here once instead of on every
// forward as the reference does:
//   * weight_norm fold  w = g * v / ||v||  (modules.py:54,64,67,134-143; models.py:327,333,346,357)
//   * the channel Flips of the flow (modules.py:165-170) folded into the coupling layers'
//     pre/post weights (see FlowStepPlan)
//   * ConvTranspose1d (models.py:333-335) rewritten as `stride` polyphase filters
//   * in_layer bias + cond_layer folded into one conditioning table (modules.py:84,91-101)
//   * conversion to the MFMA operand type and per-wave fragment-stream order.
#include <cmath>
#include <functional>
#include <string>
#include <unordered_map>
#include "qvc_plan.h"
#include "qvc_pack_util.h"

namespace {

using namespace qvc;

uint16_t to_bf16(float f) {
  uint32_t u; std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;       // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);                          // round to nearest even
  return (uint16_t)(u >> 16);
}

uint16_t to_f16(float f) {
  uint32_t u; std::memcpy(&u, &f, 4);
  uint32_t sign = (u >> 16) & 0x8000u;
  uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);   // NaN
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7bffu);  // >= 65520 would round to inf: saturate to 65504
  if (a < 0x33000001u) return (uint16_t)sign;               // < 2^-25: rounds to zero
  int32_t exp = (int32_t)(a >> 23) - 127;
  uint32_t man = (a & 0x7fffffu) | 0x800000u;
  if (exp < -14) {                                          // subnormal half
    int shift = -14 - exp + 13;                             // bits to drop from the 24-bit significand
    uint32_t half = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u), mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1u))) ++half;
    return (uint16_t)(sign | half);
  }
  uint32_t half = ((uint32_t)(exp + 15) << 10) | ((man >> 13) & 0x3ffu);
  uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) ++half;   // carries into the exponent correctly
  return (uint16_t)(sign | half);
}

struct Packer {
  const qvc_config& cfg;
  std::unordered_map<std::string, const qvc_tensor*> tensors;
  char* blob;
  int status = QVC_OK;
  std::string missing;

  const qvc_tensor* find(const std::string& name) {
    auto it = tensors.find(name);
    if (it == tensors.end()) { if (status == QVC_OK) { status = QVC_ERR_MISSING_TENSOR; missing = name; } return nullptr; }
    return it->second;
  }
  static int64_t numel(const qvc_tensor* t) { int64_t n = 1; for (int i = 0; i < t->ndim; ++i) n *= t->shape[i]; return n; }

  // Effective weight of a conv stored under `prefix` as fp32 [d0][d1][d2]; checks the shape.
  std::vector<float> weight(const std::string& prefix, int64_t d0, int64_t d1, int64_t d2) {
    std::vector<float> w((size_t)(d0 * d1 * d2), 0.f);
    auto check = [&](const qvc_tensor* t) {
      if (!t) return false;
      if (t->ndim != 3 || t->shape[0] != d0 || t->shape[1] != d1 || t->shape[2] != d2) {
        if (status == QVC_OK) { status = QVC_ERR_BAD_SHAPE; missing = prefix; }
        return false;
      }
      return true;
    };
    if (tensors.count(prefix + ".weight_v")) {
      const qvc_tensor* v = find(prefix + ".weight_v");
      const qvc_tensor* g = find(prefix + ".weight_g");
      if (!check(v) || !g) return w;
      if (numel(g) != d0) { if (status == QVC_OK) { status = QVC_ERR_BAD_SHAPE; missing = prefix + ".weight_g"; } return w; }
      const int64_t inner = d1 * d2;
      for (int64_t r = 0; r < d0; ++r) {
        double ss = 0.0;
        for (int64_t i = 0; i < inner; ++i) { double x = v->data[r * inner + i]; ss += x * x; }
        const double scale = (double)g->data[r] / std::sqrt(ss);
        for (int64_t i = 0; i < inner; ++i) w[(size_t)(r * inner + i)] = (float)(v->data[r * inner + i] * scale);
      }
    } else {
      const qvc_tensor* t = find(prefix + ".weight");
      if (!check(t)) return w;
      std::memcpy(w.data(), t->data, w.size() * 4);
    }
    return w;
  }
  std::vector<float> bias(const std::string& prefix, int64_t n) {
    std::vector<float> b((size_t)n, 0.f);
    const qvc_tensor* t = find(prefix + ".bias");
    if (!t) return b;
    if (numel(t) != n) { if (status == QVC_OK) { status = QVC_ERR_BAD_SHAPE; missing = prefix + ".bias"; } return b; }
    std::memcpy(b.data(), t->data, (size_t)n * 4);
    return b;
  }

  int cur_dtype = -1;     // operand type of the section being packed (-1: the config's WaveNet-half type)
  uint16_t cvt(float f) const { return (cur_dtype < 0 ? wn_dtype(cfg) : cur_dtype) == QVC_F16 ? to_f16(f) : to_bf16(f); }

  // wv(v, tap, ci): virtual weight; bv(v): bias (ignored when the descriptor carries no bias slot).
  void pack(const ConvDesc& d, const std::function<float(int, int, int)>& wv, const std::function<float(int)>& bv) {
    uint16_t* dst = reinterpret_cast<uint16_t*>(blob + d.w_off);
    const int KS = d.KS(), nIt = d.nIt();
    for (int chunk = 0; chunk < d.nchunk; ++chunk)
      for (int wave = 0; wave < d.WM; ++wave)
        for (int it = 0; it < nIt; ++it) {
          const int tap = it / KS, ks = it % KS;
          for (int mf = 0; mf < d.MF; ++mf)
            for (int lane = 0; lane < 64; ++lane) {
              const int v = conv_row(d, chunk, wave, mf, lane & 15);
              for (int j = 0; j < 8; ++j) {
                const int ci = ks * kKStep + (lane >> 4) * 8 + j;
                const float val = (v >= 0 && ci < d.Cin) ? wv(v, tap, ci) : 0.f;
                *dst++ = cvt(val);
              }
            }
        }
    if (d.b_off >= 0) {
      float* b = reinterpret_cast<float*>(blob + d.b_off);
      for (int v = 0; v < d.MP(); ++v) b[v] = v < d.M ? bv(v) : 0.f;
    }
  }

  // Plain Conv1d weight [Cout][Cin][k] stored under `prefix`.
  void pack_conv1d(const ConvDesc& d, const std::string& prefix, bool has_bias = true) {
    auto w = weight(prefix, d.M, d.Cin, d.taps);
    auto b = has_bias ? bias(prefix, d.M) : std::vector<float>((size_t)d.M, 0.f);
    if (status != QVC_OK) return;
    pack(d, [&](int v, int tap, int ci) { return w[((size_t)v * d.Cin + ci) * d.taps + tap]; },
         [&](int v) { return b[(size_t)v]; });
  }

  void pack_wn(const WNPlan& wn, const std::string& prefix) {
    for (int i = 0; i < wn.layers; ++i) {
      pack_conv1d(wn.in_conv[i], prefix + ".in_layers." + std::to_string(i), /*has_bias=*/false);
      pack_conv1d(wn.rs_conv[i], prefix + ".res_skip_layers." + std::to_string(i));
    }
  }
};

}  // namespace

namespace qvc {
void pack_plain_conv(const ConvDesc& d, const float* w, const float* bias, int dtype, char* base) {
  qvc_config cfg{};
  cfg.operand_dtype = dtype;
  Packer pk{cfg, {}, base};
  pk.pack(d, [&](int v, int tap, int ci) { return w[((size_t)v * d.Cin + ci) * d.taps + tap]; },
          [&](int v) { return bias ? bias[v] : 0.f; });
}
}  // namespace qvc

extern "C" int64_t qvc_blob_bytes(const qvc_config* cfg) {
  if (!cfg) return QVC_ERR_BAD_ARG;
  qvc::Plan P = qvc::build_plan(*cfg);
  return P.status == QVC_OK ? P.blob_bytes : (int64_t)P.status;
}

extern "C" int qvc_pack_weights(const qvc_config* cfg, const qvc_tensor* tensors, int32_t n_tensors,
                                void* blob_host, int64_t blob_bytes) {
  if (!cfg || !tensors || n_tensors <= 0 || !blob_host) return QVC_ERR_BAD_ARG;
  Plan P = build_plan(*cfg);
  if (P.status != QVC_OK) return P.status;
  if (blob_bytes < P.blob_bytes) return QVC_ERR_SMALL_BUFFER;
  std::memset(blob_host, 0, (size_t)P.blob_bytes);
  Packer pk{*cfg, {}, static_cast<char*>(blob_host)};
  for (int i = 0; i < n_tensors; ++i) {
    if (!tensors[i].name || !tensors[i].data || tensors[i].ndim < 0 || tensors[i].ndim > 4) return QVC_ERR_BAD_ARG;
    pk.tensors[tensors[i].name] = &tensors[i];
  }
  const int H = cfg->hidden_channels, C = cfg->inter_channels, half = C / 2, gin = cfg->gin_channels;
  pk.cur_dtype = wn_dtype(*cfg);         // ---- WaveNet half

  // ---- enc_p (models.py:71-73,583)
  pk.pack_conv1d(P.enc_pre, "enc_p.pre");
  pk.pack_wn(P.enc_wn, "enc_p.enc");
  for (int i = 0; i < cfg->enc_layers && pk.status == QVC_OK; ++i) {
    auto b = pk.bias("enc_p.enc.in_layers." + std::to_string(i), 2 * H);
    std::memcpy(pk.blob + P.enc_wn.inbias_off + (int64_t)i * 2 * H * 4, b.data(), (size_t)2 * H * 4);
  }
  pk.pack_conv1d(P.enc_proj, "enc_p.proj");

  // ---- flow (models.py:33-37; modules.py:173-224) + its rows of the cond table
  float* cond_w = reinterpret_cast<float*>(pk.blob + P.cond_w_off);
  float* cond_b = reinterpret_cast<float*>(pk.blob + P.cond_b_off);
  for (const FlowStepPlan& f : P.flow) {
    if (pk.status != QVC_OK) break;
    const std::string p = "flow.flows." + std::to_string(2 * f.layer);
    {   // pre: logical input channel c lives at physical slice channel (flipped ? half-1-c : c)
      auto w = pk.weight(p + ".pre", H, half, 1);
      auto b = pk.bias(p + ".pre", H);
      if (pk.status != QVC_OK) break;
      const bool fl = f.flipped;
      pk.pack(f.pre, [&](int v, int, int ci) { return w[(size_t)v * half + (fl ? half - 1 - ci : ci)]; },
              [&](int v) { return b[(size_t)v]; });
    }
    pk.pack_wn(f.wn, p + ".enc");
    {   // post: logical output channel c' updates physical slice channel (flipped ? half-1-c' : c')
      auto w = pk.weight(p + ".post", half, H, 1);
      auto b = pk.bias(p + ".post", half);
      if (pk.status != QVC_OK) break;
      const bool fl = f.flipped;
      pk.pack(f.post, [&](int v, int, int ci) { return w[(size_t)(fl ? half - 1 - v : v) * H + ci]; },
              [&](int v) { return b[(size_t)(fl ? half - 1 - v : v)]; });
    }
    {   // cond rows: cond_layer (weight-normed 1x1, modules.py:54) + the in_layer biases
      auto w = pk.weight(p + ".enc.cond_layer", (int64_t)cfg->flow_layers * 2 * H, gin, 1);
      auto b = pk.bias(p + ".enc.cond_layer", (int64_t)cfg->flow_layers * 2 * H);
      if (pk.status != QVC_OK) break;
      for (int l = 0; l < cfg->flow_layers; ++l) {
        auto ib = pk.bias(p + ".enc.in_layers." + std::to_string(l), 2 * H);
        for (int r = 0; r < 2 * H; ++r) {
          const int64_t row = f.cond_row0 + (int64_t)l * 2 * H + r, src = (int64_t)l * 2 * H + r;
          std::memcpy(cond_w + row * gin, w.data() + src * gin, (size_t)gin * 4);
          cond_b[row] = b[(size_t)src] + ib[(size_t)r];
        }
      }
    }
  }

  // ---- decoder (models.py:327-357)
  pk.cur_dtype = dec_dtype(*cfg);        // ---- generator half
  if (pk.status == QVC_OK) {
    auto w = pk.weight("dec.conv_pre", cfg->upsample_initial_channel, C, 7);
    if (pk.status == QVC_OK)
      pk.pack(P.conv_pre, [&](int v, int tap, int ci) { return w[((size_t)v * C + ci) * 7 + tap]; },
              [&](int) { return 0.f; });                       // bias rides in the cond table
    auto cw = pk.weight("dec.cond", cfg->upsample_initial_channel, gin, 1);
    auto cb = pk.bias("dec.cond", cfg->upsample_initial_channel);
    auto pb = pk.bias("dec.conv_pre", cfg->upsample_initial_channel);
    if (pk.status == QVC_OK)
      for (int r = 0; r < cfg->upsample_initial_channel; ++r) {
        std::memcpy(cond_w + (int64_t)(P.dec_cond_row0 + r) * gin, cw.data() + (int64_t)r * gin, (size_t)gin * 4);
        cond_b[P.dec_cond_row0 + r] = cb[(size_t)r] + pb[(size_t)r];
      }
  }
  int ch = cfg->upsample_initial_channel;
  for (size_t i = 0; i < P.stages.size() && pk.status == QVC_OK; ++i) {
    const StagePlan& st = P.stages[i];
    const ConvDesc& u = st.up;
    const int k = cfg->upsample_kernel_sizes[i], s = u.up_s, cout = u.Cout, cin = ch;
    auto w = pk.weight("dec.ups." + std::to_string(i), cin, cout, k);       // (Cin, Cout, K)
    auto b = pk.bias("dec.ups." + std::to_string(i), cout);
    if (pk.status != QVC_OK) break;
    // virtual row v = r*Cout + co, tap m' reads x[q + m' - (taps-1)] and uses w[ci][co][r + s*(taps-1-m')]
    pk.pack(u, [&](int v, int tap, int ci) {
              const int r = v / cout, co = v % cout, j = r + s * (u.taps - 1 - tap);
              return j < k ? w[((size_t)ci * cout + co) * k + j] : 0.f; },
            [&](int v) { return b[(size_t)(v % cout)]; });
    ch = cout;
    for (int j = 0; j < cfg->n_resblocks; ++j)
      for (int q = 0; q < 3; ++q) {
        const std::string rb = "dec.resblocks." + std::to_string(i * cfg->n_resblocks + j);
        // fused pairs take their own operand type (bf16 in the mixed mode); the unfused fallback runs as generator convs
        const bool fusedp = pair_supported(st.c1[(size_t)j * 3 + q], st.c2[(size_t)j * 3 + q]) && st.c1[(size_t)j * 3 + q].lp;
        pk.cur_dtype = fusedp ? pair_weight_dtype(*cfg) : dec_dtype(*cfg);
        pk.pack_conv1d(st.c1[(size_t)j * 3 + q], rb + ".convs1." + std::to_string(q));
        pk.pack_conv1d(st.c2[(size_t)j * 3 + q], rb + ".convs2." + std::to_string(q));
        pk.cur_dtype = dec_dtype(*cfg);
      }
  }
  if (pk.status == QVC_OK) pk.pack_conv1d(P.conv_post, "dec.subband_conv_post");

  // ---- synthesis FIR (1, subbands, taps): learned (models.py:357) or fixed PQMF (pqmf.py:65-76,83);
  //      the zero-stuffing gain `subbands` (models.py:405, pqmf.py:116) is folded in.
  if (pk.status == QVC_OK) {
    std::vector<float> fir;
    if (cfg->decoder == QVC_DEC_MULTISTREAM) {
      fir = pk.weight("dec.multistream_conv_post", 1, cfg->subbands, cfg->fir_taps);
    } else {
      const qvc_tensor* t = pk.find("dec.pqmf.synthesis_filter");
      if (t) {
        if (Packer::numel(t) != (int64_t)cfg->subbands * cfg->fir_taps) { pk.status = QVC_ERR_BAD_SHAPE; }
        else fir.assign(t->data, t->data + Packer::numel(t));
      }
    }
    if (pk.status == QVC_OK) {
      float* dst = reinterpret_cast<float*>(pk.blob + P.fir_off);
      for (size_t i = 0; i < fir.size(); ++i) dst[i] = fir[i] * (float)cfg->subbands;
    }
  }
  return pk.status;
}

// ---------------------------------------------------------------- enc_q blob (models.py:582: CondNormalWN with gin)
extern "C" int64_t qvc_encq_blob_bytes(const qvc_config* cfg) {
  if (!cfg) return QVC_ERR_BAD_ARG;
  EncQPlan Q = build_encq_plan(*cfg);
  return Q.status == QVC_OK ? Q.blob_bytes : (int64_t)Q.status;
}

extern "C" int qvc_encq_pack_weights(const qvc_config* cfg, const qvc_tensor* tensors, int32_t n_tensors,
                                     void* blob_host, int64_t blob_bytes) {
  if (!cfg || !tensors || n_tensors <= 0 || !blob_host) return QVC_ERR_BAD_ARG;
  EncQPlan Q = build_encq_plan(*cfg);
  if (Q.status != QVC_OK) return Q.status;
  if (blob_bytes < Q.blob_bytes) return QVC_ERR_SMALL_BUFFER;
  std::memset(blob_host, 0, (size_t)Q.blob_bytes);
  Packer pk{*cfg, {}, static_cast<char*>(blob_host)};
  for (int i = 0; i < n_tensors; ++i) {
    if (!tensors[i].name || !tensors[i].data || tensors[i].ndim < 0 || tensors[i].ndim > 4) return QVC_ERR_BAD_ARG;
    pk.tensors[tensors[i].name] = &tensors[i];
  }
  const int H = cfg->hidden_channels, gin = cfg->gin_channels, L = cfg->enc_layers;
  pk.pack_conv1d(Q.pre, "enc_q.pre");
  pk.pack_wn(Q.wn, "enc_q.enc");
  pk.pack_conv1d(Q.proj, "enc_q.proj");
  if (pk.status != QVC_OK) return pk.status;
  // conditioning rows: cond_layer (weight-normed 1x1 on g, modules.py:54,84) + the in_layer biases (modules.py:91)
  float* cond_w = reinterpret_cast<float*>(pk.blob + Q.cond_w_off);
  float* cond_b = reinterpret_cast<float*>(pk.blob + Q.cond_b_off);
  auto w = pk.weight("enc_q.enc.cond_layer", (int64_t)L * 2 * H, gin, 1);
  auto b = pk.bias("enc_q.enc.cond_layer", (int64_t)L * 2 * H);
  if (pk.status != QVC_OK) return pk.status;
  std::memcpy(cond_w, w.data(), w.size() * 4);
  for (int l = 0; l < L && pk.status == QVC_OK; ++l) {
    auto ib = pk.bias("enc_q.enc.in_layers." + std::to_string(l), 2 * H);
    for (int r = 0; r < 2 * H; ++r) cond_b[(size_t)l * 2 * H + r] = b[(size_t)l * 2 * H + r] + ib[(size_t)r];
  }
  return pk.status;
}

// ---------------------------------------------------------------- speaker encoder blob (models.py:507-518)
extern "C" int64_t qvc_spk_blob_bytes(const qvc_config* cfg) {
  if (!cfg) return QVC_ERR_BAD_ARG;
  SpkPlan S = build_spk_plan(*cfg);
  return S.status == QVC_OK ? S.blob_bytes : (int64_t)S.status;
}

extern "C" int qvc_spk_pack_weights(const qvc_config* cfg, const qvc_tensor* tensors, int32_t n_tensors,
                                    void* blob_host, int64_t blob_bytes) {
  if (!cfg || !tensors || n_tensors <= 0 || !blob_host) return QVC_ERR_BAD_ARG;
  SpkPlan S = build_spk_plan(*cfg);
  if (S.status != QVC_OK) return S.status;
  if (blob_bytes < S.blob_bytes) return QVC_ERR_SMALL_BUFFER;
  std::memset(blob_host, 0, (size_t)S.blob_bytes);
  Packer pk{*cfg, {}, static_cast<char*>(blob_host)};
  pk.cur_dtype = dec_dtype(*cfg);         // the speaker encoder runs in the generator half's operand type
  for (int i = 0; i < n_tensors; ++i) {
    if (!tensors[i].name || !tensors[i].data || tensors[i].ndim < 0 || tensors[i].ndim > 4) return QVC_ERR_BAD_ARG;
    pk.tensors[tensors[i].name] = &tensors[i];
  }
  auto matrix = [&](const std::string& name, int64_t rows, int64_t cols) -> const float* {
    const qvc_tensor* t = pk.find(name);
    if (!t) return nullptr;
    if (t->ndim != 2 || t->shape[0] != rows || t->shape[1] != cols) {
      if (pk.status == QVC_OK) { pk.status = QVC_ERR_BAD_SHAPE; pk.missing = name; }
      return nullptr;
    }
    return t->data;
  };
  auto vec = [&](const std::string& name, int64_t n) -> const float* {
    const qvc_tensor* t = pk.find(name);
    if (!t) return nullptr;
    if (Packer::numel(t) != n) { if (pk.status == QVC_OK) { pk.status = QVC_ERR_BAD_SHAPE; pk.missing = name; } return nullptr; }
    return t->data;
  };
  const int H = S.H;
  for (int l = 0; l < kSpkLayers; ++l) {
    const std::string sfx = "_l" + std::to_string(l);
    const int cin = l == 0 ? S.n_mel : H;
    const float* wih = matrix("enc_spk.lstm.weight_ih" + sfx, 4 * H, cin);
    const float* whh = matrix("enc_spk.lstm.weight_hh" + sfx, 4 * H, H);
    const float* bih = vec("enc_spk.lstm.bias_ih" + sfx, 4 * H);
    const float* bhh = vec("enc_spk.lstm.bias_hh" + sfx, 4 * H);
    if (pk.status != QVC_OK) return pk.status;
    // input projection with both biases folded (torch.nn.LSTM adds b_ih + b_hh to every gate pre-activation)
    pk.pack(S.ih[l], [&](int v, int, int ci) { return wih[(size_t)v * cin + ci]; },
            [&](int v) { return bih[v] + bhh[v]; });
    // recurrent weights: [wave][k-step][fragment 2*gate+half][lane][8]
    uint16_t* dst = reinterpret_cast<uint16_t*>(pk.blob + S.hh_off[l]);
    for (int w = 0; w < S.NW; ++w)
      for (int ks = 0; ks < S.KS; ++ks)
        for (int f = 0; f < 8; ++f)
          for (int lane = 0; lane < 64; ++lane) {
            const int row = spk_hh_row(S, w, f, lane & 15);
            for (int j = 0; j < 8; ++j) {
              const int k = ks * kKStep + (lane >> 4) * 8 + j;
              *dst++ = pk.cvt(row >= 0 && k < H ? whh[(size_t)row * H + k] : 0.f);
            }
          }
  }
  const float* lw = matrix("enc_spk.linear.weight", H, H);
  const float* lb = vec("enc_spk.linear.bias", H);
  if (pk.status != QVC_OK) return pk.status;
  std::memcpy(pk.blob + S.lin_w_off, lw, (size_t)H * H * 4);
  std::memcpy(pk.blob + S.lin_b_off, lb, (size_t)H * 4);
  return QVC_OK;
}
