// qvc_spk.hip -- SpeakerEncoder.embed_utterance (models.py:507-546) on gfx950.
//
// The step of SynthesizerTrn.infer in front of the conversion path (models.py:635): a 3-layer LSTM over
// 128-frame mel partials, Linear + ReLU + L2 norm, mean over the partials.  torch.nn.LSTM on this GPU costs
// ~10 ms whatever the batch (3 x 128 sequential steps of several launches each) -- four times the whole
// 32-utterance conversion step.  Here every layer is two launches:
//   1. the input projection of ALL frames at once, xp = W_ih x + (b_ih + b_hh): a plain 1x1 conv through the
//      MFMA conv kernel (for layer 0 computed once per mel frame and shared by the overlapping partials);
//   2. one persistent launch that walks the time steps.  A workgroup carries 16 partials (one MFMA column
//      fragment) through all steps; wave w owns hidden units [32w, 32w+32) of the four gates, so
//      gates[4H x 16] = W_hh[4H x H] * h[H x 16] is 8 row fragments x H/32 k-steps per wave, the gate
//      nonlinearities and the cell state are lane-local, and only h (16 x H operand values, double-buffered
//      in LDS) crosses waves: one barrier per step.
// W_hh (512 KB in f16 at H = 256) equals the CU's whole register file, so it cannot stay resident: KREG k-steps
// of it live in registers, the rest is re-streamed from L2 every step through a register ring whose prefetches
// run across the step boundary (the stream is periodic and data-independent).  That stream -- not the MFMA
// pipe (64 MFMAs per wave per step) -- sets the step time.
#include <hip/hip_runtime.h>
#include "qvc_launch_util.h"
#include "qvc_conv_impl.h"
#include "qvc_path.h"

namespace qvc {

template <typename T, int KS, int KREG, int KLDS, int RING, bool LAST>
__global__ __launch_bounds__(512) void lstm_layer_kernel(const LstmArgs a) {
  using O = Op<T>;
  using frag = typename O::frag;
  constexpr int KRES = KREG + KLDS;         // resident k-steps: [0, KREG) in registers, [KREG, KRES) in LDS
  constexpr int KSS = KS - KRES;            // k-steps streamed per time step
  static_assert(KSS >= 1 && RING >= 1 && KSS % RING == 0, "ring slots must repeat every time step");
  constexpr int HP = KS * 32;
  constexpr int HPs = HP + 8;               // LDS row of one partial's h (+16 B: rows start in different banks)
  constexpr int NL = KLDS * 8;              // LDS-resident fragments per wave
  constexpr int LW = NL < 4 ? NL : 4;       // ... read through a rolling window of LW registers
  extern __shared__ __align__(16) char smem[];
  T* hb = reinterpret_cast<T*>(smem);       // [2][16][HPs]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int col = lane & 15, quad = lane >> 4;
  const int p = blockIdx.x * kSpkCols + col;
  const int pc = p < a.P ? p : a.P - 1;     // columns past the last partial replay it (their stores land in padding)
  const int H = a.H, H4 = 4 * H;

  const float* xb;
  if (a.shared) {
    const int u = pc / a.n_part, i = pc - u * a.n_part;
    const int start = i + 1 < a.n_part ? i * kSpkHop : (a.F > kSpkPartial ? a.F - kSpkPartial : 0);
    xb = a.xp + ((size_t)u * a.F + start) * H4;
  } else {
    xb = a.xp + (size_t)pc * a.S * H4;
  }
  // fragment f = 2*gate + half: rows gate*H + 32w + 16*half + 4*quad + (0..3).  Units >= H (H not a multiple of
  // 32) have zero weights; their h is forced to 0 and their loads are pointed at row 0 -- no branches in the step
  // loop, so hipcc can count the outstanding loads instead of draining them (s_waitcnt vmcnt(0)) at every use.
  const int unit0 = w * 32 + quad * 4;
  const bool live[2] = {unit0 < H, unit0 + 16 < H};
  int ro[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) ro[f] = live[f & 1] ? (f >> 1) * H + unit0 + (f & 1) * 16 : 0;

  const frag* wp = static_cast<const frag*>(a.w_hh) + (size_t)w * KS * 8 * 64 + lane;
  frag areg[KREG > 0 ? KREG : 1][8];
#pragma unroll
  for (int k = 0; k < KREG; ++k)
#pragma unroll
    for (int f = 0; f < 8; ++f) areg[k][f] = wp[(k * 8 + f) * 64];
  // LDS-resident k-steps: each wave keeps its own fragments, one 16-byte slot per lane (conflict-free b128 access)
  frag* wl = reinterpret_cast<frag*>(smem + (size_t)2 * kSpkCols * HPs * sizeof(T)) + (size_t)w * NL * 64 + lane;
#pragma unroll
  for (int i = 0; i < NL; ++i) wl[i * 64] = wp[(KREG * 8 + i) * 64];
  // streamed k-steps: slot s % RING is refilled with k-step s + RING right after the MFMAs that read it, so RING
  // k-steps (8 KiB per wave each) stay in flight through the gate phase and the barrier
  frag ring[RING][8];
#pragma unroll
  for (int u = 0; u < RING; ++u)
#pragma unroll
    for (int f = 0; f < 8; ++f) ring[u][f] = wp[((KRES + u) * 8 + f) * 64];

  for (int i = tid; i < kSpkCols * HPs; i += blockDim.x) reinterpret_cast<uint32_t*>(hb)[i] = 0u;   // h0 = 0 (both buffers)
  float c[2][4], hf[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 4; ++j) { c[h][j] = 0.f; hf[h][j] = 0.f; }

  f32x4 xv[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) xv[f] = *reinterpret_cast<const f32x4*>(xb + ro[f]);
  T* hs = LAST ? nullptr : static_cast<T*>(a.hseq) + (size_t)p * a.S * HP + unit0;   // [P16][S][HP]
  __syncthreads();

  for (int t = 0; t < a.S; ++t) {
    // the stream addresses repeat every step; keep hipcc from hoisting those loads out of the loop (it would
    // try to hold all of W_hh in registers and spill)
    // (an opaque OFFSET, not an opaque pointer: the latter loses the address space and turns the stream into
    // flat loads, which also count on lgkmcnt and serialise with the LDS reads)
    int wofs = 0;
    asm volatile("" : "+v"(wofs));
    const frag* wt = wp + wofs;
    const T* hcur = hb + (size_t)(t & 1) * kSpkCols * HPs + col * HPs + quad * 8;
    T* hnxt = hb + (size_t)((t + 1) & 1) * kSpkCols * HPs + col * HPs;
    f32x4 acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    // order: streamed k-steps first (prefetched during the previous gate phase), then LDS-, then register-resident
    auto kmap = [](int kk) { return kk < KSS ? KRES + kk : (kk - KSS < KLDS ? KREG + (kk - KSS) : kk - KSS - KLDS); };
    frag bq[2];
    bq[0] = *reinterpret_cast<const frag*>(hcur + kmap(0) * 32);
    frag lw[LW > 0 ? LW : 1];
#pragma unroll
    for (int i = 0; i < LW; ++i) lw[i] = wl[i * 64];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int ks = kmap(kk);
      if (kk + 1 < KS) bq[(kk + 1) & 1] = *reinterpret_cast<const frag*>(hcur + kmap(kk + 1) * 32);
      if (kk < KSS) {
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch distance: hipcc would sink the loads to their use
#pragma unroll
        for (int f = 0; f < 8; ++f) acc[f] = O::mfma(ring[kk % RING][f], bq[kk & 1], acc[f]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 8; ++f) ring[kk % RING][f] = wt[((KRES + (kk + RING) % KSS) * 8 + f) * 64];
        __builtin_amdgcn_sched_barrier(0);
      } else if (ks >= KREG) {
        if constexpr (NL > 0) {
#pragma unroll
          for (int f = 0; f < 8; ++f) {
            const int i = (ks - KREG) * 8 + f;
            acc[f] = O::mfma(lw[i % LW], bq[kk & 1], acc[f]);
            if (i + LW < NL) lw[i % LW] = wl[(i + LW) * 64];
          }
        }
      } else {
#pragma unroll
        for (int f = 0; f < 8; ++f) acc[f] = O::mfma(areg[ks][f], bq[kk & 1], acc[f]);
      }
    }
    // + input projection of this step, then fetch the next step's into the same registers.  Issued here, the
    // fetch (cold: xp is streamed from HBM when many partials are in flight) has the whole gate phase to land;
    // issued at the top of the step it would sit in front of the ring refills in the in-order return queue and
    // stall the MFMAs behind an HBM round trip (measured: 4.8 vs 3.4 us per step at 96 partials).
    {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < 8; ++f) acc[f] += xv[f];
      const float* xn = xb + (size_t)(t + 1 < a.S ? t + 1 : t) * H4;
#pragma unroll
      for (int f = 0; f < 8; ++f) xv[f] = *reinterpret_cast<const f32x4*>(xn + ro[f]);
    }
    // gates (torch.nn.LSTM order i, f, g, o), cell and hidden state of this lane's 2 x 4 units
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      typename O::quad hq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gi = fast_sigmoid(acc[0 + h][j]), gf = fast_sigmoid(acc[2 + h][j]);
        const float gg = fast_tanh(acc[4 + h][j]), go = fast_sigmoid(acc[6 + h][j]);
        c[h][j] = gf * c[h][j] + gi * gg;
        hf[h][j] = live[h] ? go * fast_tanh(c[h][j]) : 0.f;
        hq[j] = O::cvt(hf[h][j]);
      }
      *reinterpret_cast<typename O::quad*>(hnxt + unit0 + h * 16) = hq;
      if constexpr (!LAST) *reinterpret_cast<typename O::quad*>(hs + (size_t)t * HP + h * 16) = hq;
    }
    __syncthreads();
  }
  if constexpr (LAST) {
    if (p < a.P) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (live[h]) *reinterpret_cast<float4*>(a.hfin + (size_t)p * H + unit0 + h * 16) = make_float4(hf[h][0], hf[h][1], hf[h][2], hf[h][3]);
    }
  }
}

// embeds = relu(linear(h)) / ||.||_2 per partial (models.py:514-518), mean over the utterance's partials
// (models.py:539), not re-normalised (:540).  One workgroup per utterance.
__global__ __launch_bounds__(256) void spk_embed_kernel(const SpkEmbedArgs a) {
  const float* hfin = a.hfin; const float* lw = a.lw; const float* lb = a.lb; float* g = a.g;
  const int n_part = a.n_part, H = a.H;
  __shared__ float s_h[512];
  __shared__ float s_red[4];
  const int u = blockIdx.x, tid = threadIdx.x;
  float mean[2] = {0.f, 0.f};
  for (int i = 0; i < n_part; ++i) {
    __syncthreads();
    for (int k = tid; k < H; k += 256) s_h[k] = hfin[((size_t)u * n_part + i) * H + k];
    __syncthreads();
    float e[2] = {0.f, 0.f};
    float ss = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = tid + r * 256;
      if (row < H) {
        const float* wr = lw + (size_t)row * H;
        float s = lb[row];
#pragma unroll 16                                     // 16 weight loads in flight (one per trip = a cache round trip each)
        for (int k = 0; k < H; k += 4) {
          const float4 w4 = *reinterpret_cast<const float4*>(wr + k);
          s = fmaf(w4.x, s_h[k], s); s = fmaf(w4.y, s_h[k + 1], s); s = fmaf(w4.z, s_h[k + 2], s); s = fmaf(w4.w, s_h[k + 3], s);
        }
        e[r] = s > 0.f ? s : 0.f;
        ss += e[r] * e[r];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((tid & 63) == 0) s_red[tid >> 6] = ss;
    __syncthreads();
    const float norm = sqrtf(s_red[0] + s_red[1] + s_red[2] + s_red[3]);
    mean[0] += e[0] / norm; mean[1] += e[1] / norm;       // 0/0 = NaN exactly as the reference's division
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = tid + r * 256;
    if (row < H) g[(size_t)u * H + row] = mean[r] / (float)n_part;
  }
}

template <typename T, int KS, int KREG, int KLDS, int RING, bool LAST>
static int launch_lstm_last(const LstmArgs& a, hipStream_t stream) {
  constexpr int HPs = KS * 32 + 8;
  const size_t lds = (size_t)2 * kSpkCols * HPs * sizeof(T) + (size_t)KS * KLDS * 8 * 1024;
  auto kern = lstm_layer_kernel<T, KS, KREG, KLDS, RING, LAST>;
  static std::atomic<uint32_t> lds_ok{0};                  // > 64 KiB dynamic LDS: opt in once per device
  if (!allow_big_lds(lds_ok, reinterpret_cast<const void*>(kern))) return QVC_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(a.P, kSpkCols)), dim3((unsigned)KS * 64), lds, stream, a);
  return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
}

template <typename T, int KS, int KREG, int KLDS, int RING>
static int launch_lstm_variant(const LstmArgs& a, hipStream_t stream) {
  if ((a.hseq == nullptr) == (a.hfin == nullptr)) return QVC_ERR_BAD_ARG;
  return a.hseq ? launch_lstm_last<T, KS, KREG, KLDS, RING, false>(a, stream) : launch_lstm_last<T, KS, KREG, KLDS, RING, true>(a, stream);
}

template <typename T, int KS>
static int launch_lstm_ks(const LstmArgs& a, hipStream_t stream) {
  // <KREG, KLDS, RING>: k-steps resident in registers / in LDS (64 KiB each at 8 waves), ring slots of the rest
  // Measured on MI355X at H = 256 (3 layers x 128 steps, one workgroup): all streamed 1.99 ms; 2 k-steps in LDS
  // 1.68 ms; 2 in LDS + 2 in registers 1.24 ms (more residency spills).  Per step (3.0 us): ~0.4 us per streamed
  // k-step (64 KiB through the 64 B/clk L1), 0.3 us for the xp fetch (another 64 KiB), 0.25 us gate math, the
  // rest 64 MFMAs per wave + LDS round trip + barrier (ablations: no gates 1.15 ms, no xp fetch 1.14 ms).
  if constexpr (KS == 8) return launch_lstm_variant<T, 8, 2, 2, 2>(a, stream);
  else if constexpr (KS % 4 == 0) return launch_lstm_variant<T, KS, 0, 0, 4>(a, stream);
  else if constexpr (KS % 2 == 0) return launch_lstm_variant<T, KS, 0, 0, 2>(a, stream);
  else return launch_lstm_variant<T, KS, 0, 0, 1>(a, stream);
}

template <typename T>
static int launch_lstm(const LstmArgs& a, int KS, hipStream_t stream) {
  switch (KS) {
    case 1: return launch_lstm_ks<T, 1>(a, stream);
    case 2: return launch_lstm_ks<T, 2>(a, stream);
    case 3: return launch_lstm_ks<T, 3>(a, stream);
    case 4: return launch_lstm_ks<T, 4>(a, stream);
    case 5: return launch_lstm_ks<T, 5>(a, stream);
    case 6: return launch_lstm_ks<T, 6>(a, stream);
    case 7: return launch_lstm_ks<T, 7>(a, stream);
    case 8: return launch_lstm_ks<T, 8>(a, stream);
  }
  return QVC_ERR_BAD_CONFIG;
}

struct SpkHipBackend {
  hipStream_t stream;
  int conv(const ConvDesc& d, const ConvArgs& a, int batch, int epi, int dtype) { return launch_conv(d, a, batch, epi, dtype, stream); }
  int lstm(const LstmArgs& a, int KS, int dtype) {
    return dtype == QVC_F16 ? launch_lstm<_Float16>(a, KS, stream) : launch_lstm<__bf16>(a, KS, stream);
  }
  int spk_embed(const SpkEmbedArgs& a) {
    hipLaunchKernelGGL(spk_embed_kernel, dim3((unsigned)a.utterances), dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
  }
};

}  // namespace qvc

using namespace qvc;

extern "C" int64_t qvc_spk_workspace_bytes(const qvc_config* cfg, int32_t utterances, int32_t mel_frames) {
  if (!cfg || utterances <= 0 || mel_frames <= 0) return QVC_ERR_BAD_ARG;
  SpkPlan S = build_spk_plan(*cfg);
  if (S.status != QVC_OK) return S.status;
  return carve_spk_workspace(S, utterances, mel_frames).bytes;
}

extern "C" int qvc_speaker_embed(const qvc_config* cfg, const void* spk_blob_dev, const float* mel, float* g,
                                 int32_t utterances, int32_t mel_frames,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
  if (!cfg || !spk_blob_dev || !mel || !g || !workspace || utterances <= 0 || mel_frames <= 0) return QVC_ERR_BAD_ARG;
  SpkPlan S = build_spk_plan(*cfg);
  if (S.status != QVC_OK) return S.status;
  const int U = utterances, F = mel_frames;
  const SpkWorkspace W = carve_spk_workspace(S, U, F);
  if (workspace_bytes < W.bytes) return QVC_ERR_SMALL_BUFFER;
  if ((int64_t)U * spk_partials(F) > (1 << 20)) return QVC_ERR_BAD_ARG;
  SpkHipBackend be{static_cast<hipStream_t>(stream)};
  return spk_path(S, dec_dtype(*cfg), static_cast<const char*>(spk_blob_dev), static_cast<char*>(workspace), W, mel, g,
                  U, F, be);
}
#ifdef QVC_SATCOUNT
namespace qvc { QVC_SAT_READER(sat_count_spk) }
#endif
