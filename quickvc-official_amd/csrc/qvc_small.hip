// qvc_small.hip -- the HBM-bound kernels around the conv trunk:
//   * cond_gemv_kernel   : all 1x1 conditioning convs on g (modules.py:54,84; models.py:328,372)
//   * sample_kernel      : z_p = mu + noise*exp(logs) (models.py:93-94), noise read in (B,C,T)
//   * istft_synth_kernel : exp / pi*sin / 16-point inverse real DFT / Hann overlap-add /
//                          envelope / x4 zero-stuff / 63-tap synthesis FIR in ONE pass
//                          (models.py:394-406, pqmf.py:106-117) -- no complex tensors, no
//                          zero-stuffed intermediate, each post-conv frame read once (+halo).
//   * fm_to_cm_kernel    : frame-major -> (B,C,T) for the unit-test conv entry point.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "qvc_launch_util.h"
#include "qvc_kernels.h"
#include "qvc_tail_impl.h"

namespace qvc {

// ------------------------------------------------------------------ cond GEMV
// out[b][row] = bias[row] + sum_k w[row][k] * g[b][k].  A workgroup owns kGR rows x kGB utterances: its weight rows
// (contiguous in memory) and the g vectors ([k][b], so the 32 utterances of one k sit in 32 different banks) are
// staged into LDS with ALL loads of a thread in flight at once -- one memory round trip per workgroup -- and each
// thread then owns one (row, utterance) pair and walks k out of LDS.  ~0.1 GFLOP in total: latency-sized, it only
// has to stay out of the way.  (Round 1 walked the weight row from global memory, 16 loads at a time: four
// dependent round trips per workgroup, 16-19 us.)
constexpr int kGR = 16, kGB = 32, kGS = kGB + 1, kGT = kGR * kGB;
// the g region doubles as the [kGB][kGR + 1] output transpose buffer: small gin must not let that run into the weights
__host__ __device__ constexpr int gemv_g_floats(int gin) { return gin * kGS > kGB * (kGR + 1) ? gin * kGS : kGB * (kGR + 1); }
__global__ __launch_bounds__(kGT) void cond_gemv_kernel(const GemvArgs a) {
  extern __shared__ float s_gv[];                       // [max(gin*kGS, kGB*(kGR+1))] g / output transpose, then [kGR][gin] weights
  float* s_g = s_gv;
  float* s_w = s_gv + gemv_g_floats(a.gin);
  const int tid = threadIdx.x;
  const int r = tid >> 5, bl = tid & 31;
  const int row0 = blockIdx.x * kGR;
  const int row = row0 + r;
  const int wq = (kGR * a.gin) >> 2;                    // float4 pieces of the weight block (gin % 4 == 0)
  const int wlim = (a.rows - row0 < kGR ? a.rows - row0 : kGR) * (a.gin >> 2);
  constexpr int kWU = 4, kGU = 16;
  for (int b0 = 0; b0 < a.batch; b0 += kGB) {
    __syncthreads();
    float4 wv[kWU];
    if (b0 == 0) {
      // (gin <= 512, checked by the launcher: kWU * kGT float4 cover the 16 rows)
#pragma unroll
      for (int u = 0; u < kWU; ++u) {
        const int i = tid + u * kGT;
        wv[u] = i < wlim ? reinterpret_cast<const float4*>(a.w + (size_t)row0 * a.gin)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const int items = a.gin * kGB;
    for (int base = tid; base < items; base += kGT * kGU) {
      float v[kGU];
#pragma unroll
      for (int u = 0; u < kGU; ++u) {                   // coalesced along k
        const int idx = base + u * kGT;
        const int bb = idx / a.gin, k = idx - bb * a.gin;
        v[u] = (idx < items && b0 + bb < a.batch) ? a.g[(size_t)(b0 + bb) * a.gin + k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kGU; ++u) {
        const int idx = base + u * kGT;
        const int bb = idx / a.gin, k = idx - bb * a.gin;
        if (idx < items) s_g[k * kGS + bb] = v[u];      // [k][kGS] keeps the transposing store conflict-free
      }
    }
    if (b0 == 0) {
#pragma unroll
      for (int u = 0; u < kWU; ++u) {
        const int i = tid + u * kGT;
        if (i < wq) reinterpret_cast<float4*>(s_w)[i] = wv[u];
      }
    }
    __syncthreads();
    float res = 0.f;
    if (row < a.rows && b0 + bl < a.batch) {
      const float* wr = s_w + r * a.gin;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int k = 0;
#pragma unroll 8
      for (; k + 4 <= a.gin; k += 4) {
        const float4 w4 = *reinterpret_cast<const float4*>(wr + k);
        s0 = fmaf(w4.x, s_g[(k + 0) * kGS + bl], s0); s1 = fmaf(w4.y, s_g[(k + 1) * kGS + bl], s1);
        s2 = fmaf(w4.z, s_g[(k + 2) * kGS + bl], s2); s3 = fmaf(w4.w, s_g[(k + 3) * kGS + bl], s3);
      }
      res = (s0 + s1) + (s2 + s3) + a.bias[row];
    }
    // the table is [utterance][row]: through LDS, so that 16 neighbouring lanes store the 16 rows of one utterance
    // (64 contiguous bytes) instead of every lane a 4-byte piece of a different 26 KB-apart line
    __syncthreads();                                    // everybody is done reading s_g
    s_g[bl * (kGR + 1) + r] = res;
    __syncthreads();
    {
      const int ob = tid >> 4, orow = tid & 15;         // kGT = 32 utterances x 16 rows
      if (row0 + orow < a.rows && b0 + ob < a.batch) a.out[(size_t)(b0 + ob) * a.rows + row0 + orow] = s_g[ob * (kGR + 1) + orow];
    }
  }
}

int launch_gemv(const GemvArgs& a, void* stream) {
  if (a.rows <= 0) return QVC_OK;
  const size_t lds = ((size_t)gemv_g_floats(a.gin) + (size_t)kGR * a.gin) * 4;
  if (a.gin % 4 || a.gin > 512 || lds > 160 * 1024) return QVC_ERR_BAD_CONFIG;
  static std::atomic<uint32_t> lds_ok{0};                  // > 64 KiB dynamic LDS: opt in once per device
  if (!allow_big_lds(lds_ok, reinterpret_cast<const void*>(cond_gemv_kernel))) return QVC_ERR_LAUNCH;
  hipLaunchKernelGGL(cond_gemv_kernel, dim3((unsigned)ceil_div(a.rows, kGR)), dim3(kGT), lds,
                     static_cast<hipStream_t>(stream), a);
  return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
}

// ------------------------------------------------------------------ sampling
// z[b][t][c] = mu + noise[b][c][t] * exp(logs): the noise arrives in the reference's (B, C, T) layout, everything
// else is frame-major, so a 32 x 32 tile goes through LDS -- both the read (along t) and the write (along c) are
// coalesced (reading the noise with c fastest cost a 64-byte sector per 4 bytes).
__global__ __launch_bounds__(256) void sample_kernel(const SampleArgs a) {
  __shared__ float s_n[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32, b = blockIdx.z;
#pragma unroll
  for (int cc = ty; cc < 32; cc += 8) {
    const int c = c0 + cc, t = t0 + tx;
    s_n[cc][tx] = (c < a.C && t < a.frames) ? a.noise[((size_t)b * a.C + c) * a.frames + t] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int tt = ty; tt < 32; tt += 8) {
    const int t = t0 + tt, c = c0 + tx;
    if (t < a.frames && c < a.C) {
      const size_t bt = (size_t)b * a.frames + t;
      const float mu = a.stats[bt * 2 * a.C + c], logs = a.stats[bt * 2 * a.C + a.C + c];
      a.z[bt * a.C + c] = mu + s_n[tx][tt] * expf(logs);
    }
  }
}

int launch_sample(const SampleArgs& a, void* stream) {
  if (a.batch <= 0 || a.frames <= 0 || a.C <= 0) return QVC_ERR_BAD_ARG;
  hipLaunchKernelGGL(sample_kernel, dim3((unsigned)ceil_div(a.frames, 32), (unsigned)ceil_div(a.C, 32), (unsigned)a.batch), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
}

// ------------------------------------------------------------------ iSTFT + band synthesis tail
// (formulas and the per-item math: qvc_tail_impl.h)
// One block = kOT consecutive output samples = kOT/4 band samples (+-7/8 halo) = kOT/16 frames (+-3/4).
constexpr int kOT = 912;                   // output samples per block: 57 + 7 = 64 frames x 4 bands = exactly one
                                           // (frame, band) item per thread in the DFT phase (1024 needed two rounds)
constexpr int kNY = kOT / 4 + 15;          // band samples needed: [a0-7, a0+kOT/4+7]
constexpr int kNFR = kOT / 16 + 7;         // frames needed: [f0-3, f0+kOT/16+3]

__global__ __launch_bounds__(256) void istft_synth_kernel(const TailArgs a) {
  __shared__ __attribute__((aligned(16))) float s_post[kNFR * kPostC];    // 20.4 KB
  __shared__ float s_xw[kBands][kNFR][17];                                // windowed frames (padded)
  __shared__ float s_y[kBands][kNY + 1];

  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int o0 = blockIdx.x * kOT;
  const int a0 = o0 >> 2;                   // first band sample owned by this block
  const int f_lo = (o0 >> 4) - 3;           // first frame staged
  const int Lpad = 4 * (a.F - 1);           // band signal length of the (padded) buffers
  const int Fb = ragged_len(a.rg, b, a.F);  // this utterance's frames are [Flo, Fb) of the buffer (ragged batches,
  const int Flo = ragged_lo(a.rg, b);       // streaming windows): its iSTFT envelope starts / ends there
  const int L = Fb > 0 ? 4 * (Fb - 1) : 0;  // its band signal is [4*Flo, L); samples outside are zeros
  const float* pb = a.post + (size_t)b * a.F * kPostC;

  // ---- stage frames [f_lo, f_lo+kNFR) x 72 channels (contiguous in memory), float4 coalesced
  {   // all of a thread's loads go out before its first LDS store (one per loop trip = serialised round trips)
    constexpr int kChunks = kNFR * (kPostC / 4), kPer = (kChunks + 255) / 256;
    float4 v[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int i = tid + u * 256;
      const int fr = i / (kPostC / 4), c4 = i - fr * (kPostC / 4);
      const int t = f_lo + fr;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < kChunks && t >= Flo && t < Fb) v[u] = *reinterpret_cast<const float4*>(pb + (size_t)t * kPostC + c4 * 4);
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int i = tid + u * 256;
      if (i < kChunks) *reinterpret_cast<float4*>(&s_post[i * 4]) = v[u];
    }
  }
  __syncthreads();

  // ---- per (frame, band): polar -> 16-point inverse real DFT -> Hann window
  for (int item = tid; item < kNFR * kBands; item += 256) {
    const int fr = item >> 2, k = item & 3;
    float xw[16];
    tail_dft(&s_post[fr * kPostC + k * 2 * kBins], xw);
#pragma unroll
    for (int m = 0; m < 16; ++m) s_xw[k][fr][m] = xw[m];
  }
  __syncthreads();

  // ---- overlap-add + envelope: band samples n = a0 - 7 + i
  for (int item = tid; item < kNY * kBands; item += 256) {
    const int k = item / kNY, i = item - k * kNY;
    const int n = a0 - 7 + i;
    const float y = tail_ola(n, Flo, Fb, L, [&](int t, int m) { return s_xw[k][t - f_lo][m]; });
    s_y[k][i] = y;
    if (a.y_mb && i >= 7 && i < 7 + kOT / 4 && n < Lpad)
      a.y_mb[((size_t)b * kBands + k) * Lpad + n] = y;
  }
  __syncthreads();

  // ---- polyphase synthesis FIR: thread -> 4 consecutive outputs o = o0 + 4*tid + r
  if (tid < kOT / 4) {
    const int ia = tid + 7;                                  // s_y index of band sample a = a0 + tid
    float out[4];
    tail_fir<kTaps>([&](int k, int d) { return s_y[k][ia + d]; }, a.fir, out);
    const int o = o0 + 4 * tid;
    const int n_out = 4 * Lpad;
    if (o >= 4 * L || o < 16 * Flo) out[0] = out[1] = out[2] = out[3] = 0.f;   // outside this utterance (both bounds are multiples of 4)
    if (o + 3 < n_out) {
      *reinterpret_cast<float4*>(a.out + (size_t)b * n_out + o) = make_float4(out[0], out[1], out[2], out[3]);
    } else {
      for (int r = 0; r < 4; ++r) if (o + r < n_out) a.out[(size_t)b * n_out + o + r] = out[r];
    }
  }
}

int launch_tail(const TailArgs& a, void* stream) {
  const int n_out = 16 * (a.F - 1);
  if (n_out <= 0) return QVC_ERR_BAD_ARG;
  hipLaunchKernelGGL(istft_synth_kernel, dim3((unsigned)ceil_div(n_out, kOT), (unsigned)a.batch), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
}

// ------------------------------------------------------------------ frame-major -> channel-major
__global__ __launch_bounds__(256) void fm_to_cm_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                       int batch, int frames, int channels) {
  const size_t n = (size_t)batch * frames * channels;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i % frames);
    const size_t bc = i / frames;
    const int c = (int)(bc % channels);
    const size_t b = bc / channels;
    dst[i] = src[(b * frames + t) * channels + c];
  }
}

int launch_fm_to_cm(const float* src, float* dst, int batch, int frames, int channels, void* stream) {
  const size_t n = (size_t)batch * frames * channels;
  const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(fm_to_cm_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, dst, batch,
                     frames, channels);
  return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
}

// ------------------------------------------------------------------ batched strided copies (qvc_stream.h)
struct CopyBatchArgs {
  CopyDesc d[kCopyBatchMax];
  uint32_t first[kCopyBatchMax + 1];   // first workgroup of descriptor i; first[n] = grid size
  uint32_t vec[kCopyBatchMax];         // bytes per element: 16 when every address / pitch / width allows it, else 4
  int32_t n;
};
constexpr int kCopyPerThread = 4;      // elements per thread: 16 KiB per workgroup at 16 bytes per element

template <typename V>
__device__ inline void copy_span(const CopyDesc& d, uint32_t wg) {
  const uint32_t wv = d.width / (uint32_t)sizeof(V);
  const uint32_t total = wv * d.rows;
  const uint32_t e0 = wg * (256u * kCopyPerThread) + threadIdx.x;
  V v[kCopyPerThread];
  uint32_t off[kCopyPerThread];
#pragma unroll
  for (int k = 0; k < kCopyPerThread; ++k) {
    const uint32_t e = e0 + (uint32_t)k * 256u;
    const uint32_t row = e / wv, col = e - row * wv;
    off[k] = e < total ? row * d.dpitch + col * (uint32_t)sizeof(V) : 0xffffffffu;
    v[k] = V{};
    if (e < total && d.src) v[k] = *reinterpret_cast<const V*>(static_cast<const char*>(d.src) + (size_t)row * d.spitch + (size_t)col * sizeof(V));
  }
#pragma unroll
  for (int k = 0; k < kCopyPerThread; ++k)
    if (off[k] != 0xffffffffu) *reinterpret_cast<V*>(static_cast<char*>(d.dst) + off[k]) = v[k];
}

__global__ __launch_bounds__(256) void copy_batch_kernel(const CopyBatchArgs a) {
  int i = 0;
  while (i + 1 < a.n && blockIdx.x >= a.first[i + 1]) ++i;    // <= 12 descriptors: a scalar scan
  if (a.vec[i] == 16) copy_span<uint4>(a.d[i], blockIdx.x - a.first[i]);
  else copy_span<uint32_t>(a.d[i], blockIdx.x - a.first[i]);
}

int launch_copy_batch(const CopyDesc* d, int n, void* stream) {
  if (n <= 0) return QVC_OK;
  if (!d || n > kCopyBatchMax) return QVC_ERR_BAD_ARG;
  CopyBatchArgs a{};
  a.n = 0;
  uint32_t wgs = 0;
  for (int i = 0; i < n; ++i) {
    const CopyDesc& c = d[i];
    if (!c.dst || (c.width | c.dpitch | c.spitch) & 3u || (reinterpret_cast<uintptr_t>(c.dst) | reinterpret_cast<uintptr_t>(c.src)) & 3u)
      return QVC_ERR_BAD_ARG;
    if (c.width == 0 || c.rows == 0) continue;
    const bool wide = !(((c.width | c.dpitch | c.spitch) & 15u) || ((reinterpret_cast<uintptr_t>(c.dst) | reinterpret_cast<uintptr_t>(c.src)) & 15u));
    const uint32_t v = wide ? 16u : 4u;
    const uint64_t total = (uint64_t)(c.width / v) * c.rows;
    // 32-bit element and destination-offset arithmetic in the kernel
    if (total >= (1ull << 31) || (uint64_t)c.rows * c.dpitch >= (1ull << 32)) return QVC_ERR_BAD_ARG;
    a.d[a.n] = c; a.vec[a.n] = v; a.first[a.n] = wgs;
    wgs += (uint32_t)((total + 256u * kCopyPerThread - 1) / (256u * kCopyPerThread));
    ++a.n;
  }
  if (a.n == 0) return QVC_OK;
  a.first[a.n] = wgs;
  hipLaunchKernelGGL(copy_batch_kernel, dim3(wgs), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
}

// ------------------------------------------------------------------ conv dispatcher
int launch_conv(const ConvDesc& d, ConvArgs a, int batch, int epi, int dtype, void* stream, int* nf_out) {
  a.Cin = d.Cin; a.CinP = d.CinP; a.taps = d.taps; a.dil = d.dil; a.left = d.left;
  a.KS = d.KS(); a.nIt = d.nIt(); a.nchunk = d.nchunk; a.M = d.M;
  a.up_s = d.up_s; a.up_p = d.up_p; a.Cout = d.Cout; a.lp = d.lp; a.ksize = d.ksize;
  if (d.lp && (epi != EPI_STD || !a.y16 || a.y32 || a.y32b || a.res || a.res16 || d.MF % 2)) return QVC_ERR_BAD_CONFIG;
  if (dtype == QVC_F16) return launch_conv_typed<_Float16>(d, a, batch, epi, stream, nf_out);
  if (dtype == QVC_BF16) return launch_conv_typed<__bf16>(d, a, batch, epi, stream, nf_out);
  return QVC_ERR_BAD_ARG;
}

int launch_post_tail(const ConvDesc& d, PostTailArgs a, int batch, int dtype, void* stream) {
  a.c.Cin = d.Cin; a.c.CinP = d.CinP; a.c.taps = d.taps; a.c.dil = d.dil; a.c.left = d.left;
  a.c.KS = d.KS(); a.c.nIt = d.nIt(); a.c.nchunk = d.nchunk; a.c.M = d.M;
  a.c.up_s = d.up_s; a.c.up_p = d.up_p; a.c.Cout = d.Cout;
  if (dtype == QVC_F16) return launch_post_tail_typed<_Float16>(d, a, batch, stream);
  if (dtype == QVC_BF16) return launch_post_tail_typed<__bf16>(d, a, batch, stream);
  return QVC_ERR_BAD_ARG;
}

bool wn_stack_supported(const ConvDesc& din, int layers) { return wn_stack_ok(din, layers); }

// qvc_wn2.hip: the continuous-stream stack kernel for the shipped shape (debug switch "wn_kernel" = 1: always the generic one)
template <typename T> int launch_wn_stack2_typed(const ConvDesc& din, const WnStackArgs& a, int batch, void* stream);
bool wn_stack2_supported(const ConvDesc& din, const WnStackArgs& a);
int wn_stack_variant(const ConvDesc& din, const WnStackArgs& a) { return debug_get(DBG_WN_KERNEL) == 0 && wn_stack2_supported(din, a) ? 2 : 1; }

int launch_wn_stack(const ConvDesc& din, const WnStackArgs& a, int batch, int dtype, void* stream) {
  if (wn_stack_variant(din, a) == 2) {
    if (dtype == QVC_F16) return launch_wn_stack2_typed<_Float16>(din, a, batch, stream);
    if (dtype == QVC_BF16) return launch_wn_stack2_typed<__bf16>(din, a, batch, stream);
    return QVC_ERR_BAD_ARG;
  }
  if (dtype == QVC_F16) return launch_wn_stack_typed<_Float16>(din, a, batch, stream);
  if (dtype == QVC_BF16) return launch_wn_stack_typed<__bf16>(din, a, batch, stream);
  return QVC_ERR_BAD_ARG;
}

int launch_wn(const ConvDesc& din, WnArgs a, int batch, int dtype, void* stream, int* nf_out) {
  if (dtype == QVC_F16) return launch_wn_typed<_Float16>(din, a, batch, stream, nf_out);
  if (dtype == QVC_BF16) return launch_wn_typed<__bf16>(din, a, batch, stream, nf_out);
  return QVC_ERR_BAD_ARG;
}

int launch_pair3(const ConvDesc* d1, const ConvDesc* d2, const PairArgs3& a, int batch, int dtype, void* stream, int* nf_out) {
  if (a.n < 1 || a.n > 3) return QVC_ERR_BAD_ARG;
  for (int i = 0; i < a.n; ++i) {
    if (!pair_supported(d1[i], d2[i]) || !d1[i].lp || !d2[i].lp) return QVC_ERR_BAD_CONFIG;
    if (d1[i].MF != d1[0].MF || d1[i].WM != d1[0].WM || d1[i].CinP != d1[0].CinP) return QVC_ERR_BAD_CONFIG;
    if (a.p[i].T != a.p[0].T || a.p[i].C != a.p[0].C || a.p[i].CP != a.p[0].CP) return QVC_ERR_BAD_ARG;
  }
  if (dtype == QVC_F16) return launch_pair_typed<_Float16, _Float16>(d1, a, batch, stream, nf_out);
  if (dtype == QVC_BF16) return launch_pair_typed<__bf16, __bf16>(d1, a, batch, stream, nf_out);
  if (dtype == QVC_BF16X) return launch_pair_typed<__bf16, _Float16>(d1, a, batch, stream, nf_out);   // bf16 operands, f16 stream
  return QVC_ERR_BAD_ARG;
}

int launch_chain(const ConvDesc* d1, const ConvDesc* d2, const ChainArgs& a, int batch, int dtype, void* stream, int* nf_out) {
  if (!chain_supported(d1, d2, a.n)) return QVC_ERR_BAD_CONFIG;
  for (int q = 1; q < a.n; ++q)
    if (a.p[q].T != a.p[0].T || a.p[q].C != a.p[0].C || a.p[q].CP != a.p[0].CP || a.p[q].bs != a.p[0].bs) return QVC_ERR_BAD_ARG;
  if (dtype == QVC_F16) return launch_chain_typed<_Float16, _Float16>(d1, d2, a, batch, stream, nf_out);
  if (dtype == QVC_BF16) return launch_chain_typed<__bf16, __bf16>(d1, d2, a, batch, stream, nf_out);
  if (dtype == QVC_BF16X) return launch_chain_typed<__bf16, _Float16>(d1, d2, a, batch, stream, nf_out);
  return QVC_ERR_BAD_ARG;
}

int launch_pair(const ConvDesc& d1, const ConvDesc& d2, PairArgs a, int batch, int dtype, void* stream, int* nf_out) {
  PairArgs3 a3; a3.p[0] = a; a3.n = 1;
  return launch_pair3(&d1, &d2, a3, batch, dtype, stream, nf_out);
}

}  // namespace qvc
