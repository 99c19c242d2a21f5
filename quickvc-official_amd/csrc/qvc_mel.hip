// qvc_mel.hip -- mel_processing.wave_to_mel (mel_processing.py:15-98) on gfx950, all fp32.
//
//   wave (U, N) --reflect pad--> Hann STFT n_fft/hop (center=False) --> sqrt(re^2+im^2+1e-6) --> mel basis --> log(clamp 1e-5)
//
// The STFT is an implicit GEMM on the f32 MFMA (v_mfma_f32_16x16x4_f32: bitwise an fmaf chain, 155 TFLOP/s):
//   spec[2*bins x frames] = W[2*bins x n_fft] * X[n_fft x frames],  X[k][f] = padded_wave[f*hop + k],
// W = Hann-windowed cos / -sin rows.  Frames overlap (hop < n_fft), so a workgroup stages ONE contiguous piece of
// the padded waveform into LDS (reflect padding applied on the way in) and a frame is just an offset into it,
// exactly like the taps of the conv kernel.  A wave owns the cos and the sin rows of the same 2 x 16 bins, so the
// magnitude is lane-local in the epilogue; no complex tensor, no FFT plan, nothing but the spectrogram is written.
// The mel stage exploits the filters' sparsity (each triangle spans a few dozen bins): one thread per
// (utterance, frame, filter) walks its own bin range.
#include <hip/hip_runtime.h>
#include "qvc_launch_util.h"
#include <cmath>
#include <cstring>
#include <vector>
#include "qvc_plan.h"

namespace qvc {

typedef float mf32x4 __attribute__((ext_vector_type(4)));

constexpr int kMelNF = 2;                    // 16-frame column fragments per wave
constexpr int kMelFrames = kMelNF * 16;      // frames per workgroup
constexpr int kMelBinsPerBlock = 128;        // 4 waves x 2 x 16 bins
constexpr int kMelPadEvery = 4;              // floats of LDS padding per hop (frame starts land in different banks)

struct MelTable {       // byte offsets into the packed table
  int32_t n_fft = 0, hop = 0, n_mels = 0, bins = 0, nchunk = 0, ksteps = 0;
  int64_t dft_off = 0;  // fp32 [chunk][wave 4][k16-step][frag 4 = (bin half, cos|sin)][lane 64][4]
  int64_t basis_off = 0;   // fp32 [n_mels][bins]
  int64_t range_off = 0;   // int32 [n_mels][2]: first bin, one past the last bin with a non-zero weight
  int64_t bytes = 0;
};

inline int mel_validate(int n_fft, int hop, int n_mels) {
  if (n_fft < 16 || n_fft % 16 || hop < 4 || hop % 4 || hop > n_fft || (n_fft - hop) % 2 || n_mels < 1 || n_mels > 4096 || n_fft > 16384)
    return QVC_ERR_BAD_CONFIG;
  return QVC_OK;
}

inline MelTable mel_table(int n_fft, int hop, int n_mels) {
  MelTable t;
  t.n_fft = n_fft; t.hop = hop; t.n_mels = n_mels; t.bins = n_fft / 2 + 1;
  t.nchunk = ceil_div(t.bins, kMelBinsPerBlock); t.ksteps = n_fft / 16;
  int64_t off = 0;
  t.dft_off = off; off = align_up(off + (int64_t)t.nchunk * 4 * t.ksteps * 4 * 64 * 4 * 4, 256);
  t.basis_off = off; off = align_up(off + (int64_t)n_mels * t.bins * 4, 256);
  t.range_off = off; off = align_up(off + (int64_t)n_mels * 2 * 4, 256);
  t.bytes = off;
  return t;
}

inline int mel_frames(int n_fft, int hop, int samples) {
  const int pad = (n_fft - hop) / 2;
  const int total = samples + 2 * pad;
  return total < n_fft ? 0 : (total - n_fft) / hop + 1;
}

struct StftArgs {
  const float* wave; int32_t samples, frames, n_fft, hop, pad, bins, binsP, ksteps;
  const float* dft; float* spec;     // spec: [U][frames][binsP]
};

// LDS index of padded-waveform sample i of the staged piece (4 floats of padding per hop)
__device__ __forceinline__ int mel_lds_idx(int i, int hop) { return i + (i / hop) * kMelPadEvery; }

__global__ __launch_bounds__(256) void stft_mag_kernel(const StftArgs a) {
  extern __shared__ __align__(16) float s_x[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int u = blockIdx.y, chunk = blockIdx.z;
  const int f0 = blockIdx.x * kMelFrames;
  // stage padded samples [f0*hop, f0*hop + (kMelFrames-1)*hop + n_fft), reflect padding as F.pad(mode='reflect')
  const int piece = (kMelFrames - 1) * a.hop + a.n_fft;
  const float* wv = a.wave + (size_t)u * a.samples;
  constexpr int kStageU = 8;                          // loads in flight per thread (one per loop trip = serialised round trips)
  for (int base = tid; base < piece; base += 256 * kStageU) {
    float v[kStageU];
#pragma unroll
    for (int j = 0; j < kStageU; ++j) {
      const int i = base + j * 256;
      int s = f0 * a.hop + i - a.pad;                 // index into the unpadded waveform
      if (s < 0) s = -s;
      if (s >= a.samples) s = 2 * (a.samples - 1) - s;
      v[j] = 0.f;
      if (i < piece && s >= 0 && s < a.samples) v[j] = wv[s];   // frames past the end of the utterance read zeros
    }
#pragma unroll
    for (int j = 0; j < kStageU; ++j) {
      const int i = base + j * 256;
      if (i < piece) s_x[mel_lds_idx(i, a.hop)] = v[j];
    }
  }
  __syncthreads();

  mf32x4 acc[4][kMelNF];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < kMelNF; ++n) acc[m][n] = mf32x4{0.f, 0.f, 0.f, 0.f};
  const mf32x4* ap = reinterpret_cast<const mf32x4*>(a.dft) + ((size_t)(chunk * 4 + w) * a.ksteps * 4) * 64 + lane;
  // B: lane (kq, col) holds x[frame col][16 s + 4 kq + 0..3] -- 4 consecutive samples, one ds_read_b128
  int boff[kMelNF];
#pragma unroll
  for (int n = 0; n < kMelNF; ++n) boff[n] = (n * 16 + col) * (a.hop + kMelPadEvery) + kq * 4;
  constexpr int PF = 2;                                // k16-steps of A in flight
  mf32x4 ar[PF + 1][4];
#pragma unroll
  for (int p = 0; p < PF; ++p)
#pragma unroll
    for (int m = 0; m < 4; ++m) ar[p][m] = ap[((size_t)p * 4 + m) * 64];
  for (int s0 = 0; s0 < a.ksteps; s0 += PF + 1) {
#pragma unroll
    for (int p = 0; p < PF + 1; ++p) {
      const int s = s0 + p;
      if (s < a.ksteps) {
        if (s + PF < a.ksteps) {
#pragma unroll
          for (int m = 0; m < 4; ++m) ar[(p + PF) % (PF + 1)][m] = ap[((size_t)(s + PF) * 4 + m) * 64];
        }
        const int k0 = s * 16;                         // a 16-sample step never straddles a hop boundary (hop % 16 == 0 or handled below)
        mf32x4 bv[kMelNF];
#pragma unroll
        for (int n = 0; n < kMelNF; ++n) {
          const int k = k0 + kq * 4;
          bv[n] = *reinterpret_cast<const mf32x4*>(s_x + boff[n] - kq * 4 + k + (k / a.hop) * kMelPadEvery);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < kMelNF; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m)
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[p][m][j], bv[n][j], acc[m][n], 0, 0, 0);
      }
    }
  }
  // fragments: m = 2*half + part (part 0: cos rows -> re, 1: -sin rows -> im) of bins chunk*128 + w*32 + half*16 + ...
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int bin0 = chunk * kMelBinsPerBlock + w * 32 + half * 16 + kq * 4;
    if (bin0 >= a.binsP) continue;
#pragma unroll
    for (int n = 0; n < kMelNF; ++n) {
      const int f = f0 + n * 16 + col;
      if (f >= a.frames) continue;
      const mf32x4 re = acc[2 * half][n], im = acc[2 * half + 1][n];
      float4 o;
      o.x = sqrtf(re[0] * re[0] + im[0] * im[0] + 1e-6f); o.y = sqrtf(re[1] * re[1] + im[1] * im[1] + 1e-6f);
      o.z = sqrtf(re[2] * re[2] + im[2] * im[2] + 1e-6f); o.w = sqrtf(re[3] * re[3] + im[3] * im[3] + 1e-6f);
      *reinterpret_cast<float4*>(a.spec + ((size_t)u * a.frames + f) * a.binsP + bin0) = o;
    }
  }
}

struct MelArgs {
  const float* spec; const float* basis; const int32_t* range; float* mel;
  int32_t utterances, frames, bins, binsP, n_mels;
};

// mel[u][m][f] = log(max(sum_b basis[m][b] * spec[u][f][b], 1e-5))   (mel_processing.py:70-73, :8-9)
__global__ __launch_bounds__(256) void mel_log_kernel(const MelArgs a) {
  const size_t total = (size_t)a.utterances * a.n_mels * a.frames;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int f = (int)(i % a.frames);
    const int m = (int)((i / a.frames) % a.n_mels);
    const int u = (int)(i / ((size_t)a.frames * a.n_mels));
    const int lo = a.range[2 * m], hi = a.range[2 * m + 1];
    const float* sp = a.spec + ((size_t)u * a.frames + f) * a.binsP;
    const float* bs = a.basis + (size_t)m * a.bins;
    float s = 0.f;
#pragma unroll 8
    for (int b = lo; b < hi; ++b) s = fmaf(bs[b], sp[b], s);
    a.mel[i] = logf(fmaxf(s, 1e-5f));
  }
}

}  // namespace qvc

using namespace qvc;

extern "C" int64_t qvc_mel_table_bytes(int32_t n_fft, int32_t n_mels) {
  if (mel_validate(n_fft, 4, n_mels) != QVC_OK) return QVC_ERR_BAD_CONFIG;
  return mel_table(n_fft, 4, n_mels).bytes;
}

extern "C" int qvc_mel_pack_tables(int32_t n_fft, int32_t hop, int32_t n_mels, const float* mel_basis_host,
                                   void* table_host, int64_t table_bytes) {
  if (!mel_basis_host || !table_host) return QVC_ERR_BAD_ARG;
  int st = mel_validate(n_fft, hop, n_mels);
  if (st != QVC_OK) return st;
  const MelTable t = mel_table(n_fft, hop, n_mels);
  if (table_bytes < t.bytes) return QVC_ERR_SMALL_BUFFER;
  std::memset(table_host, 0, (size_t)t.bytes);
  char* base = static_cast<char*>(table_host);
  // periodic Hann window (torch.hann_window default) times the DFT twiddles, in double, rounded once to fp32
  const double two_pi = 6.283185307179586476925286766559;
  std::vector<double> win((size_t)n_fft);
  for (int n = 0; n < n_fft; ++n) win[(size_t)n] = 0.5 - 0.5 * std::cos(two_pi * n / n_fft);
  float* dft = reinterpret_cast<float*>(base + t.dft_off);
  for (int c = 0; c < t.nchunk; ++c)
    for (int w = 0; w < 4; ++w)
      for (int s = 0; s < t.ksteps; ++s)
        for (int m = 0; m < 4; ++m)
          for (int lane = 0; lane < 64; ++lane) {
            const int bin = c * kMelBinsPerBlock + w * 32 + (m >> 1) * 16 + (lane & 15);
            for (int j = 0; j < 4; ++j) {
              const int k = s * 16 + (lane >> 4) * 4 + j;
              float v = 0.f;
              if (bin < t.bins) {
                const long long ph = ((long long)bin * k) % n_fft;      // exact phase reduction
                const double ang = two_pi * (double)ph / n_fft;
                v = (float)((m & 1) ? -win[(size_t)k] * std::sin(ang) : win[(size_t)k] * std::cos(ang));
              }
              *dft++ = v;
            }
          }
  std::memcpy(base + t.basis_off, mel_basis_host, (size_t)n_mels * t.bins * 4);
  int32_t* range = reinterpret_cast<int32_t*>(base + t.range_off);
  for (int m = 0; m < n_mels; ++m) {
    int lo = t.bins, hi = 0;
    for (int b = 0; b < t.bins; ++b)
      if (mel_basis_host[(size_t)m * t.bins + b] != 0.f) { if (b < lo) lo = b; hi = b + 1; }
    if (hi == 0) lo = 0;
    range[2 * m] = lo; range[2 * m + 1] = hi;
  }
  return QVC_OK;
}

extern "C" int64_t qvc_mel_workspace_bytes(int32_t n_fft, int32_t hop, int32_t utterances, int32_t samples) {
  if (utterances <= 0 || samples <= 0) return QVC_ERR_BAD_ARG;
  if (mel_validate(n_fft, hop, 1) != QVC_OK) return QVC_ERR_BAD_CONFIG;
  const int frames = mel_frames(n_fft, hop, samples);
  if (frames <= 0 || samples <= (n_fft - hop) / 2) return QVC_ERR_BAD_ARG;      // reflect padding needs pad < samples
  const int64_t binsP = align_up(n_fft / 2 + 1, 4);
  return align_up((int64_t)utterances * frames * binsP * 4, 256);
}

extern "C" int qvc_wave_to_mel(const void* table_dev, int32_t n_fft, int32_t hop, int32_t n_mels,
                               const float* wave, float* mel, int32_t utterances, int32_t samples,
                               void* workspace, int64_t workspace_bytes, void* stream) {
  if (!table_dev || !wave || !mel || !workspace) return QVC_ERR_BAD_ARG;
  int st = mel_validate(n_fft, hop, n_mels);
  if (st != QVC_OK) return st;
  const int64_t need = qvc_mel_workspace_bytes(n_fft, hop, utterances, samples);
  if (need < 0) return (int)need;
  if (workspace_bytes < need) return QVC_ERR_SMALL_BUFFER;
  if (hop % 16) return QVC_ERR_BAD_CONFIG;             // a 16-sample k-step must not straddle a hop boundary
  const MelTable t = mel_table(n_fft, hop, n_mels);
  const int frames = mel_frames(n_fft, hop, samples);
  const char* tb = static_cast<const char*>(table_dev);
  hipStream_t s = static_cast<hipStream_t>(stream);
  StftArgs sa;
  sa.wave = wave; sa.samples = samples; sa.frames = frames; sa.n_fft = n_fft; sa.hop = hop; sa.pad = (n_fft - hop) / 2;
  sa.bins = t.bins; sa.binsP = (int)align_up(t.bins, 4); sa.ksteps = t.ksteps;
  sa.dft = reinterpret_cast<const float*>(tb + t.dft_off); sa.spec = static_cast<float*>(workspace);
  const int piece = (kMelFrames - 1) * hop + n_fft;
  const size_t lds = (size_t)(piece + (piece / hop + 1) * kMelPadEvery) * 4;
  if (lds > 160 * 1024) return QVC_ERR_BAD_CONFIG;
  static std::atomic<uint32_t> lds_ok{0};                  // > 64 KiB dynamic LDS: opt in once per device
  if (!allow_big_lds(lds_ok, reinterpret_cast<const void*>(stft_mag_kernel))) return QVC_ERR_LAUNCH;
  hipLaunchKernelGGL(stft_mag_kernel, dim3((unsigned)ceil_div(frames, kMelFrames), (unsigned)utterances, (unsigned)t.nchunk), dim3(256), lds, s, sa);
  if (hipGetLastError() != hipSuccess) return QVC_ERR_LAUNCH;
  MelArgs ma;
  ma.spec = sa.spec; ma.basis = reinterpret_cast<const float*>(tb + t.basis_off);
  ma.range = reinterpret_cast<const int32_t*>(tb + t.range_off); ma.mel = mel;
  ma.utterances = utterances; ma.frames = frames; ma.bins = t.bins; ma.binsP = sa.binsP; ma.n_mels = n_mels;
  const size_t total = (size_t)utterances * n_mels * frames;
  hipLaunchKernelGGL(mel_log_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0, s, ma);
  return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
}
