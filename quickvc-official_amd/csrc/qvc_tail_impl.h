// qvc_tail_impl.h -- device-side pieces of the iSTFT + band-synthesis tail (models.py:394-406, pqmf.py:106-117),
// shared by istft_synth_kernel (qvc_small.hip) and the fused conv_post + tail kernel (qvc_post_tail_impl.h) so that
// the two produce the same bits.
// Geometry (n_fft 16, hop 4, 4 sub-bands, 63 taps):
//   band signal   y_k[n], n in [0, 4(F-1)):  y = (sum_t w[m] x_t[m]) / (sum_t w[m]^2), m = n + 8 - 4t
//   output        out[o], o in [0, 16(F-1)): out[o] = sum_k sum_n fir[k][4n - o + 31] * y_k[n]
#pragma once
#include <hip/hip_runtime.h>

namespace qvc {

constexpr int kBands = 4, kBins = 9, kPostC = kBands * 2 * kBins, kTaps = 63;

// cos / sin(2*pi*j/16), j = 0..15
#define QVC_C16 {1.f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.f, -0.38268343236508977f,   \
                 -0.70710678118654752f, -0.92387953251128674f, -1.f, -0.92387953251128674f, -0.70710678118654752f,     \
                 -0.38268343236508977f, 0.f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f}
#define QVC_S16 {0.f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f, 1.f, 0.92387953251128674f,     \
                 0.70710678118654752f, 0.38268343236508977f, 0.f, -0.38268343236508977f, -0.70710678118654752f,        \
                 -0.92387953251128674f, -1.f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f}

// One (frame, band): 9 log-magnitudes + 9 phase arguments -> polar -> 16-point inverse real DFT -> periodic Hann.
// `sp` = the band's 18 post-conv values; xw[m] = windowed sample m of the frame.
__device__ __forceinline__ void tail_dft(const float* sp, float (&xw)[16]) {
  float re[kBins], im[kBins];
#pragma unroll
  for (int q = 0; q < kBins; ++q) {
    // hardware transcendentals: v_exp_f32 (1 ulp), v_sin/v_cos_f32 take revolutions.  The phase
    // pi*sin(p) lies in [-pi, pi] = [-0.5, 0.5] revolutions, the sweet spot of v_sin/v_cos (abs err
    // ~1e-6); p itself is reduced with an exact fract() first.
    const float mag = __builtin_amdgcn_exp2f(1.4426950408889634f * sp[q]);
    const float pr = sp[kBins + q] * 0.15915494309189535f;              // p / 2pi
    const float sp_ = __builtin_amdgcn_sinf(pr - floorf(pr));            // sin(p)
    const float rev = 0.5f * sp_;                                        // pi*sin(p) / 2pi
    re[q] = mag * __builtin_amdgcn_cosf(rev); im[q] = mag * __builtin_amdgcn_sinf(rev);
  }
  constexpr float C16[16] = QVC_C16;      // twiddles: compile-time constants after unrolling
  constexpr float S16[16] = QVC_S16;
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    float acc = re[0] + ((m & 1) ? -re[8] : re[8]);      // imaginary parts of bins 0 and 8 are ignored
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      const int j = (q * m) & 15;
      acc += 2.f * (re[q] * C16[j] - im[q] * S16[j]);
    }
    const float win = 0.5f - 0.5f * C16[m];               // periodic Hann(16)
    xw[m] = acc * (1.f / 16.f) * win;
  }
}

// Overlap-add + envelope for band sample n of an utterance whose frames are [Flo, Fb) (band signal [4*Flo, L)):
// `xw(t, m)` returns windowed sample m of frame t.
template <class XW>
__device__ __forceinline__ float tail_ola(int n, int Flo, int Fb, int L, XW xw) {
  float y = 0.f;
  if (n >= 4 * Flo && n < L) {
    constexpr float C16b[16] = QVC_C16;
    float num = 0.f, env = 0.f;
    const int t_hi = (n + 8) >> 2;                         // frames with m = n + 8 - 4t in [0, 16)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int t = t_hi - d, m = n + 8 - 4 * t;
      if (t >= Flo && t < Fb && m < 16) {
        const float w = 0.5f - 0.5f * C16b[m];
        num += xw(t, m);
        env += w * w;
      }
    }
    y = num / env;
  }
  return y;
}

// Polyphase synthesis FIR: the 4 consecutive outputs o = 4a + r of band sample index a; `ys(k, d)` returns band
// sample a + d of band k, d in [-7, 8]; s_fir = [band][FS] taps.  With `s_fir` pointing at the taps in MEMORY
// (FS = 63) every tap address is wave-uniform and known at compile time, so the 252 taps arrive through scalar
// loads into SGPRs -- no LDS read per FMA (staged in LDS they were 4/5 of this phase's LDS instructions).
template <int FS, class YS>
__device__ __forceinline__ void tail_fir(YS ys, const float* __restrict__ s_fir, float (&out)[4]) {
  out[0] = out[1] = out[2] = out[3] = 0.f;
  // constant address space: tells the compiler the taps are read-only for the whole launch, which is what lets a
  // wave-uniform address become an s_load (through the generic pointer it emitted per-lane global loads)
  typedef const float __attribute__((address_space(4))) ctap;
  const ctap* taps = (const ctap*)(uintptr_t)s_fir;
#pragma unroll
  for (int k = 0; k < kBands; ++k) {
#pragma unroll
    for (int d = -7; d <= 8; ++d) {
      const float yv = ys(k, d);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 4 * d - r + 31;                      // tap index, compile-time after unrolling
        if (j >= 0 && j < kTaps) out[r] = fmaf(taps[k * FS + j], yv, out[r]);
      }
    }
  }
}

}  // namespace qvc
