// qvc_conv_impl.h -- the implicit-GEMM Conv1d kernel for gfx950 (MFMA 16x16x32, wave64).
//
// One kernel covers every conv of the hot path:
//   D[v][q] = sum_{tap, ci} Wv[v][tap][ci] * act(X[q + tap*dil - left][ci])
// GEMM view: M = output channels (virtual rows v), N = frames q, K = taps*Cin.
//
// Mapping to the hardware
//   * workgroup = 4 waves (256 threads) = one [4*MF*16 rows] x [NF*16 frames] output tile.
//   * the activation tile (NF*16 + halo frames, all Cin channels) is staged ONCE into LDS,
//     frame-major, converted to the MFMA operand type with the leaky-ReLU applied on the way
//     in; rows are XOR-swizzled per 16-byte chunk so the ds_read_b128 B-fragment reads of the
//     16 lanes of a group spread over the banks.  A tap is just a row offset into the tile,
//     so dilation costs nothing.
//   * the 4 waves split M: each wave streams its own pre-packed A fragments straight from
//     global memory (one coalesced 1 KiB global_load_dwordx4 per fragment, L2 resident, each
//     fragment reused for NF MFMAs) -- weights never touch LDS, no barrier in the K loop.
//   * fp32 accumulators; the epilogue fuses bias, per-utterance conditioning, residual add,
//     MRF averaging, leaky-ReLU + down-conversion, the WaveNet gate and the res/skip split.
#pragma once
#include <hip/hip_runtime.h>
#include "qvc_kernels.h"

namespace qvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Op;
template <> struct Op<_Float16> {
  using frag = f16x8; using quad = f16x4;
  static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ _Float16 cvt(float f) {
    return (_Float16)__builtin_amdgcn_fmed3f(f, -65504.f, 65504.f);   // saturate instead of inf
  }
};
template <> struct Op<__bf16> {
  using frag = bf16x8; using quad = bf16x4;
  static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ __bf16 cvt(float f) { return (__bf16)f; }
};

__device__ __forceinline__ float lrelu(float x, float slope) { return x > 0.f ? x : x * slope; }
__device__ __forceinline__ float fast_sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); exact limits at +-inf, ~1e-7 abs error elsewhere
  return 1.f - 2.f / (__expf(2.f * x) + 1.f);
}

// 16-byte-chunk swizzle of a tile row (see tools: bank simulation in DESIGN.md).
__device__ __forceinline__ int swz(int row, int mode) {
  return mode == 0 ? (row & 15) : (mode == 1 ? (row & 7) : ((row >> 1) & 3));
}

template <typename T, int MF, int NF, int EPI>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
  using O = Op<T>;
  using frag = typename O::frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, chunk = blockIdx.z;
  constexpr int NT = NF * 16;
  const int q0 = blockIdx.x * NT;
  const int R = NT + (a.taps - 1) * a.dil;
  const int rowbytes = a.CinP * 2;
  const int cpr = a.CinP >> 3;                                  // 16-byte chunks per row
  const int smode = (cpr & 15) == 0 ? 0 : ((cpr & 7) == 0 ? 1 : 2);
  const int t_base = q0 - a.left;                               // input frame of tile row 0

  // ------------------------------------------------------------------ stage the activation tile
  if (a.x_kind == XK_F32_FM) {
    const float* xb = static_cast<const float*>(a.x) + (size_t)b * a.x_bs + a.x_c0;
    const float slope = a.slope_in;
    for (int idx = tid; idx < R * cpr; idx += 256) {
      const int r = idx / cpr, c8 = idx - r * cpr;
      const int ti = t_base + r;
      bool ok; int src;
      if (a.reflect) { ok = ti >= 0 && ti <= a.T_in; src = ti == 0 ? 1 : ti - 1; }
      else { ok = ti >= 0 && ti < a.T_in; src = ti; }
      ok = ok && (c8 * 8 < a.Cin);
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (ok) {
        const float4* p = reinterpret_cast<const float4*>(xb + (size_t)src * a.x_ts + c8 * 8);
        v0 = p[0]; v1 = p[1];
      }
      frag h;
      h[0] = O::cvt(lrelu(v0.x, slope)); h[1] = O::cvt(lrelu(v0.y, slope));
      h[2] = O::cvt(lrelu(v0.z, slope)); h[3] = O::cvt(lrelu(v0.w, slope));
      h[4] = O::cvt(lrelu(v1.x, slope)); h[5] = O::cvt(lrelu(v1.y, slope));
      h[6] = O::cvt(lrelu(v1.z, slope)); h[7] = O::cvt(lrelu(v1.w, slope));
      *reinterpret_cast<frag*>(smem + r * rowbytes + ((c8 ^ swz(r, smode)) << 4)) = h;
    }
  } else if (a.x_kind == XK_OP_FM) {
    const T* xb = static_cast<const T*>(a.x) + (size_t)b * a.x_bs + a.x_c0;
    for (int idx = tid; idx < R * cpr; idx += 256) {
      const int r = idx / cpr, c8 = idx - r * cpr;
      const int ti = t_base + r;
      const bool ok = ti >= 0 && ti < a.T_in && (c8 * 8 < a.Cin);
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (ok) v = *reinterpret_cast<const uint4*>(xb + (size_t)ti * a.x_ts + c8 * 8);
      *reinterpret_cast<uint4*>(smem + r * rowbytes + ((c8 ^ swz(r, smode)) << 4)) = v;
    }
  } else {  // XK_F32_CM: (B, C, T) -- consecutive threads walk along T (coalesced), transposed into the tile
    const float* xb = static_cast<const float*>(a.x) + (size_t)b * a.x_bs;
    const float slope = a.slope_in;
    for (int idx = tid; idx < R * a.CinP; idx += 256) {
      const int c = idx / R, r = idx - c * R;
      const int ti = t_base + r;
      float v = 0.f;
      if (c < a.Cin && ti >= 0 && ti < a.T_in) v = lrelu(xb[(size_t)c * a.x_ts + ti], slope);
      *reinterpret_cast<T*>(smem + r * rowbytes + (((c >> 3) ^ swz(r, smode)) << 4) + (c & 7) * 2) = O::cvt(v);
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ K loop: A from global, B from LDS
  f32x4 acc[MF][NF];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const frag* ap = static_cast<const frag*>(a.w) + ((size_t)(chunk * kWaves + wave) * a.nIt * MF) * 64 + lane;
  frag a_nxt[MF];
#pragma unroll
  for (int m = 0; m < MF; ++m) a_nxt[m] = ap[m * 64];

  const int lrow = lane & 15, lq = lane >> 4;
  int tap = 0, ks = 0;
  for (int it = 0; it < a.nIt; ++it) {
    frag a_cur[MF];
#pragma unroll
    for (int m = 0; m < MF; ++m) a_cur[m] = a_nxt[m];
    if (it + 1 < a.nIt) {
#pragma unroll
      for (int m = 0; m < MF; ++m) a_nxt[m] = ap[((size_t)(it + 1) * MF + m) * 64];
    }
    const int row0 = tap * a.dil + lrow;
    const int ch = ks * 4 + lq;
    frag bf[NF];
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      const int row = row0 + n * 16;
      bf[n] = *reinterpret_cast<const frag*>(smem + row * rowbytes + ((ch ^ swz(row, smode)) << 4));
    }
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n) acc[m][n] = O::mfma(a_cur[m], bf[n], acc[m][n]);
    if (++ks == a.KS) { ks = 0; ++tap; }
  }

  // ------------------------------------------------------------------ epilogue
  if constexpr (EPI == EPI_GAU) {
    // rows: m = 0 tanh half, m = 1 sigmoid half of channels chunk*64 + wave*16 + ...
    static_assert(EPI != EPI_GAU || MF == 2, "GAU epilogue pairs two fragments per wave");
    const int H = a.gau_H;
    const int ch0 = chunk * 64 + wave * 16 + lq * 4;
    if (ch0 < H) {
      const float* bb = a.bbias + (size_t)b * a.bbias_bs;
      const float4 bt = *reinterpret_cast<const float4*>(bb + ch0);
      const float4 bs = *reinterpret_cast<const float4*>(bb + H + ch0);
      T* yb = static_cast<T*>(a.y16) + (size_t)b * a.y16_bs + ch0;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int q = q0 + n * 16 + lrow;
        if (q < a.Nq) {
          const f32x4 t = acc[0][n], s = acc[MF - 1][n];
          typename O::quad o;
          o[0] = O::cvt(fast_tanh(t[0] + bt.x) * fast_sigmoid(s[0] + bs.x));
          o[1] = O::cvt(fast_tanh(t[1] + bt.y) * fast_sigmoid(s[1] + bs.y));
          o[2] = O::cvt(fast_tanh(t[2] + bt.z) * fast_sigmoid(s[2] + bs.z));
          o[3] = O::cvt(fast_tanh(t[3] + bt.w) * fast_sigmoid(s[3] + bs.w));
          *reinterpret_cast<typename O::quad*>(yb + (size_t)q * a.y16_ts) = o;
        }
      }
    }
  } else {
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int v = ((chunk * kWaves + wave) * MF + m) * 16 + lq * 4;
      if (v >= a.M) continue;
      int ph = 0, co = v;
      if (a.up_s > 1) { ph = v / a.Cout; co = v - ph * a.Cout; }
      float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.bias) bias = *reinterpret_cast<const float4*>(a.bias + v);
      if (a.bbias) {
        const float4 bb = *reinterpret_cast<const float4*>(a.bbias + (size_t)b * a.bbias_bs + v);
        bias.x += bb.x; bias.y += bb.y; bias.z += bb.z; bias.w += bb.w;
      }
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int q = q0 + n * 16 + lrow;
        const int o = q * a.up_s + ph - a.up_p;
        if (q >= a.Nq || o < 0 || o >= a.T_out) continue;
        float4 val = make_float4(acc[m][n][0] + bias.x, acc[m][n][1] + bias.y, acc[m][n][2] + bias.z, acc[m][n][3] + bias.w);
        if (a.y32b && v >= a.split) {        // skip half of the WN 1x1: out += val
          float* p = a.y32b + (size_t)b * a.y32_bs + (size_t)o * a.y32_ts + (v - a.split);
          float4 old = *reinterpret_cast<float4*>(p);
          old.x += val.x; old.y += val.y; old.z += val.z; old.w += val.w;
          *reinterpret_cast<float4*>(p) = old;
          continue;
        }
        if (a.res) {
          const float4 rr = *reinterpret_cast<const float4*>(a.res + (size_t)b * a.res_bs + (size_t)o * a.res_ts + a.res_c0 + co);
          val.x = rr.x + a.res_sign * val.x; val.y = rr.y + a.res_sign * val.y;
          val.z = rr.z + a.res_sign * val.z; val.w = rr.w + a.res_sign * val.w;
        }
        if (a.y32) {
          float* p = a.y32 + (size_t)b * a.y32_bs + (size_t)o * a.y32_ts + a.y32_c0 + co;
          float4 out = make_float4(val.x * a.y_scale, val.y * a.y_scale, val.z * a.y_scale, val.w * a.y_scale);
          if (a.y_accum) {
            const float4 old = *reinterpret_cast<const float4*>(p);
            out.x += old.x; out.y += old.y; out.z += old.z; out.w += old.w;
          }
          *reinterpret_cast<float4*>(p) = out;
        }
        if (a.y16) {
          typename O::quad h;
          h[0] = O::cvt(lrelu(val.x, a.slope_out)); h[1] = O::cvt(lrelu(val.y, a.slope_out));
          h[2] = O::cvt(lrelu(val.z, a.slope_out)); h[3] = O::cvt(lrelu(val.w, a.slope_out));
          *reinterpret_cast<typename O::quad*>(static_cast<T*>(a.y16) + (size_t)b * a.y16_bs + (size_t)o * a.y16_ts + co) = h;
        }
      }
    }
  }
}

// ------------------------------------------------------------------ launch-side tile selection
struct TileChoice { int NF; int blocks; size_t lds; };

inline TileChoice choose_tile(const ConvDesc& d, int Nq, int batch, int epi, const int* nf_list, int n_nf) {
  const int halo = (d.taps - 1) * d.dil;
  const int rowbytes = d.CinP * 2;
  TileChoice best{0, 0, 0};
  double best_cost = 1e300;
  for (int i = 0; i < n_nf; ++i) {
    const int NF = nf_list[i];
    if (d.MF * NF * 4 > 160 || (NF == 10 && d.MF > 2)) continue;   // accumulator registers / built variants
    const size_t lds = (size_t)(NF * 16 + halo) * rowbytes;
    if (lds > 160 * 1024) continue;
    const int tiles = ceil_div(Nq, NF * 16);
    const long blocks = (long)tiles * batch * d.nchunk;
    const long per_cu = (blocks + 255) / 256;
    // work per block ~ frames computed + a fixed overhead (staging, launch, epilogue), in frame units
    const double cost = (double)per_cu * (NF * 16 + 0.35 * halo + 24.0);
    if (cost < best_cost) { best_cost = cost; best = TileChoice{NF, (int)blocks, lds}; }
  }
  (void)epi;
  return best;
}

template <typename T, int MF, int NF, int EPI>
inline int launch_one(const ConvArgs& a, int batch, int Nq, size_t lds, hipStream_t stream) {
  auto kern = conv_mfma_kernel<T, MF, NF, EPI>;
  static bool attr_done = false;                             // one-time opt-in for > 64 KiB dynamic LDS
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return QVC_ERR_LAUNCH;
    attr_done = true;
  }
  dim3 grid((unsigned)ceil_div(Nq, NF * 16), (unsigned)batch, (unsigned)a.nchunk);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
  return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
}

template <typename T, int MF, int EPI>
inline int launch_nf(const ConvDesc& d, const ConvArgs& a, int batch, hipStream_t stream, int* nf_out) {
  static const int nfs[] = {2, 4, 5, 8, 10};
  const TileChoice tc = choose_tile(d, a.Nq, batch, EPI, nfs, 5);
  if (nf_out) *nf_out = tc.NF;
  switch (tc.NF) {
    case 2: return launch_one<T, MF, 2, EPI>(a, batch, a.Nq, tc.lds, stream);
    case 4: return launch_one<T, MF, 4, EPI>(a, batch, a.Nq, tc.lds, stream);
    case 5: return launch_one<T, MF, 5, EPI>(a, batch, a.Nq, tc.lds, stream);
    case 8: return launch_one<T, MF, 8, EPI>(a, batch, a.Nq, tc.lds, stream);
    case 10:
      if constexpr (MF <= 2) return launch_one<T, MF, 10, EPI>(a, batch, a.Nq, tc.lds, stream);
      return QVC_ERR_BAD_CONFIG;
    default: return QVC_ERR_BAD_CONFIG;                      // tile does not fit LDS
  }
}

template <typename T>
int launch_conv_typed(const ConvDesc& d, const ConvArgs& a, int batch, int epi, void* stream_v, int* nf_out) {
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (epi == EPI_GAU) {
    if (d.MF != 2) return QVC_ERR_BAD_CONFIG;
    return launch_nf<T, 2, EPI_GAU>(d, a, batch, stream, nf_out);
  }
  switch (d.MF) {
    case 1: return launch_nf<T, 1, EPI_STD>(d, a, batch, stream, nf_out);
    case 2: return launch_nf<T, 2, EPI_STD>(d, a, batch, stream, nf_out);
    case 3: return launch_nf<T, 3, EPI_STD>(d, a, batch, stream, nf_out);
    case 4: return launch_nf<T, 4, EPI_STD>(d, a, batch, stream, nf_out);
    default: return QVC_ERR_BAD_CONFIG;
  }
}

}  // namespace qvc
