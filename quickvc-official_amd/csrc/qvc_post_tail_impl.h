// qvc_post_tail_impl.h -- subband_conv_post (k 7, reflect-padded, models.py:388-390) + the iSTFT / band-synthesis
// tail (models.py:394-406) in ONE launch: the 72-channel post-conv frames go from the MFMA accumulators through LDS
// straight into the inverse DFT -- they are never written to memory (the two-launch form hands 46 MB over per step).
//
// One workgroup (4 waves = 2 along the rows, 3 fragments each [72 -> 96 rows] x 2 along the frames) owns NT = 2*NF*16
// post-conv frames [f_lo, f_lo + NT) and the NT - 7 output frames in their middle (the tail's +-3/4-frame halo is
// recomputed by the neighbours: 5.8 % at NT = 128).  Phases, with a barrier between each:
//   stage   act(mean of the three ResBlock outputs) for input rows [f_lo - 3 - 1(reflect), ...) -> LDS tile (operand type)
//   GEMM    acc = conv_post                                   (same K order as conv_mfma_kernel: same bits)
//   post    acc + bias -> LDS fp32 [frame][72], OVER the dead input tile
//   DFT     per (frame, band): 18 values -> 16 windowed samples, in place
//   OLA     band samples [a0 - 7, a0 + OT/4 + 7]
//   FIR     OT output samples, 16-byte stores
// Uses conv_post's packed weights as they are (ConvDesc MF 3 / WM 2 / 1 chunk, checked by post_tail_supported).
#pragma once
#include "qvc_conv_impl.h"
#include "qvc_tail_impl.h"

namespace qvc {

template <int NF> struct PostTailGeom {
  static constexpr int NT = 2 * NF * 16;        // post-conv frames of the tile
  static constexpr int OF = NT - 7;             // output frames owned
  static constexpr int OT = OF * 16;            // output samples owned
  static constexpr int NY = OT / 4 + 15;        // band samples needed
};
inline size_t post_tail_lds(int nf, int taps, int CinP) {
  const int NT = 2 * nf * 16, NY = (NT - 7) * 4 + 15;
  const size_t tile = std::max<size_t>((size_t)(NT + taps - 1) * CinP * 2, (size_t)NT * kPostC * 4);
  return align_up((int64_t)tile, 16) + (size_t)kBands * (NY + 1) * 4;
}

template <typename T, int NF>
// (three workgroups per CU at NF 4, four at NF 2: the phases of one workgroup are serial -- memory, MFMA, VALU/LDS
//  in turn -- so it is the neighbours that fill each unit; measured 94 -> 78 us going from two to three)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NF == 2 ? 4 : 3, NF == 2 ? 4 : 3)))
void post_tail_kernel(const PostTailArgs A) {
  using O = Op<T>;
  using frag = typename O::frag;
  using G = PostTailGeom<NF>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MF = 3, WM = 2;
  constexpr int NT = G::NT, OT = G::OT, NY = G::NY;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int lrow = lane & 15, lq = lane >> 4;
  const int b = blockIdx.y;
  const int o0 = blockIdx.x * OT;
  const int a0 = o0 >> 2;                       // first band sample owned by this block
  const int f_lo = (o0 >> 4) - 3;               // first post-conv frame of the tile
  const int Lpad = 4 * (A.F - 1);
  const int n_out = 4 * Lpad;
  const int Fb = ragged_len(A.rg, b, A.F);      // this utterance's post-conv frames are [Flo, Fb)
  const int Flo = ragged_lo(A.rg, b);
  const int L = Fb > 0 ? 4 * (Fb - 1) : 0;
  float* outb = A.out + (size_t)b * n_out;

  if (f_lo >= Fb || f_lo + NT <= Flo) {         // no frame of this utterance in the tile: its samples are zeros
    for (int j = tid; j < OT / 4; j += 256) {
      const int o = o0 + 4 * j;
      if (o + 3 < n_out) *reinterpret_cast<float4*>(outb + o) = make_float4(0.f, 0.f, 0.f, 0.f);
      else for (int r = 0; r < 4; ++r) if (o + r < n_out) outb[o + r] = 0.f;
    }
    return;
  }

  const int taps = A.c.taps;
  const int R = NT + taps - 1;
  const int rowbytes = A.c.CinP * 2;
  const int cpr = A.c.CinP >> 3;
  const Swz sm = swz_mode(cpr);
  const int t_base = f_lo - A.c.left;           // input frame (after the reflect pad) of tile row 0
  const int Tin = ragged_len(A.c.rg, b, A.c.T_in);
  const int Tlo = ragged_lo(A.c.rg, b);
  const int in_bytes = R * rowbytes, post_bytes = NT * kPostC * 4;
  const int tile_bytes = ((in_bytes > post_bytes ? in_bytes : post_bytes) + 15) & ~15;   // == post_tail_lds()'s first region
  float* s_post = reinterpret_cast<float*>(smem);
  float* s_y = reinterpret_cast<float*>(smem + tile_bytes);       // [band][NY + 1]

  // ------------------------------------------------------------------ stage: act(mean(x, x2, x3)) -> operand-type tile
  // A thread keeps its 16-byte column c8 and walks rows r0, r0 + rstep, ... (256 threads = rstep whole rows when
  // cpr divides 256: 128 channels -> 16 rows per pass); all loads of a batch go out before the first conversion.
  {
    constexpr int kU = 5;
    const size_t boff = (size_t)b * A.c.x_bs + A.c.x_c0;
    const T* xb = static_cast<const T*>(A.c.x) + boff;
    const T* xb2 = static_cast<const T*>(A.c.x2) + boff;
    const T* xb3 = static_cast<const T*>(A.c.x3) + boff;
    const float slope = A.c.slope_in;
    const bool regular = (256 % cpr) == 0;
    const int rstep = regular ? 256 / cpr : 0;
    const int total = R * cpr;
    for (int base = tid; base < total; base += 256 * kU) {
      uint4 v1[kU], v2[kU], v3[kU];
      int dst[kU];
      int r = base / cpr, c8 = base - r * cpr;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int idx = base + u * 256;
        if (!regular) { r = idx / cpr; c8 = idx - r * cpr; }
        const int ti = t_base + r;
        bool ok; int src;
        if (A.c.reflect) { ok = ti >= Tlo && ti <= Tin; src = ti == Tlo ? Tlo + 1 : ti - 1; }
        else { ok = ti >= Tlo && ti < Tin; src = ti; }
        ok = ok && idx < total && (c8 * 8 < A.c.Cin);
        v1[u] = make_uint4(0u, 0u, 0u, 0u); v2[u] = v1[u]; v3[u] = v1[u];
        if (ok) {
          const size_t o = (size_t)src * A.c.x_ts + c8 * 8;
          v1[u] = *reinterpret_cast<const uint4*>(xb + o);
          v2[u] = *reinterpret_cast<const uint4*>(xb2 + o);
          v3[u] = *reinterpret_cast<const uint4*>(xb3 + o);
        }
        dst[u] = idx < total ? r * rowbytes + ((rotc(c8, sm) ^ swz(r, sm)) << 4) : -1;
        r += rstep;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (dst[u] < 0) continue;
        frag h1, h2, h3, h;
        __builtin_memcpy(&h1, &v1[u], 16); __builtin_memcpy(&h2, &v2[u], 16); __builtin_memcpy(&h3, &v3[u], 16);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          h[i] = O::cvt(lrelu(((float)h1[i] + (float)h2[i] + (float)h3[i]) * (1.f / 3.f), slope));
        *reinterpret_cast<frag*>(smem + dst[u]) = h;
      }
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ GEMM (A from global through the ring, B from the tile)
  f32x4 acc[MF][NF];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const frag* ap = static_cast<const frag*>(A.c.w) + ((size_t)wm * A.c.nIt * MF) * 64 + lane;
  if (wm == 0) {
    gemm_loop<T, MF, NF, QVC_PF_CONV>(acc, ap, A.c.nIt, A.c.KS, 1, smem, rowbytes, sm, wn * (NF * 16) + lrow, lq, 0);
  } else {        // rows 48..95: the third fragment (80..95) is all padding -- 72 channels -- and is neither loaded nor multiplied
    f32x4 (&acc2)[2][NF] = reinterpret_cast<f32x4 (&)[2][NF]>(acc);
    gemm_loop<T, 2, NF, QVC_PF_CONV, MF>(acc2, ap, A.c.nIt, A.c.KS, 1, smem, rowbytes, sm, wn * (NF * 16) + lrow, lq, 0);
  }
  __syncthreads();                              // every wave is done with the input tile

  // ------------------------------------------------------------------ post-conv frames -> LDS fp32 [frame][72]
#pragma unroll
  for (int m = 0; m < MF; ++m) {
    const int v = (wm * MF + m) * 16 + lq * 4;
    if (v >= kPostC) continue;
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (A.c.bias) bias = *reinterpret_cast<const float4*>(A.c.bias + v);
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      const int fr = wn * (NF * 16) + n * 16 + lrow;
      const int t = f_lo + fr;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t >= Flo && t < Fb) val = make_float4(acc[m][n][0] + bias.x, acc[m][n][1] + bias.y, acc[m][n][2] + bias.z, acc[m][n][3] + bias.w);
      *reinterpret_cast<float4*>(&s_post[fr * kPostC + v]) = val;
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ per (frame, band): polar -> inverse DFT -> window, in place
  for (int item = tid; item < NT * kBands; item += 256) {
    const int fr = item >> 2, k = item & 3;
    float* sp = &s_post[fr * kPostC + k * 2 * kBins];
    float xw[16];
    tail_dft(sp, xw);
#pragma unroll
    for (int m = 0; m < 16; ++m) sp[m] = xw[m];
  }
  __syncthreads();

  // ------------------------------------------------------------------ overlap-add + envelope: band samples n = a0 - 7 + i
  for (int item = tid; item < NY * kBands; item += 256) {
    const int k = item / NY, i = item - k * NY;
    const int n = a0 - 7 + i;
    s_y[k * (NY + 1) + i] = tail_ola(n, Flo, Fb, L, [&](int t, int m) { return s_post[(t - f_lo) * kPostC + k * 2 * kBins + m]; });
  }
  __syncthreads();

  // ------------------------------------------------------------------ synthesis FIR: 4 consecutive outputs per item
  for (int j = tid; j < OT / 4; j += 256) {
    const int ia = j + 7;
    float out[4];
    tail_fir<kTaps>([&](int k, int d) { return s_y[k * (NY + 1) + ia + d]; }, A.fir, out);
    const int o = o0 + 4 * j;
    if (o >= 4 * L || o < 16 * Flo) out[0] = out[1] = out[2] = out[3] = 0.f;
    if (o + 3 < n_out) {
      *reinterpret_cast<float4*>(outb + o) = make_float4(out[0], out[1], out[2], out[3]);
    } else {
      for (int r = 0; r < 4; ++r) if (o + r < n_out) outb[o + r] = out[r];
    }
  }
}

template <typename T, int NF>
inline int launch_post_tail_nf(const ConvDesc& d, const PostTailArgs& a, int batch, hipStream_t stream) {
  auto kern = post_tail_kernel<T, NF>;
  const size_t lds = post_tail_lds(NF, d.taps, d.CinP);
  if (lds > 160 * 1024) return QVC_ERR_BAD_CONFIG;
  static std::atomic<uint32_t> lds_ok{0};
  if (!allow_big_lds(lds_ok, reinterpret_cast<const void*>(kern))) return QVC_ERR_LAUNCH;
  const int n_out = 16 * (a.F - 1);
  hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(n_out, PostTailGeom<NF>::OT), (unsigned)batch), dim3(256), lds, stream, a);
  return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
}

template <typename T>
int launch_post_tail_typed(const ConvDesc& d, const PostTailArgs& a, int batch, void* stream_v) {
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (!post_tail_supported(d) || a.F < 2 || !a.c.x2 || !a.c.x3 || a.c.x_kind != XK_OP_FM) return QVC_ERR_BAD_CONFIG;
  if (debug_get(DBG_POST_TAIL_NF) == 2) return launch_post_tail_nf<T, 2>(d, a, batch, stream);
  return launch_post_tail_nf<T, 4>(d, a, batch, stream);
}

}  // namespace qvc
