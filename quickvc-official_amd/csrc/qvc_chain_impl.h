// qvc_chain_impl.h -- a whole ResBlock1 (modules.py:147-154: three pairs x = x + conv2(lrelu(conv1(lrelu(x)))) with
// dilations d0, d1, d2) in ONE launch, for the chains whose receptive field is short (kernel 3: 12 frames per side).
//
// Why: the fused pair kernel (rbpair_kernel) reads a chain's stream twice (tile + residual) and writes it once per
// pair, and its memory phases do not overlap its GEMMs (204 us of GEMMs + 68 us of memory phases per
// three-chain launch, additive).  For the k = 3 chain those phases are as long as for k = 11 while its GEMMs are a
// quarter of the work -- so its three pairs are chained on chip: the stream is read ONCE (tile + 2 x 12 halo frames),
// stays in LDS as a raw f16 tile (the residual) next to the activated operand tile, and is written ONCE.
// Price: 2 x halo frames recomputed per tile (144-frame tile, 120 produced: +20 % of the chain's MFMA work, the
// cheapest chain) and one more barrier per pair.
//
// Same math per element as three rbpair launches: same K order, the stream rounded to its memory type after every
// pair, the intermediates zeroed outside the utterance -- results are bit-identical (GPU test).
#pragma once
#include "qvc_conv_impl.h"

namespace qvc {

template <typename T, int MF, int NF, int NWV, typename TS>
__global__ __launch_bounds__(NWV * 64) void rbchain_kernel(const ChainArgs A) {
  using O = Op<T>;
  using frag = typename O::frag;
  using OS = Op<TS>;
  using sfrag = typename OS::frag;
  static_assert(MF % 2 == 0, "lane-packed rows in 16-byte pieces");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTHR = NWV * 64;
  constexpr int ROWS = NF * 16;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);      // all waves split the rows (WN = 1)
  const int lrow = lane & 15, lq = lane >> 4;
  const int b = blockIdx.y;
  const PairArgs a0 = A.p[0];                 // a copy: a reference into the by-value arguments would send them to scratch
  const int C = a0.C, CP = a0.CP;
  const int rowbytes = CP * 2, cpr = CP >> 3;
  const Swz sm = swz_mode(cpr);
  const int q0 = blockIdx.x * A.NT;
  const int F0 = q0 - A.halo;                  // frame of tile row 0
  const int Tb = ragged_len(A.rg, b, a0.T), Tlo = ragged_lo(A.rg, b);
  if (q0 >= Tb) return;
  char* act = smem;                            // activated operand tile: row mrg + r <-> frame F0 + r
  char* raw = smem + (size_t)(ROWS + 2 * A.margin) * rowbytes;   // the stream itself (residual), row r <-> frame F0 + r
  const int mrg = A.margin;
  const int cb = wm * MF * 16 + lq * 4 * MF;   // first of this lane's 4 * MF consecutive channels

  {   // ---- stage: raw x and lrelu(x) (operand type); margins of the operand tile = zeros
    const TS* xb = static_cast<const TS*>(a0.x) + (size_t)b * a0.bs;
    for (int i = tid; i < 2 * mrg * cpr; i += NTHR) {
      const int r = i / cpr, c8 = i - r * cpr;
      const int row = r < mrg ? r : ROWS + r;
      *reinterpret_cast<uint4*>(act + row * rowbytes + ((rotc(c8, sm) ^ swz(row, sm)) << 4)) = make_uint4(0u, 0u, 0u, 0u);
    }
    const int total = ROWS * cpr;
    constexpr int kU = 16;
    const int rstep = NTHR / cpr, cstep = NTHR - rstep * cpr;
    for (int base = tid; base < total; base += NTHR * kU) {
      uint4 v[kU];
      const int r_0 = base / cpr, c_0 = base - r_0 * cpr;
      int r = r_0, c8 = c_0;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int idx = base + u * NTHR;
        const int ti = F0 + r;
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        if (idx < total && ti >= Tlo && ti < Tb && c8 * 8 < C) v[u] = *reinterpret_cast<const uint4*>(xb + (size_t)ti * C + c8 * 8);
        c8 += cstep; r += rstep;
        if (c8 >= cpr) { c8 -= cpr; ++r; }
      }
      r = r_0; c8 = c_0;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int idx = base + u * NTHR;
        if (idx < total) {
          frag o;
          if constexpr (std::is_same<T, TS>::value) {
            frag h; __builtin_memcpy(&h, &v[u], 16);
            o = lrelu8<T>(h, a0.slope);
          } else {
            sfrag h; __builtin_memcpy(&h, &v[u], 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = O::cvt(lrelu((float)h[e], a0.slope));
          }
          *reinterpret_cast<uint4*>(raw + r * rowbytes + ((rotc(c8, sm) ^ swz(r, sm)) << 4)) = v[u];
          *reinterpret_cast<frag*>(act + (mrg + r) * rowbytes + ((rotc(c8, sm) ^ swz(mrg + r, sm)) << 4)) = o;
        }
        c8 += cstep; r += rstep;
        if (c8 >= cpr) { c8 -= cpr; ++r; }
      }
    }
  }
  __syncthreads();

  for (int q = 0; q < A.n; ++q) {
    PairArgs a = A.p[0];                       // scalar selects (a dynamic index into the kernel arguments would go through scratch)
    if (q == 1) a = A.p[1];
    if (q == 2) a = A.p[2];
    const int h2 = (a.k - 1) / 2, h1 = h2 * a.dil;
    const bool last = q == A.n - 1;
    f32x4 acc[MF][NF];
    {   // ---- GEMM1 (dilation d_q) over the whole tile, bias + lrelu -> the operand tile, in place
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      const frag* ap = static_cast<const frag*>(a.w1) + ((size_t)wm * a.nIt * MF) * 64 + lane;
      gemm_loop<T, MF, NF, QVC_PF_CONV>(acc, ap, a.nIt, a.KS, a.dil, act, rowbytes, sm, mrg - h1 + lrow, lq, 0);
      float4 bias[MF];
#pragma unroll
      for (int m = 0; m < MF; ++m) bias[m] = *reinterpret_cast<const float4*>(a.b1 + cb + m * 4);
      __syncthreads();                         // every wave is done reading the operand tile
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int r = n * 16 + lrow;
        const int f = F0 + r;
        const bool inside = f >= Tlo && f < Tb;                  // conv2 zero-pads outside the utterance
        char* rowp = act + (mrg + r) * rowbytes;
        const int sw = swz(mrg + r, sm);
#pragma unroll
        for (int m = 0; m < MF; m += 2) {
          const int v = cb + m * 4;
          if (v >= CP) continue;
          frag h;
          if (inside && v < C) {
            h[0] = O::cvt(lrelu(acc[m][n][0] + bias[m].x, a.slope)); h[1] = O::cvt(lrelu(acc[m][n][1] + bias[m].y, a.slope));
            h[2] = O::cvt(lrelu(acc[m][n][2] + bias[m].z, a.slope)); h[3] = O::cvt(lrelu(acc[m][n][3] + bias[m].w, a.slope));
            h[4] = O::cvt(lrelu(acc[m + 1][n][0] + bias[m + 1].x, a.slope)); h[5] = O::cvt(lrelu(acc[m + 1][n][1] + bias[m + 1].y, a.slope));
            h[6] = O::cvt(lrelu(acc[m + 1][n][2] + bias[m + 1].z, a.slope)); h[7] = O::cvt(lrelu(acc[m + 1][n][3] + bias[m + 1].w, a.slope));
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (T)0.f;
          }
          *reinterpret_cast<frag*>(rowp + ((rotc(v >> 3, sm) ^ sw) << 4)) = h;
        }
      }
    }
    __syncthreads();
    {   // ---- GEMM2 (dilation 1) + bias + residual (raw tile) -> the stream: back into both tiles, or out to memory
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      const frag* ap = static_cast<const frag*>(a.w2) + ((size_t)wm * a.nIt * MF) * 64 + lane;
      gemm_loop<T, MF, NF, QVC_PF_CONV>(acc, ap, a.nIt, a.KS, 1, act, rowbytes, sm, mrg - h2 + lrow, lq, 0);
      float4 bias[MF];
#pragma unroll
      for (int m = 0; m < MF; ++m) bias[m] = *reinterpret_cast<const float4*>(a.b2 + cb + m * 4);
      if (!last) __syncthreads();              // every wave is done reading the intermediate: the next pair's input may overwrite it
      TS* yb = static_cast<TS*>(A.n == 3 ? A.p[2].y : A.p[1].y) + (size_t)b * a.bs;   // the chain's output (scalar select)
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int r = n * 16 + lrow;
        const int f = F0 + r;
        const bool inside = f >= Tlo && f < Tb;
        const int swr = swz(r, sm), swa = swz(mrg + r, sm);
#pragma unroll
        for (int m = 0; m < MF; m += 2) {
          const int v = cb + m * 4;
          if (v >= C) continue;
          char* rp = raw + r * rowbytes + ((rotc(v >> 3, sm) ^ swr) << 4);
          sfrag r8 = *reinterpret_cast<const sfrag*>(rp);
          sfrag h;
          h[0] = OS::cvt(acc[m][n][0] + bias[m].x + (float)r8[0]); h[1] = OS::cvt(acc[m][n][1] + bias[m].y + (float)r8[1]);
          h[2] = OS::cvt(acc[m][n][2] + bias[m].z + (float)r8[2]); h[3] = OS::cvt(acc[m][n][3] + bias[m].w + (float)r8[3]);
          h[4] = OS::cvt(acc[m + 1][n][0] + bias[m + 1].x + (float)r8[4]); h[5] = OS::cvt(acc[m + 1][n][1] + bias[m + 1].y + (float)r8[5]);
          h[6] = OS::cvt(acc[m + 1][n][2] + bias[m + 1].z + (float)r8[6]); h[7] = OS::cvt(acc[m + 1][n][3] + bias[m + 1].w + (float)r8[7]);
          if (last) {
            if (r >= A.halo && r < A.halo + A.NT && f < Tb) *reinterpret_cast<sfrag*>(yb + (size_t)f * C + v) = h;
          } else {
            if (!inside) {
#pragma unroll
              for (int e = 0; e < 8; ++e) h[e] = (TS)0.f;        // the next pair's convs zero-pad outside the utterance
            }
            *reinterpret_cast<sfrag*>(rp) = h;
            frag o;
            if constexpr (std::is_same<T, TS>::value) {
              o = lrelu8<T>(h, a.slope);
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = O::cvt(lrelu((float)h[e], a.slope));
            }
            *reinterpret_cast<frag*>(act + (mrg + r) * rowbytes + ((rotc(v >> 3, sm) ^ swa) << 4)) = o;
          }
        }
      }
    }
    if (!last) __syncthreads();
  }
}

template <typename T, int MF, int NF, int NWV, typename TS>
inline int launch_chain_one(const ChainArgs& a, int batch, size_t lds, hipStream_t stream) {
  auto kern = rbchain_kernel<T, MF, NF, NWV, TS>;
  static std::atomic<uint32_t> lds_ok{0};                  // > 64 KiB dynamic LDS: opt in once per device
  if (!allow_big_lds(lds_ok, reinterpret_cast<const void*>(kern))) return QVC_ERR_LAUNCH;
  const dim3 grid((unsigned)ceil_div(a.p[0].T, a.NT), (unsigned)batch);
  hipLaunchKernelGGL(kern, grid, dim3(NWV * 64), lds, stream, a);
  return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
}

template <typename T, int MF, int NWV, typename TS>
inline int launch_chain_nf(int NF, const ChainArgs& a, int batch, size_t lds, hipStream_t stream) {
  switch (NF) {
    case 9: if constexpr (MF * 9 * 4 <= 96) return launch_chain_one<T, MF, 9, NWV, TS>(a, batch, lds, stream); break;
    case 8: if constexpr (MF * 8 * 4 <= 96) return launch_chain_one<T, MF, 8, NWV, TS>(a, batch, lds, stream); break;
    case 6: if constexpr (MF * 6 * 4 <= (MF > 2 ? 80 : 96)) return launch_chain_one<T, MF, 6, NWV, TS>(a, batch, lds, stream); break;
    case 5: return launch_chain_one<T, MF, 5, NWV, TS>(a, batch, lds, stream);
    case 4: return launch_chain_one<T, MF, 4, NWV, TS>(a, batch, lds, stream);
    default: break;
  }
  return QVC_ERR_BAD_CONFIG;
}

template <typename T, typename TS>
int launch_chain_typed(const ConvDesc* d1, const ConvDesc* d2, ChainArgs a, int batch, void* stream_v, int* nf_out) {
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (!chain_supported(d1, d2, a.n)) return QVC_ERR_BAD_CONFIG;
  ChainGeom g;
  const int NF = chain_pick_nf(d1, a.n, &g);
  a.halo = g.halo; a.margin = g.margin; a.NT = g.NT;
  if (nf_out) *nf_out = NF;
  switch (d1[0].WM * 10 + d1[0].MF) {
    case 42: return launch_chain_nf<T, 2, 4, TS>(NF, a, batch, g.lds, stream);
    case 44: return launch_chain_nf<T, 4, 4, TS>(NF, a, batch, g.lds, stream);
    case 82: return launch_chain_nf<T, 2, 8, TS>(NF, a, batch, g.lds, stream);
    case 84: return launch_chain_nf<T, 4, 8, TS>(NF, a, batch, g.lds, stream);
    default: return QVC_ERR_BAD_CONFIG;
  }
}

}  // namespace qvc
