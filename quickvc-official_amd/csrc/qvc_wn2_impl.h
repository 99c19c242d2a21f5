// qvc_wn2_impl.h -- the WaveNet stack (modules.py:69-114, and a coupling layer's pre / post 1x1 convs, modules.py:212-217)
// with a CONTINUOUS weight stream.
//
// What bounds a WaveNet layer on this path (profiles/r03_wn_stamps.txt, tools/stream_probe): every CU carries 48 frames
// through a layer and has to take in the layer's whole 864 KB of weights for them -- each byte is used by exactly one
// wave for three MFMAs.  A CU's vector-memory path delivers ~58-60 B/clk of such an L2-resident stream when all 256
// CUs pull at once (tools/stream_probe), i.e. 14.9 k cycles per layer, against 10.4 k cycles of MFMA work.  The
// round-2 kernel (wn_stack_kernel) moved 32 B/clk: its K loops prefetch three k-steps ahead, stop streaming at every
// phase boundary (gate math, barriers, epilogue, the next GEMM's ring priming) and spend ~45 instructions per k-step
// on addresses and loop control.  Here
//   * the launch's weights are ONE stream per wave -- layer l's k-tap conv, its 1x1, layer l+1's conv ... -- prefetched
//     SIX k-steps ahead through a six-slot register ring that never drains (a slot is refilled right after its MFMAs): while a wave does its gate math, waits at
//     a barrier or runs an epilogue, the next GEMM's fragments are already on their way;
//   * a layer's 36 k-steps are fully unrolled for the shipped shape (hidden 192, kernel 5): ring slots, taps and LDS
//     offsets are compile-time constants, the B-fragment address of a k-step is one of ten per-lane bases computed once
//     per launch plus an immediate, so a k-step is 2 global loads + 3 LDS reads + 6 MFMAs and almost nothing else;
//   * the first fragments are requested before anything else in the kernel, so they travel under the prologue.
// Same math, same K order per output as wn_stack_kernel: results are bit-identical (GPU test).
#pragma once
#include <utility>
#include "qvc_conv_impl.h"

namespace qvc {

// f(integral_constant<int, 0>{}), f(<1>), ... f(<N-1>): a fully unrolled loop whose index is a compile-time constant
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Developer experiment (tools/conv_bench): re-synchronise the workgroup's waves every QVC_WN2_SYNC k-steps of GEMM1.
// The three waves of a SIMD do not advance together -- the oldest wins every arbitration and finishes its k-tap GEMM
// in 6 k cycles, the youngest needs 14 k -- and the idea was that a wave idling at the layer's barrier streams nothing.
// Measured (profiles/r03_wn_stamps.txt): with a plain s_barrier every 6 k-steps ALL waves take 14.2 k cycles and the
// layer gets slower (12.8 vs 12.2 us): the CU's intake of ~50 B/clk is the bound whoever issues the loads.  Off.
#ifndef QVC_WN2_SYNC
#define QVC_WN2_SYNC 0
#endif

// the shapes this kernel is built for
inline bool wn2_supported(const ConvDesc& din, const WnStackArgs& a) {
  return wn_layout_ok(din) && din.CinP == 192 && din.taps == 5 && a.layers >= 1 && a.layers <= 4 && wn_stack_nf(din.taps, a.layers) == 3 &&
         a.HP == 192 && a.H <= 192 && a.KS == 6 && a.nIt1 == 30 && (!a.w_post || a.post_mf == 1) && (!a.w_pre || a.pre_KS <= 6);
}

template <typename T, int KS, int TAPS, int PM>
__global__ __launch_bounds__(KS * 2 * 64) void wn_stack2_kernel(const WnStackArgs a) {
  using O = Op<T>;
  using frag = typename O::frag;
  using quad = typename O::quad;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HP = KS * 32, RB = HP * 2, CPR = HP / 8;
  constexpr int NF = 3, NB = NF * 16, OUTF = kWnOutFrames, ON = 3;
  constexpr int LEFT = (TAPS - 1) / 2, R = NB + TAPS - 1;
  constexpr int NK1 = TAPS * KS, NK2 = KS, NKL = NK1 + NK2;
  constexpr int RING = 6, PF = RING;          // a k-step's slot is refilled right after its MFMAs: six k-steps (12 KiB per wave) in flight
  static_assert(NKL % RING == 0, "a layer must start on ring slot 0");
  static_assert(CPR % 8 == 0 && CPR % 16 != 0, "row swizzle = row & 7 (see swz_mode)");
  constexpr int ACTS = R * RB;                // byte offset of the gated-activation tile
  constexpr int NTH = KS * 2 * 64;
  // fp32 residual stream and skip sum of the window: lane-private 16-byte slots in LDS ([n][thread]), touched only in a
  // layer's epilogue -- in registers (24 per lane) they pushed the K loops' ring into scratch memory
  constexpr int XRES = ACTS + NB * RB, OACC = XRES + NF * NTH * 16;
  const Swz sm{0, 7, 0};

  const int tid = threadIdx.x, lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lq = lane >> 4;
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * OUTF;
#ifdef QVC_STAMP
  unsigned long long* const st_ = a.stamps ? a.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + wm) * 32 : nullptr;
  if (st_ && lane == 0) st_[27] = __builtin_amdgcn_s_memrealtime();
#endif
  QVC_ST(0);
  const int Tb = ragged_len(a.rg, b, a.T);
  const int Tlo = ragged_lo(a.rg, b);
  if (q0 >= Tb) return;
  const int halo = LEFT * a.layers;
  const int w0 = q0 - halo;                   // first frame of the window; column j <-> frame w0 + j
  char* acts = smem + ACTS;

  const int ch0 = wm * 16 + lq * 4;           // this lane's four channels
  // ---- the weight stream: request the first PF k-steps of layer 0 before anything else
  // (issuing the prologue's own few loads -- x, the running skip sum -- ahead of the ring was tried: the prologue got
  //  longer, 8.9 k -> 10.7 k cycles, profiles/r03_wn_stamps.txt)
  frag ring[RING][2];
  const frag* ap1 = static_cast<const frag*>(a.w_in[0]) + ((size_t)wm * NK1 * 2) * 64 + lane;
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    ring[u][0] = ap1[(size_t)(u * 2) * 64];
    ring[u][1] = ap1[(size_t)(u * 2 + 1) * 64];
  }

  // B-fragment bases: tile row (tap + lrow), 16-byte chunk (ks*4 + lq) ^ ((tap + lrow) & 7).  The swizzle touches the
  // low three bits of the chunk only: bits 0-1 = lq, bit 2 = ks & 1 -- so one base per (tap, ks parity), and
  // (ks >> 1) * 128 + n * 16 * RB are immediates.
  int boff[TAPS];                             // even k-steps; odd ones: ^ 64 (chunk bit 2)
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const int row = t + lrow;
    boff[t] = row * RB + ((lq ^ (row & 7)) << 4);
  }
  float4* const xres = reinterpret_cast<float4*>(smem + XRES) + tid;     // [n * NTH]
  float4* const oacc = reinterpret_cast<float4*>(smem + OACC) + tid;

  // zero the x tile once: rows outside the window and K-padding channels must stay finite zeros
  for (int i = tid; i < (R * RB) >> 4; i += NTH) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);

  f32x4 xr[NF];
  if (a.w_pre) {
    // fused `pre` 1x1 (modules.py:212): the z slice of the window through the (still unused) acts tile
    const int pcpr = a.pre_KS * 4;
    for (int i = tid; i < (NB * RB) >> 4; i += NTH) reinterpret_cast<uint4*>(acts)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const float* zb = a.z + (size_t)b * a.z_bs + a.pre_c0;
    for (int idx = tid; idx < NB * pcpr; idx += NTH) {
      const int r = idx / pcpr, c8 = idx - r * pcpr;
      const int q = w0 + r;
      if (q >= Tlo && q < Tb && c8 * 8 < a.pre_cin) {
        const float4* p = reinterpret_cast<const float4*>(zb + (size_t)q * a.z_ts + c8 * 8);
        const float4 v0 = p[0], v1 = p[1];
        frag h;
        h[0] = O::cvt(v0.x); h[1] = O::cvt(v0.y); h[2] = O::cvt(v0.z); h[3] = O::cvt(v0.w);
        h[4] = O::cvt(v1.x); h[5] = O::cvt(v1.y); h[6] = O::cvt(v1.z); h[7] = O::cvt(v1.w);
        *reinterpret_cast<frag*>(acts + r * RB + ((rotc(c8, sm) ^ swz(r, sm)) << 4)) = h;
      }
    }
    __syncthreads();
    f32x4 pacc[1][NF];
#pragma unroll
    for (int n = 0; n < NF; ++n) pacc[0][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const frag* app = static_cast<const frag*>(a.w_pre) + ((size_t)wm * a.pre_KS) * 64 + lane;
    gemm_loop<T, 1, NF, QVC_PF_STACK>(pacc, app, a.pre_KS, a.pre_KS, 1, acts, RB, sm, lrow, lq, 0);
    float4 bp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ch0 < a.H) bp = *reinterpret_cast<const float4*>(a.b_pre + ch0);
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      const int q = w0 + n * 16 + lrow;
      const bool in = ch0 < a.H && q >= Tlo && q < Tb;
      xr[n] = in ? f32x4{pacc[0][n][0] + bp.x, pacc[0][n][1] + bp.y, pacc[0][n][2] + bp.z, pacc[0][n][3] + bp.w}
                 : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
#pragma unroll
  for (int n = 0; n < NF; ++n) {
    const int q = w0 + n * 16 + lrow;
    if (!a.w_pre) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ch0 < a.H && q >= Tlo && q < Tb) v = *reinterpret_cast<const float4*>(a.x0 + (size_t)b * a.bs + (size_t)q * a.H + ch0);
      xr[n] = f32x4{v.x, v.y, v.z, v.w};
    }
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.accum && ch0 < a.H && q >= q0 && q < q0 + OUTF && q < Tb)      // continue a previous launch's skip sum
      o = *reinterpret_cast<const float4*>(a.out + (size_t)b * a.bs + (size_t)q * a.H + ch0);
    oacc[n * NTH] = o;
  }
  __syncthreads();                            // tile zeroed before anybody writes x into it
  // the residual stream's operand-type copy in the x tile (GEMM1's B operand)
  const int xt_off = (lrow + LEFT) * RB + (((ch0 >> 3) ^ ((lrow + LEFT) & 7)) << 4) + (ch0 & 7) * 2;   // + n * 16 * RB
  auto put_x = [&](const f32x4 (&x)[NF]) {
    if (ch0 >= a.H) return;
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      quad h;
      h[0] = O::cvt(x[n][0]); h[1] = O::cvt(x[n][1]); h[2] = O::cvt(x[n][2]); h[3] = O::cvt(x[n][3]);
      *reinterpret_cast<quad*>(smem + xt_off + n * 16 * RB) = h;
    }
  };
  put_x(xr);
#pragma unroll
  for (int n = 0; n < NF; ++n) xres[n * NTH] = make_float4(xr[n][0], xr[n][1], xr[n][2], xr[n][3]);
  __syncthreads();
  QVC_ST(1);

  for (int l = 0; l < a.layers; ++l) {
    const bool last = a.final_layer && l == a.layers - 1;      // the network's last layer has no residual half
    const bool has_next = l + 1 < a.layers;
    const int mf2 = last ? 1 : 2;
    const frag* ap2 = static_cast<const frag*>(a.w_rs[l]) + ((size_t)wm * NK2 * mf2) * 64 + lane;
    const frag* apn = static_cast<const frag*>(a.w_in[has_next ? l + 1 : l]) + ((size_t)wm * NK1 * 2) * 64 + lane;
    const int s2 = mf2 * 64, d2 = (mf2 - 1) * 64;
    // k-step kk of THIS layer's stream (kk >= NKL: the next layer's), into ring slot kk % RING.  Always exactly two
    // loads, from valid addresses: a branch around a load would make the compiler's wait counts assume the path without
    // it and drain the ring (the network's last 1x1 has one fragment per k-step: loaded twice; behind the launch's
    // last layer the ring refills from that layer's own first k-steps, never used)
    auto prefetch = [&](auto kk_c) {
      constexpr int kk = decltype(kk_c)::value;
      constexpr int slot = kk % RING;
      if constexpr (kk < NK1) {
        ring[slot][0] = ap1[(size_t)(kk * 2) * 64];
        ring[slot][1] = ap1[(size_t)(kk * 2 + 1) * 64];
      } else if constexpr (kk < NKL) {
        const frag* p = ap2 + (kk - NK1) * s2;
        ring[slot][0] = p[0];
        ring[slot][1] = p[d2];
      } else {
        ring[slot][0] = apn[(size_t)((kk - NKL) * 2) * 64];
        ring[slot][1] = apn[(size_t)((kk - NKL) * 2 + 1) * 64];
      }
    };
    // conditioning / bias rows of the gate: requested a few k-steps before GEMM1 ends, used right after it
    const float* bb = a.bbias + (size_t)b * a.bbias_bs + (size_t)l * 2 * a.H + (ch0 < a.H ? ch0 : 0);
    float4 bt, bs;

    {   // ---- GEMM1 (k taps) + conditioning + gate -> acts tile
      f32x4 acc[2][NF];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      frag bf[2][NF];
      auto read_b = [&](auto k_c, frag (&dst)[NF]) {
        constexpr int k = decltype(k_c)::value;
        constexpr int tap = k / KS, ks = k % KS;
        const char* bp = smem + ((ks & 1) ? (boff[tap] ^ 64) : boff[tap]) + (ks >> 1) * 128;
#pragma unroll
        for (int n = 0; n < NF; ++n) dst[n] = *reinterpret_cast<const frag*>(bp + n * 16 * RB);
      };
      read_b(std::integral_constant<int, 0>{}, bf[0]);
      auto step = [&](auto k_c) {
        constexpr int k = decltype(k_c)::value;
        if constexpr (k + 1 < NK1) { if (!QVC_ABL(6)) read_b(std::integral_constant<int, k + 1>{}, bf[(k + 1) & 1]); }
        if constexpr (k == NK1 - 6) { bt = *reinterpret_cast<const float4*>(bb); bs = *reinterpret_cast<const float4*>(bb + a.H); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m][n] = O::mfma(ring[k % RING][m], bf[k & 1][n], acc[m][n]);
        __builtin_amdgcn_sched_barrier(0);
        if (!QVC_ABL(5)) prefetch(std::integral_constant<int, k + PF>{});      // into the slot these MFMAs have just read
        if constexpr (QVC_WN2_SYNC > 0 && (k + 1) % (QVC_WN2_SYNC > 0 ? QVC_WN2_SYNC : 1) == 0 && k + 1 < NK1) __builtin_amdgcn_s_barrier();
      };
      static_for<NK1>(step);
#ifdef QVC_STAMP
      if (l < 4) QVC_ST(2 + 6 * l);
#endif
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int j = n * 16 + lrow;
        const f32x4 t = acc[0][n], sg = acc[1][n];
        quad o;
        if (ch0 < a.H) {
          o[0] = O::cvt(fast_tanh(t[0] + bt.x) * fast_sigmoid(sg[0] + bs.x));
          o[1] = O::cvt(fast_tanh(t[1] + bt.y) * fast_sigmoid(sg[1] + bs.y));
          o[2] = O::cvt(fast_tanh(t[2] + bt.z) * fast_sigmoid(sg[2] + bs.z));
          o[3] = O::cvt(fast_tanh(t[3] + bt.w) * fast_sigmoid(sg[3] + bs.w));
        } else {
          o[0] = o[1] = o[2] = o[3] = (T)0.f;
        }
        *reinterpret_cast<quad*>(acts + j * RB + ((rotc(ch0 >> 3, sm) ^ swz(j, sm)) << 4) + (ch0 & 7) * 2) = o;
      }
    }
    // res / skip biases: requested before the barrier, used after GEMM2
    // (unconditional loads from clamped addresses: a branch around a load makes the compiler's wait counts assume the path
    //  without it, and GEMM2's k-steps then wait for the prefetches issued one step earlier)
    const float* brs = a.b_rs[l] + (ch0 < a.H ? ch0 : 0);
    const float4 b0 = *reinterpret_cast<const float4*>(brs);
    const float4 b1 = *reinterpret_cast<const float4*>(brs + (last ? 0 : a.H));
#ifdef QVC_STAMP
    if (l < 4) QVC_ST(3 + 6 * l);
#endif
    __syncthreads();                          // acts complete; every wave is done reading the x tile
#ifdef QVC_STAMP
    if (l < 4) QVC_ST(4 + 6 * l);
#endif
    {   // ---- GEMM2 (1x1): x += res, out += skip   (modules.py:104-112)
      f32x4 acc[2][NF];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      frag bf[2][NF];
      auto read_b = [&](auto k_c, frag (&dst)[NF]) {
        constexpr int ks = decltype(k_c)::value;
        const char* bp = smem + ((ks & 1) ? (boff[0] ^ 64) : boff[0]) + (ACTS + (ks >> 1) * 128);
#pragma unroll
        for (int n = 0; n < NF; ++n) dst[n] = *reinterpret_cast<const frag*>(bp + n * 16 * RB);
      };
      read_b(std::integral_constant<int, 0>{}, bf[0]);
      auto step = [&](auto k_c) {
        constexpr int k = decltype(k_c)::value;       // k-step of GEMM2; stream position NK1 + k
        if constexpr (k + 1 < NK2) { if (!QVC_ABL(6)) read_b(std::integral_constant<int, k + 1>{}, bf[(k + 1) & 1]); }
        __builtin_amdgcn_sched_barrier(0);
        if (!last) {
#pragma unroll
          for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m][n] = O::mfma(ring[(NK1 + k) % RING][m], bf[k & 1][n], acc[m][n]);
        } else {
#pragma unroll
          for (int n = 0; n < NF; ++n) acc[0][n] = O::mfma(ring[(NK1 + k) % RING][0], bf[k & 1][n], acc[0][n]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!QVC_ABL(5)) prefetch(std::integral_constant<int, NK1 + k + PF>{});
      };
      static_for<NK2>(step);
#ifdef QVC_STAMP
      if (l < 4) QVC_ST(5 + 6 * l);
#endif
      if (ch0 < a.H) {
        if (!last) {
          f32x4 xn[NF];
#pragma unroll
          for (int n = 0; n < NF; ++n) {
            const int q = w0 + n * 16 + lrow;
            const bool in = q >= Tlo && q < Tb;                  // the convs zero-pad x outside the utterance
            const float4 xo = xres[n * NTH];
            float4 oo = oacc[n * NTH];
            xn[n][0] = in ? xo.x + acc[0][n][0] + b0.x : 0.f;
            xn[n][1] = in ? xo.y + acc[0][n][1] + b0.y : 0.f;
            xn[n][2] = in ? xo.z + acc[0][n][2] + b0.z : 0.f;
            xn[n][3] = in ? xo.w + acc[0][n][3] + b0.w : 0.f;
            oo.x += acc[1][n][0] + b1.x; oo.y += acc[1][n][1] + b1.y; oo.z += acc[1][n][2] + b1.z; oo.w += acc[1][n][3] + b1.w;
            xres[n * NTH] = make_float4(xn[n][0], xn[n][1], xn[n][2], xn[n][3]);
            oacc[n * NTH] = oo;
          }
          put_x(xn);                          // safe: all waves are past GEMM1 of this layer
        } else {
#pragma unroll
          for (int n = 0; n < NF; ++n) {
            float4 oo = oacc[n * NTH];
            oo.x += acc[0][n][0] + b0.x; oo.y += acc[0][n][1] + b0.y; oo.z += acc[0][n][2] + b0.z; oo.w += acc[0][n][3] + b0.w;
            oacc[n * NTH] = oo;
          }
        }
      }
    }
    ap1 = apn;
#ifdef QVC_STAMP
    if (l < 4) QVC_ST(6 + 6 * l);
#endif
    __syncthreads();                          // x tile updated / acts tile free for the next layer
#ifdef QVC_STAMP
    if (l < 4) QVC_ST(7 + 6 * l);
#endif
  }

  if constexpr (PM > 0) {
    if (a.w_post) {
      // fused `post` 1x1 + coupling update (modules.py:214-217)
#pragma unroll
      for (int n = 0; n < ON; ++n) {
        const int j = n * 16 + lrow;
        quad h;
        if (ch0 < a.H) {
          const float4 oo = oacc[n * NTH];
          h[0] = O::cvt(oo.x); h[1] = O::cvt(oo.y); h[2] = O::cvt(oo.z); h[3] = O::cvt(oo.w);
        } else {
          h[0] = h[1] = h[2] = h[3] = (T)0.f;
        }
        *reinterpret_cast<quad*>(acts + j * RB + ((rotc(ch0 >> 3, sm) ^ swz(j, sm)) << 4) + (ch0 & 7) * 2) = h;
      }
      __syncthreads();
      f32x4 qacc[PM][ON];
#pragma unroll
      for (int m = 0; m < PM; ++m)
#pragma unroll
        for (int n = 0; n < ON; ++n) qacc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      const frag* apq = static_cast<const frag*>(a.w_post) + ((size_t)wm * KS * PM) * 64 + lane;
      gemm_loop<T, PM, ON, QVC_PF_STACK>(qacc, apq, KS, KS, 1, acts, RB, sm, lrow, lq, 0);
#pragma unroll
      for (int m = 0; m < PM; ++m) {
        const int v = (wm * PM + m) * 16 + lq * 4;
        if (v >= a.post_m) continue;
        const float4 bq = *reinterpret_cast<const float4*>(a.b_post + v);
        float4 zin[ON];                      // all loads of the read-modify-write before its first store
#pragma unroll
        for (int n = 0; n < ON; ++n) {
          const int q = w0 + n * 16 + lrow;
          zin[n] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (q >= q0 && q < q0 + OUTF && q < Tb)
            zin[n] = *reinterpret_cast<const float4*>(a.z + (size_t)b * a.z_bs + (size_t)q * a.z_ts + a.post_c0 + v);
        }
#pragma unroll
        for (int n = 0; n < ON; ++n) {
          const int q = w0 + n * 16 + lrow;
          if (q >= q0 && q < q0 + OUTF && q < Tb) {
            float* p = a.z + (size_t)b * a.z_bs + (size_t)q * a.z_ts + a.post_c0 + v;
            float4 zz = zin[n];
            zz.x += a.post_sign * (qacc[m][n][0] + bq.x); zz.y += a.post_sign * (qacc[m][n][1] + bq.y);
            zz.z += a.post_sign * (qacc[m][n][2] + bq.z); zz.w += a.post_sign * (qacc[m][n][3] + bq.w);
            *reinterpret_cast<float4*>(p) = zz;
          }
        }
      }
      return;
    }
  }
  // ---- store the skip sum (and, when another launch continues the stack, the residual stream) of the output tile
  if (ch0 < a.H) {
#pragma unroll
    for (int n = 0; n < ON; ++n) {
      const int q = w0 + n * 16 + lrow;
      if (q >= q0 && q < q0 + OUTF && q < Tb) {
        const size_t off = (size_t)b * a.bs + (size_t)q * a.H + ch0;
        *reinterpret_cast<float4*>(a.out + off) = oacc[n * NTH];
        if (a.x_out) *reinterpret_cast<float4*>(a.x_out + off) = xres[n * NTH];
      }
    }
  }
#ifdef QVC_STAMP
  QVC_ST(26);
  if (st_ && lane == 0) st_[28] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <typename T>
int launch_wn_stack2_typed(const ConvDesc& din, const WnStackArgs& a, int batch, void* stream_v) {
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (!wn2_supported(din, a)) return QVC_ERR_BAD_CONFIG;
  constexpr int KS = 6, TAPS = 5;
  const size_t lds = (size_t)(48 + TAPS - 1 + 48) * KS * 64 + (size_t)2 * 3 * (KS * 2 * 64) * 16;   // tiles + fp32 residual / skip slots
  const dim3 grid((unsigned)ceil_div(a.T, kWnOutFrames), (unsigned)batch), block(KS * 2 * 64);
  if (a.w_post) {
    auto kern = wn_stack2_kernel<T, KS, TAPS, 1>;
    static std::atomic<uint32_t> lds_ok{0};                  // > 64 KiB dynamic LDS: opt in once per device
    if (!allow_big_lds(lds_ok, reinterpret_cast<const void*>(kern))) return QVC_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  } else {
    auto kern = wn_stack2_kernel<T, KS, TAPS, 0>;
    static std::atomic<uint32_t> lds_ok{0};
    if (!allow_big_lds(lds_ok, reinterpret_cast<const void*>(kern))) return QVC_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  }
  return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
}

}  // namespace qvc
