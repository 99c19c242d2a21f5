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
#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include "qvc_kernels.h"
#include "qvc_launch_util.h"

#ifndef QVC_SWZ_ROT
#define QVC_SWZ_ROT 1                  // 0: the round-1 swizzle of 256-byte-multiple rows (A/B builds of tools/conv_bench)
#endif
namespace qvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Debug build only (-DQVC_SATCOUNT, libqvc_hip_sat.so; never defined in the product library): every fp32 -> f16
// conversion that saturates counts itself, so that an activation range the f16 streams cannot carry shows up as a
// number (qvc_debug_saturations) instead of passing silently.  One counter per translation unit (internal linkage).
#ifdef QVC_SATCOUNT
static __device__ unsigned long long g_sat_count = 0;
#define QVC_SAT_READER(name)                                                                        \
  unsigned long long name(bool reset) {                                                             \
    unsigned long long v = 0, z = 0;                                                                \
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_sat_count), 8) != hipSuccess) return ~0ull;            \
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_sat_count), &z, 8) != hipSuccess) return ~0ull;     \
    return v;                                                                                       \
  }
#endif

template <typename T> struct Op;
template <> struct Op<_Float16> {
  using frag = f16x8; using quad = f16x4;
  static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ _Float16 cvt(float f) {
#ifdef QVC_SATCOUNT
    if (__builtin_fabsf(f) > 65504.f) atomicAdd(&g_sat_count, 1ull);
#endif
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
// Gate math on the hardware transcendental units: v_exp_f32 (2^x) and v_rcp_f32 are 1-ulp instructions;
// an IEEE division costs ~10 VALU instructions and made the gate as expensive as the layer's MFMAs.
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); exact limits at +-inf (rcp(inf) = 0), ~2e-7 abs error elsewhere
  return 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.8853900817779268f * x) + 1.f);
}

// 16-byte-chunk swizzle of a tile row: chunk' = rotc(chunk) ^ ((row >> sh) & mask), branch-free; all modes are
// invariant under row += 16.  Rows of 16k chunks (256-byte multiples: every pair / up-sampler / conv_post tile) use
// rotc = the low four chunk bits rotated right by one, XOR (row & 7): a ds_read_b128 lane group is 8 lanes on chunk c
// and 8 lanes on chunk c + 1 of eight different rows each (MI355X_MICROARCH.md, LDS), so the chunk's low bit has to
// pick the 128-byte half and the row's low three bits the slot inside it -- conflict-free for EVERY row offset.
// The round-1 scheme (chunk ^ (row & 15)) was conflict-free for even offsets only: odd tap shifts (dilations 1, 3, 5
// x odd taps, almost half of all B reads) were 2-way, 8 LDS cycles instead of 4 (tools/lds_swizzle_sim.py;
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.28 on the pair kernels).  Rows of 8k chunks keep mask 7 (the WaveNet
// tiles), others (row >> 1) & 3.
struct Swz { int sh, mask, rot; };
__device__ __forceinline__ Swz swz_mode(int cpr) {
  return (cpr & 15) == 0 ? Swz{0, QVC_SWZ_ROT ? 7 : 15, QVC_SWZ_ROT} : ((cpr & 7) == 0 ? Swz{0, 7, 0} : Swz{1, 3, 0});
}
__device__ __forceinline__ int swz(int row, Swz m) { return (row >> m.sh) & m.mask; }
__device__ __forceinline__ int rotc(int chunk, Swz m) {
  return m.rot ? ((chunk & ~15) | ((chunk & 1) << 3) | ((chunk >> 1) & 7)) : chunk;
}

// leaky ReLU on 8 packed operand values (slope <= 1): max(x, slope*x), in operand arithmetic
template <typename T>
__device__ __forceinline__ typename Op<T>::frag lrelu8(typename Op<T>::frag v, float slope) {
  typename Op<T>::frag r;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float f = (float)v[i];
    r[i] = (T)(f > 0.f ? f : f * slope);     // exact in fp32, one rounding back to T
  }
  return r;
}

// f16: max(x, slope*x) with the product formed in fp32 (v_fma_mix, exact as above) and a packed max: 12 VALU
// instructions per 16-byte chunk instead of ~45 for the compare / select form.
template <>
__device__ __forceinline__ f16x8 lrelu8<_Float16>(f16x8 v, float slope) {
  f16x8 s;
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = (_Float16)((float)v[i] * slope);
  return __builtin_elementwise_max(v, s);
}

// A fragments are prefetched PF k-steps ahead through a register ring of PF+1 slots.  Bytes in flight per
// wave = PF * MF KiB: the weight stream of a layer is cold (touched once per step), so small-tile kernels
// (WaveNet layer) need a deep ring to cover the L2/MALL round trip; big-tile kernels are MFMA-paced.
#ifndef QVC_PF_CONV
#define QVC_PF_CONV 3
#endif
// k-step rotation of a workgroup (see gemm_loop); QVC_NO_ROT=1 disables it for A/B measurements
// (measured: no effect on gfx950 -- the L2 is not the limiter -- so it is off; QVC_ROTATE=1 enables it)
#ifdef QVC_ROTATE
#define QVC_ROT(n) ((int)((blockIdx.x * 5u + blockIdx.y * 3u + blockIdx.z) % (unsigned)(n)))
#else
#define QVC_ROT(n) 0
#endif
// developer ablation switches for tools/conv_bench (never defined in the product build)
#ifdef QVC_ABLATE
#define QVC_ABL(bit) ((QVC_ABLATE >> (bit)) & 1)
#else
#define QVC_ABL(bit) 0
#endif
// developer phase stamps for tools/conv_bench (never defined in the product build): s_memtime at the phase boundaries of
// the WaveNet stack kernel, written to WnStackArgs::stamps [workgroup][wave][32] at the end of the kernel
#ifdef QVC_STAMP
#define QVC_ST(slot) do { if (st_ && lane == 0) st_[slot] = __builtin_amdgcn_s_memtime(); } while (0)   // straight to memory: a local array would live in scratch
#else
#define QVC_ST(slot) ((void)0)
#endif
#ifndef QVC_PF_WN
#define QVC_PF_WN 3
#endif
// prefetch depth of the whole-stack WaveNet kernel (weight-stream bound: see DESIGN.md)
#ifndef QVC_PF_STACK
#define QVC_PF_STACK 3
#endif

// gemm_prime issues the first kPF k-steps of a stream into the ring; gemm_loop_primed runs the K loop on a ring
// primed that way.  (Priming the NEXT GEMM's ring before the epilogue and barrier of the current one was tried
// in the WaveNet stack kernel -- 8 short GEMMs per launch -- and bought nothing: 14.3 vs 14.1 us per layer.)
// AS = fragments per k-step in the stream (default MF: the wave walks all of its packed fragments; a wave that
// skips all-padding fragments of its rows walks the first MF of AS)
template <typename T, int MF, int kPF, int AS = MF>
__device__ __forceinline__ void gemm_prime(typename Op<T>::frag (&ar)[kPF + 1][MF], const typename Op<T>::frag* ap, int nIt) {
#pragma unroll
  for (int u = 0; u < kPF; ++u)
    if (u < nIt) {
#pragma unroll
      for (int m = 0; m < MF; ++m) ar[u][m] = ap[((size_t)u * AS + m) * 64];
    }
}

template <typename T, int MF, int NF, int kPF, int AS = MF>
__device__ __forceinline__ void gemm_loop_primed(f32x4 (&acc)[MF][NF], typename Op<T>::frag (&ar)[kPF + 1][MF],
                                                 const typename Op<T>::frag* ap, int nIt, int KS, int dil,
                                                 const char* tile, int rowbytes, Swz sm, int colrow, int lq) {
  // Software pipeline (all register indices static after unrolling RING = kPF+1 steps, RING even):
  //   A fragments: global -> register ring, kPF k-steps ahead (plain loads, hipcc counts them);
  //   B fragments: LDS -> a double buffer, ONE k-step ahead.  Left to itself hipcc keeps two
  //   B buffers and alternates ds_read / wait / MF MFMAs, which exposes the LDS latency on every
  //   fragment (seen in the ISA: ~55 % MFMA issue in the loop).  Here all NF reads of step i+1 are
  //   in flight while the MF*NF MFMAs of step i run; sched_barrier pins that order.
  using O = Op<T>;
  using frag = typename O::frag;
  constexpr int RING = kPF + 1;
  static_assert(RING % 2 == 0, "the B double buffer needs an even ring");
  frag bf[2][NF];
  const int nstride = 16 * rowbytes;
  int pf = kPF < nIt ? kPF : nIt;                               // k-step of the next A prefetch (== nIt: none left)
  int tap = 0, ks = 0;                                          // k-step whose B fragments are read next
  auto read_b = [&](frag (&dst)[NF]) {
    const int row0 = tap * dil + colrow;
    const char* bp = tile + row0 * rowbytes + ((rotc(ks * 4 + lq, sm) ^ swz(row0, sm)) << 4);
#pragma unroll
    for (int n = 0; n < NF; ++n) dst[n] = *reinterpret_cast<const frag*>(bp + n * nstride);
    if (++ks == KS) { ks = 0; ++tap; }
  };
  if (nIt > 0) read_b(bf[0]);
  for (int it0 = 0; it0 < nIt; it0 += RING) {
#pragma unroll
    for (int u = 0; u < RING; ++u) {
      const int it = it0 + u;
      if (it < nIt) {                                           // wave-uniform
        if (it + kPF < nIt && !QVC_ABL(5)) {
#pragma unroll
          for (int m = 0; m < MF; ++m) ar[(u + kPF) % RING][m] = ap[((size_t)pf * AS + m) * 64];
          ++pf;
        }
        if (it + 1 < nIt && !QVC_ABL(6)) read_b(bf[(u + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);                      // loads stay above this step's MFMAs
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
          for (int m = 0; m < MF; ++m) acc[m][n] = O::mfma(ar[u][m], bf[u & 1][n], acc[m][n]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// The K loop shared by the kernels: acc[m][n] += A(stream) x B(LDS tile).  `ap` already points at this
// wave's fragment stream (+lane); B rows start at `colrow` (+ tap*dil); rows are `rowbytes` wide.
// A fragments go through a register ring, prefetched kPF k-steps ahead with plain loads (hipcc tracks
// them).  An inline-asm variant with hand-counted vmcnt waits was tried and rejected: no faster, and
// the compiler copied not-yet-landed asm outputs in one instantiation (wrong results).
template <typename T, int MF, int NF, int kPF, int AS = MF>
__device__ __forceinline__ void gemm_loop(f32x4 (&acc)[MF][NF], const typename Op<T>::frag* ap, int nIt, int KS, int dil,
                                          const char* tile, int rowbytes, Swz sm, int colrow, int lq, int /*rot*/) {
  typename Op<T>::frag ar[kPF + 1][MF];
  gemm_prime<T, MF, kPF, AS>(ar, ap, nIt);
  gemm_loop_primed<T, MF, NF, kPF, AS>(acc, ar, ap, nIt, KS, dil, tile, rowbytes, sm, colrow, lq);
}

// WM waves along M, WN = 4/WM along the frames; block tile = [WM*MF*16 rows] x [WN*NF*16 frames].
// CL ("chunk loop"): the workgroup stages its tile once and walks all the row chunks itself (grid z = 1) instead of
// one workgroup per chunk each staging the same tile -- for convs whose staging is the expensive part (up-sampler 1:
// the mean of three streams) and that still leave >= 2 workgroups per CU.
template <typename T, int MF, int NF, int WM, int EPI, bool CL = false>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
  using O = Op<T>;
  using frag = typename O::frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WN = kWaves / WM;
  constexpr int NT = WN * NF * 16;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * NT;
  const int R = NT + (a.taps - 1) * a.dil;
  const int rowbytes = a.CinP * 2;
  const int cpr = a.CinP >> 3;                                  // 16-byte chunks per row
  const Swz sm = swz_mode(cpr);
  const int t_base = q0 - a.left;                               // input frame of tile row 0
  const int Tin = ragged_len(a.rg, b, a.T_in);                  // this utterance's input rows are [Tlo, Tin) (ragged batches,
  const int Tlo = ragged_lo(a.rg, b);                           // streaming windows); rows outside read as the conv's zero padding
  if (a.rg.lens && t_base >= Tin + a.reflect) return;           // tile past the end of the utterance: nothing anybody reads

  // ------------------------------------------------------------------ stage the activation tile
  // Loads are issued in batches per thread before anything is converted or stored, so that one
  // workgroup alone keeps tens of KiB of reads in flight.
  constexpr int kU = 4;
  if (a.x_kind == XK_F32_FM) {
    const float* xb = static_cast<const float*>(a.x) + (size_t)b * a.x_bs + a.x_c0;
    const float slope = a.slope_in;
    const int total = R * cpr;
    for (int base = tid; base < total; base += 256 * kU) {
      float4 v0[kU], v1[kU];
      int dst[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int idx = base + u * 256;
        const int r = idx / cpr, c8 = idx - r * cpr;
        const int ti = t_base + r;
        bool ok; int src;
        if (a.reflect) { ok = ti >= Tlo && ti <= Tin; src = ti == Tlo ? Tlo + 1 : ti - 1; }
        else { ok = ti >= Tlo && ti < Tin; src = ti; }
        ok = ok && idx < total && (c8 * 8 < a.Cin);
        v0[u] = make_float4(0.f, 0.f, 0.f, 0.f); v1[u] = v0[u];
        if (ok) {
          const float4* p = reinterpret_cast<const float4*>(xb + (size_t)src * a.x_ts + c8 * 8);
          v0[u] = p[0]; v1[u] = p[1];
        }
        dst[u] = idx < total ? r * rowbytes + ((rotc(c8, sm) ^ swz(r, sm)) << 4) : -1;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (dst[u] < 0) continue;
        frag h;
        h[0] = O::cvt(lrelu(v0[u].x, slope)); h[1] = O::cvt(lrelu(v0[u].y, slope));
        h[2] = O::cvt(lrelu(v0[u].z, slope)); h[3] = O::cvt(lrelu(v0[u].w, slope));
        h[4] = O::cvt(lrelu(v1[u].x, slope)); h[5] = O::cvt(lrelu(v1[u].y, slope));
        h[6] = O::cvt(lrelu(v1[u].z, slope)); h[7] = O::cvt(lrelu(v1[u].w, slope));
        *reinterpret_cast<frag*>(smem + dst[u]) = h;
      }
    }
  } else if (a.x_kind == XK_OP_FM && a.x2) {
    // mean of three operand-type tensors (MRF average taken by the consumer), then the activation
    const size_t boff = (size_t)b * a.x_bs + a.x_c0;
    const T* xb = static_cast<const T*>(a.x) + boff;
    const T* xb2 = static_cast<const T*>(a.x2) + boff;
    const T* xb3 = static_cast<const T*>(a.x3) + boff;
    const float slope = a.slope_in;
    const int total = R * cpr;
    for (int base = tid; base < total; base += 256 * kU) {
      uint4 v1[kU], v2[kU], v3[kU];
      int dst[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int idx = base + u * 256;
        const int r = idx / cpr, c8 = idx - r * cpr;
        const int ti = t_base + r;
        bool ok; int src;
        if (a.reflect) { ok = ti >= Tlo && ti <= Tin; src = ti == Tlo ? Tlo + 1 : ti - 1; }
        else { ok = ti >= Tlo && ti < Tin; src = ti; }
        ok = ok && idx < total && (c8 * 8 < a.Cin);
        v1[u] = make_uint4(0u, 0u, 0u, 0u); v2[u] = v1[u]; v3[u] = v1[u];
        if (ok) {
          const size_t o = (size_t)src * a.x_ts + c8 * 8;
          v1[u] = *reinterpret_cast<const uint4*>(xb + o);
          v2[u] = *reinterpret_cast<const uint4*>(xb2 + o);
          v3[u] = *reinterpret_cast<const uint4*>(xb3 + o);
        }
        dst[u] = idx < total ? r * rowbytes + ((rotc(c8, sm) ^ swz(r, sm)) << 4) : -1;
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
  } else if (a.x_kind == XK_OP_FM) {
    const T* xb = static_cast<const T*>(a.x) + (size_t)b * a.x_bs + a.x_c0;
    const int total = R * cpr;
    for (int base = tid; base < total; base += 256 * kU * 2) {
      uint4 v[kU * 2];
      int dst[kU * 2];
#pragma unroll
      for (int u = 0; u < kU * 2; ++u) {
        const int idx = base + u * 256;
        const int r = idx / cpr, c8 = idx - r * cpr;
        const int ti = t_base + r;
        const bool ok = idx < total && ti >= Tlo && ti < Tin && (c8 * 8 < a.Cin);
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        if (ok) v[u] = *reinterpret_cast<const uint4*>(xb + (size_t)ti * a.x_ts + c8 * 8);
        dst[u] = idx < total ? r * rowbytes + ((rotc(c8, sm) ^ swz(r, sm)) << 4) : -1;
      }
#pragma unroll
      for (int u = 0; u < kU * 2; ++u)
        if (dst[u] >= 0) {
          if (a.slope_in != 1.f) {
            frag h; __builtin_memcpy(&h, &v[u], 16);
            h = lrelu8<T>(h, a.slope_in);
            *reinterpret_cast<frag*>(smem + dst[u]) = h;
          } else {
            *reinterpret_cast<uint4*>(smem + dst[u]) = v[u];
          }
        }
    }
  } else {  // XK_F32_CM: (B, C, T) -- consecutive threads walk along T (coalesced), transposed into the tile
    // (32 scalar loads in flight per thread: the 32-frame x 256-channel tile of enc_p.pre in ONE memory round trip --
    //  with 8 it took four dependent ones on an input that comes cold from HBM)
    constexpr int kCM = 32;
    const float* xb = static_cast<const float*>(a.x) + (size_t)b * a.x_bs;
    const float slope = a.slope_in;
    const int total = R * a.CinP;
    for (int base = tid; base < total; base += 256 * kCM) {
      float v[kCM];
      int dst[kCM];
#pragma unroll
      for (int u = 0; u < kCM; ++u) {
        const int idx = base + u * 256;
        const int c = idx / R, r = idx - c * R;
        const int ti = t_base + r;
        v[u] = 0.f;
        if (idx < total && c < a.Cin && ti >= Tlo && ti < Tin) v[u] = xb[(size_t)c * a.x_ts + ti];
        dst[u] = idx < total ? r * rowbytes + ((rotc(c >> 3, sm) ^ swz(r, sm)) << 4) + (c & 7) * 2 : -1;
      }
#pragma unroll
      for (int u = 0; u < kCM; ++u)
        if (dst[u] >= 0) *reinterpret_cast<T*>(smem + dst[u]) = O::cvt(lrelu(v[u], slope));
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ K loop: A from global (ring), B from LDS
  const int lrow = lane & 15, lq = lane >> 4;
  const int chunk_lo = CL ? 0 : (int)blockIdx.z, chunk_hi = CL ? a.nchunk : chunk_lo + 1;
  for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
  f32x4 acc[MF][NF];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const frag* ap = static_cast<const frag*>(a.w) + ((size_t)(chunk * WM + wm) * a.nIt * MF) * 64 + lane;
  // Polyphase rows: phase ph of a transposed conv (kernel k, stride s) has ceil((k - ph) / s) real taps; the packed
  // stream pads every phase to `taps` with zero weights, and with left = taps - 1 those zeros are the FIRST taps.
  // This wave's rows start in phase ph0 (the phase with the most taps among them): skip the k-steps all of its rows
  // have zeros in (k 16, s 5: 3 real taps out of 4 for four of the five phases -- a fifth of the up-sampler's work).
  int it0 = 0;
  if (a.up_s > 1 && a.ksize > 0) {
    const int ph0 = ((chunk * WM + wm) * MF * 16) / a.Cout;
    const int real = ph0 < a.up_s ? (a.ksize - ph0 + a.up_s - 1) / a.up_s : 0;
    const int tap0 = a.taps - (real < a.taps ? real : a.taps);
    it0 = tap0 * a.KS;
  }
  gemm_loop<T, MF, NF, QVC_PF_CONV>(acc, ap + (size_t)it0 * MF * 64, a.nIt - it0, a.KS, a.dil, smem, rowbytes, sm,
                                    wn * (NF * 16) + lrow + (it0 / a.KS) * a.dil, lq, QVC_ROT(a.nIt));

  // ------------------------------------------------------------------ epilogue
  const int qw = q0 + wn * (NF * 16);                           // first frame of this wave's columns
  if constexpr (EPI == EPI_GAU) {
    // fragments [0, MF/2) = tanh rows, [MF/2, MF) = sigmoid rows of the same channels
    static_assert(EPI != EPI_GAU || MF % 2 == 0, "GAU epilogue pairs fragments");
    constexpr int HF = MF / 2;
    const int H = a.gau_H;
    const float* bb = a.bbias + (size_t)b * a.bbias_bs;
#pragma unroll
    for (int f = 0; f < HF; ++f) {
      const int ch0 = ((chunk * WM + wm) * HF + f) * 16 + lq * 4;
      if (ch0 >= H) continue;
      const float4 bt = *reinterpret_cast<const float4*>(bb + ch0);
      const float4 bs = *reinterpret_cast<const float4*>(bb + H + ch0);
      T* yb = static_cast<T*>(a.y16) + (size_t)b * a.y16_bs + ch0;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int q = qw + n * 16 + lrow;
        if (q < a.Nq) {
          const f32x4 t = acc[f][n], s = acc[HF + f][n];
          typename O::quad o;
          o[0] = O::cvt(fast_tanh(t[0] + bt.x) * fast_sigmoid(s[0] + bs.x));
          o[1] = O::cvt(fast_tanh(t[1] + bt.y) * fast_sigmoid(s[1] + bs.y));
          o[2] = O::cvt(fast_tanh(t[2] + bt.z) * fast_sigmoid(s[2] + bs.z));
          o[3] = O::cvt(fast_tanh(t[3] + bt.w) * fast_sigmoid(s[3] + bs.w));
          *reinterpret_cast<typename O::quad*>(yb + (size_t)q * a.y16_ts) = o;
        }
      }
    }
  } else if constexpr (EPI == EPI_SAMPLE) {
    // fragments [0, MF/2) = mu rows, [MF/2, MF) = log-sigma rows of the same channels: sample in registers
    static_assert(EPI != EPI_SAMPLE || MF % 2 == 0, "paired rows");
    constexpr int HF = MF / 2;
    const int C = a.gau_H;
#pragma unroll
    for (int f = 0; f < HF; ++f) {
      const int ch0 = ((chunk * WM + wm) * HF + f) * 16 + lq * 4;
      if (ch0 >= C) continue;
      const float4 bm = *reinterpret_cast<const float4*>(a.bias + ch0);
      const float4 bl = *reinterpret_cast<const float4*>(a.bias + C + ch0);
      const float* nb = a.noise + (size_t)b * a.noise_bs + (size_t)ch0 * a.noise_ts;
      float* zb = a.y32 + (size_t)b * a.y32_bs + ch0;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int q = qw + n * 16 + lrow;
        if (q < a.Nq) {
          const f32x4 mu = acc[f][n], ls = acc[HF + f][n];
          const float n0 = nb[q], n1 = nb[(size_t)a.noise_ts + q], n2 = nb[2 * (size_t)a.noise_ts + q], n3 = nb[3 * (size_t)a.noise_ts + q];
          *reinterpret_cast<float4*>(zb + (size_t)q * a.y32_ts) =
              make_float4((mu[0] + bm.x) + n0 * expf(ls[0] + bl.x), (mu[1] + bm.y) + n1 * expf(ls[1] + bl.y),
                          (mu[2] + bm.z) + n2 * expf(ls[2] + bl.z), (mu[3] + bm.w) + n3 * expf(ls[3] + bl.w));
        }
      }
    }
  } else {
  bool packed = false;
  if constexpr (MF % 2 == 0) if (a.lp) {
    packed = true;
    // lane-packed rows (ConvDesc::lp; only a y16 output -- checked by launch_conv): the lane's MF quads are the
    // 4*MF consecutive virtual rows v0.., all of one phase, stored as MF/2 16-byte pieces per frame
    const int v0 = (chunk * WM + wm) * MF * 16 + lq * 4 * MF;
    if (v0 < a.M) {
      int ph = 0, co = v0;
      if (a.up_s > 1) { ph = v0 / a.Cout; co = v0 - ph * a.Cout; }
      float4 bias[MF];
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        bias[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) bias[m] = *reinterpret_cast<const float4*>(a.bias + v0 + m * 4);
        if (a.bbias) {
          const float4 bb = *reinterpret_cast<const float4*>(a.bbias + (size_t)b * a.bbias_bs + v0 + m * 4);
          bias[m].x += bb.x; bias[m].y += bb.y; bias[m].z += bb.z; bias[m].w += bb.w;
        }
      }
      T* yb = static_cast<T*>(a.y16) + (size_t)b * a.y16_bs + co;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int q = qw + n * 16 + lrow;
        const int o = q * a.up_s + ph - a.up_p;
        if (q >= a.Nq || o < 0 || o >= a.T_out) continue;
        frag h[MF / 2];
#pragma unroll
        for (int m = 0; m < MF; ++m) {
          h[m >> 1][(m & 1) * 4 + 0] = O::cvt(lrelu(acc[m][n][0] + bias[m].x, a.slope_out));
          h[m >> 1][(m & 1) * 4 + 1] = O::cvt(lrelu(acc[m][n][1] + bias[m].y, a.slope_out));
          h[m >> 1][(m & 1) * 4 + 2] = O::cvt(lrelu(acc[m][n][2] + bias[m].z, a.slope_out));
          h[m >> 1][(m & 1) * 4 + 3] = O::cvt(lrelu(acc[m][n][3] + bias[m].w, a.slope_out));
        }
#pragma unroll
        for (int k2 = 0; k2 < MF / 2; ++k2) *reinterpret_cast<frag*>(yb + (size_t)o * a.y16_ts + k2 * 8) = h[k2];
      }
    }
  }
  if (!packed) {
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int v = ((chunk * WM + wm) * MF + m) * 16 + lq * 4;
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
        const int q = qw + n * 16 + lrow;
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
        if (a.res16) {
          const typename O::quad rr = *reinterpret_cast<const typename O::quad*>(
              static_cast<const T*>(a.res16) + (size_t)b * a.res_bs + (size_t)o * a.res_ts + co);
          val.x = (float)rr[0] + a.res_sign * val.x; val.y = (float)rr[1] + a.res_sign * val.y;
          val.z = (float)rr[2] + a.res_sign * val.z; val.w = (float)rr[3] + a.res_sign * val.w;
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
  }   // chunk
}


// ------------------------------------------------------------------ fused ResBlock1 pair
// y = x + conv2(lrelu(conv1(lrelu(x))))  (modules.py:148-153) in ONE kernel.  The activated input tile
// is staged once; GEMM1 produces the intermediate for NT + 2*h2 frames (its own 'same' halo), which
// after bias + leaky-ReLU OVERWRITES the input tile in LDS (the input is dead by then), GEMM2 runs
// from there, and the epilogue adds the raw residual and stores the operand-type result.  Compared with two launches this removes the
// intermediate's HBM round trip and one staging pass; the price is NF+1 instead of NF column
// fragments in GEMM1.  The residual stream is carried in the operand type (costs 0.35 dB).
// NWV = waves per workgroup (4, or 8 with all of them along M: see wide_pair_layout in qvc_plan.h)
//
// One launch carries pair q of up to three independent ResBlock chains (PairArgs3): workgroup x works for chain
// x % n.  The chains differ in kernel size, so the workgroups sharing a CU have different durations and drift out
// of phase -- the memory phases (staging, epilogue) of one run under the GEMM phases of another.
//
// Rows are lane-packed (ConvDesc::lp): a lane's MF accumulator quads are 4*MF consecutive channels, so with an
// even MF the intermediate goes to LDS and the residual / result move through memory in 16-byte pieces.
// TS = type of the residual STREAM in memory (x, y); T = MFMA operand type.  They differ in the mixed mode
// (QVC_BF16X: bf16 operands, f16 stream): the identity path of 18 chained pairs then carries 11 mantissa bits --
// rounding it to bf16 at every pair is what costs all-bf16 its waveform SNR (measured) -- at the same bytes.
template <typename T, int MF, int NF, int WM, int NWV, typename TS = T>
__global__ __launch_bounds__(NWV * 64) void rbpair_kernel(const PairArgs3 A) {
  using O = Op<T>;
  using frag = typename O::frag;
  using quad = typename O::quad;
  using OS = Op<TS>;
  using sfrag = typename OS::frag;
  using squad = typename OS::quad;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WN = NWV / WM;
  constexpr int NTHR = NWV * 64;
  constexpr int NF1 = NF + 1;
  constexpr int NT = WN * NF * 16;       // output frames per block
  constexpr int N1P = WN * NF1 * 16;     // intermediate frames computed per block (>= NT + 2*h2)
  constexpr bool kWide = MF % 2 == 0;    // 8 consecutive channels per 16-byte piece

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int lrow = lane & 15, lq = lane >> 4;
  const int b = blockIdx.y;
  const int nj = A.n;
  const int chain = nj > 1 ? (A.chain_major ? (int)blockIdx.z : (int)(blockIdx.x % (unsigned)nj)) : 0;
  const int tile = (nj > 1 && !A.chain_major) ? (int)(blockIdx.x / (unsigned)nj) : (int)blockIdx.x;
  const int q0 = tile * NT;
  PairArgs a = A.p[0];                   // scalar selects: a dynamic index into the kernel arguments would go through scratch
  if (chain == 1) a = A.p[1];
  if (chain == 2) a = A.p[2];
  const int h2 = (a.k - 1) / 2, h1 = h2 * a.dil;
  const int Rx = N1P + 2 * h1;           // staged rows; row 0 <-> frame q0 - h2 - h1
  const int rowbytes = a.CP * 2;
  const int cpr = a.CP >> 3;
  const Swz sm = swz_mode(cpr);
  const TS* xb = static_cast<const TS*>(a.x) + (size_t)b * a.bs;
  const int cb = wm * MF * 16 + lq * 4 * MF;   // first of this lane's 4*MF consecutive channels
  const int Tb = ragged_len(A.rg, b, a.T);     // this utterance occupies rows [Tlo, Tb): both convs zero-pad at ITS ends
  const int Tlo = ragged_lo(A.rg, b);
  if (q0 >= Tb) return;                        // tile past the end of the utterance (ragged batches)
#ifdef QVC_STAMP
  // wave 0 of every workgroup: 64 slots; 0-6 the phases, 8 = tile loads issued, 9 = first data back, 10 = converted and written
  unsigned long long* st_ = (A.stamps && wave == 0) ? A.stamps + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64 : nullptr;
#endif
  QVC_ST(0);
#ifdef QVC_STAMP
  if (st_ && lane == 0) st_[11] = __builtin_amdgcn_s_memrealtime();      // 100 MHz: with slots 0 / 6 the clock the chip holds
#endif

  if (!QVC_ABL(0)) {   // ---- stage lrelu(x): every load of the tile is in flight before the first conversion
    // (no accumulator is live yet, so the registers are free: one memory round trip per tile instead of two)
    const int t_base = q0 - h2 - h1;
    const int total = Rx * cpr;
    constexpr int kU = 16;
    const int rstep = NTHR / cpr, cstep = NTHR - rstep * cpr;
    for (int base = tid; base < total; base += NTHR * kU) {
      uint4 v[kU];
      const int r_0 = base / cpr, c_0 = base - r_0 * cpr;
      int r = r_0, c8 = c_0;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int idx = base + u * NTHR;
        const int ti = t_base + r;
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        if (idx < total && ti >= Tlo && ti < Tb && c8 * 8 < a.C) v[u] = *reinterpret_cast<const uint4*>(xb + (size_t)ti * a.C + c8 * 8);
        c8 += cstep; r += rstep;
        if (c8 >= cpr) { c8 -= cpr; ++r; }
      }
      r = r_0; c8 = c_0;
#ifdef QVC_STAMP
      if (base == tid) {
        QVC_ST(8);
        asm volatile("s_waitcnt vmcnt(15)" ::: "memory");      // the first of the 16 loads is back
        QVC_ST(9);
      }
#endif
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int idx = base + u * NTHR;
        if (idx < total) {
          frag o;
          if constexpr (std::is_same<T, TS>::value) {
            frag h; __builtin_memcpy(&h, &v[u], 16);
            o = lrelu8<T>(h, a.slope);
          } else {                                   // stream type -> fp32 -> activation -> operand type
            sfrag h; __builtin_memcpy(&h, &v[u], 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = O::cvt(lrelu((float)h[e], a.slope));
          }
          *reinterpret_cast<frag*>(smem + r * rowbytes + ((rotc(c8, sm) ^ swz(r, sm)) << 4)) = o;
        }
        c8 += cstep; r += rstep;
        if (c8 >= cpr) { c8 -= cpr; ++r; }
      }
    }
  }
  QVC_ST(10);
  __syncthreads();
  QVC_ST(1);

  {   // ---- GEMM1 over N1P frames, then bias + lrelu -> intermediate tile (in place of the input tile)
    f32x4 acc[MF][NF1];
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF1; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const frag* ap = static_cast<const frag*>(a.w1) + ((size_t)wm * a.nIt * MF) * 64 + lane;
    if (!QVC_ABL(1)) gemm_loop<T, MF, NF1, QVC_PF_CONV>(acc, ap, a.nIt, a.KS, a.dil, smem, rowbytes, sm, wn * (NF1 * 16) + lrow, lq, QVC_ROT(a.nIt));
    QVC_ST(2);
    float4 bias[MF];                     // the bias array is padded to WM*MF*16 entries (zeros past C)
#pragma unroll
    for (int m = 0; m < MF; ++m) bias[m] = *reinterpret_cast<const float4*>(a.b1 + cb + m * 4);
    __syncthreads();                     // every wave is done reading the input tile
    QVC_ST(3);
#pragma unroll
    for (int n = 0; n < NF1; ++n) {
      const int jr = wn * (NF1 * 16) + n * 16 + lrow;            // intermediate row <-> frame q0 - h2 + jr
      const int f = q0 - h2 + jr;
      const bool inside = f >= Tlo && f < Tb;                      // conv2 zero-pads outside [0, T)
      char* rowp = smem + jr * rowbytes;
      const int sw = swz(jr, sm);
      if constexpr (kWide) {
#pragma unroll
        for (int m = 0; m < MF; m += 2) {
          const int v = cb + m * 4;                              // channels v .. v+7, one 16-byte chunk
          if (v >= a.CP) continue;
          frag h;
          if (inside && v < a.C) {
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
      } else {
#pragma unroll
        for (int m = 0; m < MF; ++m) {
          const int v = cb + m * 4;
          if (v >= a.CP) continue;
          quad h;
          if (inside && v < a.C) {
            h[0] = O::cvt(lrelu(acc[m][n][0] + bias[m].x, a.slope)); h[1] = O::cvt(lrelu(acc[m][n][1] + bias[m].y, a.slope));
            h[2] = O::cvt(lrelu(acc[m][n][2] + bias[m].z, a.slope)); h[3] = O::cvt(lrelu(acc[m][n][3] + bias[m].w, a.slope));
          } else {
            h[0] = h[1] = h[2] = h[3] = (T)0.f;
          }
          *reinterpret_cast<quad*>(rowp + ((rotc(v >> 3, sm) ^ sw) << 4) + (v & 7) * 2) = h;
        }
      }
    }
  }
  __syncthreads();
  QVC_ST(4);

  {   // ---- GEMM2 over NT frames (dilation 1) + bias + residual
    f32x4 acc[MF][NF];
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const frag* ap = static_cast<const frag*>(a.w2) + ((size_t)wm * a.nIt * MF) * 64 + lane;
    if (!QVC_ABL(2)) gemm_loop<T, MF, NF, QVC_PF_CONV>(acc, ap, a.nIt, a.KS, 1, smem, rowbytes, sm, wn * (NF * 16) + lrow, lq, QVC_ROT(a.nIt));
    QVC_ST(5);
    float4 bias[MF];
#pragma unroll
    for (int m = 0; m < MF; ++m) bias[m] = *reinterpret_cast<const float4*>(a.b2 + cb + m * 4);
    const TS* xres = static_cast<const TS*>(a.x) + (size_t)b * a.bs;
    TS* yb = static_cast<TS*>(a.y) + (size_t)b * a.bs;
    const int qw = q0 + wn * (NF * 16) + lrow;
    // All residual loads go out before the first store: x and y may alias as far as the compiler knows, so a
    // load-add-store per fragment compiles to load, s_waitcnt vmcnt(0), store -- MF*NF serialised memory round
    // trips at the tail of every workgroup (seen in the ISA; it was a third of the launch time).
    if constexpr (kWide) {
      uint4 rr[MF / 2][NF];
#pragma unroll
      for (int m = 0; m < MF; m += 2) {
        const int v = cb + m * 4;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
          const int q = qw + n * 16;
          const bool ok = v < a.C && q < Tb && !QVC_ABL(3);
          rr[m / 2][n] = *reinterpret_cast<const uint4*>(xres + (size_t)(ok ? q : 0) * a.C + (ok ? v : 0));
        }
      }
#pragma unroll
      for (int m = 0; m < MF; m += 2) {
        const int v = cb + m * 4;
        if (v >= a.C || (QVC_ABL(3) && acc[0][0][0] != 12345.f)) continue;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
          const int q = qw + n * 16;
          if (q >= Tb) continue;
          sfrag r8; __builtin_memcpy(&r8, &rr[m / 2][n], 16);
          sfrag h;
          h[0] = OS::cvt(acc[m][n][0] + bias[m].x + (float)r8[0]); h[1] = OS::cvt(acc[m][n][1] + bias[m].y + (float)r8[1]);
          h[2] = OS::cvt(acc[m][n][2] + bias[m].z + (float)r8[2]); h[3] = OS::cvt(acc[m][n][3] + bias[m].w + (float)r8[3]);
          h[4] = OS::cvt(acc[m + 1][n][0] + bias[m + 1].x + (float)r8[4]); h[5] = OS::cvt(acc[m + 1][n][1] + bias[m + 1].y + (float)r8[5]);
          h[6] = OS::cvt(acc[m + 1][n][2] + bias[m + 1].z + (float)r8[6]); h[7] = OS::cvt(acc[m + 1][n][3] + bias[m + 1].w + (float)r8[7]);
          *reinterpret_cast<sfrag*>(yb + (size_t)q * a.C + v) = h;
        }
      }
    } else {
      squad rr[MF][NF];
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        const int v = cb + m * 4;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
          const int q = qw + n * 16;
          const bool ok = v < a.C && q < Tb && !QVC_ABL(3);
          rr[m][n] = *reinterpret_cast<const squad*>(xres + (size_t)(ok ? q : 0) * a.C + (ok ? v : 0));
        }
      }
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        const int v = cb + m * 4;
        if (v >= a.C || (QVC_ABL(3) && acc[0][0][0] != 12345.f)) continue;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
          const int q = qw + n * 16;
          if (q >= Tb) continue;
          const squad r4 = rr[m][n];
          squad h;
          h[0] = OS::cvt(acc[m][n][0] + bias[m].x + (float)r4[0]); h[1] = OS::cvt(acc[m][n][1] + bias[m].y + (float)r4[1]);
          h[2] = OS::cvt(acc[m][n][2] + bias[m].z + (float)r4[2]); h[3] = OS::cvt(acc[m][n][3] + bias[m].w + (float)r4[3]);
          *reinterpret_cast<squad*>(yb + (size_t)q * a.C + v) = h;
        }
      }
    }
#ifdef QVC_STAMP
    __builtin_amdgcn_s_waitcnt(0);         // the stores have left the wave's queue (vmcnt covers stores on this target)
#endif
    QVC_ST(6);
#ifdef QVC_STAMP
    if (st_ && lane == 0) st_[12] = __builtin_amdgcn_s_memrealtime();
#endif
  }
}


// ------------------------------------------------------------------ fused WaveNet layer
// One wave per 16 channels (blockDim = HP/16 waves <= WV, see wn_layout in qvc_plan.h): wave w holds the tanh,
// sigmoid, res and skip rows of channels [16w, 16w+16), so one workgroup owns ALL 2h gate rows of its NF*16
// frames, gates them into an LDS tile and runs the 1x1 res/skip GEMM from there: one launch per layer, the
// gated activations never leave the CU.
template <typename T, int NF, bool LAST, int WV>
__global__ __launch_bounds__(WV * 64) void wn_layer_kernel(const WnArgs a) {
  using O = Op<T>;
  using frag = typename O::frag;
  using quad = typename O::quad;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int FW = 1;
  constexpr int NT = NF * 16;
  constexpr int MF1 = 2 * FW, MF2 = LAST ? FW : 2 * FW;

  const int tid = threadIdx.x, lane = tid & 63, NTH = blockDim.x;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lq = lane >> 4;
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * NT;
  const int halo = a.taps - 1, left = halo / 2;
  const int R = NT + halo;
  const int rowbytes = a.HP * 2;
  const int cpr = a.HP >> 3;
  const Swz sm = swz_mode(cpr);
  char* acts = smem + R * rowbytes;                              // second tile: NT rows of gated activations
  const float* xb = a.x_in + (size_t)b * a.bs;
  const int Tb = ragged_len(a.rg, b, a.T), Tlo = ragged_lo(a.rg, b);
  if (q0 >= Tb) return;

  if (!QVC_ABL(0)) {   // ---- stage x (fp32 -> operand type), rows [q0-left, q0-left+R)
    const int total = R * cpr;
    constexpr int kU = 4;
    for (int base = tid; base < total; base += NTH * kU) {
      float4 v0[kU], v1[kU];
      int dst[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int idx = base + u * NTH;
        const int r = idx / cpr, c8 = idx - r * cpr;
        const int ti = q0 - left + r;
        const bool ok = idx < total && ti >= Tlo && ti < Tb && (c8 * 8 < a.H);
        v0[u] = make_float4(0.f, 0.f, 0.f, 0.f); v1[u] = v0[u];
        if (ok) {
          const float4* p = reinterpret_cast<const float4*>(xb + (size_t)ti * a.H + c8 * 8);
          v0[u] = p[0]; v1[u] = p[1];
        }
        dst[u] = idx < total ? r * rowbytes + ((rotc(c8, sm) ^ swz(r, sm)) << 4) : -1;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (dst[u] < 0) continue;
        frag h;
        h[0] = O::cvt(v0[u].x); h[1] = O::cvt(v0[u].y); h[2] = O::cvt(v0[u].z); h[3] = O::cvt(v0[u].w);
        h[4] = O::cvt(v1[u].x); h[5] = O::cvt(v1[u].y); h[6] = O::cvt(v1[u].z); h[7] = O::cvt(v1[u].w);
        *reinterpret_cast<frag*>(smem + dst[u]) = h;
      }
    }
  }
  __syncthreads();

  {   // ---- GEMM1 (k taps) + conditioning + gate -> acts tile
    f32x4 acc[MF1][NF];
#pragma unroll
    for (int m = 0; m < MF1; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const frag* ap = static_cast<const frag*>(a.w_in) + ((size_t)wm * a.nIt1 * MF1) * 64 + lane;
    if (!QVC_ABL(1)) gemm_loop<T, MF1, NF, QVC_PF_WN>(acc, ap, a.nIt1, a.KS, 1, smem, rowbytes, sm, lrow, lq, QVC_ROT(a.nIt1));
    const float* bb = a.bbias + (size_t)b * a.bbias_bs;
#pragma unroll
    for (int f = 0; f < FW; ++f) {
      const int ch0 = (wm * FW + f) * 16 + lq * 4;
      if (ch0 >= a.HP) continue;
      float4 bt = make_float4(0.f, 0.f, 0.f, 0.f), bs = bt;
      if (ch0 < a.H) { bt = *reinterpret_cast<const float4*>(bb + ch0); bs = *reinterpret_cast<const float4*>(bb + a.H + ch0); }
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int j = n * 16 + lrow;
        const f32x4 t = acc[f][n], sg = acc[FW + f][n];
        quad o;
        if (ch0 < a.H) {
          o[0] = O::cvt(fast_tanh(t[0] + bt.x) * fast_sigmoid(sg[0] + bs.x));
          o[1] = O::cvt(fast_tanh(t[1] + bt.y) * fast_sigmoid(sg[1] + bs.y));
          o[2] = O::cvt(fast_tanh(t[2] + bt.z) * fast_sigmoid(sg[2] + bs.z));
          o[3] = O::cvt(fast_tanh(t[3] + bt.w) * fast_sigmoid(sg[3] + bs.w));
        } else {
          o[0] = o[1] = o[2] = o[3] = (T)0.f;                    // K padding of the 1x1 must be finite
        }
        *reinterpret_cast<quad*>(acts + j * rowbytes + ((rotc(ch0 >> 3, sm) ^ swz(j, sm)) << 4) + (ch0 & 7) * 2) = o;
      }
    }
  }
  __syncthreads();

  {   // ---- GEMM2 (1x1) + residual / skip updates
    f32x4 acc[MF2][NF];
#pragma unroll
    for (int m = 0; m < MF2; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const frag* ap = static_cast<const frag*>(a.w_rs) + ((size_t)wm * a.KS * MF2) * 64 + lane;
    if (!QVC_ABL(2)) gemm_loop<T, MF2, NF, QVC_PF_WN>(acc, ap, a.KS, a.KS, 1, acts, rowbytes, sm, lrow, lq, QVC_ROT(a.KS));
    // all loads of the epilogue are issued before the first dependent store (they are independent
    // L2 round trips; issued one by one they cost ~5 us per layer)
    float4 xin[FW][NF], oin[FW][NF];
#pragma unroll
    for (int f = 0; f < FW; ++f) {
      const int ch0 = (wm * FW + f) * 16 + lq * 4;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int q = q0 + n * 16 + lrow;
        const bool ok = ch0 < a.H && q < Tb && !QVC_ABL(3);
        const size_t off = (size_t)b * a.bs + (size_t)q * a.H + ch0;
        xin[f][n] = make_float4(0.f, 0.f, 0.f, 0.f); oin[f][n] = xin[f][n];
        if (ok) {
          if constexpr (!LAST) xin[f][n] = *reinterpret_cast<const float4*>(a.x_in + off);
          oin[f][n] = *reinterpret_cast<const float4*>(a.oacc + off);
        }
      }
    }
#pragma unroll
    for (int f = 0; f < FW; ++f) {
      const int ch0 = (wm * FW + f) * 16 + lq * 4;
      if (ch0 >= a.H || QVC_ABL(3)) continue;
      const float4 b0 = *reinterpret_cast<const float4*>(a.b_rs + ch0);
      float4 b1 = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (!LAST) b1 = *reinterpret_cast<const float4*>(a.b_rs + a.H + ch0);
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int q = q0 + n * 16 + lrow;
        if (q >= Tb) continue;
        const size_t off = (size_t)b * a.bs + (size_t)q * a.H + ch0;
        float4 sk;
        if constexpr (!LAST) {
          const float4 xi = xin[f][n];
          *reinterpret_cast<float4*>(a.x_out + off) = make_float4(xi.x + acc[f][n][0] + b0.x, xi.y + acc[f][n][1] + b0.y,
                                                                  xi.z + acc[f][n][2] + b0.z, xi.w + acc[f][n][3] + b0.w);
          sk = make_float4(acc[FW + f][n][0] + b1.x, acc[FW + f][n][1] + b1.y, acc[FW + f][n][2] + b1.z, acc[FW + f][n][3] + b1.w);
        } else {
          sk = make_float4(acc[f][n][0] + b0.x, acc[f][n][1] + b0.y, acc[f][n][2] + b0.z, acc[f][n][3] + b0.w);
        }
        float4 o = oin[f][n];
        o.x += sk.x; o.y += sk.y; o.z += sk.z; o.w += sk.w;
        *reinterpret_cast<float4*>(a.oacc + off) = o;
      }
    }
  }
}


// ------------------------------------------------------------------ whole WaveNet stack, one launch
// PM = A fragments per wave of the optional fused post conv (0: none).  One wave per 16 channels, as above.
template <typename T, int NF, int PM, int WV>
__global__ __launch_bounds__(WV * 64) void wn_stack_kernel(const WnStackArgs a) {
  using O = Op<T>;
  using frag = typename O::frag;
  using quad = typename O::quad;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int FW = 1;
  constexpr int NB = NF * 16;                 // frames carried by this workgroup (output tile + halo)
  constexpr int MF = 2 * FW;
  // NF 3 / 6: 32 output frames per workgroup; NF 2: 16 (tiny batches: twice the workgroups and a third less MFMA
  // work per layer for the same weight stream -- the workgroups would not fill the chip either way).
  constexpr int OUTF = NF == 2 ? 16 : kWnOutFrames;
  // the skip sum is only needed for the column fragments that overlap the output frames
  constexpr int OLO = NF > 3 ? 1 : 0, ON = NF == 2 ? 2 : 3;

  const int tid = threadIdx.x, lane = tid & 63, NTH = blockDim.x;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lq = lane >> 4;
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * OUTF;
#ifdef QVC_STAMP
  unsigned long long* const st_ = a.stamps ? a.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + wm) * 32 : nullptr;
  if (st_ && lane == 0) st_[27] = __builtin_amdgcn_s_memrealtime();
#endif
  QVC_ST(0);
  const int Tb = ragged_len(a.rg, b, a.T);    // this utterance occupies rows [Tlo, Tb): x is zero outside at every layer
  const int Tlo = ragged_lo(a.rg, b);
  if (q0 >= Tb) return;
  const int left = (a.taps - 1) / 2;
  const int halo = left * a.layers;
  const int w0 = q0 - halo;                   // first frame of the window; column j <-> frame w0 + j
  const int rowbytes = a.HP * 2;
  const int cpr = a.HP >> 3;
  const Swz sm = swz_mode(cpr);
  const int R = NB + a.taps - 1;              // x tile rows; row r <-> frame w0 - left + r
  char* acts = smem + R * rowbytes;

  // zero the x tile once: rows outside the window and K-padding channels must stay finite zeros
  for (int i = tid; i < (R * rowbytes) >> 4; i += NTH) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);

  // x residual stream and skip sum of this wave's channels live in registers for the whole stack
  f32x4 xr[FW][NF], outr[FW][ON];
  if (a.w_pre) {
    // fused `pre` 1x1 (modules.py:212): stage the z slice of the window into the (still unused) acts tile,
    // one small GEMM, and the result IS the residual stream -- no launch, no round trip through memory
    const int pcpr = a.pre_KS * 4;                              // 16-byte chunks per staged row
    for (int i = tid; i < (NB * rowbytes) >> 4; i += NTH) reinterpret_cast<uint4*>(acts)[i] = make_uint4(0u, 0u, 0u, 0u);
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
        *reinterpret_cast<frag*>(acts + r * rowbytes + ((rotc(c8, sm) ^ swz(r, sm)) << 4)) = h;
      }
    }
    __syncthreads();
    f32x4 pacc[FW][NF];
#pragma unroll
    for (int f = 0; f < FW; ++f)
#pragma unroll
      for (int n = 0; n < NF; ++n) pacc[f][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const frag* app = static_cast<const frag*>(a.w_pre) + ((size_t)wm * a.pre_KS * FW) * 64 + lane;
    gemm_loop<T, FW, NF, QVC_PF_STACK>(pacc, app, a.pre_KS, a.pre_KS, 1, acts, rowbytes, sm, lrow, lq, 0);
#pragma unroll
    for (int f = 0; f < FW; ++f) {
      const int ch0 = (wm * FW + f) * 16 + lq * 4;
      float4 bp = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ch0 < a.H) bp = *reinterpret_cast<const float4*>(a.b_pre + ch0);
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int q = w0 + n * 16 + lrow;
        const bool in = ch0 < a.H && q >= Tlo && q < Tb;
        xr[f][n] = in ? f32x4{pacc[f][n][0] + bp.x, pacc[f][n][1] + bp.y, pacc[f][n][2] + bp.z, pacc[f][n][3] + bp.w}
                      : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
#pragma unroll
  for (int f = 0; f < FW; ++f) {
    const int ch0 = (wm * FW + f) * 16 + lq * 4;
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      const int q = w0 + n * 16 + lrow;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!a.w_pre) {
        if (ch0 < a.H && q >= Tlo && q < Tb) v = *reinterpret_cast<const float4*>(a.x0 + (size_t)b * a.bs + (size_t)q * a.H + ch0);
        xr[f][n] = f32x4{v.x, v.y, v.z, v.w};
      }
      if (n >= OLO && n < OLO + ON) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.accum && ch0 < a.H && q >= q0 && q < q0 + OUTF && q < Tb)      // continue a previous launch's skip sum
          o = *reinterpret_cast<const float4*>(a.out + (size_t)b * a.bs + (size_t)q * a.H + ch0);
        outr[f][n - OLO] = f32x4{o.x, o.y, o.z, o.w};
      }
    }
  }
  __syncthreads();                            // tile zeroed before anybody writes x into it
  auto put_x = [&]() {
#pragma unroll
    for (int f = 0; f < FW; ++f) {
      const int ch0 = (wm * FW + f) * 16 + lq * 4;
      if (ch0 >= a.H) continue;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int r = n * 16 + lrow + left;
        quad h;
        h[0] = O::cvt(xr[f][n][0]); h[1] = O::cvt(xr[f][n][1]); h[2] = O::cvt(xr[f][n][2]); h[3] = O::cvt(xr[f][n][3]);
        *reinterpret_cast<quad*>(smem + r * rowbytes + ((rotc(ch0 >> 3, sm) ^ swz(r, sm)) << 4) + (ch0 & 7) * 2) = h;
      }
    }
  };
  put_x();
  __syncthreads();
  QVC_ST(1);

  for (int l = 0; l < a.layers; ++l) {
    const bool last = a.final_layer && l == a.layers - 1;      // the network's last layer has no residual half
    {   // ---- GEMM1 (k taps) + conditioning + gate -> acts tile
      f32x4 acc[MF][NF];
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      const frag* ap = static_cast<const frag*>(a.w_in[l]) + ((size_t)wm * a.nIt1 * MF) * 64 + lane;
      if (!QVC_ABL(1)) gemm_loop<T, MF, NF, QVC_PF_STACK>(acc, ap, a.nIt1, a.KS, 1, smem, rowbytes, sm, lrow, lq, 0);
#ifdef QVC_STAMP
      if (l < 4) QVC_ST(2 + 6 * l);
#endif
      const float* bb = a.bbias + (size_t)b * a.bbias_bs + (size_t)l * 2 * a.H;
#pragma unroll
      for (int f = 0; f < FW; ++f) {
        const int ch0 = (wm * FW + f) * 16 + lq * 4;
        if (ch0 >= a.HP) continue;
        float4 bt = make_float4(0.f, 0.f, 0.f, 0.f), bs = bt;
        if (ch0 < a.H) { bt = *reinterpret_cast<const float4*>(bb + ch0); bs = *reinterpret_cast<const float4*>(bb + a.H + ch0); }
#pragma unroll
        for (int n = 0; n < NF; ++n) {
          const int j = n * 16 + lrow;
          const f32x4 t = acc[f][n], sg = acc[FW + f][n];
          quad o;
          if (ch0 < a.H && QVC_ABL(4)) {
            o[0] = O::cvt(t[0] + bt.x); o[1] = O::cvt(t[1] + bt.y); o[2] = O::cvt(sg[2] + bs.z); o[3] = O::cvt(sg[3] + bs.w);
          } else if (ch0 < a.H) {
            o[0] = O::cvt(fast_tanh(t[0] + bt.x) * fast_sigmoid(sg[0] + bs.x));
            o[1] = O::cvt(fast_tanh(t[1] + bt.y) * fast_sigmoid(sg[1] + bs.y));
            o[2] = O::cvt(fast_tanh(t[2] + bt.z) * fast_sigmoid(sg[2] + bs.z));
            o[3] = O::cvt(fast_tanh(t[3] + bt.w) * fast_sigmoid(sg[3] + bs.w));
          } else {
            o[0] = o[1] = o[2] = o[3] = (T)0.f;
          }
          *reinterpret_cast<quad*>(acts + j * rowbytes + ((rotc(ch0 >> 3, sm) ^ swz(j, sm)) << 4) + (ch0 & 7) * 2) = o;
        }
      }
    }
#ifdef QVC_STAMP
    if (l < 4) QVC_ST(3 + 6 * l);
#endif
    __syncthreads();                          // acts complete; every wave is done reading the x tile
#ifdef QVC_STAMP
    if (l < 4) QVC_ST(4 + 6 * l);
#endif
    {   // ---- GEMM2 (1x1): x += res, out += skip   (modules.py:104-112)
      f32x4 acc[MF][NF];
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* brs = a.b_rs[l];
      if (!last) {
        const frag* ap = static_cast<const frag*>(a.w_rs[l]) + ((size_t)wm * a.KS * MF) * 64 + lane;
        if (!QVC_ABL(2)) gemm_loop<T, MF, NF, QVC_PF_STACK>(acc, ap, a.KS, a.KS, 1, acts, rowbytes, sm, lrow, lq, 0);
#ifdef QVC_STAMP
        if (l < 4) QVC_ST(5 + 6 * l);
#endif
#pragma unroll
        for (int f = 0; f < FW; ++f) {
          const int ch0 = (wm * FW + f) * 16 + lq * 4;
          if (ch0 >= a.H) continue;
          const float4 b0 = *reinterpret_cast<const float4*>(brs + ch0);
          const float4 b1 = *reinterpret_cast<const float4*>(brs + a.H + ch0);
#pragma unroll
          for (int n = 0; n < NF; ++n) {
            const int q = w0 + n * 16 + lrow;
            const bool in = q >= Tlo && q < Tb;                  // the convs zero-pad x outside the utterance
            xr[f][n][0] = in ? xr[f][n][0] + acc[f][n][0] + b0.x : 0.f;
            xr[f][n][1] = in ? xr[f][n][1] + acc[f][n][1] + b0.y : 0.f;
            xr[f][n][2] = in ? xr[f][n][2] + acc[f][n][2] + b0.z : 0.f;
            xr[f][n][3] = in ? xr[f][n][3] + acc[f][n][3] + b0.w : 0.f;
            if (n >= OLO && n < OLO + ON) {
              outr[f][n - OLO][0] += acc[FW + f][n][0] + b1.x; outr[f][n - OLO][1] += acc[FW + f][n][1] + b1.y;
              outr[f][n - OLO][2] += acc[FW + f][n][2] + b1.z; outr[f][n - OLO][3] += acc[FW + f][n][3] + b1.w;
            }
          }
        }
        put_x();                              // safe: all waves are past GEMM1 of this layer
      } else {
        f32x4 (&acl)[FW][NF] = reinterpret_cast<f32x4 (&)[FW][NF]>(acc);
        const frag* ap = static_cast<const frag*>(a.w_rs[l]) + ((size_t)wm * a.KS * FW) * 64 + lane;
        gemm_loop<T, FW, NF, QVC_PF_STACK>(acl, ap, a.KS, a.KS, 1, acts, rowbytes, sm, lrow, lq, 0);
#pragma unroll
        for (int f = 0; f < FW; ++f) {
          const int ch0 = (wm * FW + f) * 16 + lq * 4;
          if (ch0 >= a.H) continue;
          const float4 b0 = *reinterpret_cast<const float4*>(brs + ch0);
#pragma unroll
          for (int n = 0; n < NF; ++n) {
            if (n >= OLO && n < OLO + ON) {
              outr[f][n - OLO][0] += acl[f][n][0] + b0.x; outr[f][n - OLO][1] += acl[f][n][1] + b0.y;
              outr[f][n - OLO][2] += acl[f][n][2] + b0.z; outr[f][n - OLO][3] += acl[f][n][3] + b0.w;
            }
          }
        }
      }
    }
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
      // fused `post` 1x1 + coupling update (modules.py:214-217): the skip sum goes (operand type) into the acts
      // tile, one small GEMM, and z[:, post slice] -= m for the output frames.  The skip sum itself is not stored.
#pragma unroll
      for (int f = 0; f < FW; ++f) {
        const int ch0 = (wm * FW + f) * 16 + lq * 4;
        if (ch0 >= a.HP) continue;
#pragma unroll
        for (int n = OLO; n < OLO + ON; ++n) {
          const int j = n * 16 + lrow;
          quad h;
          if (ch0 < a.H) {
            h[0] = O::cvt(outr[f][n - OLO][0]); h[1] = O::cvt(outr[f][n - OLO][1]);
            h[2] = O::cvt(outr[f][n - OLO][2]); h[3] = O::cvt(outr[f][n - OLO][3]);
          } else {
            h[0] = h[1] = h[2] = h[3] = (T)0.f;
          }
          *reinterpret_cast<quad*>(acts + j * rowbytes + ((rotc(ch0 >> 3, sm) ^ swz(j, sm)) << 4) + (ch0 & 7) * 2) = h;
        }
      }
      __syncthreads();
      f32x4 qacc[PM][ON];
#pragma unroll
      for (int m = 0; m < PM; ++m)
#pragma unroll
        for (int n = 0; n < ON; ++n) qacc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      const frag* apq = static_cast<const frag*>(a.w_post) + ((size_t)wm * a.KS * PM) * 64 + lane;
      gemm_loop<T, PM, ON, QVC_PF_STACK>(qacc, apq, a.KS, a.KS, 1, acts, rowbytes, sm, OLO * 16 + lrow, lq, 0);
#pragma unroll
      for (int m = 0; m < PM; ++m) {
        const int v = (wm * PM + m) * 16 + lq * 4;
        if (v >= a.post_m) continue;
        const float4 bq = *reinterpret_cast<const float4*>(a.b_post + v);
        float4 zin[ON];                      // all loads of the read-modify-write before its first store
#pragma unroll
        for (int n = 0; n < ON; ++n) {
          const int q = w0 + (OLO + n) * 16 + lrow;
          zin[n] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (q >= q0 && q < q0 + OUTF && q < Tb)
            zin[n] = *reinterpret_cast<const float4*>(a.z + (size_t)b * a.z_bs + (size_t)q * a.z_ts + a.post_c0 + v);
        }
#pragma unroll
        for (int n = 0; n < ON; ++n) {
          const int q = w0 + (OLO + n) * 16 + lrow;
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
#pragma unroll
  for (int f = 0; f < FW; ++f) {
    const int ch0 = (wm * FW + f) * 16 + lq * 4;
    if (ch0 >= a.H) continue;
#pragma unroll
    for (int n = OLO; n < OLO + ON; ++n) {
      const int q = w0 + n * 16 + lrow;
      if (q >= q0 && q < q0 + OUTF && q < Tb) {
        const size_t off = (size_t)b * a.bs + (size_t)q * a.H + ch0;
        *reinterpret_cast<float4*>(a.out + off) =
            make_float4(outr[f][n - OLO][0], outr[f][n - OLO][1], outr[f][n - OLO][2], outr[f][n - OLO][3]);
        if (a.x_out) *reinterpret_cast<float4*>(a.x_out + off) = make_float4(xr[f][n][0], xr[f][n][1], xr[f][n][2], xr[f][n][3]);
      }
    }
  }
#ifdef QVC_STAMP
  QVC_ST(26);
  if (st_ && lane == 0) st_[28] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ------------------------------------------------------------------ launch-side tile selection
struct TileChoice { int NF; int blocks; size_t lds; };

inline TileChoice choose_tile(const ConvDesc& d, int Nq, int batch, const int* nf_list, int n_nf) {
  const int halo = (d.taps - 1) * d.dil;
  const int rowbytes = d.CinP * 2;
  const int WN = kWaves / d.WM;
  TileChoice best{0, 0, 0};
  double best_cost = 1e300;
  for (int i = 0; i < n_nf; ++i) {
    const int NF = nf_list[i];
#ifdef QVC_FORCE_NF
    if (NF != QVC_FORCE_NF) continue;                           // developer sweep (tools/conv_bench)
#endif
    if (d.MF * NF * 4 > 160) continue;                          // accumulator registers
    // polyphase up-samplers have s * Cout rows = many row chunks, and every workgroup streams its chunk's whole K:
    // with 32-frame tiles up-sampler 0 pulls 4.9 MB of weights into every CU (45 us at the ~50 B/clk a CU takes in) --
    // 64-frame tiles halve that while the launch still has two workgroups per CU (57.9 -> 50.3 us, bit-identical)
    if (d.up_s > 1 && NF < 4 && (long)ceil_div(Nq, WN * 4 * 16) * batch * d.nchunk >= 512) continue;
    const int NT = WN * NF * 16;
    const size_t lds = (size_t)(NT + halo) * rowbytes;
    if (lds > 160 * 1024) continue;
    const int tiles = ceil_div(Nq, NT);
    const long blocks = (long)tiles * batch * d.nchunk;
    // workgroups that can share a CU (LDS, and the ~512 registers/lane of a SIMD): with >= 2 the
    // staging / epilogue of one overlaps the MFMA phase of another
    const int regs = d.MF * NF * 4 + 16 * d.MF + 8 * NF + 40;
    const int bpc = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(3, (160 * 1024) / lds), (size_t)(512 / regs)));
    const long rounds = (blocks + 256L * bpc - 1) / (256L * bpc);
    // work per block ~ frames computed + a fixed overhead (staging, launch, epilogue), in frame units
    const double work = NT + 0.35 * halo + 24.0;
    const double cost = (double)rounds * bpc * work * (bpc == 1 ? 1.3 : 1.0);
    if (cost < best_cost) { best_cost = cost; best = TileChoice{NF, (int)blocks, lds}; }
#ifndef QVC_FORCE_NF
    // latency-sized launch: even the smallest tile gives every workgroup a CU of its own -- take it (each workgroup
    // streams the same weights whatever its tile, so fewer frames per workgroup is less serial MFMA work behind them;
    // batch 1, up-sampler 0: 35 -> 22 us)
    if (i == 0 && blocks <= 256) return best;
#endif
  }
  return best;
}

template <typename T, int MF, int NF, int WM, int EPI, bool CL = false>
inline int launch_one_cl(const ConvArgs& a, int batch, size_t lds, hipStream_t stream) {
  auto kern = conv_mfma_kernel<T, MF, NF, WM, EPI, CL>;
  static std::atomic<uint32_t> lds_ok{0};                  // > 64 KiB dynamic LDS: opt in once per device
  if (!allow_big_lds(lds_ok, reinterpret_cast<const void*>(kern))) return QVC_ERR_LAUNCH;
  constexpr int NT = (kWaves / WM) * NF * 16;
  dim3 grid((unsigned)ceil_div(a.Nq, NT), (unsigned)batch, (unsigned)(CL ? 1 : a.nchunk));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
  return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
}
template <typename T, int MF, int NF, int WM, int EPI>
inline int launch_one(const ConvArgs& a, int batch, size_t lds, hipStream_t stream) {
  // chunk loop: built for the 4 x 4-fragment layout; taken when the tile is the mean of three streams (expensive to
  // stage) and the launch still has two workgroups for every CU without the chunk dimension
  if constexpr (EPI == EPI_STD && MF == 4 && WM == 4 && NF <= 5) {
    constexpr int NT = (kWaves / WM) * NF * 16;
    if (debug_get(DBG_CONV_CL) && a.nchunk >= 2 && a.x2 && (long)ceil_div(a.Nq, NT) * batch >= 512)
      return launch_one_cl<T, MF, NF, WM, EPI, true>(a, batch, lds, stream);
  }
  return launch_one_cl<T, MF, NF, WM, EPI, false>(a, batch, lds, stream);
}

template <typename T, int MF, int WM, int EPI>
inline int launch_nf(const ConvDesc& d, const ConvArgs& a, int batch, hipStream_t stream, int* nf_out) {
  static const int nfs[] = {2, 4, 5, 8, 10};
  const TileChoice tc = choose_tile(d, a.Nq, batch, nfs, 5);
  if (nf_out) *nf_out = tc.NF;
  switch (tc.NF) {
    case 2: return launch_one<T, MF, 2, WM, EPI>(a, batch, tc.lds, stream);
    case 4: return launch_one<T, MF, 4, WM, EPI>(a, batch, tc.lds, stream);
    case 5: if constexpr (MF * 5 * 4 <= 160) return launch_one<T, MF, 5, WM, EPI>(a, batch, tc.lds, stream); break;
    case 8: if constexpr (MF * 8 * 4 <= 160) return launch_one<T, MF, 8, WM, EPI>(a, batch, tc.lds, stream); break;
    case 10: if constexpr (MF * 10 * 4 <= 160) return launch_one<T, MF, 10, WM, EPI>(a, batch, tc.lds, stream); break;
    default: break;
  }
  return QVC_ERR_BAD_CONFIG;                                 // no tile fits LDS / registers
}

template <typename T>
int launch_conv_typed(const ConvDesc& d, const ConvArgs& a, int batch, int epi, void* stream_v, int* nf_out) {
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (epi == EPI_GAU) {
    if (d.WM != 4) return QVC_ERR_BAD_CONFIG;
    switch (d.MF) {
      case 2: return launch_nf<T, 2, 4, EPI_GAU>(d, a, batch, stream, nf_out);
      case 4: return launch_nf<T, 4, 4, EPI_GAU>(d, a, batch, stream, nf_out);
      case 6: return launch_nf<T, 6, 4, EPI_GAU>(d, a, batch, stream, nf_out);
      default: return QVC_ERR_BAD_CONFIG;
    }
  }
  if (epi == EPI_SAMPLE) {
    if (d.WM != 4 || !d.gau || !a.noise || !a.y32 || !a.bias) return QVC_ERR_BAD_CONFIG;
    switch (d.MF) {
      case 2: return launch_nf<T, 2, 4, EPI_SAMPLE>(d, a, batch, stream, nf_out);
      case 4: return launch_nf<T, 4, 4, EPI_SAMPLE>(d, a, batch, stream, nf_out);
      case 6: return launch_nf<T, 6, 4, EPI_SAMPLE>(d, a, batch, stream, nf_out);
      default: return QVC_ERR_BAD_CONFIG;
    }
  }
  if (d.WM > kWaves) return QVC_ERR_BAD_CONFIG;            // 8-wave layouts exist for the fused pair kernel only
  switch (d.WM * 10 + d.MF) {
    case 41: return launch_nf<T, 1, 4, EPI_STD>(d, a, batch, stream, nf_out);
    case 42: return launch_nf<T, 2, 4, EPI_STD>(d, a, batch, stream, nf_out);
    case 43: return launch_nf<T, 3, 4, EPI_STD>(d, a, batch, stream, nf_out);
    case 44: return launch_nf<T, 4, 4, EPI_STD>(d, a, batch, stream, nf_out);
    case 23: return launch_nf<T, 3, 2, EPI_STD>(d, a, batch, stream, nf_out);
    case 24: return launch_nf<T, 4, 2, EPI_STD>(d, a, batch, stream, nf_out);
    case 14: return launch_nf<T, 4, 1, EPI_STD>(d, a, batch, stream, nf_out);
    default: return QVC_ERR_BAD_CONFIG;
  }
}

// ---- fused pair: tile choice + dispatch
inline TileChoice choose_pair_tile(const ConvDesc* ds, int n, int T, int batch) {
  static const int nfs[] = {2, 4, 5, 8, 10}; // measured at B=32, MF 4: NF 5 (2 workgroups/CU) beats 4 and 8 by 7-35 %
  const ConvDesc& d = ds[0];
  const int NWV = block_waves(d);
  const int WN = NWV / d.WM;
  int halo1 = 0;                              // the chains of a launch share one tile shape: size it for the widest halo
  for (int i = 0; i < n; ++i) halo1 = std::max(halo1, (ds[i].taps - 1) * ds[i].dil);
  const int rowbytes = d.CinP * 2;
  TileChoice best{0, 0, 0};
  double best_cost = 1e300;
  for (int NF : nfs) {
#ifdef QVC_FORCE_NF
    if (NF != QVC_FORCE_NF) continue;
#endif
    if (d.MF * (NF + 1) * 4 > 160 || (NF == 10 && d.MF > 2)) continue;
    const size_t lds = (size_t)(WN * (NF + 1) * 16 + halo1) * rowbytes;
    if (lds > 160 * 1024) continue;
    const int NT = WN * NF * 16;
    const long blocks = (long)ceil_div(T, NT) * batch * n;
    // workgroups that can share a CU: 2 x 4 waves or 1 x 8 waves (two waves per SIMD either way)
    const int bpc = NWV == 8 ? 1 : (int)std::min<size_t>(2, (160 * 1024) / lds);
    const long rounds = (blocks + 256L * bpc - 1) / (256L * bpc);
    // per-workgroup work in frame units: both GEMMs + staging/epilogue; a single 4-wave workgroup per CU
    // cannot overlap its memory phases with another one's MFMA phases
    const double work = WN * (NF + 1) * 16 + NT + 0.35 * halo1 + 32.0;
    const double cost = rounds * bpc * work * ((bpc == 1 && NWV == 4) ? 1.3 : 1.0);
    if (cost < best_cost) { best_cost = cost; best = TileChoice{NF, (int)blocks, lds}; }
  }
  return best;
}

template <typename T, int MF, int NF, int WM, int NWV, typename TS>
inline int launch_pair_one(const PairArgs3& a, int batch, size_t lds, hipStream_t stream) {
  auto kern = rbpair_kernel<T, MF, NF, WM, NWV, TS>;
  static std::atomic<uint32_t> lds_ok{0};                  // > 64 KiB dynamic LDS: opt in once per device
  if (!allow_big_lds(lds_ok, reinterpret_cast<const void*>(kern))) return QVC_ERR_LAUNCH;
  constexpr int NT = (NWV / WM) * NF * 16;
  const unsigned tiles = (unsigned)ceil_div(a.p[0].T, NT);
  const dim3 grid = a.chain_major ? dim3(tiles, (unsigned)batch, (unsigned)a.n) : dim3(tiles * (unsigned)a.n, (unsigned)batch, 1);
  hipLaunchKernelGGL(kern, grid, dim3(NWV * 64), lds, stream, a);
  return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
}

template <typename T, int MF, int WM, int NWV, typename TS>
inline int launch_pair_nf(const ConvDesc* ds, const PairArgs3& a, int batch, hipStream_t stream, int* nf_out) {
  const TileChoice tc = choose_pair_tile(ds, a.n, a.p[0].T, batch);
  if (nf_out) *nf_out = tc.NF;
  switch (tc.NF) {
    case 2: return launch_pair_one<T, MF, 2, WM, NWV, TS>(a, batch, tc.lds, stream);
    case 4: return launch_pair_one<T, MF, 4, WM, NWV, TS>(a, batch, tc.lds, stream);
    case 5: return launch_pair_one<T, MF, 5, WM, NWV, TS>(a, batch, tc.lds, stream);
    case 8: if constexpr (MF * 9 * 4 <= 160) return launch_pair_one<T, MF, 8, WM, NWV, TS>(a, batch, tc.lds, stream); break;
    case 10: if constexpr (MF * 11 * 4 <= 96) return launch_pair_one<T, MF, 10, WM, NWV, TS>(a, batch, tc.lds, stream); break;
    default: break;
  }
  return QVC_ERR_BAD_CONFIG;
}

template <typename T, typename TS>
int launch_pair_typed(const ConvDesc* ds, const PairArgs3& a, int batch, void* stream_v, int* nf_out) {
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const ConvDesc& d = ds[0];
  switch (d.WM * 10 + d.MF) {
    case 41: return launch_pair_nf<T, 1, 4, 4, TS>(ds, a, batch, stream, nf_out);
    case 42: return launch_pair_nf<T, 2, 4, 4, TS>(ds, a, batch, stream, nf_out);
    case 43: return launch_pair_nf<T, 3, 4, 4, TS>(ds, a, batch, stream, nf_out);
    case 44: return launch_pair_nf<T, 4, 4, 4, TS>(ds, a, batch, stream, nf_out);
    case 23: return launch_pair_nf<T, 3, 2, 4, TS>(ds, a, batch, stream, nf_out);
    case 24: return launch_pair_nf<T, 4, 2, 4, TS>(ds, a, batch, stream, nf_out);
    case 14: return launch_pair_nf<T, 4, 1, 4, TS>(ds, a, batch, stream, nf_out);
    case 82: return launch_pair_nf<T, 2, 8, 8, TS>(ds, a, batch, stream, nf_out);
    case 83: return launch_pair_nf<T, 3, 8, 8, TS>(ds, a, batch, stream, nf_out);
    case 84: return launch_pair_nf<T, 4, 8, 8, TS>(ds, a, batch, stream, nf_out);
    default: return QVC_ERR_BAD_CONFIG;
  }
}

// ---- fused WaveNet layer dispatch
inline int wn_wave_bucket(int waves) { return waves <= 4 ? 4 : (waves <= 8 ? 8 : (waves <= 12 ? 12 : 16)); }

template <typename T, int NF, int WV>
inline int launch_wn_one(const WnArgs& a, int waves, int batch, hipStream_t stream) {
  const size_t lds = (size_t)(NF * 16 + a.taps - 1 + NF * 16) * a.HP * 2;
  dim3 grid((unsigned)ceil_div(a.T, NF * 16), (unsigned)batch), block((unsigned)waves * 64);
  // hidden = 256 with 64-frame tiles needs (64 + taps-1 + 64) * 512 B > 64 KiB of dynamic LDS: opt in per device
  static std::atomic<uint32_t> lds_ok_last{0}, lds_ok_mid{0};
  if (a.last) {
    auto kern = wn_layer_kernel<T, NF, true, WV>;
    if (lds > 64 * 1024 && !allow_big_lds(lds_ok_last, reinterpret_cast<const void*>(kern))) return QVC_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  } else {
    auto kern = wn_layer_kernel<T, NF, false, WV>;
    if (lds > 64 * 1024 && !allow_big_lds(lds_ok_mid, reinterpret_cast<const void*>(kern))) return QVC_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  }
  return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
}

template <typename T, int NF>
inline int launch_wn_nf(const WnArgs& a, int waves, int batch, hipStream_t stream) {
  switch (wn_wave_bucket(waves)) {
    case 4: return launch_wn_one<T, NF, 4>(a, waves, batch, stream);
    case 8: return launch_wn_one<T, NF, 8>(a, waves, batch, stream);
    case 12: return launch_wn_one<T, NF, 12>(a, waves, batch, stream);
    default: return launch_wn_one<T, NF, 16>(a, waves, batch, stream);
  }
}

template <typename T>
int launch_wn_typed(const ConvDesc& din, const WnArgs& a, int batch, void* stream_v, int* nf_out) {
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  // 32-frame tiles unless that leaves most CUs without a workgroup anyway and 64 halves the weight traffic
  const long blocks32 = (long)ceil_div(a.T, 32) * batch;
  const int NF = blocks32 >= 512 ? 4 : 2;
  if (nf_out) *nf_out = NF;
  if (!wn_layout_ok(din)) return QVC_ERR_BAD_CONFIG;
  return NF == 4 ? launch_wn_nf<T, 4>(a, din.WM, batch, stream) : launch_wn_nf<T, 2>(a, din.WM, batch, stream);
}

// ---- whole-stack WaveNet dispatch
template <typename T, int NF, int PM, int WV>
inline int launch_wn_stack_pm(const WnStackArgs& a, int waves, int batch, hipStream_t stream) {
  auto kern = wn_stack_kernel<T, NF, PM, WV>;
  const size_t lds = (size_t)(NF * 16 + a.taps - 1 + NF * 16) * a.HP * 2;
  static std::atomic<uint32_t> lds_ok{0};                  // > 64 KiB dynamic LDS: opt in once per device
  if (!allow_big_lds(lds_ok, reinterpret_cast<const void*>(kern))) return QVC_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(a.T, NF == 2 ? 16 : kWnOutFrames), (unsigned)batch), dim3((unsigned)waves * 64), lds, stream, a);
  return hipGetLastError() == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH;
}

template <typename T, int NF, int PM>
inline int launch_wn_stack_wv(const WnStackArgs& a, int waves, int batch, hipStream_t stream) {
  switch (wn_wave_bucket(waves)) {
    case 4: return launch_wn_stack_pm<T, NF, PM, 4>(a, waves, batch, stream);
    case 8: return launch_wn_stack_pm<T, NF, PM, 8>(a, waves, batch, stream);
    case 12: return launch_wn_stack_pm<T, NF, PM, 12>(a, waves, batch, stream);
    default: return launch_wn_stack_pm<T, NF, PM, 16>(a, waves, batch, stream);
  }
}

template <typename T>
int launch_wn_stack_typed(const ConvDesc& din, const WnStackArgs& a, int batch, void* stream_v) {
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (!wn_stack_ok(din, a.layers)) return QVC_ERR_BAD_CONFIG;
  int nf = wn_stack_nf(a.taps, a.layers);
  const int pm = a.w_post ? a.post_mf : 0;
  if (pm > 1) return QVC_ERR_BAD_CONFIG;
  // tiny batches (<= 64 of the 32-frame tiles: a quarter of the CUs): 16-frame tiles, when the halo fits a 32-column window
  if (nf == 3 && 16 + (a.taps - 1) * a.layers <= 32 && (long)batch * ceil_div(a.T, kWnOutFrames) <= 64)
    return pm ? launch_wn_stack_wv<T, 2, 1>(a, din.WM, batch, stream) : launch_wn_stack_wv<T, 2, 0>(a, din.WM, batch, stream);
  if (nf == 3) return pm ? launch_wn_stack_wv<T, 3, 1>(a, din.WM, batch, stream) : launch_wn_stack_wv<T, 3, 0>(a, din.WM, batch, stream);
  return pm ? launch_wn_stack_wv<T, 6, 1>(a, din.WM, batch, stream) : launch_wn_stack_wv<T, 6, 0>(a, din.WM, batch, stream);
}

}  // namespace qvc
