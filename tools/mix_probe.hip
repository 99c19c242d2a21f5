// tools/mix_probe.hip -- developer probe: does a CU's traffic to far memory hold back a weight stream that hits in L2?
//
// The fused pair kernel's GEMM phases and memory phases do not overlap on a CU.  This probe puts both
// on every CU without the kernel around them: in each 8-wave workgroup (one per CU) waves 0-3 run a GEMM-like loop --
// two 1-KiB weight fragments from an L2-resident buffer per 22 MFMAs, four k-steps ahead, the pair kernel's ratio -- and
// waves 4-7 fetch 64-KiB "tiles" from a 2-GiB buffer (16 loads of 16 bytes per lane in flight, then a wait), either
//   mode 0: not at all,
//   mode 1: with plain global_load_dwordx4 into VGPRs,
//   mode 2: with global_load_lds_dwordx4 (LDS-DMA, no VGPR return path),
// back to back or with pauses (duty).  Prints the GEMM waves' cycles per k-step and the tile waves' bytes per clock.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mix_probe.hip -o tools/mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

constexpr int kSteps = 1500;          // k-steps per GEMM wave
constexpr int kWFrags = 256;          // weight fragments per GEMM wave (256 KiB; 4 waves = 1 MiB, shared by every workgroup)

template <int MODE>
__global__ __launch_bounds__(512) void mix(const uint4* __restrict__ w, const uint4* __restrict__ far, size_t far_tiles, int pause,
                                            unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __shared__ int done;
  if (threadIdx.x == 0) done = 0;
  __syncthreads();
  if (wave < 4) {
    const uint4* base = w + (size_t)wave * kWFrags * 64 + lane;
    f32x4 acc[22];
#pragma unroll
    for (int i = 0; i < 22; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 bfr = {(_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)1.f, (_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)1.f};
    uint4 ring[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) { ring[u][0] = base[(size_t)(2 * u) * 64]; ring[u][1] = base[(size_t)(2 * u + 1) * 64]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i0 = 0; i0 < kSteps; i0 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f16x8 a0, a1; __builtin_memcpy(&a0, &ring[u][0], 16); __builtin_memcpy(&a1, &ring[u][1], 16);
        const int nx = ((i0 + u + 4) * 2) % kWFrags;
        ring[u][0] = base[(size_t)nx * 64]; ring[u][1] = base[(size_t)(nx + 1) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 11; ++m) {
          acc[2 * m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bfr, acc[2 * m], 0, 0, 0);
          acc[2 * m + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bfr, acc[2 * m + 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[blockIdx.x * 16 + wave] = t1 - t0; atomicAdd(&done, 1); }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 22; ++i) s += acc[i][0];
    if (s == 12345.678f) sink[0] = s;
  } else {
    // tile waves: until the GEMM waves are done (bounded: at most kSteps tiles)
    const int tw = wave - 4;
    uint4 x = make_uint4(0, 0, 0, 0);
    unsigned long long bytes = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (MODE != 0) {
      const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)tw * 16 * 1024;
      size_t tile = ((size_t)blockIdx.x * 977 + 13) % far_tiles;
      for (int it = 0; it < kSteps; ++it) {
        if (*(volatile int*)&done >= 4) break;
        const uint4* src = far + tile * 4096 + (size_t)tw * 1024 + lane;     // a tile = 4096 x 16 B = 64 KiB; this wave's quarter
        if constexpr (MODE == 1) {
          uint4 v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = src[(size_t)u * 64];
#pragma unroll
          for (int u = 0; u < 16; ++u) { x.x ^= v[u].x; x.y ^= v[u].y; x.z ^= v[u].z; x.w ^= v[u].w; }
        } else {
#pragma unroll
          for (int u = 0; u < 16; ++u) glds16(src + (size_t)u * 64, lds_base + u * 1024);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const uint4 v = *reinterpret_cast<const uint4*>(smem + (size_t)tw * 16 * 1024 + lane * 16);
          x.x ^= v.x;
        }
        bytes += 16 * 1024;
        tile = (tile * 1103515245ull + 12345ull + blockIdx.x) % far_tiles;
        for (int p = 0; p < pause; ++p) __builtin_amdgcn_s_sleep(100);    // ~6400 cycles per unit
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[blockIdx.x * 16 + wave] = t1 - t0; out[blockIdx.x * 16 + 8 + tw] = bytes; }
    if ((x.x ^ x.y ^ x.z ^ x.w) == 0x12345678u) sink[1] = 1.f;
  }
}

template <int MODE>
void run(const char* name, const uint4* w, const uint4* far, size_t far_tiles, int pause, unsigned long long* dout, float* sink) {
  auto k = mix<MODE>;
  const size_t lds = 100 * 1024;                     // one workgroup per CU
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  CK(hipMemset(dout, 0, 256 * 16 * 8));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, w, far, far_tiles, pause, dout, sink);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(256 * 16);
  CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
  double gc = 0, tc = 0, tb = 0;
  for (int g = 0; g < 256; ++g) for (int wv = 0; wv < 4; ++wv) { gc += (double)h[g * 16 + wv]; tc += (double)h[g * 16 + 4 + wv]; tb += (double)h[g * 16 + 8 + wv]; }
  gc /= 1024; tc /= 1024;
  printf("%-34s pause %d : GEMM waves %7.1f cycles per k-step (22 MFMA = 352 at the pipe's rate, one wave per SIMD) | tile waves %6.1f B/clk/CU\n",
         name, pause, gc / kSteps, tc > 0 ? tb / 256 / tc : 0.0);
}

int main() {
  const size_t wbytes = (size_t)4 * kWFrags * 1024;
  uint4* w; CK(hipMalloc(&w, wbytes));
  std::vector<uint16_t> h(wbytes / 2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(0x3000 + ((i * 40503u) & 0x7ff));
  CK(hipMemcpy(w, h.data(), wbytes, hipMemcpyHostToDevice));
  const size_t far_tiles = 32768;                     // x 64 KiB = 2 GiB
  uint4* far; CK(hipMalloc(&far, far_tiles * 65536)); CK(hipMemset(far, 1, far_tiles * 65536));
  unsigned long long* dout; CK(hipMalloc(&dout, 256 * 16 * 8));
  float* sink; CK(hipMalloc(&sink, 8));
  run<0>("no tile traffic", w, far, far_tiles, 0, dout, sink);
  for (int pause : {0, 1, 4}) {
    run<1>("tiles by global_load (VGPR)", w, far, far_tiles, pause, dout, sink);
    run<2>("tiles by LDS-DMA", w, far, far_tiles, pause, dout, sink);
  }
  run<0>("no tile traffic", w, far, far_tiles, 0, dout, sink);
  return 0;
}
