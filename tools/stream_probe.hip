// tools/stream_probe.hip -- developer probe: how fast can ONE CU take in a weight stream that is L2-resident and read
// by every CU at once (the WaveNet stack's situation: 864 KB per layer into every CU, each byte used by one wave)?
//   mode 0: global_load_dwordx4 -> VGPR, DEPTH loads in flight per wave (register ring)
//   mode 1: global_load_lds_dwordx4 (LDS-DMA) into a per-wave LDS ring of DEPTH 1-KiB slots, read back with ds_read_b128
// optional MFMA work per fragment pair (6 MFMAs per 2 fragments = the WaveNet layer's ratio) to see what the stream
// keeps beside a busy matrix pipe.  Prints bytes per clock per CU (s_memtime) and GB/s per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_probe.hip -o tools/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// LDS-DMA through inline asm: hipcc does not count asm memory operations, so it emits no vmcnt(0) in front of the ds_read
// that follows (with the builtin it drains the whole ring before every read).  M0 = the wave-uniform LDS byte address.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// fragments: [wave][nfrag][64 lanes][16 B]; every workgroup reads the same buffer (as the WaveNet weights are shared)
template <int MODE, int DEPTH, int MFMA>
__global__ __launch_bounds__(768) void probe(const uint4* __restrict__ w, int nfrag, int reps, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint4* base = w + (size_t)wave * nfrag * 64 + lane;
  f32x4 acc[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  f16x8 bfr = {(_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)1.f, (_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)1.f};
  uint4 x = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int total = nfrag * reps;
  if constexpr (MODE == 0) {
    uint4 ring[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) ring[u] = base[(size_t)(u % nfrag) * 64];
    for (int i0 = 0; i0 < total; i0 += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        const uint4 v = ring[u];
        int nx = i0 + u + DEPTH; nx = nx < total ? nx % nfrag : 0;
        ring[u] = base[(size_t)nx * 64];
        if constexpr (MFMA > 0) {
          f16x8 a; __builtin_memcpy(&a, &v, 16);
#pragma unroll
          for (int m = 0; m < MFMA; ++m) acc[m % 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bfr, acc[m % 3], 0, 0, 0);
        } else {
          x.x ^= v.x; x.y ^= v.y; x.z ^= v.z; x.w ^= v.w;
        }
      }
    }
  } else {
    char* ring = smem + (size_t)wave * DEPTH * 1024;
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
    // (the source address is per lane; the LDS destination is wave-uniform base + lane * 16)
#pragma unroll
    for (int u = 0; u < DEPTH - 1; ++u) {
      const int nx = u % nfrag;
      glds16(base + (size_t)nx * 64, ring_lds + u * 1024);
    }
    for (int i0 = 0; i0 < total; i0 += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        const int i = i0 + u;
        {   // refill the slot consumed in the previous trip (DEPTH-1 loads stay in flight)
          int nx = i + DEPTH - 1; nx = nx < total ? nx % nfrag : 0;
          glds16(base + (size_t)nx * 64, ring_lds + ((u + DEPTH - 1) % DEPTH) * 1024);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
        const uint4 v = *reinterpret_cast<const uint4*>(ring + u * 1024 + lane * 16);
        if constexpr (MFMA > 0) {
          f16x8 a; __builtin_memcpy(&a, &v, 16);
#pragma unroll
          for (int m = 0; m < MFMA; ++m) acc[m % 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bfr, acc[m % 3], 0, 0, 0);
        } else {
          x.x ^= v.x; x.y ^= v.y; x.z ^= v.z; x.w ^= v.w;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
  const float s = acc[0][0] + acc[1][1] + acc[2][2] + (float)(x.x ^ x.y ^ x.z ^ x.w);
  if (s == 12345.678f) sink[0] = s;
}

template <int MODE, int DEPTH, int MFMA>
void run(const char* name, const uint4* w, int nfrag, int reps, int grid, unsigned long long* dout, float* sink) {
  auto k = probe<MODE, DEPTH, MFMA>;
  const size_t lds = MODE == 1 ? (size_t)12 * DEPTH * 1024 : 0;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(768), lds, 0, w, nfrag, reps, dout, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(768), lds, 0, w, nfrag, reps, dout, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)grid * 16);
  CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0; int n = 0;
  for (int g = 0; g < grid; ++g) for (int wv = 0; wv < 12; ++wv) { cyc += (double)h[(size_t)g * 16 + wv]; ++n; }
  cyc /= n;
  const double bytes_cu = 12.0 * nfrag * reps * 1024;
  const double us = ms * 1e3 / 5;
  printf("%-26s grid %3d  depth %2d  mfma/frag %d : %7.1f B/clk/CU (in-kernel cycles %.0f)  %6.1f GB/s per CU  %6.2f TB/s chip  (%.1f us)\n", name, grid, DEPTH, MFMA,
         bytes_cu / cyc, cyc, bytes_cu / us * 1e-3, bytes_cu * grid / us * 1e-6, us);
}

int main() {
  const int nfrag = 72 * 4;                  // per wave: 72 fragments per layer x 4 layers = 288 KiB; x 12 waves = 3.4 MB (fits every XCD's L2)
  const int reps = 4;
  const size_t bytes = (size_t)12 * nfrag * 1024;
  uint4* w; CK(hipMalloc(&w, bytes));
  std::vector<uint16_t> h(bytes / 2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(0x3000 + ((i * 40503u) & 0x7ff));
  CK(hipMemcpy(w, h.data(), bytes, hipMemcpyHostToDevice));
  unsigned long long* dout; CK(hipMalloc(&dout, 256 * 16 * 8));
  float* sink; CK(hipMalloc(&sink, 4));
  for (int grid : {256, 64, 8}) {
    run<0, 2, 0>("vgpr ring", w, nfrag, reps, grid, dout, sink);
    run<0, 4, 0>("vgpr ring", w, nfrag, reps, grid, dout, sink);
    run<0, 8, 0>("vgpr ring", w, nfrag, reps, grid, dout, sink);
    run<0, 16, 0>("vgpr ring", w, nfrag, reps, grid, dout, sink);
    run<1, 4, 0>("lds-dma ring", w, nfrag, reps, grid, dout, sink);
    run<1, 8, 0>("lds-dma ring", w, nfrag, reps, grid, dout, sink);
    run<1, 12, 0>("lds-dma ring", w, nfrag, reps, grid, dout, sink);
    run<0, 8, 3>("vgpr ring + mfma", w, nfrag, reps, grid, dout, sink);
    run<0, 16, 3>("vgpr ring + mfma", w, nfrag, reps, grid, dout, sink);
    run<1, 8, 3>("lds-dma ring + mfma", w, nfrag, reps, grid, dout, sink);
    run<1, 12, 3>("lds-dma ring + mfma", w, nfrag, reps, grid, dout, sink);
  }
  return 0;
}
