// tools/launch_probe.hip -- developer probe: how long does the hardware take to START the waves of one big workgroup
// per CU?  (The WaveNet stack kernels -- 12 waves, 1 workgroup per CU, all 256 workgroups launched at once -- show their
// waves starting ~0.5 us apart: 6 us from the first wave to the last, profiles/r03_wn_stamps.txt.)
// Each wave records s_memrealtime (100 MHz) on entry; variants: threads per workgroup, VGPRs per lane, dynamic LDS, argument bytes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/launch_probe.hip -o tools/launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct BigArgs { unsigned long long* out; int spin; char pad[560]; };
struct SmallArgs { unsigned long long* out; int spin; };

template <int THREADS, int BIGREG, typename A>
__global__ __launch_bounds__(THREADS) void probe(const A a) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  extern __shared__ char smem[];
  if (BIGREG) asm volatile("v_mov_b32 v160, 0" ::: "v160");      // forces >= 161 VGPRs per lane
  // stay resident for a while (so that all workgroups of the grid are on the chip together)
  unsigned long long t = t0;
  while (t - t0 < (unsigned long long)a.spin) t = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) a.out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t0;
  if (a.spin < 0) smem[threadIdx.x] = 1;
}

template <int THREADS, int BIGREG, typename A>
void run(const char* name, int grid, size_t lds, unsigned long long* dout) {
  auto k = probe<THREADS, BIGREG, A>;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  A a{}; a.out = dout; a.spin = 2000;    // 20 us
  constexpr int W = THREADS / 64;
  std::vector<unsigned long long> h((size_t)grid * 16);
  double worst = 0, mean = 0, grid_span = 0;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemset(dout, 0, h.size() * 8));
    hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    if (rep < 2) continue;
    unsigned long long gmin = ~0ull, gmax = 0;
    double m = 0;
    for (int g = 0; g < grid; ++g) {
      unsigned long long lo = ~0ull, hi = 0;
      for (int w = 0; w < W; ++w) { lo = std::min(lo, h[(size_t)g * 16 + w]); hi = std::max(hi, h[(size_t)g * 16 + w]); }
      worst = std::max(worst, (double)(hi - lo) * 0.01); m += (double)(hi - lo) * 0.01;
      gmin = std::min(gmin, lo); gmax = std::max(gmax, hi);
    }
    mean += m / grid / 3; grid_span = std::max(grid_span, (double)(gmax - gmin) * 0.01);
  }
  printf("%-34s grid %4d x %4d threads  lds %6zu  : first->last wave of a workgroup mean %.2f us (worst %.2f), of the grid %.2f us\n", name, grid, THREADS, lds, mean, worst, grid_span);
}

int main() {
  unsigned long long* dout; CK(hipMalloc(&dout, 4096 * 16 * 8));
  run<768, 1, BigArgs>("768 thr, 161+ vgpr, 576 B args", 256, 112128, dout);
  run<768, 1, BigArgs>("768 thr, 161+ vgpr, 576 B args", 256, 0, dout);
  run<768, 0, BigArgs>("768 thr, few vgpr, 576 B args", 256, 112128, dout);
  run<768, 0, BigArgs>("768 thr, few vgpr, 576 B args", 256, 0, dout);
  run<768, 0, SmallArgs>("768 thr, few vgpr, 16 B args", 256, 0, dout);
  run<768, 1, SmallArgs>("768 thr, 161+ vgpr, 16 B args", 256, 0, dout);
  run<768, 1, SmallArgs>("768 thr, 161+ vgpr, 16 B args", 256, 112128, dout);
  run<768, 1, SmallArgs>("768 thr, 161+ vgpr, 16 B args", 64, 112128, dout);
  run<768, 1, SmallArgs>("768 thr, 161+ vgpr, 16 B args", 8, 112128, dout);
  run<384, 1, SmallArgs>("384 thr, 161+ vgpr, 16 B args", 512, 56064, dout);
  run<384, 1, SmallArgs>("384 thr, 161+ vgpr, 16 B args", 256, 56064, dout);
  run<256, 1, SmallArgs>("256 thr, 161+ vgpr, 16 B args", 512, 58000, dout);
  run<512, 1, SmallArgs>("512 thr, 161+ vgpr, 16 B args", 256, 150000, dout);
  run<1024, 0, SmallArgs>("1024 thr, few vgpr, 16 B args", 256, 0, dout);
  run<256, 0, SmallArgs>("256 thr, few vgpr, 16 B args", 1024, 0, dout);
  return 0;
}
