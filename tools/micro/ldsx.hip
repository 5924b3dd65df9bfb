// Microbenchmark 5: how fast can every CU stream the whole gathered vector through its LDS, panel
// by panel (all CUs of an XCD reading the same panel at about the same time: L2-served), and what do
// random 8-byte reads from the LDS-resident panel cost?  Decides whether an "x in LDS" SpMV can beat
// the L2-served gather (~3 clk per gathered lane in the CU's TA/L1 path).
//   hipcc --offload-arch=gfx950 -O3 ldsx.hip -o ldsx
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// V0: register-staged, single buffer: load panel -> ds_write -> barrier -> G lds gathers/thread -> barrier
// V1: LDS-DMA (global_load_lds_dwordx4), single buffer
// V2: register-staged, double buffer (next panel's loads in flight during the gathers)
template <int V, int THREADS, int PANEL_BYTES, int G>
__global__ __launch_bounds__(THREADS) void sweep(const double *__restrict__ x, size_t n, double *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int PER = PANEL_BYTES / (THREADS * 16);  // 16-byte pieces per thread and panel
  const int npanels = (int)(n * 8 / PANEL_BYTES);
  const unsigned char *src = reinterpret_cast<const unsigned char *>(x);
  double acc = 0.0;
  uint32_t rng = threadIdx.x * 2654435761u + blockIdx.x;
  u32x4 nxt[PER];
  if (V == 2) {
#pragma unroll
    for (int k = 0; k < PER; k++) nxt[k] = *reinterpret_cast<const u32x4 *>(src + (size_t)(k * THREADS + threadIdx.x) * 16);
  }
  for (int c = 0; c < npanels; c++) {
    const unsigned char *p = src + (size_t)c * PANEL_BYTES;
    unsigned char *buf = lds + (V == 2 ? (c & 1) * PANEL_BYTES : 0);
    if (V == 0) {
      u32x4 r[PER];
#pragma unroll
      for (int k = 0; k < PER; k++) r[k] = *reinterpret_cast<const u32x4 *>(p + (size_t)(k * THREADS + threadIdx.x) * 16);
#pragma unroll
      for (int k = 0; k < PER; k++) *reinterpret_cast<u32x4 *>(buf + (size_t)(k * THREADS + threadIdx.x) * 16) = r[k];
    } else if (V == 1) {
#pragma unroll
      for (int k = 0; k < PER; k++) {
        // wave-uniform LDS base; each lane lands its 16 bytes at base + lane * 16
        const uint32_t wave_off = (uint32_t)(k * THREADS + (threadIdx.x & ~63)) * 16;
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const u32x4 *>(p + (size_t)(k * THREADS + threadIdx.x) * 16),
                                         reinterpret_cast<__attribute__((address_space(3))) void *>(
                                             (__attribute__((address_space(3))) unsigned char *)buf + wave_off),
                                         16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int k = 0; k < PER; k++) *reinterpret_cast<u32x4 *>(buf + (size_t)(k * THREADS + threadIdx.x) * 16) = nxt[k];
      if (c + 1 < npanels) {
        const unsigned char *q = p + PANEL_BYTES;
#pragma unroll
        for (int k = 0; k < PER; k++) nxt[k] = *reinterpret_cast<const u32x4 *>(q + (size_t)(k * THREADS + threadIdx.x) * 16);
      }
    }
    __syncthreads();
    const double *xp = reinterpret_cast<const double *>(buf);
#pragma unroll
    for (int g = 0; g < G; g++) {
      rng = rng * 1664525u + 1013904223u;
      acc += xp[(rng >> 8) % (PANEL_BYTES / 8)];
    }
    if (V != 2) __syncthreads();
  }
  out[(size_t)blockIdx.x * THREADS + threadIdx.x] = acc;
}

template <int V, int THREADS, int PANEL_BYTES, int G>
void run(const char *name, const double *x, size_t n, double *out, int grid) {
  const size_t lds = (V == 2 ? 2 : 1) * (size_t)PANEL_BYTES;
  hipFuncSetAttribute((const void *)sweep<V, THREADS, PANEL_BYTES, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  sweep<V, THREADS, PANEL_BYTES, G><<<grid, THREADS, lds>>>(x, n, out);
  hipEventRecord(a);
  for (int r = 0; r < 3; r++) sweep<V, THREADS, PANEL_BYTES, G><<<grid, THREADS, lds>>>(x, n, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
  hipError_t e = hipGetLastError();
  const double bytes = (double)n * 8 * grid;
  printf("%-46s grid %4d: %8.1f us per sweep of x, %6.1f GB/s per block, %6.2f TB/s chip, lds gathers %.0f G/s  %s\n", name, grid,
         ms * 1e3, n * 8 / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12,
         (double)G * THREADS * grid * (n * 8 / PANEL_BYTES) / (ms * 1e-3) / 1e9, e == hipSuccess ? "" : hipGetErrorString(e));
  fflush(stdout);
  hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
  const size_t n = (size_t)1 << 22;  // config 4's vector: 33.5 MB
  double *x, *out;
  hipMalloc(&x, n * 8 + (1 << 20)); hipMalloc(&out, (size_t)2048 * 1024 * 8);
  hipMemset(x, 0, n * 8 + (1 << 20));
  run<0, 1024, 131072, 0>("V0 regs, 1024 thr, 128 KB panel, no gathers", x, n, out, 256);
  run<0, 1024, 131072, 2>("V0 regs, 1024 thr, 128 KB panel, 2 gathers", x, n, out, 256);
  run<0, 1024, 131072, 8>("V0 regs, 1024 thr, 128 KB panel, 8 gathers", x, n, out, 256);
  run<1, 1024, 131072, 0>("V1 dma,  1024 thr, 128 KB panel, no gathers", x, n, out, 256);
  run<1, 1024, 131072, 2>("V1 dma,  1024 thr, 128 KB panel, 2 gathers", x, n, out, 256);
  run<2, 1024, 65536, 0>("V2 regs dbl, 1024 thr, 64 KB panel, no gathers", x, n, out, 256);
  run<2, 1024, 65536, 1>("V2 regs dbl, 1024 thr, 64 KB panel, 1 gather", x, n, out, 256);
  run<2, 1024, 65536, 4>("V2 regs dbl, 1024 thr, 64 KB panel, 4 gathers", x, n, out, 256);
  run<0, 512, 65536, 1>("V0 regs, 512 thr, 64 KB panel, 1 gather, 2/CU", x, n, out, 512);
  run<1, 512, 65536, 1>("V1 dma,  512 thr, 64 KB panel, 1 gather, 2/CU", x, n, out, 512);
  run<0, 256, 32768, 1>("V0 regs, 256 thr, 32 KB panel, 1 gather, 4/CU", x, n, out, 1024);
  run<1, 256, 32768, 1>("V1 dma,  256 thr, 32 KB panel, 1 gather, 4/CU", x, n, out, 1024);
  return 0;
}
