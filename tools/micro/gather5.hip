// Microbenchmark 6: is the L2-served 8-byte gather bound per CU (the CU's address / L1-fill path) or per L2?
// B workgroups of 1024 threads, each holding 96 KB of LDS so that at most ONE fits a CU (the dispatcher deals
// consecutive workgroups to the XCDs round-robin): B = 1 .. 256 CUs busy, each gathering from the same window.
// If the rate per busy CU stays the same from 8 CUs (one per XCD) to 256, the bound is the CU's own path; if it
// falls as an XCD's 32 CUs join in, it is that XCD's L2.
//   hipcc --offload-arch=gfx950 -O3 gather5.hip -o gather5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(1024) void gk(const double *__restrict__ x, const uint32_t *__restrict__ idx,
                                           double *__restrict__ out, uint32_t per_thread, uint32_t n_idx) {
  extern __shared__ double pad[];
  if (threadIdx.x == 2000) pad[0] = 1.0;  // (keeps the LDS allocation)
  const uint32_t t = blockIdx.x * 1024u + threadIdx.x;
  double s = 0.0;
  // the index stream is coalesced (a wave reads 64 consecutive indices per instruction) and small next to the gathers
  const uint32_t nthreads = gridDim.x * 1024u;
  uint32_t i = t % n_idx;
  for (uint32_t k = 0; k < per_thread; k += 8) {
    uint32_t c[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { c[j] = __builtin_nontemporal_load(idx + i); i += nthreads; if (i >= n_idx) i -= n_idx; }
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = x[c[j]];
#pragma unroll
    for (int j = 0; j < 8; j++) s += v[j];
  }
  out[t] = s;
}

int main() {
  const uint32_t n_idx = 1u << 24;
  std::vector<uint32_t> h(n_idx);
  uint32_t *idx; double *x, *out;
  hipMalloc(&idx, (size_t)n_idx * 4); hipMalloc(&x, (size_t)(1u << 23) * 8); hipMalloc(&out, (size_t)256 * 1024 * 8);
  hipMemset(x, 0, (size_t)(1u << 23) * 8);
  hipFuncSetAttribute((const void *)gk, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  for (uint32_t W : {1u << 10, 1u << 17, 1u << 21}) {  // 8 KB (L1), 1 MB (L2), 16 MB (beyond an L2)
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s % W); }
    hipMemcpy(idx, h.data(), (size_t)n_idx * 4, hipMemcpyHostToDevice);
    for (int B : {1, 8, 16, 32, 64, 128, 256}) {
      const uint32_t per_thread = 2048;
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      gk<<<B, 1024, 96 * 1024>>>(x, idx, out, per_thread, n_idx);
      hipEventRecord(a);
      for (int r = 0; r < 3; r++) gk<<<B, 1024, 96 * 1024>>>(x, idx, out, per_thread, n_idx);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
      const double g = (double)B * 1024 * per_thread;
      printf("window %8.3f MB  %3d CUs busy: %7.1f G gathers/s = %5.2f G/s per CU = %.3f lanes per ns per CU\n", W * 8.0 / 1e6, B,
             g / (ms * 1e6), g / (ms * 1e6) / B, g / (ms * 1e6) / B);
      fflush(stdout);
      hipEventDestroy(a); hipEventDestroy(b);
    }
  }
  return 0;
}
