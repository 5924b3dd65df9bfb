// Microbenchmark 3: the L2-served 8-byte gather, by window size, load flavour and loads in
// flight per lane -- what bounds the panel-layout SpMV.
//   hipcc --offload-arch=gfx950 -O3 gather3.hip -o gather3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

template <int F> __device__ __forceinline__ double ld(const double *p) {
  if (F == 1) return __builtin_nontemporal_load(p);
  if (F == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}

template <int K, int F>
__global__ __launch_bounds__(256) void gk(const double *__restrict__ x, const uint32_t *__restrict__ idx,
                                          double *__restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    uint32_t c[K];
#pragma unroll
    for (int k = 0; k < K; k++) c[k] = __builtin_nontemporal_load(idx + i + (size_t)k * n);
    double v[K];
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = ld<F>(x + c[k]);
    double s = 0;
#pragma unroll
    for (int k = 0; k < K; k++) s += v[k];
    out[i] = s;
  }
}

template <int K, int F> float run(const double *x, const uint32_t *idx, double *out, size_t n, int grid) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  gk<K, F><<<grid, 256>>>(x, idx, out, n);
  hipEventRecord(a);
  for (int r = 0; r < 3; r++) gk<K, F><<<grid, 256>>>(x, idx, out, n);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms / 3;
}

int main() {
  const size_t total = (size_t)1 << 26;  // gathers per launch
  std::vector<uint32_t> h(total);
  uint32_t *idx; double *x, *out;
  hipMalloc(&idx, total * 4); hipMalloc(&x, (size_t)(1u << 23) * 8); hipMalloc(&out, total * 8 / 4);
  hipMemset(x, 0, (size_t)(1u << 23) * 8);
  for (uint32_t W : {1u << 10, 1u << 13, 1u << 16, 1u << 17, 1u << 18, 3u << 17, 1u << 19, 1u << 20, 1u << 22}) {
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s % W); }
    hipMemcpy(idx, h.data(), total * 4, hipMemcpyHostToDevice);
    for (int grid : {2048, 8192}) {
      float t[9];
      t[0] = run<4, 0>(x, idx, out, total / 4, grid);
      t[1] = run<4, 1>(x, idx, out, total / 4, grid);
      t[2] = run<4, 2>(x, idx, out, total / 4, grid);
      t[3] = run<8, 0>(x, idx, out, total / 8, grid);
      t[4] = run<8, 2>(x, idx, out, total / 8, grid);
      t[5] = run<16, 0>(x, idx, out, total / 16, grid);
      t[6] = run<16, 2>(x, idx, out, total / 16, grid);
      printf("window %8.3f MB grid %5d  G gathers/s:  K4 plain %6.1f nt %6.1f sc1 %6.1f | K8 plain %6.1f sc1 %6.1f | K16 plain %6.1f sc1 %6.1f\n",
             W * 8.0 / 1e6, grid, total / (t[0] * 1e6), total / (t[1] * 1e6), total / (t[2] * 1e6), total / (t[3] * 1e6),
             total / (t[4] * 1e6), total / (t[5] * 1e6), total / (t[6] * 1e6));
      fflush(stdout);
    }
  }
  return 0;
}
