// Microbenchmark: how fast can gfx950 gather 8-byte values at random from a window of x?
//   out[i] = sum_k x[idx[i*K + k]]  with idx uniform in [0, W)
// Built and run by hand (tools/micro): hipcc --offload-arch=gfx950 -O3 gather.hip -o gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

template <int K>
__global__ __launch_bounds__(256) void gather_kernel(const double* __restrict__ x, const uint32_t* __restrict__ idx,
                                                     double* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    uint32_t c[K];
#pragma unroll
    for (int k = 0; k < K; k++) c[k] = __builtin_nontemporal_load(idx + i + (size_t)k * n);
    double v[K];
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = x[c[k]];
    double s = 0;
#pragma unroll
    for (int k = 0; k < K; k++) s += v[k];
    out[i] = s;
  }
}

int main() {
  const size_t n = 1 << 24;  // threads-worth of outputs
  const int K = 4;
  for (uint32_t W : {1u << 13, 1u << 16, 1u << 18, 1u << 19, 1u << 20, 1u << 22, 1u << 24}) {
    std::vector<uint32_t> h((size_t)K * n);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s % W); }
    uint32_t* idx; double *x, *out;
    hipMalloc(&idx, h.size() * 4); hipMalloc(&x, (size_t)W * 8); hipMalloc(&out, n * 8);
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(x, 0, (size_t)W * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int grid : {2048, 8192, 65536}) {
      gather_kernel<K><<<grid, 256>>>(x, idx, out, n);
      hipEventRecord(a);
      for (int r = 0; r < 5; r++) gather_kernel<K><<<grid, 256>>>(x, idx, out, n);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("window %8.2f MB grid %6d: %7.1f G gathers/s  (idx stream %.0f GB/s)\n", W * 8.0 / 1e6, grid,
             5.0 * K * n / (ms * 1e-3) / 1e9, 5.0 * K * n * 4 / (ms * 1e-3) / 1e9);
    }
    hipFree(idx); hipFree(x); hipFree(out);
  }
  return 0;
}
