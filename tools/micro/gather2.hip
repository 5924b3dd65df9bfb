// Microbenchmark 2: from the bare gather towards the SpMV load phase, one step at a time.
//   V1  idx stream + gather                      (tools/micro/gather.hip)
//   V2  + 8-byte value stream, multiply, per-thread sum
//   V3  + products through LDS, workgroup barrier, each thread reads 4 staged products back
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

template <int V>
__global__ __launch_bounds__(256) void k(const double* __restrict__ x, const uint32_t* __restrict__ idx,
                                         const double* __restrict__ val, double* __restrict__ out, size_t n) {
  __shared__ double s[1024];
  const size_t stride = (size_t)gridDim.x * 1024;
  for (size_t base = (size_t)blockIdx.x * 1024; base < n; base += stride) {
    uint32_t c[4]; double v[4], g[4];
#pragma unroll
    for (int q = 0; q < 4; q++) c[q] = __builtin_nontemporal_load(idx + base + threadIdx.x + q * 256);
    if (V >= 2) {
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = __builtin_nontemporal_load(val + base + threadIdx.x + q * 256);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) g[q] = x[c[q]];
    double acc = 0;
    if (V == 1) { acc = (g[0] + g[1]) + (g[2] + g[3]); }
    if (V == 2) { acc = (v[0] * g[0] + v[1] * g[1]) + (v[2] * g[2] + v[3] * g[3]); }
    if (V == 3) {
#pragma unroll
      for (int q = 0; q < 4; q++) s[threadIdx.x + q * 256] = v[q] * g[q];
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; q++) acc += s[(threadIdx.x * 4 + q + 5) & 1023];
      __syncthreads();
    }
    out[base / 4 + threadIdx.x] = acc;
  }
}

int main() {
  const size_t n = (size_t)1 << 26;
  const uint32_t W = 1u << 18;  // 2 MB window: L2-resident
  std::vector<uint32_t> h(n);
  uint64_t s = 88172645463325252ull;
  for (auto& e : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; e = (uint32_t)(s % W); }
  uint32_t* idx; double *x, *val, *out;
  hipMalloc(&idx, n * 4); hipMalloc(&val, n * 8); hipMalloc(&x, (size_t)W * 8); hipMalloc(&out, n * 2);
  hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(x, 0, (size_t)W * 8); hipMemset(val, 0, n * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int grid : {2048, 16384, 65536}) {
    float ms[4];
    for (int v = 1; v <= 3; v++) {
      auto launch = [&]() {
        if (v == 1) k<1><<<grid, 256>>>(x, idx, val, out, n);
        if (v == 2) k<2><<<grid, 256>>>(x, idx, val, out, n);
        if (v == 3) k<3><<<grid, 256>>>(x, idx, val, out, n);
      };
      launch();
      hipEventRecord(a);
      for (int r = 0; r < 5; r++) launch();
      hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms[v], a, b);
      printf("grid %6d V%d: %7.1f G elements/s\n", grid, v, 5.0 * n / (ms[v] * 1e-3) / 1e9);
    }
  }
  return 0;
}
