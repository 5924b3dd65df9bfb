// Microbenchmark 4: does visiting a segment's elements in COLUMN order pay?  Indices uniform in
// a 1 MB window (131072 entries) -- config 4's panel -- drawn in blocks of S and, in the sorted
// arm, sorted ascending inside each block; a wave's 64 lanes take 64 consecutive indices, so the
// sorted arm's lanes share 128-byte lines.  Same index + value streams and products as gather2 V2.
//   hipcc --offload-arch=gfx950 -O3 gather4.hip -o gather4
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int K>
__global__ __launch_bounds__(256) void gk(const double *__restrict__ x, const uint32_t *__restrict__ idx,
                                          const double *__restrict__ val, double *__restrict__ out, size_t n) {
  // thread handles K consecutive chunks: element e = base + k*256 + tid (64 consecutive per wave-instruction)
  const size_t stride = (size_t)gridDim.x * 256 * K;
  for (size_t base = (size_t)blockIdx.x * 256 * K; base < n; base += stride) {
    uint32_t c[K]; double v[K], g[K];
#pragma unroll
    for (int k = 0; k < K; k++) c[k] = __builtin_nontemporal_load(idx + base + k * 256 + threadIdx.x);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = __builtin_nontemporal_load(val + base + k * 256 + threadIdx.x);
#pragma unroll
    for (int k = 0; k < K; k++) g[k] = x[c[k]];
    double acc = 0;
#pragma unroll
    for (int k = 0; k < K; k++) acc += v[k] * g[k];
    out[base / K + threadIdx.x] = acc;
  }
}

int main() {
  const size_t n = (size_t)1 << 26;
  const uint32_t W = 1u << 17;
  std::vector<uint32_t> h(n);
  uint32_t *idx; double *x, *val, *out;
  hipMalloc(&idx, n * 4); hipMalloc(&val, n * 8); hipMalloc(&x, (size_t)W * 8); hipMalloc(&out, n * 8 / 4);
  hipMemset(x, 0, (size_t)W * 8); hipMemset(val, 0, n * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int S : {0, 1024, 2048, 3277, 6554, 13107, 52428}) {
    uint64_t s = 88172645463325252ull;
    for (auto &e : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; e = (uint32_t)(s % W); }
    if (S) for (size_t i = 0; i + S <= n; i += S) std::sort(h.begin() + i, h.begin() + i + S);
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int grid : {1024, 2048}) {
      gk<8><<<grid, 256>>>(x, idx, val, out, n);
      hipEventRecord(a);
      for (int r = 0; r < 3; r++) gk<8><<<grid, 256>>>(x, idx, val, out, n);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("sorted blocks of %6d (0 = random), grid %5d: %7.1f G elements/s\n", S, grid, 3.0 * n / (ms * 1e-3) / 1e9);
      fflush(stdout);
    }
  }
  return 0;
}
