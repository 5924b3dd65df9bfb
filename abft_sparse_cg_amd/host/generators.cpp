// generators.cpp -- deterministic in-memory synthetic matrices:
// the inputs of BASELINE.json's configs, produced as the sorted symmetric COO
// triplets the reference loader hands to create_matrix (cg.cpp:394-418), without
// going through a multi-GB Matrix-Market text file.  Any row range can be
// generated on its own, so each rank of a row-partitioned run builds only its
// shard.  Shared by the cg-csr/cg-coo drivers (--synthetic) and the Python side
// (ctypes), so `-t hip` and the CPU checkers see identical inputs.
//
//   laplace5:NX,NY      5-point Laplacian on an NX x NY grid, natural ordering:
//                       diagonal 4, off-diagonals -1 (SPD).  N = NX*NY.
//   random:N,K,SEED     N a power of two.  K/2 invertible bit-mixing maps f_k on
//                       [0,N); row i is linked to f_k(i) and f_k^-1(i), so the
//                       pattern is symmetric, scattered over the whole vector,
//                       and every row has <= K off-diagonals.  Values are a hash
//                       of the unordered pair in (-1,0); diagonal = 1 + sum|off|
//                       (strictly diagonally dominant => SPD).
//   powerlaw:N,SEED     as random, with 64 maps of which a row activates 4, 16 or
//                       64 by a hash of its index (1/64 of the rows are hubs):
//                       irregular row lengths for the COO configuration.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct Spec {
  int kind = -1;  // 0 laplace5, 1 random, 2 powerlaw
  int64_t nx = 0, ny = 0, n = 0;
  int k = 0, bits = 0;
  uint64_t seed = 0;
};

bool parse(const char *s, Spec &sp) {
  long long a = 0, b = 0, c = 0;
  if (sscanf(s, "laplace5:%lld,%lld", &a, &b) == 2 && a > 0 && b > 0) {
    sp.kind = 0; sp.nx = a; sp.ny = b; sp.n = a * b;
    return sp.n <= 0x7FFFFFFF;
  }
  if (sscanf(s, "random:%lld,%lld,%lld", &a, &b, &c) == 3 && a > 1 && b >= 0) {
    sp.kind = 1; sp.n = a; sp.k = (int)b; sp.seed = (uint64_t)c;
  } else if (sscanf(s, "powerlaw:%lld,%lld", &a, &b) == 2 && a > 1) {
    sp.kind = 2; sp.n = a; sp.k = 128; sp.seed = (uint64_t)b;
  } else {
    return false;
  }
  if (sp.n & (sp.n - 1)) return false;  // power of two
  while ((1LL << sp.bits) < sp.n) sp.bits++;
  return sp.bits >= 4 && sp.bits <= 31 && sp.k <= 128;
}

inline uint64_t splitmix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// inverse of an odd multiplier modulo 2^64 (Newton)
inline uint64_t inv_odd(uint64_t a) {
  uint64_t x = a;
  for (int i = 0; i < 6; i++) x *= 2 - a * x;
  return x;
}

// One invertible mixing map on `bits`-bit integers: xor key, multiply by an odd
// constant, xorshift by half the width, multiply again.
struct Map {
  uint64_t key, m1, m2, i1, i2, mask;
  int sh;
  void init(uint64_t seed, int k, int bits) {
    mask = (1ull << bits) - 1;
    sh = bits / 2;
    key = splitmix(seed * 1315423911ull + 3 * k) & mask;
    m1 = (splitmix(seed + 7919ull * k + 1) | 1ull) & mask;
    m2 = (splitmix(seed + 104729ull * k + 2) | 1ull) & mask;
    i1 = inv_odd(m1) & mask;
    i2 = inv_odd(m2) & mask;
  }
  inline uint64_t fwd(uint64_t x) const {
    x = ((x ^ key) * m1) & mask;
    x ^= x >> sh;
    return (x * m2) & mask;
  }
  inline uint64_t inv(uint64_t y) const {
    uint64_t x = (y * i2) & mask;
    // undo x ^= x >> sh (sh >= bits/2, so two rounds recover every bit)
    uint64_t t = x ^ (x >> sh);
    t = x ^ (t >> sh);
    x = (t * i1) & mask;
    return x ^ key;
  }
};

inline double pair_value(uint64_t seed, uint64_t i, uint64_t j) {
  const uint64_t lo = i < j ? i : j, hi = i < j ? j : i;
  const uint64_t h = splitmix(splitmix(seed ^ (lo * 0x9E3779B97F4A7C15ull)) ^ hi);
  return -((double)((h >> 11) + 1) / 9007199254740994.0);  // in (-1, 0)
}

// number of maps row i drives (powerlaw only)
inline int active_maps(const Spec &sp, uint64_t i) {
  if (sp.kind == 1) return sp.k / 2;
  const uint64_t h = splitmix(sp.seed ^ (i * 0xD1B54A32D192ED03ull)) & 63u;
  return h == 0 ? 64 : (h < 5 ? 16 : 4);
}

struct RandomGen {
  Spec sp;
  std::vector<Map> maps;
  explicit RandomGen(const Spec &s) : sp(s) {
    const int nmaps = s.kind == 1 ? s.k / 2 : 64;
    maps.resize(nmaps);
    for (int k = 0; k < nmaps; k++) maps[k].init(s.seed, k, s.bits);
  }
  // sorted, de-duplicated off-diagonal columns of row i into out; returns count
  int row_cols(uint64_t i, uint32_t *out) const {
    int n = 0;
    const int mine = active_maps(sp, i);
    for (int k = 0; k < (int)maps.size(); k++) {
      if (k < mine) {
        const uint64_t j = maps[k].fwd(i);
        if (j != i) out[n++] = (uint32_t)j;
      }
      const uint64_t src = maps[k].inv(i);  // rows that point at i through map k
      if (src != i && k < active_maps(sp, src)) out[n++] = (uint32_t)src;
    }
    std::sort(out, out + n);
    return (int)(std::unique(out, out + n) - out);
  }
};

int64_t laplace_row_count(const Spec &sp, int64_t i) {
  const int64_t ix = i % sp.nx, iy = i / sp.nx;
  return 1 + (ix > 0) + (ix + 1 < sp.nx) + (iy > 0) + (iy + 1 < sp.ny);
}

}  // namespace

extern "C" {

// -> N (matrix dimension), or -1 if the spec does not parse
int64_t abft_gen_dim(const char *spec) {
  Spec sp;
  return parse(spec, sp) ? sp.n : -1;
}

// row_nnz[r - row0] for r in [row0,row1); returns the total, or -1
int64_t abft_gen_count(const char *spec, int64_t row0, int64_t row1, int64_t *row_nnz) {
  Spec sp;
  if (!parse(spec, sp) || row0 < 0 || row1 > sp.n || row0 > row1) return -1;
  int64_t total = 0;
  if (sp.kind == 0) {
    for (int64_t i = row0; i < row1; i++) {
      const int64_t c = laplace_row_count(sp, i);
      if (row_nnz) row_nnz[i - row0] = c;
      total += c;
    }
    return total;
  }
  RandomGen g(sp);
#pragma omp parallel for reduction(+ : total) schedule(static)
  for (int64_t i = row0; i < row1; i++) {
    uint32_t tmp[260];
    const int64_t c = 1 + g.row_cols((uint64_t)i, tmp);
    if (row_nnz) row_nnz[i - row0] = c;
    total += c;
  }
  return total;
}

// Fill the triplets of rows [row0,row1), sorted by (row, col), global indices.
// Arrays must hold abft_gen_count(...) entries.  Returns the count, or -1.
int64_t abft_gen_fill(const char *spec, int64_t row0, int64_t row1, uint32_t *cols, uint32_t *rows,
                      double *vals) {
  Spec sp;
  if (!parse(spec, sp) || row0 < 0 || row1 > sp.n || row0 > row1) return -1;
  const int64_t nr = row1 - row0;
  std::vector<int64_t> start((size_t)nr + 1, 0);
  if (abft_gen_count(spec, row0, row1, start.data() + 1) < 0) return -1;
  for (int64_t r = 0; r < nr; r++) start[r + 1] += start[r];
  if (sp.kind == 0) {
#pragma omp parallel for schedule(static)
    for (int64_t i = row0; i < row1; i++) {
      int64_t p = start[i - row0];
      const int64_t ix = i % sp.nx, iy = i / sp.nx;
      auto put = [&](int64_t c, double v) { cols[p] = (uint32_t)c; rows[p] = (uint32_t)i; vals[p] = v; p++; };
      if (iy > 0) put(i - sp.nx, -1.0);
      if (ix > 0) put(i - 1, -1.0);
      put(i, 4.0);
      if (ix + 1 < sp.nx) put(i + 1, -1.0);
      if (iy + 1 < sp.ny) put(i + sp.nx, -1.0);
    }
    return start[nr];
  }
  RandomGen g(sp);
#pragma omp parallel for schedule(static)
  for (int64_t i = row0; i < row1; i++) {
    uint32_t tmp[260];
    const int n = g.row_cols((uint64_t)i, tmp);
    int64_t p = start[i - row0];
    double absum = 0.0;
    for (int k = 0; k < n; k++) absum += -pair_value(sp.seed, (uint64_t)i, tmp[k]);
    bool diag_done = false;
    for (int k = 0; k <= n; k++) {
      if (!diag_done && (k == n || tmp[k] > (uint32_t)i)) {
        cols[p] = (uint32_t)i; rows[p] = (uint32_t)i; vals[p] = 1.0 + absum; p++;
        diag_done = true;
      }
      if (k < n) {
        cols[p] = tmp[k]; rows[p] = (uint32_t)i; vals[p] = pair_value(sp.seed, (uint64_t)i, tmp[k]); p++;
      }
    }
  }
  return start[nr];
}

// Cut [0,N) into `parts` contiguous row ranges of (nearly) equal nnz
// (balance by nnz, not rows).  bounds gets parts+1 entries.
int abft_gen_partition(const char *spec, int parts, int64_t *bounds) {
  Spec sp;
  if (!parse(spec, sp) || parts < 1) return -1;
  std::vector<int64_t> cnt((size_t)sp.n);
  const int64_t total = abft_gen_count(spec, 0, sp.n, cnt.data());
  if (total < 0) return -1;
  bounds[0] = 0;
  int64_t acc = 0, r = 0;
  for (int p = 1; p < parts; p++) {
    const int64_t target = total * p / parts;
    while (r < sp.n && acc + cnt[r] <= target) acc += cnt[r++];
    bounds[p] = r;
  }
  bounds[parts] = sp.n;
  return 0;
}

}  // extern "C"
