// matrix_io.cpp -- see matrix_io.h.  Our own reader: the reference pulls in the
// 500-line NIST mmio library for one header function (cg.cpp:355 -> mmio.c:192);
// the dialect it accepts is small enough to state directly.
#include "matrix_io.h"
#include "glibc_rand.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace
{
  struct Entry
  {
    uint32_t col, row;
    double value;
  };

  // reference mm_read_mtx_crd_size (mmio.c:192-220): skip comment lines, take
  // the first line that holds three integers
  bool read_size_line(FILE *f, int *M, int *N, int *nz)
  {
    char line[1100];
    while (fgets(line, sizeof(line), f))
    {
      if (line[0] == '%')
        continue;
      if (sscanf(line, "%d %d %d", M, N, nz) == 3)
        return true;
    }
    return false;
  }
}

extern "C" int abft_load_mtx(const char *path, int num_blocks, int *N, int *block_size, int *nnz,
                             uint32_t **cols, uint32_t **rows, double **vals)
{
  FILE *f = fopen(path, "r");
  if (!f)
    return ABFT_IO_OPEN;

  int width = 0, height = 0, file_nnz = 0;
  if (!read_size_line(f, &width, &height, &file_nnz) || file_nnz < 0)
  {
    fclose(f);
    return ABFT_IO_BAD_DATA;
  }
  if (width != height)
  {
    fclose(f);
    return ABFT_IO_NOT_SQUARE;
  }

  std::vector<Entry> block;
  block.reserve(2 * (size_t)file_nnz);
  for (int i = 0; i < file_nnz; i++)
  {
    int a, b;
    double v;
    if (fscanf(f, "%d %d %lg", &a, &b, &v) != 3)
    {
      fclose(f);
      return ABFT_IO_BAD_DATA;
    }
    // first integer -> column, second -> row (cg.cpp:371-381), 1-based in the file
    Entry e = {(uint32_t)(a - 1), (uint32_t)(b - 1), v};
    block.push_back(e);
    if (e.col != e.row)  // mirror every off-diagonal entry (cg.cpp:385-391)
    {
      Entry m = {e.row, e.col, v};
      block.push_back(m);
    }
  }
  fclose(f);

  std::sort(block.begin(), block.end(), [](const Entry &x, const Entry &y) {
    return x.row != y.row ? x.row < y.row : x.col < y.col;
  });

  const size_t per_block = block.size();
  const size_t total = per_block * (size_t)num_blocks;
  uint32_t *c = (uint32_t *)malloc(std::max<size_t>(total, 1) * sizeof(uint32_t));
  uint32_t *r = (uint32_t *)malloc(std::max<size_t>(total, 1) * sizeof(uint32_t));
  double *v = (double *)malloc(std::max<size_t>(total, 1) * sizeof(double));
  size_t k = 0;
  for (int j = 0; j < num_blocks; j++)  // the block repeated down the diagonal (cg.cpp:402-414)
    for (size_t i = 0; i < per_block; i++, k++)
    {
      c[k] = block[i].col + (uint32_t)j * (uint32_t)width;
      r[k] = block[i].row + (uint32_t)j * (uint32_t)width;
      v[k] = block[i].value;
    }

  *N = width * num_blocks;
  *block_size = width;
  *nnz = (int)total;
  *cols = c;
  *rows = r;
  *vals = v;
  return ABFT_IO_OK;
}

extern "C" void abft_free_triplets(uint32_t *cols, uint32_t *rows, double *vals)
{
  free(cols);
  free(rows);
  free(vals);
}

// out[i] = rand() / RAND_MAX for the first n draws after srand(seed): the
// right-hand side of the reference driver (cg.cpp:66-74), for the Python side
extern "C" void abft_glibc_rand_fill(double *out, int64_t n, unsigned seed)
{
  GlibcRand g(seed);
  for (int64_t i = 0; i < n; i++)
    out[i] = g.next() / 2147483647.0;
}
