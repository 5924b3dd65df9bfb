// matrix_io.h -- Matrix-Market input in the reference driver's dialect, and the
// synthetic generators (generators.cpp), behind one small C ABI shared by the
// C++ drivers and the Python side (libabft_host.so).
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ABFT_IO_OK = 0, ABFT_IO_OPEN = 1, ABFT_IO_NOT_SQUARE = 2, ABFT_IO_BAD_DATA = 3 };

// Load `path` the way the reference does (cg.cpp:342-418): header lines starting
// with '%' are skipped (the banner is not validated), then "M N nnz" with M == N,
// then nnz lines "%d %d %lg", 1-based, first integer = column, second = row;
// every off-diagonal entry is mirrored (so the file must hold one triangle);
// the block is sorted by (row, col) and repeated num_blocks times down the
// diagonal.  On success the three arrays are malloc()ed (free with
// abft_free_triplets) and hold *nnz COO triplets of an *N x *N matrix.
int abft_load_mtx(const char *path, int num_blocks, int *N, int *block_size, int *nnz,
                  uint32_t **cols, uint32_t **rows, double **vals);
void abft_free_triplets(uint32_t *cols, uint32_t *rows, double *vals);

void abft_glibc_rand_fill(double *out, int64_t n, unsigned seed);

// generators.cpp
int64_t abft_gen_dim(const char *spec);
int64_t abft_gen_count(const char *spec, int64_t row0, int64_t row1, int64_t *row_nnz);
int64_t abft_gen_fill(const char *spec, int64_t row0, int64_t row1, uint32_t *cols, uint32_t *rows,
                      double *vals);
int abft_gen_partition(const char *spec, int parts, int64_t *bounds);

#ifdef __cplusplus
}
#endif
