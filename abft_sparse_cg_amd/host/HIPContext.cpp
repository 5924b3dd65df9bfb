// HIPContext.cpp -- see HIPContext.h.  Shared by cg-csr and cg-coo.
#include "HIPContext.h"

#include <cstdio>
#include <cstdlib>

HIPContextBase::HIPContextBase(int format, int mode)
  : ctx_(NULL), format_(format), mode_(mode)
{
  int device = 0;
  if (const char *env = getenv("ABFT_HIP_DEVICE"))
    device = atoi(env);
  check(abft_hip_init(device, &ctx_), "abft_hip_init");
}

HIPContextBase::~HIPContextBase()
{
  if (ctx_)
  {
    report_events(true);
    abft_hip_shutdown(ctx_);
  }
}

void HIPContextBase::check(int rc, const char *what)
{
  if (rc == ABFT_OK)
    return;
  fflush(stdout);
  fprintf(stderr, "hip backend: %s failed (%d): %s\n", what, rc, abft_hip_last_error());
  exit(2);
}

// Print queued events the way the reference prints them while it runs, and
// stop like it does on a fatal one.
void HIPContextBase::report_events(bool force)
{
  if (!force && abft_hip_pending_events(ctx_) == 0)
    return;
  static abft_event events[4096];
  int count = 0, fatal = 0;
  check(abft_hip_drain_events(ctx_, events, 4096, &count, &fatal), "abft_hip_drain_events");
  char line[160];
  for (int i = 0; i < count; i++)
  {
    abft_format_event(&events[i], line, sizeof(line));
    fputs(line, stdout);
  }
  if (fatal)
  {
    fflush(stdout);
    exit(1);
  }
}

cg_matrix* HIPContextBase::create_matrix(const uint32_t *columns, const uint32_t *rows,
                                         const double *values, int N, int nnz)
{
  cg_matrix *M = new cg_matrix;
  M->N = N;
  M->nnz = nnz;
  M->handle = NULL;
  check(abft_hip_matrix_create_shard(ctx_, format_, mode_, columns, rows, values, N, N, nnz, 0,
                                     &M->handle),
        "abft_hip_matrix_create");
  return M;
}

void HIPContextBase::destroy_matrix(cg_matrix *mat)
{
  report_events(true);
  check(abft_hip_matrix_destroy(mat->handle), "abft_hip_matrix_destroy");
  delete mat;
}

cg_vector* HIPContextBase::create_vector(int N)
{
  cg_vector *v = new cg_vector;
  v->N = N;
  v->handle = NULL;
  check(abft_hip_vector_create(ctx_, N, &v->handle), "abft_hip_vector_create");
  return v;
}

void HIPContextBase::destroy_vector(cg_vector *vec)
{
  check(abft_hip_vector_destroy(vec->handle), "abft_hip_vector_destroy");
  delete vec;
}

double* HIPContextBase::map_vector(cg_vector *v)
{
  double *host = NULL;
  check(abft_hip_vector_map(v->handle, &host), "abft_hip_vector_map");
  report_events(true);  // map synchronises: anything an earlier spmv queued is visible now
  return host;
}

void HIPContextBase::unmap_vector(cg_vector *v, double *h)
{
  check(abft_hip_vector_unmap(v->handle, h), "abft_hip_vector_unmap");
}

void HIPContextBase::copy_vector(cg_vector *dst, const cg_vector *src)
{
  check(abft_hip_vector_copy(dst->handle, src->handle), "abft_hip_vector_copy");
}

double HIPContextBase::dot(const cg_vector *a, const cg_vector *b)
{
  double result = 0.0;
  check(abft_hip_dot(ctx_, a->handle, b->handle, &result), "abft_hip_dot");
  report_events(false);
  return result;
}

double HIPContextBase::calc_xr(cg_vector *x, cg_vector *r, const cg_vector *p, const cg_vector *w,
                               double alpha)
{
  double result = 0.0;
  check(abft_hip_calc_xr(ctx_, x->handle, r->handle, p->handle, w->handle, alpha, &result),
        "abft_hip_calc_xr");
  report_events(false);
  return result;
}

void HIPContextBase::calc_p(cg_vector *p, const cg_vector *r, double beta)
{
  check(abft_hip_calc_p(ctx_, p->handle, r->handle, beta), "abft_hip_calc_p");
}

void HIPContextBase::spmv(const cg_matrix *mat, const cg_vector *vec, cg_vector *result)
{
  check(abft_hip_spmv(ctx_, mat->handle, vec->handle, result->handle), "abft_hip_spmv");
}

// The host half of inject_bitflip: the same 1 + num_flips rand() draws, bit
// ranges and printf lines as the reference (CSR/CPUContext.cpp:135-159,
// COO/CPUContext.cpp:123-140); the XOR itself happens on the device.
void HIPContextBase::inject_bitflip(cg_matrix *mat, BitFlipKind kind, int num_flips)
{
  int index = rand() % mat->nnz;

  int first = 0, width;
  if (format_ == ABFT_FMT_CSR)
  {
    width = 96;                                // [0,64) value, [64,96) column
    if (kind == VALUE) width = 64;
    else if (kind == INDEX) { first = 64; width = 32; }
  }
  else
  {
    width = 128;                               // [0,64) col+row, [64,128) value
    if (kind == VALUE) { first = 64; width = 64; }
    else if (kind == INDEX) width = 64;
  }

  for (int i = 0; i < num_flips; i++)
  {
    int bit = (rand() % width) + first;
    printf("*** flipping bit %d at index %d ***\n", bit, index);
    check(abft_hip_inject(mat->handle, (uint32_t)index, &bit, 1), "abft_hip_inject");
  }
}
