// CGContext.h -- the backend plugin boundary of the cg-csr / cg-coo drivers.
//
// Interface-compatible with the reference's CGContext (reference CGContext.h:8-67):
// the same thirteen pure virtuals, in the same order and with the same signatures
// (so the vtable layout is the same), the same BitFlipKind, create() /
// list_contexts(), and a Register<T> helper that a backend instantiates at file
// scope.  Reference-style backends therefore compile against this header
// unchanged, and HIPContext.cpp drops into the reference tree unchanged
// (INTEGRATION.md).  The implementation behind it is our own: the registry is a
// function-local static vector (no dependence on link order for static
// initialisation) that keeps registration order for --list.
#pragma once
#include <cstdint>

struct cg_matrix;  // defined privately by each backend; opaque to the driver
struct cg_vector;

class CGContext
{
public:
  typedef CGContext *(*Factory)();
  enum BitFlipKind { ANY, VALUE, INDEX };  // which bits of an element -x may hit

  virtual ~CGContext() {}

  // ---- matrix: COO triplets sorted by (row, col) in, backend-private handle out.
  //      The caller frees its three arrays right after the call (cg.cpp:418-422).
  virtual cg_matrix *create_matrix(const uint32_t *columns, const uint32_t *rows,
                                   const double *values, int N, int nnz) = 0;
  virtual void destroy_matrix(cg_matrix *mat) = 0;

  // ---- vectors: contents undefined after create; map/unmap bracket host access.
  virtual cg_vector *create_vector(int N) = 0;
  virtual void destroy_vector(cg_vector *vec) = 0;
  virtual double *map_vector(cg_vector *v) = 0;
  virtual void unmap_vector(cg_vector *v, double *h) = 0;
  virtual void copy_vector(cg_vector *dst, const cg_vector *src) = 0;

  // ---- the four kernels of a CG iteration (cg.cpp:97-112).
  virtual double dot(const cg_vector *a, const cg_vector *b) = 0;
  virtual double calc_xr(cg_vector *x, cg_vector *r,                    // x += alpha p
                         const cg_vector *p, const cg_vector *w,        // r -= alpha w
                         double alpha) = 0;                             // returns r.r
  virtual void calc_p(cg_vector *p, const cg_vector *r, double beta) = 0;  // p = r + beta p
  virtual void spmv(const cg_matrix *mat, const cg_vector *vec, cg_vector *result) = 0;

  // ---- fault injection: num_flips random bits of one random element.
  virtual void inject_bitflip(cg_matrix *mat, BitFlipKind kind, int num_flips) = 0;

  // (target, mode) -> new backend; an unknown pair prints to stderr and exit(1)s.
  static CGContext *create(const char *target, const char *mode);
  // "\t<target>-<mode>" per registered pair, in registration order.
  static void list_contexts();
  // What Register<T> calls; usable directly by backends that construct lazily.
  static void add(const char *target, const char *mode, Factory make);

  // static CGContext::Register<MyContext> reg("target", "mode");   // at file scope
  template <class Backend> class Register
  {
    static CGContext *make() { return new Backend(); }

  public:
    Register(const char *target, const char *mode) { CGContext::add(target, mode, &make); }
  };

protected:
  CGContext() {}
};
