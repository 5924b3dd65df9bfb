// CGContext.h -- the backend plugin boundary of the cg-csr / cg-coo drivers.
//
// Interface-compatible with the reference's CGContext (reference CGContext.h:8-67):
// the same thirteen pure virtuals in the same order with the same signatures,
// the same BitFlipKind, create()/list_contexts(), and a Register<T> helper that a
// backend instantiates at file scope -- so reference-tree backends compile
// against this header unchanged and CSR/HIPContext.cpp drops into the reference
// tree unchanged.  The implementation is our own: the registry
// is a function-local static (no dependence on link order for static
// initialisation) and keeps registration order for --list.
#pragma once
#include <cstdint>
#include <vector>

// Opaque to the driver; each backend defines them privately.
struct cg_matrix;
struct cg_vector;

class CGContext
{
public:
  enum BitFlipKind {ANY, VALUE, INDEX};

  virtual ~CGContext() {}

  virtual cg_matrix* create_matrix(const uint32_t *columns,
                                   const uint32_t *rows,
                                   const double *values,
                                   int N, int nnz) = 0;
  virtual void       destroy_matrix(cg_matrix *mat) = 0;

  virtual cg_vector* create_vector(int N) = 0;
  virtual void       destroy_vector(cg_vector *vec) = 0;
  virtual double*    map_vector(cg_vector *v) = 0;
  virtual void       unmap_vector(cg_vector *v, double *h) = 0;
  virtual void       copy_vector(cg_vector *dst, const cg_vector *src) = 0;

  virtual double     dot(const cg_vector *a, const cg_vector *b) = 0;
  virtual double     calc_xr(cg_vector *x, cg_vector *r,
                             const cg_vector *p, const cg_vector *w,
                             double alpha) = 0;
  virtual void       calc_p(cg_vector *p, const cg_vector *r, double beta) = 0;
  virtual void       spmv(const cg_matrix *mat, const cg_vector *vec,
                          cg_vector *result) = 0;

  virtual void       inject_bitflip(cg_matrix *mat,
                                    BitFlipKind kind, int num_flips) = 0;

  // Looks (target, mode) up; unknown pair: message on stderr and exit(1).
  static CGContext* create(const char *target, const char *mode);
  // Prints "\t<target>-<mode>" per registered pair, in registration order.
  static void       list_contexts();

  typedef CGContext* (*Factory)();
  static void add(const char *target, const char *mode, Factory make);

  // `static CGContext::Register<MyContext> R("target", "mode");` at file scope
  template<class T>
  class Register
  {
  public:
    Register(const char *target, const char *mode)
    {
      CGContext::add(target, mode, &Register<T>::make);
    }
  private:
    static CGContext* make() { return new T(); }
  };

protected:
  CGContext() {}
};
