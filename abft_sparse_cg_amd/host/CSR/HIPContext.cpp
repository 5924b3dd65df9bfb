// CSR/HIPContext.cpp -- registers the hip target of cg-csr (see ../HIPContext.h).
#include "HIPContext.h"

ABFT_REGISTER_HIP_CONTEXTS(ABFT_FMT_CSR)
