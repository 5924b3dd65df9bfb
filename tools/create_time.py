import sys, time
sys.path.insert(0, '.')
import abft_sparse_cg_amd as amd
from abft_sparse_cg_amd import generators
for spec, mode, fmt in (("laplace5:3162,3162", "none", "csr"), ("laplace5:3162,3162", "secded", "csr"), ("random:4194304,24,1", "secded", "csr"), ("powerlaw:2097152,2", "sec7", "coo")):
    t0 = time.perf_counter(); cols, rows, vals, n = generators.generate(spec); t1 = time.perf_counter()
    ctx = amd.HIPContext(mode, fmt)
    t2 = time.perf_counter(); A = ctx.create_matrix(cols, rows, vals, n, len(vals)); ctx.synchronize(); t3 = time.perf_counter()
    print("%-24s %-7s %s generate %.2f s  create_matrix %.2f s  (nnz %d)" % (spec, mode, fmt, t1 - t0, t3 - t2, len(vals)), flush=True)
    ctx.close()
