#!/usr/bin/env python3
"""Host-side cost of the sharded CG loop, measured at world size 1 over RCCL with
the collectives forced on (ABFT_FORCE_COLLECTIVES=1: same host path as N ranks,
trivial device work), eager enqueue vs hipGraph replay.

    python tools/dist_overhead.py [spec ...]      (default: the 1/8 and the full config-2 matrix)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ABFT_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29731")


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    from abft_sparse_cg_amd import generators
    from abft_sparse_cg_amd.distributed import HipEngine, ShardedCG

    specs = sys.argv[1:] or ["laplace5:1118,1118", "laplace5:3162,3162"]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    for spec in specs:
        cols, rows, vals, n = generators.generate(spec)
        eng = HipEngine("none", "csr", device=0)
        cg = ShardedCG(eng, cols, rows, vals, [0, n], 0, "none")
        b = np.random.default_rng(1).random(n)
        out = {}
        for name, graph in (("eager", False), ("graph", True)):
            cg.set_rhs(b)
            cg.run_fixed(20, graph=graph)
            cg.set_rhs(b)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rr = cg.run_fixed(400, graph=graph)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[name] = (dt / 400 * 1e6, rr)
            print("%-22s %-6s %8.1f us/iteration  rr=%.17g  graph=%s" % (spec, name, dt / 400 * 1e6, rr, cg._graph),
                  flush=True)
        assert out["eager"][1] == out["graph"][1], "graph replay changed the arithmetic"
        eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
