"""Child process of tests/test_gpu_distributed.py: one rank of the sharded solver
with the real engine (HipEngine -> libabft_hip.so) on cuda:0.

  nccl  world size 1 over RCCL: device-memory aliasing, shared stream, collectives;
  gloo  world size W with every rank on the SAME GPU and host-staged collectives:
        the multi-rank shard geometry (padded columns, halo windows / all-gather,
        global event indices, fatal stop) with the real kernels.
Rank 0 prints one "RESULT <json>" line with what the parent asserts on."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    backend, rank, world, matrix, mode, flip_index, flip_bit, port = sys.argv[1:9]
    fixed = int(sys.argv[9]) if len(sys.argv) > 9 else 0
    rank, world, flip_index, flip_bit = int(rank), int(world), int(flip_index), int(flip_bit)
    import torch
    import torch.distributed as dist
    from _oracle import laplace5, random_spd, rhs
    from abft_sparse_cg_amd.distributed import HipEngine, ShardedCG
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    os.environ.setdefault("ABFT_CG_OVERLAP_BYTES", "0")  # exercise the interior / boundary split at any size
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cols, rows, vals, n = laplace5(48, 48) if matrix == "laplace" else random_spd(1500, 10, seed=3)
    counts = np.bincount(rows, minlength=n)
    cum = np.cumsum(counts)
    bounds = [0] + [int(np.searchsorted(cum, cum[-1] * g / world)) for g in range(1, world)] + [n]
    r0, r1 = bounds[rank], bounds[rank + 1]
    m = (rows >= r0) & (rows < r1)
    before = int(np.argmax(m)) if m.any() else 0
    eng = HipEngine(mode, "csr", device=0)
    cg = ShardedCG(eng, cols[m], rows[m], vals[m], bounds, before, mode, staged=(backend == "gloo"))
    cg.set_rhs(rhs(n, 1)[r0:r1])
    if flip_index >= 0 and before <= flip_index < before + int(m.sum()):
        eng.inject(cg.A, flip_index - before, [flip_bit])
    hist = []
    out = {"exit": 0, "windows": bool(cg.use_windows)}
    try:
        if fixed:
            it, hist = fixed, [cg.run_fixed(fixed)]
        else:
            it, rr = cg.solve(on_iteration=lambda i, r: hist.append(r))
        x = cg.gather_x()
        tot, mx = cg.residual_check()
        out.update(it=it, hist=hist, x=[float(v) for v in x], tot=tot, mx=mx)
    except SystemExit as e:
        out["exit"] = int(e.code)
    out["events"] = [list(e) for e in cg.events]
    out["graph"] = bool(cg._graph)
    sys.stdout.flush()
    if rank == 0:
        print("RESULT " + json.dumps(out))
        sys.stdout.flush()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
