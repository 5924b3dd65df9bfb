"""Child process of tests/test_gpu_distributed.py: one rank (world size 1) of the
sharded solver with the real engine over the nccl (RCCL) backend.  Prints one
JSON line with what the parent asserts on."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    mode, flip_index, flip_bit, port = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch
    import torch.distributed as dist
    from _oracle import laplace5, rhs
    from abft_sparse_cg_amd.distributed import HipEngine, ShardedCG
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cols, rows, vals, n = laplace5(48, 48)
    eng = HipEngine(mode, "csr", device=0)
    cg = ShardedCG(eng, cols, rows, vals, [0, n], 0, mode)
    cg.set_rhs(rhs(n, 1))
    if flip_index >= 0:
        eng.inject(cg.A, flip_index, [flip_bit])
    hist = []
    out = {"exit": 0}
    try:
        it, rr = cg.solve(on_iteration=lambda i, r: hist.append(r))
        x = cg.gather_x()
        tot, mx = cg.residual_check()
        out.update(it=it, hist=hist, x=[float(v) for v in x], tot=tot, mx=mx)
    except SystemExit as e:
        out["exit"] = int(e.code)
    out["events"] = [list(e) for e in cg.events]
    sys.stdout.flush()
    print("RESULT " + json.dumps(out))
    sys.stdout.flush()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
