"""One rank of the CPU test of bench.py's agreement step (tests/test_host_logic.py): every rank reports an
exit status, all ranks must see all of them.  Started by torch.distributed.run (gloo, no GPU)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rank = int(os.environ["RANK"])
codes = bench.agree_codes(70 if rank == 1 else 0)   # rank 1's job "failed"
again = bench.agree_codes(-6 if rank == 0 else 0)   # a second job on the same group: rank 0 died of a signal
print("AGREE rank %d %s %s" % (rank, codes, again), flush=True)
