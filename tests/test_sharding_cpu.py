"""CPU suite, part 3: the N > 1 path (problem sharding + result gather) with world_size 2 over gloo."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from lbfgspp_amd.batched import RECORD, gather_records, shard_range

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
N = 37
first, count = shard_range(N, rank, world)
local = np.zeros(count, dtype=RECORD)
for k in range(count):
    pid = first + k
    local[k] = (10 + pid %% 7, 20 + pid, 0, 0.5 * pid, 1.0 / (1 + pid))
full = gather_records(local, N, rank, world, dist=dist)
ok = all(full["nfev"][p] == 20 + p and full["fx"][p] == 0.5 * p and full["niter"][p] == 10 + p %% 7 for p in range(N))
# timing protocol of bench.py: barrier, then max over ranks
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.barrier()
dist.all_reduce(t, op=dist.ReduceOp.MAX)
ok = ok and t.item() == float(world)
print("RANK%%d %%s" %% (rank, "OK" if ok else "BAD"))
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_partitions_exactly():
    from lbfgspp_amd.batched import shard_range
    for n in (0, 1, 7, 8192, 8193):
        for w in (1, 2, 3, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
            for (f0, c0), (f1, _) in zip(blocks, blocks[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_two_rank_gather_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "RANK%d OK" % rank in o, o
