"""CPU suite, part 3: the N > 1 paths with world_size 2 over gloo: problem sharding + result gather (batched mode) and
the all-reduced bundle of the row-sharded Gram-space recursion."""
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


ROW_WORKER = r'''
import ctypes as C, os, sys
import numpy as np
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
hl = C.CDLL(%(lib)r)
n, m, K = 602, 5, 9
rng = np.random.default_rng(11)                       # every rank builds the same full problem ...
a = 1.0 + 9.0 * rng.random(n)
G = np.zeros((K + 1, n)); S = np.zeros((K, n))
x = rng.standard_normal(n); G[0] = a * x
for k in range(K):
    S[k] = -0.3 * G[k] / a * (1 + 0.2 * rng.random(n)); x = x + S[k]; G[k + 1] = a * x
accept = np.ones(K, dtype=np.uint8); accept[4] = 0
per = (n // world) // 4 * 4                           # ... and keeps the row block bench.py --workload sharded gives it
lo = rank * per; hi = n if rank == world - 1 else lo + per
nred = [0]

@C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int)
def reduce(ptr, k):
    v = np.ctypeslib.as_array(ptr, shape=(k,))
    dist.all_reduce(torch.from_numpy(v))              # in place on the callback's memory, as bench.py does over gloo
    nred[0] += 1

def run(fn, Sx, Gx, *extra):
    coef = np.zeros(2 * m); cg = C.c_double(); slots = np.zeros(m, dtype=np.int32)
    Sx = np.ascontiguousarray(Sx); Gx = np.ascontiguousarray(Gx)
    fn.restype = C.c_int
    ptr = fn(Sx.shape[1], m, K, Sx.ctypes.data_as(C.c_void_p), Gx.ctypes.data_as(C.c_void_p),
             accept.ctypes.data_as(C.c_void_p), coef.ctypes.data_as(C.c_void_p), C.byref(cg),
             slots.ctypes.data_as(C.c_void_p), *extra)
    assert ptr >= 0
    return np.concatenate([coef, [cg.value]]), slots

full, slots_full = run(hl.hl_gram_space, S, G)
mine, slots_mine = run(hl.hl_gram_space_sharded, S[:, lo:hi], G[:, lo:hi], reduce)
ok = nred[0] == K + 1                                  # one bundle per iteration (+ the initial g.g)
ok = ok and np.array_equal(slots_full, slots_mine)
ok = ok and np.allclose(mine, full, rtol=1e-9, atol=1e-12)   # the sums only differ in their association
both = [torch.zeros(2 * m + 1, dtype=torch.float64) for _ in range(world)]
dist.all_gather(both, torch.from_numpy(mine))
ok = ok and all(torch.equal(both[0], b) for b in both)       # every rank computes the same coefficients, bit for bit
print("RANK%%d %%s" %% (rank, "OK" if ok else "BAD %%r %%r" %% (mine, full)))
dist.destroy_process_group()
'''


def test_row_sharded_bundle_over_gloo(tmp_path):
    """Row-sharded Gram-space recursion: GramSpaceHistory driven as LBFGSSolver::run drives it,
    each rank holding a block of rows and every sum over rows all-reduced as one bundle per iteration.  The
    coefficients of the direction must equal the single-process ones to rounding and be identical on all ranks."""
    lib = str(tmp_path / "libhostlogic.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-I",
                           os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_logic_capi.cpp"),
                           "-o", lib])
    script = tmp_path / "row_worker.py"
    script.write_text(ROW_WORKER % {"lib": lib})
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


def test_bench_refuses_to_report_fewer_gpus_than_asked_for():
    """`python bench.py --gpus 2` on a box with fewer devices exits non-zero and prints no result line (it used to parse
    --gpus and ignore it); under a launcher whose world size differs from --gpus it refuses as well."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LBFGSX_BENCH_FORCE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    if r.returncode == 0:   # a box with >= 2 GPUs: then the line must say so
        import json
        assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 2
    else:
        assert not r.stdout.strip() and "refusing" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode != 0 and not r.stdout.strip() and "launcher started 2" in r.stderr
