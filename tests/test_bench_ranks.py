"""bench.py's N > 1 path end to end on ONE GPU: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2
--backend gloo` — two ranks (each its own process, its own shard handle of the real HIP engine) sharing GPU 0, the
collectives of sharded.py staged through the host because gloo moves host memory.  What the driver's 8-GPU run does with
--backend nccl (RCCL over xGMI) differs only in the transport: the launcher contract (RANK / LOCAL_RANK / WORLD_SIZE),
the shard protocol, the collective fit status, the all-reduced MAE and the one JSON line of rank 0 are all exercised here."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOAD = ["--workload", "syn-scaled:6000:1500:400000", "--k", "40", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-bf16-leg"]


def _line(out):
    return json.loads(out.strip().splitlines()[-1])


def test_two_ranks_on_one_gpu_reproduce_the_single_process_line():
    one = subprocess.run([sys.executable, "bench.py"] + WORKLOAD, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    a = _line(one.stdout)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), "bench.py", "--gpus", "2", "--backend", "gloo"] + WORKLOAD,
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    b = _line(two.stdout)
    assert (a["n_gpus"], b["n_gpus"]) == (1, 2) and b["steps"] == 2 and b["scaling"] == "strong"
    assert abs(a["mae"] - b["mae"]) <= 1e-12          # only the order of the final all-reduce differs
    assert b["value"] > 0 and "roofline" in b and b["config"]["parallelism"].endswith("x2")
    assert len([l for l in two.stdout.strip().splitlines() if l.startswith("{")]) == 1   # rank 0 alone prints the line


def test_one_rank_nccl_process_group_runs_every_collective_of_the_multi_gpu_path():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 --backend nccl --force-process-group`: the RCCL
    branch of bench.py / sharded.py executed on this one MI355X before the driver's 8-GPU run executes it for the first
    time — init_process_group(backend="nccl", device_id=...), all_gather_into_tensor on the library's f64 device arrays,
    the int64 status all-reduce, the f64 (sum, rows, status) all-reduce, barrier, destroy_process_group.  At world 1 there
    is no peer, so the transport itself (xGMI) is not exercised; everything above it is.  Same numbers as the plain run."""
    one = subprocess.run([sys.executable, "bench.py"] + WORKLOAD, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    a = _line(one.stdout)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), "bench.py", "--gpus", "1", "--backend", "nccl", "--force-process-group"] + WORKLOAD,
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    b = _line(run.stdout)
    assert b["n_gpus"] == 1 and "nccl" in b["config"]["collectives"] and "RCCL" in b["config"]["collectives"]
    assert a["mae"] == b["mae"]          # one rank: the all-reduce adds nothing
    assert b["value"] > 0 and "roofline" in b
