"""GPU: the class-sharded head over RCCL (``torch.distributed`` backend "nccl") around the real HIP kernels.

World size 1 exercises the whole N>1 code path on a single-GPU box (process group, the head writing straight into the
all-gather send buffer, the asynchronous gather on RCCL's stream, the per-level-stream runner, the detection gather);
world size 2 runs wherever two GPUs are visible (the round-end 8-GPU node) and is skipped otherwise.  The checks live in
tests/dist_worker.py, launched with torchrun exactly like ``bench.py --gpus N``."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [1, 2])
def test_class_sharded_head_over_rccl(world, device):
    if torch.cuda.device_count() < world:
        pytest.skip("needs {} GPUs, {} visible".format(world, torch.cuda.device_count()))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "tests", "dist_worker.py")]
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    for r in range(world):
        assert "DIST_WORKER_OK rank={} world={}".format(r, world) in out.stdout, out.stdout[-3000:]


@pytest.mark.parametrize("extra", [["--pyramid", "--classes-total", "16"], ["--classes-total", "16", "--gather", "scores"]])
def test_bench_distributed_path_runs_on_one_rank(extra, device):
    """``bench.py --force-dist`` = the N>1 timed path (RCCL process group, sharded head, gather, optional pyramid runner)
    with one rank: the configuration that crashed in round 1 (pyramid + scores gather) must produce its JSON line."""
    import json
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--force-dist", "--steps", "2", "--warmup", "1",
           "--no-sweep", "--no-cpu-baseline", "--no-end-to-end", "--no-other-precision"] + extra
    env = dict(os.environ, MASTER_PORT=str(_free_port()))
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["classes_total"] == 16
    # what makes a first real N > 1 run diagnosable: per-rank class counts, the collective timeout, a
    # standalone all-gather timing and - where the step gathers asynchronously - how long it actually waited for it
    assert d["config"]["classes_per_gpu"] == [16] and d["config"]["dist_timeout_s"] > 0
    probe = d["allgather_probe"]
    assert probe["bytes_per_rank"] == 32 << 20 and probe["ms"] > 0 and probe["algbw_gbps"] > 0
    if "--pyramid" not in extra:
        assert d["gather_wait_ms"] >= 0.0
