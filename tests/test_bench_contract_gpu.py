"""GPU: bench.py prints ONE JSON line with the driver's contract fields plus the roofline / cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _strict_loads(text):
    def refuse(name):
        raise ValueError("non-finite constant in the bench line: " + name)
    return json.loads(text, parse_constant=refuse)


def _one_line(out):
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("\n") == 1 and out.stdout.startswith("{"), out.stdout[-2000:]     # stdout is the line and nothing else
    line = out.stdout.strip()
    assert len(line) <= 6000, len(line)
    return _strict_loads(line)


def test_default_command_prints_one_compact_line(device):
    """The driver's own command, nothing disabled: one line of at most 6000 bytes of strict JSON with
    every contract key, the live roofline traffic and the cpu_baseline; the full record lands in bench_details.json."""
    details = os.path.join(REPO, "bench_details.json")
    if os.path.exists(details):
        os.remove(details)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"],
                         cwd=REPO, capture_output=True, text=True, timeout=900)
    d = _one_line(out)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "roofline_other", "stages_ms"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
    assert d["config"]["workload"].startswith("BASELINE.json configs[1]") and d["config"]["classes_total"] == 64
    assert abs(d["value"] - 64 * 20 / (d["ms_per_step"] * 20e-3)) / d["value"] < 1e-2
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["avg_launch_ms"] <= d["ms_per_step"]
    assert r["traffic"] is None or r["traffic"] > 0
    assert all(len(v) == 2 for v in d["roofline_other"].values())
    for k in ("classes_256_v1", "classes_1024_one_gpu", "pyramid_7_levels_128_classes"):
        assert d["config"][k]["pairs_per_s"] > 0, k
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    # the strict-fp32 reading has a complete record of its own, the CPU baseline names the host's physical cores
    assert d["cpu_baseline"]["host_physical_cores"] is None or d["cpu_baseline"]["host_physical_cores"] >= 1
    assert d["cpu_baseline"]["host_logical_cpus"] >= d["cpu_baseline"]["cores"]
    rs = d["roofline_strict_fp32"]
    assert rs["precision"] == "fft32" and rs["pairs_per_s"] > 0 and rs["peak"] in (157.3, 8000.0)
    assert abs(rs["frac"] - rs["achieved"] / rs["peak"]) < 1e-3 and rs["avg_launch_ms"] <= rs["ms_per_step"]
    assert len(json.dumps(rs)) <= 400
    full = _strict_loads(open(details).read())
    for k in ("other_precisions", "sweep", "live_counters", "roofline_other", "end_to_end"):
        assert k in full, k
    assert "[bench_details] {" in out.stderr


def test_bench_line_contract(device):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--classes", "8",
                          "--cpu-seconds", "1.5", "--no-end-to-end", "--no-sweep"], cwd=REPO, capture_output=True, text=True, timeout=600)
    d = _one_line(out)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-2
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert (r["bound"], r["unit"]) in (("mfma", "TFLOP/s"), ("hbm", "GB/s")) and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and isinstance(c["sample"], str)


def test_forced_distributed_line_carries_its_own_one_gpu_reference(device):
    """the N > 1 line is self-contained - rank 0 times the SAME workload on one GPU first and the line carries
    ``one_gpu_same_workload`` + ``speedup_vs_one_gpu`` (here: world size 1 under --force-dist, so the speed-up is ~1)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1",
                          "--classes-total", "128", "--no-other-precision", "--no-other-gather"], cwd=REPO, capture_output=True,
                         text=True, timeout=600, env=env)
    d = _one_line(out)
    assert d["scaling"] == "strong" and d["config"]["classes_total"] == 128
    one = d["one_gpu_same_workload"]
    assert one["pairs_per_s"] > 0 and one["ms_per_step"] > 0
    assert abs(d["speedup_vs_one_gpu"] - d["value"] / one["pairs_per_s"]) < 2e-3
    assert 0.5 < d["speedup_vs_one_gpu"] < 1.5
