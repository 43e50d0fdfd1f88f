"""GPU: bench.py prints ONE JSON line with the driver's contract fields plus the roofline / cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_bench_line_contract(device):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--classes", "8",
                          "--cpu-seconds", "1.5", "--no-end-to-end", "--no-sweep"], cwd=REPO, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
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
