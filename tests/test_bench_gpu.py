"""bench.py contract: one JSON line with the driver's keys, the roofline object of the dominant kernel and
(at N = 1) the cpu_baseline object.  Short run; the numbers themselves are not asserted."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", *flags],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run("--no-e2e", "--no-train", "--no-density-sweep")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "pairs/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.3 < r["frac"] < 1.0 and "wino_gemm" in r["kernel"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "pairs/s" and c["sample"]
    cv = d["roofline_costvol"]
    assert cv["bound"] == "hbm" and cv["peak"] == 8000.0 and 0 < cv["frac"] < 1.2
    alt = d["alt_wino_gemm_bf16x3"]                    # the opt-in bf16x3 Winograd GEMM, measured in a child process
    assert "error" not in alt, alt
    assert alt["value"] > 0 and alt["wino_gemm_ms"] > 0 and 0 < alt["bf16_pipe_frac"] < 1.0


def test_bench_train_leg_full_size():
    """The config-5 share at full size (4 pairs of 972x540, stages 1-3, forward + backward through the C ABI and
    through SpaMatFunction) under a test assertion, not only inside the driver's bench run; and --config 5 itself."""
    d = _run("--no-e2e", "--no-density-sweep", "--no-cpu-baseline", "--no-alt")
    tr = d["train"]
    assert "error" not in tr, tr
    for row in tr["by_density"]:
        assert [s["stage"] for s in row["stages"]] == [1, 2, 3]
        for s in row["stages"]:
            assert s["fwd_kernel_ms"] > 0 and s["bwd_kernel_ms"] > 0 and s["autograd_step_ms"] >= s["bwd_kernel_ms"] * 0.5
            assert 0 < s["bwd_frac_hbm"] < 1.0
    d5 = _run("--config", "5")
    assert d5["n_gpus"] == 1 and d5["value"] > 0 and "config 5" in d5["config"]["workload"]
    assert d5["roofline"]["bound"] == "hbm" and 0 < d5["roofline"]["frac"] < 1.0 and d5["config"]["grad_buckets"] == 4


def test_bench_other_configs_run():
    for cfg in ("3", "4"):
        d = _run("--config", cfg, "--no-e2e", "--no-train", "--no-density-sweep", "--no-cpu-baseline")
        assert d["value"] > 0 and ("config %s" % cfg) in d["config"]["workload"]
