"""bench.py contract: one JSON line with the driver's keys, the roofline object of the dominant kernel and
(at N = 1) the cpu_baseline object.  Short run; the numbers themselves are not asserted."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, live_traffic=False):
    # (the live PMC count of roofline.traffic is two more child processes under rocprofv3: its own test below)
    if not live_traffic:
        flags = flags + ("--no-live-traffic",)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", *flags],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run("--no-e2e", "--no-train", "--no-density-sweep", "--cpu-budget", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "pairs/s" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    # the dominant kernel's roof is the one it is closer to; both fractions are reported
    assert r["bound"] in ("mfma", "hbm") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert (r["bound"], r["unit"], r["peak"]) in (("mfma", "TFLOP/s", 157.3), ("mfma", "TFLOP/s", 2500.0),
                                                   ("hbm", "GB/s", 8000.0))
    assert r["frac"] == max(r["mfma"]["frac"], (r["hbm"] or {"frac": 0})["frac"])
    # the definition is frozen (round 5): the binding roof heads the object, both roofs and the wording travel with it,
    # and the kernel it names is the one profiles/traffic.json counted (key, time)
    assert "LARGER" in r["definition"] and "EXECUTED flops" in r["definition"]
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        tj = json.load(f)["wino_gemm"]
    assert tj["kernel_match"] in r["kernel"]
    if "rocprof_avg_ms" in tj:                          # the committed rocprofv3 --kernel-trace --stats average
        assert abs(r["ms"] - tj["rocprof_avg_ms"]) < 0.2 * tj["rocprof_avg_ms"], (r["ms"], tj["rocprof_avg_ms"])
    assert os.path.exists(os.path.join(ROOT, tj["rocprof_avg_ms_source"])), tj["rocprof_avg_ms_source"]
    # the same timed region five more times (spread of one box) and the clock / power it ran at
    vb = d["value_blocks"]
    assert vb["n_blocks"] == 5 and len(vb["blocks"]) == 5 and vb["min"] <= vb["median"] <= vb["max"]
    assert 0.5 * d["value"] < vb["median"] < 2.0 * d["value"]
    assert "smi_under_load" in d and d["smi_under_load"]["steps_run_meanwhile"] > 0
    assert d["dtype"].startswith("f32")
    assert 0.15 < r["mfma"]["frac"] < 1.0 and 0.15 < r["frac"] < 1.0 and "wino_gemm" in r["kernel"]
    assert r["traffic"] is None or r["hbm"] is None or \
        0.8 < r["traffic"] / r["hbm"]["algorithmic_bytes_per_launch"] < 1.5      # counted vs algorithmic bytes
    assert 0.3 < r["fp32_equivalent_frac_of_fp32_mfma_peak"] < 1.2
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "pairs/s" and c["sample"]
    assert c["one_thread"]["cores"] == 1 and 0 < c["one_thread"]["value"] <= c["value"] * 1.5
    cv = d["roofline_costvol"]
    assert cv["bound"] == "hbm" and cv["peak"] == 8000.0 and 0 < cv["frac"] < 1.2
    alt = d["alt_wino_gemm_fp32"]                      # round 2's fp32 MFMA Winograd GEMM, measured in a child process
    assert "error" not in alt, alt
    assert alt["value"] > 0 and alt["wino_gemm_ms"] > 0 and 0 < alt["fp32_mfma_frac"] < 1.0
    cfs = d["stage0_cost_funcs"]                       # the stage-0 branch with each --cost_func of the reference
    assert "error" not in cfs, cfs
    assert 0 < cfs["cor_ms"] and 0.7 < cfs["ssd_ms"] / cfs["cor_ms"] < 1.5 and 0.7 < cfs["cat_ms"] / cfs["cor_ms"] < 1.5


def test_costvol_density_map_and_e2e_objects():
    """roofline_costvol.frac_at_density {1.0, 0.5, 0.3, 0.1} (algorithmic bytes of the pass / time / 8 TB/s) and, in the
    e2e object, the per-kernel-family roofline table of one forward + the CPU figure for the same metric."""
    import shutil
    d = _run("--no-train", "--no-alt", "--no-valu-floor", "--cpu-budget", "2")
    cv = d["roofline_costvol"]
    assert sorted(cv["frac_at_density"]) == ["0.1", "0.3", "0.5", "1.0"]
    assert d["roofline"]["costvol_pass"]["frac_at_density"] == cv["frac_at_density"]     # inside the object the driver keeps
    assert all(0 < v < 1.2 for v in cv["frac_at_density"].values())
    assert cv["frac_at_density"]["0.1"] > cv["frac_at_density"]["0.5"] > cv["frac_at_density"]["1.0"]
    for k, v in cv["frac_at_density"].items():
        assert abs(v - cv["bytes_per_launch"] / cv["ms_at_density"][k] / 1e6 / 8000.0) < 1e-9
    e = d["e2e"]
    assert "error" not in e, e
    assert d["value_end_to_end"] >= e["value"] > 0
    cb = e["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "pairs/s" and 0 < cb["value"] < e["value"] and cb["cores"] >= 1
    assert set(e["unit_work"]) >= {"conv", "mfma"}
    if shutil.which("rocprofv3"):
        t = e["roofline"]
        assert "error" not in t, t
        assert len(t["top_kernels"]) == 5 and t["one_forward"]["kernels"] > 50
        ms = [k["ms"] for k in t["top_kernels"]]
        assert ms == sorted(ms, reverse=True) and sum(ms) <= t["one_forward"]["busy_ms"] * 1.0001
        with_frac = [k for k in t["top_kernels"] if "frac" in k]
        assert len(with_frac) >= 4 and all(0 < k["frac"] < 1.0 and k["bound"] in ("hbm", "compute") for k in with_frac)
        assert any("wino_gemm" in " ".join(k["kernel_names"]) for k in t["top_kernels"])


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


@pytest.mark.parametrize("cfg", [3, 4])
def test_bench_other_configs_run_and_match_the_oracle_at_full_size(cfg):
    """Configs 3 (KITTI 1242x378, 4 pairs per GPU) and 4 (Middlebury half-res 1512x1026, max_disp 270): the bench
    line, and -- on the bench's own full-size tensors -- sampled rows of every SpaMat/SpaVar stage against the
    oracle (rows are independent, so a row sample at full width is the full-size kernel's result) plus the
    stage-0 result of one pair against the torch-CPU oracle."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import decnet_amd
    import oracle
    from oracle import stage0 as o0
    d = _run("--config", str(cfg), "--no-e2e", "--no-train", "--no-density-sweep", "--no-cpu-baseline")
    assert d["value"] > 0 and ("config %s" % cfg) in d["config"]["workload"]
    bench.set_config(cfg)
    try:
        dev = torch.device("cuda:0")
        B = bench.DEFAULT_B
        for dens in (1.0, 0.4):
            feats, masks = bench.make_inputs(B, dev, dens, seed=77)
            for s in (1, 2, 3):
                C, H, W, D = bench.STAGES[s]
                (L, R), (rm, tm) = feats[s], masks[s]
                o, v, ss, mc = decnet_amd.spamatvar_forward(L, R, rm, tm, D)
                for b, y in ((0, 0), (B - 1, H - 1), (B // 2, H // 3)):
                    sl = (slice(b, b + 1), slice(None), slice(y, y + 1))
                    Lc, Rc = L[sl].cpu(), R[sl].cpu()
                    rmc, tmc = rm[b:b + 1, y:y + 1].cpu(), tm[b:b + 1, y:y + 1].cpu()
                    oo, so, mo = oracle.spamat_forward(Lc, Rc, rmc, tmc, D)
                    vo, _, _ = oracle.spavar_forward(Lc, Rc, rmc, tmc, oo, D)
                    # fp32 expectation over D candidates: 2e-4 px at D = 216, ~1e-6 D beyond
                    assert np.abs(o[b, y].cpu().numpy() - oo[0, 0]).max() < max(2e-4, 1.3e-6 * D), (cfg, s, dens, b, y)
                    assert np.allclose(mc[b, y].cpu().numpy(), mo[0, 0], rtol=1e-5, atol=1e-6)
                    assert np.allclose(ss[b, y].cpu().numpy(), so[0, 0], rtol=2e-5, atol=1e-6)
                    assert np.allclose(v[b, y].cpu().numpy(), vo[0, 0], rtol=2e-4, atol=2e-3)
        # stage 0 of one pair at the config's full stage-0 shape
        C0, H0, W0, D0 = bench.STAGES[0]
        params = o0.random_params(C0, 5)
        reg = decnet_amd.CostRegNetNoDown(C0, 2 * C0, "cor")
        for u, p in zip(reg.units(), params):
            u.conv.weight.data = p["w"].clone()
            u.bn.weight.data, u.bn.bias.data = p["bn"][0].clone(), p["bn"][1].clone()
            u.bn.running_mean.data, u.bn.running_var.data = p["bn"][2].clone(), p["bn"][3].clone()
        reg = reg.to(dev).eval()
        g = torch.Generator().manual_seed(5)
        left = torch.relu(torch.randn(1, C0, H0, W0, generator=g))
        right = torch.relu(torch.randn(1, C0, H0, W0, generator=g))
        with torch.no_grad():
            pred = decnet_amd.Stage0(reg)(left.to(dev), right.to(dev), D0)
            pred_o, _, _ = o0.stage0_forward(left, right, params, D0)
        assert float((pred.cpu() - pred_o).abs().max()) < 1e-3, cfg
    finally:
        bench.set_config(2)


def test_bench_collectives_run_through_rccl_in_a_world_of_one():
    """The N > 1 branch on the one GPU there is: under torch.distributed.run --nproc-per-node 1 with
    --force-collective bench.py initialises RCCL (backend nccl), barriers through it, runs the per-step
    all_gather_into_tensor(async_op=True) + work.wait() (config 2) and the 4-bucket async gradient all-reduce
    (config 5); the gathered shard must equal what was sent and the value must be that of a normal run's order."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
            "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5",
            "--warmup", "2", "--force-collective", "--no-e2e", "--no-train", "--no-density-sweep", "--no-cpu-baseline",
            "--no-alt"]
    for extra in ([], ["--config", "5"]):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][-1])
        c = d["collective"]
        print("collective:", c)
        assert c["backend"] == "nccl" and c["world"] == 1 and c["forced_at_world_1"] is True and d["value"] > 0
        if extra:
            assert c["allreduce_4_buckets_alone_ms"] > 0
        else:
            assert c["own_shard_equals_sent"] is True and c["all_gather_alone_ms"] > 0
    # plain `python bench.py --force-collective` (no launcher): a world of one by itself
    d = _run("--force-collective", "--no-e2e", "--no-train", "--no-density-sweep", "--no-cpu-baseline", "--no-alt")
    assert d["collective"]["backend"] == "nccl" and d["collective"]["own_shard_equals_sent"] is True


def test_roofline_traffic_is_counted_in_the_run():
    """Without --no-live-traffic bench.py counts the dominant kernel's HBM-side bytes itself (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE child passes) instead of quoting profiles/traffic.json; the two must agree (the committed figure is the
    same measurement from tools/profile_round.sh)."""
    import shutil
    if not shutil.which("rocprofv3"):
        pytest.skip("no rocprofv3 on this box")
    d = _run("--no-e2e", "--no-train", "--no-density-sweep", "--no-cpu-baseline", "--no-alt", "--no-valu-floor",
             live_traffic=True)
    r = d["roofline"]
    assert r["traffic_source"].startswith("counted in this run"), r["traffic_source"]
    # V in + split U^T in + M out of the GEMM: 139 + 62 + 139 MB algorithmic
    assert 0.8 * 340e6 < r["traffic"] < 1.3 * 340e6, r["traffic"]
    if r.get("traffic_committed"):
        assert abs(r["traffic"] - r["traffic_committed"]) < 0.15 * r["traffic_committed"]
