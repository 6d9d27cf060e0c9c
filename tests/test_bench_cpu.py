"""bench.py's multi-rank launch path, on CPU: `python bench.py --gpus N` without a launcher must start N
ranks itself (one per GPU) and rank 0 must print ONE line with n_gpus == N; a WORLD_SIZE that disagrees with
--gpus is an error, never a silent single-rank run.  --selftest-cpu swaps
RCCL for gloo and runs the sharding / all-gather / bucketed all-reduce plumbing only (no kernels, value null)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, *flags):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                          timeout=300, cwd=ROOT, env=env)


def test_gpus_2_starts_two_ranks_itself():
    r = _run({}, "--gpus", "2", "--selftest-cpu")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ok"] is True and d["value"] is None and d["selftest"] is True


def test_world_size_mismatch_is_an_error():
    r = _run({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2", "--selftest-cpu")
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_traffic_json_cites_files_that_exist():
    """profiles/traffic.json names the rocprofv3 summary each average came from: the file must be in the repository
    (round 5 committed the CSV under another tag than the one the json recorded)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "traffic.json")) as f:
        tj = json.load(f)
    cited = set()

    def walk(o):
        if isinstance(o, dict):
            for k, v in o.items():
                if isinstance(v, str) and v.startswith("profiles/") and k.endswith("_source"):
                    cited.add(v)
                walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)
    walk(tj)
    assert cited, "traffic.json no longer says where its averages came from"
    for c in sorted(cited):
        assert os.path.exists(os.path.join(root, c)), c


def test_unit_kinds_resolve_to_their_own_family():
    """bench.E2E_FAMILIES matches kernel names by substring, first match wins: every family a unit kind is filed under
    must come out of the matcher as itself (round 5: 'conv2d_k3s3' sat before 'deconv2d_k3s3' and swallowed it)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    fams = [p for p, _ in bench.E2E_FAMILIES]
    for kind, pre in bench.UNIT_KIND_PREFIX.items():
        assert pre in fams, (kind, pre)
        assert next(p for p in fams if p in pre + "<8>") == pre, (kind, pre)
    assert next(p for p in fams if p in "void (anonymous namespace)::deconv2d_k3s3<8>(float const*") == "deconv2d_k3s3"
