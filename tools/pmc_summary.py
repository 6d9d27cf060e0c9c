"""Aggregate rocprofv3 --pmc counter_collection CSVs into per-kernel per-launch means and write
profiles/traffic.json (the file bench.py reads for the `traffic` fields).

    python tools/pmc_summary.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <raw.json> <traffic.json> [kernel_stats.csv]

kernel_stats.csv: the `rocprofv3 --kernel-trace --stats` summary of the same command; its average duration per kernel is
recorded beside the bytes (`rocprof_avg_ms`) -- tests/test_bench_gpu.py holds bench.py's live-measured time against it.

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch (summed over the XCDs' L2s).  FETCH_SIZE is
doubled as MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE is taken as is."""
import csv
import glob
import json
import sys
from collections import defaultdict

KEYS = {                                    # substring of the kernel name -> key in traffic.json
    # exact template names: "wino_gemm" alone would also match wino_gemm_persist / wino_gemm_bf16x3.  The step's
    # GEMM is the first of these that ran (the default is the persistent kernel at Ci = 216)
    "wino_gemm_persist<": "wino_gemm",
    "wino_gemm_bf16x3": "wino_gemm",
    "wino_gemm<": "wino_gemm",
    "wino_mid_transform": "wino_mid_transform",
    "wino_head_transform": "wino_head_transform",
    "wino_input_transform": "wino_input_transform",
    "wino_output_transform": "wino_output_transform",
    "conv3d_k3_igemm": "conv3d_k3_igemm",
    "costvol_cor_ndhwc": "costvol_cor_ndhwc",
    "cout1_tap_gemm": "cout1_tap_gemm",
    "spamat_fwd_mfma<15": "spamat_fused_stage3",
    "spamat_fwd_mfma<6": "spamat_fused_stage2",
    "spamat_fwd_mfma<3": "spamat_fused_stage1",
}


def means(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for k, v in per_dispatch.items():
            acc[names[k]].append(v)
    # per-launch mean over the launches that did the full work: a kernel that exits at once on some
    # launches (the marker launches of spamat_fwd_mfma behind the sparse-row kernel) would otherwise
    # average its traffic with zeros
    out = {}
    for k, v in acc.items():
        top = max(v)
        full = [x for x in v if x >= 0.5 * top] if top > 0 else v
        out[k] = (sum(full) / len(full), len(full))
    return out


def main():
    fdir, wdir, raw_out, out = sys.argv[1:5]
    fetch, write = means(fdir, "FETCH_SIZE"), means(wdir, "WRITE_SIZE")
    raw = {"FETCH_SIZE_KB_mean": {k: v[0] for k, v in fetch.items()},
           "WRITE_SIZE_KB_mean": {k: v[0] for k, v in write.items()},
           "launches": {k: v[1] for k, v in fetch.items()}}
    json.dump(raw, open(raw_out, "w"), indent=1)
    res = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 "
                   "--warmup 1 --no-cpu-baseline --no-e2e --no-density-sweep; per-launch means (tools/pmc_summary.py). FETCH_SIZE is doubled "
                   "as MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE is taken as is. FETCH/WRITE are "
                   "L2<->fabric requests: Infinity-Cache hits are included, so for kernels whose working set "
                   "stays in the 256 MiB Infinity Cache this is an upper bound on HBM bytes."}
    for sub, key in KEYS.items():
        if key in res:
            continue
        # launch-weighted mean over every kernel name that matches (template instantiations of one kernel)
        f = [v for k, v in fetch.items() if sub in k]
        w = [v for k, v in write.items() if sub in k]
        if not f or not w:
            continue
        fm = sum(a * n for a, n in f) / sum(n for _, n in f)
        wm = sum(a * n for a, n in w) / sum(n for _, n in w)
        fb, wb = 2.0 * 1024.0 * fm, 1024.0 * wm
        res[key] = {"fetch_bytes": fb, "write_bytes": wb, "total_bytes": fb + wb, "kernel_match": sub}
    if len(sys.argv) > 5:
        rows = list(csv.DictReader(open(sys.argv[5])))
        for key, v in res.items():
            if key == "_how":
                continue
            m = [r for r in rows if v["kernel_match"] in r["Name"]]
            if m:
                calls = sum(int(r["Calls"]) for r in m)
                v["rocprof_avg_ms"] = sum(float(r["TotalDurationNs"]) for r in m) / calls / 1e6
                v["rocprof_calls"] = calls
                v["rocprof_avg_ms_source"] = "profiles/" + sys.argv[5].split("/")[-1]
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if k != "_how":
            print("%-24s fetch %8.1f MB  write %8.1f MB" % (k, v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
