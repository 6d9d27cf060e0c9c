"""Add per-launch HBM-side traffic of named kernel groups to profiles/traffic.json.

    python tools/pmc_kernels.py <FETCH_SIZE pass dir> <WRITE_SIZE pass dir> <raw.json> <traffic.json> key=substr[+substr...] ...

Each key's bytes = sum over its kernels (matched by substring of the kernel name) of the per-launch mean of
that kernel's counter, with FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (tools/pmc_summary.py
has the rest of the corrections' story).  Used for passes that launch several kernels per operator call: the
sparse-row SpaMat forward (spamat_fwd_sparse + the marker launch of spamat_fwd_mfma) at a given mask density, the
two launches of the SpaMat backward."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import means  # noqa: E402


def main():
    fdir, wdir, raw_out, out = sys.argv[1:5]
    fetch, write = means(fdir, "FETCH_SIZE"), means(wdir, "WRITE_SIZE")
    res = json.load(open(out)) if os.path.exists(out) else {}
    raw = json.load(open(raw_out)) if os.path.exists(raw_out) else {}
    for spec in sys.argv[5:]:
        key, subs = spec.split("=")
        fb = wb = 0.0
        used = {}
        for sub in subs.split("+"):
            for k, v in fetch.items():
                if sub in k:
                    fb += 2.0 * 1024.0 * v[0]
                    used[k] = {"FETCH_SIZE_KB_mean": v[0], "launches": v[1]}
            for k, v in write.items():
                if sub in k:
                    wb += 1024.0 * v[0]
                    used.setdefault(k, {})["WRITE_SIZE_KB_mean"] = v[0]
        if not used:
            print("no kernel matches", spec)
            continue
        res[key] = {"fetch_bytes": fb, "write_bytes": wb, "total_bytes": fb + wb}
        raw[key] = used
        print("%-40s fetch %8.1f MB  write %8.1f MB" % (key, fb / 1e6, wb / 1e6))
    json.dump(res, open(out, "w"), indent=1)
    json.dump(raw, open(raw_out, "w"), indent=1)


if __name__ == "__main__":
    main()
