#!/usr/bin/env python3
"""Build profiles/rNN_c2_pmc_traffic.json from two rocprofv3 counter passes (run separately, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <out>/pmc_fetch -- python bench.py --steps 2 --warmup 3 --no-graph --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <out>/pmc_write -- python bench.py --steps 2 --warmup 3 --no-graph --no-cpu-baseline
    python profiles/make_pmc_json.py <out>/pmc_fetch <out>/pmc_write profiles/r01_c2_pmc_traffic.json

Per kernel: average FETCH_SIZE / WRITE_SIZE (KB) per launch and hbm_bytes_per_launch_corrected = (2*FETCH + WRITE)*1024
(gfx950 tallies the 128-B requests of a wide coalesced read stream at 64 B: FETCH_SIZE is doubled, same guide)."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main(fetch_dir, write_dir, out):
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fe) | set(wr)):
        if "at::native" in k or k.startswith("__amd"):
            continue
        f = sum(fe.get(k, [0.0])) / max(1, len(fe.get(k, [])))
        w = sum(wr.get(k, [0.0])) / max(1, len(wr.get(k, [])))
        res[k[:160]] = {"FETCH_SIZE_KB_avg_per_launch": round(f, 1), "WRITE_SIZE_KB_avg_per_launch": round(w, 1),
                        "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024), "launches": len(fe.get(k, wr.get(k, [])))}
    json.dump(res, open(out, "w"), indent=1)
    print(f"{len(res)} kernels -> {out}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
