#!/usr/bin/env python3
"""Per-kernel matrix-pipe busy fraction from one rocprofv3 counter pass (its own run, no other trace domain):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <out>/pmc_mfma -- python3 bench.py ...
    python profiles/make_mfma_json.py <out>/pmc_mfma profiles/rNN_<what>_pmc_mfma_busy.json

mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs): the share of (SIMD x cycle) slots
of the launch in which the matrix pipe was executing (MI355X_MICROARCH.md, counters)."""
import collections
import csv
import glob
import json
import sys


def main(d, out):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k in sorted(acc):
        if "at::native" in k or k.startswith("__amd"):
            continue
        busy, act = acc[k].get("SQ_VALU_MFMA_BUSY_CYCLES", []), acc[k].get("GRBM_GUI_ACTIVE", [])
        if not act:
            continue
        b, a = sum(busy) / max(1, len(busy)), sum(act) / len(act)
        res[k[:160]] = {"GRBM_GUI_ACTIVE_avg": a, "SQ_VALU_MFMA_BUSY_CYCLES_avg": b, "launches": len(act),
                        "mfma_busy_fraction": round(b / (a / 8 * 256 * 4), 4) if a > 0 else 0.0}
    json.dump(res, open(out, "w"), indent=1)
    print(f"{len(res)} kernels -> {out}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
