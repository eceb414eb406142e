#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace statistics of bench.py for every BASELINE shape, the VQ-only
# bench, and the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, no other trace domains) for C2.
# Output goes to gpurun_out/r03/ ; the summaries are then copied into profiles/ by hand (see profiles/README.md).
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/r03"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for wl in c2 c4 c5 stage2; do
  steps=5; [ "$wl" = "c4" ] && steps=3
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$wl" -- python3 "$R/bench.py" --workload $wl --steps $steps --warmup 3 --no-cpu-baseline > "$OUT/stats_$wl.log" 2>&1 || echo "stats $wl failed"
  cp $(find "$OUT/stats_$wl" -name "*kernel_stats.csv" | head -1) "$OUT/r03_${wl}_kernel_stats_rocprofv3.csv" 2>/dev/null
  echo "[collect] $wl done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_vq" -- python3 "$R/bench.py" --vq-only --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/stats_vq.log" 2>&1 || echo "stats vq failed"
cp $(find "$OUT/stats_vq" -name "*kernel_stats.csv" | head -1) "$OUT/r03_vq_only_kernel_stats_rocprofv3.csv" 2>/dev/null
echo "[collect] vq done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" --steps 2 --warmup 3 --no-graph --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1 || echo "pmc fetch failed"
echo "[collect] pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" --steps 2 --warmup 3 --no-graph --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1 || echo "pmc write failed"
echo "[collect] pmc write done"
python3 "$R/profiles/make_pmc_json.py" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/r03_c2_pmc_traffic.json"
# keep the merge small: drop the raw traces, keep logs + summaries
rm -rf "$OUT"/stats_*/ "$OUT"/pmc_fetch "$OUT"/pmc_write
ls -la "$OUT"
