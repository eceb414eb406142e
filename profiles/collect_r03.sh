#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace statistics of bench.py for every BASELINE shape, the real-data shape
# (stage2_vq.yaml at B = 128, L = 350) and the VQ-only bench; then the counter passes, each its own run with no other trace
# domain: FETCH_SIZE / WRITE_SIZE for C2 and for the VQ-only bench, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE for both.
# Output goes to gpurun_out/r03/ ; the summaries are then copied into profiles/ by hand (see profiles/README.md).
# Usage: collect_r03.sh [stats|pmc|all|c2]
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/r03"
WHAT="${1:-all}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
stats() {   # name, bench args...
  local name="$1"; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$name" -- python3 "$R/bench.py" "$@" --no-cpu-baseline > "$OUT/stats_$name.log" 2>&1 || echo "stats $name failed"
  cp $(find "$OUT/stats_$name" -name "*kernel_stats.csv" | head -1) "$OUT/r03_${name}_kernel_stats_rocprofv3.csv" 2>/dev/null
  grep '^{' "$OUT/stats_$name.log" | tail -2 > "$OUT/r03_${name}_bench_line.json"
  rm -rf "$OUT/stats_$name"
  echo "[collect] stats $name done"
}
pmc() {     # name, counters, bench args...
  local name="$1" ctr="$2"; shift 2
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- python3 "$R/bench.py" "$@" --no-graph --no-cpu-baseline > "$OUT/pmc_$name.log" 2>&1 || echo "pmc $name failed"
  echo "[collect] pmc $name done"
}
if [ "$WHAT" = "c2" ]; then
  stats c2 --workload c2 --steps 5 --warmup 3
  pmc c2_fetch FETCH_SIZE --steps 2 --warmup 3
  pmc c2_write WRITE_SIZE --steps 2 --warmup 3
  python3 "$R/profiles/make_pmc_json.py" "$OUT/pmc_c2_fetch" "$OUT/pmc_c2_write" "$OUT/r03_c2_pmc_traffic.json"
  pmc c2_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --steps 2 --warmup 3
  python3 "$R/profiles/make_mfma_json.py" "$OUT/pmc_c2_mfma" "$OUT/r03_c2_pmc_mfma_busy.json"
  rm -rf "$OUT"/pmc_*/
fi
if [ "$WHAT" = "stats" ] || [ "$WHAT" = "all" ]; then
  stats c2 --workload c2 --steps 5 --warmup 3
  stats c5 --workload c5 --steps 5 --warmup 3
  stats stage2 --workload stage2 --steps 5 --warmup 3
  stats stage2_b128_l350 --workload stage2 --batch 128 --seq 350 --steps 5 --warmup 3
  stats vq_only --vq-only --steps 5 --warmup 2
  stats c4 --workload c4 --steps 3 --warmup 3
fi
if [ "$WHAT" = "pmc" ] || [ "$WHAT" = "all" ]; then
  pmc c2_fetch FETCH_SIZE --steps 2 --warmup 3
  pmc c2_write WRITE_SIZE --steps 2 --warmup 3
  python3 "$R/profiles/make_pmc_json.py" "$OUT/pmc_c2_fetch" "$OUT/pmc_c2_write" "$OUT/r03_c2_pmc_traffic.json"
  pmc c2_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --steps 2 --warmup 3
  python3 "$R/profiles/make_mfma_json.py" "$OUT/pmc_c2_mfma" "$OUT/r03_c2_pmc_mfma_busy.json"
  pmc vq_fetch FETCH_SIZE --vq-only --steps 2 --warmup 2
  pmc vq_write WRITE_SIZE --vq-only --steps 2 --warmup 2
  python3 "$R/profiles/make_pmc_json.py" "$OUT/pmc_vq_fetch" "$OUT/pmc_vq_write" "$OUT/r03_vq_only_pmc_traffic.json"
  pmc vq_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --vq-only --steps 2 --warmup 2
  python3 "$R/profiles/make_mfma_json.py" "$OUT/pmc_vq_mfma" "$OUT/r03_vq_only_pmc_mfma_busy.json"
  pmc s2_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --workload stage2 --batch 128 --seq 350 --steps 2 --warmup 3
  python3 "$R/profiles/make_mfma_json.py" "$OUT/pmc_s2_mfma" "$OUT/r03_stage2_b128_l350_pmc_mfma_busy.json"
  rm -rf "$OUT"/pmc_*/
fi
ls -la "$OUT"
