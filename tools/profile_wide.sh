#!/bin/bash
# Counters and kernel traces of the wide-row kernels as built: wave-level (SQ) counters and HBM traffic of the
# HumanoidAMP step and the AllegroHand post-physics kernel, and a kernel trace of tools/bench_tasks.py for both.
# usage (GPU box): tools/profile_wide.sh <tag>      -> gpurun_out/wide_<tag>/
set -u
TAG=${1:-t00}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/wide_$TAG
mkdir -p "$OUT"
while read -r name match args; do
  bash "$ROOT/tools/pmc_sq.sh" "$OUT/sq_$name" "$match" $args > "$OUT/sq_$name.log" 2>&1 || { tail -5 "$OUT/sq_$name.log"; exit 1; }
  bash "$ROOT/tools/pmc_traffic.sh" "$OUT/traffic_$name" "$match" $args > "$OUT/traffic_$name.log" 2>&1 || { tail -5 "$OUT/traffic_$name.log"; exit 1; }
  echo "$name counters done"
done <<'LIST'
amp amp_step_pair_kernel amp
allegro allegro_post_kernel allegro
LIST
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/bench_tasks.py" amp allegro humanoid > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
python3 "$ROOT/tools/kstats.py" "$OUT/trace" 12 | tee "$OUT/kernel_stats_tasks.txt"
