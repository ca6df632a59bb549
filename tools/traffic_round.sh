#!/bin/bash
# HBM traffic of the step kernels other than Ant's (tools/pmc_traffic.sh per kernel); summaries under gpurun_out/traffic_<tag>/
# usage (GPU box): tools/traffic_round.sh <tag>
set -u
TAG=${1:-t00}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/traffic_$TAG
mkdir -p "$OUT"
while read -r name match args; do
  bash "$ROOT/tools/pmc_traffic.sh" "$OUT/$name" "$match" $args > "$OUT/$name.log" 2>&1 || { tail -5 "$OUT/$name.log"; exit 1; }
  echo "$name done"
done <<'LIST'
humanoid loco_step_kernel humanoid
allegro allegro_post_kernel allegro
shadow shadow_post_kernel shadow
amp amp_step_pair_kernel amp
him humanoid_imitate_kernel him
LIST
