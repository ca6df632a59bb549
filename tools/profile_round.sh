#!/bin/bash
# Round profile: rocprofv3 kernel-trace stats of the default bench command plus
# the PMC passes; writes compact summaries into profiles/ (tracked).
# usage (on the GPU box): tools/profile_round.sh <tag>
set -u
TAG=${1:-t00}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$ROOT/gpurun_out/profiles"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
python3 - "$OUT" "$ROOT/gpurun_out/profiles/${TAG}_kernel_stats.csv" <<'PY'
import csv, glob, sys
out, dst = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(dst, "w") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        name = r["Name"]
        if len(name) > 120:
            name = name[:117] + "..."
        w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
print(open(dst).read()[:1500])
PY
bash "$ROOT/tools/pmc_passes.sh" "$OUT/pmc" > "$OUT/pmc.log" 2>&1 || { tail -20 "$OUT/pmc.log"; exit 1; }
cp "$OUT/pmc/summary.json" "$ROOT/gpurun_out/profiles/${TAG}_pmc_ant_step.json"
cat "$OUT/pmc/summary.json"
