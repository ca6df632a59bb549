#!/bin/bash
# Copy what tools/profile_round.sh <tag> and tools/profile_wide.sh <tag> left under gpurun_out/ into profiles/
# under a common prefix.  usage: tools/install_profiles.sh <tag of the collection> <prefix in profiles/> "<note on the build>"
set -eu
TAG=$1; ROUND=$2; NOTE=${3:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cp "$ROOT/gpurun_out/profiles/${TAG}_kernel_stats.csv" "$ROOT/profiles/${ROUND}_kernel_stats.csv"
cp "$ROOT/gpurun_out/profiles/${TAG}_pmc_ant_step.json" "$ROOT/profiles/${ROUND}_pmc_ant_step.json"
cp "$ROOT/gpurun_out/wide_${TAG}/kernel_stats_tasks.txt" "$ROOT/profiles/${ROUND}_kernel_stats_tasks.txt"
python3 - "$ROOT" "$TAG" "$ROUND" "$NOTE" <<'PY'
import json, subprocess, sys
root, tag, rnd, note = sys.argv[1:5]
wide = json.loads(subprocess.check_output([sys.executable, root + "/tools/wide_summary.py", root + "/gpurun_out/wide_" + tag]))
wide["_note"] = ("rocprofv3 --pmc, tools/profile_wide.sh %s (tools/pmc_sq.sh two passes + tools/pmc_traffic.sh four passes per "
                 "kernel, program directly after --; merged by tools/wide_summary.py), averages per dispatch of "
                 "tools/bench_tasks.py; SQ_WAVE_CYCLES and the ACTIVE / WAIT counters tick every 4 clocks; read bytes = "
                 "TCC_EA0_RDREQ_128B x 128 (every read request of these kernels is a 128-byte request; FETCH_SIZE shows half "
                 "of that on gfx950), write bytes = TCC_EA0_WRREQ_64B x 64.  %s" % (tag, note))
json.dump(wide, open("%s/profiles/%s_pmc_wide_kernels.json" % (root, rnd), "w"), indent=1)
s = json.load(open("%s/profiles/%s_pmc_ant_step.json" % (root, rnd)))
t = json.load(open(root + "/profiles/traffic_ant.json"))
t["fetch_bytes_per_launch"] = s["FETCH_SIZE"] * 1024 * 2
t["write_bytes_per_launch"] = s["WRITE_SIZE"] * 1024
t["hbm_bytes_per_launch"] = t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]
t["crosscheck_rdreq_bytes"] = (s["TCC_EA0_RDREQ_128B_sum"] * 128 + s["TCC_EA0_RDREQ_32B_sum"] * 32 +
                               (s["TCC_EA0_RDREQ_sum"] - s["TCC_EA0_RDREQ_128B_sum"] - s["TCC_EA0_RDREQ_32B_sum"]) * 64)
t["crosscheck_wrreq_bytes"] = s["TCC_EA0_WRREQ_64B_sum"] * 64 + (s["TCC_EA0_WRREQ_sum"] - s["TCC_EA0_WRREQ_64B_sum"]) * 32
t["source"] = ("profiles/%s_pmc_ant_step.json (rocprofv3 --pmc on `bench.py --steps 6 --warmup 2 --settle 0 --no-cpu-baseline --no-configs`, "
               "one pass per counter group, tools/pmc_passes.sh; %s)" % (rnd, note))
json.dump(t, open(root + "/profiles/traffic_ant.json", "w"), indent=1)
for k, v in wide.items():
    if isinstance(v, dict):
        print(k, {a: b for a, b in v["derived"].items() if "TBps" in a or "kernel_us" in a or "hbm_bytes_per_env" == a})
print("ant step HBM bytes per launch", t["hbm_bytes_per_launch"])
PY
