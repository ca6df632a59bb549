#!/bin/bash
# HBM traffic per launch of one kernel of tools/bench_tasks.py, from the L2's memory-side counters:
# one rocprofv3 run per counter group.
# usage: tools/pmc_traffic.sh <outdir> <kernel name substring> <bench_tasks args...>
set -u
OUT=$(realpath -m "$1"); MATCH=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_REQ_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/tools/bench_tasks.py" --steps 4 "$@" > "$OUT/pass$i.log" 2>&1
  rc=$?
  echo "pass $i ($P) rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 "$OUT/pass$i.log"; exit 1; fi
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" "$MATCH" | tee "$OUT/summary.json"
