#!/bin/bash
# Wave-level counters (instruction mix, wait fractions) of one kernel of tools/bench_tasks.py.
# usage: tools/pmc_sq.sh <outdir> <kernel name substring> <bench_tasks args...>
set -u
OUT=$(realpath -m "$1"); MATCH=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/tools/bench_tasks.py" --steps 4 "$@" > "$OUT/pass$i.log" 2>&1
  rc=$?
  echo "pass $i rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 "$OUT/pass$i.log"; exit 1; fi
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" "$MATCH" | tee "$OUT/summary.json"
