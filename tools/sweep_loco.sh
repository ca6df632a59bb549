#!/bin/bash
# Sweep the locomotion launch knobs (tile rows x LDS reservation) on the headline bench.
# usage: tools/sweep_loco.sh ant|humanoid "tiles" "reserves"
TASK=$1; TILES=$2; RES=$3
UP=$(echo $TASK | tr a-z A-Z)
for T in $TILES; do for R in $RES; do
  OUT=$(env IGX_${UP}_TILE=$T IGX_${UP}_LDS_RESERVE=$R timeout -k 10 120 python -u bench.py --task $TASK --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f us  frac %.3f' % (d['roofline']['kernel_ms']*1e3, d['roofline']['frac']))") || exit 1
  echo "$TASK tile=$T reserve=$R: $OUT"
done; done
