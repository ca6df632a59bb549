#!/bin/bash
# A/B differently built libigx variants on the headline bench (one process each).
# usage: tools/ab.sh [bench args] -- lib1.so lib2.so ...
ARGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
for L in "$@"; do
  R=$(IGX_LIB=$L timeout -k 10 120 python -u bench.py --no-cpu-baseline "${ARGS[@]}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f us  frac %.3f' % (d['roofline']['kernel_ms']*1e3, d['roofline']['frac']))") || exit 1
  echo "$(basename $L): $R"
done
