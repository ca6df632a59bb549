#!/bin/bash
# A/B of libigx builds on one box: tools/ab_libs.sh "<bench_tasks names>" lib1.so lib2.so ...  (two rounds, interleaved)
names="$1"; shift
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    IGX_LIB=$PWD/$lib PYTHONPATH=. timeout -k 10 150 python tools/bench_tasks.py $names 2>&1 | grep -v RESULT | grep "{" || exit 1
  done
done
