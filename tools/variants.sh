#!/bin/bash
# time one bench_tasks entry with every library under build_variants/ (and the in-tree one)
export PYTHONPATH=.
WHAT=$1; LINE=$2
echo "in-tree $(timeout -k 10 90 python -u tools/bench_tasks.py $WHAT 2>/dev/null | grep ^$LINE)" || exit 1
for f in build_variants/libigx_*.so; do
  echo "$f $(IGX_LIB=$PWD/$f timeout -k 10 90 python -u tools/bench_tasks.py $WHAT 2>/dev/null | grep ^$LINE)" || exit 1
done
