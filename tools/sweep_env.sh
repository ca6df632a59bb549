#!/bin/bash
# usage: tools/sweep_env.sh VAR "values" bench_tasks-name
VAR=$1; VALS=$2; NAME=$3
for V in $VALS; do
  OUT=$(env $VAR=$V PYTHONPATH=. timeout -k 10 120 python -u tools/bench_tasks.py $NAME 2>/dev/null | grep -v RESULT | tr '\n' ' ')
  echo "$VAR=$V: $OUT"
done
