#!/bin/bash
# Does the host's way of waiting for the device (interrupt or polling) change the wall-clock rate of the headline
# bench?  Alternates the two settings in fresh processes on one box.  usage (GPU box, repository root): tools/sync_latency_ab.sh
set -u
for round in 1 2 3; do
  for mode in default polling; do
    if [ $mode = polling ]; then export HSA_ENABLE_INTERRUPT=0; else unset HSA_ENABLE_INTERRUPT; fi
    R=$(timeout -k 10 120 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.5f kernel_ms %.5f value %.4e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))") || exit 1
    echo "$round $mode $R"
  done
done
