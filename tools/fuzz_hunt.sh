#!/bin/bash
# Several tools/fuzz_parity.py processes side by side on one GPU (sessions created and torn down at random beside each other):
# usage: tools/fuzz_hunt.sh <seconds per process> <first seed> [processes=4]. Logs: gpurun_out/hunt_<seed>.log, exit codes:
# gpurun_out/hunt_rc.log (0 = every scan equal to the oracle's and no watchdog report; 3 = the watchdog saw no progress for 60 s).
T=${1:-240}; S0=${2:-801}; N=${3:-4}
mkdir -p gpurun_out; : > gpurun_out/hunt_rc.log
for i in $(seq 0 $((N - 1))); do
  s=$((S0 + i))
  ( timeout $((T + 120)) python tools/fuzz_parity.py $T $s > gpurun_out/hunt_$s.log 2>&1; echo "seed $s rc=$?" >> gpurun_out/hunt_rc.log ) &
done
wait
cat gpurun_out/hunt_rc.log
tail -q -n 1 gpurun_out/hunt_*.log
