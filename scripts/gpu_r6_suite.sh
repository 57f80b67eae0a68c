#!/bin/bash
# round 6: the whole GPU suite with per-test durations, then the slot stress (pre-pack on / off) and the long fuzz runs
mkdir -p gpurun_out/r6e; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
( time timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=25 ) > gpurun_out/r6e/gpu_suite.log 2>&1 < /dev/null
tail -45 gpurun_out/r6e/gpu_suite.log
for pp in 1 0; do
  MCX_PREPACK=$pp timeout 900 python3 scripts/stress_slots.py --batches 2000 --seed 7 > gpurun_out/r6e/stress_prepack_$pp.txt 2>&1 < /dev/null
  tail -3 gpurun_out/r6e/stress_prepack_$pp.txt
done
MCX_PREPACK=1 timeout 900 python3 scripts/stress_slots.py --batches 2000 --seed 11 --set mc > gpurun_out/r6e/stress_prepack_1_mc.txt 2>&1 < /dev/null
tail -3 gpurun_out/r6e/stress_prepack_1_mc.txt
bash scripts/fuzz_r6.sh > gpurun_out/r6e/fuzz_r6.log 2>&1
cp gpurun_out/fuzz_r6.txt gpurun_out/r6e/
tail -30 gpurun_out/r6e/fuzz_r6.log
ls gpurun_out/fuzz_fail gpurun_out/stress_fail 2>/dev/null | head
