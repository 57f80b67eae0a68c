#!/bin/bash
# rocprofv3 evidence for BASELINE.json's configs 5 and 2 (the same child commands bench.py's other_configs() runs):
#   scripts/profile_configs.sh <tag>      -> gpurun_out/<tag>_cfg5/, gpurun_out/<tag>_cfg2/
tag=${1:-r4}
cd "$GRAFT_REPO_ROOT"
READS_PER_LAUNCH=8000000 bash scripts/collect_profile.sh ${tag}_cfg5 human --rlen 250 --ins 0.025 --dele 0.025 --alg nw > gpurun_out/${tag}_cfg5.log 2>&1
READS_PER_LAUNCH=1000000 bash scripts/collect_profile.sh ${tag}_cfg2 uniform --genome-mbp 4.6 --contigs 1 --repeats 20 --batch-pairs 1000000 --single-end 1 --rlen 100 > gpurun_out/${tag}_cfg2.log 2>&1
tail -3 gpurun_out/${tag}_cfg5.log gpurun_out/${tag}_cfg2.log
cat gpurun_out/${tag}_cfg5/job_classes.txt
