#!/bin/bash
# end-of-round evidence (round 6): the rocprofv3 summaries of the four workloads (BASELINE config 3 on both synthetic genomes, configs 5 and 2) — kernel trace + one
# PMC pass per counter set each, stamped with the kernel sources' hash —, a step's timeline, the kernels of the -vcf leg, run-to-run determinism of the -vcf planes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6_profiles; ulimit -c 0
R=gpurun_out/r6_profiles
what=${1:-all}
if [ $what = all ] || [ $what = human ]; then
bash scripts/collect_profile.sh r6_human human > gpurun_out/r6_human.log 2>&1
cp gpurun_out/r6_human/summary.json $R/summary_human.json; cp gpurun_out/r6_human/kernel_stats.csv $R/rocprofv3_kernel_stats_human.csv; cp gpurun_out/r6_human/bench_line.json $R/bench_line_human.json
TAG=r6 bash scripts/timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline_r6/timeline.txt $R/timeline_step.txt
fi
if [ $what = all ] || [ $what = configs ]; then
bash scripts/profile_configs.sh r6 > gpurun_out/r6_configs.log 2>&1
cp gpurun_out/r6_cfg5/summary.json $R/summary_cfg5.json; cp gpurun_out/r6_cfg2/summary.json $R/summary_cfg2.json
cp gpurun_out/r6_cfg5/kernel_stats.csv $R/rocprofv3_kernel_stats_cfg5.csv; cp gpurun_out/r6_cfg2/kernel_stats.csv $R/rocprofv3_kernel_stats_cfg2.csv
cp gpurun_out/r6_cfg5/bench_line.json $R/bench_line_cfg5.json
fi
if [ $what = all ] || [ $what = rest ]; then
bash scripts/collect_profile.sh r6_uniform uniform > gpurun_out/r6_uniform.log 2>&1
cp gpurun_out/r6_uniform/summary.json $R/summary_uniform.json; cp gpurun_out/r6_uniform/kernel_stats.csv $R/rocprofv3_kernel_stats_uniform.csv
# the -vcf leg: which kernels make up a batch with the bookkeeping on
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6_vcf_trace3; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 --vcf-reduce 1 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0 > $out/bench.json 2> $out/bench.err < /dev/null
f=$(find $out -name "*kernel_stats.csv" | head -1); cp "$f" $R/rocprofv3_kernel_stats_vcf_leg.csv
k=$(find $out -name "*kernel_trace.csv" | head -1); python3 scripts/slice_kernels.py "$k" > $R/vcf_leg_slices.txt 2>&1
rm -rf $out
timeout 1500 python3 scripts/determinism_vcf.py > $R/determinism_vcf.json 2> gpurun_out/r6_determinism_vcf.err; tail -c 300 $R/determinism_vcf.json
fi
ls -la $R
