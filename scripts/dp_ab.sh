#!/bin/bash
# k_dp_group<4> with two columns per lane (packed 16-bit halves) against one column per lane: kernel times from the trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3_dp; ulimit -c 0
for v in packed scalar; do
  [ $v = scalar ] && export MCX_DP_SCALAR=1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_dp/$v -o kt -- python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0 > gpurun_out/r3_dp/$v.log 2>&1 < /dev/null
  f=$(find gpurun_out/r3_dp/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'k_dp_' in n: print('%-28s calls %4s  total %8.2f ms  avg %7.3f ms' % (n.split('(')[0][-28:], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6))
" "$f"
  rm -rf gpurun_out/r3_dp/$v
done
