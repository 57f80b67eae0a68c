#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 3000 python -m pytest tests -m gpu -q -x --timeout 2400 -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/r4_pytest_bw.log; tail -4 gpurun_out/r4_pytest_bw.log
timeout 900 python scripts/ab_env.py --steps 3 "" "MCX_BUILD_BY_LANE=1" "" > gpurun_out/r4_ab8_cfg3.txt 2>/dev/null; cat gpurun_out/r4_ab8_cfg3.txt
timeout 900 python scripts/ab_env.py --steps 2 --rlen 250 --ins 0.025 --dele 0.025 --alg nw "" "MCX_BUILD_BY_LANE=1" > gpurun_out/r4_ab8_cfg5.txt 2>/dev/null; cat gpurun_out/r4_ab8_cfg5.txt
SECONDS=0; python bench.py > gpurun_out/r4_bench_try.json 2> gpurun_out/r4_bench_try.err; echo "bench: $SECONDS s"
python - <<'P'
import json
o=json.loads(open('gpurun_out/r4_bench_try.json').read().strip().splitlines()[-1])
print(o['value'], o['ms_per_step'], o['stage_ms_per_step'])
print('pcie', o['value_pcie_inclusive'].get('value'), 'f2f', json.dumps(o['value_file_to_file'])[:300])
print('cpu', json.dumps(o['cpu_baseline'])[:700])
print('vcf', json.dumps(o['vcf_reduce'])[:200])
print('other', o['other_genome'].get('value'), [ (c.get('value'), c.get('ms_per_step'), json.dumps(c.get('cpu_baseline'))[:300]) for c in o['other_configs']])
P
