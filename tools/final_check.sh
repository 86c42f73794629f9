#!/bin/bash
# The driver's round-end sequence on one box: GPU tests, smoke(), the default bench line.  Logs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/final_tests.log 2>&1; rc=$?
tail -2 gpurun_out/final_tests.log
[ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1; rc=$?
tail -1 gpurun_out/final_smoke.log
[ $rc -ne 0 ] && exit $rc
t0=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; rc=$?
echo "bench.py wall $(( $(date +%s) - t0 )) s, exit $rc"
python -c "
import json
d=json.loads(open('gpurun_out/final_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'hbm_side', d['roofline'].get('hbm_side_frac'), 'pmc match', d['roofline'].get('pmc_build_match'), 'cpu', d['cpu_baseline']['value'], 'rmse', d.get('rmse'))
for c in d['config'].get('other_configs', []): print(c.get('workload','')[:40], c.get('ms_per_step') or c.get('ms_per_frame'), c.get('value'), (c.get('roofline') or {}).get('pmc_build_match'))
"
exit $rc
