#!/bin/bash
# tools/variants_bench.sh "<flags1>" "<flags2>" ... — like variants.sh, but measures the loop the driver times
# (bench.py, three frames in flight): prints value (Mrays/s) and ms per step for each set of build flags.
cd fyp-bidirectionalpathtracer_amd/csrc
for v in "$@"; do
  make EXTRA="$v" > /tmp/variant_build.log 2>&1 || { echo "== $v : build failed"; tail -5 /tmp/variant_build.log; continue; }
  echo -n "== [$v] "
  (cd ../.. && timeout -k 10 240 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-single-pass --no-other-configs $BENCH_ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step')")
done
make > /dev/null 2>&1   # the flag stamp (.flags) rebuilds what the last variant left behind
