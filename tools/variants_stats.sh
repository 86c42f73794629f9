#!/bin/bash
# tools/variants_stats.sh "<flags1>" "<flags2>" ... — as variants.sh, but prints rocprofv3's average kernel durations of the
# bench loop (one frame in flight: kernels of different frames do not overlap) instead of the stage times.  ON THE GPU BOX.
for v in "$@"; do
  (cd fyp-bidirectionalpathtracer_amd/csrc && make EXTRA="$v" > /tmp/variant_build.log 2>&1) || { echo "== $v : build failed"; tail -5 /tmp/variant_build.log; continue; }
  echo "== [$v]"
  tag=vs_$(echo "$v" | tr -c 'A-Za-z0-9' '_')
  bash tools/prof_stats.sh $tag --inflight 1 2>&1 | grep -v "k_\|amdgpu.ids" | head -${VARIANT_LINES:-11} | cut -c1-70,100-140
done
(cd fyp-bidirectionalpathtracer_amd/csrc && make > /dev/null 2>&1)   # the flag stamp (.flags) rebuilds what the last variant left behind
