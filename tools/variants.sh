#!/bin/bash
# tools/variants.sh "<flags1>" "<flags2>" ... — rebuild kernels.hip (EXTRA) and bvh_build.cpp (the -D flags among them:
# HOSTEXTRA) with each set of flags ON THE GPU BOX and print
# the bench frame's stage times; restores the default build at the end.  (api.cpp sees the kernel flags too: sizes in
# kernels.h that both sides use — the stack overflow area — depend on them.)
cd fyp-bidirectionalpathtracer_amd/csrc
for v in "$@"; do
  make EXTRA="$v" > /tmp/variant_build.log 2>&1 || { echo "== $v : build failed"; tail -5 /tmp/variant_build.log; continue; }
  echo "== [$v]"
  (cd ../.. && timeout -k 10 180 python tools/stages.py 2>&1 | grep -v amdgpu.ids | tail -1)
done
make > /dev/null 2>&1   # the flag stamp (.flags) rebuilds what the last variant left behind
