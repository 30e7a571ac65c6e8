#!/bin/bash
for v in v1 v2; do
  export HSTU_BWD_W16=1 HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_w16$v.so
  bash tools/prof_pmc.sh r05_w16$v --parity-users 0 > /dev/null 2>&1
  unset HSTU_BWD_W16 HSTU_HIP_LIBRARY
  grep -c "" gpurun_out/prof_r05_w16$v/summary.md
done
