#!/bin/bash
out=gpurun_out/r4h; mkdir -p $out
T="tests/test_gpu_parity.py::test_step_matches_oracle[1d_64x4_m3_p4_odd-fused]"
for env in "" "GPE_COOP_FWD_MAX_TILES=0" "GPE_PIPE=0" "GPE_SPLIT_UPDATE=0" "GPE_FUSE_HEAD=0" "GPE_MERGE_BC=0" "GPE_HIP_LIB=$PWD/build/variants/libgpe_fpoff.so"; do
  env $env timeout 120 python -m pytest "$T" -q -x > $out/bisect.log 2>&1; echo "[$env] rc $? $(grep -E 'passed|failed' $out/bisect.log | tail -1) $(grep -E '^E +assert [0-9]' $out/bisect.log | head -1)"
done
