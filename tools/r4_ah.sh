#!/bin/bash
out=gpurun_out/r4ah; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "step_matches_oracle or orth or golden or shard or head_inside" > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.log
python tools/pinn2d_reference_size.py 2>&1 | grep -v amdgpu | grep "100, 100\|64, 64" | cut -c1-100 | tee $out/ref_size.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp
d=$out/../r4ah_tr; rocprofv3 --kernel-trace --output-format csv -d $R/$d -- python3 $R/tools/pinn2d_timeline_run.py 2,100,100,100,1 > /dev/null 2>&1
cd $R; python3 tools/step_timeline.py $d/*/*kernel_trace.csv | tee $out/timeline_100.txt; rm -rf $d
