#!/bin/bash
out=$PWD/gpurun_out/r4aj; mkdir -p $out; R=$PWD
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 $R/tools/residual_timeline_run.py > /dev/null 2> $out/tr.err || { tail -3 $out/tr.err; exit 1; }
cd $R; python3 tools/step_timeline.py $out/tr/*/*kernel_trace.csv | cut -c1-130 | tee $out/residual_timeline.txt; rm -rf $out/tr
