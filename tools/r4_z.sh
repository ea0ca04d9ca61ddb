#!/bin/bash
R=$PWD; out=$PWD/gpurun_out/r4z; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
for L in 2,100,100,100,1 2,64,64,64,1; do
  d=$out/tr_$L
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $R/tools/pinn2d_timeline_run.py $L > /dev/null 2> $out/tr_$L.err || { echo fail; tail -3 $out/tr_$L.err; exit 1; }
  cp $d/*/*kernel_trace.csv $out/pinn2d_${L}_kernel_trace.csv; rm -rf $d
  echo "== $L"; python3 $R/tools/step_timeline.py $out/pinn2d_${L}_kernel_trace.csv | tee $out/pinn2d_${L}_timeline.txt
done
