#!/bin/bash
# round 4: accuracy records on this round's build (second half of the BASELINE metric) -> gpurun_out/r4f
out=gpurun_out/r4f
mkdir -p $out
for c in ns_2d cfg2_1d cfg3_2d; do
  python tools/accuracy_nd.py --case $c --out $out/accuracy_$c.json > $out/accuracy_$c.log 2>&1; echo "$c: $(tail -1 $out/accuracy_$c.log)"
done
python tools/accuracy_cfg4.py --no-basin --out $out/accuracy_cfg4_2d_6x128_rot.json > $out/accuracy_cfg4.log 2>&1; tail -3 $out/accuracy_cfg4.log
python tools/accuracy_refine.py --tol 1e-7 > $out/accuracy_refine_tol1e-7.log 2>&1; tail -2 $out/accuracy_refine_tol1e-7.log
cp gpurun_out/accuracy_refine*.json $out/ 2>/dev/null
