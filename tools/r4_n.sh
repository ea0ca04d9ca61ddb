#!/bin/bash
out=gpurun_out/r4n; mkdir -p $out
bash tools/r4_m.sh
unset GPE_HIP_LIB
python tools/accuracy_nd.py --case ns_2d --big-grid 1024,1024 --big-epochs 0 --out $out/acc_ns_fixed_eval1024.json 2>&1 | grep -E "big grid|^mu "
python tools/accuracy_nd.py --case ns_2d --resample 100 --big-grid 1024,1024 --big-epochs 0 --out $out/acc_ns_resample_eval1024.json 2>&1 | grep -E "big grid|^mu "
python tools/accuracy_nd.py --case cfg2_1d --big-grid 1048576 --big-epochs 0 --out $out/acc_cfg2_fixed_eval1M.json 2>&1 | grep -E "big grid|^mu "
