#!/bin/bash
# round 4: cfg5 ground state, last stage on BASELINE's per-GPU grid (64 x 128 x 64 = 524 288 points)
out=gpurun_out/r4g
mkdir -p $out
python tools/accuracy_nd.py --case cfg5_3d --n 40 --epochs 2500 --final 80000 --big-grid 64,128,64 --big-epochs 3000 --out $out/accuracy_cfg5_3d_6x256_big.json 2>&1 | tee $out/accuracy_cfg5_big.log | grep -E "stage 16|big grid|^mu " 
