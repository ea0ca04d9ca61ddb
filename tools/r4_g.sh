#!/bin/bash
# round 4: cfg5 ground state with re-drawn (stratified) collocation points, last stage on BASELINE's per-GPU grid (64 x 128 x 64 = 524 288 points)
out=gpurun_out/r4g2
mkdir -p $out
python tools/accuracy_nd.py --case cfg5_3d --n 40 --lr 2e-4 --stages 16 --epochs 2500 --final 70000 --resample 100 --sets 16 --big-grid 64,128,64 --big-epochs 4000 --out $out/accuracy_cfg5_3d_6x256_resampled_then_per_gpu_grid.json 2>&1 | tee $out/accuracy_cfg5_resampled.log | grep -E "^stage (4|8|12|16)|big grid|^mu "
