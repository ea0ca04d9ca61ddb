#!/bin/bash
# packed fp32 VALU (v_pk_fma/mul/add_f32) vs plain: MI355X_MICROARCH.md cycle constants call packed f32 an anti-lever beside MFMAs
out=gpurun_out/r4ad; mkdir -p $out
for r in 1 2; do
  for wl in ns_2d_4x64 cfg5_3d_6x256; do
    bash tools/ab_bench.sh $wl build/variants/libgpe_fastbase.so build/variants/libgpe_nopk.so 2>&1 | sed "s/^/$wl  /"
  done
done | tee $out/nopk_ab.txt
