#!/bin/bash
out=gpurun_out/r4j; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_surface.py -q -m gpu -k "vary_beta_driver" > $out/pytest_vbeta.log 2>&1; echo "vbeta driver rc $? $(tail -1 $out/pytest_vbeta.log)"
GPE_HIP_LIB=$PWD/build/variants/libgpe_fpoff.so timeout -k 10 300 python -m pytest tests/test_gpu_surface.py -q -m gpu -k "notebook_driver_reproduces or refine_driver_against or vary_beta_driver" > $out/pytest_drivers_fpoff.log 2>&1; echo "drivers under -ffp-contract=off rc $? $(tail -1 $out/pytest_drivers_fpoff.log)"
GPE_FWD_B6=1 GPE_BWD_B6=1 GPE_COOP_FWD_MAX_TILES=0 timeout -k 10 300 python -m pytest tests/test_gpu_surface.py -q -m gpu -k "notebook_driver_reproduces or refine_driver_against or vary_beta_driver" > $out/pytest_drivers_b6.log 2>&1; echo "drivers under split-bf16 rc $? $(tail -1 $out/pytest_drivers_b6.log)"
bash tools/profile_round4.sh gpurun_out/r4j/prof 1
