#!/bin/bash
mkdir -p gpurun_out/r4av
timeout -k 10 700 python tools/fuzz_parity.py 1200 7 > gpurun_out/r4av/fuzz.txt 2>&1; echo "rc $?"; grep -v amdgpu gpurun_out/r4av/fuzz.txt | tail -30 | cut -c1-330
