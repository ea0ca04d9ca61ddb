#!/bin/bash
FUZZ_ONLY=320 timeout -k 10 300 python tools/fuzz_parity.py 400 11 7 6 2>&1 | grep -v amdgpu | tail -12 | cut -c1-300
