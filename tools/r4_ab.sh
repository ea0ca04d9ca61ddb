#!/bin/bash
timeout -k 10 200 python tools/engine_lifecycle_cost.py 2>&1 | grep -v amdgpu | tail -3
