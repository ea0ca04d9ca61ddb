#!/bin/bash
# prologue staging (stage_layer0 in one pass, stage_copy16) against the slab_column_sum build: identical bits, then times.  One box.
o=gpurun_out/r4ba; mkdir -p $o
prev=$PWD/build/variants/libgpe_prev2.so
timeout -k 10 120 python tools/slab_sum_bits.py 2>/dev/null > $o/bits_new.txt || { echo "bits new FAILED"; tail -5 $o/bits_new.txt; exit 1; }
GPE_HIP_LIB=$prev timeout -k 10 120 python tools/slab_sum_bits.py 2>/dev/null > $o/bits_prev.txt || { echo "bits prev FAILED"; exit 1; }
if cmp -s $o/bits_new.txt $o/bits_prev.txt; then echo "BITS IDENTICAL ($(wc -l < $o/bits_new.txt) cases)"; else echo "BITS DIFFER"; diff $o/bits_new.txt $o/bits_prev.txt; exit 1; fi
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "not rotating_trap and not ground_state_mu" > $o/pytest_gpu_without_accuracy_runs.log 2>&1; rc=$?; tail -2 $o/pytest_gpu_without_accuracy_runs.log; [ $rc = 0 ] || exit $rc
for r in 1 2; do
for wl in cfg2_1d_4x64 ns_2d_4x64 cfg1_1d_4x32; do
  for lib in new prev; do
    if [ $lib = prev ]; then export GPE_HIP_LIB=$prev; else unset GPE_HIP_LIB; fi
    timeout -k 10 100 python3 bench.py --workload $wl --steps 30 --warmup 5 --blocks 20 --no-cpu-baseline --no-alt-mode --no-parity-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; f=d.get('roofline_forward',{})
print('%-5s %-16s %9.4f ms/step  bwd %7.4f ms  fwd %7.4f ms  %.4g pts/s' % ('$lib', '$wl', d['ms_per_step'], r['avg_launch_ms'], f.get('avg_launch_ms',0), d['value']))" | tee -a $o/ab.txt
  done
done
done
unset GPE_HIP_LIB
for n in 2048 4000; do python3 tools/small_n_step.py $n 3000 2>/dev/null | tee -a $o/small_new.txt; GPE_HIP_LIB=$prev python3 tools/small_n_step.py $n 3000 2>/dev/null | sed 's/^/[prev] /' | tee -a $o/small_new.txt; done
