for mode in "0 0" "1 1"; do
  set -- $mode
  ( for i in 1 2 3 4 5 6 7 8 9 10; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed 's/.*(\([0-9]*Mhz\)).*/\1/; s/.*Power (W): /W /' | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/r2v/clk_$1$2.txt &
  GPE_FWD_B6=$1 GPE_BWD_B6=$2 python3 bench.py --workload ns_2d_4x64 --steps 1200 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; f=d.get('roofline_forward',{})
print('fwd_b6=$1 bwd_b6=$2 %8.3f ms/step  bwd %7.3f ms  fwd %7.3f ms  %.4g points/s' % (d['ms_per_step'], r['avg_launch_ms'], f.get('avg_launch_ms',0), d['value']))"
  wait
  sed -n 3,6p gpurun_out/r2v/clk_$1$2.txt
done
