#!/bin/bash
# Round 6: the headline kernel with the LDS reads of the LO A-operand tiles reduced -- every one of them (an upper
# bound), or every other one -- replaced by a 4-byte read (same instruction count: the counted waits stay exact; numerics WRONG, the
# range guard off).  Per variant: ms per launch of the full and the sigma-only kernel, shader clock and board power while it runs.
#   scripts/exp_lo_reads.sh > gpurun_out/r06_lo_reads.txt     (libraries: scripts/build_variant.sh r06_no_lo_read -DMNRF_EXP_NO_LO_READ ...)
cd "$(dirname "$0")/.."
export MNRF_GUARD=0 MNRF_BENCH_LEGS=headline
for r in 1 2; do
for L in default exp_libs/r06_half_lo_read.so exp_libs/r06_no_lo_read.so; do
  if [ "$L" == default ]; then unset MNRF_LIB; else export MNRF_LIB=$L; fi
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-train 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']; t = r['telemetry']
print('$L'.ljust(34), 'full launch %.3f ms  value %.4g rays/s  frac %.4f  sclk %s MHz  power %s W (cap %s)' % (
      r['avg_launch_ms'], d['value'], r['frac'], (t.get('sclk_mhz') or {}).get('median'), (t.get('power_w') or {}).get('median'), t.get('power_cap_w')))"
done
done
