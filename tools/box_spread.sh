#!/bin/bash
# One default-length bench line (driver contract: --steps 20 --warmup 5 defaults) reduced to the figures that separate the box from the kernels; appended to
# gpurun_out/box_spread.txt.  Run once per gpurun call (every call gets a fresh box):   for i in 1 2 3; do gpurun -- tools/box_spread.sh; done
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; mkdir -p gpurun_out
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys,socket
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; f=r['fused_layer']
print('host %s  ms_per_step %.3f  molecules/s %.1f  sclk %.0f MHz  power %.0f W  edge launch %.4f ms  frac %.3f  frac_at_measured_clock %.3f  tile_cycles %.0f  one handle fused / two launches %.3f / %.3f' % (
      socket.gethostname(), d['ms_per_step'], d['value'], r['sclk_mhz'] or 0, r['power_w'] or 0, r['avg_launch_ms'], r['frac'], r['frac_at_measured_clock'] or 0, r['tile_cycles'],
      f['one_handle_ms_per_step']['fused'], f['one_handle_ms_per_step']['two_launches_per_layer']))" | tee -a gpurun_out/box_spread.txt
