#!/bin/bash
# Engine clock / power / temperature of the GPU while the headline loop runs (rocm-smi sampled once a second): why boxes of the pool differ.
#   tools/clocks_under_load.sh > gpurun_out/clocks_under_load.txt
F="--no-cpu-baseline --no-fp32-timing --no-extras --no-full-sample --no-other-configs --steps 2500 --warmup 5"
echo "== idle"; /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction|hotspot)" | head -8
python bench.py $F > /tmp/clk_bench.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5 6; do
  echo "== under load, sample $i"; /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction|hotspot)" | head -8
  sleep 1
done
wait $BP
python -c "import json; d=json.loads(open('/tmp/clk_bench.json').read().strip().splitlines()[-1]); print('== bench:', round(d['ms_per_step'],4), 'ms/step', round(d['value'],1), 'molecules/s over', d['steps'], 'steps')"
