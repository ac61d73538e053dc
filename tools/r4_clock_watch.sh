#!/bin/bash
# ON THE GPU BOX: shader clock / power / temperature of the GPU while the headline step runs for a few seconds (rocm-smi polled from a second
# process), next to the box probes: what state is this box in when its kernels run 5-9 % slower than elsewhere?  
mkdir -p gpurun_out
python3 bench.py --steps 6000 --warmup 24 --sweeps 0 --no-cpu > gpurun_out/clockwatch_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|hotspot)|GPU use" | tr -s ' \t' ' ' | tr '\n' ';'
  echo
  sleep 0.5
done
wait $BP
python3 -c "
import json; d=json.load(open('gpurun_out/clockwatch_bench.json')); print(d['value'], d['box'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
