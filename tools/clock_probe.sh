#!/bin/bash
# Sample the shader clock / power while the headline forward kernel runs back to back.
# usage (on the GPU box): bash tools/clock_probe.sh > gpurun_out/clock_probe.txt
python bench.py --steps 80000 --warmup 100 --no-cpu-baseline --no-extras > gpurun_out/clock_probe_bench.json 2> gpurun_out/clock_probe_bench.err &
BPID=$!
for i in $(seq 1 400); do
  if ! kill -0 $BPID 2>/dev/null; then break; fi
  echo "$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power|mclk' | sed 's/.*: //' | tr '\n' ' ')"
  sleep 0.4
done
wait $BPID
echo "bench rc=$?"
tail -c 700 gpurun_out/clock_probe_bench.json
