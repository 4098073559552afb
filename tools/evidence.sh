#!/bin/bash
# The round's evidence in one call ON THE GPU BOX: bash tools/evidence.sh r03  ->  gpurun_out/<tag>_{traffic,bench_line,bench_driverform}.json,
# gpurun_out/kstats_<tag>/*.csv (copy into profiles/ afterwards).  PMC passes first (bench.py takes roofline.traffic from their record).
TAG=${1:-r03}
python tools/measure_traffic.py $TAG > gpurun_out/${TAG}_traffic.log 2>&1 || { echo "traffic failed"; exit 1; }
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err || { echo "bench failed"; exit 1; }
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > gpurun_out/${TAG}_bench_driverform.json 2>> gpurun_out/${TAG}_bench.err || { echo "bench (driver form) failed"; exit 1; }
bash tools/profile_kernels.sh $TAG > gpurun_out/${TAG}_kstats.log 2>&1 || { echo "kernel stats failed"; tail -5 gpurun_out/${TAG}_kstats.log; exit 1; }
echo evidence done
