#!/bin/bash
# One rocprofv3 --kernel-trace --stats summary per (kernel, batch size) -> gpurun_out/kstats_<tag>/<what>_<B>.csv
# (averages must be recomputable per size).  Run ON THE GPU BOX:  bash tools/profile_kernels.sh r03
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kstats_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {   # what B
  d=$OUT/tmp_$1_$2
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o p -- python3 "$ROOT/tools/run_one.py" "$1" "$2" 30 > "$OUT/$1_$2.log" 2>&1 || { echo "FAILED $1 $2"; return 1; }
  f=$(find "$d" -name "*kernel_stats.csv" | head -1)
  grep -v "at::native\|__amd_rocclr" "$f" > "$OUT/$1_$2.csv"
  rm -rf "$d"
  echo "ok $1 $2"
}
for w in fwd_cfg2; do run $w 65536 || exit 1; done
for w in vjp_cfg3 train_cfg3; do run $w 65536 || exit 1; done
run train_1region 80000 || exit 1
for w in fwd_cfg4_wide tick_cfg4; do run $w 32768 || exit 1; run $w 262144 || exit 1; done
for m in st_ks st_select fullint frenet; do run roll_$m 262144 || exit 1; run roll_$m 32768 || exit 1; done
for m in st_ks fullint frenet; do run rollvjp_$m 262144 || exit 1; run rollvjp_$m 32768 || exit 1; done
run spiral 262144 || exit 1; run spiralvjp 262144 || exit 1
run sparse_fwd 65536 || exit 1; run sparse_tick 65536 || exit 1; run sparse_vjp 80000 || exit 1; run sparse_train 80000 || exit 1
echo all done
