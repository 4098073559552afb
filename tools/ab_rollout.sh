# A/B of roll-out kernel builds (run on the GPU box): tools/ab_rollout.sh <variant...>
run() { echo "== $1"; shift; env "$@" python tools/tune_rollout.py 50 32768,262144 2>&1 | grep -v amdgpu.ids; }
run default A=1
for v in "$@"; do run "$v" IRBFN_LIB=$PWD/tools/_bin/libirbfn_$v.so; done
run "default again" A=1
