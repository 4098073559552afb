# A/B of forward-kernel builds / launch geometries on the headline config (run on the GPU box)
run() { echo "== $1"; shift; env "$@" python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['ms_per_step']*1e3,1), 'us', round(j['roofline']['frac'],3), j['roofline']['kernel'], j['roofline']['grid'], j['roofline']['block'])
"; }
run "default" A=1
for v in "$@"; do run "$v" IRBFN_LIB=$PWD/tools/_bin/libirbfn_$v.so; done
run "default again" A=1
