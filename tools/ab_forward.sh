# A/B of forward-kernel builds / launch geometries on the headline config (run on the GPU box)
run() { echo "== $1"; shift; env "$@" python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['ms_per_step']*1e3,1), 'us', round(j['roofline']['frac'],3), j['roofline']['kernel'], j['roofline']['grid'], j['roofline']['block'])
"; }
run "K1h default" A=1
run "K1h pad 24K (2 blocks/CU: 4 waves/SIMD)" IRBFN_FWD_F16_LDSPAD=24576
run "K1h pad 40K (1 block/CU.. 2/SIMD)" IRBFN_FWD_F16_LDSPAD=40960
run "K1h pad 8K" IRBFN_FWD_F16_LDSPAD=8192
run "K1h S4 QG1 (256 thr)" IRBFN_FWD_F16_S=4 IRBFN_FWD_F16_QG=1
run "K1h S2 QG1 (128 thr)" IRBFN_FWD_F16_S=2 IRBFN_FWD_F16_QG=1
