# A/B of forward-kernel builds / launch geometries on the headline config (run on the GPU box)
run() { echo "== $1"; shift; env "$@" python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['ms_per_step']*1e3,1), 'us', round(j['roofline']['frac'],3), j['roofline']['kernel'], j['roofline']['grid'], j['roofline']['block'], 'parity', j.get('parity'))
"; }
run "K1" A=1
run "K1h" IRBFN_FWD_F16=1
run "K1h S4 QG2" IRBFN_FWD_F16=1 IRBFN_FWD_F16_S=4 IRBFN_FWD_F16_QG=2
run "K1h S8 QG1" IRBFN_FWD_F16=1 IRBFN_FWD_F16_S=8 IRBFN_FWD_F16_QG=1
run "K1h S4 QG1" IRBFN_FWD_F16=1 IRBFN_FWD_F16_S=4 IRBFN_FWD_F16_QG=1
run "K1h S2 QG4" IRBFN_FWD_F16=1 IRBFN_FWD_F16_S=2 IRBFN_FWD_F16_QG=4
run "K1h terms1" IRBFN_FWD_F16=1 IRBFN_FWD_F16_TERMS=1
run "K1 again" A=1
