#!/bin/bash
# PMC counters of the narrow forward kernels K1h and K1g at config 2, one counter per pass (run ON THE GPU BOX):
#   bash tools/pmc_gram.sh  -> gpurun_out/pmc_gram.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_gram.txt
: > "$OUT"
cd /tmp && export TMPDIR=/tmp
for k in k1h k1g; do
  for c in GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM; do
    d=/tmp/pmc_$k_$c
    rm -rf "$d"
    timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv --pmc $c -d "$d" -- python3 "$ROOT/tools/run_one.py" fwd_cfg2_$k 65536 20 > /tmp/pmc.log 2>&1 || { echo "FAILED $k $c" >> "$OUT"; tail -3 /tmp/pmc.log >> "$OUT"; continue; }
    python3 - "$d" "$k" "$c" >> "$OUT" <<'PY'
import csv, glob, sys
d, k, c = sys.argv[1:4]
vals, durs = [], []
for f in glob.glob(d + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "rbf_fwd_f16" in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and "End_Timestamp" in r:
                durs.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
if vals:
    v = sorted(vals)[len(vals) // 2]
    dd = sorted(durs)[len(durs) // 2] if durs else float("nan")
    print(f"{k} {c}: median {v:.4g} over {len(vals)} launches; median duration {dd / 1e3:.1f} us")
else:
    print(f"{k} {c}: no rows")
PY
  done
done
cat "$OUT"
