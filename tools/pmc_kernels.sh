#!/bin/bash
# PMC counters of the matrix-core kernels, one counter per rocprofv3 pass (run ON THE GPU BOX):  bash tools/pmc_kernels.sh -> gpurun_out/pmc_kernels.txt
# rows: <label> <tools/run_one.py case> <batch> <kernel-name substring>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_kernels.txt
: > "$OUT"
cd /tmp && export TMPDIR=/tmp
while read -r label what B pat; do
  for c in GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA; do
    d=/tmp/pmc_${label}_$c
    rm -rf "$d"
    timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv --pmc $c -d "$d" -- python3 "$ROOT/tools/run_one.py" $what $B 20 > /tmp/pmc.log 2>&1 || { echo "FAILED $label $c" >> "$OUT"; tail -3 /tmp/pmc.log >> "$OUT"; continue; }
    python3 - "$d" "$label" "$c" "$pat" >> "$OUT" <<'PY'
import csv, glob, sys
d, k, c, pat = sys.argv[1:5]
vals, durs = [], []
for f in glob.glob(d + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and pat in r["Kernel_Name"]:
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
done <<'ROWS'
k1g fwd_cfg2_k1g 65536 rbf_fwd_f16gram
k2g vjp_cfg3 65536 rbf_vjp_f16gram
k1gwide fwd_cfg4_wide 32768 gram_wide
ROWS
cat "$OUT"
