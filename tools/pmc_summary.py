"""Summarises the rocprofv3 passes written by tools/profile.sh: per-kernel mean of every counter and
the kernel-trace stats.  Usage: python tools/pmc_summary.py gpurun_out/prof_<tag> [kernel-substring]"""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    for f in sorted(glob.glob(os.path.join(root, "stats", "*", "*kernel_stats.csv"))):
        print("== kernel stats", f)
        for r in csv.DictReader(open(f)):
            if filt in r["Name"]:
                print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} us "
                      f"min {float(r['MinNs']) / 1e3:9.1f} max {float(r['MaxNs']) / 1e3:9.1f}")
    for d in sorted(glob.glob(os.path.join(root, "pmc*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            meta = {}
            for r in csv.DictReader(open(f)):
                if filt in r["Kernel_Name"]:
                    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta[r["Kernel_Name"]] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"],
                                              r["SGPR_Count"], r["Scratch_Size"])
            for k, cs in agg.items():
                print(f"== {os.path.basename(d)} {k[:80]} grid,wg,lds,vgpr,sgpr,scratch={meta[k]}")
                for c, v in sorted(cs.items()):
                    print(f"     {c:28s} mean {sum(v) / len(v):16.1f}  n={len(v)}")


if __name__ == "__main__":
    main()
