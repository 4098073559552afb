"""HBM traffic of the headline kernel for bench.py's `roofline.traffic` (run ON THE GPU BOX):
    python tools/measure_traffic.py
Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md) of
`python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras`; per-launch means of the headline kernel ->
profiles/<tag>_traffic.json (tag = argv[1], default r03), stamped with the kernel-source fingerprint and the launch geometry (bench.py ignores the
record when either has changed).  FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3.  The forward reads
dword-coalesced rows (not 16-byte-per-lane streams), for which the guide's x2 FETCH_SIZE correction is not
calibrated: the value is stored uncorrected and the note says so."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def one_pass(counter, tag):
    out = os.path.join(ROOT, "gpurun_out", f"traffic_{tag}")
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "--pmc", counter, "-d", out, "--",
                    "python3", os.path.join(ROOT, "bench.py"), "--steps", "50", "--warmup", "5", "--no-cpu-baseline", "--no-extras"],
                   check=True, env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    agg = collections.defaultdict(list)
    meta = {}
    for f in glob.glob(os.path.join(out, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "rbf_fwd" in r["Kernel_Name"]:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
                meta[r["Kernel_Name"]] = (int(r["Grid_Size"]), int(r["Workgroup_Size"]))
    return agg, meta


def main():
    fetch, meta = one_pass("FETCH_SIZE", "fetch")
    write, _ = one_pass("WRITE_SIZE", "write")
    rec = {"_comment": __doc__.split("\n\n")[0]}
    fp = bench.kernel_fingerprint()
    for k in fetch:
        # demangled template name -> the name the library reports (irbfn_net_last_launch)
        f_kib, w_kib = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k])
        grid_threads, wg = meta[k]
        rec[k] = {"fetch_kib": f_kib, "write_kib": w_kib, "bytes": int((f_kib + w_kib) * 1024), "launches": len(fetch[k]),
                  "grid": grid_threads // wg, "block": wg, "fingerprint": fp,
                  "note": "FETCH_SIZE uncorrected (dword-coalesced reads; the x2 rule holds for 16 B/lane streams only)"}
    # key by the library's own kernel name too
    import torch  # noqa: F401
    from irbfn_amd import configs, distributed
    from irbfn_amd.model import WCRBFNet
    net = WCRBFNet.from_config(configs.model_card(2))
    net.bind(distributed.params_to_device(configs.synth_params(2)))
    net(torch.from_numpy(configs.synth_queries(2)).cuda())
    lib_name = net.last_launch()["kernel"]
    for k in list(fetch):
        if k.split("<")[0].split("::")[-1] == lib_name.split("<")[0]:
            rec[lib_name] = dict(rec[k], profiler_name=k)
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    json.dump(rec, open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json"), "w"), indent=2)
    # gpurun brings back gpurun_out/ only: copy the record from there into profiles/ after the call
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", f"{tag}_traffic.json"), "w"), indent=2)
    print(json.dumps(rec, indent=2))


if __name__ == "__main__":
    main()
