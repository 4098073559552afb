"""A/B build: tools/build_variant.py <name> <src.hip[,src.hip...]> [extra hipcc flags...]
-> tools/_bin/libirbfn_<name>.so = the regular library with the named translation units recompiled with the extra
flags (all other objects from the regular build).  Use with IRBFN_LIB=<path> (irbfn_amd/_lib.py)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irbfn_amd import build as b  # noqa: E402


def main():
    name, srcs, extra = sys.argv[1], set(sys.argv[2].split(",")), sys.argv[3:]
    b.build_lib()
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin")
    objdir = os.path.join(out, f"obj_{name}")
    os.makedirs(objdir, exist_ok=True)
    hipcc = b._hipcc()
    objs, procs = [], []
    for src, obj, flags in b.UNITS:
        if src in srcs:
            o = os.path.join(objdir, obj)
            cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={b.ARCH}", "-fPIC", "-I", b.INCLUDE, "-I", b.CSRC, *flags, *extra,
                   "-c", os.path.join(b.CSRC, src), "-o", o]
            procs.append((src, subprocess.Popen(cmd)))
            objs.append(o)
        else:
            objs.append(os.path.join(b.OBJ, obj))
    for src, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"hipcc failed for {src}")
    lib = os.path.join(out, f"libirbfn_{name}.so")
    subprocess.check_call([hipcc, "-shared", "-fPIC", f"--offload-arch={b.ARCH}", *objs, "-o", lib])
    print("built", lib)


if __name__ == "__main__":
    main()
