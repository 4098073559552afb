#!/bin/bash
# A/B build of the forward kernels: tools/build_variant.sh <name> <extra hipcc flags...>
#   -> tools/_bin/libirbfn_<name>.so (all other objects from the regular build); use with IRBFN_LIB=<path>.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/irbfn_amd/csrc/_obj
OUT=$ROOT/tools/_bin
mkdir -p $OUT/obj_$NAME
for D in 3 4 7 8; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I $ROOT/include -I $ROOT/irbfn_amd/csrc -DIRBFN_INST_D=$D "$@" \
        -c $ROOT/irbfn_amd/csrc/rbf_forward_kernels.hip -o $OUT/obj_$NAME/rbf_fwd_d$D.o &
done
wait
OTHERS=$(ls $OBJ/*.o | grep -v rbf_fwd_d)
hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS $OUT/obj_$NAME/*.o -o $OUT/libirbfn_$NAME.so
echo built $OUT/libirbfn_$NAME.so
