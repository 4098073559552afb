#!/bin/bash
# FETCH_SIZE / WRITE_SIZE only (two passes) for one command; usage: profile_fw.sh <tag> <cmd...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/pmcD" -- "$@" > "$OUT/pmcD.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/pmcE" -- "$@" > "$OUT/pmcE.log" 2>&1
