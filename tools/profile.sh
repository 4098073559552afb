#!/bin/bash
# rocprofv3 passes for one command (default: the headline bench).  Run ON THE GPU BOX via gpurun:
#   gpurun -- 'bash tools/profile.sh r01 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras'
# Writes gpurun_out/prof_<tag>/{stats,pmcA..E}; counters are collected in separate passes
# (MI355X_MICROARCH.md: 8 SQ slots, FETCH_SIZE / WRITE_SIZE do not fit one pass).
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$ROOT" && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- "$@" > "$OUT/stats.log" 2>&1
pass() { n=$1; shift; rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d "$OUT/pmc$n" -- "${CMD[@]}" > "$OUT/pmc$n.log" 2>&1; }
CMD=("$@")
pass A SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass B SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS GRBM_GUI_ACTIVE
pass C SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES
pass F SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
pass G TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum
pass H SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH
pass I TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_NC_WRITE_REQ_sum TCP_GATE_EN1_sum
pass D FETCH_SIZE
pass E WRITE_SIZE
find "$OUT" -name "*.csv" | head -40
