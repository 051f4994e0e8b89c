#!/bin/bash
# rocprofv3 passes over one command (kernel trace + separate PMC passes), summaries under gpurun_out/<tag>/.
# usage: tools/profile.sh <tag> <python script and args...>     (run from the repository root on the GPU box)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
cd /tmp
run() { # name, rocprof args...
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 "${CMD[@]}" > $OUT/$name.log 2>&1
  echo "$name exit $?" >> $OUT/status.txt
}
CMD=("$@")
for i in "${!CMD[@]}"; do case "${CMD[$i]}" in *.py) CMD[$i]="$REPO/${CMD[$i]}";; esac; done
run trace --kernel-trace --stats || exit 1
run pmc_sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
run pmc_sq2 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM
run pmc_sq3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_BUSY_avr
cd $REPO
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
