#!/bin/bash
# rocprofv3 passes over the headline bench (cfg3): kernel trace + stats, then PMC passes (each on its own,
# never combined with a trace domain), then the FETCH_SIZE calibration.  Usage: tools/gpu_profile.sh <tag> [bench args]
set -u
TAG=${1:-prof}; shift || true
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/$TAG
P="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --no-scenes $*"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --no-scenes $* > $O.kt.log 2>&1; echo "kt rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -- $P > $O.sq1.log 2>&1; echo "pmc1 rc=$?"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq2 -- $P > $O.sq2.log 2>&1; echo "pmc2 rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/fetch -- $P > $O.fetch.log 2>&1; echo "pmc3 rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/write -- $P > $O.write.log 2>&1; echo "pmc4 rc=$?"
timeout 600 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d $O/ta -- $P > $O.ta.log 2>&1; echo "pmc5 rc=$?"
timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq3 -- $P > $O.sq3.log 2>&1; echo "pmc6 rc=$?"
if [ -x build/fetch_calibration ]; then
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal -- ./build/fetch_calibration > $O.cal.log 2>&1; echo "cal rc=$?"
fi
find $O -name "*.csv" | wc -l; du -sh gpurun_out
