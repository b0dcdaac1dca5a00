#!/bin/bash
# Second GPU pass: instruction-cost microbench, rocprof kernel trace of the bench, PMC passes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 ./build/microbench > gpurun_out/microbench.txt 2>&1; echo "microbench rc=$?"
tail -12 gpurun_out/microbench.txt
# kernel trace + stats of the headline bench (cfg3)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1; echo "kt rc=$?"
find gpurun_out/prof_kt -name "*stats*" | head; 
# PMC passes on cfg3 (one fusion launch per step)
P="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_sq1 -- $P > gpurun_out/pmc_sq1.log 2>&1; echo "pmc1 rc=$?"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc_sq2 -- $P > gpurun_out/pmc_sq2.log 2>&1; echo "pmc2 rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_fetch -- $P > gpurun_out/pmc_fetch.log 2>&1; echo "pmc3 rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_write -- $P > gpurun_out/pmc_write.log 2>&1; echo "pmc4 rc=$?"
timeout 600 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d gpurun_out/pmc_ta -- $P > gpurun_out/pmc_ta.log 2>&1; echo "pmc5 rc=$?"
du -sh gpurun_out; find gpurun_out -name "*.csv" | head -30
