export TMPDIR=/tmp
P="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01k/kt -- $P > gpurun_out/r01k.kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/r01k/sq1 -- $P > gpurun_out/r01k.sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/r01k/sq3 -- $P > gpurun_out/r01k.sq3.log 2>&1
cat gpurun_out/r01k/kt/*/*kernel_stats.csv | cut -c1-200 | head -5
