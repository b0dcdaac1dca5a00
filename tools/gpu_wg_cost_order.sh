T=cudadepthmapintegration_amd/csrc/libdmi_hip_tuning.so
for v in 2097152 1048576; do for sc in dense speckle; do
DMI_DEBUG_WG_TIMES=1 DMI_LIB_OVERRIDE=$T timeout -k 10 300 python tools/gpu_wg_timeline.py --workload cfg2 --scene $sc --variant $v --tag r16d_wg_cfg2_${sc}_v$v > gpurun_out/r16d_wg_cfg2_${sc}_v$v.log 2>&1 || exit 1
done; done; echo ok
