export TMPDIR=/tmp
for sc in sparse dense; do
O=gpurun_out/r16h_cfg3_$sc
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 bench.py --workload cfg3 --scene $sc --steps 4 --warmup 2 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --no-scenes > $O.log 2>&1; echo "$sc rc=$?"
python3 tools/step_timeline.py $O
done
