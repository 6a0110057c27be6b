tools/round_profile.sh r04l > gpurun_out/r04l_profile.log 2>&1; tail -3 gpurun_out/r04l_profile.log | cut -c1-200
python3 tools/demo_timing.py 2>&1 | tail -4 > gpurun_out/r04l_demo.txt
RRL_DEMO_ISSUE=graph python3 tools/demo_timing.py 2>&1 | grep "save_every=0" | sed 's/^/RRL_DEMO_ISSUE=graph: /' >> gpurun_out/r04l_demo.txt
tools/demo_kt.sh > gpurun_out/r04l_demo_kernels.txt 2>&1
python3 tools/config_sweep.py > gpurun_out/r04l_config_sweep.jsonl 2>/dev/null
python3 tools/order_counters.py > gpurun_out/r04l_order_counters.txt 2>&1
python3 tools/scan_tail.py > gpurun_out/r04l_scan_tail.txt 2>&1
python3 tools/step_stress.py 100000 > gpurun_out/r04l_stress.txt 2>&1; echo "exit $?" >> gpurun_out/r04l_stress.txt
for s in 0 1 2 3 4 5 6; do python3 tools/soak.py $s 300 2>&1 | tail -1; done > gpurun_out/r04l_soak.txt; echo "exit $?" >> gpurun_out/r04l_soak.txt
tail -3 gpurun_out/r04l_stress.txt; tail -3 gpurun_out/r04l_soak.txt
python3 tools/ride_timing.py > gpurun_out/r04l_ride_timing.txt 2>&1; tail -3 gpurun_out/r04l_ride_timing.txt
