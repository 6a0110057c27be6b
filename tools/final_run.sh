tools/round_profile.sh r04k > gpurun_out/r04k_profile.log 2>&1; tail -3 gpurun_out/r04k_profile.log | cut -c1-200
python3 tools/demo_timing.py 2>&1 | tail -4 > gpurun_out/r04k_demo.txt
RRL_DEMO_ISSUE=graph python3 tools/demo_timing.py 2>&1 | grep "save_every=0" | sed 's/^/RRL_DEMO_ISSUE=graph: /' >> gpurun_out/r04k_demo.txt
tools/demo_kt.sh > gpurun_out/r04k_demo_kernels.txt 2>&1
python3 tools/config_sweep.py > gpurun_out/r04k_config_sweep.jsonl 2>/dev/null
python3 tools/order_counters.py > gpurun_out/r04k_order_counters.txt 2>&1
python3 tools/scan_tail.py > gpurun_out/r04k_scan_tail.txt 2>&1
python3 tools/step_stress.py 100000 > gpurun_out/r04k_stress.txt 2>&1; echo "exit $?" >> gpurun_out/r04k_stress.txt
for s in 0 1 2 3 4 5 6; do python3 tools/soak.py $s 300 2>&1 | tail -1; done > gpurun_out/r04k_soak.txt; echo "exit $?" >> gpurun_out/r04k_soak.txt
tail -3 gpurun_out/r04k_stress.txt; tail -3 gpurun_out/r04k_soak.txt
python3 tools/ride_timing.py > gpurun_out/r04k_ride_timing.txt 2>&1; tail -3 gpurun_out/r04k_ride_timing.txt
