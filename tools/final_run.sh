#!/bin/bash
# usage (GPU box, repo root; through gpurun): tools/final_run.sh <tag>
# The round's evidence of ONE tree, in one call: profiles (tools/round_profile.sh), the demo / sweep / tail / stress / soak /
# fragment runs, and -- LAST, on this very tree -- both test suites with their logs kept (round 4 shipped twelve evidence
# passes and not one pytest log).  tools/install_evidence.sh <tag> then copies everything into profiles/.
T=${1:-r06}
O=gpurun_out; mkdir -p $O
tools/round_profile.sh $T > $O/${T}_profile.log 2>&1; tail -3 $O/${T}_profile.log | cut -c1-200
python3 tools/demo_timing.py 2>&1 | tail -4 > $O/${T}_demo.txt
RRL_DEMO_ISSUE=graph python3 tools/demo_timing.py 2>&1 | grep "save_every=0" | sed 's/^/RRL_DEMO_ISSUE=graph: /' >> $O/${T}_demo.txt
tools/demo_kt.sh > $O/${T}_demo_kernels.txt 2>&1
python3 tools/config_sweep.py > $O/${T}_config_sweep.jsonl 2>/dev/null
python3 tools/order_counters.py > $O/${T}_order_counters.txt 2>&1
python3 tools/scan_tail.py > $O/${T}_scan_tail.txt 2>&1
python3 tools/multi_pose_timing.py > $O/${T}_multi_pose.txt 2>&1
python3 tools/fragment_timing.py > $O/${T}_fragments.txt 2>&1
python3 tools/step_stress.py 100000 > $O/${T}_stress.txt 2>&1; echo "exit $?" >> $O/${T}_stress.txt
rc=0; for s in 0 1 2 3 4 5 6; do python3 tools/soak.py $s 300 > $O/${T}_soak_$s.log 2>&1 || rc=1; tail -1 $O/${T}_soak_$s.log; done > $O/${T}_soak.txt; echo "exit $rc (1 = some soak.py run returned non-zero)" >> $O/${T}_soak.txt
tail -3 $O/${T}_stress.txt; tail -3 $O/${T}_soak.txt
python3 tools/ride_timing.py > $O/${T}_ride_timing.txt 2>&1; tail -3 $O/${T}_ride_timing.txt
# round 6: the chained step against the plain one, kernel by kernel (C2 and the shapes of the fuse rule)
(for S in 8,4096,4096,10000 4,4096,4096,10000 16,4096,4096,10000 32,4096,4096,10000 1,1024,1024,20000 8,2048,1024,10000; do for c in 1 0; do echo "$S RRL_CHAIN=$c"; RRL_CHAIN=$c RRL_STEP=loss tools/kt.sh ${T}_chain $S 300 2>&1 | grep -v amdgpu; done; done) > $O/${T}_chain_sweep.txt 2>&1; tail -4 $O/${T}_chain_sweep.txt
# ---- the suites, last, on this tree (csrc_sha in the first line of each log)
SHA=$(python3 -c "import bench; print(bench.csrc_sha())" 2>/dev/null)
(echo "# csrc_sha $SHA  $(date -u +%FT%TZ)  python3 -m pytest tests -q -m 'not gpu'"; python3 -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1) > $O/${T}_pytest_cpu.log
(echo "# csrc_sha $SHA  $(date -u +%FT%TZ)  python3 -m pytest tests -x -q -m gpu"; python3 -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1) > $O/${T}_pytest_gpu.log
tail -2 $O/${T}_pytest_cpu.log; tail -2 $O/${T}_pytest_gpu.log
