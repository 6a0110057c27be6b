#!/bin/bash
# usage (build container, repo root): tools/install_evidence.sh <tag> [<previous tag to remove>] -- copies what tools/final_run.sh
# (run on the GPU box through gpurun; results merged into gpurun_out/) produced into profiles/, checking that the
# profiled sources are THIS tree's (csrc_sha)
set -e
T=$1; OLD=$2
SHA=$(python3 - <<'PY'
import hashlib, os
h = hashlib.sha256()
d = "a-robust-registration-loss_amd/csrc"
for f in sorted(x for x in os.listdir(d) if x.endswith((".hip", ".h", ".inc"))) + ["../../include/rrl.h"]:
    h.update(open(os.path.join(d, f), "rb").read())
print(h.hexdigest()[:16])
PY
)
GOT=$(python3 -c "import json; print(json.load(open('gpurun_out/${T}_pmc_summary.json'))['csrc_sha'])")
[ "$SHA" = "$GOT" ] || { echo "csrc_sha of the tree ($SHA) != the profile's ($GOT): rerun tools/final_run.sh"; exit 1; }
for f in cpu gpu; do
  head -1 gpurun_out/${T}_pytest_$f.log | grep -q "csrc_sha $SHA" || { echo "pytest $f log is not of this tree"; exit 1; }
done
[ -n "$OLD" ] && git rm -q --ignore-unmatch profiles/${OLD}_*
R=${T%%[a-z]}  # r05b -> r05 (files that are per round, not per pass)
cp gpurun_out/${T}_bench.json profiles/${T}_bench.json
cp gpurun_out/${T}_bench_prof.json profiles/${T}_bench_under_rocprof.json
cp gpurun_out/${T}_stats/s_kernel_stats.csv profiles/${T}_bench_kernel_stats.csv
cp gpurun_out/${T}_stats_step/s_kernel_stats.csv profiles/${T}_step_kernel_stats.csv
cp gpurun_out/${T}_stats_step_b64/s_kernel_stats.csv profiles/${T}_step_b64_kernel_stats.csv
cp gpurun_out/${T}_pmc_summary.json gpurun_out/${T}_pmc_summary_b64.json gpurun_out/${T}_config_sweep.jsonl profiles/
cp gpurun_out/${T}_scan_hbm_traffic.json profiles/scan_hbm_traffic.json
grep -v amdgpu.ids gpurun_out/${T}_pytest_cpu.log > profiles/${R}_pytest_cpu.log
grep -v amdgpu.ids gpurun_out/${T}_pytest_gpu.log > profiles/${R}_pytest_gpu.log
(echo "# final tree (csrc_sha $SHA): tools/demo_timing.py on MI355X (one C call per epoch, the Chamfer walk riding in the scan's launch; RRL_DEMO_ISSUE=graph: the launches as a hipGraph replay); then tools/demo_kt.sh: rocprofv3 kernel averages of the one-call epoch"; cat gpurun_out/${T}_demo.txt gpurun_out/${T}_demo_kernels.txt) | grep -v amdgpu.ids > profiles/${T}_demo_epochs_per_s.txt
(echo "# (csrc_sha $SHA): tools/scan_tail.py and tools/order_counters.py on MI355X"; grep -v amdgpu.ids gpurun_out/${T}_scan_tail.txt; grep -v amdgpu.ids gpurun_out/${T}_order_counters.txt) > profiles/${T}_scan_tail.txt
(echo "# (csrc_sha $SHA): tools/ride_timing.py on MI355X -- the step with the trainers' Chamfer monitor, per step"; grep -v amdgpu.ids gpurun_out/${T}_ride_timing.txt) > profiles/${T}_ride_timing.txt
(echo "# (csrc_sha $SHA): tools/multi_pose_timing.py (RPM fragment at C2, k poses: ONE multi-pose evaluation vs pose after pose) and tools/fragment_timing.py on MI355X"; grep -v amdgpu.ids gpurun_out/${T}_multi_pose.txt; grep -v amdgpu.ids gpurun_out/${T}_fragments.txt) > profiles/${T}_fragments.txt
(echo "# FINAL tree (csrc_sha $SHA): tools/step_stress.py 100000 on MI355X (gpurun)"; grep -v amdgpu.ids gpurun_out/${T}_stress.txt) > profiles/${R}_stress.txt
(echo "# FINAL tree (csrc_sha $SHA): tools/soak.py <seed> 300 for seeds 0..6 on MI355X (gpurun): 2100 random (B, N, M, L, scale) shapes"; grep -v amdgpu.ids gpurun_out/${T}_soak.txt) > profiles/${R}_soak.txt
(echo "# (csrc_sha $SHA): the chained step (RRL_CHAIN=1, default) against the plain one (RRL_CHAIN=0), tools/kt.sh per-kernel rocprofv3 averages, us"; grep -v amdgpu.ids gpurun_out/${T}_chain_sweep.txt | cut -c1-400) > profiles/${T}_chain_sweep.txt
echo "installed profiles/${T}_* and profiles/${R}_pytest_{cpu,gpu}.log (csrc_sha $SHA)"
