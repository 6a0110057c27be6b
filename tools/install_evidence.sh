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
for f in sorted(x for x in os.listdir(d) if x.endswith((".hip", ".h"))) + ["../../include/rrl.h"]:
    h.update(open(os.path.join(d, f), "rb").read())
print(h.hexdigest()[:16])
PY
)
GOT=$(python3 -c "import json; print(json.load(open('gpurun_out/${T}_pmc_summary.json'))['csrc_sha'])")
[ "$SHA" = "$GOT" ] || { echo "csrc_sha of the tree ($SHA) != the profile's ($GOT): rerun tools/final_run.sh"; exit 1; }
[ -n "$OLD" ] && git rm -q --ignore-unmatch profiles/${OLD}_*
cp gpurun_out/${T}_bench.json profiles/${T}_bench.json
cp gpurun_out/${T}_bench_prof.json profiles/${T}_bench_under_rocprof.json
cp gpurun_out/${T}_stats/s_kernel_stats.csv profiles/${T}_bench_kernel_stats.csv
cp gpurun_out/${T}_stats_step/s_kernel_stats.csv profiles/${T}_step_kernel_stats.csv
cp gpurun_out/${T}_pmc_summary.json gpurun_out/${T}_config_sweep.jsonl profiles/
cp gpurun_out/${T}_scan_hbm_traffic.json profiles/scan_hbm_traffic.json
(echo "# round 4, final tree (csrc_sha $SHA): tools/demo_timing.py on MI355X (one C call per epoch, the Chamfer walk riding in the scan's launch; RRL_DEMO_ISSUE=graph: the nine launches as a hipGraph replay); then tools/demo_kt.sh: rocprofv3 kernel averages of the one-call epoch"; cat gpurun_out/${T}_demo.txt gpurun_out/${T}_demo_kernels.txt) > profiles/${T}_demo_epochs_per_s.txt
(echo "# round 4 (csrc_sha $SHA): tools/scan_tail.py and tools/order_counters.py on MI355X"; grep -v amdgpu.ids gpurun_out/${T}_scan_tail.txt; grep -v amdgpu.ids gpurun_out/${T}_order_counters.txt) > profiles/${T}_scan_tail.txt
(echo "# round 4 (csrc_sha $SHA): tools/ride_timing.py on MI355X -- the fused step with the trainers' Chamfer monitor, per step"; grep -v amdgpu.ids gpurun_out/${T}_ride_timing.txt) > profiles/${T}_ride_timing.txt
python3 - "$T" "$SHA" <<'PY'
import sys, json, os
T, sha = sys.argv[1], sys.argv[2]
for name, src, first in (("r04_stress.txt", f"gpurun_out/{T}_stress.txt", f"# round 4, FINAL tree (csrc_sha {sha}): tools/step_stress.py 100000 on MI355X (gpurun)"),
                         ("r04_soak.txt", f"gpurun_out/{T}_soak.txt", f"# round 4, FINAL tree (csrc_sha {sha}): tools/soak.py <seed> 300 for seeds 0..6 on MI355X (gpurun): 2100 random (B, N, M, L, scale) shapes;")):
    hdr = [l for l in open("profiles/" + name).read().splitlines() if l.startswith("#")]
    body = [l for l in open(src).read().splitlines() if "amdgpu.ids" not in l]
    open("profiles/" + name, "w").write("\n".join([first] + hdr[1:] + body) + "\n")
ride = f"gpurun_out/{T}_ride_pmc_cull_scan_chamfer.json"
if os.path.exists(ride):
    d = json.load(open(ride))
    out = {"csrc_sha": sha, "kernel": "cull_scan_chamfer_kernel",
           "profiled_command": f"tools/pmc_any.sh {T}_ride cull_scan_chamfer -- tools/ride_step.py (12 monitored C2 steps: ops.RegistrationStep(chamfer=True)); rocprofv3 --pmc, separate passes per counter group, per-launch means",
           "note": "the fused launch = 1280 scan workgroups + 1024 walk workgroups of 512 lanes (18432 wavefronts); alone the scan issues 11.4 M VALU instructions per launch (the round's pmc_summary), the walk the rest; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 wide-read correction)",
           "counters": {k: v for k, v in d.items() if not k.startswith("_")}}
    out["bytes_corrected"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
    json.dump(out, open(f"profiles/{T}_pmc_ride.json", "w"), indent=1)
PY
if [ -f gpurun_out/${T}_demo_ride_soak.txt ]; then
  (echo "# round 4, final tree (csrc_sha $SHA): tools/demo_ride_soak.py 10000 on MI355X (gpurun) -- rrl_demo_epoch with all three riding launches against"
   echo "# RRL_DEMO_RIDE=0 (every kernel in a launch of its own), deterministic backward: per-epoch loss / Chamfer / validity and final pose bit-identical"
   cat gpurun_out/${T}_demo_ride_soak.txt) > profiles/r04_demo_ride_soak.txt
fi
echo "installed profiles/${T}_* (csrc_sha $SHA)"
