#!/bin/bash
# usage (GPU box, repo root): tools/round_profile.sh <tag>
# Produces under gpurun_out/: the default bench line, the rocprofv3 --kernel-trace --stats summary of
# the SAME command, separate PMC passes for FETCH_SIZE / WRITE_SIZE (HBM traffic per kernel, the
# guide's recipe: own runs, --kernel-trace only) and two SQ counter passes -- for the TIMED shape (B = 8) and for the
# chip-filling shape bench.py reports under roofline.at_B64 (B = 64 on one GPU).
# Round 5: every profiling pass runs the step the driver TIMES -- SURVEY 8(d)'s step by direct issue
# (bench.py --issue direct: ops.LossStep -> rrl_loss_step_ex, prepared build, backward to points1.grad) -- so the PMC summary
# has a row for each of its kernels (records, scan, per-line stage, tail) and the step-total traffic is the timed step's.
TAG=${1:-r}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o s -- python3 $R/bench.py --no-cpu-baseline > $O/${TAG}_bench_prof.json 2> $O/${TAG}_bench_prof.err
STEPARGS="--no-cpu-baseline --no-extras --no-other --issue direct --no-dist --no-fresh --no-parity"
passes() {  # $1 = suffix ("" or "_b64"), $2 = extra bench arguments
  # the timed step alone (no variant / strict / counter / drop-in / Chamfer passes): per-kernel averages of the headline path
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_step$1 -o s -- python3 $R/bench.py $STEPARGS $2 > /dev/null 2> $O/${TAG}_bench_step$1.err
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${TAG}_pmc_$c$1 -o p -- python3 $R/bench.py --steps 6 --warmup 2 $STEPARGS $2 > $O/${TAG}_pmc_$c$1.log 2>&1
  done
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/${TAG}_pmc_sqa$1 -o p -- python3 $R/bench.py --steps 6 --warmup 2 $STEPARGS $2 > $O/${TAG}_pmc_sqa$1.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/${TAG}_pmc_sqb$1 -o p -- python3 $R/bench.py --steps 6 --warmup 2 $STEPARGS $2 > $O/${TAG}_pmc_sqb$1.log 2>&1
}
passes "" ""
passes "_b64" "--batch 64 --steps 100"
cd $R
python3 - "$TAG" <<'PY'
import csv, collections, json, sys, glob, hashlib, os, re
tag = sys.argv[1]
def csrc_sha():  # == bench.py csrc_sha(): the profile is only attached to a bench line of the same build
    h = hashlib.sha256()
    d = "a-robust-registration-loss_amd/csrc"
    for f in sorted(x for x in os.listdir(d) if x.endswith((".hip", ".h", ".inc"))) + ["../../include/rrl.h"]:
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]
O = "gpurun_out"
def per_kernel(counter_dir):
    f = glob.glob(f"{O}/{counter_dir}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for path in f:
        for r in csv.DictReader(open(path)):
            name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0].split("::")[-1]
            agg[name][r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    return {k: {c: sum(d.values()) / len(d) for c, d in v.items()} for k, v in agg.items()}
def short(n):
    return re.sub(r"^void ", "", n).split("(")[0].split("<")[0].split("::")[-1]
def summarise(sfx, B, what):
    STEP = f"bench.py --issue direct --no-extras --no-other --no-dist --no-fresh --no-parity --steps 6{what} (the timed one-call step: ops.LossStep -> rrl_loss_step_ex, SURVEY 8(d): backward to points1.grad)"
    out = {"csrc_sha": csrc_sha(), "profiled_command": STEP,
           "note": "rocprofv3 --pmc, separate passes per counter group; per-launch means. "
                   "FETCH_SIZE / WRITE_SIZE in KB as reported; gfx950 correction for wide (16 B/lane) streaming reads: "
                   "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section)"}
    hbm = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, v in per_kernel(f"{tag}_pmc_{c}{sfx}").items():
            hbm.setdefault(k, {}).update(v)
    for k, v in hbm.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["bytes_corrected"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    out["hbm_per_kernel"] = hbm
    sq = {}
    for d in ("sqa", "sqb"):
        for k, v in per_kernel(f"{tag}_pmc_{d}{sfx}").items():
            sq.setdefault(k, {}).update(v)
    step_rows = list(csv.DictReader(open(glob.glob(f"{O}/{tag}_stats_step{sfx}/**/*kernel_stats.csv", recursive=True)[0])))
    # the kernels of the timed step: what its kernel-trace pass saw at least `steps` times, library kernels only
    step_kernels = [short(r["Name"]) for r in step_rows if int(r["Calls"]) >= 100 and not any(x in r["Name"] for x in ("at::", "rocclr", "kd_", "sample_"))]
    out["step_kernels"] = step_kernels
    out["step_kernel_avg_us"] = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in step_rows if short(r["Name"]) in step_kernels}
    out["sq_per_kernel"] = {k: v for k, v in sq.items() if k in step_kernels or "cull" in k or "tri_" in k or "loss_" in k or "line_pair" in k}
    missing = [k for k in step_kernels if k not in out["hbm_per_kernel"] or k not in out["sq_per_kernel"]]
    out["step_kernels_without_counters"] = missing
    out["step_hbm_bytes_corrected"] = sum(out["hbm_per_kernel"].get(k, {}).get("bytes_corrected", 0.0) for k in step_kernels)
    json.dump(out, open(f"{O}/{tag}_pmc_summary{sfx}.json", "w"), indent=1)
    # the step's scan launch: cull_scan_kernel, or -- a chained step (round 6) -- cull_scan_build_kernel (records + both scans)
    scan_name = next((k for k in step_kernels if k.startswith("cull_scan")), "cull_scan_kernel")
    c_ = out["hbm_per_kernel"].get(scan_name, {})
    sq_ = out["sq_per_kernel"].get(scan_name, {})
    cull_avg = [float(r["AverageNs"]) / 1e3 for r in step_rows if short(r["Name"]) == scan_name]
    ent = None
    if "FETCH_SIZE" in c_ and "WRITE_SIZE" in c_:  # the object bench.py reports under roofline (traffic, issue_frac, pmc)
        ent = {"kernel": scan_name, "fetch_kb_raw": c_["FETCH_SIZE"], "write_kb_raw": c_["WRITE_SIZE"],
               "bytes": int(c_["bytes_corrected"]),
               "correction": "FETCH_SIZE x2 (gfx950 wide-read undercount), WRITE_SIZE as reported",
               "source": f"profiles/{tag}_pmc_summary{sfx}.json (rocprofv3 --pmc, separate passes per counter group, of " + STEP +
                         f"; per-launch means) and profiles/{tag}_step{sfx}_kernel_stats.csv",
               "step_kernels": step_kernels, "step_hbm_bytes": int(out["step_hbm_bytes_corrected"]),
               "step_kernel_avg_us": out["step_kernel_avg_us"]}
        for k in ("SQ_INSTS_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS"):
            if k in sq_:
                ent[k.lower()] = sq_[k]
        if cull_avg:
            ent["rocprof_avg_us"] = cull_avg[0]
    print(f"[{sfx or 'B=8'}] step kernels:", step_kernels, "without counters:", missing, "step HBM bytes (corrected): %.2f MB" % (out["step_hbm_bytes_corrected"] / 1e6))
    for r in step_rows[:8]:
        print(f"  {r['Name'][:44]:44s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us")
    if sq_:
        print("  cull SQ: VALU busy %.3f  LDS busy %.3f  wait_any %.3f  wait_inst %.3f of wave cycles; VALU insts %.3g" % (
            4 * sq_.get("SQ_ACTIVE_INST_VALU", 0) / max(sq_.get("SQ_BUSY_CYCLES", 1), 1), 4 * sq_.get("SQ_ACTIVE_INST_LDS", 0) / max(sq_.get("SQ_BUSY_CYCLES", 1), 1),
            sq_.get("SQ_WAIT_ANY", 0) / max(sq_.get("SQ_WAVE_CYCLES", 1), 1), sq_.get("SQ_WAIT_INST_ANY", 0) / max(sq_.get("SQ_WAVE_CYCLES", 1), 1), sq_.get("SQ_INSTS_VALU", 0)))
    return ent
rec = {"csrc_sha": csrc_sha()}
for sfx, B, what in (("", 8, ""), ("_b64", 64, " --batch 64")):
    try:
        ent = summarise(sfx, B, what)
    except Exception as exc:
        print(f"[{sfx}] summary failed: {type(exc).__name__}: {exc}")
        ent = None
    if ent:
        rec[f"B{B}_N4096_L10000_cull"] = ent
json.dump(rec, open(f"{O}/{tag}_scan_hbm_traffic.json", "w"), indent=1)
PY
# the bench line once more, now that the traffic / PMC object of exactly this build exists: the committed line carries it
if [ -f $O/${TAG}_scan_hbm_traffic.json ]; then
  cp $O/${TAG}_scan_hbm_traffic.json $R/profiles/scan_hbm_traffic.json
  (cd /tmp; python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err)
fi
tail -c 1500 $O/${TAG}_bench.json
