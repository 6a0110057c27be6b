cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/c5 -o w -- python3 /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-graph --batch 1 --points 16384 --lines 512 > /dev/null 2>&1
python3 - <<'PY'
import csv
for r in list(csv.DictReader(open("/root/repo/gpurun_out/c5/w_kernel_stats.csv")))[:9]:
    print(f"{r['Name'][:44]:44s} {r['Calls']:>4s} {float(r['AverageNs'])/1e3:8.1f}")
PY
