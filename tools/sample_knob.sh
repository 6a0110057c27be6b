for flags in "-DSAMPLE_KNOB=1" "-DSAMPLE_KNOB=2" ""; do
  RRL_HIPCC_FLAGS="$flags" python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1 || { echo "$flags: BUILD FAILED"; continue; }
  ( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sk -o d -- python3 /root/repo/tools/demo_graph_prof.py 600 > /tmp/sk.log 2>&1 )
  echo "[$flags] $(grep sample_count /tmp/sk/d_kernel_stats.csv | awk -F, '{print $(NF-4)}')"
done
python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1
