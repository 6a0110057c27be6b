timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_harness.py -q -x -k "registration or tiled or fused or graphed or harness or demo or bench or config3 or fragment" 2>&1 | tail -6
for f in 1 0; do echo "== fused=$f"; RRL_STEP_FUSED=$f tools/kt.sh b 8,4096,4096,10000 200; RRL_STEP_FUSED=$f tools/kt.sh b 1,1024,1024,20000 200; done
