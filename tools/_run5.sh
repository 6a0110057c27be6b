timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_harness.py -q -x -k "fused or registration or deterministic or graphed or fragment or demo or config3 or payload" 2>&1 | tail -3
tools/kt.sh b 8,4096,4096,10000 200
tools/kt.sh b 1,1024,1024,20000 200
