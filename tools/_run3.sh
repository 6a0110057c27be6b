timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "scan or cull or ragged or baseline or large_cloud or reuse or counters or full_size or public_loss or sort_parts or tiled or prepare" 2>&1 | tail -3
tools/kt.sh b 8,4096,4096,10000 200
tools/kt.sh b 1,1024,1024,20000 200
tools/pmc_any.sh p3 cull_scan -- tools/step_loop.py 8,4096,4096,10000 20 2>&1 | grep -v "^ \"_\|launches" | tr '\n' ' '
