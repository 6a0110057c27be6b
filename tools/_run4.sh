timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "scan or cull or ragged or baseline or large_cloud or reuse or counters or full_size or public_loss or sort_parts or tiled or prepare or empty" 2>&1 | tail -3
for p in 1 0; do echo "== persist=$p"; RRL_CULL_PERSIST=$p tools/kt.sh b 8,4096,4096,10000 200; RRL_CULL_PERSIST=$p tools/kt.sh b 1,1024,1024,20000 200; RRL_CULL_PERSIST=$p tools/kt.sh b 64,4096,4096,10000 50; done
python3 tools/scan_tail.py 2>&1 | grep -v amdgpu.ids
