for g in 8,8 8,4 8,2 8,1 4,8 4,4 4,2 4,1 2,4 2,2 2,1 1,1; do
  echo "== $g"; RRL_CULL_GEOM=$g python tools/config_sweep.py 1,1024,1024,20000 1,16384,16384,512 8,16384,16384,512 8,2048,1024,10000 2>/dev/null | cut -c1-130
done
