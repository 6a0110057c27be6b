for cfg in "8,4096,4096,10000" "1,1024,1024,20000"; do
 for red in single tiled; do for sp in 1 2 4; do
  echo "== $cfg reduce=$red sort_parts=$sp"; RRL_REDUCE=$red RRL_SORT_PARTS=$sp tools/kt.sh a "$cfg" 200
 done; done
done
