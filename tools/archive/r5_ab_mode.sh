mkdir -p gpurun_out/r5g
export SHAPE_CACHE_DIR=/tmp/shape_cache
O=gpurun_out/r5g/ab_mode.txt; : > $O
for v in main scan; do
  if [ $v = main ]; then L=""; else L="PDHG_HIP_LIB=$PWD/firstorderlp.jl_amd/csrc/variants/libpdhg_$v.so"; fi
  for mode in 0 1 2; do
    echo "== $v PDHG_TW_MODE=$mode" >> $O
    env $L timeout 600 python tools/shape_table.py --no-vendor --only "${SHAPES:-clustered}" --env "PDHG_TW_MODE=$mode" 2>&1 | grep -v "^#" | cut -c1-140 >> $O
  done
done
cat $O
