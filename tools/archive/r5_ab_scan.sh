mkdir -p gpurun_out/r5g
O=gpurun_out/r5g/${OUT:-ab}.txt; : > $O
S="${SHAPES:-config S,clustered,column-skewed,3000000,1M x 1M,30 per row}"
for v in ${VARIANTS:-main}; do
  echo "== $v" >> $O
  if [ $v = main ]; then L=""; else L="PDHG_HIP_LIB=$PWD/firstorderlp.jl_amd/csrc/variants/libpdhg_$v.so"; fi
  env $L timeout 600 python tools/shape_table.py --no-vendor --only "$S" 2>&1 | grep -v "^#" | cut -c1-140 >> $O
done
cat $O
