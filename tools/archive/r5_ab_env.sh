# dev: shape table lines under several environments of the product library:  ENVS="A=1|B=2 C=3" SHAPES="..." OUT=name
mkdir -p gpurun_out/r5g
export SHAPE_CACHE_DIR=/tmp/shape_cache PDHG_DEV=1
O=gpurun_out/r5g/${OUT:-ab_env}.txt; : > $O
IFS='|' read -ra E <<< "${ENVS:-PDHG_DEV=1}"
for e in "${E[@]}"; do
  echo "== $e" >> $O
  timeout 900 python tools/shape_table.py --no-vendor --only "${SHAPES:-config S}" --env "$e" 2>&1 | grep -v "^#" | cut -c1-170 >> $O
done
cat $O
