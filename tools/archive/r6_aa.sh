# round 6: shared wave reduction (WaveSplit) against the one-tree-per-quantity build: bits and cost of the checks
export PDHG_DEV=1
V=$PWD/firstorderlp.jl_amd/csrc/variants
for w in l1svm pagerank 200000; do
  for lib in "" $V/libpdhg_nosplit.so; do PDHG_HIP_LIB=$lib python tools/wave_split_ab.py $w 2>&1 | tail -1 | sed "s|^|${lib##*/} |"; done
done
