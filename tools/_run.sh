{
echo "== pagerank 1M"; PDHG_VERBOSE=1 timeout 600 python tools/tune_tiled.py --pagerank 1000000 --reps 1 --steps 40 "PDHG_SPMV=stream" "PDHG_SPMV=tiled PDHG_TW_NNZ_CAP=1.0" "PDHG_SPMV=tiled PDHG_TW_NNZ_CAP=1.0 PDHG_TILE_COLS=24576" "PDHG_SPMV=tiled PDHG_TW_NNZ_CAP=1.0 PDHG_TILE_COLS=16384" "PDHG_SPMV=tiled PDHG_TW_NNZ_CAP=1.0 PDHG_TILE_COLS=8192" "PDHG_SPMV=tiled PDHG_TW_NNZ_CAP=1.2 PDHG_TILE_COLS=16384"
} 2>&1 | grep "rep\|==\|tiled layout.*waves" | cut -c1-260
