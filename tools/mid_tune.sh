mkdir -p gpurun_out
{
for K in 6 7 12 14 16; do
  echo "== random 4M k=$K"
  timeout 300 python tools/tune_tiled.py --m 4000000 --n 4000000 --k $K --reps 1 --steps 40 "PDHG_TILE_SHIFT=14" "PDHG_TILE_SHIFT=15" "PDHG_TILE_SHIFT=16"
done
echo "== shard shapes of config S: rows x 10M"
timeout 300 python tools/tune_tiled.py --m 1250000 --n 10000000 --k 10 --reps 1 --steps 40 "PDHG_TILE_SHIFT=16" "PDHG_TILE_SHIFT=17" "PDHG_TILE_SHIFT=18" "PDHG_TILE_SHIFT=19"
timeout 300 python tools/tune_tiled.py --m 2500000 --n 10000000 --k 10 --reps 1 --steps 40 "PDHG_TILE_SHIFT=16" "PDHG_TILE_SHIFT=17" "PDHG_TILE_SHIFT=18"
timeout 300 python tools/tune_tiled.py --m 5000000 --n 10000000 --k 10 --reps 1 --steps 40 "PDHG_TILE_SHIFT=15" "PDHG_TILE_SHIFT=16" "PDHG_TILE_SHIFT=17"
echo "== random 30M k=10"
timeout 600 python tools/tune_tiled.py --m 30000000 --n 30000000 --k 10 --reps 1 --steps 20 "PDHG_TILE_SHIFT=16" "PDHG_TILE_SHIFT=17" "PDHG_TILE_SHIFT=18"
} > gpurun_out/mid_tune3.txt 2>&1
grep -v "^+" gpurun_out/mid_tune3.txt | tail -80
