# dev: the one-launch trust-region searches with fewer / more workgroups (L1-SVM check stages, tools/solve_demo.py --breakdown)
export PDHG_DEV=1
mkdir -p gpurun_out/r5g
O=gpurun_out/r5g/tr_wgs.txt; : > $O
for w in default 32 64 96 128 160 256; do
  echo "== PDHG_TR_COOP_WGS=$w" >> $O
  if [ $w = default ]; then E=""; else E="PDHG_TR_COOP_WGS=$w"; fi
  env $E timeout 600 python tools/solve_demo.py --workload ${WL:-l1svm} --iteration_limit 40000 --verbosity 0 --breakdown 2>/dev/null | grep -E "OPTIMAL|run_restart|update_obj|iteration_stats" >> $O
done
cat $O
