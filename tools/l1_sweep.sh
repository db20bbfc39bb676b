for L in 36 40 43 44 48 56; do
  OZK_MSM_L1=$L timeout -k 10 100 python bench.py --no-cpu-baseline --in-flight 1 --steps 40 > gpurun_out/r2_l1sweep_if1_$L.json 2>/dev/null
  OZK_MSM_L1=$L timeout -k 10 100 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/r2_l1sweep_if2_$L.json 2>/dev/null
done
echo done
