mkdir -p gpurun_out/r5c
timeout -k 10 300 python -m pytest tests/test_gpu_farm_window.py -q -m gpu -x > gpurun_out/r5c/tests.log 2>&1; tail -3 gpurun_out/r5c/tests.log
rm -f gpurun_out/r5c/nsplit2.txt
for ns in 1 2 4 8 16 32; do
  echo "=== MGPU_PAIR_NSPLIT=$ns" >> gpurun_out/r5c/nsplit2.txt
  MGPU_PAIR_NSPLIT=$ns timeout -k 10 200 python tools/farm_window_speed.py --replicas 8,32,64,128,256,512,1024 --modes w2 --lanes 1,2 --seconds 0.3 >> gpurun_out/r5c/nsplit2.txt 2>&1
done
cat gpurun_out/r5c/nsplit2.txt
