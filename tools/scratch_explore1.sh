mkdir -p gpurun_out/r5e
for bc in 0 1 0 1; do
echo "== by_chain $bc" >> gpurun_out/r5e/speed2.txt
MGPU_FARM_BY_CHAIN=$bc timeout -k 10 200 python tools/farm_window_speed.py --replicas 8,64,256,512 --modes w2 --lanes 1,2 --seconds 0.4 >> gpurun_out/r5e/speed2.txt 2>&1
done
cat gpurun_out/r5e/speed2.txt
