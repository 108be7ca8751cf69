#!/bin/bash
set -e -o pipefail
out=gpurun_out/r3q9
mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_gcmc.py tests/test_gpu_farm.py -m gpu -x -q -k "framework or decided" > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
python tools/bench_kernels.py --replicas 2048 --reps 5 --workload framework_water --decide > $out/k_fw.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof_fw -o p -- python3 /root/repo/tools/bench_kernels.py --replicas 2048 --reps 5 --workload framework_water --decide > /root/repo/$out/k_fw_rocprof.json)
rm -f $out/prof_fw/*kernel_trace.csv
python bench.py --workload framework_water --no-cpu-baseline --steps 400 > $out/b_fw.json 2> $out/b_fw.err
echo done
