mkdir -p gpurun_out/r5h
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_farm.py tests/test_gpu_run.py tests/test_gpu_gcmc.py -q -m gpu -x -k "tilt or triclinic or golden or batched_candidates or consistency or run or gcmc" > gpurun_out/r5h/tests2.log 2>&1; tail -15 gpurun_out/r5h/tests2.log
