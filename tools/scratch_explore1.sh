mkdir -p gpurun_out/r5g
timeout -k 10 600 python -m pytest tests/test_gpu_windows_oracle.py tests/test_gpu_farm_window.py tests/test_gpu_chain.py tests/test_gpu_parity.py tests/test_bench_cli.py -q -m gpu -x > gpurun_out/r5g/tests.log 2>&1; tail -15 gpurun_out/r5g/tests.log
