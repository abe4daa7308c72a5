# round 5 (second session): dual pair kernel for N1 = 3072 (N = 737 280: the lag-axis time shift and transforms)
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fft.py tests/test_gpu_fftlib.py tests/test_gpu_longwindow.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r5O_pytest.txt 2>&1; tail -2 gpurun_out/r5O_pytest.txt
for v in 0 1; do echo "XCX_DUAL=$v"; XCX_DUAL=$v timeout -k 10 300 python tools/time_post.py 1024 2>&1 | grep -E 'timeshift|whole'; done > gpurun_out/r5O_post.txt 2>&1; cat gpurun_out/r5O_post.txt
