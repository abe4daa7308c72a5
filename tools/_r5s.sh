# round 5 (second session): column loads of the known-zero stretch not issued at all (diagnostic build, fixed band)
P=$PWD/xcorr_amd/lib
XCX_LIB_PATH=$P/libxcx_zs.so timeout -k 10 600 python -m pytest tests/test_gpu_config3.py -x -q -m gpu > gpurun_out/r5s_pytest.txt 2>&1; tail -3 gpurun_out/r5s_pytest.txt
tools/ab.sh "libxcx.so libxcx_zs.so" 4 > gpurun_out/r5s_ab.txt 2>&1
cat gpurun_out/r5s_ab.txt
