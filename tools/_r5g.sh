timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5g_pytest.txt 2>&1
tail -4 gpurun_out/r5g_pytest.txt
tools/prof.sh r05a > gpurun_out/r5g_prof.txt 2>&1; tail -5 gpurun_out/r5g_prof.txt
XCX_T_STORAGE=q24 timeout -k 10 600 python tools/t24_error_table.py > gpurun_out/r05_t24_error_table.txt 2>&1; tail -1 gpurun_out/r05_t24_error_table.txt
( echo "# tools/stamps.py on the make STAMPS=1 build of the round-5 kernels, float2 storage of T (XCX_T_STORAGE=f32)"; XCX_T_STORAGE=f32 timeout -k 10 300 python tools/stamps.py;
  echo; echo "# ... and q24 storage (the default of whitened plans)"; XCX_T_STORAGE=q24 timeout -k 10 300 python tools/stamps.py ) > gpurun_out/r05_stamps.txt 2>&1
tail -12 gpurun_out/r05_stamps.txt
