set -x
export XCX_LIB_PATH=$PWD/xcorr_amd/lib/libxcx_q24.so
timeout -k 10 600 python tools/t24_error_table.py > gpurun_out/r5a_t24_error_table.txt 2>&1
tail -5 gpurun_out/r5a_t24_error_table.txt
for r in 1 2 3; do for ts in f32 q24; do
  XCX_T_STORAGE=$ts timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$ts', round(d['value']), round(d['ms_per_step'],1))"
done; done > gpurun_out/r5a_ab.txt 2>&1
cat gpurun_out/r5a_ab.txt
XCX_T_STORAGE=q24 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config3.py -x -q > gpurun_out/r5a_pytest_q24.txt 2>&1
tail -5 gpurun_out/r5a_pytest_q24.txt
