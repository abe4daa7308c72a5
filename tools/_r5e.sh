XCX_T_STORAGE=q24 timeout -k 10 600 python tools/t24_error_table.py > gpurun_out/r5e_t24_error_table.txt 2>&1
tail -2 gpurun_out/r5e_t24_error_table.txt
for r in 1 2 3; do
  for v in "libxcx.so f32" "libxcx.so q24" "libxcx_q24pr.so q24"; do
    set -- $v
    XCX_LIB_PATH=$PWD/xcorr_amd/lib/$1 XCX_T_STORAGE=$2 timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2', round(d['value']), round(d['ms_per_step'],1))"
  done
done > gpurun_out/r5e_ab.txt 2>&1
cat gpurun_out/r5e_ab.txt
timeout -k 10 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_config4.py tests/test_gpu_multirank.py -x -q > gpurun_out/r5e_pytest.txt 2>&1
tail -15 gpurun_out/r5e_pytest.txt
