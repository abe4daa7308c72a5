# round 5 (second session): persistent row pass of the 2-D pair transform (next row in flight)
P=$PWD/xcorr_amd/lib
timeout -k 10 600 python -m pytest tests/test_gpu_cc2d.py -x -q -m gpu > gpurun_out/r5r_pytest.txt 2>&1; tail -3 gpurun_out/r5r_pytest.txt
for r in 1 2; do
for v in "XCX_LIB_PATH=$P/libxcx_base.so" "XCX_LIB_PATH=$P/libxcx.so" "XCX_LIB_PATH=$P/libxcx.so XCX_CC2D_ROW_WGS=4" "XCX_LIB_PATH=$P/libxcx.so XCX_CC2D_ROW_WGS=6" "XCX_LIB_PATH=$P/libxcx_r4.so" "XCX_LIB_PATH=$P/libxcx_r4.so XCX_CC2D_ROW_WGS=8"; do
  echo "[$v] $(env $v timeout -k 10 300 python tools/bench_timelapse.py --steps 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['check'])")"
done; done > gpurun_out/r5r_sweep.txt 2>&1
cat gpurun_out/r5r_sweep.txt
