# round 5 (second session): scalar gstep lookup, q24 storage of the forward intermediate -- suite + A/B
P=$PWD/xcorr_amd/lib
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5w_pytest.txt 2>&1; tail -3 gpurun_out/r5w_pytest.txt
tools/envsweep.sh 3 "XCX_LIB_PATH=$P/libxcx_nosc.so XCX_TF_STORAGE=f32" "XCX_LIB_PATH=$P/libxcx.so XCX_TF_STORAGE=f32" "XCX_LIB_PATH=$P/libxcx.so" > gpurun_out/r5w_sweep.txt 2>&1
cat gpurun_out/r5w_sweep.txt
for v in f32 q24; do XCX_TF_STORAGE=$v timeout -k 10 300 python bench.py --stack --days 30 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stack tf=$v', round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernels_us'])"; done > gpurun_out/r5w_stack.txt 2>&1
cat gpurun_out/r5w_stack.txt
