# round 5 (second session): 2-D pair column pass addressed from scalar bases, rows without map stores
P=$PWD/xcorr_amd/lib
timeout -k 10 600 python -m pytest tests/test_gpu_cc2d.py tests/test_gpu_spectrogram.py -x -q -m gpu > gpurun_out/r5L_pytest.txt 2>&1; tail -2 gpurun_out/r5L_pytest.txt
for r in 1 2 3; do for l in libxcx_prev.so libxcx.so libxcx_c4.so; do
  echo "$l $(XCX_LIB_PATH=$P/$l timeout -k 10 300 python tools/bench_timelapse.py --steps 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['check'])")"
done; done > gpurun_out/r5L_ab.txt 2>&1; cat gpurun_out/r5L_ab.txt
