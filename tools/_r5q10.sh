# round 5 (third session): cc2d -- S2 column-block-major, T2 row-group sweep (row kernel: one per-lane offset + uniform stride)
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_cc2d.py tests/test_gpu_spectrogram.py -x -q > gpurun_out/q10_pytest.log 2>&1; rc=$?; tail -2 gpurun_out/q10_pytest.log
[ $rc -ne 0 ] && exit 1
for l in libxcx.so libxcx_t8.so; do
XCX_CC2D_PASS_MB=64 XCX_LIB_PATH=$PWD/xcorr_amd/lib/$l timeout -k 10 200 python tools/check_cc2d.py 25 61 > gpurun_out/q10_check_$l.log 2>&1 || { tail -5 gpurun_out/q10_check_$l.log; exit 1; }
done
tail -q -n 1 gpurun_out/q10_check_*.log
rm -f gpurun_out/q10_ab.log
for rep in 1 2 3; do
  for l in libxcx_t1.so libxcx_t8.so libxcx_t32.so libxcx_t128.so libxcx.so; do
    XCX_LIB_PATH=$PWD/xcorr_amd/lib/$l timeout -k 10 200 python tools/bench_timelapse.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', round(d['value']), round(d['roofline']['us_per_tile_pair'],2))" >> gpurun_out/q10_ab.log || exit 1
  done
done
cat gpurun_out/q10_ab.log
for l in libxcx_t1.so libxcx_t8.so libxcx.so; do
XCX_LIB_PATH=$PWD/xcorr_amd/lib/$l XCX_CC2D_OVERLAP=0 tools/pmc_cc2d.sh q10_$l > gpurun_out/q10_pmc_$l.log 2>&1; awk -F'","|",|,' '/k2d_rows_inv|k2d_cols/{print "'$l'", substr($1,1,50), $(NF-4)}' gpurun_out/cc2d_q10_$l/kernel_stats.csv | head -2
done
