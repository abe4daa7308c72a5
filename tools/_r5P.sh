# round 5 (second session): forward column kernel at 4 waves per SIMD (it needs 121 VGPRs; its attribute held it to 3)
P=$PWD/xcorr_amd/lib
for r in 1 2 3; do for l in libxcx.so libxcx_fc4.so; do
  echo "$l $(XCX_LIB_PATH=$P/$l timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print(round(d['value']), round(d['ms_per_step'],2), round(k['k_fwd_rows'],1), round(k['k_fwd_cols'],1))")"
done; done > gpurun_out/r5P_ab.txt 2>&1; cat gpurun_out/r5P_ab.txt
