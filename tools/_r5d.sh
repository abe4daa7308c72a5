run() { # label, env...
  lab=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lab', round(d['value']), round(d['ms_per_step'],1), d['config']['pair_batch'])"
}
for r in 1 2; do
  run "f32_c2" XCX_T_STORAGE=f32
  run "q24_c2" XCX_T_STORAGE=q24
  run "q24_c3_r252" XCX_T_STORAGE=q24 XCX_CHAINS=3 XCX_RING_MB=252
  run "q24_c3_r200" XCX_T_STORAGE=q24 XCX_CHAINS=3 XCX_RING_MB=200
  run "q24_c2_r213" XCX_T_STORAGE=q24 XCX_RING_MB=213
  run "q24_c4_r336" XCX_T_STORAGE=q24 XCX_CHAINS=4 XCX_RING_MB=336
  run "f32_c3_r252" XCX_T_STORAGE=f32 XCX_CHAINS=3 XCX_RING_MB=252
done > gpurun_out/r5d_sweep.txt 2>&1
cat gpurun_out/r5d_sweep.txt
