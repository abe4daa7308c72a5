( echo "# tools/stamps.py on the make STAMPS=1 build of the round-5 kernels, float2 storage of T (XCX_T_STORAGE=f32)"; XCX_T_STORAGE=f32 timeout -k 10 300 python tools/stamps.py;
  echo; echo "# ... and q24 storage (the default of whitened plans)"; XCX_T_STORAGE=q24 timeout -k 10 300 python tools/stamps.py ) > gpurun_out/r05_stamps.txt 2>&1
tail -8 gpurun_out/r05_stamps.txt
run() { lab=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lab', round(d['value']), round(d['ms_per_step'],1), d['config']['pair_batch'])"
}
for r in 1 2; do
  run "groups_B28" XCX_GROUPS=1
  run "plain_B28" XCX_GROUPS=0
  run "plain_B42_ring250" XCX_GROUPS=0 XCX_RING_MB=251
  run "plain_B56_ring334" XCX_GROUPS=0 XCX_RING_MB=335
  run "plain_B35_ring210" XCX_GROUPS=0 XCX_RING_MB=210
done > gpurun_out/r5h_sweep.txt 2>&1
cat gpurun_out/r5h_sweep.txt
timeout -k 10 300 python bench.py --gpus 2 --oversubscribe --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5h_selflaunch_n2.txt 2>&1; echo "rc=$?" >> gpurun_out/r5h_selflaunch_n2.txt; tail -2 gpurun_out/r5h_selflaunch_n2.txt | cut -c1-400
timeout -k 10 300 python bench.py --gpus 2 --steps 2 > gpurun_out/r5h_selflaunch_refused.txt 2>&1; echo "rc=$?" >> gpurun_out/r5h_selflaunch_refused.txt; tail -2 gpurun_out/r5h_selflaunch_refused.txt | cut -c1-300
timeout -k 10 300 python bench.py --stack --days 4 --gpus 2 --oversubscribe --steps 1 --warmup 0 > gpurun_out/r5h_selflaunch_stack.txt 2>&1; echo "rc=$?" >> gpurun_out/r5h_selflaunch_stack.txt; tail -4 gpurun_out/r5h_selflaunch_stack.txt | cut -c1-300
