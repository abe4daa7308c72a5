# round 5 (second session): shader / memory clocks and power while the bench runs (rocm-smi sampled every 0.25 s)
( timeout -k 10 300 python bench.py --steps 60 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/r5S_bench.txt 2>&1 ) &
BP=$!
sleep 8
for i in $(seq 1 40); do
  rocm-smi --showclocks --showpower --showtemp --json 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.load(sys.stdin); c=d[list(d.keys())[0]]
    print({k:v for k,v in c.items() if any(x in k.lower() for x in ('sclk','mclk','fclk','power','temp'))})
except Exception as e: print('err',e)
"
  sleep 0.25
done > gpurun_out/r5S_clocks.txt 2>&1
wait $BP
tail -1 gpurun_out/r5S_bench.txt | cut -c1-120
sort gpurun_out/r5S_clocks.txt | uniq -c | sort -rn | head -12
