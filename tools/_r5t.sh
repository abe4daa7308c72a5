# round 5 (second session): zero-stretch skip as a template parameter of the column body -- suite + A/B
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5t_pytest.txt 2>&1; tail -3 gpurun_out/r5t_pytest.txt
tools/envsweep.sh 3 "XCX_ZSKIP=0" "XCX_ZSKIP=1" > gpurun_out/r5t_sweep.txt 2>&1
cat gpurun_out/r5t_sweep.txt
