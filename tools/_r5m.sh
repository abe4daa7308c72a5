# round 5 (second session): column and row pass on disjoint CU sets (XCX_SPLIT, experiment)
XCX_SPLIT=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r5m_pytest.txt 2>&1; tail -3 gpurun_out/r5m_pytest.txt
tools/envsweep.sh 2 \
 "XCX_SPLIT=0" \
 "XCX_SPLIT=1 XCX_SPLIT_MASK=0" \
 "XCX_SPLIT=1 XCX_SPLIT_MASK=1" \
 "XCX_SPLIT=1 XCX_SPLIT_MASK=2" \
 "XCX_SPLIT=1 XCX_SPLIT_MASK=1 XCX_SPLIT_COLS=14" \
 "XCX_SPLIT=1 XCX_SPLIT_MASK=1 XCX_SPLIT_COLS=18" \
 "XCX_SPLIT=1 XCX_SPLIT_MASK=2 XCX_SPLIT_COLS=14" \
 "XCX_SPLIT=1 XCX_SPLIT_MASK=2 XCX_SPLIT_COLS=18" \
 > gpurun_out/r5m_sweep.txt 2>&1
cat gpurun_out/r5m_sweep.txt
