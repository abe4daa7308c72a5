# round 5 (second session): ring / pipeline sweep on the zero-skip kernel
tools/envsweep.sh 2 "XCX_X=0" "XCX_RING_MB=130" "XCX_RING_MB=200" "XCX_CHAINS=3 XCX_RING_MB=200" "XCX_CHAINS=3 XCX_RING_MB=240" "XCX_FLOW=1 XCX_T_STORAGE=f32" "XCX_T_STORAGE=f32" > gpurun_out/r5u_sweep.txt 2>&1
cat gpurun_out/r5u_sweep.txt
