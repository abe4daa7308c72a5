# round 5 (second session): ring / pipeline sweep on the final kernels
tools/envsweep.sh 2 "XCX_X=0" "XCX_RING_MB=190" "XCX_RING_MB=220" "XCX_CHAINS=3 XCX_RING_MB=200" "XCX_CHAINS=3 XCX_RING_MB=240" "XCX_GROUPS=0" > gpurun_out/r5M_sweep.txt 2>&1
cat gpurun_out/r5M_sweep.txt
