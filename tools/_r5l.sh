# round 5 (second session): does the smaller q24 ring let the CCF go through the Infinity Cache (plain stores)?
P=$PWD/xcorr_amd/lib
tools/envsweep.sh 2 \
 "XCX_LIB_PATH=$P/libxcx.so" \
 "XCX_LIB_PATH=$P/libxcx_plain.so" \
 "XCX_LIB_PATH=$P/libxcx_plain.so XCX_RING_MB=100" \
 "XCX_LIB_PATH=$P/libxcx_plain.so XCX_RING_MB=64" \
 "XCX_LIB_PATH=$P/libxcx.so XCX_RING_MB=100" \
 "XCX_LIB_PATH=$P/libxcx.so XCX_RING_MB=64" \
 "XCX_LIB_PATH=$P/libxcx_plain.so XCX_RING_MB=100 XCX_CHAINS=3" \
 "XCX_LIB_PATH=$P/libxcx_plain.so XCX_RING_MB=64 XCX_CHAINS=3" \
 "XCX_LIB_PATH=$P/libxcx.so XCX_FLOW=1" \
 "XCX_LIB_PATH=$P/libxcx_plain.so XCX_FLOW=1" \
 > gpurun_out/r5l_sweep.txt 2>&1
cat gpurun_out/r5l_sweep.txt
