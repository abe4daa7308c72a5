# round 5 (second session): front end with non-temporal accesses; does it then overlap with the pair phase?
P=$PWD/xcorr_amd/lib
tools/envsweep.sh 3 \
 "XCX_LIB_PATH=$P/libxcx.so" \
 "XCX_LIB_PATH=$P/libxcx_fnt.so" \
 "XCX_LIB_PATH=$P/libxcx.so -- --pipeline-front-end" \
 "XCX_LIB_PATH=$P/libxcx_fnt.so -- --pipeline-front-end" \
 > gpurun_out/r5n_sweep.txt 2>&1
cat gpurun_out/r5n_sweep.txt
