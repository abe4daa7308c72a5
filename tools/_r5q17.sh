# round 5 (third session): configs[4] with tiles built the way scripts/timelapse.py builds them (MFMA spectrogram of CCF-like rows)
timeout -k 10 200 python tools/bench_timelapse.py --steps 5 --from-ccf --no-cpu-baseline > gpurun_out/r05k_timelapse_from_ccf.txt 2>&1; tail -1 gpurun_out/r05k_timelapse_from_ccf.txt | cut -c1-400
