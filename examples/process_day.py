"""
End-to-end example on synthetic data, following xcorr's examples/core/process.py and
scripts/snr.py: init -> process -> write -> read -> postprocess -> snr, everything on the GPU
through xcorr_amd.  Run:  python examples/process_day.py [output_dir]
"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xcorr_amd as xcorr                      # noqa: E402
from xcorr_amd import core, io, signal         # noqa: E402
from xcorr_amd.synth import synthetic_traces   # noqa: E402


def main(root):
    fs = 50.
    hours = 6
    tr = synthetic_traces(3, int(hours * 3600 * fs))            # common signal, 0.5 s moveout per station
    names = ['XX.STA%d..HHZ' % i for i in range(3)]
    client = core.ArrayClient(dict(zip(names, tr)), fs, t0='2020-01-01')   # stands in for xcorr.Client (ObsPy waveforms)
    attrs = {'institution': 'example', 'author': 'example', 'source': 'synthetic'}
    files = []
    for i in range(3):
        for j in range(i + 1, 3):
            pair = f'{names[i]}-{names[j]}'
            ds = xcorr.init(pair, '2020-01-01', f'2020-01-01T0{hours}:00:00', preprocess={}, attrs=attrs,
                            sampling_rate=fs, window_length=3600., window_overlap=.5, clip_lag=(-60., 60.),
                            hash_waveforms=True)
            xcorr.process(ds, client, hash_waveforms=True)
            path = io.ncfile(pair, ds.time[0], root)
            io.write(ds, path)
            files.append(path)
            print(f'{pair}: {int((ds.status == 1).sum())} of {ds.status.size} windows -> {os.path.relpath(path, root)}')
    ds = xcorr.mfread(files)                                     # all pairs of the day as one container
    pp = xcorr.postprocess(ds, lag_lim=(-30., 30.), filter_kwargs={'frequency': 3., 'order': 2})
    lag = pp.lag
    valid = pp.status == 1
    sig = np.abs(lag) <= 5.
    noi = np.abs(lag) >= 15.
    for ip, pair in enumerate(pp.pair):
        rows = pp.cc[ip][valid[ip]]
        ratio = signal.snr(rows, sig, noi)                      # max |cc| in the signal window / rms of the noise window
        k = int(np.argmax(np.abs(rows.mean(axis=0))))
        print(f'{pair}: stack peak at lag {lag[k]:+.2f} s, median snr {np.median(ratio):.1f}')
    return pp


if __name__ == '__main__':
    out = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix='xcorr_amd_example_')
    main(out)
