"""tools/run_file.py: a recording in the reference's .bin format in, position fixes out (the
reference's file mode, README.md:177-187).  The checkout holds no recording, so a
synthetic one is written with gpsmi.synth_nav (the scene of tests/test_gpu_position.py) and the
command must print a fix within 5 m of the scene's truth (north_star's bar)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SECONDS = 22.0


def write_recording(path, scene, n_blocks):
    with open(path, 'wb') as f:
        for b in range(n_blocks):
            scene.block_raw(b).astype('<u2').tofile(f)


def test_reader_and_cpu_cold_acquisition(tmp_path):
    """CPU: the file loop (START_STREAM, short last block) and the configs[0] leg."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import run_file
    from gpsmi import ingest, synth
    sc = synth.default_scene(6, seed=7)
    path = str(tmp_path / 'rec.bin')
    write_recording(path, sc, 6)
    with open(path, 'ab') as f:
        f.write(b'\x00' * 1000)                                   # a short tail ends the stream
    blocks = list(ingest.read_raw_blocks(path))
    assert len(blocks) == 6 and np.array_equal(blocks[2], sc.block_raw(2))
    assert len(list(ingest.read_raw_blocks(path, start_stream=4))) == 2
    acq = run_file.cpu_cold_acquisition(path)
    assert acq['blocks'] == 5 and acq['wall_ms'] > 0
    assert {s for s, _, _ in acq['found']} == {s.prn for s in sc.sats}


@pytest.mark.gpu
def test_recording_to_position_fix(tmp_path):
    from gpsmi import position as P, synth_nav
    truth = np.array(P.geo_to_ecef(49.082961, 8.307581, 160.0))
    sc, info = synth_nav.geometric_scene(truth, SECONDS)
    path = str(tmp_path / 'synthetic.bin')
    write_recording(path, sc, int(SECONDS / 0.032))
    eph = str(tmp_path / 'gpsEphem.json')
    with open(eph, 'w') as f:
        json.dump({str(k): v for k, v in info['ephs'].items()}, f)
    pk = str(tmp_path / 'result.pickle')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'run_file.py'), path, '--ephemeris', eph,
                        '--save-pickle', pk, '--json'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['blocks'] == int(SECONDS / 0.032) and out['datagrams'] >= 15 and len(out['tracked']) >= 6
    assert out['fixes'] > 150
    pos = np.array(out['position']['ecef_m'])
    assert np.linalg.norm(pos - truth) < 5.0
    assert abs(out['position']['lat_deg'] - 49.082961) < 1e-4 and abs(out['position']['lon_deg'] - 8.307581) < 1e-4
    import pickle
    with open(pk, 'rb') as f:                                     # (our own run's SAVE_PICKLE file)
        dgs = pickle.load(f)
    assert len(dgs) == out['datagrams'] and len(pickle.loads(dgs[0])) == 3
