"""The two sinks of the hand-off (reference gpsrecv.py:513-517, :205-212): a UDP
datagram to the evaluation process and the SAVE_PICKLE file gpseval can replay
(gpseval.py:516-524).  No GPU."""
import pickle
import socket

import numpy as np


def test_udp_datagram_and_pickle_file_roundtrip(tmp_path):
    from gpsmi import pipeline
    frames = [{'ID': 2, 'tow': 1001, 'Crs': -12.5, 'ST': np.int64(123456), 'SAT': 9,
               'AMP': np.float32(7.5), 'CRM': np.float32(21.0), 'FRQ': np.float32(-1500.25),
               'SWP': False}]
    co_ph = {9: [(41, 700.31), (42, 700.28)], 23: [(41, 1501.9)]}
    res = pickle.dumps((65536, frames, co_ph))
    assert len(res) < 65504                                    # UDP_BUFSIZE_1, gpsglob.py:85
    rx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    rx.bind(('127.0.0.1', 0))
    rx.settimeout(5)
    tx = pipeline.make_udp_socket()
    pipeline.send_udp(tx, res, '127.0.0.1', rx.getsockname()[1])
    msg, _ = rx.recvfrom(65504)
    tx.close(); rx.close()
    skipped, fl, cp = pickle.loads(msg)                        # what gpseval.py:531 does
    assert skipped == 65536 and fl == frames and cp == co_ph
    path = tmp_path / 'x_gpsResult.pickle'
    pipeline.save_results(str(path), [res, res])
    with open(path, 'rb') as f:
        lst = pickle.load(f)
    assert [pickle.loads(r) for r in lst] == [(65536, frames, co_ph)] * 2
