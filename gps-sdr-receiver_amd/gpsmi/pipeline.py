"""The receiver's block loop and its hand-off to the evaluation process.

Mirror of the data path of ``gpsrecv.processData`` (reference
src/gpsrecv.py:445-548) without its asyncio / socket plumbing: feed it the
32-ms blocks in order, it runs the cold sweep, selects satellites, tracks them
and returns, whenever the reference would send one, the hand-off datagram
``pickle.dumps((skippedData, frameLst, coPhLst))`` (gpsrecv.py:496-519).
``Receiver.command(b'SWEEP')`` / ``b'STOP'`` are the two commands the
evaluation side can send back (gpsrecv.py:522-536).  ``send_udp`` and
``save_results`` are the reference's two sinks (gpsrecv.py:513-517, :205-212).
"""
import pickle
import socket

import numpy as np

from . import receiver as R
from .acquisition import SAT_ALL, Acquisition, getNewSats
from .engine import Config

UDP_IP = '127.0.0.1'            # gpsglob.py:79
UDP_PORT = 61431                # gpsglob.py:82


class Receiver:
    def __init__(self, cfg=None, sat_all=None, raw_u8=False, report_lag=0):
        """raw_u8: feed() takes the recorder's uint16 (Q << 8 | I) blocks exactly as streamData
        reads them from the file (gpsrecv.py:162-173); the decode to complex64 happens inside
        the GPU kernels, every datagram is byte-identical to the complex64 path's.
        report_lag = L: feed() returns a datagram L blocks after the block the reference sends it
        on (0, the default: on that block) -- the once-a-second host work then runs while the GPU
        has L newer blocks queued instead of idling through it; the datagrams are the same."""
        self.cfg = cfg or Config()
        self.raw_u8 = bool(raw_u8)
        self.sat_all = list(SAT_ALL if sat_all is None else sat_all)
        self.acq = Acquisition(self.cfg, raw_u8=self.raw_u8)
        self.pool, self.pool_no, self.pool_worker = R.initMultiProcPool(self.cfg.max_sat,
                                                                        self.cfg, self.raw_u8)
        self.pool.report_lag = int(report_lag)
        self.running = True
        self.smp_time = np.int64(0)                  # SMP_TIME, gpsrecv.py:29
        self.act_sat_set = set()
        self.co_ph_lst, self.cp_q_lst = {}, {}
        self.skipped_data = 0
        self.result_list = []                        # RESULT_LIST (SAVE_PICKLE)
        self._start_sweep()

    def _start_sweep(self, first=True):              # gpsrecv.py:450-451, :462-463, :529-533
        self.sweep_all_freq = True
        if first:
            self.doppler_freq = self.cfg.min_freq    # a SWEEP command keeps the current bin
        self.sat_lst = self.sat_all.copy()
        self.found_sats = []

    def command(self, msg):
        """b'SWEEP' restarts the cold search, b'STOP' ends the run."""
        if msg == b'SWEEP':
            self.drain()                             # (cpQLst must be current for getNewSats)
            self._start_sweep(first=False)
        elif msg == b'STOP':
            self.running = False

    def feed(self, data, skip=0):
        """One block (complex64[NGPS], or uint16[NGPS] when raw_u8); `skip` = streams lost before it
        (gpsrecv.py:469-471).  Returns the pickled hand-off or None."""
        c = self.cfg
        self.skipped_data += skip * c.ngps
        self.smp_time += (1 + skip) * c.ngps
        if self.sweep_all_freq:
            ready, self.doppler_freq, self.found_sats = self.acq.sweepAllSats(
                data, self.doppler_freq, self.sat_lst, self.found_sats,
                itSweep=c.it_sweep_all)
            if ready:
                self.sweep_all_freq = False
                dele, new = getNewSats(self.act_sat_set, self.found_sats, self.cp_q_lst,
                                       c.max_sat)
                if dele:
                    self.pool_worker, self.act_sat_set = R.delPoolStreams(
                        self.pool, self.pool_no, self.pool_worker, self.act_sat_set, dele)
                self.pool_worker, self.act_sat_set = R.initPoolStreams(
                    self.pool, self.pool_no, self.pool_worker, self.act_sat_set, new,
                    self.found_sats)
            return None
        if not self.act_sat_set:                     # (nothing acquired: the reference's satCalc
            return None                              # returns [], no datagram)
        # the block is enqueued on the GPU; the host catches up once a second, at the block
        # the reference sends its datagram on (R.satCalcLazy)
        res = None
        for batch in R.satCalcLazy(self.act_sat_set, self.pool, self.pool_worker, data,
                                   self.smp_time):
            res = self.hand_off(batch) or res
        return res

    def drain(self):
        """Absorb the blocks that are still on their way (before anything that reads the
        channel state from outside: a command, close, a test)."""
        self.pool.absorb_pending()
        for batch in self.pool.take_done():
            self.hand_off(batch)

    def hand_off(self, batch):
        """gpsrecv.py:496-519 for the K blocks of a batch, the last of which carries the
        frames: every block appends (streamNo, coPh) for the channels that correlated -- a
        satellite enters coPhLst at its first such block, in the order satCalc returned the
        results -- and when any frame exists (once a second) the datagram is built and the
        accumulators are reset."""
        ngps = self.cfg.ngps
        cp = batch.code_phase                        # [K][satellites in actSatSet order]
        good = cp >= 0
        stream_nos = [t // ngps for t in batch.smp_times]       # (np.int64, as SMP_TIME//NGPS is there)
        n_good = good.sum(axis=0)
        first = np.where(n_good > 0, good.argmax(axis=0), len(stream_nos))
        cols = cp.T.tolist()
        for col in np.argsort(first, kind='stable').tolist():
            if n_good[col] == 0:
                break
            lst = self.co_ph_lst.setdefault(batch.sats[col], [])
            if n_good[col] == len(stream_nos):
                lst.extend(zip(stream_nos, cols[col]))
            else:
                lst.extend((stream_nos[k], cols[col][k]) for k in np.flatnonzero(good[:, col]).tolist())
        frame_lst = []
        for sw_fq, sat_no, f_lst, co_ph, cp_q in batch.res:
            frame_lst += f_lst
            self.cp_q_lst[sat_no] = cp_q
        if not frame_lst:
            return None
        res = pickle.dumps((self.skipped_data, frame_lst, self.co_ph_lst))
        self.result_list.append(res)
        self.co_ph_lst = {}
        self.skipped_data = 0
        return res

    def close(self):
        self.drain()
        R.closeMultiProcPool(self.pool)
        self.acq.engine.close()


def send_udp(sock, res, ip=UDP_IP, port=UDP_PORT):
    """gpsrecv.py:513-517"""
    sock.sendto(res, (ip, port))


def save_results(path, result_list):
    """saveResults (gpsrecv.py:205-212): the list of datagrams as one pickle,
    the file LOAD_PICKLE replays in gpseval."""
    with open(path, 'wb') as f:
        pickle.dump(result_list, f)


def make_udp_socket():
    return socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
