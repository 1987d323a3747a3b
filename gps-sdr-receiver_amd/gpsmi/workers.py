"""The reference's worker message protocol served by one GPU loop.

``gpsrecv.runProc(inQ, outQ)`` (reference src/gpsrecv.py:300-337) is the body of one
OS process per satellite; the parent talks to it with five messages:

    ('initPool', no)                  -> (name, no)
    ('initInst', (satNo, freq, delay)) -> satNo          constructs the SatStream
    ('delInst', None)                 -> bool
    ('runInst', (data, smpTime))      -> (swFq, satNo, frameData, coPh, cpQ)
    ('done', None)                    -> the loop ends

Here ONE loop serves the queue pairs of all workers: every worker is a channel of one
tracking engine, and the ``runInst`` messages of a block -- the parent's ``satCalc``
(gpsrecv.py:404-417) puts one into every active worker's queue before it reads any
answer -- are collected and run as ONE ``gpsmi_trk_process`` call; each worker's answer
goes to its own ``outQ`` in the reference's tuple.  The parent-side functions of the
reference (``initMultiProcPool`` ... ``satCalc``, ``closeMultiProcPool`` with its
put('done') / join() / close() per worker in turn) work unchanged on the pool this module
returns, because the traffic on the queues is the same and every worker has a handle of its
own whose join() returns when that worker's 'done' has been consumed; they are restated
below (``q_*``) for a host that does not import the reference.

Message timing never changes results (the reference's workers are independent processes):
a burst that covers only some of the instances -- the grace period ran out, or another
message closed it -- runs those channels only, the state of the others is put back
(``receiver._sat_calc``), and their late ``runInst`` for the same block runs them then.
An exception inside the loop is sent to every worker's ``outQ`` (a parent blocked in
``outQ.get()`` gets the exception object instead of hanging) before the loop ends.

``gpsmi.receiver`` has the same functions as direct calls (no queues, no thread); this
module is for a parent that wants to keep the reference's message boundary.
"""
import queue
import threading
import time

from . import receiver as R
from .engine import Config


class WorkerLoop:
    """runProc for all workers at once.  pairs: [(inQ, outQ), ...], one per worker slot."""

    def __init__(self, pairs, cfg=None, raw_u8=False, pool=None, grace=0.05):
        self.pairs = list(pairs)
        self.cfg = cfg or Config()
        self.pool = pool if pool is not None else R.GpuPool(len(self.pairs), self.cfg, raw_u8)
        self.grace = grace                     # s to wait for the rest of a block's runInst burst
        self.worker_no = [None] * len(self.pairs)
        self.sat = [0] * len(self.pairs)       # PRN per worker slot, 0 = no instance
        self.done = [threading.Event() for _ in self.pairs]   # worker w has consumed its 'done'
        self._inbox = queue.Queue()

    def _forward(self, wno, in_q):
        while True:
            msg = in_q.get()
            self._inbox.put((wno, msg))
            if msg[0] == 'done':
                return

    def _flush(self, pending):
        """One engine call for every worker that received runInst for this block."""
        by_time = {}
        for wno, (data, smp_time) in pending.items():
            by_time.setdefault(int(smp_time), []).append(wno)
        for smp_time in sorted(by_time):       # (one group unless the parent mixes blocks)
            wnos = by_time[smp_time]
            data = pending[wnos[0]][0]
            sats = [self.sat[w] for w in wnos]
            res = R.satCalc(sats, self.pool, self.sat, data, pending[wnos[0]][1])
            for r in res:                      # (swFq, satNo, frameData, coPh, cpQ)
                self.pairs[self.sat.index(r[1])][1].put(r)
        pending.clear()

    def run(self):
        fwd = [threading.Thread(target=self._forward, args=(w, q[0]), daemon=True)
               for w, q in enumerate(self.pairs)]
        for t in fwd:
            t.start()
        alive = len(self.pairs)
        pending, first_at = {}, None
        try:
            while alive:
                timeout = None
                if pending:
                    timeout = max(0.0, first_at + self.grace - time.monotonic())
                try:
                    wno, msg = self._inbox.get(timeout=timeout)
                except queue.Empty:
                    self._flush(pending)       # the burst did not cover every instance
                    continue
                kind, out_q = msg[0], self.pairs[wno][1]
                if kind == 'runInst':
                    if not pending:
                        first_at = time.monotonic()
                    pending[wno] = msg[1]
                    if all(w in pending for w, s in enumerate(self.sat) if s):
                        self._flush(pending)
                    continue
                if pending:                    # any other message closes the burst first
                    self._flush(pending)
                if kind == 'initPool':
                    self.worker_no[wno] = msg[1]
                    out_q.put((threading.current_thread().name, msg[1]))
                elif kind == 'initInst':
                    sat_no, freq, delay = msg[1]
                    self.sat[wno] = R.open_worker(self.pool, wno, sat_no, freq, delay)
                    out_q.put(sat_no)
                elif kind == 'delInst':
                    done = R.close_worker(self.pool, wno)
                    self.sat[wno] = 0
                    out_q.put(done)
                elif kind == 'done':
                    alive -= 1
                    self.done[wno].set()
        except BaseException as err:           # a parent blocked in outQ.get() must not hang
            for _, out_q in self.pairs:
                out_q.put(err)
            raise
        finally:
            for ev in self.done:
                ev.set()
            self.pool.close()


class _Handle:
    """What the reference's closeMultiProcPool (gpsrecv.py:363-367) expects of a worker
    process: join() returns once this worker's 'done' has been consumed, then close()."""

    def __init__(self, thread, done):
        self.thread, self.done = thread, done

    def join(self, timeout=None):
        self.done.wait(timeout)

    def close(self):
        pass


def q_initMultiProcPool(poolNo, cfg=None, raw_u8=False, pool=None, make_queue=queue.Queue):
    """gpsrecv.py:340-360 -> (pool, poolNo, poolWorker) with pool = [(inQ, outQ, handle)]:
    the queues of poolNo workers, all served by one WorkerLoop thread."""
    pairs = [(make_queue(), make_queue()) for _ in range(poolNo)]
    loop = WorkerLoop(pairs, cfg, raw_u8, pool)
    th = threading.Thread(target=loop.run, name='gpsmi-workers', daemon=True)
    th.start()
    qpool = [(i, o, _Handle(th, loop.done[w])) for w, (i, o) in enumerate(pairs)]
    for wno, (in_q, _, _) in enumerate(qpool):
        in_q.put(('initPool', wno))
    for _, out_q, _ in qpool:
        out_q.get()
    return qpool, poolNo, [0] * poolNo


def q_closeMultiProcPool(pool):             # gpsrecv.py:363-367, worker by worker as there
    for in_q, _, handle in pool:
        in_q.put(('done', None))
        handle.join()
        handle.close()
    pool[0][2].thread.join()                # (the one loop behind all of them has ended)


def q_delPoolStreams(pool, poolNo, poolWorker, actSatSet, delSatSet):
    """gpsrecv.py:370-382"""
    for sat_no in delSatSet:
        wno = poolWorker.index(sat_no)
        in_q, out_q, _ = pool[wno]
        in_q.put(('delInst', None))
        if out_q.get():
            poolWorker[wno] = 0
    return poolWorker, actSatSet - delSatSet


def q_initPoolStreams(pool, poolNo, poolWorker, actSatSet, newSatSet, foundSats):
    """gpsrecv.py:385-401"""
    if len(newSatSet) > 0:
        for wno, sno in enumerate(poolWorker):
            if sno == 0:
                new_sat = newSatSet.pop()
                poolWorker[wno] = new_sat
                in_q, out_q, _ = pool[wno]
                _, _, freq, delay = [e for e in foundSats if e[1] == new_sat][0]
                in_q.put(('initInst', (new_sat, freq, delay)))
                actSatSet.add(out_q.get())
                if len(newSatSet) == 0:
                    break
    return poolWorker, actSatSet


def q_satCalc(actSatSet, pool, poolWorker, data, smpTime):
    """gpsrecv.py:404-417: runInst to every active worker, then the answers in that order."""
    out_qs = []
    for sno in actSatSet:
        in_q, out_q, _ = pool[poolWorker.index(sno)]
        in_q.put(('runInst', (data, smpTime)))
        out_qs.append(out_q)
    return [q.get() for q in out_qs]
