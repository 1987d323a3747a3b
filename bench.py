#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s ingested, 32-SV acquisition + 12-channel tracking.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--blocks NB]

Workload (BASELINE.json configs[2] with configs[1] in front of it): a resident
batch of NB 32-ms blocks of synthetic 2.048 Msps IQ (NB x 65536 complex64 in
HBM, 512 MiB at the default NB = 1024), 12 tracking channels.  One step =
  (a) one cold-acquisition search, 32 SV x 41 Doppler bins x 1 ms, on the first
      millisecond of the batch (configs[1]), and
  (b) 12-channel tracking of all NB blocks in replay mode: every block runs the
      complete SatStream.process arithmetic (carrier wipe-off, FFT code
      correlation + peak fit, prompt integrate-and-dump, amplitude statistics,
      PLL) from the state recorded at its start, and the records are copied to
      the host.  The state table is produced by the closed loop in the untimed
      set-up, and after the timed steps the bench verifies that replay returned
      exactly the closed loop's outputs -- nothing is skipped, only the order of
      evaluation differs (DESIGN.md "Closed loop and replay").
value = NB * 65536 * n_gpus / step time.  The closed loop itself (state fed back
block by block, launch-latency bound) is timed once and reported alongside.

N > 1 (launched by torch.distributed.run, one rank per GPU): the stream is
sharded in time -- each rank tracks its own NB blocks -- and the acquisition
search is sharded by SV with one RCCL all-gather of the peak records; no other
data-path collective exists.  torch is used only for the rendezvous, the
barrier and the max-over-ranks of the step time (gloo); it is not imported at
N = 1.

The printed JSON line carries `roofline` for the dominant kernel (the
correlator, trk_stream_mfma_kernel: algorithmic bytes = NB*65536*8 per launch, time
from HIP events on the engine's stream) and `cpu_baseline` (the numpy oracle
fanned over host cores like the reference's one-process-per-SV pool, on a
bounded sample of the same blocks; rank 0, N = 1 only).
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, 'gps-sdr-receiver_amd'),
                os.path.join(ROOT, 'oracle')]

NGPS = 65536
N_ACQ_BLOCKS = 5
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SEED = 7
TIMED_EVERY = 4                # steps between two kernel-timed steps
N_CH = 12


def pmc_traffic():
    """HBM bytes per correlator launch from the committed rocprofv3 --pmc run of
    tools/kernel_bench.py (same kernel, same batch; separate counter passes):
    FETCH_SIZE counts KiB and on gfx950 reports half of a wide coalesced read
    stream (MI355X_MICROARCH.md, HBM), WRITE_SIZE is exact.  None if absent."""
    path = os.path.join(ROOT, 'profiles', 'round1', 'replay_pmc_counters.txt')
    try:
        fetch = write = None
        take = False
        for line in open(path):
            if not line.startswith(' '):
                take = 'trk_stream_mfma_kernel' in line
            elif take and 'FETCH_SIZE' in line:
                fetch = float(line.split()[1])
            elif take and 'WRITE_SIZE' in line:
                write = float(line.split()[1])
        if fetch is None:
            return None
        return int(fetch * 1024 * 2 + (write or 0) * 1024)
    except OSError:
        return None


# ---------------------------------------------------------------- input data
def _gen_block(b):
    from gpsmi import synth
    return synth.default_scene(N_CH, seed=SEED).block_raw(b)


def generate_raw(first, count, workers):
    """uint16 raw IQ [count, NGPS] of scene blocks first..first+count-1."""
    with mp.get_context('fork').Pool(workers) as pool:
        blocks = pool.map(_gen_block, range(first, first + count), chunksize=4)
    return np.stack(blocks)


# -------------------------------------------------------------- CPU baseline
_CPU_RAW = None      # inherited by the forked workers instead of being pickled


def _cpu_track(args):
    sv, f0, d0 = args
    raw = _CPU_RAW
    import gps_oracle as orc
    from gpsmi import synth
    p = orc.Params()
    ss = orc.SatStream(int(sv), float(f0), p, delay=int(d0))
    blocks = [synth.raw_to_c64(r) for r in raw]
    t0 = time.perf_counter()
    for i, blk in enumerate(blocks):
        ss.process(blk, np.int64((N_ACQ_BLOCKS + i + 1) * NGPS))
    return time.perf_counter() - t0


def cpu_baseline(raw, n_blocks, cores):
    """The oracle (numpy restatement of the reference) on host cores: cold
    acquisition through the first-hit loop on one core, then 12 channels over
    n_blocks blocks, one process per channel (gpsrecv.satCalc's layout)."""
    import gps_oracle as orc
    from gpsmi import synth
    p = orc.Params()
    t = orc.sec_time(p)
    spectra = {s: orc.fft_cacode(s) for s in range(1, 33)}
    blocks = [synth.raw_to_c64(raw[b]) for b in range(N_ACQ_BLOCKS)]
    t0 = time.perf_counter()
    sat_lst, found, freq = list(range(2, 33)), [], p.min_freq
    for b in range(N_ACQ_BLOCKS):
        _, freq, found = orc.sweep_all_sats(blocks[b], freq, sat_lst, found,
                                            p.it_sweep_all, p, spectra, t)
    t_sweep = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.acq_table(blocks[0], [-5000.0 + 250.0 * i for i in range(41)],
                  list(range(1, 33)), 1, p, t=t, spectra=spectra)
    t_cfg2 = time.perf_counter() - t0
    chans = [(s, f, d) for _, s, f, d in found][:N_CH]
    global _CPU_RAW
    _CPU_RAW = raw[N_ACQ_BLOCKS:N_ACQ_BLOCKS + n_blocks]
    procs = min(len(chans), cores)
    t0 = time.perf_counter()
    with mp.get_context('fork').Pool(procs) as pool:
        per = pool.map(_cpu_track, [(s, f, d) for s, f, d in chans], chunksize=1)
    wall = time.perf_counter() - t0
    msps = n_blocks * NGPS / wall / 1e6
    return {
        'value': round(msps, 3), 'unit': 'Msamples/s', 'cores': procs,
        'kind': 'port',
        'sample': (f'{len(chans)}-channel tracking of {n_blocks} blocks '
                   f'({n_blocks * 32} ms of IQ), one process per channel on '
                   f'{procs} cores, wall {wall:.2f} s, sum of per-channel loop '
                   f'time {sum(per):.2f} s; acquisition on 1 core: first-hit '
                   f'sweep 31 SV x 50 bins x 4 ms {t_sweep * 1e3:.0f} ms, '
                   f'32 SV x 41 bins x 1 ms surface {t_cfg2 * 1e3:.0f} ms'),
        'acq_cfg2_ms': round(t_cfg2 * 1e3, 2),
        'acq_sweep_ms': round(t_sweep * 1e3, 2),
    }, found


# ---------------------------------------------------------------------- main
def main():
    # stdout carries exactly one JSON line: whatever libraries print there while we run
    # (gloo's connection banner, for one) is sent to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--blocks', type=int, default=1024)
    ap.add_argument('--cpu-blocks', type=int, default=192)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--rehearse', action='store_true',
                    help='N > 1 on a box with fewer GPUs: ranks share devices, RCCL skipped')
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit('bench.py --gpus N > 1 must be launched with '
                     'python -m torch.distributed.run --nproc-per-node N')
        a.gpus = world
    dist = None
    if world > 1:
        import torch  # noqa: F401  (plumbing only; imported before libgpsmi)
        import torch.distributed as dist
        dist.init_process_group('gloo')

    nb = a.blocks
    cores = os.cpu_count() or 1
    workers = max(1, min(16, cores // max(1, min(world, 8))))

    # ---- inputs (host, before HIP is initialised: fork pools)
    first = rank * (nb + N_ACQ_BLOCKS)           # time-sharded stream
    t0 = time.perf_counter()
    raw = generate_raw(first, nb + N_ACQ_BLOCKS, workers)
    t_gen = time.perf_counter() - t0

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu:
        cpu, _ = cpu_baseline(raw, min(a.cpu_blocks, nb), cores)

    # ---- GPU set-up
    from gpsmi import engine as E
    from gpsmi.acquisition import Acquisition
    if a.rehearse:
        local = local % E.device_count()
    cfg = E.Config(device=local)
    dev_name = E.device_name(local)
    acq = Acquisition(cfg)
    trk = E.TrkEngine(cfg, max_ch=N_CH)
    d_raw = E.DeviceBuffer(raw.nbytes, local)
    d_raw.upload(raw)
    d_iq = E.DeviceBuffer(raw.size * 8, local)
    E.unpack_u8iq(d_iq.ptr, d_raw.ptr, raw.size, local)   # streamData's decode
    d_raw.free()
    blk_bytes = NGPS * 8

    # cold acquisition with the reference's own first-hit loop -> channels
    sat_lst, found, freq = list(range(2, 33)), [], cfg.min_freq
    for b in range(N_ACQ_BLOCKS):
        _, freq, found = acq.sweepAllSats((d_iq.at(b * blk_bytes), NGPS), freq,
                                          sat_lst, found, cfg.it_sweep_all)
    chans = [(s, f, d) for _, s, f, d in found][:N_CH]
    for c, (s, f, d) in enumerate(chans):
        trk.open(c, s, f, d)

    # closed loop over the batch: the trajectory (state at block start) + outputs
    states = np.zeros((nb, N_CH), dtype=E.STATE_DTYPE)
    cl_out = np.zeros((nb, N_CH), dtype=E.OUT_DTYPE)
    trk_base = N_ACQ_BLOCKS * blk_bytes
    E.sync(local)
    t0 = time.perf_counter()
    cl_dev_ms = 0.0
    for i in range(nb):
        for c in range(N_CH):
            states[i, c] = trk.get_state(c)
        cl_out[i] = trk.process(d_iq.at(trk_base + i * blk_bytes))
        cl_dev_ms += trk.last_ms()[0]
    t_closed_with_readback = time.perf_counter() - t0
    # closed loop again with no per-block host traffic: state stays on the device
    for c, (s, f, d) in enumerate(chans):
        trk.open(c, s, f, d)
    trk.set_timing(False)                   # no kernel-timing events in the queue either
    E.sync(local)
    t0 = time.perf_counter()
    for i in range(nb):
        trk.process(d_iq.at(trk_base + i * blk_bytes), want_out=False)
    E.sync(local)
    t_closed = time.perf_counter() - t0
    trk.set_timing(True)

    # SV shard of the acquisition search + RCCL gather
    prn_all = list(range(1, 33))
    shard = prn_all[rank * 32 // world:(rank + 1) * 32 // world]
    f41 = [-5000.0 + 250.0 * i for i in range(41)]
    comm = d_send = d_recv = None
    use_rccl = world > 1 and not a.rehearse
    if world > 1:
        import ctypes as C
        from gpsmi import _lib
        lib = _lib.load()
        idb = np.zeros(_lib.COMM_ID_BYTES, np.uint8)
        if rank == 0:
            E.check(lib.gpsmi_comm_unique_id(E.ptr(idb)), 'comm_unique_id')
        import torch
        tid = torch.from_numpy(idb)
        dist.broadcast(tid, 0)
        collective = 'rehearsal: none'
        if use_rccl:
            comm = C.c_void_p()
            rc = lib.gpsmi_comm_create(E.ptr(idb), world, rank, local, C.byref(comm))
            ok = torch.tensor([1 if rc == 0 else 0])
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)        # all ranks agree on the path
            if int(ok[0]) == 1:
                collective = 'rccl all-gather over xGMI'
            else:
                err = lib.gpsmi_last_error().decode(errors='replace')
                collective = f'gloo all-gather (RCCL init failed: {err})'
                if rc == 0:
                    lib.gpsmi_comm_destroy(comm)
                comm, use_rccl = None, False
        cells = len(f41) * len(shard)
        d_send = E.DeviceBuffer(cells * 16, local)
        d_recv = E.DeviceBuffer(cells * 16 * world, local)
        gathered = np.zeros(cells * world, dtype=E.PEAK_DTYPE)

    trk.replay_load(nb, states, cl_out['delay_used'])
    # two page-locked result buffers: the read-back of step k overlaps the kernels of step k+1
    pins = [E.PinnedArray((nb, N_CH), E.OUT_DTYPE) for _ in range(2)]

    def barrier():
        if dist is not None:
            dist.barrier()
        E.sync(local)

    corr_ms, total_ms, acq_ms = [], [], []
    recording = [False]

    acq_pin = E.PinnedArray((len(f41), len(shard)), E.PEAK_DTYPE)

    def record_last():
        t, c = trk.last_ms()
        total_ms.append(t)
        corr_ms.append(c)

    def finish_search():
        """wait for the search enqueued one step ago; gather its peak records"""
        acq.engine.wait()
        if recording[0]:
            acq_ms.append(acq.engine.last_ms())
        if use_rccl:
            E.check(lib.gpsmi_comm_allgather_peaks(
                comm, d_send.ptr, d_recv.ptr, len(f41) * len(shard),
                E.ptr(gathered)), 'allgather')
        elif world > 1 and not a.rehearse:                 # labelled fallback, see `collective`
            mine = torch.from_numpy(acq_pin.array.view(np.uint8).reshape(-1).copy())
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)

    def step(k, record):
        # Software pipeline of depth two: the host enqueues step k (search on its own
        # handle and stream, tracking batch, read-back on the copy stream) and only
        # then waits for step k-1, so that neither the read-back nor the host's launch
        # work leaves the GPU idle between steps.
        if k > 0:
            finish_search()
        # device-side order: the search starts when batch k-1 has finished, i.e. it runs
        # beside the code-phase correlation of batch k and is over before the correlator
        # (the kernel the roofline is quoted for) starts
        acq.engine.after(trk)
        acq.engine.search_async(d_iq.ptr, NGPS, shard, f41, 1, acq_pin.array,
                                d_send.ptr if world > 1 else None)
        # the kernel-timing events are barrier packets in the queue (~5 us each): the
        # kernels of every fourth step are timed, the others run without them
        trk.set_timing(k % TIMED_EVERY == 0)
        trk.replay_run_async(d_iq.at(trk_base), nb)
        trk.replay_fetch_async(pins[k & 1].array)
        trk.wait_prev()
        if record and k > 0 and (k - 1) % TIMED_EVERY == 0:
            record_last()

    for k in range(a.warmup):
        step(k, False)
    if a.warmup:
        finish_search()
    trk.wait()
    barrier()
    recording[0] = True
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k, True)
    finish_search()                         # the last step: its search, ...
    trk.wait()                              # ... its kernels and its copy
    barrier()
    dt = time.perf_counter() - t0
    if (a.steps - 1) % TIMED_EVERY == 0:
        record_last()
    trk.set_timing(True)
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])

    # replay must have returned exactly what the closed loop produced
    same = all(p.array.tobytes() == cl_out.tobytes() for p in pins[:min(2, a.steps)])
    nxt = trk.replay_states(nb)
    chained = all(np.array_equal(nxt[:-1][k], states[1:][k])
                  for k in ('delay', 'freq', 'phase', 'phase_locked', 'nps',
                            'prev_sum_re', 'prev_sum_im', 'df_len'))
    locked = int(cl_out[-1]['phase_locked'].sum())

    if rank == 0:
        samples = nb * NGPS
        ms_step = dt / a.steps * 1e3
        value = samples * world / (dt / a.steps) / 1e6
        k_ms = float(np.mean(corr_ms))
        alg_bytes = samples * 8
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        line = {
            'metric': 'IQ Msamples/s ingested, 32-SV acq + 12-ch track',
            'value': round(value, 1), 'unit': 'Msamples/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(ms_step, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {
                'workload': ('BASELINE configs[2]: 12-channel tracking of '
                             f'{nb} x 32 ms blocks (65536 complex64 each, '
                             f'{samples * 8 / 2**20:.0f} MiB resident in HBM) '
                             'in replay of the closed-loop trajectory, results '
                             'copied to host (the copy of a step overlaps the '
                             'kernels of the next); preceded per step by configs[1]: '
                             '32 SV x 41 Doppler x 1 ms acquisition search'),
                'channels': len(chans), 'blocks': nb, 'sample_rate_hz': 2048000,
                'sharding': ('1 GPU' if world == 1 else
                             f'{world} ranks: stream sharded in time, '
                             f'acquisition sharded by SV, peaks by {collective}'),
            },
            'x_realtime': round(value / 2.048, 1),
            'roofline': {
                'bound': 'hbm', 'kernel': 'trk_span_kernel',
                'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                'traffic': pmc_traffic() if nb == 1024 else None,
                'kernel_ms': round(k_ms, 4),
                'algorithmic_bytes_per_launch': alg_bytes,
            },
            'kernels_ms': {
                'tracking_all': round(float(np.mean(total_ms)), 4),
                'correlator': round(k_ms, 4),
                'acquisition_search': round(float(np.mean(acq_ms)), 4),
            },
            'closed_loop': {
                'value': round(samples / t_closed / 1e6, 1),
                'unit': 'Msamples/s',
                'x_realtime': round(samples / t_closed / 2.048e6, 1),
                'us_per_block': round(t_closed / nb * 1e6, 2),
                'with_per_block_readback_msps':
                    round(samples / t_closed_with_readback / 1e6, 1),
                'device_ms_per_block': round(cl_dev_ms / nb, 4),
            },
            'checks': {'replay_equals_closed_loop': bool(same),
                       'replay_states_chain': bool(chained),
                       'channels_locked_at_end': locked,
                       'acquired': len(found)},
            'device': dev_name,
            'setup_s': {'generate_iq': round(t_gen, 2)},
        }
        if cpu is not None:
            line['cpu_baseline'] = cpu
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + '\n').encode())
        if not (same and chained):
            sys.exit('bench: replay does not reproduce the closed loop')
    for p_ in pins:
        p_.free()
    acq_pin.free()
    if comm is not None:
        lib.gpsmi_comm_destroy(comm)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
