#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s ingested, 32-SV acquisition + 12-channel tracking.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--blocks NB]
                    [--shard time|channels] [--no-cpu] [--no-extra] [--settle-steps S]

Workload (BASELINE.json configs[2] with an acquisition search in front of it): a
resident batch of NB 32-ms blocks of synthetic 2.048 Msps IQ (NB x 65536
complex64 in HBM, 512 MiB at the default NB = 1024), 12 tracking channels.  One
step =
  (a) one cold-acquisition search on the first milliseconds of the batch:
      N = 1: configs[1], 32 SV x 41 Doppler bins x 1 ms;
      N > 1: configs[3], 32 SV x 201 bins x 10 ms with the SVs sharded over the
             ranks (4 per GPU at N = 8) and one RCCL all-gather of the peak records;
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
Before the W warm-up steps the device is driven with S = 300 (--settle-steps) of the
same steps, untimed: an idle MI355X takes some 40 ms under load to reach steady clocks
(the same correlator launch: ~120 us cold, 103 us from the 100th step on and flat for
the 400 ms measured, profiles/round2/clock_settling.txt), and K = 20 steps timed from
cold measure that ramp rather than the kernels.  S is printed in the JSON line;
--settle-steps 0 gives the cold figure.

N > 1 (launched by torch.distributed.run, one rank per GPU).  Two splits of the
tracking work (gpsmi/sharding.py):
  --shard time      (default) every rank tracks its own NB blocks of the stream:
                    per-rank work is fixed, "scaling": "weak";
  --shard channels  north_star's split, one worker per SV as in the reference:
                    the 12 channels round-robin over the ranks, every rank reads
                    the same NB blocks: a fixed job is cut up, "scaling": "strong".
No data-path collective exists besides the gather of peak records.  torch is used
only for the rendezvous, the barrier and the max-over-ranks of the step time
(gloo); it is not imported at N = 1.

The printed JSON line carries `roofline` for the dominant kernel (the
correlator, trk_span_kernel: algorithmic bytes = NB*65536*8 per launch, duration
from the dispatch's own begin / end stamps -- hipExtLaunchKernel events on the
engine's stream, what a kernel trace reports), `cpu_baseline` (the numpy oracle
fanned over host cores like the reference's one-process-per-SV pool, on a bounded
sample of the same blocks; rank 0, N = 1 only) and `configs`: BASELINE configs[3]
(fine acquisition), the fused-ingest replay on raw uint16 IQ, configs[4] (16.368 Msps
tracking) and the closed loop streamed from host memory (H2D inside the timed loop, raw
uint16 next to complex64), all timed on this GPU after the timed region.  Round 3 adds
`roofline.frac_cold` (the correlator over the first steps from an idle device),
`issue_rooflines` (the FFT-bound kernels against what their SIMDs can issue; counters
from the committed profiles/round3/pmc_counters.txt), `closed_loop.batched` (R independent
receivers on one handle) and, at N > 1, `multi_gpu` (RCCL's own rank count, per-rank
rates, the gather alone, the channel-sharded split beside the time-sharded headline).
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
HBM_COPY_GBS = 6290.0          # same guide: measured float4 copy
SEED = 7
TIMED_EVERY = 4                # steps between two kernel-timed steps
N_CH = 12
PMC_FILE = os.path.join('profiles', 'round4', 'pmc_counters.txt')
ISSUE_CYCLES = 4.9             # measured cycles per packed-fp32 VALU instruction of one wave's stream
                               # (tools/probe/hazard_probe.hip); the FFT kernels are made of those
ISSUE_CLOCK_GHZ = 2.2          # the clock DESIGN section 4.4 priced its 78 us floor at


def csrc_sha256():
    """sha256 over the kernel sources (csrc/*.h, *.hip in name order): what ties a committed
    counter summary to the code it was taken from (tools/pmc_summary.py --stamp writes it)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'gps-sdr-receiver_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(d, '*.h')) + glob.glob(os.path.join(d, '*.hip'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()


def pmc_matches_sources():
    """True when the committed counter summary names the present kernel sources."""
    try:
        first = open(os.path.join(ROOT, PMC_FILE)).readline()
    except OSError:
        return False
    return first.startswith('# csrc_sha256 ') and first.split()[2] == csrc_sha256()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc pass of
    tools/kernel_bench.py (same kernel, same batch; separate counter passes): FETCH_SIZE
    counts KiB and on gfx950 reports half of a wide coalesced read stream
    (MI355X_MICROARCH.md, HBM), WRITE_SIZE is exact.  (None, reason) when there is no
    summary or it was taken from other kernel sources than the present ones (its first line
    carries their sha256): a counter from another build says nothing about this one.  Never a
    measurement of the present run, and labelled as such."""
    path = os.path.join(ROOT, PMC_FILE)
    if not pmc_matches_sources():
        return None, 'no counter summary of these kernel sources at ' + PMC_FILE
    try:
        fetch = write = None
        take = False
        for line in open(path):
            if not line.startswith(' '):
                take = kernel in line
            elif take and 'FETCH_SIZE' in line:
                fetch = float(line.split()[1])
            elif take and 'WRITE_SIZE' in line:
                write = float(line.split()[1])
        if fetch is None:
            return None, None
        return int(fetch * 1024 * 2 + (write or 0) * 1024), PMC_FILE
    except OSError:
        return None, None


def pmc_counter(kernel, counter):
    """Mean of `counter` per launch of the first kernel whose name contains `kernel` in the
    committed PMC summary (PMC_FILE; None if absent or taken from other kernel sources).  Not a
    measurement of the present run."""
    if not pmc_matches_sources():
        return None
    try:
        take = False
        for line in open(os.path.join(ROOT, PMC_FILE)):
            if not line.startswith(' '):
                take = kernel in line
            elif take and line.split()[0] == counter:
                return float(line.split()[1])
    except OSError:
        pass
    return None


def issue_roofline(kernels, duration_ms, label):
    """For kernels bound by what their SIMDs can issue (the LDS-resident FFT kernels): the time
    their VALU instructions need at the measured issue cost on all 1024 SIMDs, over the duration.
    kernels: [(name fragment in the PMC summary, launches per measured duration)]."""
    n = 0.0
    for k, times in kernels:
        v = pmc_counter(k, 'SQ_INSTS_VALU')
        n = None if (v is None or n is None) else n + v * times
    if n is None or not duration_ms:
        return {'kernel': label, 'bound': 'valu issue', 'frac': None, 'note': 'no PMC summary at ' + PMC_FILE}
    floor_ms = n / 1024.0 * ISSUE_CYCLES / (ISSUE_CLOCK_GHZ * 1e6)
    return {'kernel': label, 'bound': 'valu issue', 'valu_instructions_per_launch': int(n),
            'issue_cycles_per_instruction': ISSUE_CYCLES, 'clock_ghz': ISSUE_CLOCK_GHZ,
            'floor_ms': round(floor_ms, 4), 'duration_ms': round(duration_ms, 4),
            'frac': round(floor_ms / duration_ms, 4), 'counter_source': PMC_FILE}


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


# ------------------------------------------------- the other BASELINE configs
def measure_cfg4(E, acq, d_iq, reps=10):
    """configs[3] on this GPU alone: 32 SV x 201 bins x 10 ms, device time per search."""
    f201 = [-5000.0 + 50.0 * i for i in range(201)]
    prns = list(range(1, 33))
    ms = []
    for i in range(reps + 2):
        acq.engine.search((d_iq, 10 * 2048), prns, f201, 10)
        if i >= 2:
            ms.append(acq.engine.last_ms())
    t = float(np.median(ms))
    return {'config': 'BASELINE configs[3]: 32 SV x 201 Doppler bins x 10 ms coherent, one GPU '
                      '(all 32 SVs)',
            'us_per_search': round(t * 1e3, 1), 'searches_per_s': round(1e3 / t, 1),
            'cells': 32 * 201, 'msamples_per_s': round(20480 / t / 1e3, 2),
            'bound': 'LDS / VALU (FFT); algorithmic HBM bytes 0.75 MB per search',
            'issue_roofline': issue_roofline([('acq acq_spectrum_kernel<4', 1), ('acq acq_corr_kernel', 1)], t,
                                             'configs[3] search: acq_spectrum_kernel<4> + acq_corr_kernel')}


def measure_u8(E, local, d_raw, nb, chans, states, delay_used, expect, iters=8):
    """the same replay with the raw uint16 recording as input (2 bytes per
    sample from HBM, decoded inside the kernels that read IQ) -- its own roofline regime."""
    trk = E.TrkEngine(E.Config(device=local), max_ch=max(1, len(chans)))
    trk.set_input_format(True)
    for c, (s, f, d) in enumerate(chans):
        trk.open(c, s, f, d)
    trk.replay_load(nb, states, delay_used)
    base = d_raw.at(N_ACQ_BLOCKS * NGPS * 2)
    tot, cor = [], []
    for i in range(iters + 2):
        trk.replay_run(base, nb)
        if i >= 2:
            t, c = trk.last_ms()
            tot.append(t)
            cor.append(c)
    out = np.zeros_like(expect)
    trk.replay_fetch(out)
    trk.close()
    c, t = float(np.median(cor)), float(np.median(tot))
    gb = nb * NGPS * 2 / 1e9
    return {'config': f'fused ingest: the configs[2] replay on raw uint16 (Q<<8|I) IQ, {nb} blocks = '
                      f'{nb * NGPS * 2 / 2**20:.0f} MiB resident, 2 B per sample',
            'correlator_ms': round(c, 4), 'correlator_gbs': round(gb / c * 1e3, 1),
            'correlator_frac_of_hbm_peak': round(gb / c * 1e3 / HBM_PEAK_GBS, 4),
            'bound': 'MFMA / VALU issue (the decode adds 6 VALU operations per sample), not HBM',
            'tracking_all_ms': round(t, 4), 'msamples_per_s': round(nb * NGPS / t / 1e3, 1),
            'equals_complex64_path': bool(out.tobytes() == expect.tobytes())}


def measure_cfg5(E, local, iters=8):
    """configs[4]: 12-channel tracking at 16.368 Msps, N_CYC = 8 (block = 130944 samples),
    512 blocks = 512 MiB resident, replay from a synthetic state table (random IQ: a
    timing run; parity of this configuration is tests/test_gpu_trk.py on ref_hirate.npz)."""
    cs, n_cyc, nb = 16368, 8, 512
    ngps = cs * n_cyc
    rng = np.random.default_rng(1)
    trk = E.TrkEngine(E.Config(code_samples=cs, n_cyc=n_cyc, device=local), max_ch=N_CH)
    buf = E.DeviceBuffer(nb * ngps * 8, local)
    chunk = (rng.standard_normal((16, ngps, 2)) * 0.25).astype(np.float32)
    for i in range(0, nb, 16):
        buf.upload(chunk[:min(16, nb - i)], i * ngps * 8)
    for c in range(N_CH):
        trk.open(c, 2 + c, -4000.0 + 700.0 * c, (1137 * c + 11) % cs)
    st = np.zeros((nb, N_CH), dtype=E.STATE_DTYPE)
    for c in range(N_CH):
        st[:, c] = trk.get_state(c)
    st['phase'] = rng.uniform(0, 6.28, (nb, N_CH)).astype(np.float32)
    dly = np.broadcast_to(st['delay'][0], (nb, N_CH)).copy()
    trk.replay_load(nb, st, dly)
    trk.set_timing(False)
    for i in range(60):                    # ~60 ms under this very load: steady clocks, as the headline's
        trk.replay_run_async(buf.ptr, nb)  # settle steps (a cold device runs the same kernels ~10 % longer)
    trk.wait()
    trk.set_timing(True)
    tot, cor, cph = [], [], []
    for i in range(iters + 2):
        trk.replay_run(buf.ptr, nb)
        if i >= 2:
            t, c = trk.last_ms()
            tot.append(t)
            cor.append(c)
            cph.append(trk.last_codephase_ms())
    trk.close()
    buf.free()
    t, c, cp = float(np.median(tot)), float(np.median(cor)), float(np.median(cph))
    gb = nb * ngps * 8 / 1e9
    return {'config': 'BASELINE configs[4]: 12-channel tracking @ 16.368 Msps, N_CYC = 8, '
                      f'{nb} blocks x {ngps} complex64 = 512 MiB resident, replay, after 60 untimed '
                      'runs (steady clocks)',
            'correlator_ms': round(c, 4), 'correlator_gbs': round(gb / c * 1e3, 1),
            'correlator_frac_of_hbm_peak': round(gb / c * 1e3 / HBM_PEAK_GBS, 4),
            'codephase_ms': round(cp, 4),
            'codephase_note': 'wipe-off + fold of all 8 rows for 12 channels (96 flop per 8-byte sample, as '
                              'much arithmetic as the correlator), then one native-length 16368-point '
                              '(16 x 3 x 11 x 31) LDS correlation with fused statistics per job',
            'issue_roofline': issue_roofline([('hirate trk_fold_general_kernel', 1), ('pfa_corr_kernel<0>', 1)], cp,
                                             'configs[4] code-phase correlation: trk_fold_general_kernel + pfa_corr_kernel'),
            'tracking_all_ms': round(t, 4),
            'msamples_per_s': round(nb * ngps / t / 1e3, 1),
            'x_realtime': round(nb * ngps / t / 1e3 / 16.368, 1)}


def measure_block_length(E, local, n_cyc, iters=8):
    """The reference's other block lengths (gpsglob.py:122-124: N_CYC "currently possible are
    (32,16,8)"; 16 / 8 ms position updates "with powerful computers", README.md:24) at
    CODE_SAMPLES = 2048: 512 MiB of random IQ resident, 12 channels, replay from a synthetic state
    table (a timing run; parity: tests/test_gpu_trk.py::test_other_block_lengths_*).  The correlator
    is the span form with one M tile (gpsmi_trk_span.h, template parameter NC)."""
    cs = 2048
    ngps = cs * n_cyc
    nb = (512 << 20) // (ngps * 8)
    rng = np.random.default_rng(2)
    trk = E.TrkEngine(E.Config(n_cyc=n_cyc, device=local), max_ch=N_CH)
    buf = E.DeviceBuffer(nb * ngps * 8, local)
    chunk = (rng.standard_normal((16, ngps, 2)) * 0.25).astype(np.float32)
    for i in range(0, nb, 16):
        buf.upload(chunk[:min(16, nb - i)], i * ngps * 8)
    for c in range(N_CH):
        trk.open(c, 2 + c, -4000.0 + 700.0 * c, (173 * c + 11) % cs)
    st = np.zeros((nb, N_CH), dtype=E.STATE_DTYPE)
    for c in range(N_CH):
        st[:, c] = trk.get_state(c)
    st['phase'] = rng.uniform(0, 6.28, (nb, N_CH)).astype(np.float32)
    trk.replay_load(nb, st, np.broadcast_to(st['delay'][0], (nb, N_CH)).copy())
    trk.set_timing(False)
    for i in range(100):
        trk.replay_run_async(buf.ptr, nb)
    trk.wait()
    trk.set_timing(True)
    tot, cor = [], []
    for i in range(iters + 2):
        trk.replay_run(buf.ptr, nb)
        if i >= 2:
            t, c = trk.last_ms()
            tot.append(t)
            cor.append(c)
    variant = trk.get_option('correlator')
    trk.close()
    buf.free()
    t, c = float(np.median(tot)), float(np.median(cor))
    gb = nb * ngps * 8 / 1e9
    return {'config': f'N_CYC = {n_cyc} at 2.048 Msps ({n_cyc}-ms blocks): 12-channel tracking, {nb} blocks x '
                      f'{ngps} complex64 = 512 MiB resident, replay, after 100 untimed runs',
            'correlator': 'trk_span_kernel<8, 4, 0, 0, %d> (matrix pipe)' % n_cyc if variant else 'trk_stream_kernel (vector)',
            'correlator_ms': round(c, 4), 'correlator_gbs': round(gb / c * 1e3, 1),
            'correlator_frac_of_hbm_peak': round(gb / c * 1e3 / HBM_PEAK_GBS, 4),
            'tracking_all_ms': round(t, 4), 'msamples_per_s': round(nb * ngps / t / 1e3, 1),
            'x_realtime': round(nb * ngps / t / 1e3 / 2.048, 1)}


def measure_batched(E, local, iq_at, nb_avail, chans, Rs=(1, 8, 32, 64), max_steps=64):
    """the closed loop of R independent receivers on one handle
    (gpsmi_trk_set_streams), state fed back on the device block by block, nothing read back
    and no host wait inside the loop.  Slab s of the resident buffer holds block s of every
    stream (R blocks back to back); iq_at(block) -> device pointer."""
    res = []
    for R in Rs:
        steps = min(max_steps, nb_avail // R)
        if steps < 2:
            continue
        trk = E.TrkEngine(E.Config(device=local), max_ch=max(1, len(chans)), streams=R)
        for r in range(R):
            for c, (s, f, d) in enumerate(chans):
                trk.open(c, s, f, d, stream=r)
        trk.set_timing(False)
        best = None
        for rep in range(3):                       # (the first pass also uploads the state rows)
            E.sync(local)
            t0 = time.perf_counter()
            for i in range(steps):
                trk.process(iq_at(i * R), want_out=False)
            E.sync(local)
            t = (time.perf_counter() - t0) / steps
            best = t if best is None else min(best, t)
        trk.close()
        res.append({'R': R, 'channels': R * len(chans), 'us_per_step': round(best * 1e6, 2),
                    'msamples_per_s': round(R * NGPS / best / 1e6, 1),
                    'x_realtime_per_stream': round(NGPS / best / 2.048e6, 1)})
    return res


def measure_streamed(E, local, raw_blocks, chans, Rs=(1, 8, 32), steps=48):
    """The reference's own shape of the flow (file -> streamData -> ring buffer -> processData,
    gpsrecv.py:153-186, :76-104, :445-548): blocks arrive in (page-locked) host memory and go up
    inside the timed loop, the host running ahead of the device (gpsmi_trk_process_stream),
    closed loop, state on the device, nothing read back.  Raw uint16 (2 B/sample over PCIe, the
    decode fused into the kernels) next to complex64 (8 B/sample), R receivers per step."""
    from gpsmi.synth import raw_to_c64
    res = []
    nb = len(raw_blocks)
    for R in Rs:
        for raw_u8 in (True, False):
            trk = E.TrkEngine(E.Config(device=local), max_ch=max(1, len(chans)), streams=R)
            if raw_u8:
                trk.set_input_format(True)
            for r in range(R):
                for c, (s, f, d) in enumerate(chans):
                    trk.open(c, s, f, d, stream=r)
            dt = np.uint16 if raw_u8 else np.complex64
            ring = [E.PinnedArray((R, NGPS), dt) for _ in range(3)]
            for k, pin in enumerate(ring):                 # (three distinct slabs of the recording)
                for r in range(R):
                    blk = raw_blocks[(k * R + r) % nb]
                    pin.array[r] = blk if raw_u8 else raw_to_c64(blk)
            best = None
            for rep in range(3):
                E.sync(local)
                t0 = time.perf_counter()
                for i in range(steps):
                    trk.process_stream(ring[i % 3].array)
                trk.wait()
                t = (time.perf_counter() - t0) / steps
                best = t if best is None else min(best, t)
            trk.close()
            for pin in ring:
                pin.free()
            bps = 2 if raw_u8 else 8
            res.append({'R': R, 'input': 'raw uint16 (Q<<8|I), 2 B/sample' if raw_u8 else 'complex64, 8 B/sample',
                        'us_per_step': round(best * 1e6, 2),
                        'msamples_per_s': round(R * NGPS / best / 1e6, 1),
                        'pcie_gb_per_s': round(R * NGPS * bps / best / 1e9, 2),
                        'x_realtime_per_stream': round(NGPS / best / 2.048e6, 1)})
    return {'config': 'streamed from host: closed loop with the H2D copy of every block inside the timed '
                      'loop (gpsmi_trk_process_stream, pinned memory, no host wait per block), 12 channels '
                      'per receiver, R receivers per step',
            'runs': res}


def measure_dropin(E, local, raw, n_blocks):
    """The path a user of the reference calls: ``pipeline.Receiver.feed(block)`` per 32-ms block --
    gpsrecv.processData's loop (gpsrecv.py:466-519): cold sweep over the first blocks
    (sweepAllSats, getNewSats, initPoolStreams), then satCalc + hand-off per block, a pickled
    datagram once a second -- fed from HOST memory block by block (complex64 as streamData hands
    them on, and the recorder's raw uint16), 12 channels (MAX_SAT = 12), everything inside the
    timed loop: the host copy into page-locked memory, the upload, the kernels, the record
    read-back, the host bookkeeping and the pickling.  `tracking_only` starts the clock at the
    first tracking block (the cold sweep's five blocks are synchronous searches).
    Each format twice: `report_lag` 0 -- feed() returns a datagram on the very block the reference sends
    it on, the GPU idles through the host's once-a-second work -- and `report_lag` 16 -- the same
    datagrams 16 blocks later, the caller up to 16 blocks ahead of the device ("stream_depth"), which
    works through its queue meanwhile: the mode for a recording (gpsrecv's own hand-off to saveResults /
    UDP is asynchronous too, gpsrecv.py:496-519)."""
    import pickle
    from gpsmi.pipeline import Receiver
    from gpsmi.synth import raw_to_c64
    out = []
    for raw_u8, lag in ((False, 0), (True, 0), (False, 16), (True, 16)):
        blocks = [np.ascontiguousarray(raw[i]) if raw_u8 else raw_to_c64(raw[i]) for i in range(n_blocks)]
        best = None
        for rep in range(2):
            rx = Receiver(E.Config(device=local, max_sat=N_CH), raw_u8=raw_u8, report_lag=lag)
            n_dg, t_trk = 0, None
            E.sync(local)
            t0 = time.perf_counter()
            for i, b in enumerate(blocks):
                if t_trk is None and not rx.sweep_all_freq:
                    rx.drain()
                    t_trk, i_trk = time.perf_counter(), i
                if rx.feed(b) is not None:
                    n_dg += 1
            rx.drain()
            t1 = time.perf_counter()
            n_dg = len(rx.result_list)
            n_ch = len(rx.act_sat_set)
            last = pickle.loads(rx.result_list[-1]) if rx.result_list else None
            rx.close()
            r = {'input': 'raw uint16 (Q<<8|I), 2 B/sample' if raw_u8 else 'complex64, 8 B/sample',
                 'report_lag': lag, 'blocks': n_blocks, 'channels': n_ch, 'datagrams': n_dg,
                 'us_per_block': round((t1 - t0) / n_blocks * 1e6, 2),
                 'msamples_per_s': round(n_blocks * NGPS / (t1 - t0) / 1e6, 1),
                 'x_realtime': round(n_blocks * NGPS / (t1 - t0) / 2.048e6, 1),
                 'tracking_only': {
                     'blocks': n_blocks - i_trk,
                     'us_per_block': round((t1 - t_trk) / (n_blocks - i_trk) * 1e6, 2),
                     'x_realtime': round((n_blocks - i_trk) * NGPS / (t1 - t_trk) / 2.048e6, 1)},
                 'satellites_in_last_datagram': len(last[2]) if last else 0}
            if best is None or r['us_per_block'] < best['us_per_block']:
                best = r
        out.append(best)
    return {'what': 'pipeline.Receiver.feed per 32-ms block from host memory: processData\'s loop '
                    '(cold sweep, channel selection, satCalc, hand-off datagrams), host work and PCIe '
                    'inside the timed loop; better of two runs',
            'runs': out}


def measure_multi(E, sharding, dist, torch, lib, comm, rank, world, local, a, nb, trk_base, d_iq,
                  chans_all, states, cl_out, gather_peaks, cells, dt_mine, acq_ms, by_channel):
    """What an N > 1 run reports besides the headline (all ranks call this, after the timed
    region): the size of the RCCL communicator as RCCL reports it, every rank's own rate, the
    gather's own time, and north_star's channel split of the same tracking job next to the
    time-sharded headline.  Nothing here is part of `value`."""
    import ctypes as C
    out = {'world': world}
    # ---- the collective really spans the ranks
    n_r, my_r = C.c_int(0), C.c_int(-1)
    if comm is not None and lib.gpsmi_comm_count(comm, C.byref(n_r), C.byref(my_r)) == 0:
        out['rccl_ranks'] = int(n_r.value)
        t = torch.tensor([int(my_r.value)], dtype=torch.int64)
        allr = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allr, t)
        out['rccl_user_ranks'] = [int(x[0]) for x in allr]
    else:
        out['rccl_ranks'] = None                 # rehearsal / RCCL unavailable: see `collective`
    # ---- every rank's own step time (the headline takes the slowest)
    t = torch.tensor([dt_mine / a.steps], dtype=torch.float64)
    allt = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allt, t)
    per = [float(x[0]) for x in allt]
    nch_rank = len(chans_all) if not by_channel else None
    out['per_rank_ms_per_step'] = [round(x * 1e3, 4) for x in per]
    out['per_rank_msamples_per_s'] = [round(nb * NGPS / x / 1e6, 1) for x in per]
    out['one_gpu_equivalent_msamples_per_s'] = round(nb * NGPS / per[0] / 1e6, 1)
    # ---- the gather alone: equal counts agreed on the host first, then 20 calls
    sharding.agree_on_count(dist, cells)
    dist.barrier()
    E.sync(local)
    t0 = time.perf_counter()
    for _ in range(20):
        gather_peaks()
    tg = torch.tensor([(time.perf_counter() - t0) / 20], dtype=torch.float64)
    dist.all_reduce(tg, op=dist.ReduceOp.MAX)
    out['gather_us'] = round(float(tg[0]) * 1e6, 1)
    out['gather_bytes_per_rank'] = cells * 16
    if acq_ms:
        ta = torch.tensor([float(np.mean(acq_ms))], dtype=torch.float64)
        dist.all_reduce(ta, op=dist.ReduceOp.MAX)
        out['sharded_search_us'] = round(float(ta[0]) * 1e3, 1)
    # ---- the weak-scaling workload of a live installation: R independent receivers per GPU batched
    # on one handle (gpsmi_trk_set_streams), closed loop, state on the device; the receivers of a
    # rank read that rank's own blocks; nothing is exchanged
    R = 64
    steps = min(16, nb // R)
    if steps >= 2:
        trk3 = E.TrkEngine(E.Config(device=local), max_ch=len(chans_all), streams=R)
        for r in range(R):
            for c, (s, f, d) in enumerate(chans_all):
                trk3.open(c, s, f, d, stream=r)
        trk3.set_timing(False)
        for i in range(steps):                      # (the first pass uploads the state rows too)
            trk3.process(d_iq.at(trk_base + i * R * NGPS * 8), want_out=False)
        E.sync(local)
        dist.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            trk3.process(d_iq.at(trk_base + i * R * NGPS * 8), want_out=False)
        E.sync(local)
        dt3 = time.perf_counter() - t0
        trk3.close()
        rate, total, slowest = sharding.job_rate(dist, steps * R * NGPS, dt3)
        out['receiver_sharded'] = {
            'what': f'{R} independent receivers per GPU ({R * world} in all, {len(chans_all)} channels each), '
                    'closed loop on the device, no collective: the samples of all ranks over the slowest '
                    "rank's time",
            'receivers_per_gpu': R, 'receivers': sharding.shard_receivers(R, world - 1)[-1] + 1,
            'msamples_per_s': round(rate, 1), 'us_per_step_slowest_rank': round(slowest / steps * 1e6, 2),
            'x_realtime_per_receiver': round(NGPS / (slowest / steps) / 2.048e6, 1), 'scaling': 'weak'}
    # ---- north_star's split: the channels round-robin over the ranks, every rank all NB blocks
    if states is not None:
        mine = sharding.shard_channels(len(chans_all), rank, world)
        ms = None
        if mine:
            trk2 = E.TrkEngine(E.Config(device=local), max_ch=len(mine))
            for c, ci in enumerate(mine):
                s, f, d = chans_all[ci]
                trk2.open(c, s, f, d)
            trk2.replay_load(nb, np.ascontiguousarray(states[:, mine]),
                             np.ascontiguousarray(cl_out['delay_used'][:, mine]))
            pin = E.PinnedArray((nb, len(mine)), E.OUT_DTYPE)
            for _ in range(5):
                trk2.replay_run(d_iq.at(trk_base), nb)
        dist.barrier()
        if mine:
            E.sync(local)
            t0 = time.perf_counter()
            for _ in range(20):
                trk2.replay_run_async(d_iq.at(trk_base), nb)
                trk2.replay_fetch_async(pin.array)
                trk2.wait_prev()
            trk2.wait()
            ms = (time.perf_counter() - t0) / 20
            ok_ch = bool(pin.array.tobytes() == np.ascontiguousarray(cl_out[:, mine]).tobytes())
            trk2.close()
            pin.free()
        tc = torch.tensor([ms or 0.0, 1.0 if (not mine or ok_ch) else 0.0], dtype=torch.float64)
        tmax = tc.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tmin = tc.clone()
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        out['channel_sharded'] = {
            'what': f'the same {len(chans_all)}-channel job cut by channel (north_star, one worker per '
                    'SV as in the reference): every rank tracks its channels over all '
                    f'{nb} blocks, slowest rank',
            'channels_per_rank': [len(sharding.shard_channels(len(chans_all), r, world))
                                  for r in range(world)],
            'ms_per_step': round(float(tmax[0]) * 1e3, 4),
            'msamples_per_s': round(nb * NGPS / float(tmax[0]) / 1e6, 1) if float(tmax[0]) > 0 else None,
            'scaling': 'strong', 'outputs_equal_unsharded': bool(float(tmin[1]) == 1.0)}
    return out


# ---------------------------------------------------------------------- main
def main():
    # stdout carries exactly one JSON line: whatever libraries print there while we run
    # (gloo's connection banner, for one) is sent to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--settle-steps', type=int, default=300,
                    help='untimed steps ahead of the warm-up steps that bring the device to its '
                         'steady clocks (see the module docstring); 0 = none')
    ap.add_argument('--blocks', type=int, default=1024)
    ap.add_argument('--cpu-blocks', type=int, default=192)
    ap.add_argument('--dropin-blocks', type=int, default=640,
                    help='blocks the drop-in leg feeds through pipeline.Receiver.feed (20 s of signal)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the configs[3] / configs[4] legs')
    ap.add_argument('--overlap', action='store_true',
                    help='time the overlapped pipeline (option "corr_overlap": consecutive batches on two '
                         'streams) instead of the isolated one; either way both are measured and reported')
    ap.add_argument('--shard', choices=('time', 'channels'), default='time',
                    help='N > 1: how the tracking work is split over the ranks')
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
    dist = torch = None
    if world > 1:
        # torch bundles its own libamdhip64 under the same SONAME: it is loaded first, and
        # libgpsmi.so (not loaded yet: checked) then binds that one copy of the runtime
        assert 'gpsmi._lib' not in sys.modules, 'torch must be imported before libgpsmi.so is loaded'
        import torch
        import torch.distributed as dist
        dist.init_process_group('gloo')

    nb = a.blocks
    cores = os.cpu_count() or 1
    workers = max(1, min(16, cores // max(1, min(world, 8))))
    by_channel = world > 1 and a.shard == 'channels'

    # ---- inputs (host, before HIP is initialised: fork pools)
    first = 0 if by_channel else rank * (nb + N_ACQ_BLOCKS)      # time-sharded stream
    t0 = time.perf_counter()
    raw = generate_raw(first, nb + N_ACQ_BLOCKS, workers)
    raw0 = raw[0] if first == 0 else generate_raw(0, 1, 1)[0]    # the block every rank searches
    t_gen = time.perf_counter() - t0

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu:
        cpu, _ = cpu_baseline(raw, min(a.cpu_blocks, nb), cores)

    # ---- GPU set-up
    from gpsmi import engine as E
    from gpsmi import sharding
    from gpsmi.acquisition import Acquisition
    if a.rehearse:
        local = local % E.device_count()
    cfg = E.Config(device=local)
    dev_name = E.device_name(local)
    acq = Acquisition(cfg)
    d_raw = E.DeviceBuffer(raw.nbytes, local)
    d_raw.upload(raw)
    d_iq = E.DeviceBuffer(raw.size * 8, local)
    E.unpack_u8iq(d_iq.ptr, d_raw.ptr, raw.size, local)   # streamData's decode
    blk_bytes = NGPS * 8                                  # (d_raw stays: the fused-ingest leg reads it)
    d_iq0 = d_iq                                          # block 0 of the common stream
    if first != 0:
        d_raw0 = E.DeviceBuffer(raw0.nbytes, local)
        d_raw0.upload(raw0)
        d_iq0 = E.DeviceBuffer(raw0.size * 8, local)
        E.unpack_u8iq(d_iq0.ptr, d_raw0.ptr, raw0.size, local)
        d_raw0.free()

    # cold acquisition with the reference's own first-hit loop -> channels
    sat_lst, found, freq = list(range(2, 33)), [], cfg.min_freq
    for b in range(N_ACQ_BLOCKS):
        _, freq, found = acq.sweepAllSats((d_iq.at(b * blk_bytes), NGPS), freq,
                                          sat_lst, found, cfg.it_sweep_all)
    chans_all = [(s, f, d) for _, s, f, d in found][:N_CH]
    my_ch = sharding.shard_channels(len(chans_all), rank, world) if by_channel \
        else list(range(len(chans_all)))
    chans = [chans_all[c] for c in my_ch]
    nch = max(1, len(chans))
    trk = E.TrkEngine(cfg, max_ch=nch)
    for c, (s, f, d) in enumerate(chans):
        trk.open(c, s, f, d)

    # closed loop over the batch: the trajectory (state at block start) + outputs
    states = np.zeros((nb, nch), dtype=E.STATE_DTYPE)
    cl_out = np.zeros((nb, nch), dtype=E.OUT_DTYPE)
    trk_base = N_ACQ_BLOCKS * blk_bytes
    E.sync(local)
    t0 = time.perf_counter()
    cl_dev_ms = 0.0
    for i in range(nb):
        for c in range(nch):
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
    batched = measure_batched(E, local, lambda b: d_iq.at(trk_base + b * blk_bytes), nb, chans) \
        if rank == 0 and not a.no_extra else []

    # the acquisition leg of a step: configs[1] on one GPU, configs[3] sharded by SV on N
    prn_all = list(range(1, 33))
    if world == 1:
        acq_freqs, acq_navg = [-5000.0 + 250.0 * i for i in range(41)], 1
        shard, n_real, width = prn_all, 32, 32
    else:
        acq_freqs, acq_navg = [-5000.0 + 50.0 * i for i in range(201)], 10
        shard, n_real, width = sharding.padded_shard(prn_all, rank, world)
    cells = len(acq_freqs) * width
    comm = d_send = d_recv = gathered = lib = None
    use_rccl = world > 1 and not a.rehearse
    collective = 'none'
    if world > 1:
        import ctypes as C
        from gpsmi import _lib
        lib = _lib.load()
        idb = np.zeros(_lib.COMM_ID_BYTES, np.uint8)
        if rank == 0:
            E.check(lib.gpsmi_comm_unique_id(E.ptr(idb)), 'comm_unique_id')
        tid = torch.from_numpy(idb)
        dist.broadcast(tid, 0)
        collective = 'rehearsal: gloo all-gather'
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
        d_send = E.DeviceBuffer(cells * 16, local)
        d_recv = E.DeviceBuffer(cells * 16 * world, local)
        gathered = np.zeros(cells * world, dtype=E.PEAK_DTYPE)

    trk.replay_load(nb, states, cl_out['delay_used'])
    # two page-locked result buffers: the read-back of step k overlaps the kernels of step k+1
    pins = [E.PinnedArray((nb, nch), E.OUT_DTYPE) for _ in range(2)]

    def barrier():
        if dist is not None:
            dist.barrier()
        E.sync(local)

    corr_ms, total_ms, acq_ms = [], [], []
    recording = [False]
    acq_pin = E.PinnedArray((len(acq_freqs), width), E.PEAK_DTYPE)
    acq_n = acq_navg * 2048

    cold_corr_ms, cp_ms = [], []
    ovl = {'total': [], 'corr': [], 'cp': []}      # the same kernels' times inside the overlapped pipeline

    def record_last(cold=False, full=True):
        t, c = trk.last_ms()
        if cold == 'cold':
            cold_corr_ms.append(c)
            return
        if cold == 'overlapped':
            ovl['corr'].append(c)
            if full:
                ovl['total'].append(t)
                ovl['cp'].append(trk.last_codephase_ms())
            return
        corr_ms.append(c)
        if full:
            total_ms.append(t)
            cp_ms.append(trk.last_codephase_ms())

    def gather_peaks():
        """all-gather of the (padded, equal-sized) shard tables -> `gathered`"""
        if use_rccl:
            E.check(lib.gpsmi_comm_allgather_peaks(comm, d_send.ptr, d_recv.ptr, cells,
                                                   E.ptr(gathered)), 'allgather')
        else:                                              # labelled fallback, see `collective`
            mine = torch.from_numpy(acq_pin.array.view(np.uint8).reshape(-1).copy())
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            gathered[:] = np.concatenate([p.numpy().view(E.PEAK_DTYPE) for p in parts])

    host_ns = {'wait': 0, 'all': 0}                  # host time inside the two blocking waits / inside step()

    def finish_search():
        """wait for the search enqueued one step ago; gather its peak records"""
        tw = time.perf_counter_ns()
        acq.engine.wait()
        host_ns['wait'] += time.perf_counter_ns() - tw
        if recording[0]:
            acq_ms.append(acq.engine.last_ms())
        if world > 1:
            gather_peaks()

    def step(k, record):
        ts = time.perf_counter_ns()
        step_body(k, record)
        host_ns['all'] += time.perf_counter_ns() - ts

    def step_body(k, record):
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
        # the kernel-timing events of the other kernels are barrier packets in the queue
        # (~5 us each): the kernels of every fourth step are timed, the others run without
        # (on the steps between, only the correlator's own dispatch stamps are taken: no packet in
        # the queue, so the roofline kernel's duration is the mean over EVERY timed step)
        trk.set_timing(1 if k % TIMED_EVERY == 0 else 2)
        # (the tracking batch goes into its queue first: its first kernel is what the GPU is
        # waiting for when the host is late; the search has the whole of that kernel's time)
        trk.replay_run_async(d_iq.at(trk_base), nb)
        acq.engine.search_async(d_iq0.ptr, acq_n, shard, acq_freqs, acq_navg, acq_pin.array,
                                d_send.ptr if world > 1 else None)
        trk.replay_fetch_async(pins[k & 1].array)
        tw = time.perf_counter_ns()
        trk.wait_prev()
        host_ns['wait'] += time.perf_counter_ns() - tw
        if record and k > 0:
            record_last(cold=record, full=(k - 1) % TIMED_EVERY == 0)

    # Settling: an idle MI355X needs some 40 ms under load before its clocks stop moving - the
    # same correlator launch takes ~120 us at the start of a run and ~103 us from the 100th
    # step on, flat from there (profiles/round2/clock_settling.txt).  A 20-step timed region
    # started cold would measure the ramp, not the kernels, so the device is first driven with
    # the very step that is timed (a.settle_steps of them, the count is in the JSON line); the
    # W warm-up steps and the K timed steps follow unchanged.
    # (the correlator launches of the first 24 of them are timed: `roofline.frac_cold`, what a run
    # that starts its clock on an idle device sees)
    # Two pipelines over the same step, both measured in every run and both named in the JSON line.
    # ISOLATED (the default for the timed region): one batch at a time, every kernel alone on the chip,
    # its duration is what a roofline can be quoted on.  OVERLAPPED (option "corr_overlap"): consecutive
    # batches alternate between two streams, so that the code-phase correlation of batch k + 1 is
    # eligible while the correlator of batch k runs.  The pass that is not timed for `value` runs
    # OVL_STEPS steps behind the timed region.  (Measured: the overlapped pipeline is 2-7 % SLOWER here;
    # DESIGN.md section 4.6 has the arithmetic -- the two big kernels both live on SIMD issue slots and
    # cannot share a CU's LDS -- so the isolated one stays the default.)
    overlap = a.overlap
    OVL_STEPS = 24
    for k in range(a.settle_steps):
        step(k, 'cold' if k < 26 else False)
    if a.settle_steps:
        finish_search()
        trk.wait()
    if overlap:
        trk.set_option('corr_overlap', 1)
    for k in range(a.warmup):
        step(k, False)
    if a.warmup:
        finish_search()
    trk.wait()
    barrier()
    recording[0] = True
    host_ns['wait'] = host_ns['all'] = 0
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k, 'overlapped' if overlap else True)
    host_timed = dict(host_ns)
    finish_search()                         # the last step: its search, ...
    trk.wait()                              # ... its kernels and its copy
    barrier()
    dt = time.perf_counter() - t0
    record_last('overlapped' if overlap else False, full=(a.steps - 1) % TIMED_EVERY == 0)
    recording[0] = False
    # the other pipeline, OVL_STEPS steps (untimed for `value`)
    trk.set_option('corr_overlap', 0 if overlap else 1)
    for k in range(4):
        step(k, False)
    finish_search()
    trk.wait()
    E.sync(local)
    recording[0] = overlap                  # (the search's own time belongs to the isolated pipeline)
    t1 = time.perf_counter()
    for k in range(OVL_STEPS):
        step(k, True if overlap else 'overlapped')
    finish_search()
    trk.wait()
    other_ms_step = (time.perf_counter() - t1) / OVL_STEPS * 1e3
    recording[0] = False
    trk.set_option('corr_overlap', 0)
    trk.set_timing(True)
    # the results copy of a step alone (4.6 MB of records into page-locked memory): the pipeline hides
    # it behind the next step's kernels as long as it is shorter than a step
    copy_ms = []
    for _ in range(6):
        tc = time.perf_counter()
        trk.replay_fetch(pins[0].array)
        copy_ms.append((time.perf_counter() - tc) * 1e3)
    copy_ms = float(np.median(copy_ms[1:]))
    if dist is not None:
        dt_mine = dt
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
    checks = {'replay_equals_closed_loop': bool(same), 'replay_states_chain': bool(chained),
              'channels_locked_at_end': locked, 'acquired': len(found)}
    if dist is not None:
        okt = torch.tensor([1 if (same and chained) else 0])
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        checks['all_ranks_replay_ok'] = bool(int(okt[0]))
        # the gathered, merged search must equal the unsharded search of the same IQ
        if rank == 0:
            tabs = [gathered[r * cells:(r + 1) * cells].reshape(len(acq_freqs), width)
                    for r in range(world)]
            merged = sharding.merge_peak_tables(tabs, prn_all, world)
            full = acq.engine.search((d_iq0.ptr, acq_n), prn_all, acq_freqs, acq_navg)
            checks['gathered_search_equals_unsharded'] = bool(merged.tobytes() == full.tobytes())
        if by_channel:                      # channel-sharded outputs, merged in channel order
            wch = -(-len(chans_all) // world)
            mine = np.zeros((nb, wch), dtype=E.OUT_DTYPE)
            mine[:, :len(chans)] = cl_out[:, :len(chans)]
            send = torch.from_numpy(mine.view(np.uint8).reshape(-1).copy())
            parts = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(parts, send)     # (after the timed region: result check only)
            if rank == 0:
                per = [parts[r].numpy().view(E.OUT_DTYPE).reshape(nb, wch)
                       [:, :len(sharding.shard_channels(len(chans_all), r, world))]
                       for r in range(world)]
                merged_out = sharding.merge_channel_outputs(per, len(chans_all), world)
                checks['merged_channels'] = int((merged_out[-1]['prn'] > 0).sum())

    multi = None
    if dist is not None:
        multi = measure_multi(E, sharding, dist, torch, lib, comm, rank, world, local, a, nb,
                              trk_base, d_iq, chans_all, states if not by_channel else None,
                              cl_out if not by_channel else None, gather_peaks, cells, dt_mine,
                              acq_ms, by_channel)
    extra = []
    if rank == 0 and not a.no_extra:
        extra.append(measure_cfg4(E, acq, d_iq0.ptr))
        if world == 1:
            extra.append(measure_u8(E, local, d_raw, nb, chans, states, cl_out['delay_used'], cl_out))
            extra.append(measure_cfg5(E, local))
            extra.append(measure_block_length(E, local, 16))
            extra.append(measure_block_length(E, local, 8))
            extra.append(measure_streamed(E, local, raw[N_ACQ_BLOCKS:N_ACQ_BLOCKS + 96], chans))
    dropin = None
    if rank == 0 and world == 1 and not a.no_extra:
        dropin = measure_dropin(E, local, raw, min(nb, a.dropin_blocks) + N_ACQ_BLOCKS)

    if rank == 0:
        samples = nb * NGPS
        ms_step = dt / a.steps * 1e3
        scale = 1 if by_channel else world
        value = samples * scale / (dt / a.steps) / 1e6
        k_ms = float(np.mean(corr_ms))
        alg_bytes = samples * 8
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic('trk_span_kernel') if nb == 1024 else (None, None)
        if world == 1:
            sharding_txt = '1 GPU'
            acq_txt = 'configs[1]: 32 SV x 41 Doppler x 1 ms acquisition search'
        else:
            acq_txt = (f'configs[3]: 32 SV x 201 Doppler x 10 ms acquisition search, SVs sharded '
                       f'{width} per rank, peaks by {collective}')
            sharding_txt = (f'{world} ranks: tracking sharded '
                            + ('by channel (round-robin, every rank reads the same IQ; '
                               'north_star / reference layout)' if by_channel else
                               'in time (each rank its own blocks of the stream)')
                            + f', acquisition sharded by SV, peaks by {collective}')
        line = {
            'metric': 'IQ Msamples/s ingested, 32-SV acq + 12-ch track',
            'value': round(value, 1), 'unit': 'Msamples/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'settle_steps': a.settle_steps,
            'ms_per_step': round(ms_step, 4), 'higher_is_better': True,
            'scaling': 'strong' if by_channel else 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'split': ('one GPU' if world == 1 else
                      'channels (north_star / the reference: one worker per SV; every rank reads all the IQ)'
                      if by_channel else
                      'time (replay only): every rank replays its own blocks of the stream from a recorded '
                      'trajectory -- linear by construction; north_star\'s channel split and the receiver '
                      'split of a live installation are in multi_gpu.channel_sharded / .receiver_sharded'),
            'config': {
                'workload': ('BASELINE configs[2]: 12-channel tracking of '
                             f'{nb} x 32 ms blocks (65536 complex64 each, '
                             f'{samples * 8 / 2**20:.0f} MiB resident in HBM) '
                             'in replay of the closed-loop trajectory, results '
                             'copied to host (the copy of a step overlaps the '
                             'kernels of the next); preceded per step by ' + acq_txt),
                'channels': len(chans_all), 'blocks': nb, 'sample_rate_hz': 2048000,
                'sharding': sharding_txt,
            },
            'x_realtime': round(value / 2.048, 1),
            'pipeline': {
                'timed': ('overlapped: consecutive batches alternate between two streams (option '
                          '"corr_overlap"), the code-phase correlation of batch k + 1 runs beside the '
                          'correlator of batch k') if overlap else
                         'isolated: one batch at a time, every kernel alone on the chip',
                'other_pass_steps': OVL_STEPS + 4,
                'isolated_ms_per_step': round(other_ms_step if overlap else ms_step, 4),
                'overlapped_ms_per_step': round(ms_step if overlap else other_ms_step, 4),
                'note': f'the pipeline that is not the timed one ran {OVL_STEPS} steps behind the timed region; '
                        'roofline.kernel_ms and kernels_ms are the isolated pipeline\'s (every kernel alone '
                        'on the chip)',
                'overlapped_kernels_ms': ({'tracking_all': round(float(np.mean(ovl['total'])), 4),
                                           'correlator': round(float(np.mean(ovl['corr'])), 4),
                                           'codephase_correlation': round(float(np.mean(ovl['cp'])), 4)}
                                          if ovl['corr'] else None),
                # where the host thread spent the timed region: inside its two blocking waits (the
                # device is the bound) or enqueuing (the host is)
                'host_us_per_step': round(host_timed['all'] / a.steps / 1e3, 1),
                'host_wait_us_per_step': round(host_timed['wait'] / a.steps / 1e3, 1),
                'results_copy_ms': round(copy_ms, 4),
                'results_copy_mb': round(pins[0].array.nbytes / 1e6, 2),
            },
            'roofline': {
                'bound': 'hbm', 'kernel': 'trk_span_kernel',
                'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                'traffic': traffic, 'traffic_source': traffic_src,
                'traffic_codephase_correlation': (pmc_traffic('trk_corr_kernel<4, 0>')[0] if nb == 1024 else None),
                'traffic_note': 'HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE) from the committed counter '
                                'pass, null unless that pass was taken from the present kernel sources; the '
                                'code-phase correlation reads the 134 MB of centre rows (one fetch per block: '
                                'its channel groups share an XCD)',
                'kernel_ms': round(k_ms, 4),
                'kernel_ms_source': 'hipExtLaunchKernel begin/end stamps of the dispatch, mean over ALL '
                                    f'{len(corr_ms)} steps of the isolated pass (the kernel alone on the chip, '
                                    'steady clocks)',
                'algorithmic_bytes_per_launch': alg_bytes,
                'frac_vs_measured_copy': round(achieved / HBM_COPY_GBS, 4),
                'frac_cold': round(alg_bytes / (float(np.mean(cold_corr_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                if cold_corr_ms else round(achieved / HBM_PEAK_GBS, 4),
                'frac_cold_note': 'the same kernel over the first ~25 steps from an idle device '
                                  '(settle_steps = 0 makes the timed region itself that)',
                'step_hbm_frac': round(alg_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            },
            'kernels_ms': {
                'tracking_all': round(float(np.mean(total_ms)), 4),
                'correlator': round(k_ms, 4),
                'acquisition_search': round(float(np.mean(acq_ms)), 4),
                'codephase_correlation': round(float(np.mean(cp_ms)), 4),
            },
            'issue_rooflines': [
                issue_roofline([('replay trk_corr_kernel<4, 0>', 1)], float(np.mean(cp_ms)),
                               'trk_corr_kernel<4,0> inside the step (the search and the previous '
                               'epilogue run beside it)'),
                issue_roofline([('acq2 acq_spectrum_kernel<1', 1), ('acq2 acq_corr_kernel', 1)],
                               float(np.mean(acq_ms)),
                               'configs[1] search inside the step: acq_spectrum_kernel<1> + acq_corr_kernel'),
            ] if world == 1 else None,
            'closed_loop': {
                'value': round(samples / t_closed / 1e6, 1),
                'unit': 'Msamples/s',
                'x_realtime': round(samples / t_closed / 2.048e6, 1),
                'us_per_block': round(t_closed / nb * 1e6, 2),
                'with_per_block_readback_msps':
                    round(samples / t_closed_with_readback / 1e6, 1),
                'device_ms_per_block': round(cl_dev_ms / nb, 4),
                'batched': batched,
                'batched_note': 'R independent receivers per handle (gpsmi_trk_set_streams), each '
                                'bytewise equal to its solo closed loop (tests/test_gpu_trk.py)',
            },
            'configs': extra,
            'dropin': dropin,
            'multi_gpu': multi,
            'checks': checks,
            'device': dev_name,
            'setup_s': {'generate_iq': round(t_gen, 2)},
        }
        if cpu is not None:
            line['cpu_baseline'] = cpu
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + '\n').encode())
        if not (same and chained) or checks.get('gathered_search_equals_unsharded') is False \
                or checks.get('all_ranks_replay_ok') is False:
            sys.exit('bench: a result check failed: ' + json.dumps(checks))
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
