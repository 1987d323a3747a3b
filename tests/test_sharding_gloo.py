"""The N > 1 path on CPU: two processes (gloo), SVs sharded by
gpsmi.sharding, every rank searches its shard, the peak tables are all-gathered
and merged; the result must equal the unsharded search.  The GPU run replaces
the stand-in search (the oracle here) by gpsmi_acq_search_dev and gloo by
gpsmi_comm_allgather_peaks (RCCL); sharding and merge are the same code."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path[:0] = [os.path.join(ROOT, 'gps-sdr-receiver_amd'), os.path.join(ROOT, 'oracle')]
    import torch
    import torch.distributed as dist
    import gps_oracle as orc
    from gpsmi import sharding, synth
    from gpsmi._lib import PEAK_DTYPE
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    prns = list(range(1, 12))                       # 11 SVs: uneven split
    freqs = [-500.0, 0.0, 500.0]
    data = synth.default_scene(3, seed=5, prns=[2, 7, 11]).block(0, n=4096)
    mine, n_real, width = sharding.padded_shard(prns, rank, world)   # as bench.py does
    t = orc.acq_table(data, freqs, mine, 2, orc.Params())
    tab = np.zeros((len(freqs), width), dtype=PEAK_DTYPE)
    for k in ('argmax', 'peak', 'mean', 'std'):
        tab[k] = t[k]
    send = torch.from_numpy(tab.view(np.uint8).copy())
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    gathered = [r.numpy().view(PEAK_DTYPE).reshape(len(freqs), width) for r in recv]
    merged = sharding.merge_peak_tables(gathered, prns, world)
    tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)       # bench.py's max-over-ranks
    dist.barrier()
    if rank == 0:
        full = orc.acq_table(data, freqs, prns, 2, orc.Params())
        ok = all(np.array_equal(merged[k], full[k].astype(merged[k].dtype))
                 for k in ('argmax', 'peak', 'mean', 'std'))
        q.put((ok, float(tt[0]), merged.shape))
    dist.destroy_process_group()


def _run_world(target, world):
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_sv_sharding_and_gather_world2():
    ok, tmax, shape = _run_world(_worker, 2)
    assert ok and tmax == 2.0 and shape == (3, 11)


def test_sv_sharding_and_gather_world3_uneven():
    """11 SVs on 3 ranks (4, 3, 4): the shards are padded to equal counts for the gather."""
    ok, tmax, shape = _run_world(_worker, 3)
    assert ok and tmax == 3.0 and shape == (3, 11)


def _track_worker(rank, world, port, q):
    """Channel-sharded tracking (the reference's one-worker-per-SV layout): every rank
    runs ITS channels on the same blocks, the records are gathered and merged in channel
    order; must equal the unsharded run.  Stand-in compute: the oracle's SatStream."""
    sys.path[:0] = [os.path.join(ROOT, 'gps-sdr-receiver_amd'), os.path.join(ROOT, 'oracle')]
    import torch
    import torch.distributed as dist
    import gps_oracle as orc
    from gpsmi import sharding, synth
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    p = orc.Params()
    chans = [(21, 3200.0, 1458), (4, -2000.0, 143), (6, 1200.0, 1023)]     # of the fixture scene
    nb = 3
    sc = synth.default_scene(12, seed=7)
    blocks = [sc.block(5 + i) for i in range(nb)]
    rec_dt = np.dtype([('delay', 'i4'), ('freq', 'f4'), ('phase', 'f4'), ('code_phase', 'f8')])

    def run(idx):
        out = np.zeros((nb, len(idx)), dtype=rec_dt)
        for k, c in enumerate(idx):
            sv, f0, d0 = chans[c]
            ss = orc.SatStream(sv, f0, p, delay=d0)
            for i in range(nb):
                _, _, co_ph, _ = ss.process(blocks[i], np.int64((5 + i + 1) * p.ngps))
                out[i, k] = (ss.delay, ss.freq, ss.phase, co_ph)
        return out

    mine = sharding.shard_channels(len(chans), rank, world)
    width = -(-len(chans) // world)
    loc = np.zeros((nb, width), dtype=rec_dt)
    loc[:, :len(mine)] = run(mine)
    send = torch.from_numpy(loc.view(np.uint8).copy())
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)           # (on the GPU path the records return through each host)
    parts = []
    for r in range(world):
        n = len(sharding.shard_channels(len(chans), r, world))
        parts.append(recv[r].numpy().view(rec_dt).reshape(nb, width)[:, :n])
    merged = sharding.merge_channel_outputs(parts, len(chans), world)
    dist.barrier()
    if rank == 0:
        full = run(list(range(len(chans))))
        q.put((merged.tobytes() == full.tobytes(), merged.shape))
    dist.destroy_process_group()


def test_channel_sharded_tracking_merge_world2():
    ok, shape = _run_world(_track_worker, 2)
    assert ok and shape == (3, 3)


def _receiver_worker(rank, world, port, q):
    """Receiver-sharded tracking, the weak-scaling workload of a live installation: every rank
    tracks ITS independent receivers (each its own IQ stream), nothing is exchanged on the data
    path; the per-receiver results gathered in global receiver order must equal the one-process
    run of all receivers, and the job's rate is all samples over the slowest rank's time.
    Stand-in compute: the oracle's SatStream (on the GPU: gpsmi_trk_set_streams)."""
    sys.path[:0] = [os.path.join(ROOT, 'gps-sdr-receiver_amd'), os.path.join(ROOT, 'oracle')]
    import torch
    import torch.distributed as dist
    import gps_oracle as orc
    from gpsmi import sharding, synth
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    p = orc.Params()
    per_gpu, nb = 2, 2
    rec_dt = np.dtype([('receiver', 'i4'), ('delay', 'i4'), ('freq', 'f4'), ('code_phase', 'f8')])

    def run(ids):
        out = np.zeros(len(ids), dtype=rec_dt)
        for k, g in enumerate(ids):                      # receiver g: its own scene (seed)
            sc = synth.default_scene(2, seed=100 + g)
            s = sc.sats[0]
            ss = orc.SatStream(s.prn, round(s.doppler / 200.0) * 200.0, p, delay=int(round(s.delay)) % 2048)
            for i in range(nb):
                _, _, co_ph, _ = ss.process(sc.block(i), np.int64((i + 1) * p.ngps))
            out[k] = (g, ss.delay, ss.freq, co_ph)
        return out

    mine = sharding.shard_receivers(per_gpu, rank)
    loc = run(mine)
    send = torch.from_numpy(loc.view(np.uint8).copy())
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)           # (result check only: the data path has no collective)
    merged = sharding.merge_receiver_tables([r.numpy().view(rec_dt) for r in recv])
    rate, total, slowest = sharding.job_rate(dist, per_gpu * nb * p.ngps, 0.5 * (rank + 1))
    dist.barrier()
    if rank == 0:
        full = run(list(range(world * per_gpu)))
        q.put((merged.tobytes() == full.tobytes(), list(merged['receiver']), rate, total, slowest))
    dist.destroy_process_group()


def test_receiver_sharded_tracking_merge_world2():
    ok, ids, rate, total, slowest = _run_world(_receiver_worker, 2)
    assert ok and ids == [0, 1, 2, 3]
    assert total == 2 * 2 * 2 * 65536 and slowest == 1.0          # all samples, the slowest rank's time
    assert abs(rate - total / 1.0 / 1e6) < 1e-9
    from gpsmi import sharding
    assert sharding.job_rate(None, 1000, 0.5) == (0.002, 1000, 0.5)
    assert sharding.shard_receivers(64, 3) == list(range(192, 256))


def test_shards_tile_the_sv_list():
    from gpsmi import sharding
    prns = list(range(1, 33))
    for world in (1, 2, 3, 4, 8):
        got = sum((sharding.shard_svs(prns, r, world) for r in range(world)), [])
        assert got == prns
    assert sharding.shard_blocks(5, 1024, 3) == (5 + 3072, 5 + 4096)
    # tracking channels round-robin: 12 on 8 ranks = 2,2,2,2,1,1,1,1
    assert [len(sharding.shard_channels(12, r, 8)) for r in range(8)] == [2, 2, 2, 2, 1, 1, 1, 1]
    assert sorted(sum((sharding.shard_channels(12, r, 8) for r in range(8)), [])) == list(range(12))
    for world in (2, 3, 5, 8):      # padded shards: equal counts, pad = a repeated SV
        for r in range(world):
            mine, n, width = sharding.padded_shard(prns, r, world)
            assert len(mine) == width == -(-32 // world) and mine[:n] == sharding.shard_svs(prns, r, world)


def _worker_count_mismatch(rank, world, port, q):
    sys.path[:0] = [os.path.join(ROOT, 'gps-sdr-receiver_amd')]
    import torch.distributed as dist
    from gpsmi import sharding
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    events = []
    # a consistent count passes on every rank ...
    events.append(('ok', sharding.agree_on_count(dist, 804)))
    # ... a rank with a different count makes EVERY rank raise before any GPU collective is
    # entered (RCCL would not return from it), and the group stays usable afterwards
    try:
        sharding.agree_on_count(dist, 804 if rank != 1 else 803)
        events.append(('no error', None))
    except sharding.CountMismatch as e:
        events.append(('mismatch', '803 .. 804' in str(e)))
    events.append(('ok', sharding.agree_on_count(dist, 12)))
    dist.barrier()
    q.put((rank, events))
    dist.destroy_process_group()


def test_unequal_gather_counts_raise_on_every_rank_instead_of_hanging():
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_worker_count_mismatch, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        assert got[rank] == [('ok', 804), ('mismatch', True), ('ok', 12)], rank
