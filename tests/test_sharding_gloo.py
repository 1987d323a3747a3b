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
    mine = sharding.shard_svs(prns, rank, world)
    t = orc.acq_table(data, freqs, mine, 2, orc.Params())
    tab = np.zeros((len(freqs), len(mine)), dtype=PEAK_DTYPE)
    for k in ('argmax', 'peak', 'mean', 'std'):
        tab[k] = t[k]
    width = -(-len(prns) // world)
    send = torch.from_numpy(sharding.pad_table(tab, width).view(np.uint8).copy())
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


def test_sv_sharding_and_gather_world2():
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, tmax, shape = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and tmax == 2.0 and shape == (3, 11)


def test_shards_tile_the_sv_list():
    from gpsmi import sharding
    prns = list(range(1, 33))
    for world in (1, 2, 3, 4, 8):
        got = sum((sharding.shard_svs(prns, r, world) for r in range(world)), [])
        assert got == prns
    assert sharding.shard_blocks(5, 1024, 3) == (5 + 3072, 5 + 4096)
