"""How the path shards over the GPUs of one node.  Pure host
logic, shared by bench.py and the multi-process tests.

Acquisition: the SV list is split contiguously (32 SVs on 8 GPUs = 4 each,
BASELINE configs[3]); every rank searches its shard on the same IQ and the
one exchange of the path is an all-gather of the 16-byte peak records (RCCL on
the GPU, gpsmi_comm_allgather_peaks).  A collective needs equal counts on every
rank, so a shard is padded to ceil(nsv / world) columns by searching its last SV
again (padded_shard); merge_peak_tables drops the pad columns.

Tracking, two splits:
 * by CHANNEL (the reference's own layout, one worker per SV, gpsrecv.py:340-360,
   :404-417, and north_star's "SV channels shard"): channels round-robin over the
   ranks (12 on 8 GPUs = 2,2,2,2,1,1,1,1), every rank reads the same IQ, the
   per-channel outputs return through each rank's host and are merged in channel
   order (shard_channels / merge_channel_outputs): no collective.  A fixed job is
   cut up: strong scaling.
 * in TIME (shard_blocks): rank r tracks its own segment of the stream from a
   known trajectory (replay).  Per-rank work is fixed: weak scaling; this is what
   bench.py --gpus N times by default, the other with --shard channels.  The split exists
   for REPLAY only (a live stream has no recorded trajectory to start a later segment from).
 * by RECEIVER (shard_receivers): the weak-scaling workload of a live installation --
   independent receivers (antennas / IQ streams), R per GPU batched on one tracking handle
   (gpsmi_trk_set_streams), closed loop.  Receivers share nothing: no collective; the job's rate
   is the samples of all ranks over the slowest rank's time (job_rate)."""
import numpy as np


def shard_svs(prns, rank, world):
    """Contiguous, near-equal split of the SV list; every SV in exactly one shard."""
    n = len(prns)
    return list(prns[rank * n // world:(rank + 1) * n // world])


def shard_blocks(first_block, blocks_per_rank, rank):
    """Time sharding of the stream: rank r owns blocks_per_rank consecutive blocks."""
    lo = first_block + rank * blocks_per_rank
    return lo, lo + blocks_per_rank


def padded_shard(prns, rank, world):
    """(SV list of this rank padded to the common width by repeating its last SV,
    number of real SVs, width).  An empty shard (more ranks than SVs) searches
    prns[0] `width` times and contributes no column."""
    mine = shard_svs(prns, rank, world)
    width = -(-len(prns) // world)
    fill = mine[-1] if mine else prns[0]
    return mine + [fill] * (width - len(mine)), len(mine), width


def shard_channels(nch, rank, world):
    """Round-robin split of the tracking channels: rank r owns r, r + world, ..."""
    return list(range(rank, nch, world))


def merge_channel_outputs(parts, nch, world):
    """parts[r]: array [nb, len(shard_channels(nch, r, world))] of per-block records of
    rank r's channels -> [nb, nch] in channel order."""
    first = next(p for p in parts if p.shape[1] > 0)
    out = np.zeros((first.shape[0], nch), dtype=first.dtype)
    for r in range(world):
        idx = shard_channels(nch, r, world)
        if idx:
            out[:, idx] = parts[r]
    return out


def pad_table(table, width):
    """[nbins, nsv_r] -> [nbins, width] so that every rank gathers equal sizes."""
    out = np.zeros((table.shape[0], width), dtype=table.dtype)
    out[:, :table.shape[1]] = table
    return out


def merge_peak_tables(gathered, prns, world):
    """gathered: [world, nbins, width] padded shard tables in rank order ->
    [nbins, len(prns)] in the order of prns."""
    cols = []
    for r in range(world):
        n = len(shard_svs(prns, r, world))
        cols.append(gathered[r][:, :n])
    return np.concatenate(cols, axis=1)


def shard_receivers(per_gpu, rank):
    """Global ids of the independent receivers rank `rank` tracks: per_gpu on every GPU."""
    return list(range(rank * per_gpu, (rank + 1) * per_gpu))


def merge_receiver_tables(parts):
    """parts[r]: array [per_gpu, ...] of rank r's per-receiver results -> [world * per_gpu, ...]
    in global receiver order (the inverse of shard_receivers)."""
    return np.concatenate(parts, axis=0)


def job_rate(dist, samples, seconds):
    """Whole-job throughput of a sharded run with no data-path collective: the samples of all
    ranks over the SLOWEST rank's time (bench.py's max-over-ranks).  dist: torch.distributed or
    None (one process).  -> (Msamples/s, total samples, slowest seconds)."""
    if dist is None:
        return samples / seconds / 1e6, int(samples), float(seconds)
    import torch
    t = torch.tensor([float(samples), 0.0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    m = torch.tensor([float(seconds)], dtype=torch.float64)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return float(t[0]) / float(m[0]) / 1e6, int(t[0]), float(m[0])


class CountMismatch(ValueError):
    """The ranks do not agree on the size of a collective (the GPSMI_E_ARG of the host side)."""


def agree_on_count(dist, count):
    """Every rank of a collective must contribute the same number of records: RCCL (like NCCL)
    does not return from an all-gather whose ranks disagree.  One tiny host-side reduction
    (the process group the launcher already set up: gloo) before the GPU collective turns that
    hang into an error on EVERY rank.  dist: torch.distributed (initialised); returns count."""
    import torch
    t = torch.tensor([int(count), -int(count)], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)            # (max count, -min count)
    hi, lo = int(t[0]), -int(t[1])
    if hi != lo:
        raise CountMismatch(f'ranks disagree on the record count of the gather: {lo} .. {hi} '
                            f'(this rank: {int(count)})')
    return int(count)
