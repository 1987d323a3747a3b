"""How the path shards over the GPUs of one node: satellites for
acquisition, stream segments for tracking.  Pure host logic, shared by
bench.py and the multi-process tests; the exchange itself is one all-gather of
peak records (RCCL on the GPU, gpsmi_comm_allgather_peaks)."""
import numpy as np


def shard_svs(prns, rank, world):
    """Contiguous, near-equal split of the SV list; every SV in exactly one shard."""
    n = len(prns)
    return list(prns[rank * n // world:(rank + 1) * n // world])


def shard_blocks(first_block, blocks_per_rank, rank):
    """Time sharding of the stream: rank r owns blocks_per_rank consecutive blocks."""
    lo = first_block + rank * blocks_per_rank
    return lo, lo + blocks_per_rank


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
