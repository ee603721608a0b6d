"""Env-parallel sharding: tables are independent (no cross-table reads anywhere in pokerl/game.py), so a node runs one
process per GPU, each owning a contiguous block of tables.  No collective is needed on the step path; RNG streams are
keyed by the GLOBAL table id, so results are identical for any number of shards."""


def shard_tables(total_tables: int, rank: int, world_size: int):
    """Contiguous block of `total_tables` owned by `rank`: returns (num_local_tables, table_id_base)."""
    if not (0 <= rank < world_size):
        raise ValueError('rank out of range')
    base, rem = divmod(total_tables, world_size)
    n = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return n, start


def gather_f64(game, field=3, dist=None, device_tensors=None):
    """OPTIONAL exchange (the step path needs none): every rank receives a per-seat f64 field -- default 3 = payoffs -- of
    ALL tables of the job, [T_total, N] in global table order (ranks own contiguous blocks: shard_tables).

    dist: an initialised torch.distributed module (default: import it).  With the "nccl" backend -- RCCL on ROCm -- the
    local block is exported on the device (pk_get_f64_d) straight into the send tensor and all-gathered GPU to GPU over
    xGMI; with "gloo" the host arrays are exchanged.  Blocks may differ in size by one table (uneven shards): they are
    padded to the largest for the collective.  torch is imported here only: the package itself does not need it."""
    import numpy as np
    import torch
    if dist is None:
        import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    n = game.num_players
    counts = [None] * world
    dist.all_gather_object(counts, int(game.num_tables))
    use_gpu = dist.get_backend() == "nccl" if device_tensors is None else bool(device_tensors)
    cmax = max(counts)                     # all_gather wants equal blocks: pad to the largest shard, trim afterwards
    if use_gpu:
        from . import _lib as L
        dev = torch.device("cuda", game.device)
        # The send tensor is written by TWO streams: torch's current stream (the zero fill of the padding rows) and the
        # handle's own non-blocking stream (k_export_f64), which have no implicit order.  So: only the padding is filled, and
        # torch's stream is drained before the export is queued -- a fill landing after (or during) the export would zero
        # the payoffs (round 3's torch.zeros over the whole tensor could) -- and the handle's stream before the collective.
        local = torch.empty((cmax, n), dtype=torch.float64, device=dev)
        if cmax > game.num_tables:
            local[game.num_tables:].zero_()
        torch.cuda.current_stream(dev).synchronize()
        L.check(game._lib.pk_get_f64_d(game._h, int(field), L.C.c_void_p(local.data_ptr())), game._h)
        game.sync()
    else:
        dev = torch.device("cpu")
        local = torch.zeros((cmax, n), dtype=torch.float64)
        local[:game.num_tables] = torch.from_numpy(np.ascontiguousarray(game._f64(int(field))))
    parts = [torch.empty((cmax, n), dtype=torch.float64, device=dev) for _ in counts]
    dist.all_gather(parts, local)
    return torch.cat([p[:c] for p, c in zip(parts, counts)]).cpu().numpy()
