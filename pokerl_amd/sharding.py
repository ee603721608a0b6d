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
