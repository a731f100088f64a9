"""Pair-level data parallelism: one process per GPU, pairs sharded by rank, and a single exchange
step -- the gather of the (ragged) match arrays over RCCL/xGMI (gloo on CPU in the tests).
The reference is single-device (model_helper.py:30); pairs are independent, so no collective is
needed on the data path itself."""
import torch
import torch.distributed as dist


def shard_pairs(num_pairs, rank, world):
    """Pair ids handled by `rank` (round-robin, deterministic, covers every id exactly once)."""
    return list(range(rank, num_pairs, world))


def pack_results(results, rank, world, pairs_per_step):
    """What a rank contributes to the final gather: `results` = its steps, each a (fine, score, coarse) triple of
    per-pair lists ([n,4] fp32, [n] fp32, [n,4] int64).  Returns (rows [M,9] fp32 = fine, score, coarse; ids [M] int64);
    the global id of pair b of step i of rank r is (i * world + r) * pairs_per_step + b, i.e. steps are dealt
    round-robin over the ranks like `shard_pairs` deals pairs."""
    rows, ids = [], []
    for i, (fine, score, coarse) in enumerate(results):
        for b in range(pairs_per_step):
            rows.append(torch.cat([fine[b], score[b][:, None], coarse[b].to(fine[b].dtype)], dim=1))
            ids.append(torch.full((fine[b].shape[0],), (i * world + rank) * pairs_per_step + b, dtype=torch.int64,
                                  device=fine[b].device))
    if not rows:
        return torch.zeros((0, 9)), torch.zeros((0,), dtype=torch.int64)
    return torch.cat(rows), torch.cat(ids)


def gather_matches(rows, pair_ids, group=None):
    """All-gather ragged per-rank results.

    rows:     [M, C] float tensor (e.g. C = 9: fine x1,y1,x2,y2, score, coarse x1,y1,x2,y2)
    pair_ids: [M] int64 tensor, the pair each row belongs to
    Returns (rows_all [sum M, C], pair_ids_all [sum M]) ordered by rank then local order, on every rank.
    Three small collectives: counts (int64 [1] per rank), then one padded all_gather of the float payload and one of
    the int64 pair ids (ids travel as integers, never through a float).
    """
    if not (dist.is_available() and dist.is_initialized()):
        return rows, pair_ids
    world = dist.get_world_size(group)
    dev = rows.device
    count = torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    c = rows.shape[1]
    payload = torch.zeros((cap, c), dtype=rows.dtype, device=dev)
    payload[:rows.shape[0]] = rows
    ids = torch.zeros((cap,), dtype=torch.int64, device=dev)
    ids[:rows.shape[0]] = pair_ids.to(torch.int64)
    gathered = [torch.empty_like(payload) for _ in range(world)]
    gathered_ids = [torch.empty_like(ids) for _ in range(world)]
    dist.all_gather(gathered, payload, group=group)
    dist.all_gather(gathered_ids, ids, group=group)
    return (torch.cat([g[:n] for g, n in zip(gathered, counts)]),
            torch.cat([g[:n] for g, n in zip(gathered_ids, counts)]))
