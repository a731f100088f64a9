"""Pair-level data parallelism: one process per GPU, pairs sharded by rank, and a single exchange
step -- the gather of the (ragged) match arrays over RCCL/xGMI (gloo on CPU in the tests).
The reference is single-device (model_helper.py:30); pairs are independent, so no collective is
needed on the data path itself."""
import numpy as np
import torch
import torch.distributed as dist


def shard_pairs(num_pairs, rank, world):
    """Pair ids handled by `rank` (round-robin, deterministic, covers every id exactly once)."""
    return list(range(rank, num_pairs, world))


def pack_results(results, rank, world, pairs_per_step, device=None, dtype=torch.float32):
    """What a rank contributes to the final gather: `results` = its steps, each a (fine, score, coarse) triple of
    per-pair lists ([n,4] fp32, [n] fp32, [n,4] int64).  Returns (rows [M,9] fp32 = fine, score, coarse; ids [M] int64);
    the global id of pair b of step i of rank r is (i * world + r) * pairs_per_step + b, i.e. steps are dealt
    round-robin over the ranks like `shard_pairs` deals pairs.  `device` / `dtype`: where and as what an EMPTY contribution
    is created (a rank without results must still enter the collective with tensors on its GPU: under RCCL a CPU tensor
    on one rank against GPU tensors on the others errors or hangs)."""
    if any(len(step[0]) < pairs_per_step for step in results):
        raise ValueError(f"every step must hold {pairs_per_step} pairs")
    fine = [f for step in results for f in step[0][:pairs_per_step]]
    if not fine:
        return torch.zeros((0, 9), dtype=dtype, device=device), torch.zeros((0,), dtype=torch.int64, device=device)
    score = [t for step in results for t in step[1][:pairs_per_step]]
    coarse = [t for step in results for t in step[2][:pairs_per_step]]
    # one concatenation per column group over ALL pairs of all steps (five launches however many pairs there are: a
    # cat + full + convert per pair was 3 launches x 320 pairs at the end of a 20-step run), the ids from the host
    rows = torch.cat([torch.cat(fine), torch.cat(score)[:, None], torch.cat(coarse).to(fine[0].dtype)], dim=1)
    first = np.array([(i * world + rank) * pairs_per_step + b for i in range(len(results)) for b in range(pairs_per_step)],
                     dtype=np.int64)
    ids = torch.from_numpy(np.repeat(first, [int(f.shape[0]) for f in fine])).to(fine[0].device)
    return rows, ids


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


def stream_rounds(num_pairs, world, chunk, gather_every):
    """Number of exchange rounds of `run_pair_stream` -- a function of the global sizes only, so every rank enters the
    same number of collectives whatever its own share is."""
    per_rank = -(-num_pairs // world)
    return max(1, -(-(-(-per_rank // chunk)) // gather_every))


def run_pair_stream(num_pairs, rank, world, chunk, submit, finish, gather_every=8, device=None, group=None, on_chunk=None,
                    exchange=True):
    """BASELINE configs[3]: a stream of `num_pairs` independent pairs, pair `i` handled by rank `i % world`
    (`shard_pairs`), `chunk` pairs per submission, results exchanged every `gather_every` chunks with `gather_matches`
    (the only collective of the path).  Software-pipelined one chunk ahead: `submit(pair_ids) -> ticket` enqueues a
    chunk, `finish(ticket) -> (fine, score, coarse)` (per-pair lists) completes it, so the device never waits for the
    host between chunks.  Returns (rows [M,9], pair_ids [M]) of ALL ranks on every rank, concatenated over the rounds,
    and the number of pairs this rank processed.  exchange=False keeps the results local (no collective: a rank running
    on its own while a process group exists)."""
    mine = shard_pairs(num_pairs, rank, world)
    chunks = [mine[i:i + chunk] for i in range(0, len(mine), chunk)]
    rounds = stream_rounds(num_pairs, world, chunk, gather_every)
    out_rows, out_ids, done = [], [], 0
    ticket = submit(chunks[0]) if chunks else None
    nxt = 1
    for r in range(rounds):
        rows, ids = [], []
        for _ in range(gather_every):
            if ticket is None:
                break
            cur_ids = chunks[nxt - 1]
            following = submit(chunks[nxt]) if nxt < len(chunks) else None
            fine, score, coarse = finish(ticket)
            for j, pid in enumerate(cur_ids):
                rows.append(torch.cat([fine[j], score[j][:, None], coarse[j].to(fine[j].dtype)], dim=1))
                ids.append(torch.full((fine[j].shape[0],), pid, dtype=torch.int64, device=fine[j].device))
            done += len(cur_ids)
            if on_chunk is not None:
                on_chunk(cur_ids)
            ticket, nxt = following, nxt + 1
        if rows:
            packed = (torch.cat(rows), torch.cat(ids))
        else:
            packed = (torch.zeros((0, 9), device=device), torch.zeros((0,), dtype=torch.int64, device=device))
        g_rows, g_ids = gather_matches(*packed, group=group) if exchange else packed
        out_rows.append(g_rows)
        out_ids.append(g_ids)
    assert ticket is None, "stream_rounds under-counted the rounds"
    return torch.cat(out_rows), torch.cat(out_ids), done
