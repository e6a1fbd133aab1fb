"""Multi-GPU orchestration of the batch mode: one process per GPU, songs sharded by index,
one all-gather of the 16-byte force vectors (RCCL over xGMI when the backend is "nccl",
gloo on CPU in the tests), then every rank computes its own row block of the N x N
bl_distance matrix.  No other collective touches the data path: songs are independent
(ref src/analyze.c:33-86 keeps no cross-song state)."""
import heapq


def shard_range(total, rank, world):
    """Contiguous block of song indices [first, first + count) owned by `rank` when all
    songs cost the same (fixed-length corpora: BASELINE configs[1], configs[2])."""
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def lpt_shards(lengths, world):
    """Longest-processing-time-first assignment of songs to ranks by sample count
    (mixed-length corpora, BASELINE configs[4]).  Returns `world` lists of song indices,
    each list sorted by descending length so that the per-song kernels of one wave of the
    envelope tail see similar lengths.  Deterministic (ties broken by index)."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    heap = [(0, r) for r in range(world)]
    heapq.heapify(heap)
    shards = [[] for _ in range(world)]
    for i in order:
        load, r = heapq.heappop(heap)
        shards[r].append(i)
        heapq.heappush(heap, (load + int(lengths[i]), r))
    return shards


def gather_force_vectors(local_vecs, counts, group=None):
    """All-gather per-rank (count_r, 4) float32 tensors into the (sum counts, 4) tensor every
    rank needs for its rows.  Equal counts use all_gather_into_tensor (one RCCL call, 16 B x
    songs per rank — latency-bound on xGMI); ragged counts fall back to all_gather on padded
    blocks."""
    import torch
    import torch.distributed as dist

    world = len(counts)
    total = sum(counts)
    if not dist.is_initialized():  # plain single-process run
        return local_vecs.clone()
    out = torch.empty((total, 4), dtype=local_vecs.dtype, device=local_vecs.device)
    if len(set(counts)) == 1:
        dist.all_gather_into_tensor(out, local_vecs.contiguous(), group=group)
        return out
    m = max(counts)
    pad = torch.zeros((m, 4), dtype=local_vecs.dtype, device=local_vecs.device)
    pad[: local_vecs.shape[0]] = local_vecs
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    off = 0
    for r, c in enumerate(counts):
        out[off:off + c] = parts[r][:c]
        off += c
    return out
