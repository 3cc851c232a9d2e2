"""Multi-GPU plumbing for the first-dimension shard.

The database is partitioned by first-dimension index j across the ranks of one node; every rank sweeps
its shard and produces full-shape partial accumulators (packed words, each 28-bit field already reduced
mod its prime).  One sum-reduce of those words as integers is carry-safe for up to 16 ranks
(16 * 2^28 = 2^32), after which the root reduces each field mod its prime and continues with the
INTT / CRT lift and the folding.  torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo"
on CPU) carries the single collective; nothing else crosses ranks.
"""
from __future__ import annotations

MAX_RANKS = 16  # carry-safety bound of the packed sum


def shard_range(rank: int, world: int, dim0: int) -> tuple[int, int]:
    """contiguous j-range [j0, j1) of `rank`; dim0 and world are powers of two in every Spiral geometry"""
    if world < 1 or world > MAX_RANKS:
        raise ValueError(f"world size {world} outside [1, {MAX_RANKS}] (packed-sum carry bound)")
    if dim0 % world != 0:
        raise ValueError(f"first dimension {dim0} does not split evenly over {world} ranks")
    per = dim0 // world
    return rank * per, (rank + 1) * per


def _check(name, t, numel=None):
    """what every collective here assumes of its tensors: int64 words (the packed fields are summed as integers), contiguous, of the
    expected size -- a wrong view would not fail inside RCCL, it would silently reduce the wrong bytes (or hang on a size mismatch)"""
    import torch

    if t.dtype != torch.int64 or not t.is_contiguous() or (numel is not None and t.numel() != numel):
        raise ValueError(f"{name}: expected a contiguous int64 tensor" + (f" of {numel} words" if numel is not None else "") + f", got {t.dtype}, {tuple(t.shape)}, contiguous={t.is_contiguous()}")


def reduce_accumulators(acc, dst: int = 0, group=None):
    """sum the ranks' packed accumulators into rank `dst` (one collective).  `acc` is an int64 tensor
    viewing the words the sweep wrote; the fields never carry into each other (see module docstring)."""
    import torch.distributed as dist

    _check("reduce_accumulators: acc", acc)
    dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return acc


def reduce_scatter_accumulators(chunk, acc, group=None, async_op=False):
    """one collective: every rank receives the element-wise sum of its contiguous chunk of the ranks' accumulator
    buffers (the sweep wrote them grouped by ii mod world, Server.set_fold_ranks).  RCCL has a native reduce-scatter;
    gloo (CPU tests) does not, there the same result is an all-reduce followed by a slice."""
    import torch.distributed as dist

    _check("reduce_scatter_accumulators: acc", acc)
    _check("reduce_scatter_accumulators: chunk", chunk, acc.numel() // dist.get_world_size(group))
    if dist.get_backend(group) == "gloo":
        tmp = acc.clone()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
        n = chunk.numel()
        chunk.copy_(tmp[dist.get_rank(group) * n:(dist.get_rank(group) + 1) * n])
        return None if async_op else chunk
    work = dist.reduce_scatter_tensor(chunk, acc, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if async_op else chunk


def acc_position(ii: int, n_ranks: int, num_per: int, n_stages: int = 1) -> int:
    """where the sweep writes ciphertext ii = g + G k of the accumulator buffer (sweep.hip acc_pos): [stage s][rank g][k' < Ls] with
    k = s Ls + k', Ls = num_per / (G K).  One stage = grouped by rank (set_fold_ranks); K stages = every stage a contiguous 1/K of the
    buffer that is reduce-scattered on its own while the next stage sweeps (set_sweep_stages)."""
    g, k = ii % n_ranks, ii // n_ranks
    ls = num_per // (n_ranks * n_stages)
    return ((k // ls) * n_ranks + g) * ls + k % ls


def reduce_scatter_stages(chunk, acc, n_stages: int, group=None, async_op=False, after_stage=None):
    """the pipelined form: stage s's contiguous 1/K of `acc` is reduce-scattered into rows [s L/K, (s+1) L/K) of `chunk`.
    after_stage(s), when given, is called before stage s's collective is issued (the caller launches the sweep of stage s there);
    async_op: returns the Work handles (None entries on gloo, where the collective is synchronous)."""
    if acc.numel() % n_stages or chunk.numel() % n_stages:
        raise ValueError(f"reduce_scatter_stages: {acc.numel()} / {chunk.numel()} words do not split into {n_stages} stages")
    al, cl = acc.numel() // n_stages, chunk.numel() // n_stages
    works = []
    for s in range(n_stages):
        if after_stage is not None:
            after_stage(s)
        works.append(reduce_scatter_accumulators(chunk[s * cl:(s + 1) * cl], acc[s * al:(s + 1) * al], group=group, async_op=async_op))
    return works if async_op else chunk


def all_gather_cts(gathered, ct, group=None):
    """collect the ranks' locally folded ciphertexts in rank order (96 KiB each)"""
    import torch.distributed as dist

    _check("all_gather_cts: ct", ct)
    _check("all_gather_cts: gathered", gathered, ct.numel() * dist.get_world_size(group))
    dist.all_gather_into_tensor(gathered, ct, group=group)
    return gathered


def instances_of_rank(rank: int, world: int, factor: int) -> list[int]:
    """factor-sharded items (an item = `factor` database instances, select_params.py:297-298): rank r holds instances r, r + world, ...; the
    instances are independent -- one query, expanded and converted on every rank, no reduce -- and their responses are gathered"""
    return list(range(rank, factor, world))


def all_gather_instance_responses(gathered, mine, group=None):
    """collect every rank's block of instance responses in rank order: `mine` holds ceil(factor / world) responses of 6 x 2048 words (the ranks
    with one instance fewer leave their last slot unused), `gathered` world such blocks; instance k = slot k // world of rank k % world.  The one
    collective of the factor-sharded path, 96 KiB per instance"""
    import torch.distributed as dist

    _check("all_gather_instance_responses: mine", mine)
    _check("all_gather_instance_responses: gathered", gathered, mine.numel() * dist.get_world_size(group))
    dist.all_gather_into_tensor(gathered, mine, group=group)
    return gathered


def instance_response(gathered, k: int, world: int, slots: int, words: int = 6 * 2048):
    """instance k's response inside the gathered blocks ([world][slots][words])"""
    r, sl = k % world, k // world
    off = (r * slots + sl) * words
    return gathered[off:off + words]


def fold_ranks(world: int, num_per: int) -> int:
    """ranks taking part in the distributed fold: all of them when they divide num_per, else 1 (root folds alone)"""
    return world if world >= 1 and (world & (world - 1)) == 0 and world <= num_per else 1


def all_gather_gsw_bits(gathered, mine, group=None, async_op=False):
    """sharded expansion: collect the ranks' blocks of GSW-bit ciphertexts in rank order (1.8 MiB in all at config 2).
    async_op: returns the Work handle; the collective then runs on the backend's own stream (ordered after what the
    current stream has enqueued so far) and work.wait() makes the current stream wait for it"""
    import torch.distributed as dist

    _check("all_gather_gsw_bits: mine", mine)
    _check("all_gather_gsw_bits: gathered", gathered, mine.numel() * dist.get_world_size(group))
    work = dist.all_gather_into_tensor(gathered, mine, group=group, async_op=async_op)
    return work if async_op else gathered


def expand_shard_ok(shape, params, world: int) -> bool:
    """can the expansion be sharded over `world` ranks: query compression with the reordered layout, power-of-two ranks"""
    return world >= 1 and (world & (world - 1)) == 0 and not params.direct_upload and shape.stopround > 0 and world <= shape.dim0
