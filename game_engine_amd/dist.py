"""Multi-GPU: rooms shard embarrassingly (one LangGraph thread per room in the reference,
src/app/api/copilotkit/route.ts:24-37: rooms never interact), so there is NO collective on the
step path.  One process per GPU; each owns a contiguous range of global room indices, and the
RNG is keyed by the global index, so results do not depend on the number of GPUs.  The only
exchange is one all-gather of the fixed-size per-GPU summary (RCCL over xGMI with backend
"nccl"; gloo on CPU in the tests), after which every rank holds the whole-job summary."""
from __future__ import annotations

from typing import Any, Dict, List, Tuple

import numpy as np

from ._lib import SUMMARY_WORDS
from .stepper import summary_to_dict

# indices inside ge_summary (u64 words): every word is a sum over rooms except `turn`
_TURN_WORD = 39


def shard_range(total_rooms: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of global room indices owned by `rank`: GPU g owns
    [g*R/G, (g+1)*R/G)."""
    return (rank * total_rooms) // world, ((rank + 1) * total_rooms) // world


def shard_first_room(rooms_per_rank: int, rank: int) -> int:
    """Weak scaling: every rank owns `rooms_per_rank` rooms; global index of its room 0."""
    return rank * rooms_per_rank


def reduce_summaries(words: np.ndarray) -> np.ndarray:
    """[world, SUMMARY_WORDS] u64 -> whole-job summary words (sums wrap mod 2^64 like the
    device-side sums; `turn` is the same on all ranks)."""
    words = np.asarray(words, dtype=np.uint64).reshape(-1, SUMMARY_WORDS)
    out = words.sum(axis=0, dtype=np.uint64)
    out[_TURN_WORD] = words[0, _TURN_WORD]
    return out


def allgather_summary_words(local_words: np.ndarray, world: int, device=None, force: bool = False) -> np.ndarray:
    """All-gather of one rank's summary words; returns [world, SUMMARY_WORDS].
    `force`: run the collective even for a single rank (rehearsal of the RCCL path)."""
    local_words = np.asarray(local_words, dtype=np.uint64)
    if world == 1 and not force:
        return local_words.reshape(1, -1)
    import torch
    import torch.distributed as dist
    # int64 view: collectives have no uint64; the bits are what travels
    t = torch.from_numpy(local_words.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = torch.empty(world * SUMMARY_WORDS, dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy().view(np.uint64).reshape(world, SUMMARY_WORDS)


def allgather_summary(batch, world: int, force: bool = False) -> Dict[str, Any]:
    """Whole-job summary on every rank: device-side reduction of this rank's rooms, then the
    single all-gather of the path."""
    device = None
    if world > 1 or force:
        import torch
        import torch.distributed as dist
        if dist.get_backend() == "nccl":
            device = torch.device("cuda", torch.cuda.current_device())
    gathered = allgather_summary_words(batch.summary_words(), world, device, force)
    return summary_to_dict(reduce_summaries(gathered))
