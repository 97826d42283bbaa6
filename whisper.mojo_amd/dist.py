"""Batched-utterance data parallelism: one process per GPU, weights replicated, utterances sharded
contiguously, ONE collective per batch — an all-gather of the fixed-stride token buffers.  No intra-model
collectives exist on this path (each utterance depends only on its own mel and the read-only weights:
whisper.mojo:184-223 builds a fresh KVCache per call).

backend "nccl" (= RCCL over xGMI) on GPUs; "gloo" on CPU for the world_size-2 tests."""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [start, start+count) of `total` utterances for `rank`; the first total%world ranks get one
    extra (ragged batches are allowed)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def pack_tokens(tokens: np.ndarray, counts: np.ndarray, stride: int, rows: int) -> np.ndarray:
    """[rows, 1+stride] int32: column 0 = length, then the ids zero-padded (51 456 B per rank at 64 x 201)."""
    buf = np.zeros((rows, 1 + stride), np.int32)
    b = tokens.shape[0]
    buf[:b, 0] = counts
    w = min(stride, tokens.shape[1])
    buf[:b, 1:1 + w] = tokens[:, :w]
    buf[:b, 1:1 + w][np.arange(w)[None, :] >= np.asarray(counts)[:, None]] = 0  # what lies past an utterance's length is not its ids
    return buf


def unpack_tokens(buf: np.ndarray) -> List[List[int]]:
    return [row[1:1 + row[0]].tolist() for row in buf]


def init_from_env(backend: str | None = None):
    """torchrun contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def gather_tokens(tokens: np.ndarray, counts: np.ndarray, total: int, stride: int) -> List[List[int]]:
    """All ranks end with every utterance's ids, in utterance order.  `tokens` [b_local, >=stride], `counts`
    [b_local] are this rank's results for its shard_range(total, rank, world)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return unpack_tokens(pack_tokens(tokens, counts, stride, tokens.shape[0]))
    world, rank = dist.get_world_size(), dist.get_rank()
    rows = (total + world - 1) // world  # fixed rows per rank so a single all_gather_into_tensor works when ragged
    local = torch.from_numpy(pack_tokens(tokens, counts, stride, rows))
    on_gpu = dist.get_backend() == "nccl"
    if on_gpu:
        local = local.cuda()
    out = torch.empty((world * rows, 1 + stride), dtype=torch.int32, device=local.device)
    dist.all_gather_into_tensor(out, local)
    out = out.cpu().numpy()
    res: List[List[int]] = []
    for r in range(world):
        _, cnt = shard_range(total, r, world)
        res.extend(unpack_tokens(out[r * rows:r * rows + cnt]))
    return res


def gather_tokens_device(packed_local, total: int) -> List[List[int]]:
    """The same collective on DEVICE buffers (backend "nccl" = RCCL over xGMI): `packed_local` is this rank's [rows, 1 + stride]
    int32 CUDA tensor as Whisper.transcribe_wait_device leaves it (rows = ceil(total / world), zero rows past the shard) — no host
    round trip before the all-gather; ONE device-to-host copy of the gathered [world * rows, 1 + stride] buffer after it."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        world, out = 1, packed_local
    else:
        world = dist.get_world_size()  # (a world of one still runs the collective: the one-GPU rehearsal of the RCCL path)
        out = torch.empty((world * packed_local.shape[0], packed_local.shape[1]), dtype=torch.int32, device=packed_local.device)
        dist.all_gather_into_tensor(out, packed_local)
    rows = packed_local.shape[0]
    host = out.cpu().numpy()
    res: List[List[int]] = []
    for r in range(world):
        _, cnt = shard_range(total, r, world)
        res.extend(unpack_tokens(host[r * rows:r * rows + cnt]))
    return res


def transcribe_sharded(model, mels_for_rank, total: int, **kw) -> List[List[int]]:
    """model: a loaded whisper.Whisper on this rank's GPU; mels_for_rank: this rank's shard
    [count, n_mels, n_frames].  Returns ALL utterances' token lists on every rank."""
    res = model.transcribe_batch(mels_for_rank, **kw) if len(mels_for_rank) else []
    stride = model.last_tokens.shape[1] if len(mels_for_rank) else len(kw.get("prompt", (0,) * 4)) + 1 + kw.get("max_loop", 195)
    toks = model.last_tokens if len(mels_for_rank) else np.zeros((0, stride), np.int32)
    cnts = model.last_counts if len(mels_for_rank) else np.zeros(0, np.int32)
    del res
    return gather_tokens(toks, cnts, total, stride)
