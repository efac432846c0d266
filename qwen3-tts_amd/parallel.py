"""Data-parallel sharding of utterances over the GPUs of one node.

The decode has no inter-GPU traffic: utterances (and the 300-frame chunks of one long utterance) are
independent, weights are replicated.  `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the
CPU tests) is used once per job, to gather the waveforms.  The reference serves one request at a time on one GPU
(reference config.py:12, docker-compose.yml:17-23) and has no counterpart of this module.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .decoder import to_host


def lpt_assign(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first: item indices per rank, so that the summed frames per rank are balanced.
    Deterministic (ties broken by index), identical on every rank."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world_size
    shards: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda j: (loads[j], j))
        shards[r].append(i)
        loads[r] += int(lengths[i])
    return shards


def chunk_assign(num_chunks: int, world_size: int) -> List[List[int]]:
    """Long-form decode: chunk c of one utterance goes to rank c mod world_size."""
    return [list(range(r, num_chunks, world_size)) for r in range(world_size)]


def gather_waveforms(local: Sequence[torch.Tensor], local_ids: Sequence[int], total_items: int, dst: Optional[int] = 0,
                     group=None) -> Optional[List[np.ndarray]]:
    """Collect variable-length 1-D float32 waveforms from every rank.

    Each rank passes the waveforms it decoded (`local`, tensors on its own device) and their global item indices.
    Lengths travel first (one all_gather of int64), then every rank's waveforms as ONE flat buffer: with dst = r the
    root receives each peer's buffer directly (point-to-point, one xGMI link per peer, no ring); with dst = None
    every rank receives everything (padded all_gather).  Returns the waveforms in global item order on the
    destination rank(s), None elsewhere."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    device = local[0].device if len(local) else torch.device("cpu")
    if dist.get_backend(group) == "nccl" and device.type != "cuda":
        device = torch.device("cuda", torch.cuda.current_device())
    meta = torch.full((total_items, 2), -1, dtype=torch.int64, device=device)      # (owner rank, length) per item
    for t, i in zip(local, local_ids):
        meta[i, 0] = rank
        meta[i, 1] = t.numel()
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    owner = torch.stack([m[:, 0] for m in metas]).max(0).values.cpu().tolist()
    length = torch.stack([m[:, 1] for m in metas]).max(0).values.cpu().tolist()
    if any(o < 0 for o in owner):
        raise RuntimeError("gather_waveforms: some item was decoded by no rank")
    per_rank_items = [[i for i in range(total_items) if owner[i] == r] for r in range(world)]
    per_rank_len = [sum(length[i] for i in items) for items in per_rank_items]
    # this rank's waveforms as one flat buffer, in global item order (the order `split` below reads them back in)
    order = sorted(range(len(local_ids)), key=lambda j: local_ids[j])
    flat = (torch.cat([local[j].reshape(-1).to(torch.float32) for j in order]) if order
            else torch.empty(0, dtype=torch.float32)).to(device)

    out: List[Optional[np.ndarray]] = [None] * total_items

    def land(r: int, buf: torch.Tensor) -> None:
        """Rank r's flat buffer -> host (pinned double buffer on a GPU) and its utterances as views into that one array."""
        host = to_host(buf)
        off = 0
        for i in per_rank_items[r]:
            out[i] = host[off:off + length[i]]
            off += length[i]

    def split(bufs: List[torch.Tensor]) -> List[np.ndarray]:
        for r in range(world):
            land(r, bufs[r])
        return out  # type: ignore[return-value]

    if dst is None:
        pad = max(per_rank_len) if per_rank_len else 0
        mine = torch.zeros(pad, dtype=torch.float32, device=device)
        mine[: flat.numel()] = flat
        bufs = [torch.empty(pad, dtype=torch.float32, device=device) for _ in range(world)]
        dist.all_gather(bufs, mine, group=group)
        return split([b[:n] for b, n in zip(bufs, per_rank_len)])
    if rank == dst:
        # every peer's receive is posted first (each arrives over its own link); the root's own samples go to the host while
        # the peers are still sending, then each peer's buffer as soon as its receive has completed
        bufs = [flat if r == dst else torch.empty(per_rank_len[r], dtype=torch.float32, device=device) for r in range(world)]
        reqs = {r: dist.irecv(bufs[r], src=r, group=group) for r in range(world) if r != dst and per_rank_len[r] > 0}
        land(dst, flat)
        for r in range(world):
            if r == dst:
                continue
            if r in reqs:
                reqs[r].wait()
                if bufs[r].is_cuda:
                    torch.cuda.current_stream(bufs[r].device).synchronize()      # the receive is stream-ordered on RCCL
            land(r, bufs[r])
        return out  # type: ignore[return-value]
    if flat.numel() > 0:
        dist.send(flat, dst=dst, group=group)
    return None


def decode_sharded(decoder, items: Sequence[torch.Tensor], dst: Optional[int] = 0, group=None) -> Optional[Tuple[List[np.ndarray], int]]:
    """Every rank calls this with the SAME list of per-utterance code matrices; rank r decodes its LPT shard with its
    own `decoder` and the waveforms are gathered to `dst` (all ranks when dst is None)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    Q = decoder.cfg.num_quantizers
    lengths = [int(c.shape[0] if c.shape[-1] == Q else c.shape[-1]) for c in items]
    mine = lpt_assign(lengths, world)[rank]
    sr = decoder.cfg.sample_rate
    if not mine:
        local = []
    elif hasattr(decoder, "decode_device"):          # samples stay on the GPU until the gather
        flat, starts, lens = decoder.decode_device([items[i] for i in mine])
        local = [flat[s:s + n] for s, n in zip(starts, lens)]
    else:
        wavs, sr = decoder.decode([items[i] for i in mine])
        local = [torch.from_numpy(w) for w in wavs]
    out = gather_waveforms(local, mine, len(items), dst=dst, group=group)
    return None if out is None else (out, sr)


def decode_long_sharded(decoder, codes: torch.Tensor, chunk_size: int = 300, left_context_size: int = 25, dst: Optional[int] = 0,
                        group=None) -> Optional[torch.Tensor]:
    """One long utterance, codes [Q, T]: the chunks of chunked_decode are independent (each recomputes its own left
    context), so chunk c goes to rank c mod world_size; every rank decodes its chunks as one batch and the pieces are
    gathered in order.  Returns the [samples] waveform (numpy) on the destination rank(s)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    cfg = decoder.cfg
    T = int(codes.shape[-1])
    plan = cfg.chunk_plan(T, chunk_size, left_context_size)
    mine = chunk_assign(len(plan), world)[rank]
    up = cfg.total_upsample
    local: List[torch.Tensor] = []
    by_len = {}
    for c in mine:
        by_len.setdefault(plan[c][1] - plan[c][0], []).append(c)
    pieces = {}
    for _length, cs in by_len.items():
        batch = torch.stack([codes[:, plan[c][0]:plan[c][1]] for c in cs])
        wav = decoder.forward(batch)
        for n, c in enumerate(cs):
            pieces[c] = wav[n, 0, plan[c][2] * up:]
    local = [pieces[c] for c in mine]
    out = gather_waveforms(local, mine, len(plan), dst=dst, group=group)
    return None if out is None else np.concatenate(out)
