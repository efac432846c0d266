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
    # chunks of equal length AND equal context share a launch sequence, which then skips the conv-decoder rows under the
    # context's dropped samples - the same launches a single GPU's chunked_decode makes
    groups = {}
    for c in mine:
        groups.setdefault((plan[c][1] - plan[c][0], plan[c][2]), []).append(c)
    pieces = {}
    skip_ok = getattr(decoder, "skip_context", False)
    for (_length, ctx), cs in groups.items():
        batch = torch.stack([codes[:, plan[c][0]:plan[c][1]] for c in cs])
        wav = decoder.forward(batch, skip_frames=ctx) if (skip_ok and ctx) else decoder.forward(batch)
        for n, c in enumerate(cs):
            pieces[c] = wav[n, 0, ctx * up:]
    local = [pieces[c] for c in mine]
    out = gather_waveforms(local, mine, len(plan), dst=dst, group=group)
    return None if out is None else np.concatenate(out)


# ---- streams over the GPUs of one node ----------------------------------------------------------------------------------
class StreamRouter:
    """Which rank owns which stream.  A stream is placed once, when it opens, on the rank with the fewest open streams (lowest
    rank on ties), and never migrates: its causal state lives in that GPU's memory.  Pure bookkeeping, deterministic, so
    every rank that sees the same open / close sequence holds the same table and no message is needed to agree on it."""

    def __init__(self, world_size: int, capacity_per_rank: int):
        self.world, self.capacity = int(world_size), int(capacity_per_rank)
        self.load = [0] * self.world
        self.owner = {}
        self._next = 0

    def open(self) -> int:
        r = min(range(self.world), key=lambda j: (self.load[j], j))
        if self.load[r] >= self.capacity:
            raise RuntimeError(f"every rank's stream pool is full ({self.world} x {self.capacity} slots)")
        gid = self._next
        self._next += 1
        self.owner[gid] = r
        self.load[r] += 1
        return gid

    def rank_of(self, gid: int) -> int:
        return self.owner[gid]

    def close(self, gid: int) -> None:
        self.load[self.owner.pop(gid)] -= 1


class ShardedStreams:
    """Streams of one node, one `StreamPool` per rank.  Every rank calls open / close / push with the SAME arguments (the
    front end broadcasts the tick's work list, or every rank reads it from the same queue); a rank only runs the streams it
    owns, with no collective on the data path.  `push_gather` brings a tick's samples to one rank (or all) with the same
    point-to-point gather the bulk decode uses."""

    def __init__(self, pool, group=None):
        self.pool, self.group = pool, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.router = StreamRouter(self.world, pool.capacity)
        self._local = {}                  # global stream id -> slot in this rank's pool

    def open(self) -> int:
        gid = self.router.open()
        if self.router.rank_of(gid) == self.rank:
            self._local[gid] = self.pool.open()
        return gid

    def close(self, gid: int) -> None:
        if self.router.rank_of(gid) == self.rank:
            self.pool.close_stream(self._local.pop(gid))
        self.router.close(gid)

    def push(self, gids: Sequence[int], codes: torch.Tensor):
        """codes [S, Q, n] for the streams `gids` -> (positions in `gids` this rank owns, their new samples)."""
        mine = [k for k, g in enumerate(gids) if self.router.rank_of(g) == self.rank]
        if not mine:
            return mine, []
        return mine, self.pool.push([self._local[gids[k]] for k in mine], codes[mine])

    def push_gather(self, gids: Sequence[int], codes: torch.Tensor, dst: Optional[int] = 0) -> Optional[List[np.ndarray]]:
        mine, outs = self.push(gids, codes)
        return gather_waveforms(outs, mine, len(gids), dst=dst, group=self.group)
