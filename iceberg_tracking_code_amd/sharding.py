"""Segment sharding across the GPUs of one node (one process per GPU) and the final gather.

The independent unit of the reference loop is a SEGMENT: `track_len + 1` consecutive frames starting at a
detection frame (`counter % track_len == 0`, s1_lucaskanade_tracking.py:362; tracks are reset there, s1:440).
Frame pairs inside a segment are sequentially dependent (p1 of pair i is p0 of pair i+1), segments are not, so a
rank takes a contiguous block of segments and needs one extra frame at the end of its block.  There is no
collective on the data path; the only exchange is the gather of the per-segment track counts at the end
(`torch.distributed` all_gather: RCCL over xGMI with backend "nccl", gloo in the CPU tests).
"""
import numpy as np


def segment_count(n_frames, track_len):
    """Completed segments in a sequence of n_frames (the loop saves a segment at every detection frame > 0)."""
    if n_frames < track_len + 1:
        return 0
    return (n_frames - 1) // track_len


def segment_block(n_segments, rank, world):
    """[first, last) segment indices of `rank`: contiguous blocks, sizes differing by at most one."""
    base, extra = divmod(n_segments, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def frame_block(n_frames, track_len, rank, world):
    """[first, last) FRAME indices rank needs: its segments plus the closing frame of its last segment."""
    s0, s1 = segment_block(segment_count(n_frames, track_len), rank, world)
    if s1 <= s0:
        return 0, 0
    return s0 * track_len, s1 * track_len + 1


def gather_counts(local_counts, dist=None, device=None, group=None):
    """All ranks' per-segment track counts, concatenated in rank order (int64 numpy array).

    `dist` is torch.distributed (initialised) or None for a single process; `group` selects the process group
    (e.g. an RCCL group created beside a gloo control group).  Blocks are padded to the largest block so that one
    all_gather suffices; messages are a few hundred bytes, i.e. latency bound on any fabric.
    """
    local = np.asarray(local_counts, np.int64).ravel()
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local.copy()
    import torch
    world = dist.get_world_size(group)
    n = torch.tensor([local.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    pad = max(max(sizes), 1)
    mine = torch.zeros(pad, dtype=torch.int64, device=device)
    if local.size:
        mine[:local.size] = torch.from_numpy(local).to(mine.device)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return np.concatenate([p[:k].cpu().numpy() for p, k in zip(parts, sizes)]) if sum(sizes) else np.zeros(0, np.int64)


def gather_tables(tracks, counts, dist=None, group=None, root=0):
    """The per-segment track tables (the `tracks` arrays of s1:394-395) of all ranks, in rank order, on rank `root`.

    tracks: torch tensor (S, R, V, 2) float32 -- S segments of this rank, each padded to R rows, the first counts[s]
    rows valid; counts: torch tensor (S,) int32/int64 on the same device.  Every rank passes the same V; S may differ
    by one (segment_block).  One all_gather of the per-segment counts (a few hundred bytes: every rank learns how many
    rows the longest segment of the run has), then ONE gather to `root` of tables cut to that many rows -- not the R the
    buffers were sized for, and not to every rank: at BASELINE.json configs[3]'s length (1 350 segments of 10 000 x 3 x 2
    floats per rank) an all_gather of the padded tables would put 8 x 324 MB on every GPU and copy all of it to every
    host.  With backend "nccl" both are RCCL over xGMI, and the only exchange of a sharded run.
    `root` is a GLOBAL rank (as torch.distributed.gather's dst) and must belong to `group`; the sizes follow the group, so
    a group that is not the whole world works.  Returns on `root` a list with one (first_rows, n) numpy pair per segment of
    the group's part of the sequence [(tracks[:n], n), ...], in group-rank order; on the other ranks None.
    """
    import torch
    S = int(tracks.shape[0])
    world = 1 if dist is None or not dist.is_initialized() else dist.get_world_size(group)
    if world == 1:
        c = counts.cpu().numpy().astype(np.int64)
        t = tracks.cpu().numpy()
        return [(t[s, :c[s]].copy(), int(c[s])) for s in range(S)]
    dev = tracks.device
    rank = dist.get_rank()            # global ranks: `root` and gather's dst are global, also inside a sub-group
    n_seg = torch.tensor([S], dtype=torch.int64, device=dev)
    all_seg = [torch.zeros_like(n_seg) for _ in range(world)]
    dist.all_gather(all_seg, n_seg, group=group)
    all_seg = [int(v.item()) for v in all_seg]
    pad = max(max(all_seg), 1)
    cpad = torch.zeros(pad, dtype=torch.int64, device=dev)
    cpad[:S] = counts.to(torch.int64)
    call = [torch.zeros_like(cpad) for _ in range(world)]
    dist.all_gather(call, cpad, group=group)
    call = [c.cpu().numpy() for c in call]
    rows = max(int(max(int(c.max()) for c in call)), 1)       # the longest segment of the whole run
    if rows > int(tracks.shape[1]):
        raise ValueError("a segment count exceeds the rows of the table")
    tpad = torch.zeros((pad, rows) + tuple(tracks.shape[2:]), dtype=tracks.dtype, device=dev)
    tpad[:S] = tracks[:, :rows]
    tall = [torch.zeros_like(tpad) for _ in range(world)] if rank == root else None
    dist.gather(tpad, tall, dst=root, group=group)
    if rank != root:
        return None
    out = []
    for r in range(world):
        t = tall[r].cpu().numpy()
        for s in range(all_seg[r]):
            out.append((t[s, :call[r][s]].copy(), int(call[r][s])))
    return out
