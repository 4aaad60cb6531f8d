"""Image tiling across ranks: layout arithmetic + the single gather.

The image is cut into 8x8-pixel tiles (one per 64-lane wavefront).  Tile ``t``
(row-major over ceil(W/8) x ceil(H/8)) belongs to rank ``t % n_ranks`` and is that
rank's local tile ``t // n_ranks`` -- interleaved, so sky/ground imbalance averages
out (the reference's contiguous row blocks, Camera.txt:96-100, do not).  Each rank
renders into a compact buffer ``[tiles_per_rank][3][64]``; one gather to rank 0
over RCCL (xGMI) collects them, and ``rtk_tiles_unpermute`` scatters to row-major.
No other collective exists on the path: every (pixel, sample) is independent.

The numpy functions restate the layout for tests (CPU, gloo); the device path is
``Renderer.render_device(..., n_ranks>1)`` + ``gather_to_root`` + ``Renderer.unpermute``.
"""
from __future__ import annotations

import numpy as np


def tile_grid(width: int, height: int):
    return (width + 7) // 8, (height + 7) // 8


def tiles_per_rank(width: int, height: int, n_ranks: int) -> int:
    tx, ty = tile_grid(width, height)
    return (tx * ty + n_ranks - 1) // n_ranks


def compact_from_image(image: np.ndarray, rank: int, n_ranks: int) -> np.ndarray:
    """What rank `rank` writes for a full row-major image HxWx3: [tiles_per_rank, 3, 64]."""
    h, w, _ = image.shape
    tx, ty = tile_grid(w, h)
    tpr = tiles_per_rank(w, h, n_ranks)
    out = np.zeros((tpr, 3, 64), image.dtype)
    for local in range(tpr):
        t = local * n_ranks + rank
        if t >= tx * ty:
            continue
        x0, y0 = (t % tx) * 8, (t // tx) * 8
        for lane in range(64):
            i, j = x0 + (lane & 7), y0 + (lane >> 3)
            if i < w and j < h:
                out[local, :, lane] = image[j, i, :]
    return out


def image_from_gathered(gathered: np.ndarray, width: int, height: int, n_ranks: int) -> np.ndarray:
    """Inverse of compact_from_image over all ranks: gathered is [n_ranks, tiles_per_rank, 3, 64]."""
    tx, ty = tile_grid(width, height)
    img = np.zeros((height, width, 3), gathered.dtype)
    for t in range(tx * ty):
        rank, local = t % n_ranks, t // n_ranks
        x0, y0 = (t % tx) * 8, (t // tx) * 8
        for lane in range(64):
            i, j = x0 + (lane & 7), y0 + (lane >> 3)
            if i < width and j < height:
                img[j, i, :] = gathered[rank, local, :, lane]
    return img


def gather_to_root(local, n_ranks: int, rank: int, root: int = 0):
    """The one collective of the path: gather every rank's compact tile buffer to `root`.

    `local` is a torch tensor [tiles_per_rank, 3, 64] (CUDA tensor -> RCCL over xGMI;
    CPU tensor -> gloo in tests).  Returns [n_ranks, tiles_per_rank, 3, 64] on root,
    None elsewhere.
    """
    import torch
    import torch.distributed as dist

    if n_ranks == 1:
        return local.unsqueeze(0)
    if rank == root:
        out = torch.empty((n_ranks,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, list(out.unbind(0)), dst=root)
        return out
    dist.gather(local, None, dst=root)
    return None


class GatherPipeline:
    """Frame k's gather overlapped with frame k+1's rendering (N > 1 only).

    Each rank renders frame k into compact tile buffer ``k & 1`` (``next_buffer``); ``submit`` starts the gather of
    the buffer just rendered (asynchronously: RCCL runs it on its own stream once the render has finished) and only
    then completes the PREVIOUS frame -- waits for its gather and, on root, hands the gathered tiles to
    ``consume(gathered, slot)`` (the un-permute kernel).  The render of frame k+1 is therefore enqueued without
    waiting for gather k; a buffer is rewritten two frames later, after its gather has been waited for.  ``flush``
    completes the last frame.  Which frame is gathered when never changes a pixel.

    ``streams``: optionally one torch stream per slot (frame k is rendered on ``streams[k & 1]``): the gather of a
    frame is started, waited for and consumed under ITS stream, so that the other stream's next render never waits
    for it.  ``stage_to_host``: gather host copies (gloo rehearsal of CUDA buffers on one GPU, CPU tests)."""

    def __init__(self, n_ranks: int, rank: int, make_buffer, consume, root: int = 0, stage_to_host: bool = False, streams=None):
        self.n_ranks, self.rank, self.root = n_ranks, rank, root
        self.buffers = [make_buffer(), make_buffer()]
        self.consume = consume
        self.stage_to_host = stage_to_host
        self.streams = streams
        self.pending = None
        self.frame = 0

    def slot(self) -> int:
        return self.frame & 1

    def next_buffer(self):
        return self.buffers[self.frame & 1]

    def _on_stream(self, slot: int):
        import contextlib

        import torch

        if self.streams is None or self.streams[slot] is None:
            return contextlib.nullcontext()
        return torch.cuda.stream(self.streams[slot])

    def submit(self):
        import torch
        import torch.distributed as dist

        slot = self.frame & 1
        with self._on_stream(slot):
            local = self.buffers[slot]
            if self.stage_to_host:
                local = local.cpu()
            out = None
            if self.rank == self.root:
                out = torch.empty((self.n_ranks,) + tuple(local.shape), dtype=local.dtype, device=local.device)
                work = dist.gather(local, list(out.unbind(0)), dst=self.root, async_op=True)
            else:
                work = dist.gather(local, None, dst=self.root, async_op=True)
        previous, self.pending = self.pending, (out, work, local, slot)
        self.frame += 1
        self._finish(previous)

    def flush(self):
        previous, self.pending = self.pending, None
        self._finish(previous)

    def _finish(self, entry):
        if entry is None:
            return
        out, work, _keep_alive, slot = entry
        with self._on_stream(slot):
            work.wait()  # device tensors: the slot's stream waits for the collective; host tensors: this thread does
            if self.rank == self.root:
                self.consume(out, slot)
