"""Frame-batch sharding across ranks (one process per GPU, no data-path
collective): batch b of `batch` consecutive frames goes to rank b % world.
Frames never depend on each other, only on the stream-constant delta frame,
so every rank encodes its batches independently; the only serial step is the
frame index, which needs every chunk's size in frame order."""
import struct


def plan_batches(nframes, batch):
    """[(first_frame, count), ...] covering 0..nframes-1."""
    return [(f, min(batch, nframes - f)) for f in range(0, nframes, batch)]


def rank_batches(nbatches, world, rank):
    """Indices of the batches owned by `rank`."""
    return list(range(rank, nbatches, world))


def frames_of_rank(nframes, batch, world, rank):
    plan = plan_batches(nframes, batch)
    out = []
    for b in rank_batches(len(plan), world, rank):
        first, count = plan[b]
        out.extend(range(first, first + count))
    return out


def assemble_stream(header, chunks):
    """header bytes (file header + delta chunk) + frame chunks in frame order
    -> complete .fpv stream with its frame index (container layout: upstream
    fusion_power_video.cc:30-155, index cc:1185-1197)."""
    out = bytearray(header)
    offsets = []
    for c in chunks:
        offsets.append(len(out))
        out += c
    n = len(chunks)
    out += struct.pack("<IB", 5 + 8 * n + 8, 2)
    for o in offsets:
        out += struct.pack("<Q", o)
    out += struct.pack("<Q", n)
    return bytes(out)
