"""
Frame sharding for multi-GPU runs: frames are independent given `reservoir_in`, so a batch
is cut into contiguous frame ranges, one per rank, and each rank reads its own slice of the hop-overlapped
stream plus a one-hop halo (the `priorBlock` of its first frame, pacfileThem.py:628-631).  No collective
touches the data path; torch.distributed is only used for the barrier and the max-over-ranks of a timing.
"""


def shard_frames(n_frames, world, rank):
    """Contiguous, balanced ranges: returns (first_frame, n_local).  The first `n_frames % world` ranks get one more."""
    if world < 1 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_frames, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def shard_streams(n_streams, world, rank):
    """Chained (stream) mode across ranks, SURVEY.md 8(e): "with many streams, shard by stream" -- a stream's blocks are
    serial through its bit reservoir, so WHOLE streams go to a rank: contiguous, balanced ranges like shard_frames;
    -> (first_stream, n_local).  No data crosses ranks."""
    return shard_frames(n_streams, world, rank)


def encode_streams_sharded(handle, streams, shapes, world, rank, use_huffman=True, num_samples=None):
    """This rank's share of pacfile.encode_stereo_streams: streams [nStreams][2][samples] and one shape list per stream as
    EVERY rank sees them (or can regenerate them); the rank encodes streams [first, first + n) on its own GPU with one
    chained call and returns (first, list of .pac byte strings).  Concatenated over the ranks in rank order the lists are
    what one process returns for all streams."""
    from . import pacfile
    first, n = shard_streams(len(shapes), world, rank)
    if n == 0:
        return first, []
    ns = None if num_samples is None else list(num_samples[first:first + n])
    return first, pacfile.encode_stereo_streams(handle, streams[first:first + n], shapes[first:first + n], use_huffman, ns)


def shard_samples(first_frame, n_local, hop):
    """Sample range [start, stop) of the stream that frames first_frame .. first_frame+n_local-1 read:
    frame f covers [f*hop, f*hop + 2*hop), so the slice carries one hop of halo in front of its new hops."""
    if n_local == 0:
        return first_frame * hop, first_frame * hop
    return first_frame * hop, (first_frame + n_local + 1) * hop


def max_over_ranks(value, device=None):
    """Max of a Python float over all ranks (identity when torch.distributed is not initialised)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
