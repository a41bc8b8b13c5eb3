"""
The N > 1 path on CPU: world_size-2 gloo processes shard one stream by contiguous frame ranges with a
one-hop halo (mrcaudiocodec_amd.shard), encode their shard independently and the concatenation equals the
unsharded result -- no collective on the data path.  There is no GPU here, so the per-shard worker is the
oracle (the checker); what is under test is the sharding arithmetic and the rank plumbing bench.py uses.
"""
import os
import socket

import numpy as np
import pytest

from mrcaudiocodec_amd import shard


def test_shard_ranges_cover_exactly():
    for n in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            got = [shard.shard_frames(n, world, r) for r in range(world)]
            assert got[0][0] == 0 and sum(c for _, c in got) == n
            for (f0, c0), (f1, _) in zip(got, got[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in got) - min(c for _, c in got) <= 1
    assert shard.shard_samples(5, 3, 1024) == (5 * 1024, 9 * 1024)
    with pytest.raises(ValueError):
        shard.shard_frames(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mrcaudiocodec_amd import synth
    from oracle import fast
    s = synth.c3_stereo(n_frames)                              # every rank can regenerate the synthetic stream
    first, count = shard.shard_frames(n_frames, world, rank)
    a, b = shard.shard_samples(first, count, 1024)
    local = s[:, a:b]                                          # this rank's slice, halo included
    bl = np.array(fast.blocks_from_stream(local[0], 1024))
    br = np.array(fast.blocks_from_stream(local[1], 1024))
    assert bl.shape[0] == count
    r = fast.encode_joint_batch(bl, br, 1024, 1024)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), first=first, mantissa=r["mantissa"], bit_alloc=r["bit_alloc"],
             ms_switch=r["ms_switch"], reservoir_out=r["reservoir_out"])
    dist.barrier()
    t = shard.max_over_ranks(1.0 + rank)                       # the reduction bench.py uses for its clock
    assert t == float(world)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_encode_equals_unsharded(tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from mrcaudiocodec_amd import synth
    from oracle import fast
    n_frames, world = 7, 2
    mp.spawn(_worker, args=(world, _free_port(), n_frames, str(tmp_path)), nprocs=world, join=True)
    s = synth.c3_stereo(n_frames)
    ref = fast.encode_joint_batch(np.array(fast.blocks_from_stream(s[0], 1024)),
                                  np.array(fast.blocks_from_stream(s[1], 1024)), 1024, 1024)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert [int(p["first"]) for p in parts] == [0, 4]
    for k in ("mantissa", "bit_alloc", "ms_switch", "reservoir_out"):
        assert np.array_equal(np.concatenate([p[k] for p in parts]), ref[k]), k


def test_bench_stream_generator_shards_are_slices_of_one_stream():
    """bench.py's counter-based content: what rank r generates for its frame range is exactly its slice (with the
    one-hop halo) of the single global stream -- for the mono (configs[1]), stereo (configs[4]) and block-switching
    content."""
    torch = pytest.importorskip("torch")
    import bench
    from mrcaudiocodec_amd.shard import shard_frames, shard_samples
    dev = torch.device("cpu")
    total = 37
    for kind in ("c2", "c3", "c4"):
        whole = bench.stream_slice(torch, dev, kind, 0, total)
        for world in (2, 3, 8):
            for rank in range(world):
                first, n = shard_frames(total, world, rank)
                lo, hi = shard_samples(first, n, 1024)
                part = bench.stream_slice(torch, dev, kind, first, n)
                for ch in range(len(whole)):
                    assert torch.equal(part[ch], whole[ch][lo:hi]), (kind, world, rank)
    assert not whole[0][:1024].any() and whole[0].dtype == torch.int16


# ------------------------------------------------------------------ chained (stream) mode: whole streams per rank
def _stream_corpus():
    from mrcaudiocodec_amd import synth
    hops = 5
    tone = synth.c1_sine(hops)
    g = synth.c2_noise(hops, seed=3, sigma=0.05)
    x, sh_sw = synth.c4_transients(hops + 1)
    sh_long = [(i * 1024, 1024, 1024) for i in range(hops - 1)]
    n = len(tone)
    streams = np.stack([np.stack([tone, 0.9 * tone]), np.stack([0.5 * tone + g, 0.5 * tone - g]), np.stack([g, 0.2 * tone])])
    return streams, [sh_long[:2], sh_long[:3], sh_long[:2]], n


def _stream_worker(rank, world, port, out_dir):
    import pickle
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pacfile as opac
    streams, shapes, _ = _stream_corpus()
    first, count = shard.shard_streams(len(shapes), world, rank)
    mine = [opac.encode_stereo_stream(streams[s], shapes[s], huffman=True) for s in range(first, first + count)]   # the checker
    gathered = [None] * world
    dist.all_gather_object(gathered, (first, mine))             # (result collection only: nothing on the data path)
    if rank == 0:
        with open(os.path.join(out_dir, "all.pkl"), "wb") as f:
            pickle.dump(gathered, f)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_stream_sharded_chained_mode(tmp_path):
    # whole streams per rank (a stream is serial through its reservoir); the ranks' lists, in rank order, are the
    # single-process result.  The encoder here is the oracle: what is tested is the partition and the rank plumbing
    # (the GPU form of the same test: tests/test_gpu_chain.py::test_two_processes_share_the_streams)
    pytest.importorskip("torch")
    import pickle
    import torch.multiprocessing as mp
    from oracle import pacfile as opac
    world = 2
    mp.spawn(_stream_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    with open(os.path.join(str(tmp_path), "all.pkl"), "rb") as f:          # (written by this test's own workers)
        gathered = pickle.load(f)
    streams, shapes, _ = _stream_corpus()
    assert [g[0] for g in gathered] == [0, 2]
    got = [b for _, part in gathered for b in part]
    assert len(got) == len(shapes)
    for s in range(len(shapes)):
        assert got[s] == opac.encode_stereo_stream(streams[s], shapes[s], huffman=True), s


def test_stream_shards_cover_exactly():
    for n in (0, 1, 5, 8192):
        for world in (1, 2, 8):
            got = [shard.shard_streams(n, world, r) for r in range(world)]
            assert got[0][0] == 0 and sum(c for _, c in got) == n
