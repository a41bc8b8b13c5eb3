"""
Stream-sharded chained mode on the GPU ("with many streams, shard by stream"): two rank processes, each
with its own handle on the card, encode WHOLE streams (a stream is serial through its bit reservoir) with one chained call
each; their lists in rank order equal what a single process returns for all streams.  No collective on the data path --
gloo only carries the results to rank 0 for the comparison.

The file sorts first among the GPU tests on purpose: every process that touches the GPU here is a CHILD started before
this pytest process has made a HIP call of its own (the single-process reference is computed by a third child).
"""
import os
import pickle
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _many_streams():
    from mrcaudiocodec_amd import synth
    hops = 9
    tone = synth.c1_sine(hops)
    sh_long = [(i * 1024, 1024, 1024) for i in range(hops - 1)]
    x, sh_sw = synth.c4_transients(hops)
    streams, shapes = [], []
    for s in range(5):
        g = synth.c2_noise(hops, seed=40 + s, sigma=0.02 * (s + 1))
        streams.append(np.stack([0.3 * tone + g, 0.25 * tone - 0.5 * g]) if s != 3 else np.stack([x + 0.1 * tone, 0.6 * x + g]))
        shapes.append(sh_long[:4 + s] if s != 3 else sh_sw)
    return np.stack(streams), shapes


def _rank(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from mrcaudiocodec_amd import Handle, pacfile, shard
    streams, shapes = _many_streams()
    hd = Handle(device_id=0)                                       # (the ranks share the one card of the test box)
    if rank == world:                                              # the extra process: everything in one call
        whole = pacfile.encode_stereo_streams(hd, streams, shapes)
        hd.close()
        with open(os.path.join(out_dir, "whole.pkl"), "wb") as f:
            pickle.dump(whole, f)
        return
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, mine = shard.encode_streams_sharded(hd, streams, shapes, world, rank)
    hd.close()
    gathered = [None] * world
    dist.all_gather_object(gathered, (first, mine))                # (result collection only)
    if rank == 0:
        with open(os.path.join(out_dir, "ranks.pkl"), "wb") as f:
            pickle.dump(gathered, f)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_processes_share_the_streams(tmp_path):
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    world = 2
    mp.spawn(_rank, args=(world, port, str(tmp_path)), nprocs=world + 1, join=True)
    with open(os.path.join(str(tmp_path), "ranks.pkl"), "rb") as f:        # (files this test's own workers wrote)
        gathered = pickle.load(f)
    with open(os.path.join(str(tmp_path), "whole.pkl"), "rb") as f:
        whole = pickle.load(f)
    assert [g[0] for g in gathered] == [0, 3]
    got = [b for _, part in gathered for b in part]
    assert len(got) == 5 and got == whole
    # ... and it is the reference's file for each stream (the oracle is the checker)
    from oracle import pacfile as opac
    streams, shapes = _many_streams()
    assert got[3] == opac.encode_stereo_stream(streams[3], shapes[3], huffman=True)
