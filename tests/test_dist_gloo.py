"""CPU, world_size 2, gloo: the N>1 path of bench.py / whisper.mojo_amd/dist.py — contiguous utterance sharding and
the single collective (an all-gather of fixed-stride [len | ids] int32 buffers).  The per-rank compute is
stood in for by a deterministic function of the utterance index, so the test checks that every rank ends with every
utterance's ids in utterance order, for even and ragged splits."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_result(u, stride):
    n = 5 + (u * 7) % (stride - 5)
    ids = (np.arange(n, dtype=np.int32) * 31 + u * 1009) % 51865
    return ids


def _worker(rank, world, port, total, stride, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from whisper_mojo_amd import dist as wdist
    r, _, w = wdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    first, count = wdist.shard_range(total, rank, world)
    toks = np.zeros((count, stride), np.int32)
    cnts = np.zeros(count, np.int32)
    for i in range(count):
        ids = _fake_result(first + i, stride)
        toks[i, :len(ids)] = ids
        cnts[i] = len(ids)
    out = wdist.gather_tokens(toks, cnts, total, stride)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 5, 1])
def test_allgather_of_token_buffers_world2(total):
    stride = 40
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, stride, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [_fake_result(u, stride).tolist() for u in range(total)]
    assert res[0] == want and res[1] == want


def test_shard_range_covers_everything():
    from whisper_mojo_amd import dist as wdist
    for total in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 4, 8):
            spans = [wdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert wdist.shard_range(512, 3, 8) == (192, 64)  # BASELINE config 4: rank r owns [64r, 64r+64)


def test_pack_unpack_roundtrip():
    from whisper_mojo_amd import dist as wdist
    toks = np.array([[1, 2, 3, 0, 0], [9, 8, 7, 6, 5]], np.int32)
    buf = wdist.pack_tokens(toks, np.array([3, 5], np.int32), 5, rows=4)
    assert buf.shape == (4, 6) and buf.nbytes == 4 * 6 * 4
    assert wdist.unpack_tokens(buf) == [[1, 2, 3], [9, 8, 7, 6, 5], [], []]
    assert wdist.pack_tokens(np.zeros((64, 201), np.int32), np.zeros(64, np.int32), 200, 64).nbytes == 51456


def test_single_process_gather_is_identity():
    from whisper_mojo_amd import dist as wdist
    toks = np.array([[4, 5, 6, 7]], np.int32)
    assert wdist.gather_tokens(toks, np.array([2], np.int32), 1, 4) == [[4, 5]]
