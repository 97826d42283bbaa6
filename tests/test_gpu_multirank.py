"""The N > 1 code path on a GPU (-m gpu): `bench.py --gpus 2` launched exactly as the driver launches it
(`python -m torch.distributed.run --nproc-per-node 2 ...`), in the rehearsal mode that lets two ranks share the one GPU of
this box (WM_BENCH_BACKEND=gloo for the collective, WM_BENCH_SINGLE_DEVICE=1 so both ranks compute on cuda:0), and
`dist.transcribe_sharded` in two spawned ranks.  The gathered ids of the 2-rank run must equal a 1-rank run of the same
2·B utterances (utterance u always uses mel seed 1000 + u and nothing in an utterance's arithmetic depends on its batch).
Children are started BEFORE this process touches the GPU in these tests' own code path (they are separate processes;
nothing is exec'd from a GPU-initialised process)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return True


def _bench(n, batch, dump, extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    args = ["bench.py", "--gpus", str(n), "--steps", "4", "--warmup", "0", "--no-cpu-baseline", "--no-x4", "--no-extras", "--batch", str(batch),
            "--dump-ids", dump]
    if n == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    import json
    return json.loads(line)


def test_bench_two_ranks_equal_one_rank(hip, tmp_path):
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    r2 = _bench(2, 8, two, {"WM_BENCH_BACKEND": "gloo", "WM_BENCH_SINGLE_DEVICE": "1"})
    r1 = _bench(1, 16, one, {})
    assert r2["n_gpus"] == 2 and r2["scaling"] == "weak" and r2["config"]["parallelism"] == "dp2"
    a, b = np.load(one), np.load(two)
    assert a.shape == b.shape == (16, 4 + 1 + 99)
    assert np.array_equal(a, b)


def _sharded_worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from whisper_mojo_amd import WhisperConfig, dist as wdist, synth
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    wdist.init_from_env("gloo")
    cfg = WhisperConfig.micro()
    first, count = wdist.shard_range(total, rank, world)
    m = Whisper(cfg, max_batch=max(count, 1), device=0)
    m.load(WeightLoader.from_array(synth.synth_weights(cfg, 0)))
    mels = synth.synth_mels(cfg, first, count)  # utterance u <-> mel seed u
    out = wdist.transcribe_sharded(m, mels, total, prompt=(1, 2, 3, 4), eot=-1, max_loop=12)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()
    m.close()


@pytest.mark.parametrize("total", [7, 4])
def test_transcribe_sharded_two_ranks(hip, total):
    """dist.transcribe_sharded on two ranks (ragged split for total = 7) == one model transcribing all utterances."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from whisper_mojo_amd import WhisperConfig, synth
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    cfg = WhisperConfig.micro()
    m = Whisper(cfg, max_batch=total)
    m.load(WeightLoader.from_array(synth.synth_weights(cfg, 0)))
    want = m.transcribe_batch(synth.synth_mels(cfg, 0, total), prompt=(1, 2, 3, 4), eot=-1, max_loop=12)
    assert res[0] == want and res[1] == want


def test_device_gather_buffer_and_rccl_world_of_one(hip, tmp_path):
    """The RCCL path as far as one GPU allows: `bench.py` with the ids left on the device as the gather buffer
    (wm_transcribe_wait_device), a process group of ONE rank on backend "nccl" (RCCL initialises, all_gather_into_tensor runs on
    the device buffer), against the host path of a plain run: same ids."""
    dev, host = str(tmp_path / "dev.npy"), str(tmp_path / "host.npy")
    r1 = _bench(1, 16, dev, {"WM_BENCH_DEVICE_GATHER": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    r0 = _bench(1, 16, host, {})
    assert r1["n_gpus"] == 1
    a, b = np.load(host), np.load(dev)
    assert a.shape == b.shape == (16, 4 + 1 + 99) and np.array_equal(a, b)


def test_wait_device_packs_like_the_host(hip):
    """wm_transcribe_wait_device: the [rows, 1 + stride] gather buffer built on the device equals dist.pack_tokens of the host
    result — ragged row count, a stride wider than the pass, natural stop (utterances of different lengths), coalesced pairs."""
    import torch
    from whisper_mojo_amd import WhisperConfig, dist, synth
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    cfg = WhisperConfig.micro()
    w = synth.synth_weights(cfg, 0)
    mels = synth.synth_mels(cfg, 0, 5)
    for coalesce in (0, 2):
        m = Whisper(cfg, max_batch=5, coalesce=coalesce)
        m.load(WeightLoader.from_array(w))
        free = m.transcribe_batch(mels, prompt=(1, 2, 3, 4), eot=-1, max_loop=20)
        kw = dict(prompt=(1, 2, 3, 4), eot=free[0][4 + 5], max_loop=20)
        want = m.transcribe_batch(mels, **kw)
        assert min(len(x) for x in want) < 25  # an early stop: the device rows hold stale ids of the longer pass behind their length
        toks, cnts = m.last_tokens.copy(), m.last_counts.copy()
        for slot in (0, 1):
            m.transcribe_submit(mels, slot=slot, **kw)
        for slot in (0, 1):
            packed = torch.full((7, 1 + 30), -1, dtype=torch.int32, device="cuda")
            m.transcribe_wait_device(slot, packed)
            assert np.array_equal(packed.cpu().numpy(), dist.pack_tokens(toks, cnts, 30, 7))
            assert dist.gather_tokens_device(packed, 5) == want
        m.close()
