"""The N>1 path on CPU: two gloo ranks shard a batch of streams, decode them with the
oracle standing in for the device, and gather checksums; the result must equal the
unsharded run (streams are independent: placement cannot change a picture)."""
import os
import socket
import zlib

import numpy as np
import torch.multiprocessing as mp

import oracle
from theora_amd import shard, synth
from tests import util

SPG, W, H, NFRAMES, BASE = 2, 64, 48, 5, 777


def decode_stream_crc(stream_id):
    geom = synth.Geometry(W, H)
    rng = np.random.default_rng(shard.stream_seed(BASE, stream_id))
    st = oracle.State(W, H)
    for f in range(NFRAMES):
        util.oracle_apply(st, synth.gen_frame(geom, rng, 0 if f == 0 else 1, "mixed"))
    c = 0
    for pli in range(3):
        c = zlib.crc32(st.get_plane(oracle.FRAME_PREV, pli).tobytes(), c)
    return c


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = shard.stream_ids(rank, world, SPG)
    crcs = [decode_stream_crc(s) for s in ids]
    shard.barrier(world)
    elapsed, allc = shard.reduce_results(1.0 + rank, crcs, torch.device("cpu"))
    mx = shard.reduce_max([rank * 2.0, 5.0 - rank], torch.device("cpu"))
    per_rank = shard.gather_floats(0.25 + rank, torch.device("cpu"))     # bench.py's ms_per_step_by_rank
    if rank == 0:
        q.put((elapsed, allc, mx, per_rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_unsharded():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    elapsed, allc, mx, per_rank = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert elapsed == 2.0                      # MAX over ranks
    assert mx == [2.0, 5.0]
    assert per_rank == [0.25, 1.25]            # one value per rank, in rank order
    assert allc == [decode_stream_crc(s) for s in range(2 * SPG)]   # global stream order


def test_stream_partition():
    world, spg = 8, 4
    seen = []
    for r in range(world):
        seen += shard.stream_ids(r, world, spg)
    assert seen == list(range(32))
    assert shard.stream_seed(5, 3) != shard.stream_seed(5, 4)


def test_bench_argument_plumbing():
    """bench.py's N > 1 control flow that needs no GPU: the CPU baseline is timed behind the timed region whenever more than one
    rank runs (no rank's clock waits for rank 0's tens of seconds of CPU work), and the flags that select it parse."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    assert "cpu_late = world > 1 or args.cpu_baseline_late" in src
    assert src.index("cpu_late = world > 1") < src.index("# ---- timed region") < src.index("if nparity and rank == 0 and not args.no_cpu_baseline and cpu_late")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--cpu-baseline-late" in out.stdout and "--no-wide" in out.stdout and "--no-enc" in out.stdout
    assert shard.gather_floats(3.5, None) == [3.5]      # no process group: this rank's value
