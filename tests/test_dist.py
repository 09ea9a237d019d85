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
    """bench.py's control flow that needs no GPU: the CPU baseline is timed behind every timed region at every world size (no
    rank's clock waits for rank 0's tens of seconds of CPU work), the flags parse, and the worker-side functions -- generation,
    packing in both coefficient forms, the oracle's pictures, the CPU-baseline job -- do what the parent expects of them."""
    import subprocess
    import sys
    import tempfile
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    assert "cpu_late = True" in src
    assert src.index("# ---- timed region") < src.index("# ---- CPU baseline, behind everything the GPU did")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--cpu-baseline-late" in out.stdout and "--no-wide" in out.stdout and "--no-enc" in out.stdout
    assert "--form" in out.stdout and "--no-form16" in out.stdout and "--detail" in out.stdout
    assert shard.gather_floats(3.5, None) == [3.5]      # no process group: this rank's value
    sys.path.insert(0, root)
    import bench
    # the sequence of a stream: a key frame at the head of every interval, inter frames from the pool, shifted by the interval
    assert [bench.seq_frame(i, 0, 1, 6) for i in (0, 1, 2, 64, 65)] == [0, 2, 3, 0, 5]
    assert bench.seq_frame(64, 3, 16, 6) == 0 and bench.seq_frame(1, 3, 16, 6) == 1 + (1 + 9) % 6
    with tempfile.TemporaryDirectory() as td:
        r = bench._gen_job(dict(tag="s", dir=td, pool=2, forms=("levels", "dequant16"), size="qcif", content="dense", seed=5,
                                parity={"frames": 5}, host=True))
        assert len(r["files"]["levels"]) == 3 and len(r["files"]["dequant16"]) == 3 and len(r["balg"]) == 3
        assert r["desc_bytes"]["dequant16"] > r["desc_bytes"]["levels"] > 0
        lv = bench._npz_load(r["files"]["levels"][1])
        assert "dequant" in lv and isinstance(lv["nslots"], int)
        planes = bench._npz_load(r["oracle_planes"])
        assert planes["p0"].shape == (144, 176) and planes["p1"].shape == (72, 88)
        # ... and those planes are what the oracle decodes from the host frames the job wrote (the CPU-baseline job's input)
        import oracle
        ost = oracle.State(176, 144)
        frames = [bench._npz_load(p) for p in r["host"]]
        for i in range(5):
            bench._oracle_decode(ost, frames[bench.seq_frame(i, 0, 1, 2)])
        assert all(np.array_equal(ost.get_plane(oracle.FRAME_PREV, pli), planes["p%d" % pli]) for pli in range(3))
        ost.close()
        assert bench._cpu_job((r["host"], "qcif", 2, 4, False)) > 0 and bench._cpu_job((r["host"], "qcif", 2, 4, True)) > 0
        w = bench._gen_job(dict(tag="w", dir=td, pool=2, forms=("levels",), size="qcif", content="dense", seed=5, widen=(0.5, 1)))
        assert 0 < w["wide_tiles"] <= w["tiles"] and w["desc_bytes"]["levels"] > r["desc_bytes"]["levels"]
