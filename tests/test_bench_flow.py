"""bench.py's control flow without a GPU: the device side replaced by stand-ins that decode nothing (so the oracle comparison is
switched off with --no-parity and no number means anything) -- what is checked is that every keyed entry is produced, the line is
valid JSON under the 6 KB the driver's 8 KB tail leaves room for, the detail file is written, and the worker processes deliver.
The real thing runs on the GPU box (tools/prof_round6.sh)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeState:
    def __init__(self, w, h, fmt=0, device=None):
        self.w, self.h = w, h
        self.frame_bytes = w * h * 3 // 2

    def read_plane(self, bufi, pli):
        return np.zeros((self.h >> (1 if pli else 0), self.w >> (1 if pli else 0)), np.uint8)

    def ref_idx(self, which):
        return 0

    def close(self):
        pass


class _FakePlan:
    submitted = 0

    def __init__(self, states, descs):
        assert len(states) == len(descs) and all(d is not None for d in descs)

    def submit(self, stream=None):
        _FakePlan.submitted += 1


def test_default_line_fits_the_drivers_tail(monkeypatch, tmp_path, capsys):
    sys.path.insert(0, ROOT)
    import torch
    import theora_amd
    from theora_amd import synth
    import bench
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)

    class _S:
        cuda_stream = 0

        def synchronize(self):
            pass
    monkeypatch.setattr(torch.cuda, "Stream", _S)
    monkeypatch.setattr(torch, "device", lambda *a: None)
    monkeypatch.setattr(theora_amd, "State", _FakeState)
    monkeypatch.setattr(theora_amd, "BatchPlan", _FakePlan)
    monkeypatch.setattr(theora_amd, "synchronize", lambda: None)
    monkeypatch.setattr(theora_amd, "profile_reset", lambda: None)
    monkeypatch.setattr(theora_amd, "profile_enable", lambda on: None)
    monkeypatch.setattr(theora_amd, "profile_read", lambda: ([256, 0], [12.5, 0.0]))
    monkeypatch.setattr(synth, "upload_frame", lambda packed, device="cuda": (dict(nslots=packed["nslots"]), None))
    detail = tmp_path / "detail.json"
    monkeypatch.setattr(sys, "argv", ["bench.py", "--size", "qcif", "--all-entries", "--no-parity", "--no-cpu-baseline", "--no-pmc", "--no-enc", "--no-e2e",
                                      "--steps", "20", "--repeats", "3", "--min-time", "0", "--detail", str(detail)])
    bench.main()
    out = capsys.readouterr().out.strip().splitlines()[-1]
    if os.environ.get("THIP_SHOW_LINE"):
        sys.stderr.write("%d bytes: %s\n" % (len(out), out))
    assert len(out) < 6000, len(out)
    line = json.loads(out)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "entries"):
        assert k in line, k
    assert line["steps"] == 20 and line["n_gpus"] == 1 and line["config"]["workload"].startswith("qcif")
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    want = {"4k_smooth", "form_dequant16", "wide_0pct", "wide_1pct", "wide_10pct", "1080p_single_stream", "1080p_four_streams", "720p_single_stream",
            "1080p_single_gop16"}
    assert want <= set(line["entries"]), sorted(want - set(line["entries"]))
    for name in want:
        e = line["entries"][name]
        assert set(("value", "ms_per_step", "frac", "ws_MB", "gt_IC")) <= set(e), (name, e)
    # 16 states x 7 distinct 1080p command streams do not fit the 256 MB Infinity Cache; one 1080p stream's pool does
    assert line["entries"]["1080p_single_gop16"]["gt_IC"] and not line["entries"]["1080p_single_stream"]["gt_IC"]
    d = json.loads(detail.read_text())
    assert d["entries"]["1080p_single_gop16"]["distinct_descriptors"] == 16 * 7
    assert d["line"]["value"] == line["value"] and "wall_s" in d and d["wall_s"]["total"] > 0
    assert _FakePlan.submitted > 1000
