"""First contact of the N>1 path with a GPU: `bench.py --gpus 2` as the driver launches it, two ranks that share the
box's one GPU, torch.distributed over gloo (SMR_DIST_BACKEND=gloo; RCCL wants one device per rank).  Both ranks render their
row band with the HIP library, exchange the caption scan's halo strips through host memory (Group.exchange_halo, the path
bench.py takes when the device exchange is not available) and scan their captions.  Afterwards this process renders the same
chart on ONE canvas and compares: the stitched bands are the chart, bit for bit, and every caption has the angle the
one-canvas scan gives it.

The file sorts before the other GPU tests on purpose: the children are started before this process has made any GPU call."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WAYS, SIZE, CAPS = 20000, 2048, 600


def _launch(tmp_path, name, extra, port):
    out = tmp_path / name
    env = dict(os.environ, SMR_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--ways", str(WAYS), "--size", str(SIZE), "--captions", str(CAPS), "--dump", str(out)] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1, r.stdout[-2000:]                      # rank 0 alone prints the line
    res = json.loads(line[0])
    assert res["n_gpus"] == 2 and res["value"] > 0
    assert res["caption_scan"]["crop_alternative"]["halo_exchange"] == "host"
    bands = [np.load(out / f"band_{k}.npz") for k in range(2)]
    return res, bands


def _one_canvas(sa, W, chart_h, arrs, bands, dpi=300):
    """the chart on ONE canvas in this process, and the angles of the ranks' captions scanned on it"""
    import bench
    from smrender_amd import synth
    x_px, y_px, off, ref = arrs
    proj = sa.Projection.from_window(45.0, 15.0, 100000.0 * (W / 16384.0), float(W), float(chart_h), dpi)
    sa.set_projection(proj)
    lat, lon = synth.px_to_geo(proj, x_px, y_px)
    if len(lat) & 1:
        lat = np.append(lat, lat[-1]); lon = np.append(lon, lon[-1])
    d_lat, d_lon = sa.DeviceArray.from_numpy(lat), sa.DeviceArray.from_numpy(lon)
    d_x, d_y = sa.DeviceArray((len(lat),), np.float64), sa.DeviceArray((len(lat),), np.float64)
    sa.project(d_lat, d_lon, d_x, d_y)
    cv = sa.Canvas(W, chart_h, dpi, 0xffffff)
    st = sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / dpi, 0, 0, 0, 0, 0, 0)
    cv.draw(st, d_x, d_y, sa.DeviceArray.from_numpy(off.astype(np.int64)), sa.DeviceArray.from_numpy(ref.astype(np.int32)), len(off) - 1, len(ref))
    full = cv.read()
    angles = []
    for b in bands:
        n = len(b["cx"])
        d_ang = sa.DeviceArray((n,), np.float64)
        cv.caprot(sa.DeviceArray.from_numpy(b["cx"]), sa.DeviceArray.from_numpy(b["cy"]), n, 270, 20, 135, 0x000000, 0.0, 1.0, d_ang)
        angles.append(d_ang.numpy())
    cv.close()
    return full, angles


def _compare(full, angles, bands, chart_h):
    rows = 0
    for k, b in enumerate(bands):
        y0, h = int(b["y0"]), int(b["h"])
        assert y0 == rows, (k, y0, rows)
        rows += h
        assert np.array_equal(b["pixels"], full[y0:y0 + h]), (k, int((b["pixels"] != full[y0:y0 + h]).sum()))
        assert len(b["angles"]) == CAPS
        assert np.array_equal(b["angles"], angles[k]), (k, int((b["angles"] != angles[k]).sum()))
        assert len(np.unique(b["angles"])) > 8
    assert rows == chart_h
    # the captions near the boundary really look across it: the scan without the neighbour's rows gives other angles
    near = sum(int(((b["cy"] > bands[1]["y0"] - 100) & (b["cy"] < bands[1]["y0"] + 100)).sum()) for b in bands)
    assert near >= 10, near


def test_two_ranks_on_one_gpu_weak_and_strong(tmp_path):
    # children first: no GPU call has been made in this process yet (this file sorts first among the GPU tests)
    res_w, bands_w = _launch(tmp_path, "weak", [], 29531)
    res_s, bands_s = _launch(tmp_path, "strong", ["--strong", "--strong-size", str(SIZE)], 29533)
    assert res_w["scaling"] == "weak" and res_s["scaling"] == "strong"
    assert res_w["config"]["canvas_total"] == [SIZE, 2 * SIZE] and res_s["config"]["canvas_total"] == [SIZE, SIZE]

    import torch      # noqa: F401
    import smrender_amd as sa
    from smrender_amd import synth
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    sa.init(0)
    # weak: rank r owns band r of a SIZE x 2 SIZE chart; the chart's ways are the bands' in band-major order
    parts = [synth.synth_ways(WAYS, SIZE, SIZE, seed=1234 + b, y_off=float(b * SIZE)) for b in range(2)]
    nb = np.cumsum([0] + [len(p[0]) for p in parts]); rb = np.cumsum([0] + [len(p[3]) for p in parts])
    arrs = (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
            np.concatenate([p[2][:-1] + rb[i] for i, p in enumerate(parts)] + [[rb[-1]]]),
            np.concatenate([p[3] + nb[i] for i, p in enumerate(parts)]))
    full, angles = _one_canvas(sa, SIZE, 2 * SIZE, arrs, bands_w)
    _compare(full, angles, bands_w, 2 * SIZE)
    # a rank holds its own ways and only those of the neighbour that reach into its band
    assert all(WAYS < int(b["n_ways"]) < 2 * WAYS for b in bands_w)
    # the anchors are the ones bench.py derives for the rank
    cx, cy = bench.caption_anchors(CAPS, 1, SIZE, SIZE, SIZE)
    assert np.array_equal(cx, bands_w[1]["cx"]) and np.array_equal(cy, bands_w[1]["cy"])

    # strong: ONE SIZE x SIZE chart, every rank the ways that meet its band
    arrs = synth.synth_ways(WAYS, SIZE, SIZE, seed=4242, mean_nodes=10)
    full, angles = _one_canvas(sa, SIZE, SIZE, arrs, bands_s)
    _compare(full, angles, bands_s, SIZE)
    assert all(0 < int(b["n_ways"]) < WAYS for b in bands_s)
