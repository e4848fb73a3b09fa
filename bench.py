#!/usr/bin/env python3
"""bench.py -- headline benchmark of libsmr_hip on MI355X.

One "step" = one pass of the hot path over the synthetic C2 batch: Mercator
projection of every node (kernel 1) + the full draw of one rule (vertex/edge setup, stroke
expansion, tile binning + sort, wave-per-tile raster; kernel 2) onto a freshly cleared
16384 x 16384 canvas.  Inputs (lat/lon SoA, way table) are resident in HBM before the timed
region.  N GPUs: one process per GPU, rank r owns band r of a 16384 x (16384*N) chart (weak
scaling, no data-path collective); value = all pixels rendered / max-over-ranks time.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import smrender_amd as sa                      # noqa: E402
from smrender_amd import synth                 # noqa: E402

HBM_PEAK = 8.0e12        # B/s, MI355X_MICROARCH.md


def make_band(args, band, n_bands):
    """ways of band `band` plus the neighbours' ways that reach into it"""
    W, H = args.size, args.size
    parts = []
    for b in (band - 1, band, band + 1):
        if b < 0 or b >= n_bands:
            continue
        x, y, off, ref = synth.synth_ways(args.ways, W, H, seed=1234 + b, y_off=float(b * H))
        if b != band:
            _, y0, _, y1 = synth.way_bbox(x, y, off, ref)
            keep = (y1 > band * H - 4) & (y0 < (band + 1) * H + 4)
            x, y, off, ref = synth.select_ways(x, y, off, ref, keep)
        parts.append((x, y, off, ref))
    # neighbours first/last keeps the global drawing order (band-major way ids)
    xs = np.concatenate([p[0] for p in parts])
    ys = np.concatenate([p[1] for p in parts])
    node_base = np.cumsum([0] + [len(p[0]) for p in parts])
    ref_base = np.cumsum([0] + [len(p[3]) for p in parts])
    refs = np.concatenate([p[3] + node_base[i] for i, p in enumerate(parts)]).astype(np.int32)
    offs = np.concatenate([p[2][:-1] + ref_base[i] for i, p in enumerate(parts)] + [[ref_base[-1]]]).astype(np.int64)
    return xs, ys, offs, refs


def caption_anchors(n, rank, W, H, band_y0):
    """C4's anchors of one rank: a jittered grid over its band, in chart px"""
    rng = np.random.default_rng(4321 + rank)
    side = int(np.ceil(np.sqrt(n)))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    cx = ((gx.ravel()[:n] + rng.uniform(0.2, 0.8, n)) * W / side)
    cy = ((gy.ravel()[:n] + rng.uniform(0.2, 0.8, n)) * H / side) + band_y0
    return cx, cy


def measured_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/<round>_hbm_traffic.json,
    collected by tools/profile_round.sh in separate rocprofv3 --pmc runs of this same command and
    corrected as MI355X_MICROARCH.md prescribes); None when the workload differs from the profiled one."""
    if (args.ways, args.size, args.dpi) != (1000000, 16384, 300) or args.strong:
        return None
    import re
    best = None
    # the latest set: names sort by their numbers (r01_s5 < r01_s11 < r02_...), not as strings
    for f in sorted(os.listdir(os.path.join(ROOT, "profiles")), key=lambda n: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", n)]):
        if f.endswith("_hbm_traffic.json") and json.load(open(os.path.join(ROOT, "profiles", f))).get("workload", "").startswith("C2 "):
            best = os.path.join(ROOT, "profiles", f)
    if best is None:
        return None
    k = json.load(open(best)).get("kernels", {}).get(kernel)
    return None if k is None else round(k["hbm_bytes"])


VALU_PEAK = 256 * 4 * 2.4e9 / 2      # wave64 vector instructions/s: 4 SIMD-32 per CU, 2 cycles each (MI355X_MICROARCH.md, wave scheduling)
SALU_PEAK = 256 * 2.4e9              # the CU's one scalar unit, one instruction per cycle


def measured_insts(kernel, args):
    """executed vector / scalar instructions per launch of `kernel` from the committed counter pass
    (profiles/<round>_valu_counters.json: SQ_INSTS_VALU, SQ_INSTS_SALU of the same command); None for another workload"""
    if (args.ways, args.size, args.dpi) != (1000000, 16384, 300) or args.strong:
        return None
    import re
    best = None
    for f in sorted(os.listdir(os.path.join(ROOT, "profiles")), key=lambda n: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", n)]):
        if f.endswith("_valu_counters.json"):
            best = os.path.join(ROOT, "profiles", f)
    if best is None:
        return None
    k = json.load(open(best)).get("kernels", {}).get(kernel)
    return None if k is None else (k.get("SQ_INSTS_VALU"), k.get("SQ_INSTS_SALU"), os.path.basename(best))


def _cpu_band(job):
    """one worker of the all-cores baseline: a row band of the sample window through libcairo (module level: picklable)"""
    S, n, dpi, y0, h = job
    from oracle import pyoracle as po
    from smrender_amd import dist as smrdist
    x, y, off, ref = synth.synth_ways(n, S, S, seed=1234)
    keep = smrdist.ways_for_band(y, off, ref, y0, h, smrdist.stroke_margin_px(0.0, 25.4 / dpi, dpi))
    x, y, off, ref = synth.select_ways(x, y, off, ref, keep)
    st = po.Style(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / dpi, 0)
    t0 = time.perf_counter()
    img = po.cairo_scene([(st, po.ways_from_arrays(x, y, off, ref))], dpi, S, h, 0xffffff, 0, y0, S, S)
    return time.perf_counter() - t0, int(img[::7, ::7].astype(np.uint64).sum())


def cpu_baseline(args):
    """The reference's CPU path on a bounded sample of the same workload: our restatement of smrender's Cairo call sequence
    over the real libcairo (oracle/cairo_drive.c), else the plain-C oracle.  Test infrastructure used only as the timed
    baseline.  Runs BEFORE the library touches the GPU (the all-cores figure forks workers).
    value: ONE thread, which is all the reference can use for draw and cap angle=auto (src/smrules_cairo.c:1378-1380; no
    sm_threaded() in act_draw_ini); all_cores: as many processes as this process may run on, one row band each (what an
    operator could do by hand with N smrender runs on N windows); caption_scan: find_angle for C4-shaped captions the
    reference's way (sub-pixel cut-out replay, num_steps rotated paints through libcairo, the per-pixel metric)."""
    from oracle import pyoracle as po
    S = args.cpu_window
    n = max(1, int(round(args.ways * (S / args.size) ** 2)))
    x, y, off, ref = synth.synth_ways(n, S, S, seed=1234)
    ways = po.ways_from_arrays(x, y, off, ref)
    st = po.Style(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / args.dpi, 0)
    out = {"cores": 1, "kind": "port", "unit": "Mpx/s"}
    p = po.make_proj(45.0, 15.0, 100000.0, float(S), float(S), args.dpi)
    lat, lon = synth.px_to_geo(p, x, y)
    t0 = time.perf_counter()
    po.project(p, lat, lon)
    t_proj = time.perf_counter() - t0
    have_cairo = po.cairo() is not None
    if have_cairo:
        t0 = time.perf_counter()
        po.cairo_scene([(st, ways)], args.dpi, S, S, 0xffffff)
        t = time.perf_counter() - t0
        out["sample"] = (f"{n} ways of the C2 generator on a {S}x{S} window, smrender's Cairo call sequence "
                         f"(recording surface, group per rule, replay at dpi/72) over libcairo {po.cairo().cd_version().decode()}, 1 thread")
    else:
        t0 = time.perf_counter()
        po.oracle_scene([(st, ways)], args.dpi, S, S, 0xffffff)
        t = time.perf_counter() - t0
        out["sample"] = f"{n} ways of the C2 generator on a {S}x{S} window, plain-C oracle (libcairo not loadable), 1 thread"
    out["value"] = round(S * S / 1e6 / t, 3)
    out["seconds"] = round(t, 3)
    out["proj_nodes_per_s"] = round(len(lat) / t_proj, 1)
    if have_cairo:
        # caption scan, the reference's way, on the recording cairo_scene left in the driver (C4's shape)
        L, cd = po.orc(), po.cairo()
        import ctypes as C
        fw, fh, fx = 270, 20, 135
        r = C.c_int(); steps = L.orc_num_steps(fw, fh, args.dpi, C.byref(r)); r = r.value
        r_pt = float(np.hypot(fw, fh)) * 72 / args.dpi
        fg = np.zeros((fh, fw), np.uint32); L.orc_make_plane(po.P(fg), fw, fh, fx, 0)
        rng = np.random.default_rng(7)
        ncap = 40
        sub = np.zeros(r * r, np.uint32)
        var = C.c_double()
        t0 = time.perf_counter()
        for k in range(ncap):
            cx, cy = rng.uniform(r, S - r, 2)
            assert cd.cd_cut_out(cx * 72 / args.dpi, cy * 72 / args.dpi, r_pt, float(args.dpi), po.P(sub), len(sub)) == r
            dst = np.zeros((fh, fw), np.uint32)
            for i in range(steps):
                cd.cd_rot_paint(po.P(dst), fw, fh, po.P(sub), r, r, i * 2 * np.pi / steps)
                L.orc_dist(po.P(dst), po.P(fg), fw, fh, C.byref(var))
        tc = time.perf_counter() - t0
        out["caption_scan"] = {"captions_per_s": round(ncap / tc, 2), "ms_per_caption": round(tc / ncap * 1e3, 2), "cores": 1,
                               "sample": f"{ncap} captions, {fw}x{fh} plane, {r}-px cut-out replayed from the recording of {n} ways, {steps} rotated paints "
                                         f"(libcairo) + the per-pixel metric each"}
        # all cores: row bands of a larger window, one process each
        try:
            ncore = min(len(os.sched_getaffinity(0)), 16)
        except AttributeError:
            ncore = min(os.cpu_count() or 1, 16)
        if ncore > 1:
            import multiprocessing as mpx
            SA = min(args.size, 8192)
            na = max(1, int(round(args.ways * (SA / args.size) ** 2)))
            per = -(-SA // ncore)
            jobs = [(SA, na, args.dpi, y0, min(per, SA - y0)) for y0 in range(0, SA, per)]
            t0 = time.perf_counter()
            with mpx.get_context("fork").Pool(ncore) as pool:
                res = pool.map(_cpu_band, jobs)
            ta = time.perf_counter() - t0
            out["all_cores"] = {"cores": ncore, "value": round(SA * SA / 1e6 / ta, 2), "unit": "Mpx/s", "seconds": round(ta, 2),
                                "sample": f"{na} ways on a {SA}x{SA} window in {len(jobs)} row bands, one process each (generation of the ways included in the wall time: "
                                          f"{round(max(r_[0] for r_ in res), 2)} s is the slowest band's libcairo time)"}
    return out


def other_charts(args):
    """Outside the timed region, N=1 only: the two other single-GPU charts BASELINE.json names, each timed over 3 draws
    after one warm-up.  (a) north_star's target: 1 M ways / 10 M nodes on a 32768^2 canvas; (b) C3-like: a fractal
    coastline of 200 k vertices + 5 000 islands as ONE directional curve=1 fill, 50 depth contours as curve=1 borders,
    A0 at 600 dpi (19866 x 28087 px)."""
    out = {}

    def timed(cv, rules):
        ms = []
        for it in range(4):
            cv.clear(); sa.sync()
            t0 = time.perf_counter()
            for st, (dx, dy, doff, dref, nw, nr) in rules:
                cv.draw(st, dx, dy, doff, dref, nw, nr)
            sa.sync()
            ms.append((time.perf_counter() - t0) * 1e3)
        return float(np.mean(ms[1:]))

    def dev(arrs):
        x, y, off, ref = arrs
        return (sa.DeviceArray.from_numpy(x), sa.DeviceArray.from_numpy(y), sa.DeviceArray.from_numpy(off.astype(np.int64)),
                sa.DeviceArray.from_numpy(ref.astype(np.int32)), len(off) - 1, len(ref))

    S = 32768
    arrs = synth.synth_ways(1000000, S, S, seed=4242, mean_nodes=10)
    cv = sa.Canvas(S, S, 300, 0xffffff)
    ms = timed(cv, [(sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / 300, 0, 0, 0, 0, 0, 0), dev(arrs))])
    # the fixed cost of a pass: a chart's rule set is hundreds of small rules (the canvas is not cleared in between)
    small = {}
    for n_small in (10, 2000):
        r = [(sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / 300, 0, 0, 0, 0, 0, 0), dev(synth.synth_ways(n_small, S, S, seed=5)))]
        for st, (dx, dy, doff, dref, nw, nr) in r * 3:
            cv.draw(st, dx, dy, doff, dref, nw, nr)
        sa.sync()
        t0 = time.perf_counter()
        for st, (dx, dy, doff, dref, nw, nr) in r * 20:
            cv.draw(st, dx, dy, doff, dref, nw, nr)
        sa.sync()
        small[f"{n_small}_ways_ms"] = round((time.perf_counter() - t0) * 1e3 / 20, 3)
    cv.close()
    out["small_rules"] = dict(small, workload=f"one rule of 10 / 2000 closed ways (fill + 2 px border) on the {S}x{S} canvas, mean of 20 draws")
    out["target_32k"] = {"workload": f"1000000 closed ways / {len(arrs[0])} nodes, fill + 2 px border, {S}x{S} px @ 300 dpi (draw only, nodes projected)",
                         "ms": round(ms, 2), "Mpx_per_s": round(S * S / 1e6 / (ms * 1e-3), 1)}
    W, H = 19866, 28087
    land, rings = synth.synth_coast(200000, 5000, 50, W, H, seed=1234)
    cv = sa.Canvas(W, H, 600, 0xc0d8f0)
    st_land = sa.DrawStyle(1, 0, 0xf0e0a0, 0, 0.0, 0.0, 0, 0, 0, 0, 1, 0).set_curve(0.2)
    st_ring = sa.DrawStyle(0, 1, 0, 0x4060a0, 0.0, 0.15, 0, 0, 0, 0, 0, 0).set_curve(0.2)
    ms = timed(cv, [(st_land, dev(land)), (st_ring, dev(rings))])
    cv.close()
    out["c3_coast"] = {"workload": f"coastline 200000 vertices + 5000 islands (directional=1;curve=1 fill), 50 depth contours "
                                   f"({len(rings[3])} refs, curve=1 borders 0.15 mm), {W}x{H} px @ 600 dpi (draw only)",
                       "ms": round(ms, 2), "Mpx_per_s": round(W * H / 1e6 / (ms * 1e-3), 1)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ways", type=int, default=1000000)
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--dpi", type=int, default=300)
    ap.add_argument("--captions", type=int, default=200000, help="C4 captions scanned after the timed region (BASELINE config 4: 200 000; 0 = skip)")
    ap.add_argument("--no-replay", action="store_true", help="skip the caption scan through re-rasterised cut-outs (profiling runs: keeps the chart's own kernels apart)")
    ap.add_argument("--cpu-window", type=int, default=3072)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-charts", action="store_true", help="skip the 32k target and the C3-like chart (reported outside the timed region)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: ONE chart of --strong-size^2 px (north_star's 32k target: 1 M ways / 10 M nodes) "
                                                          "split into one row band per GPU; value = the chart's Mpx / max-over-ranks time")
    ap.add_argument("--strong-size", type=int, default=32768)
    ap.add_argument("--halo", choices=("auto", "host"), default="auto", help="N>1 caption scan: 'host' stages the halo strips through host memory "
                                                                            "instead of RCCL on views of the canvas")
    ap.add_argument("--dump", default=None, help="directory: every rank writes band_<rank>.npz (its band's pixels, its captions' anchors and angles)")
    args = ap.parse_args()

    from smrender_amd import dist as smrdist
    # the CPU baseline first: its all-cores leg forks workers, which a process that has touched the GPU must not do
    cpu_res = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        cpu_res = cpu_baseline(args)
    grp = smrdist.Group()
    rank, world = grp.rank, grp.world
    sa.init(grp.local)

    W, H = args.size, args.size
    n_bands = world
    if args.strong:
        # ONE chart, every rank a row band of it: the ways whose stroke-inflated extent meets the band (identical arithmetic
        # on both sides of a boundary, no pixel exchange for draw: DESIGN 6)
        S = args.strong_size
        W = S
        x_all, y_all, off_all, ref_all = synth.synth_ways(args.ways, S, S, seed=4242, mean_nodes=10)
        band_list = smrdist.bands(S, world)
        band_y0, H = band_list[rank]
        keep = smrdist.ways_for_band(y_all, off_all, ref_all, band_y0, H, smrdist.stroke_margin_px(0.0, 25.4 / args.dpi, args.dpi))
        x_px, y_px, way_off, ref = synth.select_ways(x_all, y_all, off_all, ref_all, keep)
        way_off, ref = way_off.astype(np.int64), ref.astype(np.int32)
        chart_h = S
        del x_all, y_all, off_all, ref_all
    else:
        x_px, y_px, way_off, ref = make_band(args, rank, n_bands)
        band_y0, chart_h = rank * H, H * n_bands
    # chart: W x chart_h px; window chosen so that the chart spans ~1.4 deg at 45N
    proj = sa.Projection.from_window(45.0, 15.0, 100000.0 * (W / 16384.0), float(W), float(chart_h), args.dpi)
    sa.set_projection(proj)
    lat, lon = synth.px_to_geo(proj, x_px, y_px)
    n_nodes, n_ways, n_refs = len(lat), len(way_off) - 1, len(ref)
    if n_nodes & 1:      # keep 16-byte alignment of every array trivially (pad one unused node)
        lat = np.append(lat, lat[-1]); lon = np.append(lon, lon[-1])
    d_lat, d_lon = sa.DeviceArray.from_numpy(lat), sa.DeviceArray.from_numpy(lon)
    d_x, d_y = sa.DeviceArray((len(lat),), np.float64), sa.DeviceArray((len(lat),), np.float64)
    d_off, d_ref = sa.DeviceArray.from_numpy(way_off), sa.DeviceArray.from_numpy(ref)
    style = sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 25.4 / args.dpi, 0, 0, 0, 0, 0, 0)
    cv = sa.Canvas(W, chart_h, args.dpi, 0xffffff, band_y0, H)
    sa.set_profiling(True)

    def step():
        sa.project(d_lat, d_lon, d_x, d_y)
        cv.clear()
        cv.draw(style, d_x, d_y, d_off, d_ref, n_ways, n_refs)

    def barrier():
        sa.sync()
        grp.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    stage = {k: 0.0 for k in ("proj_ms", "edges_ms", "prims_ms", "cover_ms", "bin_ms", "raster_ms")}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        t = sa.timings()
        for k in stage:
            stage[k] += t[k]
    barrier()
    dt = grp.max(time.perf_counter() - t0)
    tm = sa.timings()
    checksum = int(cv.read(0, 0, W, min(64, H)).astype(np.uint64).sum())

    # ---- C4, outside the timed region: caption auto-rotation scan over the rendered band ----
    cap = None
    if args.captions > 0:
        fg_w, fg_h, fg_x = 270, 20, 135            # E-aligned 2 mm caption at 300 dpi -> 72 angles, 271 px cut-out
        n_steps, r_px = sa.caprot_steps(fg_w, fg_h, args.dpi)
        halo_path = None
        if world > 1:
            halo = r_px // 2 + 2
            # strips straight from the canvas band in HBM over RCCL into device buffers (no host round trip); the host-staged
            # exchange is the path of a gloo group (SMR_DIST_BACKEND=gloo: ranks that share a GPU, tests/test_gpu_a_two_ranks.py)
            # and the fallback if the device exchange fails on a node
            halo_path = "device" if grp.device.type == "cuda" and args.halo != "host" else "host"
            if halo_path == "device":
                try:
                    grp.exchange_halo_device(cv, halo)
                except Exception as e:      # noqa: BLE001
                    print(f"rank {rank}: device halo exchange failed ({e}); host-staged exchange", file=sys.stderr)
                    halo_path = "host (device exchange failed)"
            # every rank must take the same path: the exchanges are collective
            if grp.sum(0.0 if halo_path == "device" else 1.0) > 0 and halo_path == "device":
                halo_path = "host (device exchange failed on another rank)"
            if halo_path != "device":
                above, below = grp.exchange_halo(cv.read(0, 0, W, min(halo, H)), cv.read(0, max(0, H - halo), W, min(halo, H)), halo)
                cv.set_halo(above, below)
        cx, cy = caption_anchors(args.captions, rank, W, H, band_y0)
        d_cx, d_cy = sa.DeviceArray.from_numpy(cx), sa.DeviceArray.from_numpy(cy)
        d_ang = sa.DeviceArray((args.captions,), np.float64)
        cv.caprot(d_cx, d_cy, args.captions, fg_w, fg_h, fg_x, 0x000000, 0.0, 1.0, d_ang)      # warm-up
        barrier()
        cv.caprot(d_cx, d_cy, args.captions, fg_w, fg_h, fg_x, 0x000000, 0.0, 1.0, d_ang)
        barrier()
        cap_ms = grp.max(sa.timings()["caprot_ms"])
        ang = d_ang.numpy()
        cap_bytes = (4.0 * r_px * r_px + 4.0 * fg_w * fg_h + 32.0) * args.captions          # SURVEY 8d per candidate
        crop = {"captions": args.captions * world, "caprot_ms": round(cap_ms, 3), "captions_per_s": round(args.captions * world / (cap_ms * 1e-3), 1),
                "roofline": {"kernel": "k_caprot", "bound": "hbm", "achieved": round(cap_bytes / (cap_ms * 1e-3) / 1e9, 2),
                             "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(cap_bytes / (cap_ms * 1e-3) / HBM_PEAK, 5)},
                "distinct_angles": int(len(np.unique(ang))), "halo_exchange": halo_path,
                "cutout": "crop of the canvas at the rounded anchor (smr_hip_caprot on the chart): NOT the reference's convention, 11-13 % of the %.1f tags differ"}
        # the headline of the scan is the reference's own convention (below, when it runs); the crop scan is the labelled alternative
        cap = {"workload": f"C4: {args.captions} captions per GPU, {fg_w}x{fg_h} px plane, {n_steps} trial angles, {r_px} px cut-outs",
               "angles": n_steps, "cutout_px": r_px, "plane_px": [fg_w, fg_h], "crop_alternative": crop}
        if world == 1 and not args.strong and not args.no_replay:
            # the same captions the reference's way: every cut-out re-rasterised at its anchor's sub-pixel offset (atlases of
            # windows, smr_hip_draw_windows), wall time of everything: instancing, drawing, scan
            sa.set_profiling(False)
            per = 65536 // ((r_px + 3) & ~3)
            sa.sync()
            t0 = time.perf_counter()
            ang2 = np.zeros(args.captions)
            at = None
            d2 = sa.DeviceArray((per,), np.float64)
            for k0 in range(0, args.captions, per):
                k1 = min(args.captions, k0 + per)
                if at is None:      # one atlas for the whole scan, set up again per batch (smr_hip_windows_reset)
                    at = sa.Canvas.windows(W, chart_h, args.dpi, 0xffffff, cx[k0:k1], cy[k0:k1], r_px, float(np.hypot(fg_w, fg_h)))
                else:
                    at.reset_windows(cx[k0:k1], cy[k0:k1], r_px, float(np.hypot(fg_w, fg_h)))
                at.draw_windows(style, d_x, d_y, d_off, d_ref, n_ways, n_refs)
                ax, ay = at.window_anchors()
                at.caprot(sa.DeviceArray.from_numpy(ax), sa.DeviceArray.from_numpy(ay), k1 - k0, fg_w, fg_h, fg_x, 0x000000, 0.0, 1.0, d2)
                sa.sync()
                ang2[k0:k1] = d2.numpy()[:k1 - k0]
            if at is not None:
                at.close()
            t_rep = time.perf_counter() - t0
            sa.set_profiling(True)
            tag = lambda v: np.array(["%.1f" % (q * 180 / np.pi) for q in v])
            cap.update({"cutout": "re-rasterised at the anchor's sub-pixel offset, as cairo_smr_cut_out replays the recording (smr_hip_draw_windows)",
                        "captions": args.captions, "atlases": (args.captions + per - 1) // per, "seconds": round(t_rep, 3),
                        "captions_per_s": round(args.captions / t_rep, 1),
                        "tags_differing_from_the_crop_scan": int((tag(ang) != tag(ang2)).sum())})

    if args.dump:
        os.makedirs(args.dump, exist_ok=True)
        np.savez(os.path.join(args.dump, f"band_{rank}.npz"), y0=band_y0, h=H, pixels=cv.read(), n_ways=n_ways,
                 cx=cx if cap is not None else np.zeros(0), cy=cy if cap is not None else np.zeros(0), angles=ang if cap is not None else np.zeros(0))

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        mpx = W * chart_h / 1e6
        stage = {k: v / args.steps for k, v in stage.items()}
        # SURVEY 8d, per draw pass on one band: every canvas pixel stored once, every projected vertex and every style record read
        # once.  The pass is k_cover (scan conversion, primitive by primitive: reads vertices and records) and the compositing of
        # the tiles, which stores the pixels: k_blend for the tiles that hold k_cover's alpha words only, k_raster for the others
        # (stage "raster_ms" is the two together)
        alg_cover = 16.0 * n_refs + 64.0 * n_ways
        alg_raster = 4.0 * W * H
        alg_bytes = alg_cover + alg_raster
        dom, dom_ms, dom_bytes = ("k_cover", stage["cover_ms"], alg_cover) if stage["cover_ms"] >= stage["raster_ms"] else ("k_blend+k_raster", stage["raster_ms"], alg_raster)
        achieved = dom_bytes / (dom_ms * 1e-3)
        stage_rate = alg_bytes / ((stage["cover_ms"] + stage["raster_ms"]) * 1e-3)
        res = {
            "metric": "rasterized Mpixels/s (full-chart render: projection + draw)",
            "value": round(mpx / (dt / args.steps), 2),
            "unit": "Mpx/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "i64 fixed 24.8 (raster) + f64 (projection)", "data": "synthetic",
            "config": {"workload": (f"north_star target: {args.ways} closed ways x ~10 nodes, fill #4080c0 + 2 px AA border, ONE {W}x{chart_h} px chart @ {args.dpi} dpi"
                                    if args.strong else
                                    f"C2: {args.ways} closed ways x ~32 nodes (non-zero winding), fill #4080c0 + 2 px AA border, "
                                    f"{W}x{H} px canvas per GPU @ {args.dpi} dpi"),
                       "ways_per_gpu": n_ways, "refs_per_gpu": n_refs, "nodes_per_gpu": n_nodes,
                       "canvas_total": [W, chart_h], "partition": f"{world} row band(s), one per GPU"},
            "nodes_per_s": round(n_nodes / (stage["proj_ms"] * 1e-3), 1) if stage["proj_ms"] > 0 else None,
            "stage_ms": {k: round(v, 3) for k, v in stage.items()},
            "counts": {"prims": tm["n_prims"], "edge_slots": tm["n_edges"], "tile_refs": tm["n_tile_refs"], "tiles": tm["n_tiles"],
                       "cover_prims": tm["n_cover_prims"], "cover_left": tm["n_cover_left"], "deltas": tm["n_deltas"], "cover_linear": tm["n_cover_linear"]},
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 5), "traffic": measured_traffic(dom, args),
                         "algorithmic_bytes": dom_bytes, "ms": round(dom_ms, 3)},
            "roofline_draw_pass": {"kernels": ["k_cover", "k_blend", "k_raster"], "bound": "hbm", "achieved": round(stage_rate / 1e9, 2),
                                   "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(stage_rate / HBM_PEAK, 5),
                                   "algorithmic_bytes": alg_bytes, "ms": round(stage["cover_ms"] + stage["raster_ms"], 3),
                                   "traffic": (lambda a, b, c: None if a is None or c is None else a + (b or 0) + c)(
                                       measured_traffic("k_cover", args), measured_traffic("k_blend", args), measured_traffic("k_raster", args))},
            "roofline_proj": {"kernel": "k_project", "bound": "hbm",
                              "achieved": round(32.0 * len(lat) / (stage["proj_ms"] * 1e-3) / 1e9, 2) if stage["proj_ms"] > 0 else None,
                              "peak": HBM_PEAK / 1e9, "unit": "GB/s"},
            "device": sa.device_name(), "checksum": checksum,
        }
        # the draw pass is instruction-bound, not HBM-bound: the second, honest denominator for k_cover is the vector issue rate
        # (instructions from the committed counter pass, duration measured live); the CU's scalar unit is as busy and is stated beside
        mi = measured_insts("k_cover", args)
        if mi is not None and mi[0] and stage["cover_ms"] > 0:
            v_rate, s_rate = mi[0] / (stage["cover_ms"] * 1e-3), (mi[1] or 0) / (stage["cover_ms"] * 1e-3)
            res["roofline_valu"] = {"kernel": "k_cover", "bound": "valu", "achieved": round(v_rate / 1e9, 1), "peak": round(VALU_PEAK / 1e9, 1),
                                    "unit": "G wave64 instr/s", "frac": round(v_rate / VALU_PEAK, 4), "valu_insts": mi[0], "ms": round(stage["cover_ms"], 3),
                                    "salu_insts": mi[1], "salu_achieved": round(s_rate / 1e9, 1), "salu_peak": round(SALU_PEAK / 1e9, 1),
                                    "salu_frac": round(s_rate / SALU_PEAK, 4), "counters": "profiles/" + mi[2]}
        if res["roofline_proj"]["achieved"]:
            res["roofline_proj"]["frac"] = round(res["roofline_proj"]["achieved"] * 1e9 / HBM_PEAK, 4)
        if cap is not None:
            res["caption_scan"] = cap
        if world == 1 and not args.no_other_charts:
            # the chart's files from the device canvas, outside the timed region: PNG and KAP (rank 4)
            import shutil, tempfile
            tmp = tempfile.mkdtemp(prefix="smr_bench_", dir="/tmp")
            try:
                t0 = time.perf_counter()
                ps = cv.write_png(os.path.join(tmp, "chart.png"), level=100)          # SMR_PNG_DEVICE_DEFLATE: what hipsave does
                t_png = time.perf_counter() - t0
                t0 = time.perf_counter()
                pz = cv.write_png(os.path.join(tmp, "chart_zlib.png"))
                t_pngz = time.perf_counter() - t0
                t0 = time.perf_counter()
                ks = cv.write_kap(os.path.join(tmp, "chart.kap"), "VER/2.0\r\n")
                t_kap = time.perf_counter() - t0
                res["output_files"] = {
                    "png": {"seconds": round(t_png, 3), "Mpx_per_s": round(mpx / t_png, 1), "MB": round(ps.bytes_written / 1e6, 2), "deflate": "on the device",
                            "zlib_alternative": {"seconds": round(t_pngz, 3), "MB": round(pz.bytes_written / 1e6, 2), "deflate_threads": pz.threads}},
                    "kap": {"seconds": round(t_kap, 3), "Mpx_per_s": round(mpx / t_kap, 1), "MB": round(ks.file_bytes / 1e6, 2), "colors": ks.n_colors,
                            "reduce_ms": round(ks.reduce_ms, 1), "index_ms": round(ks.index_ms, 1), "rows_ms": round(ks.rows_ms, 1)}}
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
            cv.close()
            res["other_charts"] = other_charts(args)
        if cpu_res is not None:
            res["cpu_baseline"] = cpu_res
        print(json.dumps(res), flush=True)
    cv.close()
    grp.close()


if __name__ == "__main__":
    main()
