"""Multi-GPU host logic: one process per GPU, rank r owns a band of canvas rows.

The draw path needs no data-path collective: a way is handed to every rank whose band its
stroke-inflated bounding box meets, and each rank rasterizes it with identical arithmetic
(DESIGN.md §6).  The caption scan reads up to r_px rows beyond the band: those halo strips
are exchanged with the neighbouring ranks (send/recv over RCCL on GPUs, gloo in the CPU tests).
"""
import os
import numpy as np

TILE_H = 4      # SMR_TH: bands start on tile rows so that every rank bins the same tiles


def bands(H, world):
    """Split H canvas rows into `world` contiguous bands [(y0, h)], multiples of the tile height
    except the last.  A chart with fewer tile rows than ranks leaves the last ranks an EMPTY band (h == 0):
    such a rank creates no canvas (smr_hip_canvas_create refuses band_h <= 0) and takes no part in the halo exchange
    (Group.exchange_halo skips neighbours without rows)."""
    per = -(-H // world)
    per = -(-per // TILE_H) * TILE_H
    out = []
    for r in range(world):
        y0 = min(r * per, H)
        out.append((y0, max(0, min(per, H - y0))))
    return out


def way_boxes(x, y, way_off, ref):
    xs, ys = x[ref], y[ref]
    a = way_off[:-1]
    return (np.minimum.reduceat(xs, a), np.minimum.reduceat(ys, a),
            np.maximum.reduceat(xs, a), np.maximum.reduceat(ys, a))


def ways_for_band(y_px, way_off, ref, y0, h, margin_px):
    """Boolean mask over ways: the way's y-extent, grown by `margin_px` (stroke_margin_px: half the stroke width times
    the miter limit 10, + 1 px of antialiasing, + what a curve or a wave leaves the nodes' extent by), meets rows [y0, y0+h)."""
    ys = y_px[ref]
    a = way_off[:-1]
    lo, hi = np.minimum.reduceat(ys, a), np.maximum.reduceat(ys, a)
    return (hi + margin_px >= y0) & (lo - margin_px < y0 + h)


def stroke_margin_px(fill_width_mm, border_width_mm, dpi, curve_factor=0.0, longest_segment_px=0.0, wavy_length_px=0.0):
    """Largest distance a pixel of a way can lie from the extent of its NODES: half of the widest line the
    reference strokes (cairo_smr_border_width: 2*bw + fw, src/smrules_cairo.c:880-889) times Cairo's
    default miter limit 10, plus 1 px.  curve=1 (cairo_smr_poly_curve :697-749): a Bezier stays inside the hull of its
    knots, and control_points (src/bspline_ctrl.c:64-105) puts the inner knots at most curve_factor times the segment's
    length from its end points, so the margin grows by curve_factor * the longest segment.  wavy=1 (:752-840): the
    waves' control points stand off the line by about a wave length."""
    mm2px = dpi / 25.4
    w = max(2 * border_width_mm + fill_width_mm, border_width_mm, fill_width_mm) * mm2px
    return max(w, 1.0) / 2 * 10 + 1 + curve_factor * longest_segment_px + wavy_length_px


def window_margin_px(r_px, stroke_margin):
    """The caption scan on the reference's own cut-outs (smr_hip_draw_windows) needs NO pixel halo: a rank renders
    the windows of the captions whose anchors lie in its band from geometry.  What it needs instead are the ways within a
    window's reach of its band: half a window (the anchor is its middle) plus the rule's own stroke margin."""
    return r_px / 2.0 + 1.0 + stroke_margin


def longest_segment_px(x_px, y_px, way_off, ref):
    """longest segment of any way in px (for stroke_margin_px's curve term)"""
    xs, ys = x_px[ref], y_px[ref]
    d = np.hypot(np.diff(xs), np.diff(ys))
    if len(d) == 0:
        return 0.0
    seam = np.zeros(len(d), bool)
    seam[way_off[1:-1] - 1] = True          # the "segment" from one way's last ref to the next way's first
    d = d[~seam]
    return float(d.max()) if len(d) else 0.0


def caption_levels(cx, cy, reach):
    """Dependency levels of the captions of one rule, in the host's order (the plugin's cap_levels, plugin.cpp): caption i
    must wait for an earlier caption j whose rectangle its cut-out can meet (distance <= reach[i] + reach[j]).  Every
    rank computes the same levels from the same (replicated) anchors, so the levels are consistent across bands."""
    n = len(cx)
    level = np.zeros(n, np.int64)
    if n == 0:
        return level, 0
    cell = 2 * max(1.0, float(np.max(reach)))
    grid = {}
    for i in range(n):
        gx, gy = int(np.floor(cx[i] / cell)), int(np.floor(cy[i] / cell))
        lv = 0
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                for j in grid.get((gx + dx, gy + dy), ()):
                    if np.hypot(cx[i] - cx[j], cy[i] - cy[j]) <= reach[i] + reach[j]:
                        lv = max(lv, level[j] + 1)
        level[i] = lv
        grid.setdefault((gx, gy), []).append(i)
    return level, int(level.max()) + 1


def caption_owner(cy, band_list):
    """rank that scans each caption: the band that holds the anchor's row; anchors above / below the chart go to the first / last
    band that has rows, as on a single canvas"""
    H = band_list[-1][0] + band_list[-1][1]
    row = np.clip(np.floor(cy), 0, H - 1)
    starts = np.array([y0 for y0, h in band_list if h > 0])
    ranks = np.array([r for r, (y0, h) in enumerate(band_list) if h > 0])
    return ranks[np.searchsorted(starts, row, side="right") - 1]


def scan_captions_banded(group, cy, band_list, level, n_levels, scan, paint):
    """The caption scan of ONE rule over row bands (src/smrules_cairo.c:2376-2382: every finished caption
    leaves a rectangle on the rule's private background that later captions of the same rule see).  A caption belongs to
    the rank whose band holds its anchor row; levels run in order on all ranks; after each level the chosen angles are
    summed over the ranks (every caption has one owner, the others contribute 0.0, so the sum is exact) and EVERY rank
    paints the level's rectangles, its neighbours' included, onto its own extended background before the next level:
    a rectangle that straddles a band boundary is seen on both sides, as on a single canvas.

    scan(idx) -> angles of my captions idx (np.int64 indices into the rule's captions);
    paint(idx, angles) paints the rectangles of captions idx (all ranks' captions of the level) where they meet my rows.
    Returns the angles of ALL captions (identical on every rank)."""
    n = len(cy)
    mine = caption_owner(cy, band_list) == group.rank
    angle = np.zeros(n)
    for lv in range(n_levels):
        idx = np.nonzero(level == lv)[0]
        own = idx[mine[idx]]
        part = np.zeros(len(idx))
        if len(own):
            part[mine[idx]] = scan(own)
        part = group.sum_array(part)
        angle[idx] = part
        if lv + 1 < n_levels:
            paint(idx, part)
    return angle


class Group:
    """Thin wrapper over torch.distributed for the bench/tests (backend nccl == RCCL on ROCm)."""

    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.device = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            backend = backend or os.environ.get("SMR_DIST_BACKEND", "nccl")
            # rehearsals of more ranks than the box has GPUs (gloo only: RCCL wants one device per rank) share the devices
            ndev = torch.cuda.device_count()
            if backend != "nccl" and ndev > 0:
                self.local %= ndev
            if backend == "nccl":
                torch.cuda.set_device(self.local)
                self.device = torch.device("cuda", self.local)
                dist.init_process_group("nccl", device_id=self.device)
            else:
                self.device = torch.device("cpu")
                dist.init_process_group(backend)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            if self.device.type == "cuda":
                import torch
                torch.cuda.synchronize()
            self.dist.barrier()

    def max(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def sum_array(self, a):
        """element-wise sum of a float64 array over the ranks"""
        if self.dist is None:
            return np.asarray(a, np.float64)
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a, np.float64)).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def exchange_halo_device(self, canvas, halo):
        """The same exchange without leaving the GPUs: the strips are views of the canvas band in device memory
        (smr_hip_canvas_device), sent and received over RCCL (point-to-point over one xGMI link per neighbour) into
        device buffers that smr_hip_canvas_set_halo_device copies from.  Sets the canvas's halo; returns nothing."""
        if self.dist is None:
            return
        import torch
        from . import hip as H
        dist = self.dist
        assert self.device.type == "cuda", "exchange_halo_device needs the nccl backend"
        H.sync()                                         # the library's stream has finished the band
        top, bottom = canvas.device_rows(0, min(halo, canvas.band_h)), canvas.device_rows(max(0, canvas.band_h - halo), min(halo, canvas.band_h))
        # the neighbours' strips may be shorter than `halo` (a short last band): row counts first
        counts = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        counts[self.rank] = top.shape[0]
        dist.all_reduce(counts)
        ops, bufs = [], {}
        # (both sides of a boundary decide alike: a pair exchanges only if both have rows)
        if self.rank > 0 and counts[self.rank - 1] > 0 and counts[self.rank] > 0:
            bufs["above"] = torch.empty((int(counts[self.rank - 1]), canvas.W), dtype=torch.int32, device=self.device)
            ops.append(dist.P2POp(dist.isend, top, self.rank - 1))
            ops.append(dist.P2POp(dist.irecv, bufs["above"], self.rank - 1))
        if self.rank < self.world - 1 and counts[self.rank + 1] > 0 and counts[self.rank] > 0:
            bufs["below"] = torch.empty((int(counts[self.rank + 1]), canvas.W), dtype=torch.int32, device=self.device)
            ops.append(dist.P2POp(dist.isend, bottom, self.rank + 1))
            ops.append(dist.P2POp(dist.irecv, bufs["below"], self.rank + 1))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        torch.cuda.synchronize()
        a, b = bufs.get("above"), bufs.get("below")
        canvas.set_halo_device(a.data_ptr() if a is not None else None, 0 if a is None else a.shape[0],
                               b.data_ptr() if b is not None else None, 0 if b is None else b.shape[0])

    def exchange_halo(self, top_rows, bottom_rows, halo):
        """Neighbour exchange of canvas strips (uint32 arrays [halo, W], may be shorter at the chart
        edge).  Rank r sends its first `halo` rows up to r-1 and its last `halo` rows down to r+1;
        returns (rows above my band from r-1, rows below my band from r+1), None at the chart edge."""
        if self.dist is None:
            return None, None
        import torch
        dist = self.dist

        def to_t(a):
            t = torch.from_numpy(np.ascontiguousarray(a).view(np.int32))
            return t.to(self.device)

        W = top_rows.shape[1]
        above = below = None
        # a strip may be shorter than `halo` (a band of fewer rows): every rank announces what it sends (top, bottom rows)
        counts = torch.zeros((self.world, 2), dtype=torch.int64, device=self.device)
        counts[self.rank, 0], counts[self.rank, 1] = top_rows.shape[0], bottom_rows.shape[0]
        dist.all_reduce(counts)
        ops, bufs = [], {}
        # (both sides of a boundary decide alike, as in exchange_halo_device: a pair exchanges only if both have rows; a rank
        # with an empty band, bands(), posts nothing and is sent nothing)
        if self.rank > 0 and int(counts[self.rank - 1, 1]) > 0 and top_rows.shape[0] > 0:
            bufs["above"] = torch.empty((int(counts[self.rank - 1, 1]), W), dtype=torch.int32, device=self.device)
            ops.append(dist.P2POp(dist.isend, to_t(top_rows), self.rank - 1))
            ops.append(dist.P2POp(dist.irecv, bufs["above"], self.rank - 1))
        if self.rank < self.world - 1 and int(counts[self.rank + 1, 0]) > 0 and bottom_rows.shape[0] > 0:
            bufs["below"] = torch.empty((int(counts[self.rank + 1, 0]), W), dtype=torch.int32, device=self.device)
            ops.append(dist.P2POp(dist.isend, to_t(bottom_rows), self.rank + 1))
            ops.append(dist.P2POp(dist.irecv, bufs["below"], self.rank + 1))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if "above" in bufs:
            above = bufs["above"].cpu().numpy().view(np.uint32)
        if "below" in bufs:
            below = bufs["below"].cpu().numpy().view(np.uint32)
        return above, below

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
