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
    except the last."""
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
    """Boolean mask over ways: the way's y-extent, grown by `margin_px` (half the stroke width times
    the miter limit 10, + 1 px of antialiasing), meets rows [y0, y0+h)."""
    ys = y_px[ref]
    a = way_off[:-1]
    lo, hi = np.minimum.reduceat(ys, a), np.maximum.reduceat(ys, a)
    return (hi + margin_px >= y0) & (lo - margin_px < y0 + h)


def stroke_margin_px(fill_width_mm, border_width_mm, dpi):
    """Largest distance a pixel of a way can lie from its centre line: half of the widest line the
    reference strokes (cairo_smr_border_width: 2*bw + fw, src/smrules_cairo.c:880-889) times Cairo's
    default miter limit 10, plus 1 px."""
    mm2px = dpi / 25.4
    w = max(2 * border_width_mm + fill_width_mm, border_width_mm, fill_width_mm) * mm2px
    return max(w, 1.0) / 2 * 10 + 1


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
        ops, bufs = [], {}
        if self.rank > 0:
            bufs["above"] = torch.empty((halo, W), dtype=torch.int32, device=self.device)
            ops.append(dist.P2POp(dist.isend, to_t(top_rows), self.rank - 1))
            ops.append(dist.P2POp(dist.irecv, bufs["above"], self.rank - 1))
        if self.rank < self.world - 1:
            bufs["below"] = torch.empty((halo, W), dtype=torch.int32, device=self.device)
            ops.append(dist.P2POp(dist.isend, to_t(bottom_rows), self.rank + 1))
            ops.append(dist.P2POp(dist.irecv, bufs["below"], self.rank + 1))
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
