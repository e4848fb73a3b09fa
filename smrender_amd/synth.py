"""Seeded synthetic inputs of SURVEY 8d (plain numpy; shared by tests and bench.py)."""
import numpy as np


def synth_ways(n_ways, W, H, seed=1234, rmin=4.0, rmax=64.0, mean_nodes=32, y_off=0.0):
    """C2: star-shaped closed polygons (centre uniform, radius log-uniform rmin..rmax px,
    ~Poisson(mean_nodes) vertices, radial jitter U(0.7,1), random orientation).
    Returns (x_px, y_px, way_off, ref): one closing ref per way (ref[last] == ref[first])."""
    rng = np.random.default_rng(seed)
    nv = np.clip(rng.poisson(mean_nodes, n_ways), 4, 256)
    node_off = np.concatenate([[0], np.cumsum(nv)])
    way_off = np.concatenate([[0], np.cumsum(nv + 1)]).astype(np.int64)
    N = int(node_off[-1])
    wid = np.repeat(np.arange(n_ways), nv)
    k = np.arange(N) - node_off[wid]
    cx, cy = rng.uniform(0, W, n_ways), rng.uniform(0, H, n_ways)
    r = np.exp(rng.uniform(np.log(rmin), np.log(rmax), n_ways))
    sgn = rng.choice([-1.0, 1.0], n_ways)
    ph = rng.uniform(0, 2 * np.pi, n_ways)
    ang = ph[wid] + sgn[wid] * 2 * np.pi * k / nv[wid]
    rr = r[wid] * rng.uniform(0.7, 1.0, N)
    x = cx[wid] + rr * np.cos(ang)
    y = cy[wid] + rr * np.sin(ang) + y_off
    ref = np.empty(int(way_off[-1]), np.int32)
    pos = np.arange(N) + wid            # slot of node k of way w in the ref array
    ref[pos] = np.arange(N, dtype=np.int32)
    ref[way_off[1:] - 1] = node_off[:-1].astype(np.int32)
    return x, y, way_off, ref


def select_ways(x, y, way_off, ref, keep):
    """Sub-set of ways (boolean mask over ways) with nodes/refs re-packed."""
    nref = np.diff(way_off)
    ref_keep = np.repeat(keep, nref)
    r = ref[ref_keep]
    used = np.zeros(len(x), bool)
    used[r] = True
    remap = np.cumsum(used) - 1
    new_off = np.concatenate([[0], np.cumsum(nref[keep])]).astype(np.int64)
    return x[used], y[used], new_off, remap[r].astype(np.int32)


def way_bbox(x, y, way_off, ref):
    xs, ys = x[ref], y[ref]
    a = way_off[:-1]
    return (np.minimum.reduceat(xs, a), np.minimum.reduceat(ys, a),
            np.maximum.reduceat(xs, a), np.maximum.reduceat(ys, a))


def px_to_geo(p, x, y):
    """pxf2geo (src/rdata.c:68-72) in numpy: used only to synthesise lat/lon inputs."""
    lon = x * p.wc / p.w + p.lon0
    lat = np.degrees(np.arctan(np.sinh(p.lath_len * (0.5 - y / p.h) + p.lath)))
    return lat, lon


def synth_coast(n_coast, n_islands, n_rings, W, H, seed=1234, hurst=0.75):
    """C3-like: one fractal coastline polygon of n_coast vertices (spectral synthesis of the radius over the
    bearing: star-shaped, so simple), n_islands small islands, n_rings depth contours nested around the coast.
    Returns (land, rings): two (x_px, y_px, way_off, ref) sets of closed ways; land ways run counterclockwise on the chart."""
    rng = np.random.default_rng(seed)
    th = np.linspace(0, 2 * np.pi, n_coast, endpoint=False)
    rad = np.ones(n_coast)
    k = 1
    while k < n_coast // 4:
        rad += 0.35 * k ** (-hurst - 0.5) * np.sin(k * th + rng.uniform(0, 2 * np.pi)) * rng.uniform(0.5, 1.0)
        k = k + 1 if k < 64 else int(k * 1.37) + 1
    rad = np.clip(rad, 0.35, None)
    cx, cy, R = 0.5 * W, 0.5 * H, 0.30 * min(W, H)
    ways = [(cx + R * rad * np.cos(-th), cy + R * rad * np.sin(-th))]          # y points down: -th runs counterclockwise on the chart
    for _ in range(n_islands):
        n = int(rng.integers(8, 64))
        r = np.exp(rng.uniform(np.log(5.0), np.log(80.0))) * rng.uniform(0.7, 1.0, n)
        a = rng.uniform(0, 2 * np.pi) - 2 * np.pi * np.arange(n) / n
        ix, iy = rng.uniform(0, W), rng.uniform(0, H)
        ways.append((ix + r * np.cos(a), iy + r * np.sin(a)))
    rings = []
    for j in range(n_rings):
        sc = 1.03 + 0.5 * (j + 1) / max(n_rings, 1)
        step = max(1, n_coast // max(2000, n_coast // (4 + j)))
        rings.append((cx + sc * R * rad[::step] * np.cos(-th[::step]), cy + sc * R * rad[::step] * np.sin(-th[::step])))

    def pack(ws):
        xs, ys, off, ref, base = [], [], [0], [], 0
        for wx, wy in ws:
            n = len(wx)
            xs.append(wx); ys.append(wy)
            ref.append(np.concatenate([np.arange(base, base + n), [base]]))
            base += n
            off.append(off[-1] + n + 1)
        if not ws:
            return np.zeros(0), np.zeros(0), np.zeros(1, np.int64), np.zeros(0, np.int32)
        return np.concatenate(xs), np.concatenate(ys), np.array(off, np.int64), np.concatenate(ref).astype(np.int32)
    return pack(ways), pack(rings)


def synth_wavy_knots(n_ways, W, H, seed=7, wave_px=(6.0, 30.0), closed=False):
    """Open ways as explicit Bezier paths of the shape cairo_smr_wavy records (src/smrules_cairo.c:752-840): the way is
    resampled every `d` px and every step is one curve_to whose control points sit at -45 and +45 degrees of the step's
    bearing, d / sqrt(2) from its start.  Returns [(x, y, closed)] with x, y = p0, (c1, c2, p) * n in px; closed: rings, whose
    knot ways end in one more point equal to p0 (the mark of a closed way; the path itself stays open)."""
    rng = np.random.default_rng(seed)
    ways = []
    for _ in range(n_ways):
        d = rng.uniform(*wave_px)
        n = int(rng.integers(2, 9))
        vx = rng.uniform(0, W) + np.cumsum(rng.normal(0, 4 * d, n)); vy = rng.uniform(0, H) + np.cumsum(rng.normal(0, 4 * d, n))
        if closed:                   # a ring: the way's last node is its first; the knot way gets one more point = p0 below
            cx, cy, r = rng.uniform(0.2 * W, 0.8 * W), rng.uniform(0.2 * H, 0.8 * H), rng.uniform(3 * d, 6 * d)
            n = int(rng.integers(5, 10)); a = np.sort(rng.uniform(0, 2 * np.pi, n - 1)) * rng.choice([-1, 1])
            vx, vy = cx + r * np.cos(a), cy + r * np.sin(a)
            vx, vy = np.append(vx, vx[0]), np.append(vy, vy[0])
        xs, ys = [vx[0]], [vy[0]]
        sx, sy, rest = vx[0], vy[0], 0.0
        for i in range(1, n):
            seg = np.hypot(vx[i] - sx, vy[i] - sy) + rest
            while seg > d:
                b = np.arctan2(vy[i] - sy, vx[i] - sx)
                ex, ey = sx + d * np.cos(b), sy + d * np.sin(b)
                for a in (b - np.pi / 4, b + np.pi / 4):
                    xs.append(sx + d * np.sqrt(0.5) * np.cos(a)); ys.append(sy + d * np.sqrt(0.5) * np.sin(a))
                xs.append(ex); ys.append(ey)
                sx, sy = ex, ey
                seg -= d
            rest = seg
        if len(xs) >= 4:
            if closed:
                xs.append(xs[0]); ys.append(ys[0])
            ways.append((np.array(xs), np.array(ys), closed))
    return ways
