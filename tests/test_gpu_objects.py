"""SURVEY 8f rank 3 on the GPU: the host's objects exported as structure-of-arrays (smr_hip_objects_create) and the rule
executor's filter for many rules at once (smr_hip_objects_match), checked bit by bit against apply_rule's own steps around the
REAL bs_match_attr of the reference's libsmrender (oracle/_ref, through the miniature host), on every match tag the reference's
shipped rulesets hold (tests/golden/match_tags.txt); then the selections against numpy and against the plugin path."""
import os
import numpy as np
import pytest
from oracle import pyoracle as po
import hostsim as hs

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not hs.available(), reason="oracle/_ref not built")]

HERE = os.path.dirname(os.path.abspath(__file__))
_next_id = [40_000_000]


def shipped_rules():
    out = []
    for line in open(os.path.join(HERE, "golden", "match_tags.txt")).read().split("\n"):
        if not line:
            continue
        f = line.split("\t")
        out.append((f[0], {"none": 0, "open": 1, "closed": 2}[f[1]], list(zip(f[2::2], f[3::2]))))
    return out


def bare(p):
    """a pattern without its markers"""
    if len(p) > 2 and p[0] == p[-1] and p[0] in "!~":
        p = p[1:-1]
    if len(p) > 2 and (p[0] + p[-1]) in ("//", "][", "[]"):
        p = p[1:-1]
    return p


NUMBERS = ["0", "5", "7.5", "7.51", "7.49", "10", "300", "300.0", "301", "899.9", "900", "901", "1e3", "-3", "12.5.1", "", " 8", "-", "-.5",
           "007", "3m", "1.", ".5", "15.0", "20"]
EXTRA_KEYS = ["seamark:light:range", "seamark:light:1:range", "seamark:light:12:range", "seamark:light:colour", "seamark:light_character",
              "seamark:light:2:colour", "seamark:lightx", "name", "ele", "buoy", "seamark:buoy_lateral:shape", "xbuoyx"]
EXTRA_VALS = ["covers", "awash", "always_covers", "dry", "pillar", "yes", "no", "green", "red", "green;red", "landmark"]


def make_world(seed, n_nodes, n_ways, rules):
    """objects whose tags are drawn from what the rules ask for (so that every kind of pattern meets matches and near misses)"""
    rng = np.random.default_rng(seed)
    keys = sorted({bare(k) for _, _, t in rules for k, _ in t if k and "^" not in k and "(" not in k and "|" not in k} | set(EXTRA_KEYS))
    vals = sorted({bare(v) for _, _, t in rules for _, v in t if v and "^" not in v and "(" not in v and "|" not in v} | set(EXTRA_VALS) | set(NUMBERS))
    # pairs that occur together in a rule are likelier than random ones
    pairs = [(bare(k), bare(v)) for _, _, t in rules for k, v in t if k and v and not any(c in k + v for c in "^(|")]

    def tags():
        out = {}
        for _ in range(int(rng.integers(0, 7))):
            if pairs and rng.random() < 0.6:
                k, v = pairs[int(rng.integers(len(pairs)))]
                if rng.random() < 0.3:
                    v = vals[int(rng.integers(len(vals)))]
            else:
                k, v = keys[int(rng.integers(len(keys)))], vals[int(rng.integers(len(vals)))]
            if "=" in k or ";" in k or ";" in v or "=" in v:
                continue
            out[k] = v
        return ";".join(f"{k}={v}" for k, v in out.items())

    base = _next_id[0]
    _next_id[0] += n_nodes + n_ways + 10
    nids = base + rng.permutation(n_nodes).astype(np.int64)          # NOT in id order
    lat = 44.9 + 0.2 * rng.random(n_nodes)
    lon = 14.9 + 0.2 * rng.random(n_nodes)
    hs.add_nodes(nids, lat, lon, [tags() for _ in range(n_nodes)])
    wids = base + n_nodes + np.arange(n_ways, dtype=np.int64)
    refs = []
    for w in range(n_ways):
        n = int(rng.integers(2, 12))
        r = nids[rng.integers(0, n_nodes, n)]
        if rng.random() < 0.4:
            r = np.concatenate([r, r[:1]])                            # closed
        if rng.random() < 0.1:
            r = r.copy(); r[int(rng.integers(len(r)))] = base - 5        # a node that does not exist
        refs.append(r)
        hs.add_way(int(wids[w]), r, tags())
    for i in rng.integers(0, n_nodes, n_nodes // 20):
        hs.set_visible(hs.OSM_NODE, nids[i], 0)
    for i in rng.integers(0, n_ways, n_ways // 20):
        hs.set_visible(hs.OSM_WAY, wids[i], 0)
    return nids, lat, lon, wids, refs


def unpack(bitmap, n):
    return np.unpackbits(bitmap.view(np.uint8), axis=1, bitorder="little")[:, :n]


def test_filter_equals_the_references_on_every_shipped_match_tag(hip):
    sa = hip
    rules = shipped_rules()
    assert len(rules) >= 200
    nids, lat, lon, wids, refs = make_world(5, 3000, 1500, rules)
    on_page = (np.random.default_rng(6).random(len(nids)) < 0.9).astype(np.uint8)
    objs = sa.Objects(hs.object_ptrs(hs.OSM_NODE, nids), hs.object_ptrs(hs.OSM_WAY, wids), threads=3)
    info = objs.info
    assert (info.n_nodes, info.n_ways, info.n_refs) == (len(nids), len(wids), sum(len(r) for r in refs))
    assert info.missing_refs == sum(int(np.sum(r == nids.min() - 5)) for r in refs) > 0
    # the export itself
    assert np.array_equal(objs.node_id.numpy(), nids) and np.array_equal(objs.lat.numpy(), lat) and np.array_equal(objs.lon.numpy(), lon)
    where = {int(i): k for k, i in enumerate(nids)}
    want_ref = np.array([where.get(int(i), -1) for r in refs for i in r], np.int32)
    assert np.array_equal(objs.ref.numpy(), want_ref)
    assert np.array_equal(objs.way_off.numpy(), np.concatenate([[0], np.cumsum([len(r) for r in refs])]))

    handles, kinds = [], []
    for typ, way_kind, tags in rules:
        t = hs.OSM_NODE if typ == "node" else hs.OSM_WAY
        handles.append(hs.match_rule(t, tags, way_kind)); kinds.append(t)
        # the same patterns asked of the other kind of object, and of ways of the other closedness
        handles.append(hs.match_rule(hs.OSM_WAY if t == hs.OSM_NODE else hs.OSM_NODE, tags, 2 if way_kind != 2 else 1))
        kinds.append(hs.OSM_WAY if t == hs.OSM_NODE else hs.OSM_NODE)
    d_on = sa.DeviceArray.from_numpy(on_page)
    bm, counts = objs.match(handles, d_on)
    got = bm.numpy()
    hit = 0
    for r, (h, t) in enumerate(zip(handles, kinds)):
        ids = nids if t == hs.OSM_NODE else wids
        want = hs.filter_objects(h, t, ids, on_page if t == hs.OSM_NODE else None)
        g = unpack(got[r:r + 1], len(ids))[0]
        assert np.array_equal(g, want), (r, rules[r // 2], np.flatnonzero(g != want)[:5])
        assert counts[r] == want.sum()
        assert not np.any(got[r, (len(ids) + 63) // 64:]), "words beyond the objects must be 0"
        hit += want.sum() > 0
    assert hit >= len(handles) * 2 // 5, hit      # not a test of empty sets

    # without a page test (render_all_nodes)
    bm2, counts2 = objs.match(handles[:8])
    for r in range(8):
        ids = nids if kinds[r] == hs.OSM_NODE else wids
        assert np.array_equal(unpack(bm2.numpy()[r:r + 1], len(ids))[0], hs.filter_objects(handles[r], kinds[r], ids))
    objs.close()


def test_numbers_are_read_as_bs_tod_reads_them(hip):
    """]x[ and [x] compare what bs_tod (libsmrender/bstring.c:156-188) makes of the value: no exponent, a second '.' counts as a
    decimal place, anything else ends the number"""
    sa = hip
    base = _next_id[0]; _next_id[0] += 200
    vals = NUMBERS + ["0.1", "0.30000000000000004", "123456789.125", "99999999999999999999", "1..5", "-0", "--1", "4.999999999999999999"]
    ids = base + np.arange(len(vals))
    hs.add_nodes(ids, np.full(len(vals), 45.0), np.full(len(vals), 15.0), [f"depth={v}" for v in vals])
    objs = sa.Objects(hs.object_ptrs(hs.OSM_NODE, ids), [])
    rules = [hs.match_rule(hs.OSM_NODE, [("depth", p)]) for lim in ("0", "0.1", "0.3", "5", "7.5", "12.51", "1000", "-0.5", "123456789.125", "1e20")
             for p in (f"]{lim}[", f"[{lim}]", f"!]{lim}[!", f"~[{lim}]~")]
    bm, counts = objs.match(rules)
    got = unpack(bm.numpy(), len(ids))
    for r, h in enumerate(rules):
        assert np.array_equal(got[r], hs.filter_objects(h, hs.OSM_NODE, ids)), r
    assert 0 < counts.sum() < len(rules) * len(ids)
    objs.close()


def test_selections_and_a_draw_rule_without_a_call_per_object(hip):
    """match -> select -> project -> draw, against the plugin's `draw` action called for exactly the objects the reference's
    filter lets through"""
    sa = hip
    W, H, DPI = 384, 300, 300
    proj = sa.Projection.from_window(45.0, 15.0, 25000.0, float(W), float(H), DPI)
    hs.set_chart(proj, 0xfff8e0)
    sa.set_projection(proj)
    from smrender_amd import synth
    x, y, off, ref = po.synth_ways(80, W, H, seed=31)
    lat, lon = synth.px_to_geo(proj, x, y)
    base = _next_id[0]; _next_id[0] += len(x) + len(off) + 10
    nids = base + np.arange(len(x), dtype=np.int64)
    hs.add_nodes(nids, lat, lon, [("seamark:type=rock" if k % 7 == 0 else "") for k in range(len(x))])
    kinds = ["natural=water", "natural=wood", "natural=water;name=x", "landuse=forest", ""]
    wids = []
    for w in range(len(off) - 1):
        wid = base + len(x) + w
        r = nids[ref[off[w]:off[w + 1]]].copy()
        if w % 9 == 4:
            r[1] = base - 3              # a missing node: poly_line passes over it
        hs.add_way(wid, r, kinds[w % len(kinds)])
        wids.append(wid)
    wids = np.array(wids, np.int64)
    objs = sa.Objects(hs.object_ptrs(hs.OSM_NODE, nids), hs.object_ptrs(hs.OSM_WAY, wids))
    rule_w = hs.match_rule(hs.OSM_WAY, [("natural", "water")], 2)          # closed ways only
    rule_n = hs.match_rule(hs.OSM_NODE, [("seamark:type", "/rock|wreck/")])
    bm, counts = objs.match([rule_w, rule_n])
    want_w = hs.filter_objects(rule_w, hs.OSM_WAY, wids)
    want_n = hs.filter_objects(rule_n, hs.OSM_NODE, nids)
    assert counts[0] == want_w.sum() > 5 and counts[1] == want_n.sum() > 5

    sel_nodes = objs.select_nodes(bm, 1).numpy()
    assert np.array_equal(sel_nodes, np.flatnonzero(want_n))
    d_off, d_ref, d_idx, n_ways, n_refs = objs.select_ways(bm, 0)
    assert np.array_equal(d_idx.numpy(), np.flatnonzero(want_w))
    all_off, all_ref = objs.way_off.numpy(), objs.ref.numpy()
    exp_ref = [all_ref[all_off[w]:all_off[w + 1]] for w in np.flatnonzero(want_w)]
    exp_ref = [r[r >= 0] for r in exp_ref]
    assert any(len(r) != all_off[w + 1] - all_off[w] for r, w in zip(exp_ref, np.flatnonzero(want_w))), "a selected way should have lost a node"
    assert np.array_equal(d_ref.numpy(), np.concatenate(exp_ref))
    assert np.array_equal(d_off.numpy(), np.concatenate([[0], np.cumsum([len(r) for r in exp_ref])]))
    assert n_refs == sum(len(r) for r in exp_ref)

    # the draw: nodes projected once from the export's own arrays, the rule's ways from the selection
    d_x, d_y = sa.DeviceArray((len(nids),), np.float64), sa.DeviceArray((len(nids),), np.float64)
    sa.project(objs.lat, objs.lon, d_x, d_y)
    cv = sa.Canvas(W, H, DPI, 0xfff8e0)
    st = sa.DrawStyle(1, 1, 0x4080c0, 0x000000, 0.0, 0.2, 0, 0, 0, 0, 0, 0)
    cv.draw(st, d_x, d_y, d_off, d_ref, n_ways, n_refs)
    got = cv.read()
    cv.close()
    # the plugin path: `_main` per object, for the objects apply_rule lets through
    sa.lib().smr_hip_plugin_reset()
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=#4080c0;bcolor=black;bwidth=0.2", wids[want_w != 0]) == 0
    L = sa.lib()
    want = np.empty((H, W), np.uint32)
    assert L.smr_hip_canvas_read(L.smr_hip_plugin_canvas(), 0, 0, W, H, want.ctypes.data) == 0
    sa.lib().smr_hip_plugin_reset()
    assert np.array_equal(got, want) and np.sum(got != 0xfffff8e0) > 1000
    # an empty selection is a valid (empty) batch
    none = hs.match_rule(hs.OSM_WAY, [("natural", "lava")])
    bm0, c0 = objs.match([none])
    o0, r0, i0, nw0, nr0 = objs.select_ways(bm0, 0)
    assert c0[0] == 0 and nw0 == 0 and nr0 == 0 and o0.numpy()[0] == 0
    objs.close()
