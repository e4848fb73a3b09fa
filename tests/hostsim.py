"""Python driver of the miniature smrender host (oracle/hostsim.c + the reference's own
libsmrender built into oracle/_ref/).  TEST INFRASTRUCTURE: used by tests/test_plugin*.py only."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
OSM_NODE, OSM_WAY, OSM_REL = 1, 2, 3
_hs = None


def available():
    return os.path.exists(os.path.join(REF_DIR, "libhostsim.so")) and os.path.exists(os.path.join(REF_DIR, "libsmrender_ref.so"))


def host():
    """libhostsim.so + libsmrender_ref.so loaded into the GLOBAL scope, the way the smrender executable
    (-rdynamic) and libsmrender.so present their symbols to a dlopen()ed plugin."""
    global _hs
    if _hs is None:
        C.CDLL(os.path.join(REF_DIR, "libsmrender_ref.so"), mode=C.RTLD_GLOBAL)
        L = C.CDLL(os.path.join(REF_DIR, "libhostsim.so"), mode=C.RTLD_GLOBAL)
        L.hs_set_rdata.argtypes = [C.c_double] * 9 + [C.c_int, C.c_uint]
        L.hs_rule_new.restype = C.c_void_p
        L.hs_rule_new.argtypes = [C.c_int, C.c_char_p, C.c_char_p]
        L.hs_node.argtypes = [C.c_int64, C.c_double, C.c_double, C.c_char_p]
        L.hs_way.argtypes = [C.c_int64, C.c_void_p, C.c_int, C.c_char_p]
        L.hs_rel_of_ways.argtypes = [C.c_int64, C.c_void_p, C.c_int]
        L.hs_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.hs_get_tag.restype = C.c_char_p
        L.hs_get_tag.argtypes = [C.c_int, C.c_int64, C.c_char_p]
        L.hs_poly_area.argtypes = [C.c_longlong, C.c_void_p]
        L.hs_match_rule.restype = C.c_void_p
        L.hs_match_rule.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.hs_filter.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.hs_filter_ptrs.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.hs_synth_world.argtypes = [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_void_p]
        L.hs_object_ptrs.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.hs_set_visible.argtypes = [C.c_int, C.c_int64, C.c_int]
        L.hs_way_refs.argtypes = [C.c_int64, C.c_void_p, C.c_int]
        L.hs_node_pos.argtypes = [C.c_int64, C.c_void_p]
        L.hs_next_node_id.restype = C.c_int64
        L.hs_next_way_id.restype = C.c_int64
        L.rdata_square_mm.restype = C.c_double
        L.rdata_square_nm.restype = C.c_double
        _hs = L
    return _hs


def set_chart(proj, bgcolor=0xffffff):
    """proj: a Projection / oracle Proj (same fields as struct rdata's, src/rdata.h:99-158)"""
    host().hs_set_rdata(proj.lon0, proj.w, proj.h, proj.wc, proj.lath, proj.lath_len, proj.lat_ll, proj.lon_ru, proj.lat_ru,
                        proj.dpi, bgcolor)
    host().hs_make_page_way()


def apply_rule(lib, func, obj_type, params, ids, rule_type=None):
    """what smrender does with `_action_ = func@libsmr_hip.so:params`: dlsym the three entry points,
    call _ini, _main per object (in the order given = bxtree id order), _fini.  Returns hs_apply's code."""
    H = host()
    r = H.hs_rule_new(rule_type or obj_type, func.encode(), params.encode())
    f = [C.cast(getattr(lib, f"act_{func}_{s}"), C.c_void_p) for s in ("ini", "main", "fini")]
    ids = np.ascontiguousarray(ids, np.int64)
    return H.hs_apply(r, f[0], f[1], f[2], obj_type, ids.ctypes.data, len(ids))


def add_nodes(ids, lat, lon, tags=None):
    H = host()
    for k, i in enumerate(ids):
        assert H.hs_node(int(i), float(lat[k]), float(lon[k]), (tags[k] if tags else "").encode()) == 0


def add_way(wid, refs, tags=""):
    refs = np.ascontiguousarray(refs, np.int64)
    assert host().hs_way(int(wid), refs.ctypes.data, len(refs), tags.encode()) == 0


def add_rel(rid, ways):
    ways = np.ascontiguousarray(ways, np.int64)
    assert host().hs_rel_of_ways(int(rid), ways.ctypes.data, len(ways)) == 0


def get_tag(obj_type, oid, key):
    v = host().hs_get_tag(obj_type, int(oid), key.encode())
    return None if v is None else v.decode()


def way_refs(wid):
    """refs of a way in the host's data, or None"""
    out = np.zeros(4096, np.int64)
    n = host().hs_way_refs(int(wid), out.ctypes.data, len(out))
    return None if n < 0 else out[:n].copy()


def node_pos(nid):
    """(lat, lon, tag count) of a node in the host's data, or None"""
    out = np.zeros(2)
    n = host().hs_node_pos(int(nid), out.ctypes.data)
    return None if n < 0 else (out[0], out[1], n)


def next_ids():
    """(id of the next node, id of the next way) the host's malloc_node / malloc_way + osm_*_default will hand out"""
    return int(host().hs_next_node_id()), int(host().hs_next_way_id())


def poly_area(wid):
    """(centroid lat, centroid lon, area in square nautical miles) of a closed way, as the host's poly_area sees it"""
    out = np.zeros(3)
    assert host().hs_poly_area(int(wid), out.ctypes.data) == 0
    return tuple(out)


def match_rule(obj_type, tags, way_kind=0):
    """a rule as init_rule leaves it (match tags parsed into otag[] / stag[]); tags = [(key pattern, value pattern), ...];
    way_kind 1 / 2: the rule way itself is open / closed (ACTION_OPEN_WAY / ACTION_CLOSED_WAY)"""
    n = len(tags)
    K = (C.c_char_p * max(n, 1))(*[k.encode() for k, _ in tags])
    V = (C.c_char_p * max(n, 1))(*[v.encode() for _, v in tags])
    return host().hs_match_rule(obj_type, K, V, n, way_kind)


def filter_objects(rule, obj_type, ids, on_page=None):
    """apply_rule's verdict per object (1 = `_main` would be called) through the reference's own bs_match_attr"""
    ids = np.ascontiguousarray(ids, np.int64)
    out = np.zeros(len(ids), np.uint8)
    op = None if on_page is None else np.ascontiguousarray(on_page, np.uint8)
    assert host().hs_filter(rule, obj_type, ids.ctypes.data, len(ids), None if op is None else op.ctypes.data, out.ctypes.data) == 0
    return out


def object_ptrs(obj_type, ids):
    """the host's own osm_node_t* / osm_way_t* of the given objects, as an array of pointers"""
    ids = np.ascontiguousarray(ids, np.int64)
    out = np.zeros(len(ids), np.uint64)
    assert host().hs_object_ptrs(obj_type, ids.ctypes.data, len(ids), out.ctypes.data) == 0
    return out


def set_visible(obj_type, oid, vis):
    assert host().hs_set_visible(obj_type, int(oid), int(vis)) == 0


def filter_ptrs(rule, ptrs, on_page=None):
    ptrs = np.ascontiguousarray(ptrs, np.uint64)
    out = np.zeros(len(ptrs), np.uint8)
    op = None if on_page is None else np.ascontiguousarray(on_page, np.uint8)
    assert host().hs_filter_ptrs(rule, ptrs.ctypes.data, len(ptrs), None if op is None else op.ctypes.data, out.ctypes.data) == 0
    return out


def synth_world(n_ways, refs_per_way, pairs, tags_per_way=3, node_tag_every=50, seed=1):
    """(node pointers, way pointers) of a synthetic world kept outside the object tree; `pairs` must stay alive with it"""
    K = (C.c_char_p * len(pairs))(*[k.encode() for k, _ in pairs])
    V = (C.c_char_p * len(pairs))(*[v.encode() for _, v in pairs])
    nodes = np.zeros(n_ways * (refs_per_way - 1), np.uint64)
    ways = np.zeros(n_ways, np.uint64)
    assert host().hs_synth_world(n_ways, refs_per_way, K, V, len(pairs), tags_per_way, node_tag_every, seed, nodes.ctypes.data, ways.ctypes.data) == 0
    return nodes, ways, (K, V)
