"""ctypes binding of libsmr_hip.so (include/smr_hip.h).  Mirrors the C-ABI one to one."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("SMR_HIP_LIBRARY") or os.path.join(_HERE, "libsmr_hip.so")      # the override is for diagnostic builds


class SmrHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsmr_hip error {code}: {msg}")
        self.code = code


class Projection(C.Structure):
    """smr_proj_t: the members of struct rdata (src/rdata.h:99-158) geo2pxf reads."""
    _fields_ = [("lon0", C.c_double), ("w", C.c_double), ("h", C.c_double), ("wc", C.c_double),
                ("lath", C.c_double), ("lath_len", C.c_double), ("lat_ll", C.c_double),
                ("lon_ru", C.c_double), ("lat_ru", C.c_double), ("dpi", C.c_int32), ("pad_", C.c_int32)]

    @classmethod
    def from_window(cls, mean_lat, mean_lon, scale, w_px, h_px, dpi):
        p = cls()
        _chk(lib().smr_hip_proj_from_window(C.byref(p), mean_lat, mean_lon, scale, w_px, h_px, dpi))
        return p


class DrawStyle(C.Structure):
    """smr_draw_style_t: struct actDraw (src/smrender_dev.h:252-268) without image/curve."""
    _fields_ = [("fill_used", C.c_int32), ("border_used", C.c_int32), ("fill_col", C.c_uint32),
                ("border_col", C.c_uint32), ("fill_width_mm", C.c_double), ("border_width_mm", C.c_double),
                ("ignore_open", C.c_int32), ("fill_style", C.c_int32), ("border_style", C.c_int32),
                ("curve", C.c_int32), ("directional", C.c_int32), ("direct", C.c_int32),
                ("border_ndash", C.c_int32), ("fill_ndash", C.c_int32),
                ("border_dash", C.c_double * 4), ("fill_dash", C.c_double * 4), ("curve_factor", C.c_double),
                ("evenodd", C.c_int32), ("pad_", C.c_int32)]

    def set_evenodd(self, on=1):
        """closed ways filled with libcairo's EVEN_ODD rule (not selectable in the reference: see include/smr_hip.h)"""
        self.evenodd = int(on)
        return self

    def set_dashes(self, border=(), fill=()):
        self.border_ndash, self.fill_ndash = len(border), len(fill)
        for i, v in enumerate(border): self.border_dash[i] = v
        for i, v in enumerate(fill): self.fill_dash[i] = v
        return self

    def set_curve(self, factor=0.2):
        """curve=1 with curve_factor (default DIV_PART, src/smrender_dev.h:114)"""
        self.curve, self.curve_factor = 1, factor
        return self

    def set_knots(self):
        """curve=2: every way's refs are p0 and (c1, c2, p) triples, the path cairo_smr_wavy records (wavy=1)"""
        self.curve = 2
        return self


class _DrawBatch(C.Structure):
    _fields_ = [("d_x", C.c_void_p), ("d_y", C.c_void_p), ("n_nodes", C.c_int64), ("d_way_off", C.c_void_p),
                ("d_ref", C.c_void_p), ("n_ways", C.c_int64), ("n_refs", C.c_int64),
                ("d_way_fill_col", C.c_void_p), ("d_way_border_col", C.c_void_p)]


class _CapBatch(C.Structure):
    _fields_ = [("d_cx", C.c_void_p), ("d_cy", C.c_void_p), ("n", C.c_int64), ("fg_w", C.c_int32), ("fg_h", C.c_int32),
                ("fg_x", C.c_int32), ("col", C.c_uint32), ("phase_deg", C.c_double), ("weight", C.c_double), ("d_fg", C.c_void_p)]


class Timings(C.Structure):
    _fields_ = [("proj_ms", C.c_float), ("edges_ms", C.c_float), ("prims_ms", C.c_float), ("bin_ms", C.c_float),
                ("raster_ms", C.c_float), ("caprot_ms", C.c_float), ("total_ms", C.c_float),
                ("n_prims", C.c_uint64), ("n_edges", C.c_uint64), ("n_tile_refs", C.c_uint64), ("n_tiles", C.c_uint64),
                ("cover_ms", C.c_float), ("n_cover_prims", C.c_uint32), ("n_cover_left", C.c_uint32), ("n_deltas", C.c_uint32), ("n_cover_linear", C.c_uint32)]


class PngStats(C.Structure):
    _fields_ = [("n_files", C.c_uint64), ("n_opaque", C.c_uint64), ("bytes_filtered", C.c_uint64), ("bytes_written", C.c_uint64),
                ("deflate_ms", C.c_float), ("total_ms", C.c_float), ("threads", C.c_int32), ("pad_", C.c_int32)]


class KapStats(C.Structure):
    _fields_ = [("n_colors", C.c_int32), ("depth", C.c_int32), ("palette", C.c_uint32 * 127), ("row_bytes", C.c_int64), ("file_bytes", C.c_int64),
                ("sorted_points", C.c_int64), ("reduce_ms", C.c_double), ("index_ms", C.c_double), ("rows_ms", C.c_double), ("write_ms", C.c_double)]


class ObjectsInfo(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("n_nodes", "n_ways", "n_refs", "n_node_tags", "n_way_tags", "n_strings", "pool_bytes", "missing_refs")] + \
               [(k, C.c_void_p) for k in ("d_node_id", "d_lat", "d_lon", "d_way_id", "d_way_off", "d_ref")] + \
               [(k, C.c_double) for k in ("extract_ms", "upload_ms", "index_ms", "match_host_ms", "match_device_ms")]


class WaySelection(C.Structure):
    _fields_ = [("n_ways", C.c_int64), ("n_refs", C.c_int64), ("d_way_off", C.c_void_p), ("d_ref", C.c_void_p), ("d_way_index", C.c_void_p)]


class _Windows(C.Structure):
    _fields_ = [("n", C.c_int64), ("size", C.c_void_p), ("r_px", C.c_void_p), ("cx", C.c_void_p), ("cy", C.c_void_p)]


# every symbol include/smr_hip.h declares (tests/test_abi.py checks the list against the header)
EXPORTS = [
    "smr_hip_init", "smr_hip_shutdown", "smr_hip_last_error", "smr_hip_device_name",
    "smr_hip_proj_from_window", "smr_hip_set_projection", "smr_hip_project", "smr_hip_unproject",
    "smr_hip_canvas_create", "smr_hip_canvas_destroy", "smr_hip_canvas_clear", "smr_hip_canvas_trim", "smr_hip_canvas_read", "smr_hip_canvas_device",
    "smr_hip_canvas_set_halo", "smr_hip_canvas_set_halo_device", "smr_hip_canvas_extend", "smr_hip_canvas_clone", "smr_hip_canvas_blit",
    "smr_hip_canvas_write_png", "smr_hip_canvas_write_tiles", "smr_hip_canvas_write_kap",
    "smr_hip_windows_create", "smr_hip_windows_reset", "smr_hip_draw_windows", "smr_hip_windows_anchors", "smr_hip_blit_windows",
    "smr_hip_objects_create", "smr_hip_objects_destroy", "smr_hip_objects_info", "smr_hip_objects_match",
    "smr_hip_objects_select_ways", "smr_hip_objects_select_nodes",
    "smr_hip_poly_area", "smr_hip_draw", "smr_hip_draw_pattern", "smr_hip_draw_status", "smr_hip_debug_edges", "smr_hip_caprot", "smr_hip_caprot_noalloc", "smr_hip_caprot_steps", "smr_hip_sync", "smr_hip_stream",
    "smr_hip_set_profiling", "smr_hip_timings", "smr_hip_malloc", "smr_hip_free",
    "smr_hip_memcpy_h2d", "smr_hip_memcpy_d2h", "smr_hip_plugin_canvas", "smr_hip_plugin_reset", "smr_hip_plugin_stats",
    "act_draw_ini", "act_draw_main", "act_draw_fini", "act_cap_ini", "act_cap_main", "act_cap_fini",
    "act_img_ini", "act_img_main", "act_img_fini", "act_hipsave_ini", "act_hipsave_main", "act_hipsave_fini",
]

_lib = None


def lib():
    """Load libsmr_hip.so (built in-tree by `make`).  Raises if it is missing: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise SmrHipError(-1, f"{_LIB_PATH} not built (run `make` / __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(_LIB_PATH)
        L.smr_hip_last_error.restype = C.c_char_p
        L.smr_hip_stream.restype = C.c_void_p
        L.smr_hip_proj_from_window.argtypes = [C.POINTER(Projection), C.c_double, C.c_double, C.c_double,
                                               C.c_double, C.c_double, C.c_int]
        L.smr_hip_set_projection.argtypes = [C.POINTER(Projection)]
        L.smr_hip_project.argtypes = [C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 5
        L.smr_hip_unproject.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.smr_hip_canvas_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.smr_hip_canvas_destroy.argtypes = [C.c_void_p]
        L.smr_hip_canvas_clear.argtypes = [C.c_void_p]
        L.smr_hip_canvas_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.smr_hip_canvas_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.smr_hip_canvas_set_halo.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.smr_hip_canvas_set_halo_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.smr_hip_canvas_extend.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.smr_hip_canvas_blit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        L.smr_hip_poly_area.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.smr_hip_objects_create.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
        L.smr_hip_objects_destroy.argtypes = [C.c_void_p]
        L.smr_hip_objects_info.argtypes = [C.c_void_p, C.POINTER(ObjectsInfo)]
        L.smr_hip_objects_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.smr_hip_objects_select_ways.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(WaySelection)]
        L.smr_hip_objects_select_nodes.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.smr_hip_windows_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        L.smr_hip_draw_windows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.smr_hip_windows_anchors.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.smr_hip_windows_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.smr_hip_blit_windows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        L.smr_hip_canvas_write_png.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        L.smr_hip_canvas_write_tiles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int,
                                                 C.c_int, C.c_int, C.c_void_p]
        L.smr_hip_canvas_write_kap.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p]
        L.smr_hip_canvas_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        L.smr_hip_draw.argtypes = [C.c_void_p, C.POINTER(DrawStyle), C.POINTER(_DrawBatch)]
        L.smr_hip_draw_pattern.argtypes = [C.c_void_p, C.POINTER(DrawStyle), C.POINTER(_DrawBatch), C.c_void_p, C.c_int32, C.c_int32]
        L.smr_hip_debug_edges.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.smr_hip_caprot.argtypes = [C.c_void_p, C.POINTER(_CapBatch), C.c_void_p, C.c_void_p]
        L.smr_hip_caprot_noalloc.argtypes = [C.c_void_p, C.POINTER(_CapBatch), C.c_void_p, C.c_void_p]
        L.smr_hip_caprot_steps.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.smr_hip_timings.argtypes = [C.POINTER(Timings)]
        L.smr_hip_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.smr_hip_free.argtypes = [C.c_void_p]
        L.smr_hip_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.smr_hip_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.smr_hip_device_name.argtypes = [C.c_char_p, C.c_size_t]
        L.smr_hip_plugin_canvas.restype = C.c_void_p
        _lib = L
    return _lib


def _chk(code):
    if code != 0:
        raise SmrHipError(code, lib().smr_hip_last_error().decode())


def init(device=0):
    _chk(lib().smr_hip_init(int(device)))


def shutdown():
    _chk(lib().smr_hip_shutdown())


def sync():
    _chk(lib().smr_hip_sync())


def device_name():
    buf = C.create_string_buffer(256)
    _chk(lib().smr_hip_device_name(buf, 256))
    return buf.value.decode()


def set_profiling(on):
    _chk(lib().smr_hip_set_profiling(int(bool(on))))


def timings():
    t = Timings()
    _chk(lib().smr_hip_timings(C.byref(t)))
    return {k: getattr(t, k) for k, _ in Timings._fields_}


class DeviceArray:
    """A typed device allocation made through the C-ABI (smr_hip_malloc/_memcpy_*)."""

    def __init__(self, shape, dtype):
        self.shape = tuple(np.atleast_1d(shape).tolist()) if not isinstance(shape, tuple) else shape
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _chk(lib().smr_hip_malloc(C.byref(p), max(self.nbytes, 16)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        if a.nbytes:
            _chk(lib().smr_hip_memcpy_h2d(d.ptr, a.ctypes.data, a.nbytes))
        return d

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            _chk(lib().smr_hip_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().smr_hip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def set_projection(p):
    _chk(lib().smr_hip_set_projection(C.byref(p)))


def project(d_lat, d_lon, d_x, d_y, d_ix=None, d_iy=None, d_on=None):
    """geo2pxf over device SoA arrays (DeviceArray)."""
    n = int(np.prod(d_lat.shape))
    _chk(lib().smr_hip_project(d_lat.ptr, d_lon.ptr, n, d_x.ptr, d_y.ptr,
                               d_ix.ptr if d_ix else None, d_iy.ptr if d_iy else None, d_on.ptr if d_on else None))


def poly_area(d_lat, d_lon, d_way_off, d_ref, n_ways, d_area, d_clat=None, d_clon=None):
    """poly_area (src/smfunc.c:377-432) of every closed way: signed square nautical miles, centroid in degrees (smr_hip_poly_area)"""
    _chk(lib().smr_hip_poly_area(d_lat.ptr, d_lon.ptr, int(np.prod(d_lat.shape)), d_way_off.ptr, d_ref.ptr if d_ref else None, n_ways,
                                 d_area.ptr, d_clat.ptr if d_clat else None, d_clon.ptr if d_clon else None))


class _View:
    """a typed view of device memory that something else owns"""

    def __init__(self, ptr, shape, dtype):
        self.ptr, self.shape, self.dtype = ptr, tuple(np.atleast_1d(shape).tolist()), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            _chk(lib().smr_hip_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes))
        return out


class Objects:
    """The host's nodes and ways as structure-of-arrays on the device (smr_hip_objects_*): node_ptrs / way_ptrs are arrays of
    the host's own osm_node_t* / osm_way_t* (uint64)."""

    def __init__(self, node_ptrs, way_ptrs, threads=0):
        node_ptrs = np.ascontiguousarray(node_ptrs, np.uint64)
        way_ptrs = np.ascontiguousarray(way_ptrs, np.uint64)
        h = C.c_void_p()
        _chk(lib().smr_hip_objects_create(node_ptrs.ctypes.data, len(node_ptrs), way_ptrs.ctypes.data, len(way_ptrs), int(threads), C.byref(h)))
        self.h = h
        self.info = ObjectsInfo()
        _chk(lib().smr_hip_objects_info(self.h, C.byref(self.info)))
        i = self.info
        self.words = (max(i.n_nodes, i.n_ways) + 63) // 64
        self.lat, self.lon = _View(i.d_lat, (i.n_nodes,), np.float64), _View(i.d_lon, (i.n_nodes,), np.float64)
        self.node_id, self.way_id = _View(i.d_node_id, (i.n_nodes,), np.int64), _View(i.d_way_id, (i.n_ways,), np.int64)
        self.way_off, self.ref = _View(i.d_way_off, (i.n_ways + 1,), np.int64), _View(i.d_ref, (i.n_refs,), np.int32)

    def match(self, rules, d_on_page=None):
        """rules: the host's smrule_t* (ints).  Returns (bitmap DeviceArray [n_rules, words] of uint64, counts)."""
        R = np.ascontiguousarray(rules, np.uint64)
        bm = DeviceArray((len(R), max(self.words, 1)), np.uint64)
        counts = np.zeros(len(R), np.int64)
        _chk(lib().smr_hip_objects_match(self.h, R.ctypes.data, len(R), d_on_page.ptr if d_on_page else None, bm.ptr, max(self.words, 1), counts.ctypes.data))
        _chk(lib().smr_hip_objects_info(self.h, C.byref(self.info)))
        return bm, counts

    def select_ways(self, bitmap, row):
        sel = WaySelection()
        _chk(lib().smr_hip_objects_select_ways(self.h, bitmap.ptr + row * bitmap.shape[1] * 8, C.byref(sel)))
        return (_View(sel.d_way_off, (sel.n_ways + 1,), np.int64), _View(sel.d_ref, (sel.n_refs,), np.int32),
                _View(sel.d_way_index, (sel.n_ways,), np.int32), int(sel.n_ways), int(sel.n_refs))

    def select_nodes(self, bitmap, row):
        p, n = C.c_void_p(), C.c_int64()
        _chk(lib().smr_hip_objects_select_nodes(self.h, bitmap.ptr + row * bitmap.shape[1] * 8, C.byref(p), C.byref(n)))
        return _View(p.value, (n.value,), np.int32)

    def close(self):
        if self.h:
            lib().smr_hip_objects_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def unproject(d_x, d_y, d_lon, d_lat):
    n = int(np.prod(d_x.shape))
    _chk(lib().smr_hip_unproject(d_x.ptr, d_y.ptr, n, d_lon.ptr, d_lat.ptr))


def caprot_steps(fg_w, fg_h, dpi):
    n, r = C.c_int(), C.c_int()
    _chk(lib().smr_hip_caprot_steps(fg_w, fg_h, dpi, C.byref(n), C.byref(r)))
    return n.value, r.value


class Canvas:
    """Device canvas band: the GPU counterpart of smrender's recording surface + replay."""

    def __init__(self, W, H, dpi, bgcolor=0xffffff, band_y0=0, band_h=None):
        self.W, self.H, self.dpi = W, H, dpi
        self.band_y0 = band_y0
        self.band_h = H - band_y0 if band_h is None else band_h
        h = C.c_void_p()
        _chk(lib().smr_hip_canvas_create(W, H, dpi, bgcolor, self.band_y0, self.band_h, C.byref(h)))
        self.h = h.value
        self._n_above = self._n_below = 0

    def draw(self, style, d_x, d_y, d_way_off, d_ref, n_ways, n_refs, d_fill_col=None, d_border_col=None):
        b = _DrawBatch(d_x.ptr, d_y.ptr, int(np.prod(d_x.shape)), d_way_off.ptr, d_ref.ptr if d_ref else None,
                       n_ways, n_refs, d_fill_col.ptr if d_fill_col else None, d_border_col.ptr if d_border_col else None)
        _chk(lib().smr_hip_draw(self.h, C.byref(style), C.byref(b)))

    def draw_pattern(self, style, pattern, d_x, d_y, d_way_off, d_ref, n_ways, n_refs):
        """the image fill of a rule (smr_hip_draw_pattern): pattern = [h, w] uint32 premultiplied ARGB32 pixels in host memory"""
        pattern = np.ascontiguousarray(pattern, np.uint32)
        b = _DrawBatch(d_x.ptr, d_y.ptr, int(np.prod(d_x.shape)), d_way_off.ptr, d_ref.ptr if d_ref else None, n_ways, n_refs, None, None)
        _chk(lib().smr_hip_draw_pattern(self.h, C.byref(style), C.byref(b), pattern.ctypes.data, pattern.shape[1], pattern.shape[0]))

    @classmethod
    def windows(cls, chart_w, chart_h, dpi, bgcolor, cx, cy, size, r_px=None):
        """an ATLAS of caption cut-outs the reference's way (smr_hip_windows_create): one window per anchor (chart px, host
        arrays) of `size` px (one int or one per window; r_px: find_angle's r before rounding, hypot of the caption plane),
        transparent but for the chart's background; draw_windows() then renders rules into it"""
        cx, cy = np.ascontiguousarray(cx, np.float64), np.ascontiguousarray(cy, np.float64)
        sizes = np.ascontiguousarray(np.broadcast_to(np.asarray(size, np.int32), cx.shape))
        rp = None if r_px is None else np.ascontiguousarray(np.broadcast_to(np.asarray(r_px, np.float64), cx.shape))
        w = _Windows(len(cx), sizes.ctypes.data, None if rp is None else rp.ctypes.data, cx.ctypes.data, cy.ctypes.data)
        h = C.c_void_p()
        _chk(lib().smr_hip_windows_create(chart_w, chart_h, dpi, bgcolor, C.byref(w), C.byref(h)))
        o = cls.__new__(cls)
        pitch = (int(sizes.max()) + 3) & ~3
        o.W, o.H, o.dpi, o.band_y0, o.band_h, o.h = int(sizes.max()), len(cx) * pitch, dpi, 0, len(cx) * pitch, h.value
        o._n_above = o._n_below = 0
        o.n_windows, o.window_sizes, o.window_pitch = len(cx), sizes, pitch
        return o

    def reset_windows(self, cx, cy, size, r_px=None):
        """the same atlas for another batch of anchors (smr_hip_windows_reset): no more windows than it was created for, the
        same size; nothing is allocated or freed"""
        cx, cy = np.ascontiguousarray(cx, np.float64), np.ascontiguousarray(cy, np.float64)
        sizes = np.ascontiguousarray(np.broadcast_to(np.asarray(size, np.int32), cx.shape))
        rp = None if r_px is None else np.ascontiguousarray(np.broadcast_to(np.asarray(r_px, np.float64), cx.shape))
        w = _Windows(len(cx), sizes.ctypes.data, None if rp is None else rp.ctypes.data, cx.ctypes.data, cy.ctypes.data)
        _chk(lib().smr_hip_windows_reset(self.h, C.byref(w)))
        self.n_windows, self.window_sizes = len(cx), sizes

    def draw_windows(self, style, d_x, d_y, d_way_off, d_ref, n_ways, n_refs, d_fill_col=None, d_border_col=None):
        """the rule (as draw() took it for the chart) into every window of this atlas it reaches (smr_hip_draw_windows)"""
        b = _DrawBatch(d_x.ptr, d_y.ptr, int(np.prod(d_x.shape)), d_way_off.ptr, d_ref.ptr if d_ref else None,
                       n_ways, n_refs, d_fill_col.ptr if d_fill_col else None, d_border_col.ptr if d_border_col else None)
        _chk(lib().smr_hip_draw_windows(self.h, C.byref(style), C.byref(b)))

    def window_anchors(self):
        """the anchors to hand smr_hip_caprot for this atlas's windows (host arrays)"""
        cx, cy = np.zeros(self.n_windows), np.zeros(self.n_windows)
        _chk(lib().smr_hip_windows_anchors(self.h, cx.ctypes.data, cy.ctypes.data))
        return cx, cy

    def caprot(self, d_cx, d_cy, n, fg_w, fg_h, fg_x, col, phase_deg, weight, d_angle, d_dv=None, d_fg=None):
        b = _CapBatch(d_cx.ptr, d_cy.ptr, n, fg_w, fg_h, fg_x, col, phase_deg, weight, d_fg.ptr if d_fg else None)
        _chk(lib().smr_hip_caprot(self.h, C.byref(b), d_angle.ptr, d_dv.ptr if d_dv else None))

    def caprot_noalloc(self, d_cx, d_cy, n, fg_w, fg_h, fg_x, col, phase_deg, weight, d_angle, d_dv=None, d_fg=None):
        """smr_hip_caprot_noalloc: the same scan with its tables in the library's static device array (no hipMalloc)"""
        b = _CapBatch(d_cx.ptr, d_cy.ptr, n, fg_w, fg_h, fg_x, col, phase_deg, weight, d_fg.ptr if d_fg else None)
        _chk(lib().smr_hip_caprot_noalloc(self.h, C.byref(b), d_angle.ptr, d_dv.ptr if d_dv else None))

    def clone(self):
        """device copy (act_cap's auto_sfc)"""
        h = C.c_void_p()
        _chk(lib().smr_hip_canvas_clone(self.h, C.byref(h)))
        o = Canvas.__new__(Canvas)
        o.W, o.H, o.dpi, o.band_y0, o.band_h, o.h = self.W, self.H, self.dpi, self.band_y0, self.band_h, h.value
        o._n_above, o._n_below = self._n_above, self._n_below
        return o

    def set_halo(self, above=None, below=None):
        """neighbour bands' rows for the caption scan (uint32 [n, W] arrays or None)"""
        a = None if above is None else np.ascontiguousarray(above, np.uint32)
        b = None if below is None else np.ascontiguousarray(below, np.uint32)
        _chk(lib().smr_hip_canvas_set_halo(self.h, a.ctypes.data if a is not None else None, 0 if a is None else a.shape[0],
                                           b.ctypes.data if b is not None else None, 0 if b is None else b.shape[0]))
        self._n_above, self._n_below = 0 if a is None else a.shape[0], 0 if b is None else b.shape[0]

    def set_halo_device(self, d_above, n_above, d_below, n_below):
        """the same from DEVICE pointers (ints): the receive buffers of a device-to-device exchange"""
        _chk(lib().smr_hip_canvas_set_halo_device(self.h, d_above, n_above, d_below, n_below))
        self._n_above, self._n_below = n_above, n_below

    def extend(self):
        """band + halo strips as one canvas: the private background of a banded caption rule (smr_hip_canvas_extend)"""
        h = C.c_void_p()
        _chk(lib().smr_hip_canvas_extend(self.h, C.byref(h)))
        o = Canvas.__new__(Canvas)
        o.W, o.H, o.dpi, o.h = self.W, self.H, self.dpi, h.value
        o.band_y0, o.band_h = self.band_y0 - self._n_above, self.band_h + self._n_above + self._n_below
        o._n_above = o._n_below = 0
        return o

    def blit(self, images):
        """images: [(x, y, premultiplied ARGB32 array [h, w])], painted OVER the canvas in this order (smr_hip_canvas_blit)"""
        items = np.zeros(len(images), np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("offset", "<u8")]))
        off = 0
        for k, (x, y, im) in enumerate(images):
            items[k] = (x, y, im.shape[1], im.shape[0], off)
            off += im.size
        pix = np.concatenate([np.ascontiguousarray(im, np.uint32).ravel() for _, _, im in images]) if images else np.zeros(0, np.uint32)
        _chk(lib().smr_hip_canvas_blit(self.h, pix.ctypes.data, len(pix), items.ctypes.data, len(images)))

    def write_png(self, path, rect=None, level=-1, threads=0):
        """the rectangle (x, y, w, h; band-local rows; default: the whole band) as a PNG file whose decoded pixels are those of
        libcairo's writer for the same data (smr_hip_canvas_write_png); returns the call's PngStats"""
        x, y, w, h = rect if rect else (0, 0, self.W, self.band_h)
        st = PngStats()
        _chk(lib().smr_hip_canvas_write_png(self.h, x, y, w, h, str(path).encode(), level, threads, C.byref(st)))
        return st

    def write_kap(self, path, header=""):
        """the KAP body (palette, run-length rows, offset table) behind `header` (smr_hip_canvas_write_kap); returns KapStats"""
        st = KapStats()
        _chk(lib().smr_hip_canvas_write_kap(self.h, header.encode() if header else None, str(path).encode(), C.byref(st)))
        return st

    def write_tiles(self, directory, x, y, nx, ny, tile_px=256, tx0=0, ty0=0, level=-1, threads=0):
        """nx x ny tiles of tile_px pixels from band pixel (x, y) on, as <directory>/<tx0 + i>/<ty0 + j>.png (smr_hip_canvas_write_tiles)"""
        st = PngStats()
        _chk(lib().smr_hip_canvas_write_tiles(self.h, x, y, nx, ny, tile_px, str(directory).encode(), tx0, ty0, level, threads, C.byref(st)))
        return st

    def device_ptr(self):
        """(device address of the band's first pixel, row stride in pixels)"""
        p, st = C.c_void_p(), C.c_int()
        _chk(lib().smr_hip_canvas_device(self.h, C.byref(p), C.byref(st)))
        return p.value, st.value

    def device_rows(self, row0, nrows):
        """rows [row0, row0 + nrows) of the band as a torch int32 tensor VIEW of the device memory (no copy).
        torch must have been imported BEFORE the library initialised HIP (smrender_amd.dist.Group does so): torch ships a HIP
        runtime of its own, and of two runtimes in one process only the first sees the GPU."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("device_rows: torch sees no GPU (import torch before smrender_amd.init(): two HIP runtimes in one process)")
        ptr, stride = self.device_ptr()

        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": (nrows, self.W), "typestr": "<i4", "data": (ptr + 4 * stride * row0, False),
                                      "version": 2, "strides": None}
        return torch.as_tensor(v, device="cuda")

    def debug_edges(self):
        """edge slots of the last draw pass, int32 [n, 4] (diagnostics)"""
        n = C.c_uint64()
        _chk(lib().smr_hip_debug_edges(self.h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 4), np.int32)
        if n.value:
            _chk(lib().smr_hip_debug_edges(self.h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def clear(self):
        _chk(lib().smr_hip_canvas_clear(self.h))

    def read(self, x=0, y=0, w=None, h=None):
        w = self.W - x if w is None else w
        h = self.band_h - y if h is None else h
        out = np.empty((h, w), np.uint32)
        _chk(lib().smr_hip_canvas_read(self.h, x, y, w, h, out.ctypes.data))
        return out

    def close(self):
        if self.h:
            lib().smr_hip_canvas_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
