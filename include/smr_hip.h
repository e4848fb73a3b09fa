/* smr_hip.h -- C-ABI of libsmr_hip.so, the MI355X (gfx950) back end for smrender's
 * projection -> draw (raster) -> caption-scan hot path.
 *
 * Every entry point is extern "C", takes plain pointers and sizes, returns 0 on success
 * and a negative SMR_E_* code on failure (smr_hip_last_error() gives the text).  There
 * is no CPU fallback: without a usable GPU every compute entry point fails with
 * SMR_E_NODEVICE.  One process drives one GPU (multi-GPU = one process per GPU, each
 * owning a band of canvas rows).
 *
 * Each declaration cites the reference interface (file:line under the smrender tree)
 * it replaces.  INTEGRATION.md shows the plugin-side binding.
 */
#ifndef SMR_HIP_H
#define SMR_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum
{
   SMR_OK = 0,
   SMR_E_NODEVICE = -1,     /* no HIP device / init not called */
   SMR_E_INVAL = -2,        /* bad argument */
   SMR_E_NOMEM = -3,        /* hipMalloc failed */
   SMR_E_HIP = -4,          /* a HIP runtime call failed */
   SMR_E_UNSUPPORTED = -5,  /* style feature not implemented on the GPU path (fails loudly) */
   SMR_E_CAPACITY = -6,     /* a fixed device-side capacity was exceeded (see DESIGN.md limits) */
};

/* ---- life cycle ------------------------------------------------------------------ */
int smr_hip_init(int device);                 /* selects the device, creates the stream (and a second one the draw pass forks its binning onto
                                                * and joins again before it returns: callers only ever see the first) */
int smr_hip_shutdown(void);
const char *smr_hip_last_error(void);
int smr_hip_device_name(char *buf, size_t len);

/* ---- projection: replaces geo2pxf/geo2pt/geo2pxi + is_on_page -------------------------
 * src/rdata.c:188-230 (Mercator branch :221-225), :240-255, :482-487; constants are the
 * ones init_bbox_mll (src/smrender.c:278-311) leaves in struct rdata (src/rdata.h:99-158). */
typedef struct
{
   double lon0;              /* rd->bb.ll.lon */
   double w, h;              /* rd->w, rd->h  (pixels, double) */
   double wc;                /* rd->wc */
   double lath, lath_len;    /* rd->lath, rd->lath_len */
   double lat_ll, lon_ru, lat_ru;   /* rd->bb (is_on_page) */
   int32_t dpi;              /* rd->dpi */
   int32_t pad_;
} smr_proj_t;

/* restates init_bbox_mll (src/smrender.c:278-311) + mean_lat_len from scale (:697) on the host */
int smr_hip_proj_from_window(smr_proj_t *p, double mean_lat, double mean_lon, double scale,
                             double w_px, double h_px, int dpi);
int smr_hip_set_projection(const smr_proj_t *p);

/* Batched geo2pxf over SoA node arrays.  All pointers are DEVICE pointers; x,y are
 * required, ix/iy (C round() of x,y: src/rdata.c:253-254) and onpage (is_on_page) may be
 * NULL.  Runs on the library stream; returns after enqueueing (smr_hip_sync() waits). */
int smr_hip_project(const double *d_lat, const double *d_lon, int64_t n,
                    double *d_x, double *d_y, int32_t *d_ix, int32_t *d_iy, uint8_t *d_onpage);
/* inverse, src/rdata.c:68-72 (pxf2geo) */
int smr_hip_unproject(const double *d_x, const double *d_y, int64_t n, double *d_lon, double *d_lat);

/* ---- canvas: replaces the recording surface sfc_ + its replay into ARGB32 --------------
 * src/smrules_cairo.c:308-330 (cairo_smr_init_main_image: bg paint),
 * :339-356 (cairo_smr_image_surface_from_bg: the pixels), :568-580 (get_pixel). */
typedef struct smr_canvas smr_canvas_t;

/* Full chart is W x H px at `dpi`; this process owns rows [band_y0, band_y0+band_h).
 * bgcolor uses the reference's colour ints 0xTTRRGGBB (src/smrparse.c:222-258). */
int smr_hip_canvas_create(int W, int H, int dpi, uint32_t bgcolor, int band_y0, int band_h, smr_canvas_t **out);
int smr_hip_canvas_destroy(smr_canvas_t *c);
/* a second canvas with the same geometry and pixels: the device counterpart of
 * cairo_smr_recording_surface_from_bg() (src/smrules_cairo.c:359-373), act_cap's auto_sfc (:2455-2460) */
int smr_hip_canvas_clone(smr_canvas_t *c, smr_canvas_t **out);
/* repaint the background (what cairo_smr_init_main_image does once per chart, :320-326) */
int smr_hip_canvas_clear(smr_canvas_t *c);
/* gives back the canvas's grow-only work buffers (edge arenas, bins: several times the size of a small canvas after a large rule);
 * the pixels stay, the next draw allocates again.  For canvases that wait: the atlases of a caption rule between its levels */
int smr_hip_canvas_trim(smr_canvas_t *c);
/* copy a rectangle (band-local rows) of premultiplied ARGB32 to host memory */
int smr_hip_canvas_read(smr_canvas_t *c, int x, int y, int w, int h, uint32_t *host_dst);
/* Multi-GPU caption scan: rows of the neighbouring bands the scan may read (cut-outs of
 * radius r_px around captions near a band boundary).  `above` = n_above rows ending at row band_y0-1,
 * `below` = n_below rows starting at row band_y0+band_h; host pointers, W pixels per row; the library
 * keeps device copies until the next call (n = 0 drops them).  The exchange itself is the caller's
 * (smrender_amd/dist.py: send/recv over RCCL). */
int smr_hip_canvas_set_halo(smr_canvas_t *c, const uint32_t *above, int n_above, const uint32_t *below, int n_below);
/* the same with the strips in DEVICE memory (e.g. the receive buffers of a device-to-device send/recv): no host round trip */
int smr_hip_canvas_set_halo_device(smr_canvas_t *c, const uint32_t *d_above, int n_above, const uint32_t *d_below, int n_below);
/* A second canvas whose band is this band plus its halo strips, pixels copied: the rule-private background of a banded
 * caption rule (act_cap's auto_sfc, src/smrules_cairo.c:2455-2460).  The rectangles of finished captions (:2376-2382), the
 * neighbouring ranks' included, are drawn onto it with smr_hip_draw (direct = 1); smr_hip_caprot on it scans the captions
 * anchored in the ORIGINAL band's rows. */
int smr_hip_canvas_extend(smr_canvas_t *c, smr_canvas_t **out);
/* Images painted OVER the canvas, in order: what the host's libcairo / FreeType / rsvg rasterise and the reference records into
 * its chart -- the glyphs of a caption (cairo_show_text, src/smrules_cairo.c:2384-2388) and the symbols of act_img
 * (img_place, :2754-2826) -- so that the device canvas holds the whole chart.  `pixels` is host memory,
 * premultiplied ARGB32; item k is the w x h image at pixels + offset (row stride w) whose top-left pixel lands on chart
 * pixel (x, y); parts outside the band are clipped.  Items are composited one after the other (later over earlier). */
typedef struct { int32_t x, y, w, h; uint64_t offset; } smr_blit_t;
int smr_hip_canvas_blit(smr_canvas_t *c, const uint32_t *pixels, uint64_t n_pixels, const smr_blit_t *items, int n_items);
/* PNG files straight from the device canvas.  canvas_write_png replaces the image branch of save_main_image
 * (src/smrules_cairo.c:398-412: the chart as an ARGB32 image handed to cairo_surface_write_to_png_stream) for the unrotated page;
 * canvas_write_tiles replaces the writer of create_tiles (src/smtile.c:186-222: clear_tile, cut_tile, save_image per tile,
 * file <dir>/<x>/<y>.png below the zoom directory) for a canvas that was rendered ON the tile grid: nx x ny tiles of tile_px
 * pixels whose first one has its corner at band pixel (x, y) and is named tx0/ty0.  (The reference re-rasterises its recording
 * per tile at TILE_SIZE / tile width in pt; here the caller renders the zoom level as one canvas at that scale and this call
 * cuts it.)  The decoded pixels are those of libcairo's PNG writer for the same ARGB32 data: all alpha 255 -> 8-bit RGB, else
 * RGBA un-premultiplied as (c*255 + a/2)/a.  The un-premultiply, the choice among PNG's five row filters and the filtering run
 * on the device; the filtered rows cross PCIe once; `threads` host workers (0: min(16, cores)) deflate them with zlib at `level`
 * (-1: zlib's default, as libpng's).  Rows are band-local.  `st` may be NULL.
 * level SMR_PNG_DEVICE_DEFLATE: the rows are deflated on the device too, every strip of 64 MB of a PNG, or every tile, as one block
 * with a Huffman code of its own over literals and runs (matches of distance 1: what PNG's filters leave of flat colour); only the
 * compressed bytes cross PCIe, the workers compute the IDAT chunks' CRCs and write the files.  Valid zlib streams like any, a few
 * percent larger than zlib's on a busy chart, in a tenth of the time. */
#define SMR_PNG_DEVICE_DEFLATE 100
typedef struct
{
   uint64_t n_files, n_opaque;             /* files written; how many of them are RGB (no alpha channel needed) */
   uint64_t bytes_filtered, bytes_written; /* deflate input (filter bytes + residuals) and file bytes */
   float deflate_ms, total_ms;             /* wall time inside the host's deflate workers / of the whole call */
   int32_t threads, pad_;
} smr_png_stats_t;
int smr_hip_canvas_write_png(smr_canvas_t *c, int x, int y, int w, int h, const char *path, int level, int threads, smr_png_stats_t *st);
int smr_hip_canvas_write_tiles(smr_canvas_t *c, int x, int y, int nx, int ny, int tile_px, const char *dir, int tx0, int ty0,
                               int level, int threads, smr_png_stats_t *st);
/* device pointer + row stride (pixels) of the band, for zero-copy consumers */
/* ---- KAP (BSB raster chart) from the device canvas: replaces save_kap from "OST/1" on -----------------------
 * src/smkap.c:196-245 with cairo_smr_image_surface_color_reduce (src/median_cut.c:293-350: median cut to at most 127 colours,
 * nearest colour per pixel) and bsb_compress_row (src/smkap.c:70-114).  `header` = the text gen_kap_header (:143-167) writes
 * (the caller's: title, scale, REF/PLY corners; may be NULL), then this call writes the palette lines, the rows and the
 * offset table.  The bytes equal the reference's for a libc whose qsort is stable (glibc <= 2.36: the cut depends on the
 * order of equal points).  Whole canvases only, at most 65535 x 65535 px. */
typedef struct
{
   int32_t n_colors, depth;
   uint32_t palette[127];          /* 0xffRRGGBB, in palette order (RGB/1 first) */
   int64_t row_bytes, file_bytes, sorted_points;
   double reduce_ms, index_ms, rows_ms, write_ms;
} smr_kap_stats_t;
int smr_hip_canvas_write_kap(smr_canvas_t *c, const char *header, const char *path, smr_kap_stats_t *st);
int smr_hip_canvas_device(smr_canvas_t *c, uint32_t **d_pix, int *stride_px);

/* ---- poly_area: replaces poly_area() per way --------------------------------------------
 * src/smfunc.c:377-432: signed area in square nautical miles (> 0: counterclockwise, < 0: clockwise) and centroid of every
 * CLOSED way of a batch given in degrees, sums in the reference's order.  It is what act_draw_fini sorts the ways of a
 * directional rule by (src/smrules_cairo.c:1087-1099), cap_way places a caption with (:2395-2444) and the coastline closing
 * orients its polygons with.  d_ref NULL = identity; an open way gets area 0 and a NaN centroid (the
 * reference returns -1 for it); d_clat / d_clon may be NULL.  The formula cancels about nine digits for chart-sized ways, so the
 * last bit of cos() shows as ~1e-10 square miles of area whatever the way's size (and in the centroid divided by the area: metres
 * for a 100 m islet): the device's values agree with the host's that far (sign and order of the areas are what act_draw_fini uses);
 * the plugin keeps the host's poly_area for caption anchors. */
int smr_hip_poly_area(const double *d_lat, const double *d_lon, int64_t n_nodes, const int64_t *d_way_off, const int32_t *d_ref,
                      int64_t n_ways, double *d_area, double *d_clat, double *d_clon);

/* ---- objects: the host's object tree as structure-of-arrays, and the rule executor's filter on the device -------------------
 * SURVEY 8f rank 3.  Replaces, for many rules at once, what apply_smrules does per rule and object (src/smcore.c:424-491:
 * traverse -> apply_rule :248-312 -> bs_match_attr, libsmrender/smutil.c:330-356 -> `_main`): one export of the objects
 * (smr_hip_objects_create: `nodes` / `ways` are the host's own osm_node_t* / osm_way_t*, e.g. gathered by one traverse() of the
 * bxtree, in the order the executor visits them), then per batch of rules one bitmap row per rule with a bit for every object
 * apply_rule would have called `_main` for, then per rule the selected ways as a smr_draw_batch_t's arrays or the selected
 * nodes as indices.  Tags are dictionary encoded (one id per distinct byte string); way refs become node indices on the
 * device (a ref whose node is not among `nodes` becomes -1, and selections pass over it as poly_line does,
 * src/smrules_cairo.c:853-857).  `nodes` must hold every node the ways refer to. */
typedef struct smr_objects smr_objects_t;
typedef struct
{
   int64_t n_nodes, n_ways, n_refs, n_node_tags, n_way_tags, n_strings, pool_bytes, missing_refs;
   const int64_t *d_node_id; const double *d_lat, *d_lon;                    /* [n_nodes] device: input of smr_hip_project */
   const int64_t *d_way_id; const int64_t *d_way_off; const int32_t *d_ref;  /* [n_ways], [n_ways+1], [n_refs] device */
   double extract_ms, upload_ms, index_ms;                                   /* host walk + dictionary, copies, ref look-up */
   double match_host_ms, match_device_ms;                                    /* the last smr_hip_objects_match: patterns + regexec on the host; copies, kernels, counts */
} smr_objects_info_t;
int smr_hip_objects_create(const void *const *nodes, int64_t n_nodes, const void *const *ways, int64_t n_ways, int threads, smr_objects_t **out);
int smr_hip_objects_destroy(smr_objects_t *o);
int smr_hip_objects_info(const smr_objects_t *o, smr_objects_info_t *info);
/* rules[r] = the host's smrule_t* after init_rules (match tags parsed by parse_matchtag, src/smrparse.c:150-159: patterns in
 * oo->otag[], kinds in act->stag[]).  Row r of d_bitmap (words_per_rule 64-bit words, >= (max(n_nodes, n_ways) + 63) / 64;
 * bit i of word j = object 64 j + i of the rule's type) gets a 1 where apply_rule reaches the call of `_main`: the node is on
 * the page (d_node_on_page = smr_hip_project's onpage output, NULL = render_all_nodes), an ACTION_CLOSED_WAY / ACTION_OPEN_WAY
 * rule meets a closed / open way, every match tag finds its tag (direct, /regex/, ]greater[, [less], !inverted!, ~not~), the
 * object is visible.  counts[r] (HOST) = the number of 1s; -1 for a relation rule (not handled: its row is all 0).
 * Regular expressions run on the host, once per distinct string, through regexec() on the rule's own regex_t. */
int smr_hip_objects_match(smr_objects_t *o, const void *const *rules, int n_rules, const uint8_t *d_node_on_page,
                          uint64_t *d_bitmap, int64_t words_per_rule, int64_t *counts);
/* the ways / nodes of one bitmap row, in object order.  The arrays belong to `o` and hold until its next selection */
typedef struct
{
   int64_t n_ways, n_refs;
   const int64_t *d_way_off;        /* [n_ways+1] device */
   const int32_t *d_ref;            /* [n_refs] device: node indices, missing nodes dropped */
   const int32_t *d_way_index;      /* [n_ways] device: which way of the export each one is */
} smr_way_selection_t;
int smr_hip_objects_select_ways(smr_objects_t *o, const uint64_t *d_bitmap_row, smr_way_selection_t *sel);
int smr_hip_objects_select_nodes(smr_objects_t *o, const uint64_t *d_bitmap_row, const int32_t **d_index, int64_t *n);

/* ---- draw: replaces act_draw_ini/_main/_fini for one rule --------------------------------
 * src/smrules_cairo.c:583-694 (style), :845-863 (path), :880-895 (widths), :953-1016
 * (render_poly_line: border stroke then fill / open stroke), :1019-1073, :1076-1148
 * (group pop + paint).  One call == one rule == one pushed group layer. */
typedef struct
{
   int32_t fill_used, border_used;          /* color= / bcolor= given */
   uint32_t fill_col, border_col;           /* 0xTTRRGGBB */
   double fill_width_mm, border_width_mm;   /* width= / bwidth= (0 -> 1 px hairline) */
   int32_t ignore_open;                     /* ignore_open=1 */
   int32_t fill_style, border_style;        /* draw_style_t, src/smrender_dev.h:145 (SMR_STYLE_*): style= / bstyle= */
   int32_t curve;                           /* 1: curve=1, the way as a chain of Beziers (cairo_smr_poly_curve, src/smrules_cairo.c:697-749;
                                               control_points, src/bspline_ctrl.c:64-105), solid strokes only; 2: every way's refs are
                                               explicit Bezier knots p0, (c1, c2, p) * n: the move_to + curve_to path that cairo_smr_wavy
                                               records for wavy=1 (:752-840; the caller resamples the way); one more ref
                                               equal to the first marks the way as closed (filled: cairo_fill closes the open path itself,
                                               the border stroke does not) */
   int32_t directional;                     /* directional=1 with a fill: open ways first, then ALL closed ways of the batch as one
                                               non-zero fill in fill_col (act_draw_fini, src/smrules_cairo.c:1084-1140); the caller
                                               appends the page rectangle (:1103-1109) as one more closed way when it is due */
   int32_t direct;                          /* 1: draw straight onto the canvas, no group layer -- what a context without
                                               push_group does, i.e. act_cap's auto_ctx rectangles (src/smrules_cairo.c:2376-2382) */
   /* dash= / bdash= (src/smrparse.c:1077-1102): up to 4 lengths in multiples of the BORDER width (cairo_smr_dash,
    * src/smrules_cairo.c:898-923, scales the fill's dashes by it too); 0 lengths = the style's default array
    * (dashed/pipe 7,3; dotted 1; rounddot 0,2; solid none).  Dashed strokes take libcairo's dashing stroker
    * (butt caps; round caps for rounddot and for the pipe's dots, which stay on for every later stroke of the rule) */
   int32_t border_ndash, fill_ndash;
   double border_dash[4], fill_dash[4];
   double curve_factor;                     /* curve_factor=, the reference's default is DIV_PART 0.2 (src/smrender_dev.h:114; act_draw_ini :625-631) */
   int32_t evenodd;                         /* 1: closed ways (and the compound of a directional rule) are filled with libcairo's EVEN_ODD rule
                                               instead of WINDING.  The reference never selects it (no cairo_set_fill_rule in
                                               src/smrules_cairo.c: :1012 and :1134-1138 fill with the default); it is here because the
                                               rasterizer is specified with both rules, pinned on libcairo's cairo_set_fill_rule.  Strokes
                                               are unaffected (a stroke's polygon is always filled non-zero) */
   int32_t pad_;
} smr_draw_style_t;
enum { SMR_STYLE_SOLID = 0, SMR_STYLE_DASHED = 1, SMR_STYLE_DOTTED = 2, SMR_STYLE_TRANSPARENT = 3, SMR_STYLE_PIPE = 4, SMR_STYLE_ROUNDDOT = 5 };

typedef struct
{
   /* nodes, already projected to pixel coordinates (output of smr_hip_project) */
   const double *d_x, *d_y;        /* [n_nodes] device */
   int64_t n_nodes;
   /* ways in the order the host's rule executor visits them (id order) */
   const int64_t *d_way_off;       /* [n_ways+1] device: first ref of each way */
   const int32_t *d_ref;           /* [n_refs] device: node index per ref; NULL = identity */
   int64_t n_ways, n_refs;
   /* optional per-way colours (color=%key): NULL = use the style's */
   const uint32_t *d_way_fill_col, *d_way_border_col;
} smr_draw_batch_t;

int smr_hip_draw(smr_canvas_t *c, const smr_draw_style_t *st, const smr_draw_batch_t *b);
/* The image fill of a rule (draw file=..., img on ways): render_poly_line src/smrules_cairo.c:989-997 / img_fill :2742-2751 fill
 * the CLOSED ways with a surface pattern that img_ini :2691-2704 sets up -- CAIRO_EXTEND_REPEAT, matrix scale(dpi / 72), i.e. one
 * image pixel per chart pixel, the image's top-left pixel on the chart's -- in a group of the rule's own, which act_draw_fini
 * :1081 pops BEFORE the rule's colours: call this first, smr_hip_draw after.  pattern: pat_w x pat_h premultiplied ARGB32 pixels
 * in HOST memory (the image as img_ini leaves it: scaled, transparency applied).  Of the style only directional, evenodd and curve
 * count; open ways draw nothing.  A rotated pattern (angle=) is not implemented: the caller refuses it. */
int smr_hip_draw_pattern(smr_canvas_t *c, const smr_draw_style_t *st, const smr_draw_batch_t *b, const uint32_t *pattern, int32_t pat_w, int32_t pat_h);
/* Flags about the last smr_hip_draw of this process.  SMR_DRAW_DEGRADED_ROWS: some pixel row of a polygon held no
 * vertex but more spanning edges than the exact whole-row coverage path keeps (64); such rows were sub-sampled on the
 * 15-row grid instead, so their anti-aliasing may differ from libcairo's by a few levels.  No counterpart in the
 * reference (libcairo has no such limit); the plugin logs a warning. */
#define SMR_DRAW_DEGRADED_ROWS 1u
/* SMR_DRAW_SECOND_PASS (informational, nothing is lost): some 64x4-px tile held more edge crossings on one sub-scanline than
 * the rasterizer's on-chip lists (30 crossings of a fill, 15 disjoint intervals of a stroke) and was rendered by the second
 * raster pass, whose lists live in device memory; the pixels are the same */
#define SMR_DRAW_SECOND_PASS 2u
uint32_t smr_hip_draw_status(void);

/* ---- caption auto-rotation: replaces find_angle ----------------------------------------
 * src/smrules_cairo.c:1959-2019 (find_angle), :1828-1861 (dv_sample), :1746-1759
 * (rotated OVER paint), :1670-1743 (cairo_smr_dist), :1598-1619 (bw stretch),
 * :1813-1821, :1868-1890, :1900-1956 (weight, quantize, peaks).  The cut-out is taken from
 * the device canvas at the rounded centre (difference to :1511-1533). */
typedef struct
{
   const double *d_cx, *d_cy;      /* [n] device: caption anchor in px (geo2pxf of the node) */
   int64_t n;
   int32_t fg_w, fg_h;             /* caption plane in px (cairo_smr_plane :1536-1558) */
   int32_t fg_x;                   /* first opaque column of the plane (E/W alignment), px */
   uint32_t col;                   /* caption colour 0xTTRRGGBB */
   double phase_deg, weight;       /* struct auto_rot (src/smrender_dev.h:168-174) */
   const uint32_t *d_fg;           /* optional [fg_h * fg_w] device: explicit plane pixels (premultiplied ARGB32);
                                      NULL = columns [fg_x, fg_w) opaque in `col`, the rest transparent */
} smr_cap_batch_t;

/* d_angle[n]: chosen angle in rad (find_angle's return value); d_dv may be NULL or
 * [n * num_steps] raw per-angle differences (dv_diff before weighting). */
int smr_hip_caprot(smr_canvas_t *c, const smr_cap_batch_t *b, double *d_angle, double *d_dv);
int smr_hip_caprot_steps(int fg_w, int fg_h, int dpi, int *num_steps, int *r_px);
/* smr_hip_caprot for callers that cannot allocate: tight serving loops, harnesses that forbid hipMalloc.  (It is NOT safe under
 * stream capture: it synchronises the library's stream and copies its tables from pageable host memory.)  No counterpart in the
 * reference.  Same arguments, same kernel, same results; the scan's only scratch
 * (the per-batch tables: one rotation matrix, angle and weight factor per trial angle) sits in a static device array of
 * SMR_CAPROT_NOALLOC_MAX_STEPS entries (48 KiB on every GPU the library was initialised on, until the process ends) instead
 * of the canvas's grow-only work buffers, so the call makes no hipMalloc / hipFree whatever it is given.  More trial angles
 * than that (a caption plane with a diagonal beyond ~325 mm) fail with SMR_E_CAPACITY.  All calls on one GPU share the
 * array: calls that overlap on one GPU race (the library has one stream per process, so they cannot unless two threads
 * call at once). */
#define SMR_CAPROT_NOALLOC_MAX_STEPS 1024
int smr_hip_caprot_noalloc(smr_canvas_t *c, const smr_cap_batch_t *b, double *d_angle, double *d_dv);

/* ---- the scan's cut-outs the reference's way: re-rasterised at a sub-pixel offset ---------------------------------------------
 * cairo_smr_cut_out (src/smrules_cairo.c:1511-1533) does not crop the rendered chart: it replays the recording into an r x r
 * image translated by (-x + r/2, -y + r/2) pt, so the cut-out's pixel grid carries the fraction of a pixel the caption's
 * anchor has.  smr_hip_caprot on the chart's canvas crops at the rounded anchor instead (fast, and 11-13 % of the %.1f angle
 * tags differ from the reference's: DESIGN.md 2).  These entry points render the reference's cut-outs: an ATLAS is a canvas
 * holding one window of `size` px per caption, one below the other; smr_hip_windows_create makes it (transparent, then the
 * chart's background rectangle), smr_hip_draw_windows draws a rule into every window it reaches -- the same style and batch
 * smr_hip_draw took for the chart, in the same order, the caller keeps them -- and smr_hip_caprot scans the atlas at the anchors
 * smr_hip_windows_anchors returns (the middle of each window), giving find_angle's angles for the real anchors.
 * The atlas is as wide as its largest window and gives every window that many rows (rounded up to 4): windows of different
 * sizes may share an atlas; a smaller one is limited to its own rows exactly and to the ATLAS's columns (so a polygon crossing
 * its right edge is limited a little further right than libcairo limits it: the anti-aliasing of that polygon's row at the
 * crossing can differ by a level; group windows of similar size).  An atlas holds at most 65 536 rows (240 windows of 271 px):
 * larger batches go in several atlases.  No counterpart in the reference beyond cairo_smr_cut_out
 * itself: there the cut-out is made per caption, on the CPU, by replaying the whole chart. */
typedef struct
{
   int64_t n;                      /* windows */
   const int32_t *size;            /* [n] HOST: width = height of each window in px, find_angle's round(r) (smr_hip_caprot_steps' r_px) */
   const double *r_px;             /* [n] HOST: find_angle's r before rounding, hypot(fg_w, fg_h) px (:1975): a window is centred with
                                      IT (cairo_smr_cut_out translates by r/2, :1526); NULL = size */
   const double *cx, *cy;          /* [n] HOST: the captions' anchors in chart px (geo2pxf) */
} smr_windows_t;
int smr_hip_windows_create(int chart_w, int chart_h, int dpi, uint32_t bgcolor, const smr_windows_t *w, smr_canvas_t **atlas);
/* the same atlas for the next batch of captions of the same chart: at most as many windows as it was created for, the largest as
 * wide as the atlas; pixels and window tables are set up again, the canvas and its work buffers stay (no hipMalloc / hipFree
 * between batches) */
int smr_hip_windows_reset(smr_canvas_t *atlas, const smr_windows_t *w);
int smr_hip_draw_windows(smr_canvas_t *atlas, const smr_draw_style_t *st, const smr_draw_batch_t *b);
/* call after the last draw into the atlas: it also clears what the rules left beyond the chart's rectangle (the reference's
 * recording has the chart's extents), then returns the anchors to scan at */
int smr_hip_windows_anchors(smr_canvas_t *atlas, double *cx, double *cy);      /* [n] HOST out */
/* smr_hip_canvas_blit for an atlas: what was composited onto the chart (caption text, symbols) lands in every window it meets,
 * at the window's offset ROUNDED to whole pixels: the images were rasterised on the chart's pixel grid */
int smr_hip_blit_windows(smr_canvas_t *atlas, const uint32_t *pixels, uint64_t n_pixels, const smr_blit_t *items, int n_items);

/* ---- stream / timing helpers --------------------------------------------------------- */
int smr_hip_sync(void);
void *smr_hip_stream(void);                    /* hipStream_t of the library */
/* when enabled, every stage of smr_hip_draw/_project/_caprot is bracketed by HIP events on
 * the library stream; smr_hip_timings() returns the last call's per-stage ms (and syncs). */
typedef struct
{
   float proj_ms, edges_ms, prims_ms, bin_ms, raster_ms, caprot_ms, total_ms;   /* draw: edge generation, primitive set-up,
                                                                                 binning + sort, raster */
   uint64_t n_prims, n_edges, n_tile_refs, n_tiles;
   float cover_ms;                                /* draw: scan conversion of the small primitives, primitive by primitive (k_cover) */
   uint32_t n_cover_prims, n_cover_left;          /* primitives k_cover handled / had to leave to the tile rasterizer */
   uint32_t n_deltas;                             /* coverage cells k_cover wrote */
   uint32_t n_cover_linear;                       /* of n_cover_prims: fills that took k_cover's crossing-by-crossing path (simple polygons) */
} smr_timings_t;
int smr_hip_set_profiling(int on);
int smr_hip_timings(smr_timings_t *t);

/* diagnostics: the edge slots (x1, y1, x2, y2 in 24.8 device space; all-zero slots are empty) the last draw pass on this
 * canvas generated, i.e. what libcairo's stroker/filler would hand to its scan converter.  No counterpart in the
 * reference; used by tools/ and tests to compare the geometry stage with the oracle's. */
int smr_hip_debug_edges(smr_canvas_t *c, int32_t *host_xyxy, uint64_t max_edges, uint64_t *n_edges);

/* device memory helpers so hosts without a HIP binding (ctypes, cgo, the C plugin) can
 * stage SoA arrays */
int smr_hip_malloc(void **d_ptr, size_t bytes);
int smr_hip_free(void *d_ptr);
int smr_hip_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes);
int smr_hip_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes);

/* ---- the plugin boundary itself: what smrender's rule parser binds with dlsym() --------
 * src/smrparse.c:286-300, :446-448 look up act_<func>_ini/_main/_fini in the library named
 * by `_action_ = func@libsmr_hip.so:params`; prototypes as doc/inc/lib_protos.h:1-3 and
 * libsmrender/smaction.h:99-113.  Arguments are smrule_t* / osm_obj_t* of the host
 * (smrender_amd/csrc/smr_abi.h restates their layout); declared with void* here so that this
 * header stays free of host types.
 *   draw    : act_draw_ini/_main/_fini   src/smrules_cairo.c:583, :1019, :1076
 *   cap     : act_cap_ini/_main/_fini    src/smrules_cairo.c:1243, :2447, :2487 (angle=auto scan:
 *             writes "<key>:autoangle" / "<key>:autoalign" tags, :2294-2349; the text is rasterised by the host's
 *             libcairo + FreeType and composited onto the device canvas, hide=1 leaves it out)
 *   img     : act_img_ini/_main/_fini    src/smrules_cairo.c:2716, :2829, :2858 for NODES (img_place :2754-2826: PNG / SVG
 *             symbol, scale, transparency, angle / anglekey / alignkey, angle=auto through the caption scan with the symbol as
 *             the plane); decoded by the host's libcairo / librsvg, composited onto the device canvas
 *   hipsave : writes the device canvas to `file=` (replaces save_main_image :388-412 for the GPU canvas) */
int act_draw_ini(void *rule);
int act_draw_main(void *rule, void *obj);
int act_draw_fini(void *rule);
int act_cap_ini(void *rule);
int act_cap_main(void *rule, void *obj);
int act_cap_fini(void *rule);
int act_img_ini(void *rule);
int act_img_main(void *rule, void *obj);
int act_img_fini(void *rule);
int act_hipsave_ini(void *rule);
int act_hipsave_main(void *rule, void *obj);
int act_hipsave_fini(void *rule);
/* the canvas the plugin entry points draw on (created at the first rule from rdata_get()) */
smr_canvas_t *smr_hip_plugin_canvas(void);
/* ends the chart: frees that canvas; the next _ini starts a new chart from rdata_get() (smrender
 * itself renders one chart per process, src/smrender.c:852-1416, and never needs this) */
int smr_hip_plugin_reset(void);
/* diagnostics (tools/plugin_bench.py): seconds the draw rules so far spent in [0] _main (the host's get_object per ref), [2] uploads +
 * projection, [3] smr_hip_draw, [4] the wait for it; reset != 0 clears them.  No counterpart in the reference */
int smr_hip_plugin_stats(double *out, int n, int reset);

#ifdef __cplusplus
}
#endif
#endif
