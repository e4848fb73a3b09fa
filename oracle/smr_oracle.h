/* smr_oracle.h -- TEST INFRASTRUCTURE. CPU restatement (plain C) of the reference's
 * hot path: projection (src/rdata.c, src/smrender.c:278-311), act_draw semantics
 * (src/smrules_cairo.c:583-1148) on top of the restated Cairo arithmetic (tor.h,
 * stroker, span compositing), and the act_cap auto-rotation scan
 * (src/smrules_cairo.c:1511-2019).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it; the product (libsmr_hip.so) never does.
 */
#ifndef SMR_ORACLE_H
#define SMR_ORACLE_H
#include <stdint.h>
#include "tor.h"

/* ---------- projection: src/rdata.c:188-230 (Mercator branch), :68-72, :240-245 ---------- */
typedef struct
{
   double lon0;      /* rd_.bb.ll.lon */
   double w, h;      /* rd_.w, rd_.h (pixels, double) */
   double wc;        /* rd_.wc */
   double lath;      /* rd_.lath */
   double lath_len;  /* rd_.lath_len */
   double lat_ll, lon_ru, lat_ru; /* rd_.bb for is_on_page (src/rdata.c:482-487) */
   int dpi;
} orc_proj_t;

/* src/smrender.c:278-311 init_bbox_mll (+ the 3 refinement iterations :300-310) */
void orc_init_bbox_mll(orc_proj_t *p, double mean_lat, double mean_lon, double scale, double w_px, double h_px, int dpi);
void orc_geo2pxf(const orc_proj_t *p, double lon, double lat, double *x, double *y);
void orc_pxf2geo(const orc_proj_t *p, double x, double y, double *lon, double *lat);
void orc_project(const orc_proj_t *p, const double *lat, const double *lon, int64_t n,
                 double *x, double *y, int32_t *ix, int32_t *iy, uint8_t *onpage);

/* ---------- fixed point / path ---------- */
int32_t orc_fixed_from_double(double d);
/* px (double) -> pt -> 24.8 -> replay scale: what the recording-surface path does to a vertex */
void orc_px_to_dev(const double *x_px, const double *y_px, int64_t n, int dpi, int32_t *xy_dev);
void orc_pt_to_dev(const double *xy_pt, int64_t n, int dpi, int32_t *xy_dev);

/* stroke of an open poly-line (the reference never closes paths: smrules_cairo.c:845-863):
 * butt caps, miter joins (limit 10).  hw_pt = half line width in pt.  Returns #edges,
 * out needs room for 8*(n-1) edges. */
int orc_stroke_edges(const int32_t *xy_dev, int n, double hw_pt, int dpi, tor_edge_t *out);
int orc_stroke_edges_round(const int32_t *xy_dev, int n, double hw_pt, int dpi, tor_edge_t *out);   /* + round caps; out: 8 n + 4 ORC_PEN_MAX */
/* the same poly-line stroked with a dash array (lengths in pt, offset 0) and butt or round caps:
 * libcairo's cairo-path-stroke.c dasher.  out == NULL: only counts.  Returns #edges (-1: pen too large). */
#define ORC_PEN_MAX 256
int orc_dash_edges(const int32_t *xy_dev, int n, double hw_pt, int dpi, const double *dash_pt, int ndash, int round_cap, tor_edge_t *out);
int orc_dash_edges_curve(const int32_t *xy_dev, const int32_t *tin, const int32_t *tout, const uint8_t *flag, int m, double hw_pt, int dpi,
                         const double *dash_pt, int ndash, int round_cap, tor_edge_t *out);
int orc_curve_flatten_flags(const double *start_pt, const double *curves_pt, int ncurves, int dpi, const int32_t *lim,
                            int32_t *pts, int32_t *tin, int32_t *tout, uint8_t *flag, int cap);
/* the pen (_cairo_pen_init, tolerance 0.1): vertex offsets in device 24.8; returns #vertices, -needed if > cap */
int orc_pen(double hw_pt, int dpi, int32_t *vx, int32_t *vy, int cap);
/* curve=1 (oracle/curve.c): control points (src/bspline_ctrl.c:64-105), the Bezier chain of a way in pt
 * (src/smrules_cairo.c:697-749), its decomposition in device space (cairo-spline.c, tolerance 0.1) and the stroke
 * of the decomposed curve (faces along the tangents).  out of the stroker: room for 8 * m edges. */
void orc_control_points(const double *g, const double *l, double f, double *p1, double *p2);
int orc_curve_path(const double *x_pt, const double *y_pt, int n, int closed, double f, double *start, double *curves);
void orc_curve_set_shift(int32_t sx, int32_t sy);
int orc_curve_flatten(const double *start_pt, const double *curves_pt, int ncurves, int dpi, const int32_t *lim,
                      int32_t *pts, int32_t *tin, int32_t *tout, int cap);
/* reach of a stroke beyond its path, px: _cairo_stroke_style_max_distance_from_path (miter joins, limit 10) */
double orc_stroke_reach_px(double width_pt, int dpi);
int orc_stroke_edges_tan(const int32_t *xy, const int32_t *tin, const int32_t *tout, int m, double hw_pt, int dpi, tor_edge_t *out);
/* fill polygon edges (implicitly closed): n edges */
int orc_fill_edges(const int32_t *xy_dev, int n, tor_edge_t *out);

/* ---------- canvas / compositing ---------- */
uint32_t orc_color_pixel(uint32_t smr_col);   /* 0xTTRRGGBB -> premultiplied ARGB32 source pixel */

typedef struct orc_canvas orc_canvas_t;
/* window of W x H pixels whose top-left is (x0,y0) on the full chart */
orc_canvas_t *orc_canvas_new(int x0, int y0, int W, int H, int dpi, uint32_t bgcolor);
void orc_canvas_free(orc_canvas_t *c);
/* a canvas with a replay's shift (orc_canvas_set_shift) moved down by whole rows, canvas and shift alike: window k of an atlas stands at
 * rows k * pitch, and the stroker's double arithmetic sees the rows */
void orc_canvas_move_rows(orc_canvas_t *c, int rows);
/* the image libcairo would render into (px on the full chart; default: the window itself): curves that cannot
 * reach it are drawn as lines (oracle/curve.c).  A window of a larger chart sets the chart's rectangle here. */
void orc_canvas_set_image(orc_canvas_t *c, int x0, int y0, int x1, int y1);
const uint32_t *orc_canvas_pixels(const orc_canvas_t *c);
void orc_rule_begin(orc_canvas_t *c);
void orc_rule_end(orc_canvas_t *c);
void orc_layer_edges(orc_canvas_t *c, const tor_edge_t *e, int n, int nonzero, uint32_t smr_col);
void orc_canvas_set_fill_poly(orc_canvas_t *c, int on);      /* the edges of the next orc_layer_edges calls are fills in the path's order */
void orc_canvas_set_pattern(orc_canvas_t *c, const uint32_t *pix, int w, int h);

void orc_direct_edges(orc_canvas_t *c, const tor_edge_t *e, int n, int nonzero, uint32_t smr_col);
void orc_direct_fill_pt(orc_canvas_t *c, const double *xy_pt, int n, uint32_t smr_col);
uint32_t *orc_canvas_pixels_rw(orc_canvas_t *c);

/* style of one draw rule: struct actDraw subset (src/smrender_dev.h:216-224, :252-268) */
typedef struct
{
   int fill_used, border_used;
   uint32_t fill_col, border_col;   /* 0xTTRRGGBB */
   double fill_width_mm, border_width_mm;
   int ignore_open;
   /* draw_style_t of border and fill (0 solid, 1 dashed, 2 dotted, 4 pipe, 5 rounddot: src/smrender_dev.h:145) and
    * their dash arrays in multiples of the border width, defaults of parse_dash_style already applied */
   int border_style, fill_style, border_ndash, fill_ndash;
   double border_dash[4], fill_dash[4];
   int curve;            /* curve=1: the way as a chain of Beziers (cairo_smr_poly_curve, src/smrules_cairo.c:697-749) */
   double curve_fact;    /* curve_factor=, default DIV_PART 0.2 (src/smrender_dev.h:114) */
   int evenodd;          /* cairo_fill under CAIRO_FILL_RULE_EVEN_ODD (the reference never sets a fill rule: an extension pinned on libcairo) */
} orc_draw_style_t;

/* render_poly_line (src/smrules_cairo.c:953-1016) for one way given projected px coords */
void orc_draw_way(orc_canvas_t *c, const orc_draw_style_t *st, const double *x_px, const double *y_px, int n, int closed,
                  int has_fill_col, uint32_t fill_col, int has_border_col, uint32_t border_col);

/* ---------- caption auto-rotation: src/smrules_cairo.c:1536-2019 ---------- */
/* pixman 0.40 bilinear (7-bit weights), EXTEND_NONE, OVER of the rotated cut-out onto dst */
void orc_rot_paint(uint32_t *dst, int dw, int dh, const uint32_t *src, int sw, int sh, double a);
/* cairo_smr_dist: returns avg, writes gray/TRANSPIX into dst */
double orc_dist(uint32_t *dst, const uint32_t *fg, int w, int h, double *var);
/* integer form of the per-angle sum: sum of |lum diff| * 2550000 and count */
void orc_dist_int(uint32_t *dst, const uint32_t *fg, int w, int h, int64_t *ksum, int *cnt);
/* whole find_angle given a cut-out (r x r) and plane (w x h): returns angle in rad; dv_out (num_steps) optional */
double orc_find_angle(const uint32_t *cut, int r, const uint32_t *fg, int w, int h, int num_steps,
                      double phase_deg, double weight, double *dv_out);
int orc_num_steps(int fg_w, int fg_h, int dpi, int *r_px);
void orc_make_plane(uint32_t *fg, int w_px, int h_px, int x_px, uint32_t smr_col);
uint32_t orc_bw_stretch(uint32_t col);

#endif
