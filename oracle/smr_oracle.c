/* smr_oracle.c -- TEST INFRASTRUCTURE; see smr_oracle.h.  Projection + draw path. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "smr_oracle.h"

#define DEG2RAD(x) ((x) * M_PI / 180.0)
#define RAD2DEG(x) ((x) * 180.0 / M_PI)

/* ---------------- projection ---------------- */

/* src/rdata.c:68-72 */
void orc_pxf2geo(const orc_proj_t *p, double x, double y, double *lon, double *lat)
{
   *lon = x * p->wc / p->w + p->lon0;
   *lat = RAD2DEG(atan(sinh(p->lath_len * (0.5 - y / p->h) + p->lath)));
}

/* src/rdata.c:221-225 (Mercator, no polygon window) */
void orc_geo2pxf(const orc_proj_t *p, double lon, double lat, double *x, double *y)
{
   *x = (lon - p->lon0) * p->w / p->wc;
   *y = p->h * (0.5 - (asinh(tan(DEG2RAD(lat))) - p->lath) / p->lath_len);
}

/* src/smrender.c:278-311; mean_lat_len from scale as src/smrender.c:697 */
void orc_init_bbox_mll(orc_proj_t *p, double mean_lat, double mean_lon, double scale, double w_px, double h_px, int dpi)
{
   double lat, lon, hc;
   double mean_lat_len = scale * (w_px / (double) dpi) * 2.54 / (60.0 * 1852 * 100);
   p->w = w_px; p->h = h_px; p->dpi = dpi;
   p->wc = mean_lat_len / cos(mean_lat * M_PI / 180);
   p->lon0 = mean_lon - p->wc / 2;
   p->lon_ru = mean_lon + p->wc / 2;
   hc = mean_lat_len * p->h / p->w;
   p->lat_ru = mean_lat + hc / 2.0;
   p->lat_ll = mean_lat - hc / 2.0;
   for (int i = 0; i < 3; i++)
   {
      p->lath = asinh(tan(DEG2RAD(mean_lat)));
      p->lath_len = asinh(tan(DEG2RAD(p->lat_ru))) - asinh(tan(DEG2RAD(p->lat_ll)));
      orc_pxf2geo(p, 0.0, 0.0, &lon, &lat);
      p->lat_ru = lat;
      orc_pxf2geo(p, 0.0, p->h, &lon, &lat);
      p->lat_ll = lat;
   }
}

/* batched form: geo2pxf + C round() indices (src/rdata.c:248-255) + is_on_page (:482-487) */
void orc_project(const orc_proj_t *p, const double *lat, const double *lon, int64_t n,
                 double *x, double *y, int32_t *ix, int32_t *iy, uint8_t *onpage)
{
   for (int64_t i = 0; i < n; i++)
   {
      orc_geo2pxf(p, lon[i], lat[i], &x[i], &y[i]);
      if (ix) ix[i] = (int32_t) round(x[i]);
      if (iy) iy[i] = (int32_t) round(y[i]);
      if (onpage)
         onpage[i] = !(lon[i] < p->lon0 || lon[i] > p->lon_ru || lat[i] < p->lat_ll || lat[i] > p->lat_ru);
   }
}

/* ---------------- fixed point, path ---------------- */

/* _cairo_fixed_from_double (cairo 1.16, 24.8): adds 1.5*2^(52-8) and takes the low word,
 * i.e. round-to-nearest, ties to even */
int32_t orc_fixed_from_double(double d)
{
   return (int32_t) nearbyint(d * 256.0);
}

static int32_t fixed_mul(int32_t a, int32_t b)
{
   return (int32_t) (((int64_t) a * b) >> 8);
}

/* The reference records paths in pt on a recording surface (geo2pt, src/rdata.c:240-245;
 * cairo_line_to, src/smrules_cairo.c:860) and replays them with cairo_scale(dpi/72)
 * (:348, :403).  libcairo 1.16 quantises the vertex to 24.8 in pt and applies the replay
 * matrix in fixed point with the SCALE ITSELF quantised to 24.8 (pinned by probe:
 * tests/test_oracle_cairo.py::test_replay_quantisation). */
void orc_pt_to_dev(const double *xy_pt, int64_t n, int dpi, int32_t *xy_dev)
{
   int32_t s = orc_fixed_from_double((double) dpi / 72);
   for (int64_t i = 0; i < 2 * n; i++)
      xy_dev[i] = fixed_mul(orc_fixed_from_double(xy_pt[i]), s);
}

void orc_px_to_dev(const double *x_px, const double *y_px, int64_t n, int dpi, int32_t *xy_dev)
{
   int32_t s = orc_fixed_from_double((double) dpi / 72);
   for (int64_t i = 0; i < n; i++)
   {
      /* rdata_px_unit(x, U_PT): x * 72 / dpi (src/rdata.c:307-308) */
      xy_dev[2 * i] = fixed_mul(orc_fixed_from_double(x_px[i] * 72 / dpi), s);
      xy_dev[2 * i + 1] = fixed_mul(orc_fixed_from_double(y_px[i] * 72 / dpi), s);
   }
}

int orc_fill_edges(const int32_t *p, int n, tor_edge_t *out)
{
   for (int i = 0; i < n; i++)
   {
      int j = (i + 1) % n;
      out[i].x1 = p[2 * i]; out[i].y1 = p[2 * i + 1];
      out[i].x2 = p[2 * j]; out[i].y2 = p[2 * j + 1];
   }
   return n;
}

/* ---- stroker: cairo-path-stroke-polygon.c semantics for an open poly-line, butt caps,
 * miter joins with Cairo's default limit 10 (the reference never sets join/limit).  The
 * stroke polygon's non-zero region equals the union of one quad per segment and one
 * wedge (miter quad or bevel triangle) on the outer side of every join; that union is
 * what we emit (same orientation for every piece). ---- */

typedef struct { int32_t x, y; } pt_t;

static int emit_poly(const pt_t *p, int n, tor_edge_t *out)
{
   /* orient positively (shoelace) so every piece winds the same way */
   int64_t a = 0;
   for (int i = 0; i < n; i++)
   {
      int j = (i + 1) % n;
      a += (int64_t) p[i].x * p[j].y - (int64_t) p[j].x * p[i].y;
   }
   if (a == 0) return 0;
   for (int i = 0; i < n; i++)
   {
      int j = (i + 1) % n;
      const pt_t *s = a > 0 ? &p[i] : &p[j], *e = a > 0 ? &p[j] : &p[i];
      out[i].x1 = s->x; out[i].y1 = s->y; out[i].x2 = e->x; out[i].y2 = e->y;
   }
   return n;
}

static double normalize_slope(double *dx, double *dy)
{
   double dx0 = *dx, dy0 = *dy, mag;
   if (dx0 == 0.0)
   {
      *dx = 0.0;
      if (dy0 > 0.0) { mag = dy0; *dy = 1.0; } else { mag = -dy0; *dy = -1.0; }
   }
   else if (dy0 == 0.0)
   {
      *dy = 0.0;
      if (dx0 > 0.0) { mag = dx0; *dx = 1.0; } else { mag = -dx0; *dx = -1.0; }
   }
   else
   {
      mag = hypot(dx0, dy0);
      *dx = dx0 / mag; *dy = dy0 / mag;
   }
   return mag;
}

typedef struct { pt_t ccw, cw, point; double sx, sy; int32_t ddx, ddy; } face_t;

static void compute_face(pt_t p, int32_t ddx, int32_t ddy, double hw_pt, double s, face_t *f)
{
   double sx = ddx / 256.0, sy = ddy / 256.0;
   normalize_slope(&sx, &sy);
   f->sx = sx; f->sy = sy; f->ddx = ddx; f->ddy = ddy;
   /* ctm = scale(s): to user space (distance * 1/s), normalise again, rotate, back */
   double ux = sx * (1.0 / s), uy = sy * (1.0 / s);
   normalize_slope(&ux, &uy);
   double fx = -uy * hw_pt, fy = ux * hw_pt;
   fx *= s; fy *= s;
   int32_t ox = orc_fixed_from_double(fx), oy = orc_fixed_from_double(fy);
   f->point = p;
   f->ccw.x = p.x + ox; f->ccw.y = p.y + oy;
   f->cw.x = p.x - ox; f->cw.y = p.y - oy;
}

static int sgn_cmp(double dx1, double dy1, double dx2, double dy2)
{
   double c = dx1 * dy2 - dx2 * dy1;
   return c > 0 ? 1 : c < 0 ? -1 : 0;
}

int orc_stroke_edges(const int32_t *xy, int n, double hw_pt, int dpi, tor_edge_t *out)
{
   double s = (double) dpi / 72;
   int cnt = 0, have = 0;
   face_t cur;          /* face at the END of the previous segment (at the join point) */
   pt_t p1 = { xy[0], xy[1] };
   for (int i = 1; i < n; i++)
   {
      pt_t p2 = { xy[2 * i], xy[2 * i + 1] };
      if (p1.x == p2.x && p1.y == p2.y) continue;     /* degenerate segment skipped */
      int32_t ddx = p2.x - p1.x, ddy = p2.y - p1.y;
      face_t start, end;
      compute_face(p1, ddx, ddy, hw_pt, s, &start);
      end = start;
      end.point = p2;
      end.ccw.x += ddx; end.ccw.y += ddy; end.cw.x += ddx; end.cw.y += ddy;

      if (have && !(cur.cw.x == start.cw.x && cur.cw.y == start.cw.y && cur.ccw.x == start.ccw.x && cur.ccw.y == start.ccw.y))
      {
         /* _cairo_slope_compare(in, out) < 0  <=>  in.dy*out.dx < out.dy*in.dx */
         int64_t l = (int64_t) cur.ddy * start.ddx, r = (int64_t) start.ddy * cur.ddx;
         int clockwise;
         if (l != r) clockwise = l < r;
         else clockwise = (cur.ddx > 0 || (cur.ddx == 0 && cur.ddy > 0));   /* anti-parallel */
         pt_t inpt = clockwise ? cur.cw : cur.ccw, outpt = clockwise ? start.cw : start.ccw;
         double in_dot_out = cur.sx * start.sx + cur.sy * start.sy;
         double ml = 10.0;
         int mitered = 0;
         if (2 <= ml * ml * (1 + in_dot_out))
         {
            double x1 = inpt.x / 256.0, y1 = inpt.y / 256.0, dx1 = cur.sx, dy1 = cur.sy;
            double x2 = outpt.x / 256.0, y2 = outpt.y / 256.0, dx2 = start.sx, dy2 = start.sy;
            double my = (((x2 - x1) * dy1 * dy2 - y2 * dx2 * dy1 + y1 * dx1 * dy2) / (dx1 * dy2 - dx2 * dy1));
            double mx;
            if (fabs(dy1) >= fabs(dy2)) mx = (my - y1) * dx1 / dy1 + x1;
            else mx = (my - y2) * dx2 / dy2 + x2;
            double ix = p1.x / 256.0, iy = p1.y / 256.0;
            double fdx1 = x1 - ix, fdy1 = y1 - iy, fdx2 = x2 - ix, fdy2 = y2 - iy, mdx = mx - ix, mdy = my - iy;
            if (sgn_cmp(fdx1, fdy1, mdx, mdy) != sgn_cmp(fdx2, fdy2, mdx, mdy))
            {
               pt_t q[4] = { p1, inpt, { orc_fixed_from_double(mx), orc_fixed_from_double(my) }, outpt };
               cnt += emit_poly(q, 4, out + cnt);
               mitered = 1;
            }
         }
         if (!mitered)
         {
            pt_t q[3] = { p1, inpt, outpt };
            cnt += emit_poly(q, 3, out + cnt);
         }
      }
      pt_t q[4] = { start.ccw, end.ccw, end.cw, start.cw };
      cnt += emit_poly(q, 4, out + cnt);
      cur = end; have = 1;
      p1 = p2;
   }
   return cnt;
}

/* ---------------- colour + compositing ---------------- */

static uint32_t d2s(double d)   /* _cairo_color_double_to_short */
{
   uint32_t i = (uint32_t) (d * 65536);
   i -= (i >> 16);
   return i;
}

/* src/smrules_cairo.c:90-95, :271-274 -> cairo_set_source_rgba -> premultiplied 8-bit pixel */
uint32_t orc_color_pixel(uint32_t col)
{
   double r = (double) ((col >> 16) & 0xff) / 255.0;
   double g = (double) ((col >> 8) & 0xff) / 255.0;
   double b = (double) (col & 0xff) / 255.0;
   double a = 1.0 - (double) (((col & 0x7f000000) >> 23) & 0xff) / 255.0;
   return ((d2s(a) >> 8) << 24) | ((d2s(r * a) >> 8) << 16) | ((d2s(g * a) >> 8) << 8) | (d2s(b * a) >> 8);
}

static inline uint32_t mul_7f(uint32_t a, uint32_t b) { uint32_t t = a * b + 0x7f; return (t + (t >> 8)) >> 8; }
static inline uint32_t mul_80(uint32_t a, uint32_t b) { uint32_t t = a * b + 0x80; return (t + (t >> 8)) >> 8; }
static inline uint32_t sat(uint32_t v) { return v > 255 ? 255 : v; }
#define CH(p, s) (((p) >> (s)) & 0xff)

/* cairo image span renderer, SOURCE-bounded-by-mask == lerp (used for opaque sources and
 * for the first drawing into a still-clear group): _fill_xrgb32_lerp_opaque_spans */
static uint32_t blend_lerp(uint32_t s, uint32_t a, uint32_t d)
{
   uint32_t o = 0;
   if (a == 255) return s;
   for (int sh = 0; sh < 32; sh += 8)
      o |= sat(mul_7f(CH(s, sh), a) + mul_7f(CH(d, sh), 255 - a)) << sh;
   return o;
}

/* pixman fast_composite_over_n_8_8888: d = (s IN m) OVER d */
static uint32_t blend_over_mask(uint32_t s, uint32_t m, uint32_t d)
{
   uint32_t o = 0;
   if (m == 0) return d;
   if (m != 255)
   {
      uint32_t t = 0;
      for (int sh = 0; sh < 32; sh += 8) t |= mul_80(CH(s, sh), m) << sh;
      s = t;
   }
   uint32_t ia = 255 - (s >> 24);
   for (int sh = 0; sh < 32; sh += 8)
      o |= sat(CH(s, sh) + mul_80(CH(d, sh), ia)) << sh;
   return o;
}

/* pixman over_8888_8888 (group layer painted OVER the canvas) */
static uint32_t blend_over(uint32_t s, uint32_t d)
{
   uint32_t a = s >> 24, o = 0;
   if (a == 255) return s;
   if (s == 0) return d;
   for (int sh = 0; sh < 32; sh += 8)
      o |= sat(CH(s, sh) + mul_80(CH(d, sh), 255 - a)) << sh;
   return o;
}

struct orc_canvas
{
   int x0, y0, W, H, dpi;
   uint32_t *pix, *layer;
   uint8_t *cov;
   int layer_clear;
};

orc_canvas_t *orc_canvas_new(int x0, int y0, int W, int H, int dpi, uint32_t bgcolor)
{
   orc_canvas_t *c = calloc(1, sizeof(*c));
   c->x0 = x0; c->y0 = y0; c->W = W; c->H = H; c->dpi = dpi;
   c->pix = malloc(sizeof(uint32_t) * W * H);
   c->layer = malloc(sizeof(uint32_t) * W * H);
   c->cov = malloc((size_t) W * H);
   uint32_t bg = orc_color_pixel(bgcolor);       /* paint onto a clear image == SOURCE */
   for (size_t i = 0; i < (size_t) W * H; i++) c->pix[i] = bg;
   return c;
}

void orc_canvas_free(orc_canvas_t *c) { free(c->pix); free(c->layer); free(c->cov); free(c); }
const uint32_t *orc_canvas_pixels(const orc_canvas_t *c) { return c->pix; }

/* cairo_push_group (src/smrules_cairo.c:660): drawing goes to a clear ARGB32 layer */
void orc_rule_begin(orc_canvas_t *c)
{
   memset(c->layer, 0, sizeof(uint32_t) * c->W * c->H);
   c->layer_clear = 1;
}

/* cairo_pop_group_to_source + cairo_paint (src/smrules_cairo.c:203-205, :1081-1082) */
void orc_rule_end(orc_canvas_t *c)
{
   for (size_t i = 0; i < (size_t) c->W * c->H; i++) c->pix[i] = blend_over(c->layer[i], c->pix[i]);
}

void orc_layer_edges(orc_canvas_t *c, const tor_edge_t *e, int n, int nonzero, uint32_t smr_col)
{
   uint32_t s = orc_color_pixel(smr_col);
   int lerp = c->layer_clear || (s >> 24) == 255;
   /* only the primitive's bounding box can receive coverage */
   int32_t minx = INT32_MAX, miny = INT32_MAX, maxx = INT32_MIN, maxy = INT32_MIN;
   for (int i = 0; i < n; i++)
   {
      if (e[i].y1 == e[i].y2) continue;
      int32_t lx = e[i].x1 < e[i].x2 ? e[i].x1 : e[i].x2, hx = e[i].x1 < e[i].x2 ? e[i].x2 : e[i].x1;
      int32_t ly = e[i].y1 < e[i].y2 ? e[i].y1 : e[i].y2, hy = e[i].y1 < e[i].y2 ? e[i].y2 : e[i].y1;
      if (lx < minx) minx = lx;
      if (hx > maxx) maxx = hx;
      if (ly < miny) miny = ly;
      if (hy > maxy) maxy = hy;
   }
   if (minx > maxx) return;
   int bx0 = (minx >> 8) - c->x0, bx1 = (maxx >> 8) - c->x0 + 1, by0 = (miny >> 8) - c->y0, by1 = (maxy >> 8) - c->y0 + 1;
   if (bx0 < 0) bx0 = 0;
   if (by0 < 0) by0 = 0;
   if (bx1 > c->W) bx1 = c->W;
   if (by1 > c->H) by1 = c->H;
   if (bx0 >= bx1 || by0 >= by1) return;      /* nothing drawn: the layer stays "clear" for libcairo too */
   int bw = bx1 - bx0, bh = by1 - by0;
   tor_coverage(e, n, nonzero, c->x0 + bx0, c->y0 + by0, bw, bh, c->cov);
   for (int y = 0; y < bh; y++)
      for (int x = 0; x < bw; x++)
      {
         uint32_t m = c->cov[(size_t) y * bw + x];
         if (!m) continue;
         uint32_t *d = &c->layer[(size_t) (y + by0) * c->W + x + bx0];
         *d = lerp ? blend_lerp(s, m, *d) : blend_over_mask(s, m, *d);
      }
   c->layer_clear = 0;
}

/* a fill issued on a context WITHOUT push_group (act_cap's auto_ctx, src/smrules_cairo.c:2376-2382):
 * the spans composite straight onto the surface -- lerp for an opaque source, (s IN m) OVER d else */
void orc_direct_edges(orc_canvas_t *c, const tor_edge_t *e, int n, int nonzero, uint32_t smr_col)
{
   uint32_t s = orc_color_pixel(smr_col);
   int lerp = (s >> 24) == 255;
   int32_t minx = INT32_MAX, miny = INT32_MAX, maxx = INT32_MIN, maxy = INT32_MIN;
   for (int i = 0; i < n; i++)
   {
      int32_t lx = e[i].x1 < e[i].x2 ? e[i].x1 : e[i].x2, hx = e[i].x1 < e[i].x2 ? e[i].x2 : e[i].x1;
      int32_t ly = e[i].y1 < e[i].y2 ? e[i].y1 : e[i].y2, hy = e[i].y1 < e[i].y2 ? e[i].y2 : e[i].y1;
      if (lx < minx) minx = lx;
      if (hx > maxx) maxx = hx;
      if (ly < miny) miny = ly;
      if (hy > maxy) maxy = hy;
   }
   if (minx > maxx) return;
   int bx0 = (minx >> 8) - c->x0, bx1 = (maxx >> 8) - c->x0 + 1, by0 = (miny >> 8) - c->y0, by1 = (maxy >> 8) - c->y0 + 1;
   if (bx0 < 0) bx0 = 0;
   if (by0 < 0) by0 = 0;
   if (bx1 > c->W) bx1 = c->W;
   if (by1 > c->H) by1 = c->H;
   if (bx0 >= bx1 || by0 >= by1) return;
   int bw = bx1 - bx0, bh = by1 - by0;
   tor_coverage(e, n, nonzero, c->x0 + bx0, c->y0 + by0, bw, bh, c->cov);
   for (int y = 0; y < bh; y++)
      for (int x = 0; x < bw; x++)
      {
         uint32_t m = c->cov[(size_t) y * bw + x];
         if (!m) continue;
         uint32_t *d = &c->pix[(size_t) (y + by0) * c->W + x + bx0];
         *d = lerp ? blend_lerp(s, m, *d) : blend_over_mask(s, m, *d);
      }
}

/* closed polygon given in pt on the recording surface (cairo_rectangle under translate+rotate) */
void orc_direct_fill_pt(orc_canvas_t *c, const double *xy_pt, int n, uint32_t smr_col)
{
   int32_t *dev = malloc(sizeof(int32_t) * 2 * n);
   tor_edge_t *e = malloc(sizeof(tor_edge_t) * n);
   orc_pt_to_dev(xy_pt, n, c->dpi, dev);
   orc_direct_edges(c, e, orc_fill_edges(dev, n, e), 1, smr_col);
   free(dev); free(e);
}

uint32_t *orc_canvas_pixels_rw(orc_canvas_t *c) { return c->pix; }

/* mm -> pt; width 0 -> 1 device pixel (THINLINE): src/smrules_cairo.c:67-69, src/rdata.c:50-53 */
static double mm2wu(double mm, int dpi) { return mm == 0.0 ? 1.0 * 72 / dpi : mm * 72 / 25.4; }

/* render_poly_line, src/smrules_cairo.c:953-1016 (solid styles, no pipe/image/curve) */
void orc_draw_way(orc_canvas_t *c, const orc_draw_style_t *st, const double *x_px, const double *y_px, int n, int closed,
                  int has_fill_col, uint32_t fill_col, int has_border_col, uint32_t border_col)
{
   if (n < 2) return;                                   /* :962-966 */
   if (!closed && st->ignore_open) return;              /* :1030-1033 */
   int32_t *dev = malloc(sizeof(int32_t) * 2 * n);
   tor_edge_t *e = malloc(sizeof(tor_edge_t) * 8 * n);
   orc_px_to_dev(x_px, y_px, n, c->dpi, dev);
   if (st->border_used)
   {
      /* cairo_smr_border_width, :880-889 */
      double bw;
      if (!st->fill_used) bw = mm2wu(st->border_width_mm, c->dpi);
      else if (!closed) bw = mm2wu(2 * st->border_width_mm, c->dpi) + mm2wu(st->fill_width_mm, c->dpi);
      else bw = mm2wu(2 * st->border_width_mm, c->dpi);
      int ne = orc_stroke_edges(dev, n, bw / 2, c->dpi, e);
      orc_layer_edges(c, e, ne, 1, has_border_col ? border_col : st->border_col);
   }
   if (st->fill_used)
   {
      uint32_t col = has_fill_col ? fill_col : st->fill_col;
      if (!closed)
      {
         int ne = orc_stroke_edges(dev, n, mm2wu(st->fill_width_mm, c->dpi) / 2, c->dpi, e);   /* :1005-1007 */
         orc_layer_edges(c, e, ne, 1, col);
      }
      else
      {
         int ne = orc_fill_edges(dev, n, e);                                                     /* :1012 */
         orc_layer_edges(c, e, ne, 1, col);
      }
   }
   free(dev); free(e);
}
