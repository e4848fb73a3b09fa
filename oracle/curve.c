/* curve.c -- TEST INFRASTRUCTURE; see smr_oracle.h.  curve=1 of act_draw: the way as a chain of cubic
 * Beziers (cairo_smr_poly_curve, src/smrules_cairo.c:697-749; control_points, src/bspline_ctrl.c:64-105),
 * libcairo 1.16's spline decomposition (cairo-spline.c: de Casteljau halving in 24.8 until the control
 * points are within the tolerance of the chord) and the stroke of the decomposed curve (cairo-path-stroke-
 * polygon.c: curve_to / spline_to put one face per decomposed point, its direction the TANGENT there;
 * joins only where two Beziers meet).  libcairo's sources are not in the image: restated from its published
 * algorithm and pinned against the real library (tests/test_oracle_cairo.py, tests/golden/scene_curves_*). */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "smr_oracle.h"

/* src/bspline_ctrl.c:64-105 (ISOSCELES_TRIANGLE); g, l = { A.x, A.y, B.x, B.y } */
void orc_control_points(const double *g, const double *l, double f, double *p1, double *p2)
{
   double lgt = hypot(g[2] - l[0], g[3] - l[1]);
   double ag = atan2(g[3] - g[1], g[2] - g[0]);
   double hax = g[2] - lgt * cos(ag), hay = g[3] - lgt * sin(ag);
   double a = atan2(l[1] - hay, l[0] - hax);
   p1[0] = g[2] + lgt * cos(a) * f;
   p1[1] = g[3] + lgt * sin(a) * f;
   double al = atan2(l[3] - l[1], l[2] - l[0]);
   double hbx = l[0] + lgt * cos(al), hby = l[1] + lgt * sin(al);
   a = atan2(hby - g[3], hbx - g[2]);
   p2[0] = l[0] - lgt * cos(a) * f;
   p2[1] = l[1] - lgt * sin(a) * f;
}

/* src/smrules_cairo.c:697-749: start = the move_to point, curves = { c1x c1y c2x c2y x y } per curve_to, all in pt.
 * n = ref_cnt, closed = is_closed_poly.  Returns the number of curves (room: n). */
int orc_curve_path(const double *x_pt, const double *y_pt, int n, int closed, double f, double *start, double *curves)
{
   int cnt = n, st = !closed, k = 0;
   if (!st) cnt--;
   if (cnt < 1) return 0;
   start[0] = x_pt[(st - 1 + cnt) % cnt]; start[1] = y_pt[(st - 1 + cnt) % cnt];
   for (int i = st; i < cnt; i++, k++)
   {
      int i0 = (i + cnt - 2) % cnt, i1 = (i + cnt - 1) % cnt, i2 = i % cnt, i3 = (i + 1) % cnt;
      double g[4] = { x_pt[i0], y_pt[i0], x_pt[i1], y_pt[i1] }, l[4] = { x_pt[i2], y_pt[i2], x_pt[i3], y_pt[i3] };
      double c1[2], c2[2];
      orc_control_points(g, l, f, c1, c2);
      if (st)
      {
         if (i == 1) { c1[0] = g[2]; c1[1] = g[3]; }
         if (i == cnt - 1) { c2[0] = l[0]; c2[1] = l[1]; }
      }
      double *c = curves + 6 * k;
      c[0] = c1[0]; c[1] = c1[1]; c[2] = c2[0]; c[3] = c2[1]; c[4] = x_pt[i]; c[5] = y_pt[i];
   }
   return k;
}

/* ---- cairo-spline.c ---- */
typedef struct { int32_t x, y; } pt_t;
typedef struct { pt_t a, b, c, d; } knots_t;

typedef struct
{
   int32_t *pts, *tin, *tout;
   uint8_t *flag;                  /* optional, per point: 1 = a curve_to ends here (or the move_to), 2 = that curve_to was a line_to */
   int m, cap, overflow;
   pt_t last;
} sink_t;

static void sink_point(sink_t *s, pt_t p, int32_t tdx, int32_t tdy)
{
   if (s->m >= s->cap) { s->overflow = 1; return; }
   s->pts[2 * s->m] = p.x; s->pts[2 * s->m + 1] = p.y;
   s->tin[2 * s->m] = tdx; s->tin[2 * s->m + 1] = tdy;
   s->tout[2 * s->m] = tdx; s->tout[2 * s->m + 1] = tdy;
   if (s->flag) s->flag[s->m] = 0;
   s->m++;
}

static double fx2d(int32_t f) { return f / 256.0; }

/* _cairo_spline_error_squared */
static double spline_err2(const knots_t *k)
{
   double bdx = fx2d(k->b.x - k->a.x), bdy = fx2d(k->b.y - k->a.y);
   double cdx = fx2d(k->c.x - k->a.x), cdy = fx2d(k->c.y - k->a.y);
   if (k->a.x != k->d.x || k->a.y != k->d.y)
   {
      double dx = fx2d(k->d.x - k->a.x), dy = fx2d(k->d.y - k->a.y);
      double v = dx * dx + dy * dy, u = bdx * dx + bdy * dy;
      if (u <= 0) { }
      else if (u >= v) { bdx -= dx; bdy -= dy; }
      else { bdx -= u / v * dx; bdy -= u / v * dy; }
      u = cdx * dx + cdy * dy;
      if (u <= 0) { }
      else if (u >= v) { cdx -= dx; cdy -= dy; }
      else { cdx -= u / v * dx; cdy -= u / v * dy; }
   }
   double berr = bdx * bdx + bdy * bdy, cerr = cdx * cdx + cdy * cdy;
   return berr > cerr ? berr : cerr;
}

static pt_t lerp_half(pt_t a, pt_t b)
{
   pt_t r = { a.x + ((b.x - a.x) >> 1), a.y + ((b.y - a.y) >> 1) };
   return r;
}

static void de_casteljau(knots_t *s1, knots_t *s2)
{
   pt_t ab = lerp_half(s1->a, s1->b), bc = lerp_half(s1->b, s1->c), cd = lerp_half(s1->c, s1->d);
   pt_t abbc = lerp_half(ab, bc), bccd = lerp_half(bc, cd), fin = lerp_half(abbc, bccd);
   s2->a = fin; s2->b = bccd; s2->c = cd; s2->d = s1->d;
   s1->b = ab; s1->c = abbc; s1->d = fin;
}

/* _cairo_spline_add_point: the start of a leaf with the slope towards its first control point */
static void spline_add_point(sink_t *s, pt_t p, pt_t knot)
{
   if (s->last.x == p.x && s->last.y == p.y) return;
   s->last = p;
   sink_point(s, p, knot.x - p.x, knot.y - p.y);
}

static void decompose_into(knots_t *s1, double tol2, sink_t *s, int depth)
{
   if (spline_err2(s1) < tol2 || depth > 40) { spline_add_point(s, s1->a, s1->b); return; }
   knots_t s2;
   de_casteljau(s1, &s2);
   decompose_into(s1, tol2, s, depth + 1);
   decompose_into(&s2, tol2, s, depth + 1);
}

/* _cairo_spline_intersects: does the box of the four knots meet the limit box lim = { x0, y0, x1, y1 }? */
static int spline_intersects(pt_t a, pt_t b, pt_t c, pt_t d, const int32_t *lim)
{
   const pt_t k[4] = { a, b, c, d };
   int32_t x0 = a.x, x1 = a.x, y0 = a.y, y1 = a.y;
   for (int i = 0; i < 4; i++)
   {
      if (lim[0] <= k[i].x && k[i].x <= lim[2] && lim[1] <= k[i].y && k[i].y <= lim[3]) return 1;
      if (k[i].x < x0) x0 = k[i].x;
      if (k[i].x > x1) x1 = k[i].x;
      if (k[i].y < y0) y0 = k[i].y;
      if (k[i].y > y1) y1 = k[i].y;
   }
   return !(x1 <= lim[0] || x0 >= lim[2] || y1 <= lim[1] || y0 >= lim[3]);
}

/* one curve_to from the current point a: _cairo_spline_init + _cairo_spline_decompose, or the line_to both the
 * filler and the stroker fall back to.  tout of the point the curve starts at becomes the curve's initial slope. */
static void curve_to(sink_t *s, pt_t a, pt_t b, pt_t c, pt_t d, const int32_t *lim)
{
   int32_t *tout_a = s->m > 0 ? s->tout + 2 * (s->m - 1) : NULL;
   int line = a.x == b.x && a.y == b.y && c.x == d.x && c.y == d.y;
   /* a curve that cannot reach the image is a line_to for the filler (_cairo_filler_curve_to) and for the stroker
    * (curve_to of cairo-path-stroke-polygon.c): libcairo limits the polygon to the image whenever the path leaves it */
   if (lim && !spline_intersects(a, b, c, d, lim)) line = 1;
   pt_t is = { 0, 0 }, fs = { 0, 0 };
   if (!line)
   {
      if (a.x != b.x || a.y != b.y) { is.x = b.x - a.x; is.y = b.y - a.y; }
      else if (a.x != c.x || a.y != c.y) { is.x = c.x - a.x; is.y = c.y - a.y; }
      else if (a.x != d.x || a.y != d.y) { is.x = d.x - a.x; is.y = d.y - a.y; }
      else line = 1;
   }
   if (!line)
   {
      if (c.x != d.x || c.y != d.y) { fs.x = d.x - c.x; fs.y = d.y - c.y; }
      else if (b.x != d.x || b.y != d.y) { fs.x = d.x - b.x; fs.y = d.y - b.y; }
      else line = 1;
   }
   if (line)
   {
      if (a.x == d.x && a.y == d.y) return;             /* line_to the current point: nothing */
      if (tout_a) { tout_a[0] = d.x - a.x; tout_a[1] = d.y - a.y; }
      sink_point(s, d, d.x - a.x, d.y - a.y);
      if (s->flag && !s->overflow) s->flag[s->m - 1] = 3;
      return;
   }
   if (tout_a) { tout_a[0] = is.x; tout_a[1] = is.y; }
   knots_t k = { a, b, c, d };
   s->last = a;
   decompose_into(&k, 0.1 * 0.1, s, 0);
   sink_point(s, d, fs.x, fs.y);
   if (s->flag && !s->overflow) s->flag[s->m - 1] = 1;
}

/* The recorded path replayed at dpi/72 (fixed * fixed(dpi/72) >> 8 per coordinate, like every vertex:
 * orc_pt_to_dev) and decomposed: pts[2m], tin[2m] (direction of the face the stroker puts on the segment
 * ARRIVING at the point), tout[2m] (... LEAVING it).  lim = { x0, y0, x1, y1 } in 24.8: the image (fill) or the
 * image grown by the stroke's reach (stroke); NULL: no limit.  Returns m, or -1 when cap is too small. */
/* the 24.8 translation of a replay at an offset (orc_canvas_set_shift), added to every knot before the decomposition */
static int32_t flatten_shift_[2];
void orc_curve_set_shift(int32_t sx, int32_t sy) { flatten_shift_[0] = sx; flatten_shift_[1] = sy; }

int orc_curve_flatten_flags(const double *start_pt, const double *curves_pt, int ncurves, int dpi, const int32_t *lim,
                            int32_t *pts, int32_t *tin, int32_t *tout, uint8_t *flag, int cap);
int orc_curve_flatten(const double *start_pt, const double *curves_pt, int ncurves, int dpi, const int32_t *lim,
                      int32_t *pts, int32_t *tin, int32_t *tout, int cap)
{
   return orc_curve_flatten_flags(start_pt, curves_pt, ncurves, dpi, lim, pts, tin, tout, NULL, cap);
}

/* the same with, per point, where the path's curve_to calls end (the dashing stroker puts its faces along the chords inside a
 * curve and along the tangents where one ends, orc_dash_edges_curve) */
int orc_curve_flatten_flags(const double *start_pt, const double *curves_pt, int ncurves, int dpi, const int32_t *lim,
                            int32_t *pts, int32_t *tin, int32_t *tout, uint8_t *flag, int cap)
{
   sink_t s; s.pts = pts; s.tin = tin; s.tout = tout; s.flag = flag; s.m = 0; s.cap = cap; s.overflow = 0;
   int32_t dv[6];
   orc_pt_to_dev(start_pt, 1, dpi, dv);
   dv[0] += flatten_shift_[0]; dv[1] += flatten_shift_[1];
   pt_t cur = { dv[0], dv[1] };
   sink_point(&s, cur, 0, 0);
   if (flag && cap > 0) flag[0] = 1;
   for (int k = 0; k < ncurves; k++)
   {
      orc_pt_to_dev(curves_pt + 6 * k, 3, dpi, dv);
      for (int t = 0; t < 6; t++) dv[t] += flatten_shift_[t & 1];
      pt_t b = { dv[0], dv[1] }, c = { dv[2], dv[3] }, d = { dv[4], dv[5] };
      curve_to(&s, cur, b, c, d, lim);
      cur = d;
   }
   /* a tangent of zero length (a leaf whose first control point sits on its start) would take the stroker's cusp
    * branch (a fan over the pen): not restated; the chord stands in */
   for (int i = 1; i < s.m; i++)
   {
      if (tin[2 * i] == 0 && tin[2 * i + 1] == 0) { tin[2 * i] = pts[2 * i] - pts[2 * i - 2]; tin[2 * i + 1] = pts[2 * i + 1] - pts[2 * i - 1]; }
      if (tout[2 * i] == 0 && tout[2 * i + 1] == 0) { tout[2 * i] = tin[2 * i]; tout[2 * i + 1] = tin[2 * i + 1]; }
   }
   return s.overflow ? -1 : s.m;
}
