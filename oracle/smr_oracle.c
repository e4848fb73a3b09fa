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
 * wedge (miter quad or bevel triangle) on the outer side of every join -- unless a join's
 * inner corner pokes through a short neighbouring segment (orc_stroke_risky below): then
 * the polygon itself is emitted (orc_stroke_outline_edges). ---- */

typedef struct { int32_t x, y; } pt_t;
static int64_t turn2(pt_t a, pt_t b, pt_t c) { return (int64_t) (b.x - a.x) * (c.y - b.y) - (int64_t) (b.y - a.y) * (c.x - b.x); }

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

/* join of two faces at one vertex (outer_join / inner_join of cairo-path-stroke-polygon.c): which side is
 * the outer one and, for a miter inside the limit, the miter point.  That stroker REPLACES the incoming
 * face's outer corner by the miter point and never adds the outgoing one: the outer contour runs from miter
 * point to miter point, the inner one through the incoming corner, the vertex itself and the outgoing corner. */
typedef struct { int kind; int clockwise; pt_t m; } join_t;     /* kind 0: faces coincide, 1: miter, 2: bevel */

static join_t stroke_join(const face_t *in, const face_t *out)
{
   join_t j; j.kind = 0; j.clockwise = 0; j.m = in->point;
   if (in->cw.x == out->cw.x && in->cw.y == out->cw.y && in->ccw.x == out->ccw.x && in->ccw.y == out->ccw.y) return j;
   /* _cairo_slope_compare(in, out) < 0  <=>  in.dy*out.dx < out.dy*in.dx */
   int64_t l = (int64_t) in->ddy * out->ddx, r = (int64_t) out->ddy * in->ddx;
   if (l != r) j.clockwise = l < r;
   else j.clockwise = (in->ddx > 0 || (in->ddx == 0 && in->ddy > 0));   /* anti-parallel */
   pt_t inpt = j.clockwise ? in->cw : in->ccw, outpt = j.clockwise ? out->cw : out->ccw;
   double in_dot_out = in->sx * out->sx + in->sy * out->sy;
   double ml = 10.0;
   j.kind = 2;
   if (2 <= ml * ml * (1 + in_dot_out))
   {
      double x1 = inpt.x / 256.0, y1 = inpt.y / 256.0, dx1 = in->sx, dy1 = in->sy;
      double x2 = outpt.x / 256.0, y2 = outpt.y / 256.0, dx2 = out->sx, dy2 = out->sy;
      double my = (((x2 - x1) * dy1 * dy2 - y2 * dx2 * dy1 + y1 * dx1 * dy2) / (dx1 * dy2 - dx2 * dy1));
      double mx;
      if (fabs(dy1) >= fabs(dy2)) mx = (my - y1) * dx1 / dy1 + x1;
      else mx = (my - y2) * dx2 / dy2 + x2;
      double ix = in->point.x / 256.0, iy = in->point.y / 256.0;
      double fdx1 = x1 - ix, fdy1 = y1 - iy, fdx2 = x2 - ix, fdy2 = y2 - iy, mdx = mx - ix, mdy = my - iy;
      if (sgn_cmp(fdx1, fdy1, mdx, mdy) != sgn_cmp(fdx2, fdy2, mdx, mdy))
      {
         j.kind = 1;
         j.m.x = orc_fixed_from_double(mx); j.m.y = orc_fixed_from_double(my);
      }
   }
   return j;
}

/* The polygon of that stroker is cut into closed convex pieces whose outer edges are exactly its edges and
 * whose other edges join vertices it has too, so that the scan converter makes the same decision per row:
 *   body of a segment  = quad over its four face corners, an outer corner replaced by the miter point
 *   join, miter        = triangle (vertex, incoming inner corner, outgoing inner corner)
 *   join, bevel        = quad (vertex, incoming outer, outgoing outer, incoming inner corner)            */
int orc_stroke_edges(const int32_t *xy, int n, double hw_pt, int dpi, tor_edge_t *out)
{
   double s = (double) dpi / 72;
   int cnt = 0;
   /* drop the degenerate segments first: every segment then knows its neighbours */
   pt_t *p = malloc(sizeof(pt_t) * (n > 0 ? n : 1));
   int m = 0;
   for (int i = 0; i < n; i++)
   {
      pt_t q = { xy[2 * i], xy[2 * i + 1] };
      if (m == 0 || p[m - 1].x != q.x || p[m - 1].y != q.y) p[m++] = q;
   }
   face_t prev_end;
   for (int i = 0; i + 1 < m; i++)
   {
      int32_t ddx = p[i + 1].x - p[i].x, ddy = p[i + 1].y - p[i].y;
      face_t start, end;
      compute_face(p[i], ddx, ddy, hw_pt, s, &start);
      end = start;
      end.point = p[i + 1];
      end.ccw.x += ddx; end.ccw.y += ddy; end.cw.x += ddx; end.cw.y += ddy;
      pt_t c_ccw0 = start.ccw, c_cw0 = start.cw, c_ccw1 = end.ccw, c_cw1 = end.cw;
      if (i > 0)
      {
         join_t j = stroke_join(&prev_end, &start);
         pt_t in_inner = j.clockwise ? prev_end.ccw : prev_end.cw, out_inner = j.clockwise ? start.ccw : start.cw;
         pt_t in_outer = j.clockwise ? prev_end.cw : prev_end.ccw, out_outer = j.clockwise ? start.cw : start.ccw;
         if (j.kind == 1)
         {
            if (j.clockwise) c_cw0 = j.m; else c_ccw0 = j.m;
            pt_t q[3] = { p[i], in_inner, out_inner };
            cnt += emit_poly(q, 3, out + cnt);
         }
         else if (j.kind == 2)
         {
            pt_t q[4] = { p[i], in_outer, out_outer, in_inner };
            cnt += emit_poly(q, 4, out + cnt);
         }
      }
      if (i + 2 < m)
      {
         face_t nstart;
         compute_face(p[i + 1], p[i + 2].x - p[i + 1].x, p[i + 2].y - p[i + 1].y, hw_pt, s, &nstart);
         join_t j = stroke_join(&end, &nstart);
         if (j.kind == 1) { if (j.clockwise) c_cw1 = j.m; else c_ccw1 = j.m; }
      }
      pt_t q[4] = { c_ccw0, c_ccw1, c_cw1, c_cw0 };
      cnt += emit_poly(q, 4, out + cnt);
      prev_end = end;
   }
   free(p);
   return cnt;
}

/* The same stroker's polygon ITSELF, edge by edge: the contour on the cw side forward (face corner or miter point to face
 * corner or miter point; at a bevel the two outer corners; at an inner join the incoming corner, the vertex, the outgoing
 * corner), the butt cap, the contour on the ccw side backward, the other cap.  Filled non-zero it IS what libcairo fills;
 * the pieces above are its union-of-convex-parts form, which differs where a negatively wound loop cancels coverage: an
 * inner join whose corner pokes out of the neighbouring segment's body (segments shorter than the stroke is wide, hairpins). */
int orc_stroke_outline_edges(const int32_t *xy, int n, double hw_pt, int dpi, tor_edge_t *out)
{
   double s = (double) dpi / 72;
   int cnt = 0;
   pt_t *p = malloc(sizeof(pt_t) * (n > 0 ? n : 1));
   int m = 0;
   for (int i = 0; i < n; i++)
   {
      pt_t q = { xy[2 * i], xy[2 * i + 1] };
      if (m == 0 || p[m - 1].x != q.x || p[m - 1].y != q.y) p[m++] = q;
   }
   #define EDGE(a, b) do { out[cnt].x1 = (a).x; out[cnt].y1 = (a).y; out[cnt].x2 = (b).x; out[cnt].y2 = (b).y; cnt++; } while (0)
   face_t prev_end;
   for (int i = 0; i + 1 < m; i++)
   {
      int32_t ddx = p[i + 1].x - p[i].x, ddy = p[i + 1].y - p[i].y;
      face_t start, end;
      compute_face(p[i], ddx, ddy, hw_pt, s, &start);
      end = start;
      end.point = p[i + 1];
      end.ccw.x += ddx; end.ccw.y += ddy; end.cw.x += ddx; end.cw.y += ddy;
      pt_t a_cw = start.cw, a_ccw = start.ccw, b_cw = end.cw, b_ccw = end.ccw;
      if (i > 0)
      {
         join_t j = stroke_join(&prev_end, &start);
         if (j.kind != 0)
         {
            if (j.clockwise)
            {
               /* cw is the outer side */
               if (j.kind == 1) a_cw = j.m; else EDGE(prev_end.cw, start.cw);
               EDGE(start.ccw, p[i]); EDGE(p[i], prev_end.ccw);
            }
            else
            {
               if (j.kind == 1) a_ccw = j.m; else EDGE(start.ccw, prev_end.ccw);
               EDGE(prev_end.cw, p[i]); EDGE(p[i], start.cw);
            }
         }
      }
      else EDGE(start.ccw, start.cw);
      if (i + 2 < m)
      {
         face_t nstart;
         compute_face(p[i + 1], p[i + 2].x - p[i + 1].x, p[i + 2].y - p[i + 1].y, hw_pt, s, &nstart);
         join_t j = stroke_join(&end, &nstart);
         if (j.kind == 1) { if (j.clockwise) b_cw = j.m; else b_ccw = j.m; }
      }
      else EDGE(end.cw, end.ccw);
      EDGE(a_cw, b_cw);
      EDGE(b_ccw, a_ccw);
      prev_end = end;
   }
   #undef EDGE
   free(p);
   return cnt;
}

/* Does the union of the pieces equal the polygon?  It does while at every join the inner corner of each face lies within
 * the body of the OTHER segment (then the triangle vertex / incoming corner / outgoing corner is inside both bodies and the
 * loop the inner contour makes there is wound like the rest).  A corner that pokes out of a neighbour shorter than the
 * stroke is wide makes a loop wound the other way, which cancels coverage: such a way is "risky" and takes the polygon. */
static int inside4(pt_t a, pt_t b, pt_t c, pt_t d, pt_t q)
{
   int64_t t0 = turn2(a, b, q), t1 = turn2(b, c, q), t2 = turn2(c, d, q), t3 = turn2(d, a, q);
   return (t0 >= 0 && t1 >= 0 && t2 >= 0 && t3 >= 0) || (t0 <= 0 && t1 <= 0 && t2 <= 0 && t3 <= 0);
}

int orc_stroke_risky(const int32_t *xy, int n, double hw_pt, int dpi)
{
   double s = (double) dpi / 72;
   pt_t *p = malloc(sizeof(pt_t) * (n > 0 ? n : 1));
   int m = 0, risky = 0;
   for (int i = 0; i < n; i++)
   {
      pt_t q = { xy[2 * i], xy[2 * i + 1] };
      if (m == 0 || p[m - 1].x != q.x || p[m - 1].y != q.y) p[m++] = q;
   }
   /* a corner reaches hw * sin(turn) <= hw along the other segment: only a segment shorter than the half width (and the few
    * 1/256 px of the corners' rounding) can be poked through.  The integer test comes first and is part of the rule: joins of
    * longer segments are never looked at (their corners may sit a rounding step outside a neighbour that is nearly in line) */
   const int64_t thr = (int64_t) ceil(hw_pt * s * 256.0) + 4, thr2 = thr * thr;
   for (int i = 1; i + 1 < m && !risky; i++)
   {
      face_t a0, a1, b0, b1;
      int32_t adx = p[i].x - p[i - 1].x, ady = p[i].y - p[i - 1].y, bdx = p[i + 1].x - p[i].x, bdy = p[i + 1].y - p[i].y;
      if ((int64_t) adx * adx + (int64_t) ady * ady >= thr2 && (int64_t) bdx * bdx + (int64_t) bdy * bdy >= thr2) continue;
      compute_face(p[i - 1], adx, ady, hw_pt, s, &a0);
      a1 = a0; a1.point = p[i]; a1.ccw.x += adx; a1.ccw.y += ady; a1.cw.x += adx; a1.cw.y += ady;
      compute_face(p[i], bdx, bdy, hw_pt, s, &b0);
      b1 = b0; b1.ccw.x += bdx; b1.ccw.y += bdy; b1.cw.x += bdx; b1.cw.y += bdy;
      join_t j = stroke_join(&a1, &b0);
      if (j.kind == 0) continue;
      pt_t in_inner = j.clockwise ? a1.ccw : a1.cw, out_inner = j.clockwise ? b0.ccw : b0.cw;
      if (!inside4(b0.ccw, b1.ccw, b1.cw, b0.cw, in_inner) || !inside4(a0.ccw, a1.ccw, a1.cw, a0.cw, out_inner)) risky = 1;
   }
   free(p);
   return risky;
}

/* Where a curve bends within less than half the line width, the inner corners of two consecutive faces change
 * places and the quad over the four face corners crosses itself.  libcairo resolves such spots through the winding
 * of its whole contour; the pieces here have to be convex (the GPU path takes a piece as ONE interval per sub-row),
 * so a quad that is not convex is replaced by the convex hull of its corners -- a deviation inside the stroke's own
 * inner side, measured in tests/test_oracle_cairo.py. */
static int64_t turn(pt_t a, pt_t b, pt_t c) { return (int64_t) (b.x - a.x) * (c.y - b.y) - (int64_t) (b.y - a.y) * (c.x - b.x); }

static int convex4(const pt_t *q)
{
   int pos = 0, neg = 0;
   for (int k = 0; k < 4; k++)
   {
      int64_t t = turn(q[k], q[(k + 1) & 3], q[(k + 2) & 3]);
      if (t > 0) pos = 1;
      if (t < 0) neg = 1;
   }
   return !(pos && neg);
}

static int hull4(const pt_t *q, pt_t *h)
{
   const pt_t o1[4] = { q[0], q[1], q[3], q[2] }, o2[4] = { q[0], q[2], q[1], q[3] };
   const pt_t *pick = convex4(q) ? q : convex4(o1) ? o1 : convex4(o2) ? o2 : NULL;
   if (pick) { memcpy(h, pick, sizeof(pt_t) * 4); return 4; }
   for (int k = 0; k < 4; k++)
   {
      pt_t t[3]; int m = 0;
      for (int i = 0; i < 4; i++) if (i != k) t[m++] = q[i];
      int64_t a = turn(t[0], t[1], q[k]), b = turn(t[1], t[2], q[k]), c = turn(t[2], t[0], q[k]);
      if ((a >= 0 && b >= 0 && c >= 0) || (a <= 0 && b <= 0 && c <= 0)) { memcpy(h, t, sizeof(t)); return 3; }
   }
   memcpy(h, q, sizeof(pt_t) * 4);
   return 4;
}

/* The stroke of a decomposed curve (curve_to / spline_to of cairo-path-stroke-polygon.c): the same contour
 * stroker, but the face at every point looks along the curve's TANGENT there (tin: the face closing the segment
 * that arrives, tout: the face opening the segment that leaves; they differ only where two Beziers meet, and
 * only there is a join).  With tin[i+1] = tout[i] = the chord this is orc_stroke_edges. */
int orc_stroke_edges_tan(const int32_t *xy, const int32_t *tin, const int32_t *tout, int m, double hw_pt, int dpi, tor_edge_t *out)
{
   double s = (double) dpi / 72;
   int cnt = 0;
   for (int i = 0; i + 1 < m; i++)
   {
      pt_t p0 = { xy[2 * i], xy[2 * i + 1] }, p1 = { xy[2 * i + 2], xy[2 * i + 3] };
      face_t start, end;
      compute_face(p0, tout[2 * i], tout[2 * i + 1], hw_pt, s, &start);
      compute_face(p1, tin[2 * i + 2], tin[2 * i + 3], hw_pt, s, &end);
      pt_t c_ccw0 = start.ccw, c_cw0 = start.cw, c_ccw1 = end.ccw, c_cw1 = end.cw;
      if (i > 0)
      {
         face_t prev_end;
         compute_face(p0, tin[2 * i], tin[2 * i + 1], hw_pt, s, &prev_end);
         join_t j = stroke_join(&prev_end, &start);
         pt_t in_inner = j.clockwise ? prev_end.ccw : prev_end.cw, out_inner = j.clockwise ? start.ccw : start.cw;
         pt_t in_outer = j.clockwise ? prev_end.cw : prev_end.ccw, out_outer = j.clockwise ? start.cw : start.ccw;
         if (j.kind == 1)
         {
            if (j.clockwise) c_cw0 = j.m; else c_ccw0 = j.m;
            pt_t q[3] = { p0, in_inner, out_inner };
            cnt += emit_poly(q, 3, out + cnt);
         }
         else if (j.kind == 2)
         {
            pt_t q[4] = { p0, in_outer, out_outer, in_inner };
            cnt += emit_poly(q, 4, out + cnt);
         }
      }
      if (i + 2 < m)
      {
         face_t nstart;
         compute_face(p1, tout[2 * i + 2], tout[2 * i + 3], hw_pt, s, &nstart);
         join_t j = stroke_join(&end, &nstart);
         if (j.kind == 1) { if (j.clockwise) c_cw1 = j.m; else c_ccw1 = j.m; }
      }
      pt_t q[4] = { c_ccw0, c_ccw1, c_cw1, c_cw0 }, h[4];
      int nh = hull4(q, h);
      cnt += emit_poly(h, nh, out + cnt);
   }
   return cnt;
}

/* the same for the stroke polygon itself (orc_stroke_outline_edges): libcairo's spline_to adds the two corners of the face at
 * every point of a decomposed curve to its contours, and joins only where two Beziers meet */
int orc_stroke_outline_edges_tan(const int32_t *xy, const int32_t *tin, const int32_t *tout, int m, double hw_pt, int dpi, tor_edge_t *out)
{
   double s = (double) dpi / 72;
   int cnt = 0;
   #define EDGE(a, b) do { out[cnt].x1 = (a).x; out[cnt].y1 = (a).y; out[cnt].x2 = (b).x; out[cnt].y2 = (b).y; cnt++; } while (0)
   for (int i = 0; i + 1 < m; i++)
   {
      pt_t p0 = { xy[2 * i], xy[2 * i + 1] }, p1 = { xy[2 * i + 2], xy[2 * i + 3] };
      face_t start, end;
      compute_face(p0, tout[2 * i], tout[2 * i + 1], hw_pt, s, &start);
      compute_face(p1, tin[2 * i + 2], tin[2 * i + 3], hw_pt, s, &end);
      pt_t a_ccw = start.ccw, a_cw = start.cw, b_ccw = end.ccw, b_cw = end.cw;
      if (i > 0)
      {
         face_t prev_end;
         compute_face(p0, tin[2 * i], tin[2 * i + 1], hw_pt, s, &prev_end);
         join_t j = stroke_join(&prev_end, &start);
         if (j.kind != 0)
         {
            if (j.clockwise)
            {
               if (j.kind == 1) a_cw = j.m; else EDGE(prev_end.cw, start.cw);
               EDGE(start.ccw, p0); EDGE(p0, prev_end.ccw);
            }
            else
            {
               if (j.kind == 1) a_ccw = j.m; else EDGE(start.ccw, prev_end.ccw);
               EDGE(prev_end.cw, p0); EDGE(p0, start.cw);
            }
         }
      }
      else EDGE(start.ccw, start.cw);
      if (i + 2 < m)
      {
         face_t nstart;
         compute_face(p1, tout[2 * i + 2], tout[2 * i + 3], hw_pt, s, &nstart);
         join_t j = stroke_join(&end, &nstart);
         if (j.kind == 1) { if (j.clockwise) b_cw = j.m; else b_ccw = j.m; }
      }
      else EDGE(end.cw, end.ccw);
      EDGE(a_cw, b_cw);
      EDGE(b_ccw, a_ccw);
   }
   #undef EDGE
   return cnt;
}

/* ---------------- dashed strokes ----------------
 * libcairo 1.16 sends a stroke with a dash array through the older stroker (cairo-path-stroke.c,
 * _cairo_path_fixed_stroke_dashed_to_polygon): every "on" stretch of a segment is one sub-edge
 * with faces of its own, consecutive sub-edges across a vertex are joined (miter, limit 10), all
 * other ends are capped (butt: the end edge; round: a fan over the pen's vertices).  The edges it
 * collects bound exactly the union of the closed convex pieces emitted here (quad per sub-edge,
 * wedge per join, half disc per round cap, cut into pieces of at most four edges), which is the
 * form the tor restatement and the GPU path take them in.                                     */

typedef struct { int32_t dx, dy; } slope_t;

static int slope_cmp(slope_t a, slope_t b)          /* _cairo_slope_compare */
{
   int64_t l = (int64_t) a.dy * b.dx, r = (int64_t) b.dy * a.dx;
   if (l != r) return l < r ? -1 : 1;
   if (a.dx == 0 && a.dy == 0 && b.dx == 0 && b.dy == 0) return 0;
   if (a.dx == 0 && a.dy == 0) return 1;
   if (b.dx == 0 && b.dy == 0) return -1;
   if ((a.dx ^ b.dx) < 0 || (a.dy ^ b.dy) < 0)
      return (a.dx > 0 || (a.dx == 0 && a.dy > 0)) ? -1 : 1;
   return 0;
}

/* the stroker's pen: a polygon of an even number of vertices around the half line width */
int orc_pen(double hw_pt, int dpi, int32_t *vx, int32_t *vy, int cap)
{
   const double s = (double) dpi / 72, tol = 0.1;
   double major = hw_pt * sqrt(s * s);
   int n;
   if (tol >= 4 * major) n = 1;
   else if (tol >= major) n = 4;
   else
   {
      n = (int) ceil(2 * M_PI / acos(1 - tol / major));
      if (n % 2) n++;
      if (n < 4) n = 4;
   }
   if (n > cap) return -n;
   for (int i = 0; i < n; i++)
   {
      double th = 2 * M_PI * i / (double) n;
      vx[i] = orc_fixed_from_double(hw_pt * cos(th) * s);
      vy[i] = orc_fixed_from_double(hw_pt * sin(th) * s);
   }
   return n;
}

typedef struct
{
   double s, hw;
   int round_cap, npen;
   int32_t pvx[ORC_PEN_MAX], pvy[ORC_PEN_MAX];
   slope_t pcw[ORC_PEN_MAX], pccw[ORC_PEN_MAX];
   tor_edge_t *out; int cnt;                       /* out == NULL: count only */
} dasher_t;

typedef struct { pt_t ccw, cw, point; double ux, uy; slope_t dev; } dface_t;

static void d_emit(dasher_t *d, const pt_t *q, int n)
{
   tor_edge_t tmp[4];
   int k = emit_poly(q, n, tmp);
   if (d->out) memcpy(d->out + d->cnt, tmp, sizeof(tor_edge_t) * k);
   d->cnt += k;
}

static void d_face(const dasher_t *d, pt_t p, slope_t dev, double ux, double uy, dface_t *f)
{
   double fx = -uy * d->hw, fy = ux * d->hw;
   fx *= d->s; fy *= d->s;
   int32_t ox = orc_fixed_from_double(fx), oy = orc_fixed_from_double(fy);
   f->point = p; f->ux = ux; f->uy = uy; f->dev = dev;
   f->ccw.x = p.x + ox; f->ccw.y = p.y + oy; f->cw.x = p.x - ox; f->cw.y = p.y - oy;
}

static int pen_active_ccw(const dasher_t *d, slope_t sl)       /* _cairo_pen_find_active_ccw_vertex_index */
{
   slope_t rev = { -sl.dx, -sl.dy };
   int i;
   for (i = d->npen - 1; i >= 0; i--)
      if (slope_cmp(d->pccw[i], rev) >= 0 && slope_cmp(d->pcw[i], rev) < 0) break;
   if (i < 0) i = d->npen - 1;
   return i;
}

static int d_fan_points(const dasher_t *d, pt_t mid, slope_t in, slope_t outv, pt_t from, pt_t to, pt_t *poly);

/* round cap at face f, f pointing out of the stroke: half disc ccw -> pen vertices -> cw */
static void d_round_cap(dasher_t *d, const dface_t *f)
{
   slope_t in = f->dev, outv = { -f->dev.dx, -f->dev.dy };
   pt_t poly[ORC_PEN_MAX + 2];
   int m = d_fan_points(d, f->point, in, outv, f->ccw, f->cw, poly);
   if (m <= 2) return;
   /* convex: fan of quads from the first point */
   for (int k = 1; k + 1 < m; k += 2)
   {
      pt_t q[4] = { poly[0], poly[k], poly[k + 1], poly[k + 2 < m ? k + 2 : k + 1] };
      d_emit(d, q, k + 2 < m ? 4 : 3);
   }
}

static void d_cap_trailing(dasher_t *d, const dface_t *f) { if (d->round_cap) d_round_cap(d, f); }

static void d_cap_leading(dasher_t *d, const dface_t *f)
{
   if (!d->round_cap) return;
   dface_t r = *f;
   r.ux = -f->ux; r.uy = -f->uy; r.dev.dx = -f->dev.dx; r.dev.dy = -f->dev.dy;
   r.cw = f->ccw; r.ccw = f->cw;
   d_round_cap(d, &r);
}

static void d_join(dasher_t *d, const dface_t *in, const dface_t *out)
{
   if (in->cw.x == out->cw.x && in->cw.y == out->cw.y && in->ccw.x == out->ccw.x && in->ccw.y == out->ccw.y) return;
   slope_t so = { out->cw.x - out->point.x, out->cw.y - out->point.y }, si = { in->cw.x - in->point.x, in->cw.y - in->point.y };
   int clockwise = slope_cmp(so, si) < 0;
   pt_t inpt = clockwise ? in->ccw : in->cw, outpt = clockwise ? out->ccw : out->cw;
   double in_dot_out = -in->ux * out->ux + -in->uy * out->uy, ml = 10.0;
   if (2 <= ml * ml * (1 - in_dot_out))
   {
      double x1 = inpt.x / 256.0, y1 = inpt.y / 256.0, dx1 = in->ux * d->s, dy1 = in->uy * d->s;
      double x2 = outpt.x / 256.0, y2 = outpt.y / 256.0, dx2 = out->ux * d->s, dy2 = out->uy * d->s;
      double my = (((x2 - x1) * dy1 * dy2 - y2 * dx2 * dy1 + y1 * dx1 * dy2) / (dx1 * dy2 - dx2 * dy1));
      double mx;
      if (fabs(dy1) >= fabs(dy2)) mx = (my - y1) * dx1 / dy1 + x1;
      else mx = (my - y2) * dx2 / dy2 + x2;
      double ix = in->point.x / 256.0, iy = in->point.y / 256.0;
      double fdx1 = x1 - ix, fdy1 = y1 - iy, fdx2 = x2 - ix, fdy2 = y2 - iy, mdx = mx - ix, mdy = my - iy;
      if (sgn_cmp(fdx1, fdy1, mdx, mdy) != sgn_cmp(fdx2, fdy2, mdx, mdy))
      {
         pt_t q[4] = { in->point, inpt, { orc_fixed_from_double(mx), orc_fixed_from_double(my) }, outpt };
         d_emit(d, q, 4);
         return;
      }
   }
   pt_t q[3] = { in->point, inpt, outpt };
   d_emit(d, q, 3);
}

/* the pen's vertices the stroker passes between the directions `in` and `outv` on the side of the ccw corners (_tessellate_fan
 * with _cairo_pen_find_active_ccw_vertices): poly = from, vertices..., to; a round cap is the case outv = -in */
static int d_fan_points(const dasher_t *d, pt_t mid, slope_t in, slope_t outv, pt_t from, pt_t to, pt_t *poly)
{
   const int n = d->npen;
   int m = 0;
   poly[m++] = from;
   if (n > 0)
   {
      int start = pen_active_ccw(d, in);
      if (slope_cmp(d->pccw[start], in) < 0) start = start == 0 ? n - 1 : start - 1;
      int stop = pen_active_ccw(d, outv), none = 0;
      if (slope_cmp(d->pcw[stop], outv) > 0)
      {
         stop = stop + 1 == n ? 0 : stop + 1;
         if (slope_cmp(d->pccw[stop], in) < 0) none = 1;
      }
      int npoints = start - stop;
      stop = stop == 0 ? n - 1 : stop - 1;
      if (npoints < 0) npoints += n;
      if (!none && npoints > 1)
         for (int i = start; i != stop; i = i == 0 ? n - 1 : i - 1)
         {
            pt_t p = { mid.x + d->pvx[i], mid.y + d->pvy[i] };
            poly[m++] = p;
         }
   }
   poly[m++] = to;
   return m;
}

/* a ROUND join (what the dashing stroker puts between the segments of a decomposed curve): the fan over the pen's vertices on
 * the outer side, from the pivot; with no vertex in between it is the bevel's triangle */
static void d_join_round(dasher_t *d, const dface_t *in, const dface_t *out)
{
   if (in->cw.x == out->cw.x && in->cw.y == out->cw.y && in->ccw.x == out->ccw.x && in->ccw.y == out->ccw.y) return;
   slope_t so = { out->cw.x - out->point.x, out->cw.y - out->point.y }, si = { in->cw.x - in->point.x, in->cw.y - in->point.y };
   int clockwise = slope_cmp(so, si) < 0;
   pt_t poly[ORC_PEN_MAX + 3];
   int m;
   if (clockwise) m = d_fan_points(d, in->point, in->dev, out->dev, in->ccw, out->ccw, poly + 1);
   else
   {
      /* the other side: the same walk along the reversed path, whose ccw corners are this one's cw corners */
      slope_t ri = { -out->dev.dx, -out->dev.dy }, ro = { -in->dev.dx, -in->dev.dy };
      m = d_fan_points(d, in->point, ri, ro, out->cw, in->cw, poly + 1);
   }
   poly[0] = in->point;
   m++;
   for (int k = 1; k + 1 < m; k += 2)
   {
      pt_t q[4] = { poly[0], poly[k], poly[k + 1], poly[k + 2 < m ? k + 2 : k + 1] };
      d_emit(d, q, k + 2 < m ? 4 : 3);
   }
}

/* the dash state along a path (_cairo_stroker_dash_start with offset 0, _cairo_stroker_dash_step) and the stroker's faces */
typedef struct
{
   const double *dash; int ndash, di, on;
   double dash_remain;
   int has_first, has_cur;
   dface_t first, cur;
   pt_t p1;
} dstate_t;

static void d_pen_init(dasher_t *D, double hw_pt, int dpi, int need_pen)
{
   D->npen = need_pen ? orc_pen(hw_pt, dpi, D->pvx, D->pvy, ORC_PEN_MAX) : 0;
   for (int i = 0; i < D->npen; i++)
   {
      int pr = i == 0 ? D->npen - 1 : i - 1, nx = i + 1 == D->npen ? 0 : i + 1;
      D->pcw[i].dx = D->pvx[i] - D->pvx[pr]; D->pcw[i].dy = D->pvy[i] - D->pvy[pr];
      D->pccw[i].dx = D->pvx[nx] - D->pvx[i]; D->pccw[i].dy = D->pvy[nx] - D->pvy[i];
   }
}

/* _cairo_stroker_line_to_dashed; round_join: inside a decomposed curve */
static void d_line_to(dasher_t *D, dstate_t *S, pt_t p2, int round_join)
{
   const int starts_on = 1;
   const pt_t p1 = S->p1;
   if (p1.x == p2.x && p1.y == p2.y) return;
   slope_t dev = { p2.x - p1.x, p2.y - p1.y };
   double sdx = dev.dx / 256.0 * (1.0 / D->s), sdy = dev.dy / 256.0 * (1.0 / D->s);
   double mag = normalize_slope(&sdx, &sdy);
   double remain = mag;
   pt_t a = p1;
   while (remain)
   {
      double step = S->dash_remain < remain ? S->dash_remain : remain;
      remain -= step;
      double dx2 = sdx * (mag - remain) * D->s, dy2 = sdy * (mag - remain) * D->s;
      pt_t b = { orc_fixed_from_double(dx2) + p1.x, orc_fixed_from_double(dy2) + p1.y };
      if (S->on)
      {
         dface_t ss, se;
         d_face(D, a, dev, sdx, sdy, &ss);
         se = ss;
         if (a.x != b.x || a.y != b.y)
         {
            se.point = b;
            se.ccw.x += b.x - a.x; se.ccw.y += b.y - a.y; se.cw.x += b.x - a.x; se.cw.y += b.y - a.y;
            pt_t q[4] = { ss.ccw, se.ccw, se.cw, ss.cw };
            d_emit(D, q, 4);
         }
         if (S->has_cur) { if (round_join) d_join_round(D, &S->cur, &ss); else d_join(D, &S->cur, &ss); S->has_cur = 0; }
         else if (!S->has_first && starts_on) { S->first = ss; S->has_first = 1; }
         else d_cap_leading(D, &ss);
         if (remain) d_cap_trailing(D, &se);
         else { S->cur = se; S->has_cur = 1; }
      }
      else if (S->has_cur) { d_cap_trailing(D, &S->cur); S->has_cur = 0; }
      /* _cairo_stroker_dash_step */
      S->dash_remain -= step;
      if (S->dash_remain < 1.0 / 512)
      {
         if (++S->di == S->ndash) S->di = 0;
         S->on = !S->on;
         S->dash_remain += S->dash[S->di];
      }
      a = b;
   }
   if (S->on && !S->has_cur)
   {
      /* the segment ends exactly where a dash begins: a (degenerate) leading cap at the vertex */
      d_face(D, p2, dev, sdx, sdy, &S->cur);
      d_cap_leading(D, &S->cur);
      S->has_cur = 1;
   }
   S->p1 = p2;
}

static void d_state_init(dstate_t *S, const double *dash, int ndash, pt_t p0)
{
   S->dash = dash; S->ndash = ndash; S->di = 0; S->on = 1; S->dash_remain = dash[0];
   S->has_first = S->has_cur = 0; S->p1 = p0;
}

int orc_dash_edges(const int32_t *xy, int n, double hw_pt, int dpi, const double *dash, int ndash, int round_cap, tor_edge_t *out)
{
   dasher_t D;
   D.s = (double) dpi / 72; D.hw = hw_pt; D.round_cap = round_cap; D.out = out; D.cnt = 0;
   d_pen_init(&D, hw_pt, dpi, round_cap);
   if (D.npen < 0) return -1;
   dstate_t S;
   pt_t p0 = { xy[0], xy[1] };
   d_state_init(&S, dash, ndash, p0);
   for (int i = 1; i < n; i++)
   {
      pt_t p2 = { xy[2 * i], xy[2 * i + 1] };
      d_line_to(&D, &S, p2, 0);
   }
   /* _cairo_stroker_add_caps; a path of coincident points only draws nothing with our dash arrays
    * unless caps are round -- the reference's ways with >= 2 distinct nodes never get there */
   if (S.has_first) d_cap_leading(&D, &S.first);
   if (S.has_cur) d_cap_trailing(&D, &S.cur);
   return D.cnt;
}

/* The dashing stroker along a decomposed Bezier chain (_cairo_stroker_curve_to of libcairo 1.16's cairo-path-stroke.c): where a
 * curve begins and a dash is on, a face along the curve's initial tangent, joined to what came before with the context's join
 * (miter); the decomposed points are plain dashed line_to's, faces along the CHORDS, joined with ROUND joins; where the curve
 * ends, a face along its final tangent, round-joined to the last chord's.  flag: orc_curve_flatten_flags. */
int orc_dash_edges_curve(const int32_t *xy, const int32_t *tin, const int32_t *tout, const uint8_t *flag, int m, double hw_pt, int dpi,
                         const double *dash, int ndash, int round_cap, tor_edge_t *out)
{
   dasher_t D;
   D.s = (double) dpi / 72; D.hw = hw_pt; D.round_cap = round_cap; D.out = out; D.cnt = 0;
   d_pen_init(&D, hw_pt, dpi, 1);
   if (D.npen < 0) return -1;
   dstate_t S;
   pt_t p0 = { xy[0], xy[1] };
   d_state_init(&S, dash, ndash, p0);
   for (int i = 1; i < m;)
   {
      int j = i;
      while (j < m - 1 && !(flag[j] & 1)) j++;
      if (flag[j] & 2)
      {
         for (int k = i; k <= j; k++) { pt_t p = { xy[2 * k], xy[2 * k + 1] }; d_line_to(&D, &S, p, 0); }
         i = j + 1;
         continue;
      }
      if (S.on)
      {
         slope_t is = { tout[2 * (i - 1)], tout[2 * (i - 1) + 1] };
         double ux = is.dx / 256.0 * (1.0 / D.s), uy = is.dy / 256.0 * (1.0 / D.s);
         dface_t f;
         if (normalize_slope(&ux, &uy) != 0)
         {
            d_face(&D, S.p1, is, ux, uy, &f);
            if (S.has_cur) d_join(&D, &S.cur, &f);
            else if (!S.has_first) { S.first = f; S.has_first = 1; }
            S.cur = f; S.has_cur = 1;
         }
      }
      for (int k = i; k <= j; k++) { pt_t p = { xy[2 * k], xy[2 * k + 1] }; d_line_to(&D, &S, p, 1); }
      if (S.on)
      {
         slope_t fs = { tin[2 * j], tin[2 * j + 1] };
         double ux = fs.dx / 256.0 * (1.0 / D.s), uy = fs.dy / 256.0 * (1.0 / D.s);
         dface_t f;
         if (normalize_slope(&ux, &uy) != 0)
         {
            d_face(&D, S.p1, fs, ux, uy, &f);
            if (S.has_cur) d_join_round(&D, &S.cur, &f);
            S.cur = f; S.has_cur = 1;
         }
      }
      i = j + 1;
   }
   if (S.has_first) d_cap_leading(&D, &S.first);
   if (S.has_cur) d_cap_trailing(&D, &S.cur);
   return D.cnt;
}

/* A solid stroke with ROUND caps (add_cap / add_fan of cairo-path-stroke-polygon.c put the pen's vertices between the two corners
 * of the path's first and last face): the pieces of orc_stroke_edges -- always the pieces, never the polygon itself -- and, as the
 * dasher's round caps are cut, a fan of quads over the pen's vertices beyond the first segment's start and the last one's end */
int orc_stroke_edges_round(const int32_t *xy, int n, double hw_pt, int dpi, tor_edge_t *out)
{
   int cnt = orc_stroke_edges(xy, n, hw_pt, dpi, out);
   dasher_t D;
   D.s = (double) dpi / 72; D.hw = hw_pt; D.round_cap = 1; D.out = out; D.cnt = cnt;
   D.npen = orc_pen(hw_pt, dpi, D.pvx, D.pvy, ORC_PEN_MAX);
   if (D.npen < 0) return -1;
   for (int i = 0; i < D.npen; i++)
   {
      int pr = i == 0 ? D.npen - 1 : i - 1, nx = i + 1 == D.npen ? 0 : i + 1;
      D.pcw[i].dx = D.pvx[i] - D.pvx[pr]; D.pcw[i].dy = D.pvy[i] - D.pvy[pr];
      D.pccw[i].dx = D.pvx[nx] - D.pvx[i]; D.pccw[i].dy = D.pvy[nx] - D.pvy[i];
   }
   int has_first = 0;
   dface_t first, last;
   for (int i = 0; i + 1 < n; i++)
   {
      pt_t p1 = { xy[2 * i], xy[2 * i + 1] }, p2 = { xy[2 * i + 2], xy[2 * i + 3] };
      if (p1.x == p2.x && p1.y == p2.y) continue;
      face_t f;
      compute_face(p1, p2.x - p1.x, p2.y - p1.y, hw_pt, D.s, &f);
      dface_t d;
      d.ccw = f.ccw; d.cw = f.cw; d.point = p1; d.ux = f.sx; d.uy = f.sy; d.dev.dx = p2.x - p1.x; d.dev.dy = p2.y - p1.y;
      if (!has_first) { first = d; has_first = 1; }
      last = d;
      last.point = p2;
      last.ccw.x += d.dev.dx; last.ccw.y += d.dev.dy; last.cw.x += d.dev.dx; last.cw.y += d.dev.dy;
   }
   if (has_first) { d_cap_leading(&D, &first); d_cap_trailing(&D, &last); }
   return D.cnt;
}

/* the same along a decomposed curve: the tangent stroker's pieces (orc_stroke_edges_tan) and a half disc at either end, the end
 * faces along the curve's first and last tangent */
int orc_stroke_edges_round_tan(const int32_t *xy, const int32_t *tin, const int32_t *tout, int m, double hw_pt, int dpi, tor_edge_t *out)
{
   if (m < 2) return 0;
   int cnt = orc_stroke_edges_tan(xy, tin, tout, m, hw_pt, dpi, out);
   dasher_t D;
   D.s = (double) dpi / 72; D.hw = hw_pt; D.round_cap = 1; D.out = out; D.cnt = cnt;
   d_pen_init(&D, hw_pt, dpi, 1);
   if (D.npen < 0) return -1;
   face_t f;
   dface_t first, last;
   pt_t p0 = { xy[0], xy[1] }, pe = { xy[2 * m - 2], xy[2 * m - 1] };
   compute_face(p0, tout[0], tout[1], hw_pt, D.s, &f);
   first.ccw = f.ccw; first.cw = f.cw; first.point = p0; first.ux = f.sx; first.uy = f.sy; first.dev.dx = tout[0]; first.dev.dy = tout[1];
   compute_face(pe, tin[2 * m - 2], tin[2 * m - 1], hw_pt, D.s, &f);
   last.ccw = f.ccw; last.cw = f.cw; last.point = pe; last.ux = f.sx; last.uy = f.sy; last.dev.dx = tin[2 * m - 2]; last.dev.dy = tin[2 * m - 1];
   d_cap_leading(&D, &first); d_cap_trailing(&D, &last);
   return D.cnt;
}

/* ---------------- colour + compositing ---------------- */

static uint32_t d2s(double d)   /* _cairo_color_double_to_short as libcairo 1.16 has it (cairo-color.c): round to nearest of d * 65535.
                                 * Pinned against the real libcairo on 40 of the ~0.8 % of translucent colours for which the older
                                 * form (d * 65536 truncated, minus its carry) gives another 8-bit channel (tests/test_oracle_cairo.py) */
{
   return (uint32_t) (uint16_t) (d * 65535.0 + 0.5);
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
   int img[4];           /* the image's rectangle on the chart, px (curve limits) */
   int cap_round;        /* line cap of the current rule's context (cairo_set_line_cap is never undone) */
   int32_t shift_x, shift_y;   /* 24.8: the translation of a replay at an offset (cairo_smr_cut_out), added to every device vertex */
   int win_lim_on; int32_t win_lim[2];      /* a replay at an offset (orc_canvas_set_shift): the sides of the image libcairo replays into, 24.8 */
   int fill_poly;        /* the edges of the next orc_layer_edges are a fill's, one per segment in the path's order (tor_set_path_order) */
   int no_cuts;          /* the edges of the next orc_layer_edges are a stroke's PIECES: their inner edges are not the polygon's, no cuts at the image's sides */
   const uint32_t *pat; int pat_w, pat_h;   /* orc_canvas_set_pattern: the source of orc_layer_edges is this image, repeated from the chart's origin */
};

orc_canvas_t *orc_canvas_new(int x0, int y0, int W, int H, int dpi, uint32_t bgcolor)
{
   orc_canvas_t *c = calloc(1, sizeof(*c));
   c->x0 = x0; c->y0 = y0; c->W = W; c->H = H; c->dpi = dpi;
   c->img[0] = x0; c->img[1] = y0; c->img[2] = x0 + W; c->img[3] = y0 + H;
   c->pix = malloc(sizeof(uint32_t) * W * H);
   c->layer = malloc(sizeof(uint32_t) * W * H);
   c->cov = malloc((size_t) W * H);
   uint32_t bg = orc_color_pixel(bgcolor);       /* paint onto a clear image == SOURCE */
   for (size_t i = 0; i < (size_t) W * H; i++) c->pix[i] = bg;
   return c;
}

/* A replay of the recording translated by (tx, ty) pt under cairo_scale(dpi/72) (cairo_smr_cut_out, src/smrules_cairo.c:1524-1529):
 * libcairo 1.16 transforms the recorded 24.8 path with the scale quantised to 24.8 and the matrix's translation x0 = tx * s
 * quantised SEPARATELY (_cairo_path_fixed_offset_and_scale: p = fixed_mul(scale, p) + fixed(x0)), so the shifted replay is the
 * unshifted device path plus one constant 24.8 vector; strokes are computed on the shifted path. */
/* the same window further down: the canvas and the replay's shift moved by whole rows (an atlas of windows holds window k at rows
 * k * pitch: a check that nothing depends on where a window stands).  After orc_canvas_set_shift. */
void orc_canvas_move_rows(orc_canvas_t *c, int rows) { c->y0 += rows; c->shift_y += rows * 256; c->img[1] += rows; c->img[3] += rows; }

void orc_canvas_set_shift(orc_canvas_t *c, double tx_pt, double ty_pt)
{
   double s = (double) c->dpi / 72;
   c->shift_x = orc_fixed_from_double(tx_pt * s); c->shift_y = orc_fixed_from_double(ty_pt * s);
   /* What the replay is limited to.  libcairo does not replay the recording onto the cut-out's image: the image compositor makes the
    * source's pixels first (_pixman_image_for_recording of cairo-image-source.c), in an image of its own that covers the SAMPLED area
    * of the source, and that area is whole units of the SOURCE, i.e. whole pt (_cairo_pattern_sampled_area: the centres of the corner
    * pixels through the pattern's matrix, 0.495 of a source unit further out for a bilinear filter, floor and ceil), taken back to
    * device pixels with floor and ceil again.  A polygon that leaves that image is limited to it (cairo-spans-compositor.c), and the
    * cuts at its left and right side add vertices, which decide the rows the scan converter sub-samples: at 300 dpi the sides lie 2 to
    * 6 px beyond the cut-out's own, wherever the anchor's fraction of a pt puts them.  (Top and bottom change no row inside the
    * cut-out.  Near the page's sides the sampled area is the recording's extents at most: left out.) */
   {
      const double X = -tx_pt;                                      /* the source coordinate of the cut-out's left side */
      const double sx1 = floor(0.5 / s + X - 0.495), sx2 = floor(((double) c->W - 0.5) / s + X + 0.495) + 1.0;
      c->win_lim_on = 1;
      c->win_lim[0] = (int32_t) floor((sx1 - X) * s) * 256; c->win_lim[1] = (int32_t) ceil((sx2 - X) * s) * 256;
      /* and the image a decomposed curve is tested against (draw_way_curve's limits), top and bottom likewise */
      const double Y = -ty_pt;
      const double sy1 = floor(0.5 / s + Y - 0.495), sy2 = floor(((double) c->H - 0.5) / s + Y + 0.495) + 1.0;
      c->img[0] = c->win_lim[0] / 256; c->img[2] = c->win_lim[1] / 256;
      c->img[1] = (int) floor((sy1 - Y) * s); c->img[3] = (int) ceil((sy2 - Y) * s);
   }
   orc_curve_set_shift(c->shift_x, c->shift_y);      /* curve.c keeps it for orc_curve_flatten (one shifted canvas at a time) */
}

void orc_canvas_set_image(orc_canvas_t *c, int x0, int y0, int x1, int y1) { c->img[0] = x0; c->img[1] = y0; c->img[2] = x1; c->img[3] = y1; }

/* _cairo_stroke_style_max_distance_from_path: butt caps 0.5, miter joins sqrt(2) * limit, times the width, in device px */
double orc_stroke_reach_px(double width_pt, int dpi)
{
   double e = 0.5;
   if (e < M_SQRT2 * 10.0) e = M_SQRT2 * 10.0;
   e *= width_pt;
   return e * hypot((double) dpi / 72, 0.0);
}

void orc_canvas_free(orc_canvas_t *c) { if (c->shift_x || c->shift_y) orc_curve_set_shift(0, 0); free(c->pix); free(c->layer); free(c->cov); free(c); }
const uint32_t *orc_canvas_pixels(const orc_canvas_t *c) { return c->pix; }

/* cairo_push_group (src/smrules_cairo.c:660): drawing goes to a clear ARGB32 layer */
void orc_rule_begin(orc_canvas_t *c)
{
   memset(c->layer, 0, sizeof(uint32_t) * c->W * c->H);
   c->layer_clear = 1;
   c->cap_round = 0;
}

/* cairo_pop_group_to_source + cairo_paint (src/smrules_cairo.c:203-205, :1081-1082) */
void orc_rule_end(orc_canvas_t *c)
{
   for (size_t i = 0; i < (size_t) c->W * c->H; i++) c->pix[i] = blend_over(c->layer[i], c->pix[i]);
}

/* The image fill of a rule (img_ini src/smrules_cairo.c:2691-2704: cairo_pattern_create_for_surface, matrix scale(1 / PX2PT_SCALE)
 * = one image pixel per chart pixel at angle 0, CAIRO_EXTEND_REPEAT; render_poly_line :989-997 / img_fill :2742-2751 fill the
 * closed ways with it): while a pattern is set, orc_layer_edges takes every pixel's source from the image (premultiplied ARGB32,
 * kept by the caller) and composites it as pixman does an image source through a mask: (s IN m) OVER d.  NULL: colours again. */
/* the next orc_layer_edges calls of a caller of its own (the compound fill of a directional rule) bring fills in the path's order */
void orc_canvas_set_fill_poly(orc_canvas_t *c, int on) { c->fill_poly = on; }

void orc_canvas_set_pattern(orc_canvas_t *c, const uint32_t *pix, int w, int h) { c->pat = pix; c->pat_w = w; c->pat_h = h; }

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
   {
      /* the clip of the replay is the recording surface's rectangle: whole pt (floor / ceil of the page in pt), through the
       * replay's 24.8 scale -- at 300 dpi a little MORE than the page, at 600 dpi a little less */
      const int32_t sc = orc_fixed_from_double((double) c->dpi / 72);
      const int32_t xl = fixed_mul((int32_t) floor((double) c->img[0] * 72 / c->dpi) * 256, sc), xr = fixed_mul((int32_t) ceil((double) c->img[2] * 72 / c->dpi) * 256, sc);
      /* (a replay at an offset, cairo_smr_cut_out: limited to the image libcairo makes of the source there, orc_canvas_set_shift) */
      if (c->win_lim_on && !c->no_cuts && getenv("ORC_NO_LIMITS") == NULL) tor_set_limits(1, c->win_lim[0], c->win_lim[1]);
      else
      tor_set_limits(getenv("ORC_NO_LIMITS") == NULL && !c->no_cuts && !c->shift_x && !c->shift_y, xl, xr);
   }
   tor_set_path_order(c->fill_poly);
   tor_coverage(e, n, nonzero, c->x0 + bx0, c->y0 + by0, bw, bh, c->cov);
   tor_set_path_order(0);
   tor_set_limits(0, 0, 0);
   for (int y = 0; y < bh; y++)
      for (int x = 0; x < bw; x++)
      {
         uint32_t m = c->cov[(size_t) y * bw + x];
         if (!m) continue;
         uint32_t *d = &c->layer[(size_t) (y + by0) * c->W + x + bx0];
         if (c->pat)
         {
            int u = (c->x0 + bx0 + x) % c->pat_w, v = (c->y0 + by0 + y) % c->pat_h;
            if (u < 0) u += c->pat_w;
            if (v < 0) v += c->pat_h;
            *d = blend_over_mask(c->pat[(size_t) v * c->pat_w + u], m, *d);
         }
         else *d = lerp ? blend_lerp(s, m, *d) : blend_over_mask(s, m, *d);
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
   tor_set_path_order(1);                                                           /* (only fills come this way) */
   tor_coverage(e, n, nonzero, c->x0 + bx0, c->y0 + by0, bw, bh, c->cov);      /* (straight onto an image: no clip, no limits) */
   tor_set_path_order(0);
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
   for (int i = 0; i < n; i++) { dev[2 * i] += c->shift_x; dev[2 * i + 1] += c->shift_y; }
   orc_direct_edges(c, e, orc_fill_edges(dev, n, e), 1, smr_col);
   free(dev); free(e);
}

uint32_t *orc_canvas_pixels_rw(orc_canvas_t *c) { return c->pix; }

/* mm -> pt; width 0 -> 1 device pixel (THINLINE): src/smrules_cairo.c:67-69, src/rdata.c:50-53 */
static double mm2wu(double mm, int dpi) { return mm == 0.0 ? 1.0 * 72 / dpi : mm * 72 / 25.4; }

/* cairo_smr_dash (:898-923) + cairo_stroke: dash lengths = mm2wu(border width) * ds[i]; rounddot and the pipe's
 * dots switch the context to round caps, and nothing ever switches it back (c->cap_round lives as long as
 * the rule's context) */
typedef struct { const int32_t *tin, *tout; const uint8_t *flag; } curve_faces_t;      /* a decomposed curve (orc_curve_flatten_flags), or NULL */
static void stroke_styled(orc_canvas_t *c, const int32_t *dev, int n, double width_pt, int style, double bwidth_mm,
                          const double *ds, int len, uint32_t col, const curve_faces_t *cf)
{
   double dash[4], l = 0;
   int nd = 0;
   for (; nd < len && nd < 4; nd++) { dash[nd] = mm2wu(bwidth_mm, c->dpi) * ds[nd]; l += dash[nd]; }
   const int was_round = c->cap_round;
   if (style == 5) c->cap_round = 1;
   if (style == 4) { c->cap_round = 1; dash[0] = 0; dash[1] = l; nd = 2; }
   tor_edge_t *e;
   int ne, outline_used = 0, solid = nd == 0;
   /* a solid stroke on a context whose caps an earlier stroke of the rule has switched to round (:909-921, never undone): the
    * stroke's pieces and a half disc over the pen's vertices at either end (orc_stroke_edges_round) */
   if (nd == 0 && c->cap_round)
   {
      e = malloc(sizeof(tor_edge_t) * (8 * n + 4 * ORC_PEN_MAX));
      ne = orc_stroke_edges_round(dev, n, width_pt / 2, c->dpi, e);
   }
   else if (nd == 0)
   {
      e = malloc(sizeof(tor_edge_t) * 8 * n);
      /* the polygon itself for a way with a poked-through join, its pieces otherwise (ORC_STROKE_OUTLINE=0 / 1: always the
       * pieces / always the polygon, for measurements) */
      const char *mode = getenv("ORC_STROKE_OUTLINE");
      int outline = mode && mode[0] == '1' ? 1 : mode && mode[0] == '0' ? 0 : orc_stroke_risky(dev, n, width_pt / 2, c->dpi);
      outline_used = outline;
      ne = outline ? orc_stroke_outline_edges(dev, n, width_pt / 2, c->dpi, e) : orc_stroke_edges(dev, n, width_pt / 2, c->dpi, e);
   }
   else
   {
      if (cf) ne = orc_dash_edges_curve(dev, cf->tin, cf->tout, cf->flag, n, width_pt / 2, c->dpi, dash, nd, c->cap_round, NULL);
      else ne = orc_dash_edges(dev, n, width_pt / 2, c->dpi, dash, nd, c->cap_round, NULL);
      e = malloc(sizeof(tor_edge_t) * (ne > 0 ? ne : 1));
      if (ne > 0 && cf) orc_dash_edges_curve(dev, cf->tin, cf->tout, cf->flag, n, width_pt / 2, c->dpi, dash, nd, c->cap_round, e);
      else if (ne > 0) orc_dash_edges(dev, n, width_pt / 2, c->dpi, dash, nd, c->cap_round, e);
   }
   /* a dashed stroke goes through libcairo's other stroker, which adds every dash, join and cap as a polygon of its own: those
    * pieces ARE its edges, and are cut at the page's sides like any; the pieces of a solid stroke are ours */
   c->no_cuts = solid && !outline_used;
   if (ne > 0) orc_layer_edges(c, e, ne, 1, col);
   c->no_cuts = 0;
   free(e);
   /* ORC_NO_STICKY_CAPS=1 (tests: that the stickiness matters in their scenes): the caps go back to what they were */
   { const char *ns = getenv("ORC_NO_STICKY_CAPS"); if (ns && ns[0] == '1') c->cap_round = was_round; }
}

/* render_poly_line with curve=1 (cairo_smr_poly -> cairo_smr_poly_curve, :926-937): border stroke, then fill /
 * open stroke, each over the way's Bezier chain; solid strokes only (a dashed curve takes libcairo's other
 * stroker: not restated, nothing is drawn and the caller is told) */
int orc_curve_unsupported_;
static void draw_way_curve(orc_canvas_t *c, const orc_draw_style_t *st, const double *x_px, const double *y_px, int n, int closed,
                           int has_fill_col, uint32_t fill_col, int has_border_col, uint32_t border_col)
{
   double *xp = malloc(sizeof(double) * 2 * n), *yp = xp + n, start[2];
   double *curves = malloc(sizeof(double) * 6 * n);
   for (int i = 0; i < n; i++) { xp[i] = x_px[i] * 72 / c->dpi; yp[i] = y_px[i] * 72 / c->dpi; }     /* geo2pt */
   int nc;
   if (st->curve == 2)
   {
      /* the way's points ARE the path: move_to p0, then one curve_to per (c1, c2, p) triple, as cairo_smr_wavy records them
       * (src/smrules_cairo.c:752-840: cairo_move_to :800, cairo_curve_to :772); the path is never closed */
      /* a closed way carries one more point, equal to the first: the path stays open (cairo_fill closes it itself, :1009-1013;
       * the border's cairo_stroke does not) */
      if (closed && (n - 2) % 3 != 0) { orc_curve_unsupported_ = 1; free(curves); free(xp); return; }
      nc = ((closed ? n - 1 : n) - 1) / 3;
      start[0] = xp[0]; start[1] = yp[0];
      for (int k = 0; k < nc; k++)
         for (int t = 0; t < 3; t++) { curves[6 * k + 2 * t] = xp[1 + 3 * k + t]; curves[6 * k + 2 * t + 1] = yp[1 + 3 * k + t]; }
   }
   else nc = orc_curve_path(xp, yp, n, closed, st->curve_fact, start, curves);
   int cap = 64 * (nc + 2), m = 0;
   int32_t *pts = NULL;
   tor_edge_t *e = NULL;
   /* one decomposition per cairo_stroke / cairo_fill: the limits differ (the stroker grows the image by its reach) */
   /* a dashed stroke: libcairo's other stroker, which knows no limits for curves; with the points, where the curve_to's end */
   uint8_t *flag = NULL;
   #define FLATTEN_DASHED() do { \
      for (;;) { \
         pts = realloc(pts, sizeof(int32_t) * 6 * cap); flag = realloc(flag, cap); \
         m = orc_curve_flatten_flags(start, curves, nc, c->dpi, NULL, pts, pts + 2 * cap, pts + 4 * cap, flag, cap); \
         if (m >= 0) break; \
         cap *= 4; } } while (0)
   #define FLATTEN(reach_px) do { \
      int32_t lim[4]; double rf = (reach_px); \
      lim[0] = c->img[0] * 256 - orc_fixed_from_double(rf); lim[1] = c->img[1] * 256 - orc_fixed_from_double(rf); \
      lim[2] = c->img[2] * 256 + orc_fixed_from_double(rf); lim[3] = c->img[3] * 256 + orc_fixed_from_double(rf); \
      for (;;) { \
         pts = realloc(pts, sizeof(int32_t) * 6 * cap); \
         m = orc_curve_flatten(start, curves, nc, c->dpi, lim, pts, pts + 2 * cap, pts + 4 * cap, cap); \
         if (m >= 0) break; \
         cap *= 4; } \
      e = realloc(e, sizeof(tor_edge_t) * (8 * (m + 1) + 4 * ORC_PEN_MAX)); } while (0)
   if (st->border_used)
   {
      double bw;
      if (!st->fill_used) bw = mm2wu(st->border_width_mm, c->dpi);
      else if (!closed) bw = mm2wu(2 * st->border_width_mm, c->dpi) + mm2wu(st->fill_width_mm, c->dpi);
      else bw = mm2wu(2 * st->border_width_mm, c->dpi);
      if (st->border_ndash > 0)                      /* (a dash array makes a dashed stroke whatever the style says: stroke_styled) */
      {
         FLATTEN_DASHED();
         curve_faces_t cf = { pts + 2 * cap, pts + 4 * cap, flag };
         uint32_t col = has_border_col ? border_col : st->border_col;
         if (m >= 2)
         {
            stroke_styled(c, pts, m, bw, st->border_style == 4 ? 1 : st->border_style, st->border_width_mm, st->border_dash, st->border_ndash, col, &cf);
            if (st->border_style == 4) stroke_styled(c, pts, m, bw * 2.5, 4, st->border_width_mm, st->border_dash, st->border_ndash, col, &cf);
         }
         goto border_done;
      }
      FLATTEN(orc_stroke_reach_px(bw, c->dpi));
      const char *om = getenv("ORC_STROKE_OUTLINE");
      /* (the rule's caps gone round, :909-921: always the pieces, with a half disc at either end, as for straight ways) */
      const int tan_outline = om && om[0] == '1' && !c->cap_round;
      int ne = m < 2 ? 0 : c->cap_round ? orc_stroke_edges_round_tan(pts, pts + 2 * cap, pts + 4 * cap, m, bw / 2, c->dpi, e) : tan_outline ? orc_stroke_outline_edges_tan(pts, pts + 2 * cap, pts + 4 * cap, m, bw / 2, c->dpi, e) : orc_stroke_edges_tan(pts, pts + 2 * cap, pts + 4 * cap, m, bw / 2, c->dpi, e);
      c->no_cuts = !tan_outline;
      if (ne > 0) orc_layer_edges(c, e, ne, 1, has_border_col ? border_col : st->border_col);
      c->no_cuts = 0;
   }
   border_done:
   if (st->fill_used)
   {
      uint32_t col = has_fill_col ? fill_col : st->fill_col;
      double fw = mm2wu(st->fill_width_mm, c->dpi);
      if (!closed && st->fill_ndash > 0)
      {
         FLATTEN_DASHED();
         curve_faces_t cf = { pts + 2 * cap, pts + 4 * cap, flag };
         if (m >= 2) stroke_styled(c, pts, m, fw, st->fill_style, st->border_width_mm, st->fill_dash, st->fill_ndash, col, &cf);
         goto all_done;
      }
      FLATTEN(closed ? 0.0 : orc_stroke_reach_px(fw, c->dpi));
      const char *om = getenv("ORC_STROKE_OUTLINE");
      const int tan_outline = om && om[0] == '1' && !c->cap_round;
      int ne = m < 2 ? 0 : closed ? orc_fill_edges(pts, m, e) : c->cap_round ? orc_stroke_edges_round_tan(pts, pts + 2 * cap, pts + 4 * cap, m, fw / 2, c->dpi, e) : tan_outline ? orc_stroke_outline_edges_tan(pts, pts + 2 * cap, pts + 4 * cap, m, fw / 2, c->dpi, e) : orc_stroke_edges_tan(pts, pts + 2 * cap, pts + 4 * cap, m, fw / 2, c->dpi, e);
      c->no_cuts = !closed && !tan_outline;
      c->fill_poly = closed;
      if (ne > 0) orc_layer_edges(c, e, ne, closed ? !st->evenodd : 1, col);
      c->fill_poly = 0;
      c->no_cuts = 0;
   }
   all_done:
   #undef FLATTEN
   #undef FLATTEN_DASHED
   free(e); free(pts); free(curves); free(xp); free(flag);
}

/* render_poly_line, src/smrules_cairo.c:953-1016 (no image/curve) */
void orc_draw_way(orc_canvas_t *c, const orc_draw_style_t *st, const double *x_px, const double *y_px, int n, int closed,
                  int has_fill_col, uint32_t fill_col, int has_border_col, uint32_t border_col)
{
   if (n < 2) return;                                   /* :962-966 */
   if (!closed && st->ignore_open) return;              /* :1030-1033 */
   if (st->curve) { draw_way_curve(c, st, x_px, y_px, n, closed, has_fill_col, fill_col, has_border_col, border_col); return; }
   int32_t *dev = malloc(sizeof(int32_t) * 2 * n);
   tor_edge_t *e = malloc(sizeof(tor_edge_t) * 8 * n);
   orc_px_to_dev(x_px, y_px, n, c->dpi, dev);
   for (int i = 0; i < n; i++) { dev[2 * i] += c->shift_x; dev[2 * i + 1] += c->shift_y; }
   if (st->border_used)
   {
      /* cairo_smr_border_width, :880-889 */
      double bw;
      if (!st->fill_used) bw = mm2wu(st->border_width_mm, c->dpi);
      else if (!closed) bw = mm2wu(2 * st->border_width_mm, c->dpi) + mm2wu(st->fill_width_mm, c->dpi);
      else bw = mm2wu(2 * st->border_width_mm, c->dpi);
      uint32_t col = has_border_col ? border_col : st->border_col;
      stroke_styled(c, dev, n, bw, st->border_style == 4 ? 1 : st->border_style, st->border_width_mm, st->border_dash, st->border_ndash, col, NULL);
      if (st->border_style == 4)       /* the pipe's dots, :979-986 */
         stroke_styled(c, dev, n, bw * 2.5, 4, st->border_width_mm, st->border_dash, st->border_ndash, col, NULL);
   }
   if (st->fill_used)
   {
      uint32_t col = has_fill_col ? fill_col : st->fill_col;
      if (!closed)
      {
         /* :1005-1007; the fill's dashes scale with the BORDER width */
         stroke_styled(c, dev, n, mm2wu(st->fill_width_mm, c->dpi), st->fill_style, st->border_width_mm, st->fill_dash, st->fill_ndash, col, NULL);
      }
      else
      {
         int ne = orc_fill_edges(dev, n, e);                                                     /* :1012 */
         c->fill_poly = 1;
         orc_layer_edges(c, e, ne, !st->evenodd, col);
         c->fill_poly = 0;
      }
   }
   free(dev); free(e);
}
