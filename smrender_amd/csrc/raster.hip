/* raster.hip -- draw path of libsmr_hip: device-space vertex setup, stroke expansion, edge
 * setup, per-tile binning/sort and the wave-per-tile Cairo-compatible rasterizer.
 *
 * Replaces (reference file:line): cairo_smr_poly_line src/smrules_cairo.c:845-863,
 * cairo_smr_border_width :880-889, render_poly_line :953-1016, act_draw_fini's group
 * pop+paint :1081-1082 and the libcairo/pixman replay that save_main_image triggers
 * (:388-412).  The coverage arithmetic is the one of libcairo 1.16's tor scan converter
 * (256 x 15 sample grid, sub-row centre sampling, exact full-row path), restated and
 * pinned in oracle/tor.c; this file must stay bit-equal to it.
 */
#include "smr_dev.h"
#include "smr_host.h"
#include <algorithm>
#include <cmath>
#include <string.h>
#include <vector>

/* ------------------------------------------------------------------------------------ */
/* vertex setup: px (double) -> pt -> 24.8 -> replay scale (fixed * fixed(dpi/72) >> 8)   */
/* ------------------------------------------------------------------------------------ */
__device__ __forceinline__ int64_t smr_find_way(const int64_t *__restrict__ way_off, int64_t n_ways, int64_t g)
{
   int64_t lo = 0, hi = n_ways;       /* way_off[lo] <= g < way_off[hi] */
   while (hi - lo > 1)
   {
      int64_t mid = (lo + hi) >> 1;
      if (way_off[mid] <= g) lo = mid; else hi = mid;
   }
   return lo;
}

__global__ void k_dev_points(const double *__restrict__ x, const double *__restrict__ y,
                             const int32_t *__restrict__ ref, int64_t n_refs, int64_t n_nodes, int dpi, int32_t sfix, double far_x, double far_y,
                             const int64_t *__restrict__ way_off, int64_t n_ways,
                             int2 *__restrict__ pts, int32_t *__restrict__ refway, uint32_t *__restrict__ status,
                             const int32_t *__restrict__ node_win, const int2 *__restrict__ win_shift)
{
   int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   int64_t stride = (int64_t) gridDim.x * blockDim.x;
   for (; g < n_refs; g += stride)
   {
      int64_t node = ref ? (int64_t) ref[g] : g;
      /* a ref outside the node arrays: reported (SMR_E_INVAL before anything is drawn), and nothing is read out of bounds */
      if (node < 0 || node >= n_nodes) { atomicOr(status, SMR_ST_BADBATCH); node = 0; }
      /* the edge arithmetic is exact while an edge's 60 * dy (24.8) fits 31 bits and its numerators 63 (smr_orient, smr_edge_cell):
       * 139 000 px of height.  A node more than SMR_FAR_PX beyond the chart (or not a number) is reported instead of wrapping
       * around; nothing out there is ever seen, so a caller clips such ways first (the plugin does: far_clip) */
      if (!(x[node] >= -SMR_FAR_PX && x[node] <= far_x) || !(y[node] >= -SMR_FAR_PX && y[node] <= far_y)) atomicOr(status, SMR_ST_FARAWAY);
      double xp = x[node] * 72 / dpi, yp = y[node] * 72 / dpi;      /* rdata_px_unit(.., U_PT) */
      int32_t fx = (int32_t) rint(xp * 256.0), fy = (int32_t) rint(yp * 256.0);
      int2 p;
      p.x = (int32_t) (((int64_t) fx * sfix) >> 8);
      p.y = (int32_t) (((int64_t) fy * sfix) >> 8);
      /* a window of a replay at an offset (smr_hip_draw_windows): libcairo adds the replay matrix's translation, quantised to
       * 24.8 by itself, to the scaled device point (_cairo_path_fixed_offset_and_scale); strokes are built on the shifted path */
      if (node_win) { const int2 sh = win_shift[node_win[node]]; p.x += sh.x; p.y += sh.y; }
      pts[g] = p;
      refway[g] = (int32_t) smr_find_way(way_off, n_ways, g);
   }
}

enum { SMR_PASS_NORMAL = 0, SMR_PASS_OPEN = 1, SMR_PASS_COMPOUND = 2 };

/* per-way classification: is_closed_poly (src/smcoast.c:115-125), short-way skip
 * (src/smrules_cairo.c:962-966), ignore_open (:1030-1033); decides which of the two
 * primitives of the way exist.  wayflags: bit0 closed, bit1 border prim, bit2 fill polygon,
 * bit3 fill-as-stroke (open way) */
__global__ void k_way_info(const int64_t *__restrict__ way_off, const int32_t *__restrict__ ref, int64_t n_ways, int64_t n_refs,
                           int border_used, int fill_used, int ignore_open, int mode, uint8_t *__restrict__ wayflags,
                           uint32_t *__restrict__ any_open_fill, uint32_t *__restrict__ first_stroked, uint32_t *__restrict__ first_open,
                           uint32_t *__restrict__ status)
{
   int64_t w = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (w >= n_ways) return;
   int64_t a = way_off[w], b = way_off[w + 1];
   /* the way table must start at 0, ascend and end at n_refs: every kernel downstream indexes the refs through it */
   if (a < 0 || b < a || b > n_refs || (w == 0 && a != 0) || (w == n_ways - 1 && b != n_refs))
   {
      atomicOr(status, SMR_ST_BADBATCH);
      wayflags[w] = 0;
      return;
   }
   int64_t n = b - a;
   int closed = 0;
   if (n >= 4)
   {
      int64_t first = ref ? (int64_t) ref[a] : a, last = ref ? (int64_t) ref[b - 1] : b - 1;
      closed = first == last;
   }
   uint8_t f = closed ? 1 : 0;
   if (n >= 2 && !(!closed && ignore_open))
   {
      if (border_used) f |= 2;
      if (fill_used) f |= closed ? 4 : 8;
   }
   /* directional rules (act_draw_main :1039-1052, act_draw_fini :1084-1140): pass 1 draws the open ways only,
    * pass 2 the closed ones as ONE compound fill */
   if (mode == SMR_PASS_OPEN && closed) f = 1;
   if (mode == SMR_PASS_COMPOUND) f = closed ? (f & (1 | 4)) : 0;
   wayflags[w] = f;
   if (f & 8) atomicOr(any_open_fill, 1u);
   if (first_stroked && (f & 2)) atomicMin(first_stroked, (uint32_t) w);
   if (first_open && (f & 8)) atomicMin(first_open, (uint32_t) w);
}

/* fill polygon: edge g = pts[g] -> pts[g+1]; the slot of the last ref closes the polygon
 * (cairo_fill closes sub-paths implicitly) */
__global__ void k_fill_edges(const int2 *__restrict__ pts, const int64_t *__restrict__ way_off,
                             const int32_t *__restrict__ refway, int64_t n_refs,
                             const uint8_t *__restrict__ wayflags, smr_edge_t *__restrict__ edges,
                             const uint8_t *__restrict__ skip)
{
   int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= n_refs) return;
   int64_t w = refway[g];
   smr_edge_t e; e.x1 = e.y1 = e.x2 = e.y2 = 0;
   if ((wayflags[w] & 4) && !(skip && skip[g] == 1))
   {
      int64_t a = way_off[w], b = way_off[w + 1];
      int64_t nx = g + 1;
      if (skip) while (nx < b && skip[nx] == 1) nx++;      /* points of the border's decomposition only (k_curve) */
      int2 p = pts[g], q = pts[nx < b ? nx : a];
      e.x1 = p.x; e.y1 = p.y; e.x2 = q.x; e.y2 = q.y;
   }
   edges[g] = e;
}

/* ------------------------------------------------------------------------------------ */
/* curve=1: the way as a chain of cubic Beziers (cairo_smr_poly_curve, src/smrules_cairo.c:697-749;
 * control_points, src/bspline_ctrl.c:64-105), decomposed like libcairo 1.16 does at replay
 * (cairo-spline.c: de Casteljau halving in 24.8 until both control points are within 0.1 px of
 * the chord; restated and pinned in oracle/curve.c).  ONE LANE per ref slot: slot 0 of a way is
 * the move_to, every other slot one curve_to.  A count pass, a scan, an emit pass; what comes out
 * is a longer poly-line per way plus, per point, the direction of the stroker's face on the
 * segment arriving (tin) and leaving (tout): libcairo's stroker puts the faces of a decomposed
 * curve along its TANGENTS and joins only where two Beziers meet.  Everything downstream (edges,
 * primitives, bins, raster) runs on the decomposed way as on any other.                        */
/* ------------------------------------------------------------------------------------ */
struct curve_args_t
{
   const double *x, *y; const int32_t *ref;
   const int2 *pts; const int64_t *way_off; const int32_t *refway; const uint8_t *wayflags;
   int64_t n_refs, n_ways;
   int dpi; int32_t sfix; double f;
   int border_used, fill_used;
   int knots;                          /* curve=2: the way's refs are p0 and (c1, c2, p) triples, the path cairo_smr_wavy records
                                          (src/smrules_cairo.c:752-840); one more ref equal to the first marks the way as closed
                                          (filled, :1009-1013): the path itself stays open, cairo_fill closes it with a straight line */
   int32_t W256, H256;                 /* the chart in 24.8 */
   int32_t reach[4];                   /* reach of the border stroke (no fill / open / closed way) and of the open ways' fill stroke, 24.8 */
   uint32_t *cnt;                      /* count pass */
   const uint32_t *off;                /* emit pass: first point of every slot */
   int2 *fpts, *tin, *tout; int32_t *frefway; int64_t *fway_off;
   uint8_t *fskip;                     /* 1: the point exists for the border stroke only, 2: the fill arrives here by a line_to (see k_curve) */
   uint8_t *pflag;                     /* dashed strokes (k_dash): 1 = a curve_to ends at the point (or the move_to), 3 = that curve_to was a line_to */
   uint8_t *bskip;                     /* a solid border beside a dashed fill stroke (open ways): fskip's marks for the BORDER, whose stroker has image
                                          limits where the fill stroke's dashing stroker, which the decomposition then follows, has none */
   int border_dashed, fill_dashed;     /* libcairo's dashing stroker knows no image limits for curves: its decomposition is never a line_to for them */
   uint32_t *status;
   /* ill-conditioned control points (see smr_control_points): recorded by the first count pass (fix_n counts them, fix their
    * slot and inputs), computed on the host with its libm, and looked up by slot (sorted) in the passes that follow */
   uint32_t *fix_n; struct smr_curvefix_t *fix; uint32_t fix_cnt;
   /* windows of an atlas (smr_hip_draw_windows): the shift of a node's window, the rows that are a way's image */
   const int32_t *node_win; const int2 *win_shift; const int2 *way_clip; const int4 *way_lim;
};

#define SMR_CURVEFIX_CAP 8192u
struct smr_curvefix_t { int64_t g; double v[8]; };      /* in: gax gay gbx gby lax lay lbx lby; back from the host: c1x c1y c2x c2y */

#define SMR_CURVE_STACK 28

struct smr_knots_t { int2 a, b, c, d; };

__device__ __forceinline__ double smr_spline_err2(const smr_knots_t &k)      /* _cairo_spline_error_squared */
{
   double bdx = (k.b.x - k.a.x) / 256.0, bdy = (k.b.y - k.a.y) / 256.0;
   double cdx = (k.c.x - k.a.x) / 256.0, cdy = (k.c.y - k.a.y) / 256.0;
   if (k.a.x != k.d.x || k.a.y != k.d.y)
   {
      const double dx = (k.d.x - k.a.x) / 256.0, dy = (k.d.y - k.a.y) / 256.0;
      const double v = dx * dx + dy * dy;
      double u = bdx * dx + bdy * dy;
      if (u <= 0) { }
      else if (u >= v) { bdx -= dx; bdy -= dy; }
      else { bdx -= u / v * dx; bdy -= u / v * dy; }
      u = cdx * dx + cdy * dy;
      if (u <= 0) { }
      else if (u >= v) { cdx -= dx; cdy -= dy; }
      else { cdx -= u / v * dx; cdy -= u / v * dy; }
   }
   const double berr = bdx * bdx + bdy * bdy, cerr = cdx * cdx + cdy * cdy;
   return berr > cerr ? berr : cerr;
}

__device__ __forceinline__ int2 smr_lerp_half(int2 a, int2 b) { return make_int2(a.x + ((b.x - a.x) >> 1), a.y + ((b.y - a.y) >> 1)); }

/* src/bspline_ctrl.c:64-105.  Returns true when a control point's direction is the angle of a vector that is all rounding
 * noise: the segment runs straight back along its neighbour (a way that doubles back, A B A, or nodes mirrored exactly), so
 * the helper point falls onto the segment's other end and atan2 sees the last bits of cos and sin.  The reference's result
 * then depends on its libm to the last bit, which the device's does not share: such control points come from the host. */
__device__ __forceinline__ bool smr_control_points(double gax, double gay, double gbx, double gby, double lax, double lay, double lbx, double lby,
                                                   double f, double &p1x, double &p1y, double &p2x, double &p2y)
{
   const double lgt = smr_hypot(gbx - lax, gby - lay);
   const double ag = atan2(gby - gay, gbx - gax);
   const double hax = gbx - lgt * cos(ag), hay = gby - lgt * sin(ag);
   double a = atan2(lay - hay, lax - hax);
   p1x = gbx + lgt * cos(a) * f;
   p1y = gby + lgt * sin(a) * f;
   const double al = atan2(lby - lay, lbx - lax);
   const double hbx = lax + lgt * cos(al), hby = lay + lgt * sin(al);
   a = atan2(hby - gby, hbx - gbx);
   p2x = lax - lgt * cos(a) * f;
   p2y = lay - lgt * sin(a) * f;
   const double v1 = (lax - hax) * (lax - hax) + (lay - hay) * (lay - hay), v2 = (hbx - gbx) * (hbx - gbx) + (hby - gby) * (hby - gby);
   const double lim = 1e-10 * lgt * lgt;
   return lgt > 0 && (v1 < lim || v2 < lim);
}

/* the same on the host, with the libm the reference itself would use (expression by expression src/bspline_ctrl.c:64-105) */
static void smr_control_points_host(const double *v, double f, double *out)
{
   const double gax = v[0], gay = v[1], gbx = v[2], gby = v[3], lax = v[4], lay = v[5], lbx = v[6], lby = v[7];
   const double lgt = hypot(gbx - lax, gby - lay);
   const double ag = atan2(gby - gay, gbx - gax);
   const double hax = gbx - lgt * cos(ag), hay = gby - lgt * sin(ag);
   double a = atan2(lay - hay, lax - hax);
   out[0] = gbx + lgt * cos(a) * f;
   out[1] = gby + lgt * sin(a) * f;
   const double al = atan2(lby - lay, lbx - lax);
   const double hbx = lax + lgt * cos(al), hby = lay + lgt * sin(al);
   a = atan2(hby - gby, hbx - gbx);
   out[2] = lax - lgt * cos(a) * f;
   out[3] = lay - lgt * sin(a) * f;
}

template <bool EMIT>
__global__ void k_curve(curve_args_t A)
{
   const int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= A.n_refs) return;
   const int64_t w = A.refway[g];
   const uint8_t wf = A.wayflags[w];
   const int64_t a0 = A.way_off[w], b0 = A.way_off[w + 1], n = b0 - a0, j = g - a0;
   const bool closed = wf & 1;
   uint32_t cnt = 0;
   const uint32_t base = EMIT ? A.off[g] : 0;
   int2 last = make_int2(0, 0);
   bool fill_line = false, border_line = false;
   auto sink = [&](int2 p, int32_t tdx, int32_t tdy, bool interior, bool is_line = false)
   {
      if (EMIT)
      {
         if (A.pflag) A.pflag[base + cnt] = interior ? 0 : is_line ? 3 : 1;
         if (A.bskip) A.bskip[base + cnt] = !border_line ? 0 : interior ? 1 : 2;
         /* a tangent of zero length would take libcairo's cusp branch (a fan over the pen): the chord stands in */
         if (tdx == 0 && tdy == 0) { tdx = p.x - last.x; tdy = p.y - last.y; }
         const uint32_t q = base + cnt;
         A.fpts[q] = p; A.tin[q] = make_int2(tdx, tdy); A.frefway[q] = (int32_t) w;
         A.fskip[q] = !fill_line ? 0 : interior ? 1 : 2;
         if (interior) A.tout[q] = make_int2(tdx, tdy);
      }
      last = p;
      cnt++;
   };
   if (A.knots && closed && (n - 2) % 3 != 0 && (wf & (2 | 4 | 8)))
   {
      if (j == 0) atomicOr(A.status, SMR_ST_KNOTS_CLOSED);      /* a closed knot way is p0, (c1, c2, p) * m, p0 */
   }
   else if (A.knots && closed && j == n - 1) { }                /* the closing ref only says "closed": cairo_fill closes the path itself */
   else if ((wf & (2 | 4 | 8)) && n >= 2)
   {
      const int64_t npt = closed ? n - 1 : n;
      const int st = closed ? 0 : 1;
      if (j == 0)
      {
         const int2 p = A.pts[a0 + (A.knots ? 0 : (st - 1 + npt) % npt)];      /* (knots: i0..i3 below are unused) */
         last = p;
         sink(p, 0, 0, false);
         if (EMIT) A.fway_off[w] = base;
      }
      else if (A.knots && j % 3 != 0) { }      /* a control point: its curve belongs to the slot of the curve's end point */
      else
      {
         const int64_t i = closed ? j - 1 : j;
         const int64_t i0 = (i + npt - 2) % npt, i1 = (i + npt - 1) % npt, i2 = i % npt, i3 = (i + 1) % npt;
         auto node = [&](int64_t k) { return A.ref ? (int64_t) A.ref[a0 + k] : a0 + k; };
         const int64_t n0 = node(i0), n1 = node(i1), n2 = node(i2), n3 = node(i3);
         /* geo2pt = geo2pxf * 72 / dpi (src/rdata.c:240-245, :307-308) */
         const double gax = A.x[n0] * 72 / A.dpi, gay = A.y[n0] * 72 / A.dpi, gbx = A.x[n1] * 72 / A.dpi, gby = A.y[n1] * 72 / A.dpi;
         const double lax = A.x[n2] * 72 / A.dpi, lay = A.y[n2] * 72 / A.dpi, lbx = A.x[n3] * 72 / A.dpi, lby = A.y[n3] * 72 / A.dpi;
         smr_knots_t K;
         if (A.knots) { K.a = A.pts[a0 + j - 3]; K.b = A.pts[a0 + j - 2]; K.c = A.pts[a0 + j - 1]; K.d = A.pts[a0 + j]; }
         else
         {
            double c1x, c1y, c2x, c2y;
            if (smr_control_points(gax, gay, gbx, gby, lax, lay, lbx, lby, A.f, c1x, c1y, c2x, c2y))
            {
               if (A.fix_cnt)
               {
                  uint32_t lo = 0, hi = A.fix_cnt;              /* the host's values, by slot */
                  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (A.fix[mid].g < g) lo = mid + 1; else hi = mid; }
                  if (lo < A.fix_cnt && A.fix[lo].g == g) { c1x = A.fix[lo].v[0]; c1y = A.fix[lo].v[1]; c2x = A.fix[lo].v[2]; c2y = A.fix[lo].v[3]; }
               }
               else if (!EMIT && A.fix_n)
               {
                  const uint32_t k = atomicAdd(A.fix_n, 1u);
                  if (k < SMR_CURVEFIX_CAP)
                  {
                     smr_curvefix_t fx; fx.g = g;
                     fx.v[0] = gax; fx.v[1] = gay; fx.v[2] = gbx; fx.v[3] = gby; fx.v[4] = lax; fx.v[5] = lay; fx.v[6] = lbx; fx.v[7] = lby;
                     A.fix[k] = fx;
                  }
               }
            }
            if (st)
            {
               if (i == 1) { c1x = gbx; c1y = gby; }
               if (i == npt - 1) { c2x = lax; c2y = lay; }
            }
            auto dev = [&](double v) { return (int32_t) (((int64_t) (int32_t) rint(v * 256.0) * A.sfix) >> 8); };
            const int2 sh = A.node_win ? A.win_shift[A.node_win[n1]] : make_int2(0, 0);      /* as k_dev_points shifts the nodes */
            K.a = A.pts[a0 + i1]; K.d = A.pts[a0 + i2];
            K.b = make_int2(dev(c1x) + sh.x, dev(c1y) + sh.y); K.c = make_int2(dev(c2x) + sh.x, dev(c2y) + sh.y);
         }
         last = K.a;
         bool line = K.a.x == K.b.x && K.a.y == K.b.y && K.c.x == K.d.x && K.c.y == K.d.y;
         /* _cairo_spline_intersects against the image (grown by the stroke's reach): a curve that cannot reach it is a line_to */
         {
            const int32_t r = A.border_used ? A.reach[!A.fill_used ? 0 : closed ? 2 : 1] : closed ? 0 : A.reach[3];
            /* the image: the chart, or for a window of an atlas the image libcairo makes of the source there (way_lim, windows_setup) */
            const int32_t ix0 = A.way_lim ? A.way_lim[w].x : 0, ix1 = A.way_lim ? A.way_lim[w].y : A.W256;
            const int32_t iy0 = A.way_lim ? A.way_lim[w].z : A.way_clip ? A.way_clip[w].x * 256 : 0, iy1 = A.way_lim ? A.way_lim[w].w : A.way_clip ? A.way_clip[w].y * 256 : A.H256;
            const int32_t lx0 = ix0 - r, ly0 = iy0 - r, lx1 = ix1 + r, ly1 = iy1 + r;
            const int2 kk[4] = { K.a, K.b, K.c, K.d };
            bool inside = false;
            int32_t x0 = K.a.x, x1 = K.a.x, y0 = K.a.y, y1 = K.a.y;
#pragma unroll
            for (int t = 0; t < 4; t++)
            {
               if (lx0 <= kk[t].x && kk[t].x <= lx1 && ly0 <= kk[t].y && kk[t].y <= ly1) inside = true;
               x0 = min(x0, kk[t].x); x1 = max(x1, kk[t].x); y0 = min(y0, kk[t].y); y1 = max(y1, kk[t].y);
            }
            const bool fill_unlimited = !closed && A.fill_dashed;
            /* an open way's dashed fill stroke beside a SOLID border: the decomposition follows the fill stroke's stroker (no limits),
             * and the border passes over what its own limits would have made a line_to (bskip) */
            const bool follow_fill = A.border_used && !A.border_dashed && A.fill_used && fill_unlimited;
            const bool unlimited = A.border_used ? (A.border_dashed != 0 || follow_fill) : fill_unlimited;
            const bool outside = !inside && (x1 <= lx0 || x0 >= lx1 || y1 <= ly0 || y0 >= ly1);
            if (!unlimited && outside) line = true;
            border_line = follow_fill && outside && !line;
            /* the fill of a bordered way has limits of its own (the image; grown by the fill stroke's reach for open ways):
             * a curve inside the border's reach but outside those is decomposed for the border and a line_to for the fill,
             * which skips its inner points (k_fill_edges, k_stroke_edges mode 1) */
            if (A.border_used && A.fill_used)
            {
               const int32_t r2 = closed ? 0 : A.reach[3];       /* open ways: the fill is a stroke of the fill width */
               bool in0 = false;
#pragma unroll
               for (int t = 0; t < 4; t++) if (ix0 - r2 <= kk[t].x && kk[t].x <= ix1 + r2 && iy0 - r2 <= kk[t].y && kk[t].y <= iy1 + r2) in0 = true;
               fill_line = !fill_unlimited && !in0 && (x1 <= ix0 - r2 || x0 >= ix1 + r2 || y1 <= iy0 - r2 || y0 >= iy1 + r2);
            }
         }
         int2 is = make_int2(0, 0), fs = make_int2(0, 0);
         if (!line)
         {
            if (K.a.x != K.b.x || K.a.y != K.b.y) is = make_int2(K.b.x - K.a.x, K.b.y - K.a.y);
            else if (K.a.x != K.c.x || K.a.y != K.c.y) is = make_int2(K.c.x - K.a.x, K.c.y - K.a.y);
            else if (K.a.x != K.d.x || K.a.y != K.d.y) is = make_int2(K.d.x - K.a.x, K.d.y - K.a.y);
            else line = true;
         }
         if (!line)
         {
            if (K.c.x != K.d.x || K.c.y != K.d.y) fs = make_int2(K.d.x - K.c.x, K.d.y - K.c.y);
            else if (K.b.x != K.d.x || K.b.y != K.d.y) fs = make_int2(K.d.x - K.b.x, K.d.y - K.b.y);
            else line = true;
         }
         if (line)
         {
            if (K.a.x != K.d.x || K.a.y != K.d.y)
            {
               if (EMIT) A.tout[base - 1] = make_int2(K.d.x - K.a.x, K.d.y - K.a.y);
               sink(K.d, K.d.x - K.a.x, K.d.y - K.a.y, false, true);
            }
         }
         else
         {
            if (EMIT) A.tout[base - 1] = is;
            /* _cairo_spline_decompose_into without the recursion: left halves first, right halves wait on a stack */
            smr_knots_t stk[SMR_CURVE_STACK];
            int sp = 0;
            smr_knots_t cur = K;
            int2 lastpt = K.a;                         /* spline->last_point */
            for (;;)
            {
               if (smr_spline_err2(cur) < 0.1 * 0.1 || sp >= SMR_CURVE_STACK)
               {
                  if (sp >= SMR_CURVE_STACK) atomicOr(A.status, SMR_ST_CURVE_DEPTH);
                  if (lastpt.x != cur.a.x || lastpt.y != cur.a.y)
                  {
                     lastpt = cur.a;
                     sink(cur.a, cur.b.x - cur.a.x, cur.b.y - cur.a.y, true);
                  }
                  if (sp == 0) break;
                  cur = stk[--sp];
                  continue;
               }
               const int2 ab = smr_lerp_half(cur.a, cur.b), bc = smr_lerp_half(cur.b, cur.c), cd = smr_lerp_half(cur.c, cur.d);
               const int2 abbc = smr_lerp_half(ab, bc), bccd = smr_lerp_half(bc, cd), fin = smr_lerp_half(abbc, bccd);
               smr_knots_t s2; s2.a = fin; s2.b = bccd; s2.c = cd; s2.d = cur.d;
               stk[sp++] = s2;
               cur.b = ab; cur.c = abbc; cur.d = fin;
            }
            sink(K.d, fs.x, fs.y, false);
         }
      }
   }
   if (!EMIT) A.cnt[g] = cnt;
}

__global__ void k_curve_finish(int64_t *fway_off, const int64_t *way_off, const uint32_t *off, const uint8_t *wayflags, int64_t n_ways, uint32_t total)
{
   const int64_t w = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (w > n_ways) return;
   /* ways that emit nothing (not drawn) still need their (empty) range */
   fway_off[w] = w == n_ways ? (int64_t) total : (int64_t) off[way_off[w]];
}

/* ---- stroker (cairo-path-stroke-polygon.c semantics, see oracle/smr_oracle.c): butt caps,
 * miter joins limit 10; one quad per segment + one wedge per join, all wound the same way ---- */
struct smr_face_t { int2 ccw, cw; double sx, sy; int32_t ddx, ddy; };

__device__ __forceinline__ void smr_normalize(double &dx, double &dy)
{
   if (dx == 0.0) { dy = dy > 0.0 ? 1.0 : -1.0; }
   else if (dy == 0.0) { dx = dx > 0.0 ? 1.0 : -1.0; }
   else { double mag = smr_hypot(dx, dy); dx = dx / mag; dy = dy / mag; }
}

__device__ __forceinline__ smr_face_t smr_face(int2 p, int32_t ddx, int32_t ddy, double hw_pt, double s)
{
   smr_face_t f;
   double sx = ddx / 256.0, sy = ddy / 256.0;
   smr_normalize(sx, sy);
   f.sx = sx; f.sy = sy; f.ddx = ddx; f.ddy = ddy;
   double ux = sx * (1.0 / s), uy = sy * (1.0 / s);
   smr_normalize(ux, uy);
   double fx = -uy * hw_pt, fy = ux * hw_pt;
   fx *= s; fy *= s;
   int32_t ox = (int32_t) rint(fx * 256.0), oy = (int32_t) rint(fy * 256.0);
   f.ccw = make_int2(p.x + ox, p.y + oy);
   f.cw = make_int2(p.x - ox, p.y - oy);
   return f;
}

__device__ __forceinline__ int smr_emit_poly(const int2 *p, int n, smr_edge_t *out)
{
   int64_t a = 0;
   for (int i = 0; i < n; i++)
   {
      int j = i + 1 == n ? 0 : i + 1;
      a += (int64_t) p[i].x * p[j].y - (int64_t) p[j].x * p[i].y;
   }
   if (a == 0) return 0;
   for (int i = 0; i < n; i++)
   {
      int j = i + 1 == n ? 0 : i + 1;
      int2 s = a > 0 ? p[i] : p[j], e = a > 0 ? p[j] : p[i];
      smr_edge_t ed; ed.x1 = s.x; ed.y1 = s.y; ed.x2 = e.x; ed.y2 = e.y;
      out[i] = ed;
   }
   return n;
}

__device__ __forceinline__ int smr_sgn_cmp(double dx1, double dy1, double dx2, double dy2)
{
   double c = dx1 * dy2 - dx2 * dy1;
   return c > 0 ? 1 : c < 0 ? -1 : 0;
}

__device__ __forceinline__ int64_t smr_turn(int2 a, int2 b, int2 c) { return (int64_t) (b.x - a.x) * (c.y - b.y) - (int64_t) (b.y - a.y) * (c.x - b.x); }

__device__ __forceinline__ bool smr_convex4(int2 a, int2 b, int2 c, int2 d)
{
   const int64_t t0 = smr_turn(a, b, c), t1 = smr_turn(b, c, d), t2 = smr_turn(c, d, a), t3 = smr_turn(d, a, b);
   const bool pos = t0 > 0 || t1 > 0 || t2 > 0 || t3 > 0, neg = t0 < 0 || t1 < 0 || t2 < 0 || t3 < 0;
   return !(pos && neg);
}

/* convex hull of the corners of a quad that may cross itself; same decisions as hull4 of oracle/smr_oracle.c */
__device__ __forceinline__ int smr_hull4(const int2 *q, int2 *h)
{
   if (smr_convex4(q[0], q[1], q[2], q[3])) { h[0] = q[0]; h[1] = q[1]; h[2] = q[2]; h[3] = q[3]; return 4; }
   if (smr_convex4(q[0], q[1], q[3], q[2])) { h[0] = q[0]; h[1] = q[1]; h[2] = q[3]; h[3] = q[2]; return 4; }
   if (smr_convex4(q[0], q[2], q[1], q[3])) { h[0] = q[0]; h[1] = q[2]; h[2] = q[1]; h[3] = q[3]; return 4; }
#pragma unroll
   for (int k = 0; k < 4; k++)
   {
      const int2 t0 = q[k == 0 ? 1 : 0], t1 = q[k <= 1 ? 2 : 1], t2 = q[k <= 2 ? 3 : 2];
      const int64_t a = smr_turn(t0, t1, q[k]), b = smr_turn(t1, t2, q[k]), c = smr_turn(t2, t0, q[k]);
      if ((a >= 0 && b >= 0 && c >= 0) || (a <= 0 && b <= 0 && c <= 0)) { h[0] = t0; h[1] = t1; h[2] = t2; h[3] = t2; return 3; }
   }
   h[0] = q[0]; h[1] = q[1]; h[2] = q[2]; h[3] = q[3];
   return 4;
}

/* join of two faces at one vertex (outer_join / inner_join of cairo-path-stroke-polygon.c, restated in
 * oracle/smr_oracle.c stroke_join): kind 0 = the faces coincide, 1 = miter inside the limit (m = miter
 * point), 2 = bevel; clockwise = the cw side is the outer one */
struct smr_join_t { int kind; bool clockwise; int2 m; };

__device__ __forceinline__ smr_join_t smr_join(const smr_face_t &in, const smr_face_t &out, int2 v)
{
   smr_join_t j; j.kind = 0; j.clockwise = false; j.m = v;
   if (in.cw.x == out.cw.x && in.cw.y == out.cw.y && in.ccw.x == out.ccw.x && in.ccw.y == out.ccw.y) return j;
   const int64_t l = (int64_t) in.ddy * out.ddx, r = (int64_t) out.ddy * in.ddx;
   j.clockwise = l != r ? l < r : (in.ddx > 0 || (in.ddx == 0 && in.ddy > 0));
   const int2 inpt = j.clockwise ? in.cw : in.ccw, outpt = j.clockwise ? out.cw : out.ccw;
   const double in_dot_out = in.sx * out.sx + in.sy * out.sy;
   j.kind = 2;
   if (2 <= 100.0 * (1 + in_dot_out))
   {
      double x1 = inpt.x / 256.0, y1 = inpt.y / 256.0, dx1 = in.sx, dy1 = in.sy;
      double x2 = outpt.x / 256.0, y2 = outpt.y / 256.0, dx2 = out.sx, dy2 = out.sy;
      double my = (((x2 - x1) * dy1 * dy2 - y2 * dx2 * dy1 + y1 * dx1 * dy2) / (dx1 * dy2 - dx2 * dy1));
      double mx;
      if (fabs(dy1) >= fabs(dy2)) mx = (my - y1) * dx1 / dy1 + x1;
      else mx = (my - y2) * dx2 / dy2 + x2;
      double ix = v.x / 256.0, iy = v.y / 256.0;
      double fdx1 = x1 - ix, fdy1 = y1 - iy, fdx2 = x2 - ix, fdy2 = y2 - iy, mdx = mx - ix, mdy = my - iy;
      if (smr_sgn_cmp(fdx1, fdy1, mdx, mdy) != smr_sgn_cmp(fdx2, fdy2, mdx, mdy))
      {
         j.kind = 1;
         j.m = make_int2((int32_t) rint(mx * 256.0), (int32_t) rint(my * 256.0));
      }
   }
   return j;
}

/* mode 0: border strokes, widths per cairo_smr_border_width; mode 1: fill of open ways drawn
 * as a stroke of the fill width (src/smrules_cairo.c:1003-1008).
 * libcairo's stroker runs an outer contour from miter point to miter point and an inner contour through the
 * incoming face corner, the vertex and the outgoing face corner.  That polygon is cut into closed convex
 * pieces whose outer edges are exactly its edges and whose other edges join vertices it has too (so the scan
 * converter decides every row as it does for libcairo's polygon):
 *   slots 4..7  body of the segment: quad over its four face corners, an outer corner replaced by the miter point
 *   slots 0..3  join at the segment's start: miter -> triangle (vertex, incoming inner, outgoing inner corner),
 *               bevel -> quad (vertex, incoming outer, outgoing outer, incoming inner corner)                  */
/* TAN: the way is a decomposed curve; the face at a point looks along tin (arriving segment) / tout (leaving
 * segment) instead of along the chords, and consecutive points are taken as they come (oracle: orc_stroke_edges_tan) */
/* Is the union of a way's pieces its stroke polygon?  Not where the inner corner of a face pokes through a neighbouring
 * segment shorter than the stroke's half width: the loop the inner contour makes there is wound against the rest and cancels
 * coverage (oracle/smr_oracle.c: orc_stroke_risky, the same rule step by step).  The lane of a segment SHORTER than the half
 * width (plus the corners' rounding: an integer test that is part of the rule) looks at the joins at its two ends; a way with
 * a poked-through join gets bit 16 (border) / 32 (fill stroke of an open way) and is then stroked as the polygon itself */
__device__ __forceinline__ bool smr_inside4(int2 a, int2 b, int2 c, int2 d, int2 q)
{
   const int64_t t0 = smr_turn(a, b, q), t1 = smr_turn(b, c, q), t2 = smr_turn(c, d, q), t3 = smr_turn(d, a, q);
   return (t0 >= 0 && t1 >= 0 && t2 >= 0 && t3 >= 0) || (t0 <= 0 && t1 <= 0 && t2 <= 0 && t3 <= 0);
}

__device__ __forceinline__ bool smr_join_poked(int2 q0, int2 p1, int2 p2, double hw, double s)
{
   const int32_t adx = p1.x - q0.x, ady = p1.y - q0.y, bdx = p2.x - p1.x, bdy = p2.y - p1.y;
   const smr_face_t a1 = smr_face(p1, adx, ady, hw, s), b0 = smr_face(p1, bdx, bdy, hw, s);
   const smr_join_t j = smr_join(a1, b0, p1);
   if (j.kind == 0) return false;
   const int2 in_inner = j.clockwise ? a1.ccw : a1.cw, out_inner = j.clockwise ? b0.ccw : b0.cw;
   const int2 a0c = make_int2(a1.ccw.x - adx, a1.ccw.y - ady), a0w = make_int2(a1.cw.x - adx, a1.cw.y - ady);
   const int2 b1c = make_int2(b0.ccw.x + bdx, b0.ccw.y + bdy), b1w = make_int2(b0.cw.x + bdx, b0.cw.y + bdy);
   return !smr_inside4(b0.ccw, b1c, b1w, b0.cw, in_inner) || !smr_inside4(a0c, a1.ccw, a1.cw, a0w, out_inner);
}

__global__ void k_stroke_risk(const int2 *__restrict__ pts, const int64_t *__restrict__ way_off, const int32_t *__restrict__ refway, int64_t n_refs,
                              uint8_t *wayflags, int mode, int dpi, double hw_nofill_pt, double hw_open_pt, double hw_closed_pt, double hw_fill_pt,
                              int fill_used, int force)
{
   const int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= n_refs) return;
   const int64_t w = refway[g];
   const uint8_t wf = wayflags[w];
   if (!(mode == 0 ? (wf & 2) : (wf & 8))) return;
   const int64_t a = way_off[w], b = way_off[w + 1];
   if (force) { if (g == a) atomicOr(reinterpret_cast<uint32_t *>(wayflags) + (w >> 2), (mode == 0 ? 16u : 32u) << (8 * (w & 3))); return; }
   if (g + 1 >= b) return;
   const int2 p1 = pts[g], p2 = pts[g + 1];
   const int32_t ddx = p2.x - p1.x, ddy = p2.y - p1.y;
   if (ddx == 0 && ddy == 0) return;
   const double hw = mode == 1 ? hw_fill_pt : (!fill_used ? hw_nofill_pt : ((wf & 1) ? hw_closed_pt : hw_open_pt));
   const double s = (double) dpi / 72;
   const int64_t thr = (int64_t) ceil(hw * s * 256.0) + 4;
   if ((int64_t) ddx * ddx + (int64_t) ddy * ddy >= thr * thr) return;
   bool poked = false;
   for (int64_t k = g - 1; k >= a; k--)
   {
      const int2 q = pts[k];
      if (q.x != p1.x || q.y != p1.y) { poked = smr_join_poked(q, p1, p2, hw, s); break; }
   }
   for (int64_t k = g + 2; k < b && !poked; k++)
   {
      const int2 q = pts[k];
      if (q.x != p2.x || q.y != p2.y) { poked = smr_join_poked(p1, p2, q, hw, s); break; }
   }
   /* the way's flag byte, set through its aligned word (other bytes: other ways, OR-ing 0 into them) */
   if (poked) atomicOr(reinterpret_cast<uint32_t *>(wayflags) + (w >> 2), (mode == 0 ? 16u : 32u) << (8 * (w & 3)));
}

template <bool TAN>
__global__ void k_stroke_edges(const int2 *__restrict__ pts, const int64_t *__restrict__ way_off,
                               const int32_t *__restrict__ refway, int64_t n_refs,
                               const uint8_t *__restrict__ wayflags, int mode, int dpi,
                               double hw_nofill_pt, double hw_open_pt, double hw_closed_pt, double hw_fill_pt,
                               int fill_used, smr_edge_t *__restrict__ edges,
                               const int2 *__restrict__ tin, const int2 *__restrict__ tout, const uint8_t *__restrict__ skip)
{
   int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= n_refs) return;
   int64_t w = refway[g];
   smr_edge_t loc[SMR_SLOTS];
   smr_edge_t z; z.x1 = z.y1 = z.x2 = z.y2 = 0;
#pragma unroll
   for (int i = 0; i < SMR_SLOTS; i++) loc[i] = z;
   uint8_t wf = wayflags[w];
   int64_t a = way_off[w], b = way_off[w + 1];
   bool on = mode == 0 ? (wf & 2) != 0 : (wf & 8) != 0;
   /* TAN, fill stroke of a bordered open way: points that only the border's decomposition has are passed over, and the
    * segment across them is the line_to libcairo's stroker draws there, faces along the chord (k_curve) */
   const bool sk = TAN && skip != nullptr;      /* (mode 0: bskip, a solid border beside a dashed fill stroke) */
   if (sk && skip[g] == 1) on = false;
   if (on && g + 1 < b)
   {
      int64_t nx = g + 1, pv = g - 1;
      if (sk) { while (nx + 1 < b && skip[nx] == 1) nx++; while (pv > a && skip[pv] == 1) pv--; }
      const bool chord_out = sk && skip[nx] == 2, chord_in = sk && skip[g] == 2;
      int2 p1 = pts[g], p2 = pts[nx];
      if (TAN || p1.x != p2.x || p1.y != p2.y)
      {
         double hw = mode == 1 ? hw_fill_pt : (!fill_used ? hw_nofill_pt : ((wf & 1) ? hw_closed_pt : hw_open_pt));
         double s = (double) dpi / 72;
         int32_t ddx = p2.x - p1.x, ddy = p2.y - p1.y;
         if (TAN && !chord_out) { const int2 t = tout[g]; ddx = t.x; ddy = t.y; }
         smr_face_t start = smr_face(p1, ddx, ddy, hw, s);
         int2 c_ccw0 = start.ccw, c_cw0 = start.cw;
         int2 c_ccw1 = make_int2(start.ccw.x + ddx, start.ccw.y + ddy), c_cw1 = make_int2(start.cw.x + ddx, start.cw.y + ddy);
         smr_face_t endf = start;
         if (TAN)
         {
            const int2 t = !chord_out ? tin[nx] : make_int2(ddx, ddy);
            endf = smr_face(p2, t.x, t.y, hw, s); c_ccw1 = endf.ccw; c_cw1 = endf.cw;
         }
         /* previous non-degenerate segment of the same way ends at p1 */
         int64_t k = g - 1;
         int2 q0 = p1;
         bool have = false;
         while (!TAN && k >= a)
         {
            q0 = pts[k];
            if (q0.x != p1.x || q0.y != p1.y) { have = true; break; }
            k--;
         }
         if (TAN) have = g > a;
         /* a way with a poked-through join (k_stroke_risk): the stroker's polygon itself, edge by edge, filled non-zero --
          * the contour forward on the cw side, backward on the ccw side, an inner join through the vertex, a butt cap at
          * either end of the way (oracle: orc_stroke_outline_edges) */
         const bool outline = (wf & (mode == 0 ? 16 : 32)) != 0;      /* (a decomposed curve: only when every stroke is asked for as a polygon) */
         int no = 0;
         /* (straight to memory: a register array indexed by a running count would live in scratch) */
         smr_edge_t *const oline = edges + (size_t) SMR_SLOTS * g;
         auto edge = [&](int2 u, int2 v) { smr_edge_t ed; ed.x1 = u.x; ed.y1 = u.y; ed.x2 = v.x; ed.y2 = v.y; oline[no++] = ed; };
         if (have)
         {
            int32_t pdx = p1.x - q0.x, pdy = p1.y - q0.y;
            if (TAN) { const int2 t = !chord_in ? tin[g] : make_int2(p1.x - pts[pv].x, p1.y - pts[pv].y); pdx = t.x; pdy = t.y; }
            smr_face_t cur = smr_face(p1, pdx, pdy, hw, s);     /* face of the previous segment at its END point */
            smr_join_t j = smr_join(cur, start, p1);
            int2 in_inner = j.clockwise ? cur.ccw : cur.cw, out_inner = j.clockwise ? start.ccw : start.cw;
            int2 in_outer = j.clockwise ? cur.cw : cur.ccw, out_outer = j.clockwise ? start.cw : start.ccw;
            if (j.kind == 1) { if (j.clockwise) c_cw0 = j.m; else c_ccw0 = j.m; }
            if (outline)
            {
               if (j.kind != 0)
               {
                  if (j.clockwise)
                  {
                     if (j.kind == 2) edge(cur.cw, start.cw);
                     edge(start.ccw, p1); edge(p1, cur.ccw);
                  }
                  else
                  {
                     if (j.kind == 2) edge(start.ccw, cur.ccw);
                     edge(cur.cw, p1); edge(p1, start.cw);
                  }
               }
            }
            else if (j.kind == 1)
            {
               int2 q[3] = { p1, in_inner, out_inner };
               smr_emit_poly(q, 3, loc);
            }
            else if (j.kind == 2)
            {
               int2 q[4] = { p1, in_outer, out_outer, in_inner };
               smr_emit_poly(q, 4, loc);
            }
         }
         else if (outline) edge(start.ccw, start.cw);
         /* next non-degenerate segment of the same way starts at p2 */
         k = g + 2;
         have = false;
         while (!TAN && k < b)
         {
            q0 = pts[k];
            if (q0.x != p2.x || q0.y != p2.y) { have = true; break; }
            k++;
         }
         int64_t nn = nx + 1;
         if (sk) while (nn + 1 < b && skip[nn] == 1) nn++;
         if (TAN) have = nx + 1 < b;
         if (have)
         {
            int32_t ndx = q0.x - p2.x, ndy = q0.y - p2.y;
            if (TAN) { const int2 t = !(sk && skip[nn] == 2) ? tout[nx] : make_int2(pts[nn].x - p2.x, pts[nn].y - p2.y); ndx = t.x; ndy = t.y; }
            smr_face_t end = TAN ? endf : smr_face(p2, ddx, ddy, hw, s), nstart = smr_face(p2, ndx, ndy, hw, s);
            smr_join_t j = smr_join(end, nstart, p2);
            if (j.kind == 1) { if (j.clockwise) c_cw1 = j.m; else c_ccw1 = j.m; }
         }
         else if (outline) edge(c_cw1, c_ccw1);
         int2 q[4] = { c_ccw0, c_ccw1, c_cw1, c_cw0 };
         if (outline)
         {
            edge(c_cw0, c_cw1); edge(c_ccw1, c_ccw0);
            for (; no < SMR_SLOTS; no++) oline[no] = z;
            return;
         }
         else if (TAN)
         {
            /* a curve that bends within half the line width makes this quad cross itself: the piece has to be convex
             * (one interval per sub-row), so the hull of the four corners stands in (oracle: hull4) */
            int2 h[4];
            const int nh = smr_hull4(q, h);
            smr_emit_poly(h, nh, loc + 4);
         }
         else smr_emit_poly(q, 4, loc + 4);
      }
   }
   smr_edge_t *out = edges + (size_t) SMR_SLOTS * g;
#pragma unroll
   for (int i = 0; i < SMR_SLOTS; i++) out[i] = loc[i];
}

/* ------------------------------------------------------------------------------------ */
/* dashed strokes: libcairo's cairo-path-stroke.c dasher (restated in oracle/smr_oracle.c,  */
/* orc_dash_edges).  The dash state runs along the whole way, so ONE LANE walks one way;     */
/* a first pass counts the pieces (4 edge slots each), a scan places them, a second pass      */
/* writes them: quad per "on" stretch, wedge per join of two stretches across a vertex,       */
/* half disc per round cap (fan of quads over the pen's vertices).                            */
/* ------------------------------------------------------------------------------------ */
struct smr_dface_t { int2 ccw, cw, point; double ux, uy; int32_t ddx, ddy; };

__device__ __forceinline__ int smr_slope_cmp(int32_t adx, int32_t ady, int32_t bdx, int32_t bdy)     /* _cairo_slope_compare */
{
   const int64_t l = (int64_t) ady * bdx, r = (int64_t) bdy * adx;
   if (l != r) return l < r ? -1 : 1;
   const bool az = adx == 0 && ady == 0, bz = bdx == 0 && bdy == 0;
   if (az && bz) return 0;
   if (az) return 1;
   if (bz) return -1;
   if ((adx ^ bdx) < 0 || (ady ^ bdy) < 0) return (adx > 0 || (adx == 0 && ady > 0)) ? -1 : 1;
   return 0;
}

/* The pieces of segment g -> g + 1 of a straight way's solid stroke, as k_stroke_edges<false> cuts them (slots 0..3: the join at
 * the segment's first point, 4..7: the segment's quad with its mitered corners); never the stroker's polygon itself.  loc comes
 * zeroed; false for a degenerate segment.  For k_dash's solid strokes with round caps (oracle: orc_stroke_edges + the caps). */
__device__ bool smr_stroke_ref_pieces(const int2 *__restrict__ pts, int64_t a, int64_t b, int64_t g, double hw, double s, smr_edge_t *loc,
                                      smr_face_t &start_face, int32_t &ddx_out, int32_t &ddy_out)
{
   const int2 p1 = pts[g], p2 = pts[g + 1];
   if (p1.x == p2.x && p1.y == p2.y) return false;
   const int32_t ddx = p2.x - p1.x, ddy = p2.y - p1.y;
   const smr_face_t start = smr_face(p1, ddx, ddy, hw, s);
   int2 c_ccw0 = start.ccw, c_cw0 = start.cw;
   int2 c_ccw1 = make_int2(start.ccw.x + ddx, start.ccw.y + ddy), c_cw1 = make_int2(start.cw.x + ddx, start.cw.y + ddy);
   int64_t k = g - 1;
   int2 q0 = p1;
   bool have = false;
   while (k >= a)
   {
      q0 = pts[k];
      if (q0.x != p1.x || q0.y != p1.y) { have = true; break; }
      k--;
   }
   if (have)
   {
      const smr_face_t cur = smr_face(p1, p1.x - q0.x, p1.y - q0.y, hw, s);
      const smr_join_t j = smr_join(cur, start, p1);
      const int2 in_inner = j.clockwise ? cur.ccw : cur.cw, out_inner = j.clockwise ? start.ccw : start.cw;
      const int2 in_outer = j.clockwise ? cur.cw : cur.ccw, out_outer = j.clockwise ? start.cw : start.ccw;
      if (j.kind == 1)
      {
         if (j.clockwise) c_cw0 = j.m; else c_ccw0 = j.m;
         int2 q[3] = { p1, in_inner, out_inner };
         smr_emit_poly(q, 3, loc);
      }
      else if (j.kind == 2)
      {
         int2 q[4] = { p1, in_outer, out_outer, in_inner };
         smr_emit_poly(q, 4, loc);
      }
   }
   k = g + 2;
   have = false;
   while (k < b)
   {
      q0 = pts[k];
      if (q0.x != p2.x || q0.y != p2.y) { have = true; break; }
      k++;
   }
   if (have)
   {
      const smr_face_t end = smr_face(p2, ddx, ddy, hw, s), nstart = smr_face(p2, q0.x - p2.x, q0.y - p2.y, hw, s);
      const smr_join_t j = smr_join(end, nstart, p2);
      if (j.kind == 1) { if (j.clockwise) c_cw1 = j.m; else c_ccw1 = j.m; }
   }
   int2 q[4] = { c_ccw0, c_ccw1, c_cw1, c_cw0 };
   smr_emit_poly(q, 4, loc + 4);
   start_face = start; ddx_out = ddx; ddy_out = ddy;
   return true;
}

/* The same for segment g of a decomposed curve, as k_stroke_edges<true> cuts its pieces: faces along the tangents (along the chord
 * where skip marks a line_to), the points skip passes over left out, the quad's hull.  The faces at the segment's two ends come back
 * for the caps (oracle: orc_stroke_edges_round_tan). */
__device__ bool smr_stroke_ref_pieces_tan(const int2 *__restrict__ pts, const int2 *__restrict__ tin, const int2 *__restrict__ tout,
                                          const uint8_t *__restrict__ skip, int64_t a, int64_t b, int64_t g, double hw, double s, smr_edge_t *loc,
                                          smr_face_t &start_face, int2 &start_dir, smr_face_t &end_face, int2 &end_dir, int2 &end_pt)
{
   const bool sk = skip != nullptr;
   if ((sk && skip[g] == 1) || g + 1 >= b) return false;
   int64_t nx = g + 1, pv = g - 1;
   if (sk) { while (nx + 1 < b && skip[nx] == 1) nx++; while (pv > a && skip[pv] == 1) pv--; }
   const bool chord_out = sk && skip[nx] == 2, chord_in = sk && skip[g] == 2;
   const int2 p1 = pts[g], p2 = pts[nx];
   int32_t ddx = p2.x - p1.x, ddy = p2.y - p1.y;
   if (!chord_out) { const int2 t = tout[g]; ddx = t.x; ddy = t.y; }
   const smr_face_t start = smr_face(p1, ddx, ddy, hw, s);
   int2 c_ccw0 = start.ccw, c_cw0 = start.cw;
   const int2 te = !chord_out ? tin[nx] : make_int2(ddx, ddy);
   const smr_face_t endf = smr_face(p2, te.x, te.y, hw, s);
   int2 c_ccw1 = endf.ccw, c_cw1 = endf.cw;
   if (g > a)
   {
      const int2 t = !chord_in ? tin[g] : make_int2(p1.x - pts[pv].x, p1.y - pts[pv].y);
      const smr_face_t cur = smr_face(p1, t.x, t.y, hw, s);
      const smr_join_t j = smr_join(cur, start, p1);
      const int2 in_inner = j.clockwise ? cur.ccw : cur.cw, out_inner = j.clockwise ? start.ccw : start.cw;
      const int2 in_outer = j.clockwise ? cur.cw : cur.ccw, out_outer = j.clockwise ? start.cw : start.ccw;
      if (j.kind == 1)
      {
         if (j.clockwise) c_cw0 = j.m; else c_ccw0 = j.m;
         int2 q[3] = { p1, in_inner, out_inner };
         smr_emit_poly(q, 3, loc);
      }
      else if (j.kind == 2)
      {
         int2 q[4] = { p1, in_outer, out_outer, in_inner };
         smr_emit_poly(q, 4, loc);
      }
   }
   if (nx + 1 < b)
   {
      int64_t nn = nx + 1;
      if (sk) while (nn + 1 < b && skip[nn] == 1) nn++;
      const int2 t = !(sk && skip[nn] == 2) ? tout[nx] : make_int2(pts[nn].x - p2.x, pts[nn].y - p2.y);
      const smr_face_t nstart = smr_face(p2, t.x, t.y, hw, s);
      const smr_join_t j = smr_join(endf, nstart, p2);
      if (j.kind == 1) { if (j.clockwise) c_cw1 = j.m; else c_ccw1 = j.m; }
   }
   int2 q[4] = { c_ccw0, c_ccw1, c_cw1, c_cw0 }, h[4];
   const int nh = smr_hull4(q, h);
   smr_emit_poly(h, nh, loc + 4);
   start_face = start; start_dir = make_int2(ddx, ddy); end_face = endf; end_dir = te; end_pt = p2;
   return true;
}

struct dash_args_t
{
   const int2 *pts; const int64_t *way_off; const uint8_t *wayflags; int64_t n_ways;
   int mode;                       /* 0 border, 1 fill of open ways drawn as a stroke, 2 the pipe's dots */
   int dpi, fill_used;
   double hw_nofill, hw_open, hw_closed, hw_fill;       /* half widths in pt (the dots' already scaled) */
   double dash[4]; int ndash;                           /* pt */
   int cap_mode;                   /* 0 butt, 1 round, 2 round except on the rule's first stroked way (pipe), 3 round on the ways after
                                      *first_way (the first open way, whose fill stroke of style pipe / rounddot switches the rule's caps) */
   int round_only;                 /* ways whose caps are butt have no pieces here: their solid stroke is k_stroke_edges' */
   int solid;                      /* no dashes: the pieces of the solid stroke (smr_stroke_ref_pieces) and a round cap at either end */
   const uint32_t *first_way;
   const int2 *pen; int pen_off[3], pen_n[3];           /* pens of the nofill / open / closed width (fill stroke: [0]) */
   uint32_t *piece_cnt;            /* count pass */
   const uint32_t *piece_off;      /* emit pass */
   smr_edge_t *edges;
   /* a decomposed curve (k_curve): the slopes at the points and where the curve_to's end; NULL for straight ways */
   const int2 *tin, *tout; const uint8_t *pflag;
   const uint8_t *skip;            /* solid: the points of the one decomposition this stroke passes over (k_curve's fskip / bskip) */
};

struct smr_dasher_t
{
   double s, hw; bool round;
   const int2 *pen; int npen;
   smr_edge_t *out; uint32_t cnt;

   __device__ __forceinline__ void piece(const int2 *q, int n)
   {
      if (out)
      {
         smr_edge_t tmp[4];
         smr_edge_t z; z.x1 = z.y1 = z.x2 = z.y2 = 0;
         tmp[0] = tmp[1] = tmp[2] = tmp[3] = z;
         smr_emit_poly(q, n, tmp);
         smr_edge_t *o = out + 4 * (size_t) cnt;
         o[0] = tmp[0]; o[1] = tmp[1]; o[2] = tmp[2]; o[3] = tmp[3];
      }
      cnt++;
   }
   /* four slots as they come (a piece smr_emit_poly has wound already); an empty group is no piece */
   __device__ __forceinline__ void raw4(const smr_edge_t *e)
   {
      bool any = false;
      for (int i = 0; i < 4; i++) any = any || e[i].x1 != 0 || e[i].y1 != 0 || e[i].x2 != 0 || e[i].y2 != 0;
      if (!any) return;
      if (out) { smr_edge_t *o = out + 4 * (size_t) cnt; o[0] = e[0]; o[1] = e[1]; o[2] = e[2]; o[3] = e[3]; }
      cnt++;
   }
   __device__ __forceinline__ int2 pv(int i) const { return pen[i]; }
   __device__ __forceinline__ int2 pcw(int i) const { int2 a = pen[i], b = pen[i == 0 ? npen - 1 : i - 1]; return make_int2(a.x - b.x, a.y - b.y); }
   __device__ __forceinline__ int2 pccw(int i) const { int2 a = pen[i], b = pen[i + 1 == npen ? 0 : i + 1]; return make_int2(b.x - a.x, b.y - a.y); }
   __device__ int active_ccw(int32_t dx, int32_t dy) const           /* _cairo_pen_find_active_ccw_vertex_index */
   {
      int i;
      for (i = npen - 1; i >= 0; i--)
      {
         const int2 c = pccw(i), w = pcw(i);
         if (smr_slope_cmp(c.x, c.y, -dx, -dy) >= 0 && smr_slope_cmp(w.x, w.y, -dx, -dy) < 0) break;
      }
      return i < 0 ? npen - 1 : i;
   }
   /* which pen vertices the stroker passes between the directions (dx, dy) and (ox, oy) on the side of the ccw corners
    * (_tessellate_fan with _cairo_pen_find_active_ccw_vertices; oracle: d_fan_points): start, (exclusive) stop, stepping down */
   __device__ bool fan_range2(int32_t dx, int32_t dy, int32_t ox, int32_t oy, int &start, int &stop) const
   {
      const int n = npen;
      if (n <= 0) return false;
      start = active_ccw(dx, dy);
      { const int2 c = pccw(start); if (smr_slope_cmp(c.x, c.y, dx, dy) < 0) start = start == 0 ? n - 1 : start - 1; }
      stop = active_ccw(ox, oy);
      { const int2 w = pcw(stop);
        if (smr_slope_cmp(w.x, w.y, ox, oy) > 0)
        {
           stop = stop + 1 == n ? 0 : stop + 1;
           const int2 c = pccw(stop);
           if (smr_slope_cmp(c.x, c.y, dx, dy) < 0) return false;
        } }
      int npoints = start - stop;
      stop = stop == 0 ? n - 1 : stop - 1;
      if (npoints < 0) npoints += n;
      return npoints > 1;
   }
   /* a round cap on a face of direction (dx, dy): the half turn to (-dx, -dy) */
   __device__ bool fan_range(int32_t dx, int32_t dy, int &start, int &stop) const { return fan_range2(dx, dy, -dx, -dy, start, stop); }
   /* a ROUND join (between the segments of a decomposed curve): the fan over the pen's vertices on the outer side, from the pivot;
    * with no vertex in between it is the bevel's triangle (oracle: d_join_round) */
   __device__ void join_round(const smr_dface_t &in, const smr_dface_t &o)
   {
      if (in.cw.x == o.cw.x && in.cw.y == o.cw.y && in.ccw.x == o.ccw.x && in.ccw.y == o.ccw.y) return;
      const bool clockwise = smr_slope_cmp(o.cw.x - o.point.x, o.cw.y - o.point.y, in.cw.x - in.point.x, in.cw.y - in.point.y) < 0;
      int start = 0, stop = 0;
      bool have;
      int2 from, to;
      if (clockwise) { have = fan_range2(in.ddx, in.ddy, o.ddx, o.ddy, start, stop); from = in.ccw; to = o.ccw; }
      else { have = fan_range2(-o.ddx, -o.ddy, -in.ddx, -in.ddy, start, stop); from = o.cw; to = in.cw; }   /* the reversed path's ccw side */
      const int n = npen;
      int m = 3;                                  /* the pivot, from, to */
      if (have) for (int i = start; i != stop; i = i == 0 ? n - 1 : i - 1) m++;
      const int2 mid = in.point;
      auto poly = [&](int j) -> int2
      {
         if (j == 0) return mid;
         if (j == 1) return from;
         if (j == m - 1) return to;
         int i = start - (j - 2);
         if (i < 0) i += n;
         const int2 v = pen[i];
         return make_int2(mid.x + v.x, mid.y + v.y);
      };
      for (int k = 1; k + 1 < m; k += 2)
      {
         int2 q[4] = { mid, poly(k), poly(k + 1), poly(k + 2 < m ? k + 2 : k + 1) };
         piece(q, k + 2 < m ? 4 : 3);
      }
   }
   /* half disc ccw -> pen vertices -> cw, cut into a fan of quads from the first point */
   __device__ void round_cap(int2 ccw, int2 cw, int2 mid, int start, int stop)
   {
      const int n = npen;
      int m = 2;
      for (int i = start; i != stop; i = i == 0 ? n - 1 : i - 1) m++;
      auto poly = [&](int j) -> int2
      {
         if (j == 0) return ccw;
         if (j == m - 1) return cw;
         int i = start - (j - 1);
         if (i < 0) i += n;
         const int2 v = pen[i];
         return make_int2(mid.x + v.x, mid.y + v.y);
      };
      for (int k = 1; k + 1 < m; k += 2)
      {
         int2 q[4] = { ccw, poly(k), poly(k + 1), poly(k + 2 < m ? k + 2 : k + 1) };
         piece(q, k + 2 < m ? 4 : 3);
      }
   }
   __device__ __forceinline__ smr_dface_t face(int2 p, int32_t ddx, int32_t ddy, double ux, double uy) const
   {
      smr_dface_t f;
      double fx = -uy * hw, fy = ux * hw;
      fx *= s; fy *= s;
      const int32_t ox = (int32_t) rint(fx * 256.0), oy = (int32_t) rint(fy * 256.0);
      f.point = p; f.ux = ux; f.uy = uy; f.ddx = ddx; f.ddy = ddy;
      f.ccw = make_int2(p.x + ox, p.y + oy); f.cw = make_int2(p.x - ox, p.y - oy);
      return f;
   }
   __device__ void join(const smr_dface_t &in, const smr_dface_t &o)
   {
      if (in.cw.x == o.cw.x && in.cw.y == o.cw.y && in.ccw.x == o.ccw.x && in.ccw.y == o.ccw.y) return;
      const bool clockwise = smr_slope_cmp(o.cw.x - o.point.x, o.cw.y - o.point.y, in.cw.x - in.point.x, in.cw.y - in.point.y) < 0;
      const int2 inpt = clockwise ? in.ccw : in.cw, outpt = clockwise ? o.ccw : o.cw;
      const double in_dot_out = -in.ux * o.ux + -in.uy * o.uy;
      if (2 <= 100.0 * (1 - in_dot_out))
      {
         double x1 = inpt.x / 256.0, y1 = inpt.y / 256.0, dx1 = in.ux * s, dy1 = in.uy * s;
         double x2 = outpt.x / 256.0, y2 = outpt.y / 256.0, dx2 = o.ux * s, dy2 = o.uy * s;
         double my = (((x2 - x1) * dy1 * dy2 - y2 * dx2 * dy1 + y1 * dx1 * dy2) / (dx1 * dy2 - dx2 * dy1));
         double mx;
         if (fabs(dy1) >= fabs(dy2)) mx = (my - y1) * dx1 / dy1 + x1;
         else mx = (my - y2) * dx2 / dy2 + x2;
         double ix = in.point.x / 256.0, iy = in.point.y / 256.0;
         double fdx1 = x1 - ix, fdy1 = y1 - iy, fdx2 = x2 - ix, fdy2 = y2 - iy, mdx = mx - ix, mdy = my - iy;
         if (smr_sgn_cmp(fdx1, fdy1, mdx, mdy) != smr_sgn_cmp(fdx2, fdy2, mdx, mdy))
         {
            int2 q[4] = { in.point, inpt, make_int2((int32_t) rint(mx * 256.0), (int32_t) rint(my * 256.0)), outpt };
            piece(q, 4);
            return;
         }
      }
      int2 q[3] = { in.point, inpt, outpt };
      piece(q, 3);
   }
};

template <bool EMIT>
__global__ void k_dash(dash_args_t A)
{
   const int64_t w = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (w >= A.n_ways) return;
   const uint8_t wf = A.wayflags[w];
   const bool on = A.mode == 1 ? (wf & 8) != 0 : (wf & 2) != 0;
   smr_dasher_t D;
   D.cnt = 0; D.out = nullptr;
   if (on && !(A.round_only && !(A.cap_mode == 1 || (A.cap_mode == 2 && (uint32_t) w != *A.first_way) || (A.cap_mode == 3 && (uint32_t) w > *A.first_way))))
   {
      const int64_t a = A.way_off[w], b = A.way_off[w + 1];
      const int cls = A.mode == 1 || !A.fill_used ? 0 : ((wf & 1) ? 2 : 1);
      D.hw = A.mode == 1 ? A.hw_fill : (!A.fill_used ? A.hw_nofill : ((wf & 1) ? A.hw_closed : A.hw_open));
      D.s = (double) A.dpi / 72;
      D.round = A.cap_mode == 1 || (A.cap_mode == 2 && (uint32_t) w != *A.first_way) || (A.cap_mode == 3 && (uint32_t) w > *A.first_way);
      D.pen = A.pen + A.pen_off[cls]; D.npen = A.pen_n[cls];
      if (EMIT) D.out = A.edges + 4 * (size_t) A.piece_off[w];
      if (A.solid)
      {
         /* a solid stroke whose caps are round: its pieces, then the half disc over the pen's vertices at the first segment's
          * start (the face turned round) and at the last segment's end (add_cap / add_fan of cairo-path-stroke-polygon.c) */
         bool has_first = false;
         smr_face_t ff, lf; int32_t fdx = 0, fdy = 0, ldx = 0, ldy = 0;
         ff.ccw = ff.cw = lf.ccw = lf.cw = make_int2(0, 0);
         int2 fpt = make_int2(0, 0), lpt = make_int2(0, 0);
         for (int64_t g = a; g + 1 < b; g++)
         {
            smr_edge_t loc[8];
            smr_edge_t z; z.x1 = z.y1 = z.x2 = z.y2 = 0;
            for (int i = 0; i < 8; i++) loc[i] = z;
            smr_face_t sf; int32_t ddx, ddy;
            if (A.tin)
            {
               /* along a decomposed curve: the end faces lie along the tangents */
               smr_face_t ef; int2 sd, ed, ep;
               if (!smr_stroke_ref_pieces_tan(A.pts, A.tin, A.tout, A.skip, a, b, g, D.hw, D.s, loc, sf, sd, ef, ed, ep)) continue;
               D.raw4(loc); D.raw4(loc + 4);
               if (!has_first) { has_first = true; ff = sf; fdx = sd.x; fdy = sd.y; fpt = A.pts[g]; }
               lf = ef; ldx = ed.x; ldy = ed.y; lpt = ep;
               continue;
            }
            if (!smr_stroke_ref_pieces(A.pts, a, b, g, D.hw, D.s, loc, sf, ddx, ddy)) continue;
            D.raw4(loc); D.raw4(loc + 4);
            if (!has_first) { has_first = true; ff = sf; fdx = ddx; fdy = ddy; fpt = A.pts[g]; }
            lf = sf; lf.ccw.x += ddx; lf.ccw.y += ddy; lf.cw.x += ddx; lf.cw.y += ddy; ldx = ddx; ldy = ddy; lpt = A.pts[g + 1];
         }
         if (D.round && has_first)
         {
            int cs, ce;
            if (D.fan_range(-fdx, -fdy, cs, ce)) D.round_cap(ff.cw, ff.ccw, fpt, cs, ce);
            if (D.fan_range(ldx, ldy, cs, ce)) D.round_cap(lf.ccw, lf.cw, lpt, cs, ce);
         }
         if (!EMIT) A.piece_cnt[w] = D.cnt;
         return;
      }
      /* _cairo_stroker_dash_start, offset 0 */
      int di = 0; bool dash_on = true;
      double dash_remain = A.dash[0];
      bool has_first = false, has_cur = false;
      smr_dface_t first, cur;
      int2 p1 = A.pts[a];
      /* _cairo_stroker_line_to_dashed; round_join: inside a decomposed curve */
      auto line_to = [&](const int2 p2, const bool round_join)
      {
         if (p1.x == p2.x && p1.y == p2.y) return;
         const int32_t ddx = p2.x - p1.x, ddy = p2.y - p1.y;
         double sdx = ddx / 256.0 * (1.0 / D.s), sdy = ddy / 256.0 * (1.0 / D.s), mag;
         if (sdx == 0.0) { if (sdy > 0.0) { mag = sdy; sdy = 1.0; } else { mag = -sdy; sdy = -1.0; } }
         else if (sdy == 0.0) { if (sdx > 0.0) { mag = sdx; sdx = 1.0; } else { mag = -sdx; sdx = -1.0; } }
         else { mag = smr_hypot(sdx, sdy); sdx = sdx / mag; sdy = sdy / mag; }
         /* the caps of this segment's dashes all look along (ddx, ddy) or against it */
         int ts = 0, te = 0, ls = 0, le = 0;
         bool tcap = false, lcap = false;
         if (D.round) { tcap = D.fan_range(ddx, ddy, ts, te); lcap = D.fan_range(-ddx, -ddy, ls, le); }
         double remain = mag;
         int2 pa = p1;
         uint32_t guard = 0;
         while (remain > 0 && ++guard < (1u << 20))        /* the host refuses dash arrays that could spin here */
         {
            const double step = dash_remain < remain ? dash_remain : remain;
            remain -= step;
            const double dx2 = sdx * (mag - remain) * D.s, dy2 = sdy * (mag - remain) * D.s;
            const int2 pb = make_int2((int32_t) rint(dx2 * 256.0) + p1.x, (int32_t) rint(dy2 * 256.0) + p1.y);
            if (dash_on)
            {
               smr_dface_t ss = D.face(pa, ddx, ddy, sdx, sdy), se = ss;
               if (pa.x != pb.x || pa.y != pb.y)
               {
                  se.point = pb;
                  se.ccw.x += pb.x - pa.x; se.ccw.y += pb.y - pa.y; se.cw.x += pb.x - pa.x; se.cw.y += pb.y - pa.y;
                  int2 q[4] = { ss.ccw, se.ccw, se.cw, ss.cw };
                  D.piece(q, 4);
               }
               if (has_cur) { if (round_join) D.join_round(cur, ss); else D.join(cur, ss); has_cur = false; }
               else if (!has_first) { first = ss; has_first = true; }                    /* the dash pattern starts "on" */
               else if (lcap) D.round_cap(ss.cw, ss.ccw, ss.point, ls, le);            /* leading cap: the face turned round */
               if (remain) { if (tcap) D.round_cap(se.ccw, se.cw, se.point, ts, te); }
               else { cur = se; has_cur = true; }
            }
            else if (has_cur)
            {
               /* cur belongs to the previous segment: its own direction decides the pen range */
               if (D.round)
               {
                  int cs, ce;
                  if (D.fan_range(cur.ddx, cur.ddy, cs, ce)) D.round_cap(cur.ccw, cur.cw, cur.point, cs, ce);
               }
               has_cur = false;
            }
            /* _cairo_stroker_dash_step */
            dash_remain -= step;
            if (dash_remain < 1.0 / 512)
            {
               if (++di == A.ndash) di = 0;
               dash_on = !dash_on;
               dash_remain += A.dash[di];
            }
            pa = pb;
         }
         if (dash_on && !has_cur)
         {
            /* the segment ends exactly where a dash begins: a leading cap at the vertex */
            cur = D.face(p2, ddx, ddy, sdx, sdy);
            if (lcap) D.round_cap(cur.cw, cur.ccw, cur.point, ls, le);
            has_cur = true;
         }
         p1 = p2;
      };
      /* a face at the current point along a curve's tangent */
      auto tangent_face = [&](const int2 t) -> smr_dface_t
      {
         double sdx = t.x / 256.0 * (1.0 / D.s), sdy = t.y / 256.0 * (1.0 / D.s);
         if (sdx == 0.0) sdy = sdy > 0.0 ? 1.0 : -1.0;
         else if (sdy == 0.0) sdx = sdx > 0.0 ? 1.0 : -1.0;
         else { const double mag = smr_hypot(sdx, sdy); sdx = sdx / mag; sdy = sdy / mag; }
         return D.face(p1, t.x, t.y, sdx, sdy);
      };
      if (!A.pflag)
         for (int64_t g = a + 1; g < b; g++) line_to(A.pts[g], false);
      else
         /* _cairo_stroker_curve_to of libcairo's dashing stroker (oracle: orc_dash_edges_curve): where a curve begins and a dash is
          * on, a face along its initial tangent, joined to what came before with the context's join; the decomposed points are
          * dashed line_to's with faces along the chords and ROUND joins; where it ends, a face along the final tangent */
         for (int64_t i = a + 1; i < b;)
         {
            int64_t j = i;
            while (j < b - 1 && !(A.pflag[j] & 1)) j++;
            if (A.pflag[j] & 2)
            {
               for (int64_t k = i; k <= j; k++) line_to(A.pts[k], false);
               i = j + 1;
               continue;
            }
            if (dash_on)
            {
               const int2 t = A.tout[i - 1];
               if (t.x != 0 || t.y != 0)
               {
                  const smr_dface_t f = tangent_face(t);
                  if (has_cur) D.join(cur, f);
                  else if (!has_first) { first = f; has_first = true; }
                  cur = f; has_cur = true;
               }
            }
            for (int64_t k = i; k <= j; k++) line_to(A.pts[k], true);
            if (dash_on)
            {
               const int2 t = A.tin[j];
               if (t.x != 0 || t.y != 0)
               {
                  const smr_dface_t f = tangent_face(t);
                  if (has_cur) D.join_round(cur, f);
                  cur = f; has_cur = true;
               }
            }
            i = j + 1;
         }
      /* _cairo_stroker_add_caps */
      if (D.round)
      {
         int cs, ce;
         if (has_first && D.fan_range(-first.ddx, -first.ddy, cs, ce)) D.round_cap(first.cw, first.ccw, first.point, cs, ce);
         if (has_cur && D.fan_range(cur.ddx, cur.ddy, cs, ce)) D.round_cap(cur.ccw, cur.cw, cur.point, cs, ce);
      }
   }
   if (!EMIT) A.piece_cnt[w] = D.cnt;
}

/* id room of a way whose strokes are dashed: enough 8-slot groups for its refs and for each of its piece lists */
__global__ void k_id_groups(const int64_t *__restrict__ way_off, int64_t n_ways, const uint32_t *__restrict__ c0,
                            const uint32_t *__restrict__ c1, const uint32_t *__restrict__ c2, uint32_t *__restrict__ out)
{
   const int64_t w = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (w >= n_ways) return;
   uint32_t g = (uint32_t) (way_off[w + 1] - way_off[w]);
   if (c0) g = max(g, (c0[w] + 1) >> 1);
   if (c1) g = max(g, (c1[w] + 1) >> 1);
   if (c2) g = max(g, (c2[w] + 1) >> 1);
   out[w] = g;
}

/* 0xTTRRGGBB -> premultiplied source pixel, per way (color=%key rules) */
__global__ void k_color_pixels(const uint32_t *__restrict__ col, int64_t n, uint32_t *__restrict__ pix)
{
   int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   uint32_t c = col[i];
   double r = (double) ((c >> 16) & 0xff) / 255.0, g = (double) ((c >> 8) & 0xff) / 255.0, b = (double) (c & 0xff) / 255.0;
   double a = 1.0 - (double) (((c & 0x7f000000u) >> 23) & 0xff) / 255.0;
   auto d2s = [](double d) { return (uint32_t) (uint16_t) (d * 65535.0 + 0.5); };      /* _cairo_color_double_to_short of libcairo 1.16 */
   pix[i] = ((d2s(a) >> 8) << 24) | ((d2s(r * a) >> 8) << 16) | ((d2s(g * a) >> 8) << 8) | (d2s(b * a) >> 8);
}

/* ------------------------------------------------------------------------------------ */
/* primitive setup: ONE WAVEFRONT per primitive.  Reads the primitive's edge slots once
 * (coalesced), and leaves
 *   - its items (64-slot chunks) with their binning boxes,
 *   - its bbox, validity against the full chart, colour,
 *   - its ROW MASK: one bit per chart row, set when libcairo's tor converter sub-samples the
 *     row (an edge starts or ends inside it, or active edges cross inside it), clear when it
 *     takes the exact full-row path.  The decision is global per (primitive, row), so it is
 *     made here once instead of by every tile.
 * Item ids are a closed form of (way, kind, chunk): base_w = (way_off[w] >> 3) + w is strictly
 * increasing in w and leaves room for ceil(n/8) chunks per kind, so no scan is needed and ids
 * ascend in drawing order (way, border before fill, chunk).                               */
/* ------------------------------------------------------------------------------------ */
#ifndef SMR_LONGCAP
/* k_cover (below): which primitives it takes, and what its arenas must hold for them (k_prim_setup adds it up) */
#define SMR_CV_WC 128                              /* cells per row, one guard cell on either side included */
#define SMR_CV_SL 16                               /* edges spanning one vertex-free row */
#define SMR_CV_MAXSLOTS 1024u
#define SMR_CV_MAXGROUPS (SMR_CV_MAXSLOTS / 64u)
#ifndef SMR_CV_WAVES
#define SMR_CV_WAVES 4
#endif
#define SMR_CV_DBLOCK 4096u                        /* alpha words a wavefront takes from the arena with one atomic */
#define SMR_CV_RBLOCK 1024u                        /* strip-table entries likewise */
#define SMR_CV_CHUNK 8u                            /* primitives a wavefront takes from the queue with one atomic */

#define SMR_LONGCAP 256      /* edges spanning a whole pixel row, kept in LDS per primitive */
#endif
#define SMR_ACTCAP 8         /* active edges per vertex-free row in the lane-per-row analysis */

/* a long primitive (SMR_BIG_SLOTS edge slots and more: a contour line, a coastline) is not set up by ONE wavefront: k_prim_setup
 * only registers it, and its items, its box, its vertex rows and its vertex-free rows are then spread over all wavefronts
 * (k_big_items, k_big_finish, k_big_vrows, k_big_rows) */
struct smr_bigacc_t
{
   int32_t minx, miny, maxx, maxy, bin_maxx;     /* atomicMin / atomicMax over the items */
   uint32_t item_flags, prim_id, valid;          /* SMR_PF_* of the items; the id the items carry (compound or own); set by k_big_finish */
   int32_t clip0, clip1;                         /* the rows the primitive may touch (a window of an atlas), else 0 .. H */
   int32_t lim_l, lim_r, lim_on;                 /* smr_setup_cuts, for k_big_vrows */
   int32_t sparse;                               /* a stroke's arena (8 slots per ref or 4 per piece, empty ones zeroed), pieces or the stroker's polygon */
};

struct setup_args_t
{
   const smr_edge_t *edges;
   const int64_t *way_off;
   const uint8_t *wayflags;
   int64_t n_ways;
   uint64_t off_border, off_fill, off_open;
   uint32_t border_pix, fill_pix;
   const uint32_t *way_border_pix, *way_fill_pix;
   int W, H;
   smr_prim_t *prims;
   smr_item_t *items;
   uint32_t *mask;
   uint32_t mask_cap;         /* words */
   uint32_t *small;           /* [1] first_prim [2] status [4] mask words demanded */
   uint32_t compound;         /* primitive id all items belong to (directional compound fill), 0xffffffff: their own */
   /* dashed strokes: piece lists (4 slots per piece) instead of the 8 slots per ref, and an id room of their own */
   int K;                     /* primitives per way: border, [the pipe's dots,] fill */
   uint64_t off_dots;
   const uint32_t *po_border, *po_dots, *po_open;      /* [n_ways+1] piece offsets; NULL: the plain layout */
   const uint32_t *id_off;                             /* [n_ways+1] in 8-slot groups; NULL: way_off */
   uint32_t scancap;                                   /* spanning edges per row the analysis of a stroke compares (smr_rowscan_t::cap) */
   int band_y0, band_h;                                /* this canvas's rows (k_cover's arenas are sized for them) */
   uint32_t *biglist;                                  /* long primitives, left to the k_big_* kernels ([0] = their number, then (id, first item)) */
   struct smr_bigacc_t *bigacc;                        /* [biglist entry] what those kernels need of them */
   int cv_split;                                       /* k_cover will take a primitive's strips in parts: their strip-table entries are handed out here (small[15]) */
   const int2 *way_clip;                               /* [n_ways] rows [x, y) a way's primitives may touch: its window of an atlas (smr_hip_draw_windows); NULL: the chart */
   int32_t lim_l, lim_r; int lim_on;                   /* the replay's clip box, left and right side, 24.8 (smr_setup_cuts) */
   uint32_t *win_first; int win_pitch;                  /* [windows] the first primitive that is anything in a window of an atlas (small[1] for the chart): every window is a
                                                          replay of its own for libcairo, and the first drawing into its still-clear group is a SOURCE (k_tile_sort) */
   const int4 *way_lim;                                /* [n_ways] the same per way (x, y; z, w: its top and bottom, for k_curve) (a window of an atlas: the image libcairo makes of the source there); NULL: lim_l, lim_r */
   int evenodd;                                        /* fill polygons take libcairo's EVEN_ODD rule (smr_draw_style_t.evenodd) */
   /* solid strokes with round caps (the rule's caps are sticky, src/smrules_cairo.c:909-921) come as piece lists of the dasher with
    * one endless dash: they stay OUR pieces (no cuts at the page's sides).  mix_border: only the ways after small[14] (the first
    * open way) have such a list, at off_border_po; the others keep the plain layout at off_border */
   int solid_border, solid_open, mix_border;
   uint64_t off_border_po;
};

struct setup_lds_t
{
   int4 l_edge[SMR_LONGCAP];       /* p1x, p1y, dxs, dy of the oriented edge */
   int act0[64][SMR_ACTCAP + 1], act1[64][SMR_ACTCAP + 1];
   uint32_t nlong;
};

/* wave-wide min / max through DPP row shifts and broadcasts (VALU only; the __shfl_xor form is an LDS round trip per step):
 * lane 63 ends up with the reduction of all 64 lanes, v_readlane hands it to everyone */
__device__ __forceinline__ int smr_wave_min(int v)
{
   v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
   v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
   v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
   v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
   v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
   v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
   return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int smr_wave_max(int v)
{
   v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
   v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
   v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
   v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
   v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
   v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
   return __builtin_amdgcn_readlane(v, 63);
}

/* libcairo replays the chart under the recording surface's rectangle as clip, and a polygon built under a clip is LIMITED to
 * the clip's box (_cairo_polygon_init_with_clip): an edge that crosses the box's left or right side is cut there
 * (_add_clipped_edge, cairo-polygon.c), the part outside going straight down the side.  The coverage is the same, but the cut
 * is a vertex, and the pixel row of a vertex is sub-sampled.  Restated in oracle/tor.c (limit_cuts), step by step the same
 * integers.  Only for primitives whose edges ARE the polygon's (fills, strokes as polygons): the inner edges of a stroke's
 * pieces are not libcairo's. */
__device__ __forceinline__ int64_t smr_floor_div(int64_t a, int64_t b)
{
   int64_t q = a / b;
   if ((a % b != 0) && ((a < 0) != (b < 0))) q--;
   return q;
}
__device__ __forceinline__ int64_t smr_y_for_x(int64_t p1x, int64_t p1y, int64_t p2x, int64_t p2y, int64_t x)
{
   if (x == p1x) return p1y;
   if (x == p2x) return p2y;
   const int64_t dx = p2x - p1x;
   return dx ? p1y + smr_floor_div((x - p1x) * (p2y - p1y), dx) : p1y;
}
__device__ __forceinline__ int64_t smr_x_for_y(int64_t p1x, int64_t p1y, int64_t p2x, int64_t p2y, int64_t y)
{
   if (y == p1y) return p1x;
   if (y == p2y) return p2x;
   const int64_t dy = p2y - p1y;
   return dy ? p1x + smr_floor_div((y - p1y) * (p2x - p1x), dy) : p1x;
}
__device__ __forceinline__ void smr_setup_cuts(const smr_edge_t &e, int32_t lim_l, int32_t lim_r, int row0, int nrows, uint32_t *mask)
{
   if (e.y1 == e.y2) return;
   const bool up = e.y1 > e.y2;
   const int64_t p1x = up ? e.x2 : e.x1, p1y = up ? e.y2 : e.y1, p2x = up ? e.x1 : e.x2, p2y = up ? e.y1 : e.y2;
   const int64_t L = lim_l, R = lim_r, pleft = min(p1x, p2x), pright = max(p1x, p2x);
   if ((L <= pleft && pright <= R) || pright <= L || R <= pleft) return;
   int64_t top_y = p1y, bot_y = p2y, left_y, right_y, cut[2];
   int n = 0;
   if (p1x <= p2x)
   {
      if (pleft >= L) left_y = top_y;
      else { left_y = smr_y_for_x(p1x, p1y, p2x, p2y, L); if (smr_x_for_y(p1x, p1y, p2x, p2y, left_y) < L) left_y++; }
      left_y = min(left_y, bot_y);
      if (top_y < left_y) { cut[n++] = left_y; top_y = left_y; }
      if (pright <= R) right_y = bot_y;
      else { right_y = smr_y_for_x(p1x, p1y, p2x, p2y, R); if (smr_x_for_y(p1x, p1y, p2x, p2y, right_y) > R) right_y--; }
      right_y = max(right_y, top_y);
      if (bot_y > right_y) cut[n++] = right_y;
   }
   else
   {
      if (pright <= R) right_y = top_y;
      else { right_y = smr_y_for_x(p1x, p1y, p2x, p2y, R); if (smr_x_for_y(p1x, p1y, p2x, p2y, right_y) > R) right_y++; }
      right_y = min(right_y, bot_y);
      if (top_y < right_y) { cut[n++] = right_y; top_y = right_y; }
      if (pleft >= L) left_y = bot_y;
      else { left_y = smr_y_for_x(p1x, p1y, p2x, p2y, L); if (smr_x_for_y(p1x, p1y, p2x, p2y, left_y) < L) left_y--; }
      left_y = max(left_y, top_y);
      if (bot_y > left_y) cut[n++] = left_y;
   }
   for (int k = 0; k < n; k++)
   {
      const int gy = smr_to_grid_y((int32_t) cut[k]);
      if (gy <= 0 || gy % SMR_GRID_Y == 0) continue;          /* on a row's boundary: no part starts or ends inside a row */
      const int r = gy / SMR_GRID_Y - row0;
      if (r >= 0 && r < nrows) atomicOr(&mask[r >> 5], 1u << (r & 31));
   }
}

/* mark the rows in which edge o starts or ends, and remember o if it spans a whole row */
__device__ __forceinline__ void smr_setup_edge(const smr_oedge_t &o, int row0, int nrows, uint32_t *lmask, setup_lds_t &L)
{
   if (o.dir == 0) return;
   if (o.ytop % SMR_GRID_Y != 0)
   {
      int r = o.ytop / SMR_GRID_Y - row0;             /* ytop >= 0 whenever the row is on the chart */
      if (o.ytop >= 0 && r >= 0 && r < nrows) atomicOr(&lmask[r >> 5], 1u << (r & 31));
   }
   if (o.ybot % SMR_GRID_Y != 0)
   {
      int r = (o.ybot > 0 ? (o.ybot - 1) / SMR_GRID_Y : -1) - row0;
      if (o.ybot > 0 && r >= 0 && r < nrows) atomicOr(&lmask[r >> 5], 1u << (r & 31));
   }
   int first_full = o.ytop >= 0 ? (o.ytop + SMR_GRID_Y - 1) / SMR_GRID_Y : -((-o.ytop) / SMR_GRID_Y);   /* ceil(ytop/15) */
   if ((first_full + 1) * SMR_GRID_Y <= o.ybot)
   {
      uint32_t k = atomicAdd(&L.nlong, 1u);
      if (k < SMR_LONGCAP)
      {
         L.l_edge[k] = make_int4(o.p1x, o.p1y, o.dxs, o.D / 60);
      }
   }
}

/* What lies beyond a side of the limits libcairo has projected onto that side (_add_clipped_edge: a vertical edge there) before its scan
 * converter looks: two edges that cross out there do not cross for it.  An edge is that vertical piece in a row when it is beyond the
 * side at the row's first AND last sample (c0, cl); one that reaches the side at the row's bottom is still its own line there, and the
 * cell libcairo steps it to (c1) lies beyond the side.  (An edge that crosses a side inside the row has a cut there, and the row is
 * sub-sampled anyway.)  lo = INT32_MIN, hi = INT32_MAX: no limits.  Oracle: tor_coverage's row decision. */
__device__ __forceinline__ bool smr_beyond(int c0, int32_t lo, int32_t hi) { return c0 < lo || c0 > hi; }
__device__ __forceinline__ void smr_project(int &c0, int &c1, int cl, int32_t lo, int32_t hi)
{
   if (c0 < lo && cl < lo) c0 = c1 = lo;
   else if (c0 > hi && cl > hi) c0 = c1 = hi;
}

/* A fill's slots are the path's segments in the path's order, the order in which libcairo's polygon got its edges.  Two edges that
 * START at a row's first sub-row in the same cell -- the two sides of a vertex a hair above the row's first sample line -- stand in
 * libcairo's active list in that order (merge_sorted_edges keeps ties as they come), and when it is right side before left side
 * their cells at the row's bottom come out descending: can_do_full_row says no and the row is sub-sampled (oracle:
 * tor_set_path_order, pinned on libcairo).  Slot idx looks at the vertex it starts from, with the slot that ends there. */
__device__ __forceinline__ void smr_tie_check(const smr_edge_t *__restrict__ E, uint32_t edge_cnt, uint32_t idx, const smr_edge_t &e,
                                              int row0, int nrows, uint32_t *mask, int32_t cl_lo, int32_t cl_hi)
{
   if (e.y2 <= e.y1) return;                          /* the side that LEAVES the vertex downwards */
   const smr_oedge_t o = smr_orient(e);
   if (o.dir == 0 || o.ytop % SMR_GRID_Y != 0 || o.ybot < o.ytop + SMR_GRID_Y) return;
   const int r = o.ytop / SMR_GRID_Y - row0;
   if (o.ytop < 0 || r < 0 || r >= nrows) return;
   /* the slot before, passing over empty ones (a node given twice) */
   uint32_t p = idx;
   smr_edge_t q;
   for (uint32_t k = 0;; k++)
   {
      if (k >= edge_cnt) return;
      p = p == 0 ? edge_cnt - 1 : p - 1;
      q = E[p];
      if (q.x1 != q.x2 || q.y1 != q.y2) break;
   }
   if (q.x2 != e.x1 || q.y2 != e.y1 || q.y1 <= q.y2) return;      /* it arrives at the same vertex from below */
   const smr_oedge_t op = smr_orient(q);
   if (op.dir == 0 || op.ytop != o.ytop || op.ybot < o.ytop + SMR_GRID_Y) return;
   const int rs0 = o.ytop, rs1 = rs0 + SMR_GRID_Y;
   auto cell = [&](const smr_oedge_t &g, int rs) { return smr_edge_cell(256 - 30ll * g.p1y, 2 * g.dxs, g.D, smr_rcp((double) g.D), g.p1x, rs); };
   int t_this = cell(o, rs0), b_this = cell(o, rs1), t_prev = cell(op, rs0), b_prev = cell(op, rs1);
   if (smr_beyond(t_this, cl_lo, cl_hi)) smr_project(t_this, b_this, cell(o, rs1 - 1), cl_lo, cl_hi);
   if (smr_beyond(t_prev, cl_lo, cl_hi)) smr_project(t_prev, b_prev, cell(op, rs1 - 1), cl_lo, cl_hi);
   if (t_this != t_prev) return;
   const bool prev_first = p < idx;                   /* (the path's first segment stands before its last) */
   if (prev_first ? b_prev > b_this : b_this > b_prev) atomicOr(&mask[r >> 5], 1u << (r & 31));
}

#define SMR_SCANCAP (64u * (SMR_ACTCAP + 1))      /* act0 / act1 of setup_lds_t as flat arrays */
#define SMR_BIG_SLOTS 4096u   /* edge slots from which a primitive with an overflowing row-spanning list goes to k_big_rows */

/* analysis of the vertex-free rows of one primitive: which of them libcairo's scan converter sub-samples after all
 * because two of the edges spanning the row cross inside it.  Used by the wavefront that sets the primitive up and,
 * for long primitives, by the wavefronts of k_big_rows (one window of 64 rows each). */
struct smr_rowscan_t
{
   int row0, nrows; uint32_t edge_cnt;
   const smr_edge_t *E;
   setup_lds_t *L;
   uint32_t *lmask;               /* LDS: bit r - mbase = row r of the primitive is sub-sampled */
   int mbase;                     /* first row lmask holds (a multiple of 64) */
   int lane;
   uint32_t *status;
   uint32_t list_n;               /* entries of L->l_edge that count; more than SMR_LONGCAP: the list overflowed, scan every slot */
   const smr_item_t *items;       /* the primitive's items (64 slots each, with their boxes), or NULL: build_list skips the items that
                                     cannot reach the window instead of reading every slot of a long primitive once per window */
   int32_t cl_lo, cl_hi;          /* the sides of the limits (smr_project); INT32_MIN / INT32_MAX: none */
   uint32_t cap;                  /* spanning edges per row that row_by_wave compares: SMR_ROWCAP for fills (what k_raster's exact path
                                     holds of one row in one tile: fill chunks bring every edge left of a tile along), SMR_SCANCAP for
                                     strokes, whose pieces reach a tile only with the items binned there */

   /* one vertex-free row decided by the whole wavefront: lanes walk the candidate edges (the LDS list of row-spanning
    * edges, or every edge slot when that list overflowed), the active ones' cells at the top and the bottom of the row
    * go to LDS, then every pair is compared */
   __device__ void row_by_wave(int r)
   {
#ifdef SMR_KO_SETUP
      if (SMR_KO_SETUP & 2) return;
#endif
      const int rs0 = (row0 + r) * SMR_GRID_Y, rs1 = rs0 + SMR_GRID_Y;
      int *f0 = &L->act0[0][0], *f1 = &L->act1[0][0];
      uint32_t cnt = 0;
      const uint32_t ncand = list_n <= SMR_LONGCAP ? list_n : edge_cnt;
      SMR_WAVE_SYNC();
      auto visit = [&](uint32_t idx)
      {
         int32_t p1x = 0, p1y = 0, dxs = 0, D = 1, ytop = 0, ybot = 0;
         bool have = false;
         if (idx < ncand)
         {
            if (list_n <= SMR_LONGCAP)
            {
               const int4 le = L->l_edge[idx];
               p1x = le.x; p1y = le.y; dxs = le.z; D = 60 * le.w; ytop = smr_to_grid_y(le.y); ybot = smr_to_grid_y(le.y + le.w);
               have = true;
            }
            else
            {
               smr_oedge_t o = smr_orient(E[idx]);
               p1x = o.p1x; p1y = o.p1y; dxs = o.dxs; D = o.D; ytop = o.ytop; ybot = o.ybot;
               have = o.dir != 0;
            }
         }
         bool act = have && ytop <= rs0 && ybot >= rs1;
         unsigned long long am = __ballot(act);
         if (!am) return;
         uint32_t pos = cnt + __popcll(am & ((1ull << lane) - 1ull));
         if (act && pos < cap)
         {
            /* the rounded cells at the top and the bottom of the row (exact either way: 32 bits for a short edge, else the
             * reciprocal estimate with its integer correction) */
            if (abs(dxs) < 4096 && D <= 60 * 4096)
            {
               const float rf = __builtin_amdgcn_rcpf((float) D);
               const int c32 = (int) ((uint32_t) rs0 * 512u + 256u - 30u * (uint32_t) p1y);
               const int n0 = __mul24(2 * dxs, c32) + (D >> 1);
               auto cell32 = [&](int n) -> int
               {
                  const int q = (int) fmaf((float) n, rf, -0.5f);
                  const int rem = __mul24(q, -D) + n;
                  return p1x + q + (rem >= D ? 1 : 0) + (rem >> 31);
               };
               int c0 = cell32(n0), c1 = cell32(n0 + __mul24(2 * dxs, 512 * SMR_GRID_Y));
               if (smr_beyond(c0, cl_lo, cl_hi)) smr_project(c0, c1, cell32(n0 + __mul24(2 * dxs, 512 * (SMR_GRID_Y - 1))), cl_lo, cl_hi);
               f0[pos] = c0; f1[pos] = c1;
            }
            else
            {
               const int64_t ec = 256 - 30ll * p1y;
               const double inv = smr_rcp((double) D);
               int c0 = smr_edge_cell(ec, 2 * dxs, D, inv, p1x, rs0), c1 = smr_edge_cell(ec, 2 * dxs, D, inv, p1x, rs1);
               if (smr_beyond(c0, cl_lo, cl_hi)) smr_project(c0, c1, smr_edge_cell(ec, 2 * dxs, D, inv, p1x, rs1 - 1), cl_lo, cl_hi);
               f0[pos] = c0; f1[pos] = c1;
            }
         }
         cnt += __popcll(am);
      };
      if (list_n > SMR_LONGCAP && items)
      {
         /* the list overflowed: every slot is a candidate, but only the items whose box meets the row hold any */
         const uint32_t nit = (edge_cnt + 63u) >> 6;
         for (uint32_t kb = 0; kb < nit; kb += 64)
         {
            bool ov = false;
            if (kb + lane < nit)
            {
               const smr_item_t *it = items + kb + lane;
               ov = it->cnt_flags != 0 && (it->maxy >> 8) >= row0 + r && (it->miny >> 8) <= row0 + r;
            }
            unsigned long long om = __ballot(ov);
            while (om)
            {
               const int j = __ffsll((long long) om) - 1;
               om &= om - 1;
               visit((kb + (uint32_t) j) * 64u + (uint32_t) lane);
            }
         }
      }
      else
         for (uint32_t base_i = 0; base_i < ncand; base_i += 64) visit(base_i + lane);
      SMR_WAVE_SYNC();
      bool viol = false;
      /* more spanning edges than the exact full-row path holds: the row is sub-sampled instead (reported) */
      if (cnt > cap) { if (lane == 0) atomicOr(status, SMR_ST_DEGRADED); viol = true; }
      else
         for (uint32_t i = lane; i < cnt; i += 64)
         {
            const int a0 = f0[i], a1 = f1[i];
            for (uint32_t j = 0; j < cnt; j++) if (a0 < f0[j] && a1 > f1[j]) viol = true;
         }
      if (__ballot(viol) && lane == 0) lmask[(r - mbase) >> 5] |= 1u << (r & 31);
      else if (cnt > SMR_ROWCAP && lane == 0) atomicOr(status, SMR_ST_ROWRISK);
      SMR_WAVE_SYNC();
   }

   /* 64 rows at a time, lane = row: every vertex-free row collects the edges of the list spanning it and checks that
    * their order at the top of the row is their order at the bottom; rows with more than SMR_ACTCAP active edges, and
    * windows whose list overflowed, are left to row_by_wave */
   __device__ void window(int rb)
   {
      const int r = rb + lane;
      const bool mine = r < nrows && !((lmask[(r - mbase) >> 5] >> (r & 31)) & 1u);
      const int rs0 = (row0 + r) * SMR_GRID_Y, rs1 = rs0 + SMR_GRID_Y;
      int cnt = 0;
      bool over = false;
      const unsigned long long rows = __ballot(mine);
      if (!rows) return;
      /* few vertex-free rows in this batch (or no list): one wavefront pass over the candidates per row is cheaper
       * than every lane walking the whole list */
      if (list_n > SMR_LONGCAP || __popcll(rows) * 8 <= (int) list_n)
      {
         unsigned long long rm = rows;
         while (rm)
         {
            int j = __ffsll((long long) rm) - 1;
            rm &= rm - 1;
            row_by_wave(rb + j);
         }
         return;
      }
      /* first which edges span the row (a compare per candidate), then their cells in a loop all lanes walk together:
       * the lanes' active edges sit at different places of the list, and evaluating inside the scan would make every
       * lane pay for every other lane's evaluations */
      /* the rows every list entry spans, once per entry (lanes = entries) instead of once per entry and row; they sit in act1's
       * storage, which the second phase only writes after the scan is through */
      int2 *span = reinterpret_cast<int2 *>(&L->act1[0][0]);
      static_assert(sizeof(L->act1) >= SMR_LONGCAP * sizeof(int2), "act1 holds the list's row spans during the scan");
      for (uint32_t i = lane; i < list_n; i += 64)
      {
         const int4 le = L->l_edge[i];
         const int ytop = smr_to_grid_y(le.y), ybot = smr_to_grid_y(le.y + le.w);
         /* rows r with ytop <= 15 r and 15 (r + 1) <= ybot (ytop >= 0 for every row on the chart; rows below 0 do not exist) */
         span[i] = make_int2(ytop > 0 ? (ytop + SMR_GRID_Y - 1) / SMR_GRID_Y : 0, (ybot >= 0 ? ybot / SMR_GRID_Y : -1) - 1);
      }
      SMR_WAVE_SYNC();
      const int myrow = row0 + r;
      /* eight spans at a time, read before any append: the appends go to LDS too, and the compiler keeps every later read
       * behind them */
      for (uint32_t i0 = 0; i0 < list_n; i0 += 8)
      {
         int2 sp[8];
#pragma unroll
         for (int u = 0; u < 8; u++) sp[u] = i0 + u < list_n ? span[i0 + u] : make_int2(1, 0);
#pragma unroll
         for (int u = 0; u < 8; u++)
            if (mine && !over && sp[u].x <= myrow && myrow <= sp[u].y)
            {
               if (cnt < SMR_ACTCAP) L->act0[lane][cnt++] = (int) (i0 + u);
               else over = true;
            }
      }
      SMR_WAVE_SYNC();
      const int kmax = smr_wave_max(over ? 0 : cnt);
      for (int k = 0; k < kmax; k++)
      {
         const bool on = !over && k < cnt;
         int4 le = make_int4(0, 0, 0, 1);
         if (on) le = L->l_edge[L->act0[lane][k]];
         const int32_t D = 60 * le.w;
         /* short edges (|dx|, dy < 16 px) evaluate in 32 bits, as in k_raster's crossing loop */
         if (!__ballot(on && !(abs(le.z) < 4096 && D <= 60 * 4096)))
         {
            if (on)
            {
               const float rf = __builtin_amdgcn_rcpf((float) D);
               const int c32 = (int) ((uint32_t) rs0 * 512u + 256u - 30u * (uint32_t) le.y);
               const int n0 = __mul24(2 * le.z, c32) + (D >> 1);
               auto cell32 = [&](int n) -> int
               {
                  const int q = (int) fmaf((float) n, rf, -0.5f);
                  const int rem = __mul24(q, -D) + n;
                  return le.x + q + (rem >= D ? 1 : 0) + (rem >> 31);
               };
               int c0 = cell32(n0), c1 = cell32(n0 + __mul24(2 * le.z, 512 * SMR_GRID_Y));
               if (smr_beyond(c0, cl_lo, cl_hi)) smr_project(c0, c1, cell32(n0 + __mul24(2 * le.z, 512 * (SMR_GRID_Y - 1))), cl_lo, cl_hi);
               L->act0[lane][k] = c0; L->act1[lane][k] = c1;
            }
         }
         else if (on)
         {
            const int64_t ec = 256 - 30ll * le.y;
            const double inv = smr_rcp((double) D);
            int c0 = smr_edge_cell(ec, 2 * le.z, D, inv, le.x, rs0), c1 = smr_edge_cell(ec, 2 * le.z, D, inv, le.x, rs1);
            if (smr_beyond(c0, cl_lo, cl_hi)) smr_project(c0, c1, smr_edge_cell(ec, 2 * le.z, D, inv, le.x, rs1 - 1), cl_lo, cl_hi);
            L->act0[lane][k] = c0; L->act1[lane][k] = c1;
         }
      }
      bool viol = false;
#ifdef SMR_KO_SETUP
      if (!(SMR_KO_SETUP & 4))
#endif
      if (!over)
      {
         /* the lane's cells into registers once (2 cnt reads instead of 2 cnt^2), then every pair */
         int c0[SMR_ACTCAP], c1[SMR_ACTCAP];
#pragma unroll
         for (int i = 0; i < SMR_ACTCAP; i++) { c0[i] = i < cnt ? L->act0[lane][i] : INT32_MAX; c1[i] = i < cnt ? L->act1[lane][i] : INT32_MAX; }
         /* a slot beyond cnt holds (MAX, MAX): it is never left of anything at the top, and nothing is right of it at the bottom */
#pragma unroll
         for (int i = 0; i < SMR_ACTCAP; i++)
         {
            if (i >= kmax) break;                    /* kmax: the wavefront's largest cnt (uniform) */
#pragma unroll
            for (int j = 0; j < SMR_ACTCAP; j++)
            {
               if (j >= kmax) break;
               if (i != j && c0[i] < c0[j] && c1[i] > c1[j] && j < cnt) viol = true;
            }
         }
      }
      if (mine && viol) atomicOr(&lmask[(r - mbase) >> 5], 1u << (r & 31));
      unsigned long long om = __ballot(over);
      while (om)
      {
         int j = __ffsll((long long) om) - 1;
         om &= om - 1;
         row_by_wave(rb + j);
      }
   }

   /* the list of the edges that span a row of the window [rb, rb + 64): one pass over the primitive's edge slots.
    * Returns false when the window has no vertex-free row (nothing to analyse). */
   __device__ bool build_list(int rb)
   {
      const int r = rb + lane;
      const bool mine = r < nrows && !((lmask[(r - mbase) >> 5] >> (r & 31)) & 1u);
      if (!__ballot(mine)) return false;
      const int wr0 = row0 + rb, wr1 = min(row0 + rb + 63, row0 + nrows - 1);
      if (lane == 0) L->nlong = 0;
      SMR_WAVE_SYNC();
      auto visit = [&](uint32_t idx)
      {
         if (idx >= edge_cnt) return;
         const smr_oedge_t o = smr_orient(E[idx]);
         if (o.dir == 0) return;
         const int first_full = o.ytop >= 0 ? (o.ytop + SMR_GRID_Y - 1) / SMR_GRID_Y : -((-o.ytop) / SMR_GRID_Y);
         const int last_full = (o.ybot >= 0 ? o.ybot / SMR_GRID_Y : -((-o.ybot + SMR_GRID_Y - 1) / SMR_GRID_Y)) - 1;
         if (first_full > last_full || first_full > wr1 || last_full < wr0) return;
         const uint32_t k = atomicAdd(&L->nlong, 1u);
         if (k < SMR_LONGCAP) L->l_edge[k] = make_int4(o.p1x, o.p1y, o.dxs, o.D / 60);
      };
      if (items)
      {
         /* lanes = 64 items at a time: whose box (of its slots' end points) meets the window's rows? */
         const uint32_t nit = (edge_cnt + 63u) >> 6;
         for (uint32_t kb = 0; kb < nit; kb += 64)
         {
            bool ov = false;
            if (kb + lane < nit)
            {
               const smr_item_t *it = items + kb + lane;
               ov = it->cnt_flags != 0 && (it->maxy >> 8) >= wr0 && (it->miny >> 8) <= wr1;
            }
            unsigned long long om = __ballot(ov);
            while (om)
            {
               const int j = __ffsll((long long) om) - 1;
               om &= om - 1;
               visit((kb + (uint32_t) j) * 64u + (uint32_t) lane);
            }
         }
      }
      else
         for (uint32_t idx = lane; idx < edge_cnt; idx += 64) visit(idx);
      SMR_WAVE_SYNC();
      list_n = L->nlong;
      return true;
   }
};

#define SMR_MASK_BLOCK 2048u   /* words a wavefront takes from the mask arena with one atomic */

__device__ __forceinline__ void smr_prim_setup_one(const setup_args_t &A, const int64_t p, const int lane, setup_lds_t &L, uint32_t *lmask,
                                                   uint32_t &arena_pos, uint32_t &arena_left, uint32_t &first_valid,
                                                   unsigned long long &need_alpha, uint32_t &need_strips)
{
   const int64_t w = p / A.K;
   const int kind = (int) (p - w * A.K);
   const int role = A.K == 3 ? kind : 2 * kind;                  /* 0 border, 1 the pipe's dots, 2 fill */
   const uint8_t wf = A.wayflags[w];
   const int64_t a = A.way_off[w], b = A.way_off[w + 1], n = b - a;
   const int64_t ua = A.id_off ? (int64_t) A.id_off[w] : a, ub = A.id_off ? (int64_t) A.id_off[w + 1] : b;
   const int64_t base = (ua >> 3) + w, span = (ub >> 3) + (w + 1) - base;
   const int64_t item0 = A.K * base + kind * span;
   const bool on = role != 2 ? (wf & 2) != 0 : (wf & 12) != 0;
   /* strokes keep 8 slots per ref (or 4 per dash piece), empty ones zeroed; a solid stroke of a way with a poked-through join
    * (k_stroke_risk) holds the stroker's polygon in them and is filled non-zero like a fill polygon */
   const bool sparse = role != 2 || (wf & 8);
   const uint32_t *const pob = A.po_border && !(A.mix_border && (uint32_t) w <= A.small[14]) ? A.po_border : nullptr;
   const bool pieces = sparse && !(role == 0 ? (wf & 16) != 0 && !pob : role == 2 ? (wf & 32) != 0 && !A.po_open : false);
   /* libcairo limits a polygon to the image it renders into; for a window of an atlas that image is the window's rows */
   const int clip0 = A.way_clip ? A.way_clip[w].x : 0, clip1 = A.way_clip ? A.way_clip[w].y : A.H;
   const int32_t lim_l = A.way_lim ? A.way_lim[w].x : A.lim_l, lim_r = A.way_lim ? A.way_lim[w].y : A.lim_r;
   smr_prim_t P;
   P.edge_off = 0; P.edge_cnt = 0; P.color = 0; P.minx = P.miny = INT32_MAX; P.maxx = P.maxy = INT32_MIN;
   P.flags = 0; P.mask_off = 0; P.row0 = 0; P.nrows = 0; P.drow_off = 0; P.pad_[0] = P.pad_[1] = P.pad_[2] = 0;
   int64_t nitems = 0;
   if (on)
   {
      auto stroke_range = [&](uint64_t off, const uint32_t *po)
      {
         if (po) { P.edge_off = off + 4ull * po[w]; P.edge_cnt = 4u * (po[w + 1] - po[w]); }
         else { P.edge_off = off + (uint64_t) SMR_SLOTS * a; P.edge_cnt = (uint32_t) (SMR_SLOTS * n); }
      };
      if (role == 0) stroke_range(pob ? A.off_border_po : A.off_border, pob);
      else if (role == 1) stroke_range(A.off_dots, A.po_dots);
      else if (wf & 4) { P.edge_off = A.off_fill + (uint64_t) a; P.edge_cnt = (uint32_t) n; }
      else stroke_range(A.off_open, A.po_open);
      P.color = role != 2 ? (A.way_border_pix ? A.way_border_pix[w] : A.border_pix) : (A.way_fill_pix ? A.way_fill_pix[w] : A.fill_pix);
      nitems = ((int64_t) P.edge_cnt + 63) >> 6;
   }
   /* every slot of this kind's id range is written: unused ones as invalid */
   for (int64_t i = nitems + lane; i < span; i += 64)
   {
      smr_item_t z; z.edge_off = 0; z.prim = (uint32_t) p; z.cnt_flags = 0; z.color = 0; z.minx = z.miny = z.maxx = z.maxy = 0;
      A.items[item0 + i] = z;
   }
   if (!on) { if (lane == 0) A.prims[p] = P; return; }
   const smr_edge_t *E = A.edges + P.edge_off;
   /* (the rule of cairo_fill only: a stroke's polygon, pieces or not, is filled non-zero whatever the context's fill rule) */
   const uint32_t item_flags = SMR_PF_VALID | (pieces ? SMR_PF_PIECES : 0u) | (nitems > 1 ? SMR_PF_MULTI : 0u) |
                               (A.evenodd && role == 2 && (wf & 4) ? SMR_PF_EVENODD : 0u);
   if (P.edge_cnt >= SMR_BIG_SLOTS && A.biglist)
   {
      if (lane == 0)
      {
         const uint32_t b = atomicAdd(&A.biglist[0], 1u);
         A.biglist[1 + 2 * b] = (uint32_t) p; A.biglist[2 + 2 * b] = (uint32_t) item0;
         smr_bigacc_t a;
         a.minx = a.miny = INT32_MAX; a.maxx = a.maxy = a.bin_maxx = INT32_MIN;
         a.item_flags = item_flags; a.prim_id = A.compound != 0xffffffffu ? A.compound : (uint32_t) p; a.valid = 0;
         a.clip0 = clip0; a.clip1 = clip1; a.sparse = sparse;
         a.lim_l = lim_l; a.lim_r = lim_r;
         a.lim_on = A.lim_on && (!pieces || (role == 0 ? pob != nullptr && !A.solid_border : role == 1 ? A.po_dots != nullptr : (wf & 8) && A.po_open != nullptr && !A.solid_open));
         A.bigacc[b] = a;
         A.prims[p] = P;                 /* edge range and colour; not valid before k_big_finish */
      }
      return;
   }

   /* ---- pass A: items + bbox ---- */
   int bin_maxx = INT32_MIN;
   smr_edge_t e0; e0.x1 = e0.y1 = e0.x2 = e0.y2 = 0;
   for (int64_t k = 0; k < nitems; k++)
   {
      uint32_t idx = (uint32_t) (k * 64) + lane;
      smr_edge_t e; e.x1 = e.y1 = e.x2 = e.y2 = 0;
      if (idx < P.edge_cnt) e = E[idx];
      if (k == 0) e0 = e;
      bool has = idx < P.edge_cnt && (e.x1 != e.x2 || e.y1 != e.y2 || !sparse);
      /* fill polygons: every vertex counts (one per slot); strokes: empty slots are all zero */
      if (sparse && e.x1 == 0 && e.y1 == 0 && e.x2 == 0 && e.y2 == 0) has = false;
      int mnx = has ? min(e.x1, e.x2) : INT32_MAX, mxx = has ? max(e.x1, e.x2) : INT32_MIN;
      int mny = has ? min(e.y1, e.y2) : INT32_MAX, mxy = has ? max(e.y1, e.y2) : INT32_MIN;
      /* the exact full-row path evaluates an edge at the row boundaries, up to half a sub-row beyond
       * its end points: the BINNING box of an edge that can span a whole row grows by that reach
       * (the primitive's own box, which clips the coverage like libcairo's extents, does not) */
      int ext = 0;
      if (has && mxy - mny >= 239) ext = (int) (((int64_t) (mxx - mnx) * 9) / (mxy - mny)) + 2;
      int bmn = smr_wave_min(has ? mnx - ext : INT32_MAX), bmx = smr_wave_max(has ? mxx + ext : INT32_MIN);
      mnx = smr_wave_min(mnx); mny = smr_wave_min(mny); mxx = smr_wave_max(mxx); mxy = smr_wave_max(mxy);
      P.minx = min(P.minx, mnx); P.miny = min(P.miny, mny); P.maxx = max(P.maxx, mxx); P.maxy = max(P.maxy, mxy);
      bin_maxx = max(bin_maxx, bmx);
      if (lane == 0)
      {
         smr_item_t it;
         it.edge_off = (uint32_t) (P.edge_off + (uint64_t) k * 64);
         it.prim = A.compound != 0xffffffffu ? A.compound : (uint32_t) p;
         uint32_t cnt = min(64u, P.edge_cnt - (uint32_t) (k * 64));
         it.cnt_flags = mnx <= mxx ? (cnt | (item_flags << 8)) : 0;
         it.color = P.color;
         it.minx = bmn; it.miny = mny; it.maxx = bmx; it.maxy = mxy;
         if (A.way_clip)
         {
            it.miny = max(it.miny, clip0 * 256); it.maxy = min(it.maxy, clip1 * 256 - 1);
            if (it.miny > it.maxy) it.cnt_flags = 0;
         }
         A.items[item0 + k] = it;
      }
   }
   if (A.way_clip && P.minx <= P.maxx)
   {
      P.miny = max(P.miny, clip0 * 256); P.maxy = min(P.maxy, clip1 * 256 - 1);
      if (P.miny > P.maxy) { P.minx = INT32_MAX; P.maxx = INT32_MIN; }
   }
   bool valid = P.minx <= P.maxx;
   if (valid)
   {
      int c0 = P.minx >> 8, c1 = P.maxx >> 8, r0 = P.miny >> 8, r1 = P.maxy >> 8;
      /* a member of a compound path (directional rule) left or right of the chart draws nothing, but its vertices still decide
       * which rows of the ONE polygon libcairo sub-samples: it keeps its row mask and its items (which no tile takes) */
      const bool off_x = c1 < 0 || c0 >= A.W, off_y = r1 < clip0 || r0 >= clip1;
      if (off_y || (off_x && A.compound == 0xffffffffu)) valid = false;
   }
   /* a primitive off the chart needs no flag in its items: their boxes are off the chart too */
   if (!valid) { if (lane == 0) A.prims[p] = P; return; }
   P.flags = item_flags;
   P.row0 = max(clip0, P.miny >> 8);
   P.nrows = min(clip1 - 1, P.maxy >> 8) - P.row0 + 1;
   const uint32_t nwords = ((uint32_t) P.nrows + 31u) >> 5;
   /* fill chunks are binned up to the primitive's right edge: edges left of a tile carry winding into it */
   if (!pieces && nitems > 1)
      for (int64_t i = lane; i < nitems; i += 64) A.items[item0 + i].maxx = bin_maxx;

   /* ---- pass B: row mask ---- */
   for (uint32_t i = lane; i < nwords; i += 64) lmask[i] = 0;
   if (lane == 0) L.nlong = 0;
   SMR_WAVE_SYNC();
   /* cuts at the page's sides: for edges that are libcairo's own -- fills, strokes as polygons, and the pieces of a DASHED stroke
    * (its other stroker adds every dash, join and cap as a polygon of its own); not the pieces of a solid stroke */
   const bool dashed = role == 0 ? pob != nullptr && !A.solid_border : role == 1 ? A.po_dots != nullptr : (wf & 8) && A.po_open != nullptr && !A.solid_open;
   const bool cuts = A.lim_on && (!pieces || dashed) && (P.minx < lim_l || P.maxx > lim_r);
   if (nitems == 1)
   {
      if ((uint32_t) lane < P.edge_cnt) { smr_setup_edge(smr_orient(e0), P.row0, P.nrows, lmask, L); if (cuts) smr_setup_cuts(e0, lim_l, lim_r, P.row0, P.nrows, lmask); }
   }
   else
   {
      for (uint32_t idx = lane; idx < P.edge_cnt; idx += 64)
      {
         const smr_edge_t e = E[idx];
         smr_setup_edge(smr_orient(e), P.row0, P.nrows, lmask, L);
         if (cuts) smr_setup_cuts(e, lim_l, lim_r, P.row0, P.nrows, lmask);
      }
   }
   if (!sparse)
   {
      /* a fill polygon: the two sides of a vertex at a row's first sample (smr_tie_check) */
      const bool lim = A.lim_on != 0;
      for (uint32_t idx = lane; idx < P.edge_cnt; idx += 64)
         smr_tie_check(E, P.edge_cnt, idx, nitems == 1 ? e0 : E[idx], P.row0, P.nrows, lmask, lim ? lim_l : INT32_MIN, lim ? lim_r : INT32_MAX);
   }
   SMR_WAVE_SYNC();
   const uint32_t nlong = L.nlong;
   smr_rowscan_t RS;
   RS.row0 = P.row0; RS.nrows = P.nrows; RS.edge_cnt = P.edge_cnt; RS.E = E; RS.L = &L; RS.lmask = lmask; RS.lane = lane;
   RS.status = &A.small[2]; RS.list_n = nlong; RS.mbase = 0;
   { const bool lim = A.lim_on && (!pieces || dashed); RS.cl_lo = lim ? lim_l : INT32_MIN; RS.cl_hi = lim ? lim_r : INT32_MAX; }
   RS.items = nitems > 64 ? A.items + item0 : nullptr;
   RS.cap = pieces ? A.scancap : SMR_ROWCAP;
#ifdef SMR_KO_SETUP
   if (SMR_KO_SETUP & 1) { }
   else
#endif
   if (nlong > SMR_LONGCAP)
   {
      for (int rb = 0; rb < P.nrows; rb += 64) if (RS.build_list(rb)) { RS.window(rb); SMR_WAVE_SYNC(); }
   }
   else if (nlong > 0)
   {
      for (int rb = 0; rb < P.nrows; rb += 64) RS.window(rb);
   }
   SMR_WAVE_SYNC();
   /* mask words come from a block this wavefront owns: one global atomic per SMR_MASK_BLOCK words
    * instead of one per primitive (2 M atomics on one address would take longer than the set-up) */
   uint32_t moff;
   if (nwords > arena_left)
   {
      const uint32_t take = nwords > SMR_MASK_BLOCK / 4 ? nwords : SMR_MASK_BLOCK;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&A.small[4], take);
      base = __shfl(base, 0);
      if (take == nwords) moff = base;                  /* a tall primitive gets its words directly */
      else { arena_pos = base; arena_left = take; moff = arena_pos; arena_pos += nwords; arena_left -= nwords; }
   }
   else { moff = arena_pos; arena_pos += nwords; arena_left -= nwords; }
   if ((uint64_t) moff + nwords > A.mask_cap)
   {
      /* the host sees the demand in small[4], enlarges the arena and runs the set-up again */
      if (lane == 0) { atomicOr(&A.small[2], SMR_ST_MASK_ARENA); P.flags = 0; A.prims[p] = P; }
      return;
   }
   for (uint32_t i = lane; i < nwords; i += 64) A.mask[moff + i] = lmask[i];
   P.mask_off = moff;
   first_valid = min(first_valid, A.compound != 0xffffffffu ? A.compound : (uint32_t) p);
   if (A.win_first && lane == 0) atomicMin(&A.win_first[clip0 / A.win_pitch], A.compound != 0xffffffffu ? A.compound : (uint32_t) p);
   /* what k_cover will need for this primitive at most: one alpha word per column of its box and strip of four rows */
   if (A.compound == 0xffffffffu && P.edge_cnt <= SMR_CV_MAXSLOTS)
   {
      const int wp = (P.maxx >> 8) - (P.minx >> 8) + 3;
      const int dr0 = max(P.row0, A.band_y0), dr1 = min(P.row0 + P.nrows, A.band_y0 + A.band_h);
      if (wp <= SMR_CV_WC && dr1 > dr0)
      {
         const int first = A.band_y0 + ((dr0 - A.band_y0) & ~(SMR_TH - 1));
         const uint32_t nstrips = (uint32_t) ((dr1 - 1 - first) >> 2) + 1u;
         need_alpha += (unsigned long long) nstrips * (uint32_t) wp; need_strips += nstrips;
         if (A.cv_split && lane == 0) P.drow_off = atomicAdd(&A.small[15], nstrips);
      }
   }
   if (lane == 0) A.prims[p] = P;
}

#ifndef SMR_SETUP_WAVES
#define SMR_SETUP_WAVES 2       /* 21 KB of LDS per workgroup: 14 wavefronts per CU where 4 per workgroup left room for 12 */
#endif
#define SMR_SETUP_CHUNK 16      /* most primitives a wavefront takes from its XCD's queue with one atomic */

/* persistent wavefronts, each sets up primitive gwave, gwave + nwaves, ... (2 M one-wave workgroups
 * would be bound by the workgroup dispatch rate, not by the work) */
__global__ __launch_bounds__(64 * SMR_SETUP_WAVES) void k_prim_setup(setup_args_t A, int mask_words, int chunk)
{
   extern __shared__ uint32_t lmask_all[];         /* [SMR_SETUP_WAVES][mask_words] */
   __shared__ setup_lds_t LL[SMR_SETUP_WAVES];
   const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      /* uniform: keeps the loop control in SGPRs */
   uint32_t arena_pos = 0, arena_left = 0, first_valid = 0xffffffffu, need_strips = 0;
   unsigned long long need_alpha = 0;
   /* XCD x takes the chunks x, x + 8, ... of `chunk` consecutive primitives (workgroups b and b+8 share an XCD), one chunk per
    * atomic: borders cost several times what fills do and a contour line thousands of times what an island does, so a static
    * share leaves most wavefronts waiting for the slowest */
   const int64_t nprim = (int64_t) A.K * A.n_ways;
   const int64_t xcd = blockIdx.x & 7u;
   uint32_t *const ctr = A.small + 128 + 8 * xcd;
   for (;;)
   {
      uint32_t k = 0;
      if (lane == 0) k = atomicAdd(ctr, 1u);
      const int64_t p0 = ((int64_t) (uint32_t) __builtin_amdgcn_readfirstlane(k) * 8 + xcd) * chunk;
      if (p0 >= nprim) break;
      const int64_t p1 = min(p0 + chunk, nprim);
      for (int64_t p = p0; p < p1; p++)
      {
         smr_prim_setup_one(A, p, lane, LL[wv], lmask_all + (size_t) wv * mask_words, arena_pos, arena_left, first_valid, need_alpha, need_strips);
         SMR_WAVE_SYNC();
      }
   }
   if (lane == 0 && first_valid != 0xffffffffu) atomicMin(&A.small[1], first_valid);
   if (lane == 0 && need_strips) { atomicAdd(&A.small[11], need_strips); atomicAdd((unsigned long long *) (A.small + 12), need_alpha); }
}

/* ---- long primitives, step 1: items and boxes.  The items of all long primitives are dealt round robin to the wavefronts
 * (item k of the b-th primitive to wavefront (items before + k) mod nwaves); the primitive's box grows by atomics ---- */
__global__ __launch_bounds__(256) void k_big_items(const smr_prim_t *__restrict__ prims, const smr_edge_t *__restrict__ edges, smr_item_t *items,
                                                   const uint32_t *__restrict__ biglist, smr_bigacc_t *acc)
{
   const int lane = threadIdx.x & 63;
   const uint32_t gwave = blockIdx.x * 4u + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = gridDim.x * 4u;
   const uint32_t nbig = biglist[0];
   /* every wavefront takes one contiguous run of the items of all long primitives: it meets one or two primitives, whose boxes it
    * grows with one set of atomics each (dealt round robin, every item brought five atomics on its primitive's accumulator) */
   uint64_t total = 0;
   for (uint32_t b = 0; b < nbig; b++) total += (prims[biglist[1 + 2 * b]].edge_cnt + 63u) >> 6;
   const uint64_t per = (total + nwaves - 1) / nwaves, lo = (uint64_t) gwave * per, hi = min(total, lo + per);
   uint64_t first = 0;
   for (uint32_t b = 0; b < nbig && first < hi; b++)
   {
      const smr_prim_t P = prims[biglist[1 + 2 * b]];
      const uint32_t item0 = biglist[2 + 2 * b], nitems = (P.edge_cnt + 63u) >> 6;
      if (first + nitems <= lo) { first += nitems; continue; }
      const uint32_t item_flags = acc[b].item_flags, prim_id = acc[b].prim_id;
      const bool sparse = acc[b].sparse != 0;       /* as in smr_prim_setup_one: a fill polygon has a vertex in every slot, a stroke's empty slots are all zero */
      const smr_edge_t *E = edges + P.edge_off;
      int amnx = INT32_MAX, amny = INT32_MAX, amxx = INT32_MIN, amxy = INT32_MIN, abmx = INT32_MIN;
      const uint32_t k0 = lo > first ? (uint32_t) (lo - first) : 0u, k1 = (uint32_t) min((uint64_t) nitems, hi - first);
      for (uint32_t k = k0; k < k1; k++)
      {
         const uint32_t idx = k * 64u + (uint32_t) lane;
         smr_edge_t e; e.x1 = e.y1 = e.x2 = e.y2 = 0;
         if (idx < P.edge_cnt) e = E[idx];
         bool has = idx < P.edge_cnt && (e.x1 != e.x2 || e.y1 != e.y2 || !sparse);
         if (sparse && e.x1 == 0 && e.y1 == 0 && e.x2 == 0 && e.y2 == 0) has = false;
         int mnx = has ? min(e.x1, e.x2) : INT32_MAX, mxx = has ? max(e.x1, e.x2) : INT32_MIN;
         int mny = has ? min(e.y1, e.y2) : INT32_MAX, mxy = has ? max(e.y1, e.y2) : INT32_MIN;
         int ext = 0;                     /* as in smr_prim_setup_one: the reach of the exact full-row path beyond an edge's end points */
         if (has && mxy - mny >= 239) ext = (int) (((int64_t) (mxx - mnx) * 9) / (mxy - mny)) + 2;
         const int bmn = smr_wave_min(has ? mnx - ext : INT32_MAX), bmx = smr_wave_max(has ? mxx + ext : INT32_MIN);
         mnx = smr_wave_min(mnx); mny = smr_wave_min(mny); mxx = smr_wave_max(mxx); mxy = smr_wave_max(mxy);
         if (lane == 0)
         {
            smr_item_t it;
            it.edge_off = (uint32_t) (P.edge_off + (uint64_t) k * 64);
            it.prim = prim_id;
            it.cnt_flags = mnx <= mxx ? (min(64u, P.edge_cnt - k * 64u) | (item_flags << 8)) : 0;
            it.color = P.color;
            it.minx = bmn; it.miny = max(mny, acc[b].clip0 * 256); it.maxx = bmx; it.maxy = min(mxy, acc[b].clip1 * 256 - 1);
            if (it.miny > it.maxy) it.cnt_flags = 0;
            items[item0 + k] = it;
         }
         if (mnx <= mxx) { amnx = min(amnx, mnx); amny = min(amny, mny); amxx = max(amxx, mxx); amxy = max(amxy, mxy); abmx = max(abmx, bmx); }
      }
      if (lane == 0 && amnx <= amxx)
      {
         atomicMin(&acc[b].minx, amnx); atomicMin(&acc[b].miny, amny); atomicMax(&acc[b].maxx, amxx); atomicMax(&acc[b].maxy, amxy);
         atomicMax(&acc[b].bin_maxx, abmx);
      }
      first += nitems;
   }
}

/* ---- step 2, one lane per long primitive: chart test, rows, mask words (zeroed here, ORed into by step 3) ---- */
__global__ void k_big_finish(smr_prim_t *prims, const uint32_t *__restrict__ biglist, smr_bigacc_t *acc, uint32_t *mask, uint32_t mask_cap,
                             uint32_t *small, int W, int H, uint32_t *win_first, int win_pitch)
{
   const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
   if (b >= biglist[0]) return;
   const uint32_t p = biglist[1 + 2 * b];
   smr_prim_t P = prims[p];
   const smr_bigacc_t a = acc[b];
   P.minx = a.minx; P.miny = a.miny; P.maxx = a.maxx; P.maxy = a.maxy;
   bool valid = P.minx <= P.maxx;
   if (valid)
   {
      P.miny = max(P.miny, a.clip0 * 256); P.maxy = min(P.maxy, a.clip1 * 256 - 1);
      const int c0 = P.minx >> 8, c1 = P.maxx >> 8, r0 = P.miny >> 8, r1 = P.maxy >> 8;
      if (c1 < 0 || P.miny > P.maxy || r1 < a.clip0 || c0 >= W || r0 >= a.clip1) valid = false;
   }
   if (!valid) { prims[p] = P; return; }
   P.row0 = max(a.clip0, P.miny >> 8);
   P.nrows = min(a.clip1 - 1, P.maxy >> 8) - P.row0 + 1;
   const uint32_t nwords = ((uint32_t) P.nrows + 31u) >> 5;
   const uint32_t moff = atomicAdd(&small[4], nwords);
   if ((uint64_t) moff + nwords > mask_cap) { atomicOr(&small[2], SMR_ST_MASK_ARENA); P.nrows = 0; prims[p] = P; return; }
   for (uint32_t i = 0; i < nwords; i++) mask[moff + i] = 0;
   P.mask_off = moff; P.flags = a.item_flags;
   prims[p] = P;
   acc[b].valid = 1;
   atomicMin(&small[1], a.prim_id);
   if (win_first) atomicMin(&win_first[a.clip0 / win_pitch], a.prim_id);
}

/* ---- step 3: the rows in which an edge starts or ends (smr_setup_edge, into the arena), and the binning box of a fill's
 * chunks (up to the primitive's right edge) ---- */
__global__ __launch_bounds__(256) void k_big_vrows(const smr_prim_t *__restrict__ prims, const smr_edge_t *__restrict__ edges, smr_item_t *items,
                                                   const uint32_t *__restrict__ biglist, const smr_bigacc_t *__restrict__ acc, uint32_t *mask)
{
   const int lane = threadIdx.x & 63;
   const uint32_t gwave = blockIdx.x * 4u + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = gridDim.x * 4u;
   const uint32_t nbig = biglist[0];
   uint64_t total = 0;                      /* contiguous runs of items per wavefront, as in k_big_items */
   for (uint32_t b = 0; b < nbig; b++) total += (prims[biglist[1 + 2 * b]].edge_cnt + 63u) >> 6;
   const uint64_t per = (total + nwaves - 1) / nwaves, lo = (uint64_t) gwave * per, hi = min(total, lo + per);
   uint64_t first = 0;
   for (uint32_t b = 0; b < nbig && first < hi; b++)
   {
      const smr_prim_t P = prims[biglist[1 + 2 * b]];
      const uint32_t item0 = biglist[2 + 2 * b], nitems = (P.edge_cnt + 63u) >> 6;
      if (first + nitems <= lo) { first += nitems; continue; }
      const smr_bigacc_t a = acc[b];
      if (a.valid)
      {
         const smr_edge_t *E = edges + P.edge_off;
         uint32_t *m = mask + P.mask_off;
         const uint32_t k0 = lo > first ? (uint32_t) (lo - first) : 0u, k1 = (uint32_t) min((uint64_t) nitems, hi - first);
         for (uint32_t k = k0; k < k1; k++)
         {
            const uint32_t idx = k * 64u + (uint32_t) lane;
            if (idx < P.edge_cnt)
            {
               const smr_edge_t e = E[idx];
               const smr_oedge_t o = smr_orient(e);
               if (a.lim_on) smr_setup_cuts(e, a.lim_l, a.lim_r, P.row0, P.nrows, m);
               if (!a.sparse) smr_tie_check(E, P.edge_cnt, idx, e, P.row0, P.nrows, m, a.lim_on ? a.lim_l : INT32_MIN, a.lim_on ? a.lim_r : INT32_MAX);
               if (o.dir != 0)
               {
                  if (o.ytop % SMR_GRID_Y != 0)
                  {
                     const int r = o.ytop / SMR_GRID_Y - P.row0;
                     if (o.ytop >= 0 && r >= 0 && r < P.nrows) atomicOr(&m[r >> 5], 1u << (r & 31));
                  }
                  if (o.ybot % SMR_GRID_Y != 0)
                  {
                     const int r = (o.ybot > 0 ? (o.ybot - 1) / SMR_GRID_Y : -1) - P.row0;
                     if (o.ybot > 0 && r >= 0 && r < P.nrows) atomicOr(&m[r >> 5], 1u << (r & 31));
                  }
               }
            }
            if (lane == 0 && !(a.item_flags & SMR_PF_PIECES)) items[item0 + k].maxx = a.bin_maxx;
         }
      }
      first += nitems;
   }
}

/* long primitives: the windows of 64 rows of every primitive in the list are dealt round robin to persistent wavefronts;
 * each reads the window's two mask words (vertex rows, written by k_big_vrows), builds the window's list of spanning
 * edges with one pass over the primitive's slots and ORs the rows found crossing back into the arena */
__global__ __launch_bounds__(64 * SMR_SETUP_WAVES) void k_big_rows(const smr_prim_t *__restrict__ prims, const smr_edge_t *__restrict__ edges,
                                                                   const smr_item_t *__restrict__ items, const uint32_t *__restrict__ biglist, uint32_t *mask, uint32_t *status,
                                                                   uint32_t scancap, const smr_bigacc_t *__restrict__ acc)
{
   __shared__ setup_lds_t LL[SMR_SETUP_WAVES];
   __shared__ uint32_t wmask[SMR_SETUP_WAVES][2];
   const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      /* uniform: keeps the loop control in SGPRs */
   const uint32_t gwave = blockIdx.x * SMR_SETUP_WAVES + wv, nwaves = gridDim.x * SMR_SETUP_WAVES;
   const uint32_t nbig = biglist[0];
   uint32_t first = 0;                                  /* windows of the primitives before this one */
   for (uint32_t b = 0; b < nbig; b++)
   {
      const smr_prim_t P = prims[biglist[1 + 2 * b]];
      const uint32_t nwin = ((uint32_t) P.nrows + 63u) >> 6;
      /* window w of this primitive belongs to wavefront (first + w) mod nwaves */
      uint32_t w = (gwave + nwaves - first % nwaves) % nwaves;
      for (; w < nwin; w += nwaves)
      {
         const int rb = (int) (w * 64);
         const uint32_t nw = min(2u, (((uint32_t) P.nrows + 31u) >> 5) - 2u * w);
         if (lane < 2) wmask[wv][lane] = (uint32_t) lane < nw ? mask[P.mask_off + 2u * w + lane] : 0xffffffffu;
         SMR_WAVE_SYNC();
         smr_rowscan_t RS;
         RS.row0 = P.row0; RS.nrows = P.nrows; RS.edge_cnt = P.edge_cnt; RS.E = edges + P.edge_off; RS.L = &LL[wv];
         RS.lmask = &wmask[wv][0]; RS.mbase = rb;
         RS.lane = lane; RS.status = status; RS.list_n = 0; RS.items = items + biglist[2 + 2 * b];
         RS.cap = (P.flags & SMR_PF_PIECES) ? scancap : SMR_ROWCAP;
         RS.cl_lo = acc[b].lim_on ? acc[b].lim_l : INT32_MIN; RS.cl_hi = acc[b].lim_on ? acc[b].lim_r : INT32_MAX;
         const uint32_t before0 = wmask[wv][0], before1 = wmask[wv][1];
         if (RS.build_list(rb)) RS.window(rb);
         SMR_WAVE_SYNC();
         if (lane < (int) nw)
         {
            const uint32_t now = wmask[wv][lane], was = lane == 0 ? before0 : before1;
            if (now != was) atomicOr(&mask[P.mask_off + 2u * w + lane], now);
         }
         SMR_WAVE_SYNC();
      }
      first += nwin;
   }
}

/* ------------------------------------------------------------------------------------ */
/* directional rules: all closed ways of the rule are ONE path with ONE non-zero fill
 * (act_draw_fini, src/smrules_cairo.c:1084-1140).  Every way was set up like a fill of its
 * own (items, box, row mask over ITS edges); the compound primitive is their union:
 * box = union, row mask = OR of the members' masks, and rows in which no member has a vertex
 * are decided over the edges of ALL members.                                              */
/* ------------------------------------------------------------------------------------ */
__global__ void k_compound_merge(smr_prim_t *prims, int64_t n_ways, uint32_t compound, uint32_t *mask, uint32_t *covered)
{
   int64_t w = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (w >= n_ways) return;
   const smr_prim_t P = prims[2 * w + 1];
   if (!(P.flags & SMR_PF_VALID)) return;
   smr_prim_t *C = &prims[compound];
   atomicMin(&C->minx, P.minx); atomicMin(&C->miny, P.miny); atomicMax(&C->maxx, P.maxx); atomicMax(&C->maxy, P.maxy);
   const uint32_t nwords = ((uint32_t) P.nrows + 31u) >> 5;
   const uint32_t sh = (uint32_t) P.row0 & 31u, base = (uint32_t) P.row0 >> 5;
   for (uint32_t i = 0; i < nwords; i++)
   {
      const uint32_t m = mask[P.mask_off + i];
      const uint32_t nb = min(32u, (uint32_t) P.nrows - 32u * i);
      const uint32_t cv = nb == 32u ? 0xffffffffu : ((1u << nb) - 1u);
      if (m << sh) atomicOr(&mask[base + i], m << sh);
      atomicOr(&covered[base + i], cv << sh);
      if (sh)
      {
         if (m >> (32u - sh)) atomicOr(&mask[base + i + 1], m >> (32u - sh));
         if (cv >> (32u - sh)) atomicOr(&covered[base + i + 1], cv >> (32u - sh));
      }
   }
}

/* one wavefront per chart row that some member covers and in which no member has a vertex: collect the
 * edges of all members spanning the row, compare every pair (order at the top = order at the bottom?).
 * More than SMR_ROWCAP such edges: the row is sub-sampled instead (status SMR_ST_DEGRADED: the one place
 * where the result may leave libcairo's arithmetic, by the difference between 15 samples and the exact
 * trapezoid on that row's edge pixels). */
__global__ __launch_bounds__(64) void k_compound_rows(const smr_edge_t *__restrict__ edges, const smr_item_t *__restrict__ items, int64_t n_items,
                                                      uint32_t compound, uint32_t *mask, const uint32_t *__restrict__ covered, int H, uint32_t *status,
                                                      int lim_on, int32_t lim_l, int32_t lim_r, const int4 *__restrict__ win_lim, int win_pitch)
{
   __shared__ int f0[SMR_ROWCAP], f1[SMR_ROWCAP];
   const int r = blockIdx.x, lane = threadIdx.x;
   if (r >= H) return;
   if (!((covered[r >> 5] >> (r & 31)) & 1u) || ((mask[r >> 5] >> (r & 31)) & 1u)) return;
   const int rs0 = r * SMR_GRID_Y, rs1 = rs0 + SMR_GRID_Y;
   uint32_t cnt = 0;
   /* (what lies beyond a side of libcairo's limits is projected onto that side before its scan converter looks: smr_rowscan_t;
    * in an atlas the row's window has the limits) */
   int32_t cl_lo = lim_on ? lim_l : INT32_MIN, cl_hi = lim_on ? lim_r : INT32_MAX;
   if (lim_on && win_lim) { const int4 wl = win_lim[r / win_pitch]; cl_lo = wl.x; cl_hi = wl.y; }
   /* lanes = 64 items at a time: only the members' items whose box meets the row hold a candidate */
   for (int64_t kb = 0; kb < n_items; kb += 64)
   {
      bool ov = false;
      if (kb + lane < n_items)
      {
         const smr_item_t *it = items + kb + lane;
         ov = it->cnt_flags != 0 && it->prim == compound && (it->maxy >> 8) >= r && (it->miny >> 8) <= r;
      }
      unsigned long long om = __ballot(ov);
      while (om)
      {
         const int j = __ffsll((long long) om) - 1;
         om &= om - 1;
         const uint32_t eo = items[kb + j].edge_off, ec_ = items[kb + j].cnt_flags & 0xffu;
         smr_oedge_t o; o.dir = 0; o.ytop = o.ybot = 0;
         if ((uint32_t) lane < ec_) o = smr_orient(edges[(size_t) eo + lane]);
         const bool act = o.dir != 0 && o.ytop <= rs0 && o.ybot >= rs1;
         const unsigned long long am = __ballot(act);
         if (!am) continue;
         const uint32_t pos = cnt + __popcll(am & ((1ull << lane) - 1ull));
         if (act && pos < SMR_ROWCAP)
         {
            const int64_t ec = 256 - 30ll * o.p1y;
            const double inv = smr_rcp((double) o.D);
            int c0 = smr_edge_cell(ec, 2 * o.dxs, o.D, inv, o.p1x, rs0), c1 = smr_edge_cell(ec, 2 * o.dxs, o.D, inv, o.p1x, rs1);
            if (smr_beyond(c0, cl_lo, cl_hi)) smr_project(c0, c1, smr_edge_cell(ec, 2 * o.dxs, o.D, inv, o.p1x, rs1 - 1), cl_lo, cl_hi);
            f0[pos] = c0; f1[pos] = c1;
         }
         cnt += __popcll(am);
      }
   }
   __syncthreads();
   bool viol = false;
   if (cnt > SMR_ROWCAP) { viol = true; if (lane == 0) atomicOr(status, SMR_ST_DEGRADED); }
   else if ((uint32_t) lane < cnt)
   {
      const int a0 = f0[lane], a1 = f1[lane];
      for (uint32_t j = 0; j < cnt; j++) if (a0 < f0[j] && a1 > f1[j]) viol = true;
   }
   if (__ballot(viol) && lane == 0) atomicOr(&mask[r >> 5], 1u << (r & 31));
}

__global__ void k_compound_init(smr_prim_t *prims, uint32_t compound, uint64_t edge_off, uint32_t edge_cnt, uint32_t color, int H, int evenodd)
{
   smr_prim_t P;
   P.edge_off = edge_off; P.edge_cnt = edge_cnt; P.color = color;
   P.minx = P.miny = INT32_MAX; P.maxx = P.maxy = INT32_MIN;
   P.flags = SMR_PF_VALID | SMR_PF_MULTI | (evenodd ? SMR_PF_EVENODD : 0u); P.mask_off = 0; P.row0 = 0; P.nrows = H; P.drow_off = 0; P.pad_[0] = P.pad_[1] = P.pad_[2] = 0;
   prims[compound] = P;
}

/* ------------------------------------------------------------------------------------ */
/* binning: item box -> 64x4 tiles of this band                                            */
/* ------------------------------------------------------------------------------------ */
__device__ __forceinline__ bool smr_item_tiles(const smr_item_t &p, int W, int band_y0, int band_h,
                                               int &tx0, int &tx1, int &ty0, int &ty1)
{
   if (!((p.cnt_flags >> 8) & SMR_PF_VALID)) return false;
   int c0 = p.minx >> 8, c1 = p.maxx >> 8, r0 = (p.miny >> 8) - band_y0, r1 = (p.maxy >> 8) - band_y0;
   if (c1 < 0 || r1 < 0 || c0 >= W || r0 >= band_h) return false;
   if (c0 < 0) c0 = 0;
   if (r0 < 0) r0 = 0;
   if (c1 >= W) c1 = W - 1;
   if (r1 >= band_h) r1 = band_h - 1;
   tx0 = c0 / SMR_TW; tx1 = c1 / SMR_TW; ty0 = r0 / SMR_TH; ty1 = r1 / SMR_TH;
   return true;
}

template <bool SCATTER>
__global__ void k_bin(const smr_item_t *__restrict__ items, int64_t n_items, int W, int band_y0, int band_h, int TX,
                      uint32_t *__restrict__ counts, const uint32_t *__restrict__ tile_off, uint32_t *__restrict__ entries)
{
   int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n_items) return;
   smr_item_t p = items[i];
   int tx0, tx1, ty0, ty1;
   if (!smr_item_tiles(p, W, band_y0, band_h, tx0, tx1, ty0, ty1)) return;
   for (int ty = ty0; ty <= ty1; ty++)
      for (int tx = tx0; tx <= tx1; tx++)
      {
         uint32_t t = (uint32_t) ty * TX + tx;
         uint32_t pos = atomicAdd(&counts[t], 1u);
         if (SCATTER) entries[tile_off[t] + pos] = (uint32_t) i;
      }
}

/* exclusive scan of n uint32 (three small kernels; n up to a few 10 M) */
#define SCAN_B 1024
__global__ __launch_bounds__(256) void k_scan_block(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                    uint32_t *__restrict__ sums, uint32_t n, uint32_t *maxv)
{
   __shared__ uint32_t s[256];
   uint32_t base = blockIdx.x * SCAN_B, t = threadIdx.x;
   uint32_t v[4], tot = 0;
#pragma unroll
   for (int k = 0; k < 4; k++) { uint32_t i = base + t * 4 + k; v[k] = i < n ? in[i] : 0; tot += v[k]; }
   if (maxv)                     /* the largest element too (entries of the fullest tile: k_tile_sort's LDS) */
   {
      const int m = smr_wave_max((int) max(max(v[0], v[1]), max(v[2], v[3])));
      if ((t & 63u) == 0 && m > 0) atomicMax(maxv, (uint32_t) m);
   }
   s[t] = tot;
   __syncthreads();
   for (int d = 1; d < 256; d <<= 1)
   {
      uint32_t x = t >= (uint32_t) d ? s[t - d] : 0;
      __syncthreads();
      s[t] += x;
      __syncthreads();
   }
   uint32_t excl = s[t] - tot;
#pragma unroll
   for (int k = 0; k < 4; k++) { uint32_t i = base + t * 4 + k; if (i < n) out[i] = excl; excl += v[k]; }
   if (t == 255) sums[blockIdx.x] = s[255];
}

__global__ __launch_bounds__(1024) void k_scan_sums(uint32_t *__restrict__ sums, uint32_t nb, uint32_t *__restrict__ total)
{
   /* single block, serial over chunks of its size (256 threads in a draw pass: four wavefronts find room on a CU beside
    * k_cover's, sixteen do not, and the whole binning waited for k_cover's last wavefront) */
   __shared__ uint32_t s[1024];
   __shared__ uint32_t carry;
   const uint32_t t = threadIdx.x, B = blockDim.x;
   if (t == 0) carry = 0;
   __syncthreads();
   for (uint32_t base = 0; base < nb; base += B)
   {
      uint32_t i = base + t, v = i < nb ? sums[i] : 0;
      s[t] = v;
      __syncthreads();
      for (uint32_t d = 1; d < B; d <<= 1)
      {
         uint32_t x = t >= d ? s[t - d] : 0;
         __syncthreads();
         s[t] += x;
         __syncthreads();
      }
      if (i < nb) sums[i] = carry + s[t] - v;
      __syncthreads();
      if (t == B - 1) carry += s[B - 1];
      __syncthreads();
   }
   if (t == 0) *total = carry;
}

__global__ __launch_bounds__(256) void k_scan_add(uint32_t *__restrict__ out, const uint32_t *__restrict__ sums, uint32_t n)
{
   uint32_t base = blockIdx.x * SCAN_B, t = threadIdx.x, add = sums[blockIdx.x];
#pragma unroll
   for (int k = 0; k < 4; k++) { uint32_t i = base + t * 4 + k; if (i < n) out[i] += add; }
   if (blockIdx.x == gridDim.x - 1 && t == 0) out[n] = 0;   /* overwritten by the host-side total below */
}

__global__ void k_set_u32(uint32_t *p, const uint32_t *v) { *p = *v; }


/* per-tile sort of the scattered item indices (ascending = drawing order), then the tile's
 * work stream: one smr_thdr_t per entry with the item header, the tile's bits of the
 * primitive's row mask and the primitive's pixel extents relative to the tile */
#define SMR_TILE_PURE 0x80000000u      /* tile_cnt: the tile's stream holds alpha words only */
#define SORT_LDS 4096        /* most entries of a tile sorted in LDS; beyond, the same network runs in global memory */
struct sort_args_t
{
   const uint32_t *tile_off;
   uint32_t *entries;
   const smr_item_t *items;
   const smr_prim_t *prims;
   const uint32_t *mask;
   const uint32_t *first_prim;
   const uint32_t *win_first; int win_pitch;      /* an atlas of windows: the first primitive per window (setup_args_t) */
   smr_thdr_t *hdrs;
   uint32_t n_tiles;
   uint32_t lds_cap;          /* entries of one tile that fit its wavefront's share of LDS (a power of two up to SORT_LDS) */
   int TX, band_y0, band_h;
   const uint2 *drows;        /* k_cover's strip table (primitives with SMR_PF_DELTA) */
   uint32_t *tile_cnt;        /* [n_tiles] entries of the tile's work stream (entries that bring nothing are left out) */
   int vec;                   /* a sparse rule: every wavefront looks at 64 tiles at once and sorts the few that hold anything */
   uint32_t pure_bit;         /* SMR_TILE_PURE, or 0 when every tile goes to k_raster: ORed into tile_cnt of a tile whose stream is k_cover's alpha words only */
   uint32_t *pure_list, *pure_n;   /* a sparse rule (vec): the tiles that got the bit, for k_blend to go by instead of looking at every tile of the chart */
};

/* SORT_WAVES independent wavefronts per workgroup, one tile each (a chart of 4 M tiles is 4 M one-wave workgroups otherwise,
 * and their dispatch, not the sorting, takes the time); no workgroup barrier: a wavefront's LDS traffic is in program order */
#ifndef SORT_WAVES
#define SORT_WAVES 4
#endif
__global__ __launch_bounds__(64 * SORT_WAVES) void k_tile_sort(sort_args_t A)
{
   extern __shared__ uint32_t sall[];              /* [SORT_WAVES][A.lds_cap]: sized for the fullest tile of the launch */
   const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   uint32_t *s = sall + (size_t) wv * A.lds_cap;
   const uint32_t wid = smr_xcd_remap(blockIdx.x, gridDim.x) * SORT_WAVES + wv;       /* an XCD's workgroups sort neighbouring tiles */
   const uint32_t lane = threadIdx.x & 63u;
   const uint32_t t0 = A.vec ? wid * 64u : wid;
   if (t0 >= A.n_tiles) return;
   unsigned long long todo = 1ull;
   if (A.vec)
   {
      const uint32_t tl = t0 + lane;
      const uint32_t nl = tl < A.n_tiles ? A.tile_off[tl + 1] - A.tile_off[tl] : 0u;
      if (tl < A.n_tiles && nl == 0u) A.tile_cnt[tl] = 0u;
      todo = __ballot(nl != 0u);
   }
   while (todo)
   {
   const uint32_t tile = t0 + (uint32_t) __ffsll((long long) todo) - 1u;
   todo &= todo - 1ull;
   const uint32_t beg = A.tile_off[tile], n = A.tile_off[tile + 1] - beg;
   if (n == 0) { if (lane == 0) A.tile_cnt[tile] = 0; continue; }
   uint32_t N = 2;
   while (N < n) N <<= 1;
   const bool in_lds = N <= A.lds_cap;
   if (n >= 2 && in_lds)
   {
      for (uint32_t i = lane; i < N; i += 64) s[i] = i < n ? A.entries[beg + i] : 0xffffffffu;
      SMR_WAVE_SYNC();
      for (uint32_t k = 2; k <= N; k <<= 1)
      {
         for (uint32_t j = k >> 1; j > 0; j >>= 1)
         {
            for (uint32_t i = lane; i < N; i += 64)
            {
               uint32_t p = (j == (k >> 1)) ? (i ^ (k - 1)) : (i ^ j);
               if (p > i)
               {
                  uint32_t a = s[i], b = s[p];
                  if (a > b) { s[i] = b; s[p] = a; }
               }
            }
            SMR_WAVE_SYNC();
         }
      }
   }
   else if (n >= 2)
   {
      /* rare: more items on one 64x4 tile than fit in LDS -- same network in global
       * memory through L1-bypassing (agent-scope) accesses */
      uint32_t *e = A.entries + beg;
      for (uint32_t k = 2; k <= N; k <<= 1)
      {
         for (uint32_t j = k >> 1; j > 0; j >>= 1)
         {
            for (uint32_t i = lane; i < N; i += 64)
            {
               uint32_t p = (j == (k >> 1)) ? (i ^ (k - 1)) : (i ^ j);
               if (p > i && p < n)
               {
                  uint32_t a = __hip_atomic_load(&e[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  uint32_t b = __hip_atomic_load(&e[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  if (a > b)
                  {
                     __hip_atomic_store(&e[i], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                     __hip_atomic_store(&e[p], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  }
               }
            }
            __builtin_amdgcn_s_waitcnt(0);
            SMR_WAVE_SYNC();
         }
      }
   }
   /* the work stream */
   const int tx = tile % A.TX, ty = tile / A.TX;
   const int row0 = A.band_y0 + ty * SMR_TH, col0 = tx * SMR_TW;
   const uint32_t first_prim = A.win_first ? A.win_first[row0 / A.win_pitch] : *A.first_prim;
   const unsigned long long lt_mask = (1ull << lane) - 1ull;
   uint32_t kept = 0, last_prim = 0xffffffffu;
   bool other = false;         /* an entry that is not an alpha word: the tile is k_raster's */
   for (uint32_t i0 = 0; i0 < n; i0 += 64)
   {
      const uint32_t i = i0 + lane;
      const bool valid = i < n;
      uint32_t idx = 0;
      if (!valid) idx = 0;
      else if (n < 2) idx = A.entries[beg + i];
      else if (in_lds) idx = s[i];
      else idx = __hip_atomic_load(&A.entries[beg + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      smr_item_t it; it.edge_off = 0; it.prim = 0xfffffffeu; it.cnt_flags = 0; it.color = 0; it.minx = it.miny = it.maxx = it.maxy = 0;
      smr_prim_t P; P.flags = 0; P.row0 = 0; P.nrows = 0; P.mask_off = 0; P.minx = P.miny = P.maxx = P.maxy = 0; P.drow_off = 0;
      if (valid) { it = A.items[idx]; P = A.prims[it.prim]; }
      smr_thdr_t h;
      h.edge_off = it.edge_off; h.prim = it.prim; h.color = it.color;
      h.cnt_flags = it.cnt_flags | (it.prim == first_prim ? 1u << 16 : 0u);
      uint32_t bits = 0;
#pragma unroll
      for (int r = 0; r < SMR_TH; r++)
      {
         int rr = row0 + r - P.row0;
         uint32_t bit = 1;      /* rows outside the mask hold no edge of the primitive */
         if (valid && rr >= 0 && rr < P.nrows) bit = (A.mask[P.mask_off + ((uint32_t) rr >> 5)] >> (rr & 31)) & 1u;
         bits |= bit << r;
         if (row0 + r >= (P.miny >> 8) && row0 + r <= (P.maxy >> 8)) bits |= 16u << r;
      }
      h.rowbits = bits;
      h.colmin = (P.minx >> 8) - col0; h.colmax = (P.maxx >> 8) - col0;
      h.dcounts = 0;
      bool keep = valid;
      /* the primitive of the entry before this one in the tile's order */
      uint32_t prev = __shfl_up(it.prim, 1);
      if (lane == 0) prev = last_prim;
      last_prim = __shfl(it.prim, 63);
      if (valid && (P.flags & SMR_PF_DELTA))
      {
         /* k_cover has the primitive's alpha: the entry names the words of the tile's strip (k_cover's strips start on tile rows).
          * One entry per primitive and tile does, and none where the strip has nothing for the tile's columns */
         const int dr0 = max(P.row0, A.band_y0), dr1 = min(P.row0 + P.nrows, A.band_y0 + A.band_h);
         const int first = A.band_y0 + ((dr0 - A.band_y0) & ~(SMR_TH - 1));
         const int sidx = (row0 - first) >> 2, nstrips = ((dr1 - 1 - first) >> 2) + 1;
         uint2 d = make_uint2(0u, 0u);
         if (row0 >= first && sidx < nstrips) d = A.drows[P.drow_off + (uint32_t) sidx];
         h.edge_off = d.x; h.dcounts = d.y; h.cnt_flags |= SMR_PF_DELTA << 8;
         const int x0 = smr_strip_x0(d.y), nw = (int) smr_strip_n(d.y);
         if (it.prim == prev || nw == 0 || x0 + nw <= col0 || x0 >= col0 + SMR_TW) keep = false;
      }
      const unsigned long long km = __ballot(keep);
      if (__ballot(keep && !(P.flags & SMR_PF_DELTA))) other = true;
      if (keep) A.hdrs[beg + kept + (uint32_t) __popcll(km & lt_mask)] = h;
      kept += (uint32_t) __popcll(km);
      if (valid && in_lds && n >= 2) A.entries[beg + i] = idx;
   }
   if (lane == 0)
   {
      const bool pure = kept && !other && A.pure_bit;
      A.tile_cnt[tile] = kept | (pure ? A.pure_bit : 0u);
      if (pure && A.pure_list) A.pure_list[atomicAdd(A.pure_n, 1u)] = tile;
   }
   SMR_WAVE_SYNC();            /* the next tile of this wavefront reuses the LDS block */
   }
}

/* ------------------------------------------------------------------------------------ */
/* the rasterizer: one wavefront per 64x4 tile                                             */
/* ------------------------------------------------------------------------------------ */
/* a pattern pass's source pixel at chart pixel (px, row) */
__device__ __forceinline__ uint32_t smr_pat_pix(const uint32_t *__restrict__ pat, int w, int h, int px, int row)
{
   int u = px % w, v = row % h;
   if (u < 0) u += w;
   if (v < 0) v += h;
   return pat[(size_t) v * w + u];
}

struct raster_args_t
{
   const uint32_t *blend_list, *blend_list_n;      /* k_blend, a sparse rule: the tiles to blend (k_tile_sort's list) */
   const smr_prim_t *prims;
   const smr_thdr_t *hdrs;    /* per-tile sorted work stream (k_tile_sort) */
   const smr_edge_t *edges;
   const uint32_t *tile_off;
   const uint32_t *tile_cnt;  /* entries of each tile's work stream (k_tile_sort leaves out what brings nothing) */
   uint32_t *canvas;          /* band pixels, stride W */
   uint32_t *status;
   uint32_t *tile_ctr;        /* [8 * xcd]: next tile of the XCD's columns, zero at launch */
   uint32_t *blend_ctr;       /* the same for k_blend */
   const uint32_t *pat;       /* a pattern pass (smr_hip_draw_pattern): every primitive's source is this premultiplied ARGB32 image, tiled from the
                                 chart's top-left pixel (CAIRO_EXTEND_REPEAT, one image pixel per chart pixel); NULL: the primitives' colours */
   int pat_w, pat_h;
   uint32_t strided;          /* k_raster<false>: no queue, wavefront i of an XCD looks at the tiles i, i + waves, ... of its columns (the pure ones are k_blend's) */
   int W, band_y0, band_h, TX;
   uint32_t n_tiles;
   uint32_t chunk;            /* tiles a wavefront takes from its XCD's queue with one atomic */
   int direct;                /* no group layer: primitives composite straight onto the canvas */
   /* tiles whose crossing lists overflow in the fast instantiation write nothing and queue here; the second instantiation
    * (k_raster<true>) renders them from lists in global memory that hold every edge slot binned to the tile */
   uint32_t *ovf_list;        /* [n_tiles] */
   uint32_t *ovf_cnt;         /* [0] tiles queued, [1] most entries of one of them */
   uint32_t ovf_n;            /* k_raster<true>: tiles in the list */
   const uint32_t *alpha;     /* k_cover's alpha words (entries with SMR_PF_DELTA) */
   uint32_t *big_bucket;      /* k_raster<true>: [wavefront][big_cap][64 lanes] */
   uint32_t big_cap;          /* entries per sub-row (a crossing takes one, an interval two) */
};

/* a sub-row's list of crossings (fills) or disjoint intervals (strokes): LDS in the fast instantiation, global memory (through
 * the L2: a wavefront reads back what it wrote) in the one for overflowed tiles */
template <bool BIG> struct smr_bucket_t;
template <> struct smr_bucket_t<false>
{
   uint16_t *p;
   __device__ __forceinline__ int get(int i) const { return p[i]; }
   __device__ __forceinline__ void set(int i, int v) const { p[i] = (uint16_t) v; }
};
template <> struct smr_bucket_t<true>
{
   uint32_t *p;               /* entry i of this lane at p[64 * i] */
   __device__ __forceinline__ int get(int i) const { return (int) __hip_atomic_load(&p[(size_t) i << 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
   __device__ __forceinline__ void set(int i, int v) const { __hip_atomic_store(&p[(size_t) i << 6], (uint32_t) v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};

struct raster_lds_t
{
   int cover[SMR_TH][SMR_TW];
   int uarea[SMR_TH][SMR_TW];
   int left[SMR_TH];
   uint32_t rowlist[SMR_TH][SMR_ROWCAP];
   int redge[SMR_TH][16][5];  /* p1x, p1y, dxs, D, dir of the first 16 edges spanning each full row */
   /* the crossing buckets are drained (spans of the sub-sampled rows) before the rank sort of the rarely taken slow
    * full-row path needs its keys: they share the storage */
   union
   {
      uint16_t bucket[SMR_NSUB][SMR_BKS];
      struct { unsigned long long key[64]; int perm[64]; } srt;
   };
};

__device__ __forceinline__ void smr_cell_add(raster_lds_t &L, int row, int ix, int dcover, int duarea)
{
   if (ix < 0) { atomicAdd(&L.left[row], dcover); return; }
   if (ix >= SMR_TW) return;
   if (dcover) atomicAdd(&L.cover[row][ix], dcover);
   if (duarea) atomicAdd(&L.uarea[row][ix], duarea);
}

/* libcairo tor: cell_list_render_edge -- exact coverage of one edge across a full pixel
 * row.  (q1,r1) = x at the first sub-row centre of the row, (q2,r2) = same one row below,
 * X0 = left edge of the tile in 24.8.  All remainders are in units of 1/D. */
__device__ void smr_render_edge(raster_lds_t &L, int row, int32_t X0, int32_t dxs, int32_t D,
                                int64_t q1, int32_t r1_, int64_t q2, int32_t r2_, int sign)
{
   int64_t r1 = r1_, r2 = r2_;
   if (dxs != 0)
   {
      /* step back from the sample location (half a sub-row) to the row boundary */
      int64_t n = (int64_t) dxs * 1024;
      int64_t dq = n / D, dr = n % D;          /* C truncation, like the original */
      q1 -= dq / 2; r1 -= dr / 2;
      if (r1 < 0) { --q1; r1 += D; } else if (r1 >= D) { ++q1; r1 -= D; }
      q2 -= dq / 2; r2 -= dr / 2;
      if (r2 < 0) { --q2; r2 += D; } else if (r2 >= D) { ++q2; r2 -= D; }
   }
   q1 -= X0; q2 -= X0;
   int64_t ix1 = q1 >> 8, ix2 = q2 >> 8;
   int fx1 = (int) (q1 & 255), fx2 = (int) (q2 & 255);
   if (ix1 == ix2)
   {
      int ix = ix1 < -1 ? -1 : ix1 > SMR_TW ? SMR_TW : (int) ix1;
      smr_cell_add(L, row, ix, sign * SMR_GRID_Y, sign * (fx1 + fx2) * SMR_GRID_Y);
      return;
   }
   if (ix2 < ix1)
   {
      int64_t t = ix1; ix1 = ix2; ix2 = t;
      int f = fx1; fx1 = fx2; fx2 = f;
      t = q1; q1 = q2; q2 = t;
      t = r1; r1 = r2; r2 = t;
   }
   int64_t dx = (q2 - q1) * D + (r2 - r1);
   int64_t tmp = (ix1 + 1) * 256 * (int64_t) D;
   tmp -= q1 * D + r1;
   tmp *= SMR_GRID_Y;
   int64_t yq = tmp / dx, yr = tmp % dx;
   {
      int ix = ix1 < -1 ? -1 : ix1 > SMR_TW ? SMR_TW : (int) ix1;
      smr_cell_add(L, row, ix, (int) (sign * yq), (int) (sign * yq * (256 + fx1)));
   }
   int64_t y_last = yq;
   if (ix1 + 1 < ix2)
   {
      int64_t fq = (int64_t) SMR_GRID_Y * 256 * D / dx, fr = (int64_t) SMR_GRID_Y * 256 * D % dx;
      ++ix1;
      do
      {
         yq += fq; yr += fr;
         if (yr >= dx) { yq++; yr -= dx; }
         int ix = ix1 < -1 ? -1 : ix1 > SMR_TW ? SMR_TW : (int) ix1;
         smr_cell_add(L, row, ix, (int) (sign * (yq - y_last)), (int) (sign * (yq - y_last) * 256));
         y_last = yq;
         ++ix1;
      } while (ix1 != ix2);
   }
   {
      int ix = ix2 < -1 ? -1 : ix2 > SMR_TW ? SMR_TW : (int) ix2;
      smr_cell_add(L, row, ix, (int) (sign * (SMR_GRID_Y - y_last)), (int) (sign * (SMR_GRID_Y - y_last) * fx2));
   }
}

#define SMR_RASTER_WAVES 4      /* independent wavefronts per workgroup, one tile at a time each */
#define SMR_BIG_SLOTS_MAX (1u << 20)      /* most edge slots of one tile (entries x 64) the second pass's lists are sized for */
#ifndef SMR_TILE_CHUNK
#define SMR_TILE_CHUNK 2        /* tiles a wavefront takes from its XCD's queue with one atomic on a dense chart (C2: 53 entries per
                                   tile; 1 is 0.5 % faster there); a sparse chart (C3: 0.5 entries per tile, most tiles empty) pays
                                   for the atomics, not for the tiles: 8 at a time there (raster 5.3 -> 2.2 ms per rule) */
#endif

/* -DSMR_SECTION_PROFILE: s_memtime stamps around the sections of k_raster, summed per section into
 * status[14..] (diagnostic build only; the stamps cost ~10 % and no output depends on them) */
#ifdef SMR_SECTION_PROFILE
#define TICK(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define TICK(i) do { } while (0)
#endif
/* -DSMR_COUNT_PROFILE: how often the parts of k_raster run, summed into small[192..] (diagnostic build only) */
#ifdef SMR_COUNT_PROFILE
#define COUNT(i) do { cnts[i]++; } while (0)
#define COUNTN(i, n) do { cnts[i] += (unsigned long long) (n); } while (0)
#else
#define COUNT(i) do { } while (0)
#define COUNTN(i, n) do { } while (0)
#endif

/* 4 waves per SIMD (128 VGPRs, a few spilled) measured 10 % faster than the 3 the allocator settles on by itself:
 * the kernel waits on LDS and L2 more than it issues */
#ifndef SMR_RASTER_WPE
#define SMR_RASTER_WPE 4
#endif
template <bool BIG>
__attribute__((amdgpu_waves_per_eu(SMR_RASTER_WPE, SMR_RASTER_WPE)))
__global__ __launch_bounds__(64 * SMR_RASTER_WAVES) void k_raster(raster_args_t A)
{
   __shared__ raster_lds_t LL[SMR_RASTER_WAVES];
   /* the wavefront's index is uniform, but the compiler cannot know: without readfirstlane every value derived from it
    * (tile, work-stream position, primitive state, all loop control) lives in VGPRs and branches through exec masks */
   const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   raster_lds_t &L = LL[wave];
   const int lane = threadIdx.x & 63;
   const unsigned long long lt_mask = (1ull << lane) - 1ull;
   const int my_row = lane / SMR_GRID_Y;               /* sub-row lane -> pixel row (lanes 60..63 idle) */
   smr_bucket_t<BIG> bk;
   if constexpr (BIG) bk.p = A.big_bucket + (size_t) (blockIdx.x * SMR_RASTER_WAVES + wave) * A.big_cap * 64u + lane;
   else bk.p = L.bucket[lane < SMR_NSUB ? lane : 0];
   const int BK = BIG ? (int) A.big_cap : SMR_BK;      /* entries a sub-row's list holds */

#pragma unroll
   for (int r = 0; r < SMR_TH; r++) { L.cover[r][lane] = 0; L.uarea[r][lane] = 0; }
   if (lane < SMR_TH) L.left[lane] = 0;
   SMR_WAVE_SYNC();

   /* persistent wavefronts: tile = gwave, gwave + nwaves, ... (1 M workgroup launches would cost more
    * than the work of the many empty or light tiles) */
#ifdef SMR_SECTION_PROFILE
   unsigned long long prof[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, tlast = __builtin_amdgcn_s_memtime();
#endif
#ifdef SMR_COUNT_PROFILE
   unsigned long long cnts[24] = { 0 };
#endif
   /* XCD-aware tile order: workgroups b and b+8 share an XCD and its L2, so XCD x rasterizes the tile
    * columns tx = x, x+8, ... and walks each column top to bottom: the waves resident on one XCD work on
    * vertically adjacent tiles at the same time, and an item's 1 KB of edge slots, needed by every tile
    * the item spans, comes from HBM once and from that L2 afterwards */
   const uint32_t xcd = blockIdx.x & 7u;
   const uint32_t TYt = A.n_tiles / A.TX;
   const uint32_t ncol = ((uint32_t) A.TX + 7u - xcd) >> 3;
   /* ... and the tiles of an XCD's columns are handed out dynamically, A.chunk vertically adjacent ones at a time: a chart's
    * load per tile varies, and a static share leaves wavefronts idle at the end of the launch */
   const uint32_t ktotal = ncol * TYt;
   uint32_t *const ctr = BIG ? A.tile_ctr : A.tile_ctr + 8u * xcd;
   /* without the queue (A.strided): the lanes of wavefront i look at the tiles i + (64 r + lane) * waves of the XCD's columns in
    * round r, so that neighbouring tiles, which tend to cost alike, go to different wavefronts */
   const bool strided = !BIG && A.strided != 0u;
   const uint32_t wx = (gridDim.x >> 3) * SMR_RASTER_WAVES, wi = (blockIdx.x >> 3) * SMR_RASTER_WAVES + wave;
   const uint32_t kstride = strided ? wx : 1u;
   uint32_t round = 0;
   for (;;)
   {
   uint32_t k0 = 0;
   if (strided) { k0 = round * 64u * wx + wi; round++; }
   else
   {
      if (lane == 0) k0 = atomicAdd(ctr, BIG ? 1u : A.chunk);
      k0 = __builtin_amdgcn_readfirstlane(k0);
   }
   if (k0 >= (BIG ? A.ovf_n : ktotal)) break;
   const uint32_t k1 = BIG ? k0 + 1u : strided ? ktotal : min(k0 + A.chunk, ktotal);
   /* a sparse rule leaves most tiles empty: the lanes look at 64 tiles of the chunk at once, and only those with entries are visited */
   for (uint32_t kb = k0; kb < k1; kb += strided ? ktotal : 64u)
   {
   unsigned long long todo = 1ull;
   if (!BIG)
   {
      const uint32_t kl = kb + (uint32_t) lane * kstride;
      const uint32_t tc = kl < k1 ? A.tile_cnt[(uint32_t) (kl % TYt) * A.TX + (xcd + 8u * (kl / TYt))] : 0u;
      todo = __ballot(tc != 0u && !(tc & SMR_TILE_PURE));      /* (a pure tile is k_blend's) */
   }
   while (todo)
   {
      const uint32_t kg = kb + ((uint32_t) __ffsll((long long) todo) - 1u) * kstride;
      todo &= todo - 1ull;
      /* the overflowed tiles come from the list (one queue: they are few), the others from the XCD's columns */
      const uint32_t tile = BIG ? A.ovf_list[kg] : (uint32_t) (kg % TYt) * A.TX + (xcd + 8u * (kg / TYt));
      const int tx = (int) (tile % (uint32_t) A.TX), ty = (int) (tile / (uint32_t) A.TX);
      const uint32_t beg = A.tile_off[tile], end = beg + (A.tile_cnt[tile] & ~SMR_TILE_PURE);
      if (beg == end) continue;
      COUNT(0);
      const int32_t X0 = tx * SMR_TW * 256, X1 = X0 + SMR_TW * 256;
      const int row0 = A.band_y0 + ty * SMR_TH;          /* absolute pixel row of the tile */
      const int ysbase = row0 * SMR_GRID_Y;
      const int my_ys = ysbase + lane;
      uint32_t layer[SMR_TH] = { 0, 0, 0, 0 };
      if (A.direct)
      {
         const int pxd = tx * SMR_TW + lane;
#pragma unroll
         for (int r = 0; r < SMR_TH; r++)
            if (pxd < A.W && ty * SMR_TH + r < A.band_h) layer[r] = A.canvas[(size_t) (ty * SMR_TH + r) * A.W + pxd];
      }

      /* state of the primitive being accumulated */
      uint32_t cur = 0xffffffffu, cur_flags = 0, cur_color = 0;
      uint32_t partial = 0, rowcnt[SMR_TH] = { 0, 0, 0, 0 };
      int cnt = 0, lw = 0;
      int cl = 1, cr = 0;            /* stroke pieces: the interval being grown (empty while cl > cr), tile-relative 24.8 */
      bool ovf = false, touched = false;
      bool tile_bad = false;         /* a list of this tile overflowed: what it holds from there on is not the tile */
      /* merge [l, r] into the lane's list of disjoint covered intervals (LDS bucket) */
      auto push_interval = [&](int l, int r)
      {
         for (int i = 0; i < cnt;)
         {
            int ia = bk.get(2 * i), ib = bk.get(2 * i + 1);
            if (ia <= r && l <= ib)
            {
               l = min(l, ia); r = max(r, ib);
               cnt--;
               bk.set(2 * i, bk.get(2 * cnt)); bk.set(2 * i + 1, bk.get(2 * cnt + 1));
               i = 0;
            }
            else i++;
         }
         if (2 * cnt + 2 <= BK) { bk.set(2 * cnt, l); bk.set(2 * cnt + 1, r); cnt++; }
         else ovf = true;
      };

      /* the tile's work stream is read 64 headers at a time (one coalesced 32-byte access per lane) and
       * handed out by v_readlane; the edge slots of entry k+1 are in flight while entry k is processed */
      uint4 hb0 = make_uint4(0u, 0xffffffffu, 0u, 0u), hb1 = make_uint4(0u, 0u, 0u, 0u);
      uint32_t bbase = beg;
      auto load_batch = [&]()
      {
         const uint32_t k = bbase + lane;
         if (k < end)
         {
            const uint4 *hp = reinterpret_cast<const uint4 *>(&A.hdrs[k]);
            hb0 = hp[0]; hb1 = hp[1];
         }
      };
      auto load_edge = [&](uint32_t k) -> smr_edge_t
      {
         smr_edge_t e; e.x1 = e.y1 = e.x2 = e.y2 = 0;
         if (k < end)
         {
            const int j = (int) (k - bbase);
            const uint32_t eo = __builtin_amdgcn_readlane(hb0.x, j), cf = __builtin_amdgcn_readlane(hb0.z, j);
            if (cf & (SMR_PF_DELTA << 8))
            {
               /* this column's alpha word of the strip instead of an edge slot */
               const uint32_t info = __builtin_amdgcn_readlane(hb1.w, j);
               const uint32_t i = (uint32_t) (tx * SMR_TW + lane - smr_strip_x0(info));
               if (i < smr_strip_n(info)) e.x1 = (int32_t) A.alpha[(size_t) eo + i];
            }
            else if ((uint32_t) lane < (cf & 0xffu)) e = A.edges[(size_t) eo + lane];
         }
         return e;
      };
      load_batch();
      smr_edge_t edge1 = load_edge(beg);
      uint32_t cur_rowin = 0; int cur_colmin = 0, cur_colmax = -1; bool cur_first = false;

      for (uint32_t ei = beg; ei <= end; ei++)
      {
         const smr_edge_t my_edge = edge1;
         smr_item_t it;
         it.prim = 0xffffffffu; it.cnt_flags = 0; it.edge_off = 0; it.color = 0;
         uint32_t it_rowbits = 0; int it_colmin = 0, it_colmax = -1;
         if (ei < end)
         {
            const int j = (int) (ei - bbase);
            it.edge_off = __builtin_amdgcn_readlane(hb0.x, j); it.prim = __builtin_amdgcn_readlane(hb0.y, j);
            it.cnt_flags = __builtin_amdgcn_readlane(hb0.z, j); it.color = __builtin_amdgcn_readlane(hb0.w, j);
            it_rowbits = __builtin_amdgcn_readlane(hb1.x, j);
            it_colmin = (int) __builtin_amdgcn_readlane(hb1.y, j); it_colmax = (int) __builtin_amdgcn_readlane(hb1.z, j);
            if (ei + 1 == bbase + 64) { bbase += 64; load_batch(); }
            edge1 = load_edge(ei + 1);
         }
         TICK(0);      /* header hand-out + the previous entry's crossing loop */

         /* ================= flush: the previous primitive is complete ================= */
         if (ei < end) COUNT(1);
         if (it.prim != cur && cur != 0xffffffffu && touched)
         {
            const bool pieces = (cur_flags & SMR_PF_PIECES) != 0;
            COUNT(2); if (pieces) COUNT(3);
            const uint32_t mask = (cur_flags & SMR_PF_EVENODD) ? 1u : ~0u;
            const smr_edge_t *E = A.edges;

            /* ---- sub-sampled rows: crossings -> spans -> cells ---- */
            const bool mine = lane < SMR_NSUB && ((partial >> my_row) & 1u);
            auto add_span = [&](int x1, int x2)
            {
               if (x1 == x2) return;
               int ix1 = x1 >> 8, ix2 = x2 >> 8, fx1 = x1 & 255, fx2 = x2 & 255;
               if (ix1 != ix2)
               {
                  smr_cell_add(L, my_row, ix1, 1, 2 * fx1);
                  smr_cell_add(L, my_row, ix2, -1, -2 * fx2);
               }
               else smr_cell_add(L, my_row, ix1, 0, 2 * (fx1 - fx2));
            };
            COUNTN(4, smr_wave_max(mine ? cnt : 0));
#ifdef SMR_KO
            if (SMR_KO & 8) cnt = 0;
#endif
            if (mine && pieces)
            {
               /* the interval still open takes the stored ones it meets along and goes straight to the cells: it needs no
                * room in the list (pushed there, a full list dropped it after the overflow had been looked at) */
               if (cl <= cr)
               {
                  for (int i = 0; i < cnt;)
                  {
                     const int ia = bk.get(2 * i), ib = bk.get(2 * i + 1);
                     if (ia <= cr && cl <= ib)
                     {
                        cl = min(cl, ia); cr = max(cr, ib);
                        cnt--;
                        bk.set(2 * i, bk.get(2 * cnt)); bk.set(2 * i + 1, bk.get(2 * cnt + 1));
                        i = 0;
                     }
                     else i++;
                  }
                  add_span(cl, cr);
               }
               for (int i = 0; i < cnt; i++) add_span(bk.get(2 * i), bk.get(2 * i + 1));
            }
            else if (mine)
            {
               for (int i = 1; i < cnt; i++)
               {
                  const int v = bk.get(i);
                  int k = i - 1;
                  while (k >= 0 && (bk.get(k) >> 1) > (v >> 1)) { bk.set(k + 1, bk.get(k)); k--; }
                  bk.set(k + 1, v);
               }
               int winding = lw;
               int xstart = (((uint32_t) winding & mask) != 0) ? 0 : -1;
               for (int i = 0; i < cnt; i++)
               {
                  const int bv = bk.get(i);
                  int xend = bv >> 1;
                  winding += (bv & 1) ? 1 : -1;
                  if (((uint32_t) winding & mask) == 0)
                  {
                     if (i + 1 < cnt && (bk.get(i + 1) >> 1) == xend) continue;
                     if (xstart >= 0) add_span(xstart, xend);
                     xstart = -1;
                  }
                  else if (xstart < 0) xstart = xend;
               }
               if (xstart >= 0) add_span(xstart, SMR_TW * 256);
            }

            if (__ballot(ovf)) tile_bad = true;
            TICK(1);      /* flush: spans of the sub-sampled rows */
            /* ---- rows that take libcairo's exact full-row path ---- */
            uint32_t fullrows = 0;
#pragma unroll
            for (int r = 0; r < SMR_TH; r++) if (!((partial >> r) & 1u) && rowcnt[r] > 0) fullrows |= 1u << r;
            /* Pieces need no edges from beyond the items binned here: a piece lies wholly inside one item, and one that spans
             * a vertex-free row has exactly its left and its right edge active in it, so the pieces of items that do not
             * reach this tile add +1 and -1 to the winding left of it -- nothing.  (Fill chunks are binned up to the
             * primitive's right edge for the same purpose.) */
            SMR_WAVE_SYNC();
            uint32_t nmax = 0;
#pragma unroll
            for (int r = 0; r < SMR_TH; r++) if ((fullrows >> r) & 1u) nmax = max(nmax, rowcnt[r]);
#ifdef SMR_KO
            if (SMR_KO & 1) fullrows = 0;
#endif
            if (fullrows && nmax <= 16)
            {
               COUNT(5); COUNTN(16, nmax);
               /* ---- fast form: lane = (row, edge slot), 16 slots per row, all full rows at once ---- */
               const int grp = lane >> 4, es = lane & 15;
               const uint32_t my_n = grp == 0 ? rowcnt[0] : grp == 1 ? rowcnt[1] : grp == 2 ? rowcnt[2] : rowcnt[3];
               const bool onr = ((fullrows >> grp) & 1u) && (uint32_t) es < my_n;
               const int rs0 = ysbase + grp * SMR_GRID_Y;
               int32_t c0 = 0, c1 = 0, dir = 0, D = 1, q1 = 0, r1 = 0, q2 = 0, r2 = 0, dxs = 0;
               smr_oedge_t o; o.p1x = o.p1y = o.dxs = 0; o.D = 1; o.dir = 0;
               if (onr) { const int *re = L.redge[grp][es]; o.p1x = re[0]; o.p1y = re[1]; o.dxs = re[2]; o.D = re[3]; o.dir = re[4]; }
               /* short edges (|dx|, dy < 16 px: every edge of a chart of small polygons) evaluate in 32 bits, as in the crossing loop */
               const bool allshort = !__ballot(onr && !(abs(o.dxs) < 4096 && o.D <= 60 * 4096));
               if (onr)
               {
                  D = o.D; dir = o.dir; dxs = o.dxs;
                  int32_t hq, hr;
                  if (allshort)
                  {
                     const float rf = __builtin_amdgcn_rcpf((float) o.D);
                     /* floor(n / D) and the remainder: truncating fp32 estimate within 1 of the floor, exact correction */
                     auto qr32 = [&](int n, int32_t &q, int32_t &r)
                     {
                        const int qe = (int) fmaf((float) n, rf, -0.5f);
                        const int rr = __mul24(qe, -o.D) + n;
                        const int adj = (rr >= o.D ? 1 : 0) + (rr >> 31);
                        q = qe + adj; r = rr - __mul24(adj, o.D);
                     };
                     const int c32 = (int) ((uint32_t) rs0 * 512u + 256u - 30u * (uint32_t) o.p1y);
                     const int n1 = __mul24(2 * o.dxs, c32);
                     qr32(n1, q1, r1);
                     qr32(n1 + __mul24(2 * o.dxs, 512 * SMR_GRID_Y), q2, r2);
                     const int nn = o.dxs * 1024;
                     int32_t qa, ra;
                     qr32(abs(nn), qa, ra);
                     hq = nn < 0 ? -(qa >> 1) : (qa >> 1); hr = nn < 0 ? -(ra >> 1) : (ra >> 1);
                  }
                  else
                  {
                     const int64_t ec = 256 - 30ll * o.p1y;
                     const double inv = smr_rcp((double) o.D);
                     smr_edge_qr(ec, 2 * o.dxs, o.D, inv, rs0, q1, r1);
                     smr_edge_qr(ec, 2 * o.dxs, o.D, inv, rs0 + SMR_GRID_Y, q2, r2);
                     const int64_t nn = (int64_t) dxs * 1024;
                     int64_t dq = (int64_t) trunc((double) nn * inv);
                     int64_t dr = nn - dq * D;
                     if (nn >= 0) { if (dr < 0) { dq--; dr += D; } else if (dr >= D) { dq++; dr -= D; } }
                     else { if (dr > 0) { dq++; dr -= D; } else if (dr <= -(int64_t) D) { dq--; dr += D; } }
                     hq = (int32_t) (dq / 2); hr = (int32_t) (dr / 2);
                  }
                  q1 += o.p1x; q2 += o.p1x;
                  c0 = q1 + (r1 >= (D >> 1) ? 1 : 0); c1 = q2 + (r2 >= (D >> 1) ? 1 : 0);
                  /* step back from the sample location (half a sub-row) to the row boundary */
                  q1 -= hq; r1 -= hr;
                  if (r1 < 0) { --q1; r1 += D; } else if (r1 >= D) { ++q1; r1 -= D; }
                  q2 -= hq; r2 -= hr;
                  if (r2 < 0) { --q2; r2 += D; } else if (r2 >= D) { ++q2; r2 -= D; }
                  q1 -= X0; q2 -= X0;
               }
               /* position in the row's (cell, next cell) order, winding before me, neighbours' cells */
               int wsum = 0, pred_c0 = INT32_MIN, succ_c0 = INT32_MAX;
               bool has_pred = false, has_succ = false;
               for (uint32_t j = 0; j < nmax; j++)
               {
                  const int src = (lane & 48) + (int) j;
                  const int kc0 = __shfl(c0, src), kc1 = __shfl(c1, src), kdir = __shfl(dir, src);
                  if (j < my_n && (int) j != es)
                  {
                     const bool before = kc0 < c0 || (kc0 == c0 && (kc1 < c1 || (kc1 == c1 && (int) j < es)));
                     if (before) { wsum += kdir; pred_c0 = max(pred_c0, kc0); has_pred = true; }
                     else { succ_c0 = min(succ_c0, kc0); has_succ = true; }
                  }
               }
               const int wincl = wsum + dir;
               /* (no "the last edge closes whatever is open": the paths are closed, so a span still open after the tile's last
                * edge ends at an edge of an item right of the tile, which is not binned here: it covers the rest of the tile) */
               const bool is_end = onr && ((uint32_t) wincl & mask) == 0 && (!has_succ || succ_c0 != c0);
               const bool is_start = onr && (!has_pred || (((uint32_t) wsum & mask) == 0 && pred_c0 != c0));
               /* every span-boundary edge adds its exact trapezoid coverage to the cells it crosses inside
                * the tile (libcairo's cell_list_render_edge; the per-cell y steps are floors of one
                * 64-bit quotient each, estimated with one fp64 multiply and corrected exactly) */
               if (is_start || is_end)
               {
                  const int sgn = is_start ? 1 : -1;
                  int ix1 = (int) (q1 >> 8), ix2 = (int) (q2 >> 8), fx1 = q1 & 255, fx2 = q2 & 255;
                  if (ix1 == ix2)
                  {
                     const int ix = ix1 < -1 ? -1 : ix1 > SMR_TW ? SMR_TW : ix1;
                     smr_cell_add(L, grp, ix, sgn * SMR_GRID_Y, sgn * (fx1 + fx2) * SMR_GRID_Y);
                  }
                  else
                  {
                     COUNT(17);
                     int32_t qa = q1, ra = r1, qb = q2, rb = r2;
                     if (ix2 < ix1)
                     {
                        int t = ix1; ix1 = ix2; ix2 = t;
                        t = fx1; fx1 = fx2; fx2 = t;
                        qa = q2; ra = r2; qb = q1; rb = r1;
                     }
                     if (ix2 < 0) atomicAdd(&L.left[grp], sgn * SMR_GRID_Y);
                     else if (ix1 < SMR_TW)
                     {
                        const int64_t dx = (int64_t) (qb - qa) * D + (rb - ra);
                        const int64_t tmp0 = ((int64_t) (256 - fx1) * D - ra) * SMR_GRID_Y;
                        const int64_t step = (int64_t) (SMR_GRID_Y * 256) * D;
                        const double inv = smr_rcp(smr_i64_to_f64(dx));
                        /* y (in sub-rows) at which the edge crosses the right boundary of cell `cell` */
                        auto Y = [&](int cell) -> int
                        {
                           const int64_t n = tmp0 + (int64_t) (cell - ix1) * step;
                           int y = (int) (smr_i64_to_f64(n) * inv);
                           int64_t rem = n - (int64_t) y * dx;
                           if (rem < 0) y--; else if (rem >= dx) y++;
                           return y;
                        };
                        int k = ix1, ylast = 0;
                        if (ix1 < 0) { ylast = Y(-1); atomicAdd(&L.left[grp], sgn * ylast); k = 0; }
                        const int kend = min(ix2, SMR_TW - 1);
                        for (; k <= kend; k++)
                        {
                           const int yhi = k == ix2 ? SMR_GRID_Y : Y(k);
                           const int cov = yhi - ylast;
                           const int fac = k == ix1 ? 256 + fx1 : k == ix2 ? fx2 : 256;
                           if (cov) { atomicAdd(&L.cover[grp][k], sgn * cov); atomicAdd(&L.uarea[grp][k], sgn * cov * fac); }
                           ylast = yhi;
                        }
                     }
                  }
               }
               fullrows = 0;
            }
            for (int r = 0; r < SMR_TH; r++)
            {
               if (!((fullrows >> r) & 1u)) continue;
               uint32_t n = rowcnt[r];
               if (n > SMR_ROWCAP) { if (lane == 0) atomicOr(A.status, SMR_ST_ROWCAP); continue; }
               COUNT(6);
               const int rs0 = ysbase + r * SMR_GRID_Y;
               int64_t q0 = 0, q1 = 0; int32_t r0 = 0, r1 = 0, c0 = 0, c1 = 0, dxs = 0, D = 1, dir = 0;
               const bool on = (uint32_t) lane < n;
               if (on)
               {
                  smr_oedge_t o = smr_orient(E[L.rowlist[r][lane]]);
                  dxs = o.dxs; D = o.D; dir = o.dir;
                  smr_edge_x(o.p1x, o.p1y, o.dxs, o.D, rs0, q0, r0);
                  smr_edge_x(o.p1x, o.p1y, o.dxs, o.D, rs0 + SMR_GRID_Y, q1, r1);
                  c0 = smr_cell(q0, r0, D, dxs); c1 = smr_cell(q1, r1, D, dxs);
               }
               /* rank sort by (cell, next cell) */
               unsigned long long key = ((unsigned long long) (uint32_t) (c0 + 0x40000000) << 32) | (uint32_t) (c1 + 0x40000000);
               L.srt.key[lane] = key;
               SMR_WAVE_SYNC();
               int rank = 0;
               for (uint32_t j = 0; j < n; j++)
               {
                  unsigned long long kj = L.srt.key[j];
                  rank += (kj < key || (kj == key && j < (uint32_t) lane)) ? 1 : 0;
               }
               if (on) L.srt.perm[rank] = lane;
               SMR_WAVE_SYNC();
               int src = on ? L.srt.perm[lane] : lane;
               q0 = __shfl(q0, src); q1 = __shfl(q1, src);
               r0 = __shfl(r0, src); r1 = __shfl(r1, src); c0 = __shfl(c0, src); c1 = __shfl(c1, src);
               dxs = __shfl(dxs, src); D = __shfl(D, src); dir = __shfl(dir, src);
               int c0n = __shfl_down(c0, 1);
               /* winding prefix */
               int wsum = on ? dir : 0;
#pragma unroll
               for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(wsum, d); if (lane >= d) wsum += t; }
               bool last = (uint32_t) lane + 1 == n;
               bool is_end = on && ((uint32_t) wsum & mask) == 0 && (last || c0n != c0);
               int prev_end = __shfl_up((int) is_end, 1);
               bool is_start = on && (lane == 0 || prev_end);
               if (is_start) smr_render_edge(L, r, X0, dxs, D, q0, r0, q1, r1, +1);
               else if (is_end) smr_render_edge(L, r, X0, dxs, D, q0, r0, q1, r1, -1);
               SMR_WAVE_SYNC();
            }
            SMR_WAVE_SYNC();

            TICK(2);      /* flush: full rows */
            /* ---- cells -> alpha -> ordered blend into the group layer; lane = column ---- */
            const uint32_t spix = cur_color;
            const bool lerp = (spix >> 24) == 255u || (cur_first && !A.direct);
            /* libcairo composites inside the primitive's pixel extents only */
            const bool col_in = lane >= cur_colmin && lane <= cur_colmax;
#ifdef SMR_KO
            if (!(SMR_KO & 2))
#endif
#pragma unroll
            for (int r = 0; r < SMR_TH; r++)
            {
               int c = L.cover[r][lane];
               int ua = L.uarea[r][lane];
               const int lf = L.left[r];
               if (!__ballot((c | ua) != 0) && lf == 0) continue;        /* nothing of this primitive on the row */
               L.cover[r][lane] = 0; L.uarea[r][lane] = 0;
               COUNT(7);
               /* interior rows carry no cover change at all: the prefix sum is the left cover */
               c = __ballot(c != 0) ? smr_wave_scan(c) + lf : lf;
               int area = c * 512 - ua;
               int a = (area + (area << 4) + 256) >> 9;
               a = a < 0 ? 0 : a > 255 ? 255 : a;
               if (!col_in || !((cur_rowin >> r) & 1u)) a = 0;
               if (A.pat)
               {
                  /* an image source: pixman's OVER through the coverage mask, pixel by pixel */
                  if (a) layer[r] = smr_blend_over_mask(smr_pat_pix(A.pat, A.pat_w, A.pat_h, tx * SMR_TW + lane, row0 + r), (uint32_t) a, layer[r]);
                  continue;
               }
               if (lerp && !__ballot(a != 0 && a != 255)) layer[r] = a ? spix : layer[r];      /* no edge pixel on the row */
               else if (__ballot(a != 0)) COUNT(8);
               if (lerp && !__ballot(a != 0 && a != 255)) { }
               else if (a) layer[r] = lerp ? smr_blend_lerp(spix, (uint32_t) a, layer[r]) : smr_blend_over_mask(spix, (uint32_t) a, layer[r]);
            }
            SMR_WAVE_SYNC();
            if (lane < SMR_TH) L.left[lane] = 0;
            SMR_WAVE_SYNC();
         }
         TICK(3);         /* flush: cells -> alpha -> blend */
         if (ei >= end) break;

         /* ================= a new primitive begins ================= */
         if (it.prim != cur)
         {
            cur = it.prim; cur_flags = (it.cnt_flags >> 8) & 0xffu; cur_color = it.color;
            cnt = 0; lw = 0; cl = 1; cr = 0; ovf = false; touched = false;
#pragma unroll
            for (int r = 0; r < SMR_TH; r++) rowcnt[r] = 0;
            partial = it_rowbits & 0xfu; cur_rowin = (it_rowbits >> 4) & 0xfu;
            cur_colmin = it_colmin; cur_colmax = it_colmax; cur_first = (it.cnt_flags >> 16) & 1u;
            if (cur_flags & SMR_PF_DELTA)
            {
               /* k_cover has run the scan converter over this primitive: the lane holds its column's alpha of the four rows */
               COUNT(18);
               partial = 0;
               const uint32_t a4 = (uint32_t) my_edge.x1;
               if (__ballot(a4 != 0))
               {
                  const uint32_t spix = cur_color;
                  const bool lerp = (spix >> 24) == 255u || (cur_first && !A.direct);
#pragma unroll
                  for (int r = 0; r < SMR_TH; r++)
                  {
                     const uint32_t a = (a4 >> (8 * r)) & 255u;
                     if (!__ballot(a != 0)) continue;
                     if (A.pat) { if (a) layer[r] = smr_blend_over_mask(smr_pat_pix(A.pat, A.pat_w, A.pat_h, tx * SMR_TW + lane, row0 + r), a, layer[r]); continue; }
                     if (lerp && !__ballot(a != 255u && a != 0)) layer[r] = a ? spix : layer[r];      /* no edge pixel on the row */
                     else if (a) layer[r] = lerp ? smr_blend_lerp(spix, a, layer[r]) : smr_blend_over_mask(spix, a, layer[r]);
                  }
               }
            }
         }
         if (cur_flags & SMR_PF_DELTA) continue;       /* (further items of the same primitive bring nothing) */

         /* ================= one item: up to 64 edge slots, one per lane ================= */
         const uint32_t icnt = it.cnt_flags & 0xffu;
         const bool pieces = (cur_flags & SMR_PF_PIECES) != 0;
         smr_oedge_t o; o.dir = 0; o.ytop = 0; o.ybot = 0; o.p1x = o.p1y = o.dxs = 0; o.D = 1;
         if ((uint32_t) lane < icnt) o = smr_orient(my_edge);
         const int32_t eminx = o.p1x + (o.dxs < 0 ? o.dxs : 0), emaxx = o.p1x + (o.dxs > 0 ? o.dxs : 0);
         const bool inband = o.dir != 0 && o.ytop < ysbase + SMR_NSUB && o.ybot > ysbase && eminx < X1;

         /* full rows: remember who spans them */
         if (~partial & 0xfu)
         {
            COUNT(9);
#pragma unroll
            for (int r = 0; r < SMR_TH; r++)
            {
               if ((partial >> r) & 1u) continue;
               int rs0 = ysbase + r * SMR_GRID_Y, rs1 = rs0 + SMR_GRID_Y;
               bool act = o.dir != 0 && o.ytop <= rs0 && o.ybot >= rs1;
               unsigned long long am = __ballot(act);
               if (pieces && rowcnt[r] + __popcll(am) > SMR_ROWCAP)
               {
                  /* the row's list is about to overflow: only the pieces (4 aligned lanes) that come near the tile from here
                   * on.  Any whole piece may stay away: it adds its +1 and its -1 beside the tile (see the flush) */
                  int pmn = o.dir != 0 ? eminx : INT32_MAX, pmx = o.dir != 0 ? emaxx : INT32_MIN;
                  pmn = min(pmn, __shfl_xor(pmn, 1)); pmx = max(pmx, __shfl_xor(pmx, 1));
                  pmn = min(pmn, __shfl_xor(pmn, 2)); pmx = max(pmx, __shfl_xor(pmx, 2));
                  act = act && pmx >= X0 - 1024 && pmn < X1 + 1024;
                  /* and not the sides that two neighbouring pieces share: the same edge twice, once down and once up (slot 3 of
                   * piece k is slot 1 of piece k - 1).  The two sort next to each other and leave the winding as it was */
                  const int m4 = lane & 3;
                  int src = m4 == 3 ? lane - 6 : m4 == 1 ? lane + 6 : lane;
                  if (src < 0 || src > 63) src = lane;
                  const int t_x = __shfl(o.p1x, src), t_y = __shfl(o.p1y, src), t_dx = __shfl(o.dxs, src), t_D = __shfl(o.D, src);
                  const int t_dir = __shfl(o.dir, src), t_act = __shfl((int) act, src);
                  const bool twin = src != lane && act && t_act && t_x == o.p1x && t_y == o.p1y && t_dx == o.dxs && t_D == o.D && t_dir == -o.dir;
                  act = act && !twin;
                  am = __ballot(act);
               }
               if (am)
               {
                  touched = true;
                  uint32_t pos = rowcnt[r] + __popcll(am & lt_mask);
                  if (act && pos < SMR_ROWCAP) L.rowlist[r][pos] = it.edge_off + lane;
                  if (act && pos < 16) { int *re = L.redge[r][pos]; re[0] = o.p1x; re[1] = o.p1y; re[2] = o.dxs; re[3] = o.D; re[4] = o.dir; }
                  rowcnt[r] += __popcll(am);
               }
            }
         }

         TICK(4);         /* orient + full-row bookkeeping */
#define XTICK(i) TICK(i)
#define XCOUNT(i) COUNT(i)
#define XCOUNTN(i, n) COUNTN(i, n)
#include "raster_cross.inc"
#undef XTICK
#undef XCOUNT
#undef XCOUNTN
      }

      TICK(7);
      if (tile_bad)
      {
         /* nothing of this tile is written: the fast instantiation hands it to the second one, which has room for every
          * edge slot binned here; should that one run out all the same, the pass fails (and the host restores the canvas) */
         if (lane == 0)
         {
            if constexpr (BIG) atomicOr(A.status, SMR_ST_BUCKET_OVERFLOW);
            else
            {
               A.ovf_list[atomicAdd(&A.ovf_cnt[0], 1u)] = tile;
               atomicMax(&A.ovf_cnt[1], end - beg);
            }
         }
         continue;
      }
      /* pop_group_to_source + paint: layer OVER canvas */
      const int px = tx * SMR_TW + lane;
      if (px < A.W)
      {
#pragma unroll
         for (int r = 0; r < SMR_TH; r++)
         {
            int row = ty * SMR_TH + r;
            if (row < A.band_h && (layer[r] || A.direct))
            {
               uint32_t *d = A.canvas + (size_t) row * A.W + px;
               *d = A.direct ? layer[r] : smr_blend_over(layer[r], *d);
            }
         }
      }
      TICK(8);            /* layer OVER canvas */
   }
   }
   }
#ifdef SMR_SECTION_PROFILE
   if (lane == 0) for (int i = 0; i < 12; i++) atomicAdd((unsigned long long *) (A.status + 14) + i, prof[i]);
#endif
#ifdef SMR_COUNT_PROFILE
   if (lane == 0) for (int i = 0; i < 24; i++) atomicAdd((unsigned long long *) (A.status + 190) + i, cnts[i]);
#endif
}

/* Tiles whose stream holds nothing but k_cover's alpha words (SMR_TILE_PURE: on a chart of small polygons nearly all of them)
 * need none of k_raster's machinery: no lists, no cells, no LDS.  One wavefront per tile reads the tile's headers 64 at a time,
 * every lane its column's word of each entry, four entries in flight, and blends the four rows into the layer in registers:
 * the statements of k_raster's own branch for such an entry, so the pixels are the same.  Twice k_raster's wavefronts per SIMD. */
#define SMR_BLEND_WAVES 4
#ifndef SMR_BLEND_BATCH
#define SMR_BLEND_BATCH 16      /* entries whose words a wavefront holds in LDS at a time (at most 32: one mask word per lane; 16: 4 KB per wavefront, ten wavefronts per SIMD -- with 32 the kernel was as slow as without the rounds, with 8 slower than with 16) */
#endif
#ifndef SMR_BLEND_FLIGHT
#define SMR_BLEND_FLIGHT 4      /* loads in flight */
#endif
__global__ __launch_bounds__(64 * SMR_BLEND_WAVES) void k_blend(raster_args_t A)
{
   __shared__ uint32_t qa[SMR_BLEND_WAVES][SMR_BLEND_BATCH][64];     /* per wavefront: the words of a batch of entries, column by column */
   const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const uint32_t xcd = blockIdx.x & 7u;                /* k_raster's tile order: an XCD walks its tile columns top to bottom */
   const uint32_t TYt = A.n_tiles / A.TX;
   const uint32_t ncol = ((uint32_t) A.TX + 7u - xcd) >> 3;
   const uint32_t ktotal = ncol * TYt;
   /* one loop for both ways of finding tiles (the body once: as a function called from two places it went to scratch memory):
    * a sparse rule goes by k_tile_sort's list of its few tiles, a chart full of tiles looks at the tiles of its XCD's share */
   const bool by_list = A.blend_list != nullptr;
   uint32_t *const ctr = by_list ? A.blend_ctr + 65u : A.blend_ctr + 8u * xcd;
   const uint32_t n_list = by_list ? *A.blend_list_n : 0u;
   /* (list) as many tiles per take as leaves every wavefront of the launch one take: a short list is spread one tile per wavefront */
   const uint32_t take = by_list ? max(1u, min(64u, n_list / (gridDim.x * SMR_BLEND_WAVES))) : A.chunk;
   uint32_t kb = 0, k1 = 0;
   for (;;)
   {
      uint32_t tile_l;
      unsigned long long todo;
      if (kb >= k1)
      {
         /* the tiles are handed out dynamically: with k_raster's static share this kernel took 14.6 ms instead of 11.3 on C2 (and with
          * sixteen queues per XCD instead of one the same 11.0: the queue's atomics are not what it waits for) */
         uint32_t k0 = 0;
         if (lane == 0) k0 = atomicAdd(ctr, take);
         k0 = __builtin_amdgcn_readfirstlane(k0);
         if (k0 >= (by_list ? n_list : ktotal)) break;  /* (every wavefront gets here: the queue only grows) */
         kb = k0; k1 = min(k0 + take, by_list ? n_list : ktotal);
      }
      {
         const uint32_t kl = kb + (uint32_t) lane;
         if (by_list)
         {
            tile_l = kl < k1 ? A.blend_list[kl] : 0u;
            todo = __ballot(kl < k1);
         }
         else
         {
            tile_l = (uint32_t) (kl % TYt) * A.TX + (xcd + 8u * (kl / TYt));
            const uint32_t tc = kl < k1 ? A.tile_cnt[tile_l] : 0u;
            todo = __ballot((tc & SMR_TILE_PURE) != 0u);
         }
         kb += 64u;
      }
      while (todo)
      {
         const uint32_t tile = (uint32_t) __builtin_amdgcn_readlane((int) tile_l, __ffsll((long long) todo) - 1);
         todo &= todo - 1ull;
         const int tx = (int) (tile % (uint32_t) A.TX), ty = (int) (tile / (uint32_t) A.TX);
         const uint32_t beg = A.tile_off[tile], end = beg + (A.tile_cnt[tile] & ~SMR_TILE_PURE);
         const int px = tx * SMR_TW + lane;
         /* (four names, not an array: the compiler copied the whole array wherever two paths of a row met) */
         uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0;
         if (A.direct && px < A.W)
         {
            const uint32_t *cp = A.canvas + (size_t) (ty * SMR_TH) * A.W + px;
            const int nr = A.band_h - ty * SMR_TH;
            if (nr > 0) l0 = cp[0];
            if (nr > 1) l1 = cp[A.W];
            if (nr > 2) l2 = cp[2 * (size_t) A.W];
            if (nr > 3) l3 = cp[3 * (size_t) A.W];
         }
         uint32_t cur = 0xffffffffu;
         for (uint32_t bbase = beg; bbase < end; bbase += 64u)
         {
            uint4 hb0 = make_uint4(0u, 0xffffffffu, 0u, 0u);
            uint32_t hinfo = 0;
            const uint32_t k = bbase + (uint32_t) lane;
            if (k < end)
            {
               const uint4 *hp = reinterpret_cast<const uint4 *>(&A.hdrs[k]);
               hb0 = hp[0]; hinfo = hp[1].w;
            }
            const int n = (int) min(64u, end - bbase);
            /* the headers are here before the first word is asked for: the loop below reads them with v_readlane on several paths, and
             * a wait for "the headers, wherever they are still on their way" placed behind a word's load is a wait for that word too
             * (one counter, in order): the words came one after the other */
            __builtin_amdgcn_s_waitcnt(0x0f70);         /* vmcnt(0) */
            /* SMR_BLEND_BATCH entries at a time.  A lane's column is empty in most entries of a tile (a border's strip is two short runs with
             * nothing between them, a fill ends somewhere), and a blend costs the whole wavefront ~25 instructions per row whoever
             * needs it: so every lane first notes WHICH of the batch's entries have a word for it (the words go to LDS), then the
             * wavefront blends in rounds, every lane its own next entry -- as many rounds as the busiest column has entries,
             * each lane's entries in the tile's order, so every pixel sees the blends it saw before */
            for (int jb = 0; jb < n; jb += SMR_BLEND_BATCH)
            {
               const int nb = min(SMR_BLEND_BATCH, n - jb);
               uint32_t nz = 0, firstmask = 0;
               for (int j0 = 0; j0 < nb; j0 += SMR_BLEND_FLIGHT)
               {
                  uint32_t a4[SMR_BLEND_FLIGHT];
                  bool take[SMR_BLEND_FLIGHT];
#pragma unroll
                  for (int u = 0; u < SMR_BLEND_FLIGHT; u++)
                  {
                     a4[u] = 0; take[u] = false;
                     if (j0 + u < nb)
                     {
                        const int jj = jb + j0 + u;
                        const uint32_t prim = __builtin_amdgcn_readlane(hb0.y, jj);
                        if (prim == cur) continue;                       /* (further entries of a primitive bring nothing) */
                        cur = prim; take[u] = true;
                        const uint32_t eo = __builtin_amdgcn_readlane(hb0.x, jj), info = __builtin_amdgcn_readlane(hinfo, jj);
                        firstmask |= ((__builtin_amdgcn_readlane(hb0.z, jj) >> 16) & 1u) << (j0 + u);
                        const uint32_t i = (uint32_t) (px - smr_strip_x0(info));
                        if (i < smr_strip_n(info)) a4[u] = A.alpha[(size_t) eo + i];
                     }
                  }
#pragma unroll
                  for (int u = 0; u < SMR_BLEND_FLIGHT; u++)
                     if (take[u])
                     {
                        qa[wv][j0 + u][lane] = a4[u];
                        nz |= (a4[u] != 0 ? 1u : 0u) << (j0 + u);
                     }
               }
               while (__ballot(nz != 0))
               {
                  const bool have = nz != 0;
                  const int jq = have ? __ffs((int) nz) - 1 : 0;
                  nz &= nz - 1u;
                  const uint32_t aw = have ? qa[wv][jq][lane] : 0u;
                  const uint32_t spix = (uint32_t) __builtin_amdgcn_ds_bpermute((jb + jq) << 2, (int) hb0.w);
                  const bool le = (spix >> 24) == 255u || (((firstmask >> jq) & 1u) != 0u && !A.direct);
                  /* k_raster's statements for such an entry, the blends without their per-lane shortcuts (the arithmetic gives
                   * the same for a = 0 and a = 255: smr_blend_lerp_all) */
#define SMR_BLEND_ROW(L, R) do { \
                     const uint32_t a = (aw >> (8 * (R))) & 255u; \
                     if (__ballot(a != 0)) \
                     { \
                        if (A.pat) L = smr_blend_over_mask_all(smr_pat_pix(A.pat, A.pat_w, A.pat_h, px, A.band_y0 + ty * SMR_TH + (R)), a, L); \
                        else \
                        { \
                           if (__ballot(a != 0 && !le)) { const uint32_t t = smr_blend_over_mask_all(spix, a, L); L = le ? L : t; } \
                           if (__ballot(a != 0 && le)) \
                           { \
                              if (!__ballot(le && a != 255u && a != 0)) L = le && a ? spix : L;      /* no edge pixel on the row */ \
                              else { const uint32_t t = smr_blend_lerp_all(spix, a, L); L = le ? t : L; } \
                           } \
                        } \
                     } } while (0)
                  SMR_BLEND_ROW(l0, 0); SMR_BLEND_ROW(l1, 1); SMR_BLEND_ROW(l2, 2); SMR_BLEND_ROW(l3, 3);
#undef SMR_BLEND_ROW
               }
            }
         }
         /* pop_group_to_source + paint: layer OVER canvas */
         if (px < A.W)
         {
            uint32_t *cp = A.canvas + (size_t) (ty * SMR_TH) * A.W + px;
            const int nr = A.band_h - ty * SMR_TH;
#define SMR_BLEND_OUT(L, R) do { if (nr > (R) && (L || A.direct)) { uint32_t *d = cp + (size_t) (R) * A.W; *d = A.direct ? L : smr_blend_over(L, *d); } } while (0)
            SMR_BLEND_OUT(l0, 0); SMR_BLEND_OUT(l1, 1); SMR_BLEND_OUT(l2, 2); SMR_BLEND_OUT(l3, 3);
#undef SMR_BLEND_OUT
         }
      }
   }
}

/* ------------------------------------------------------------------------------------ */
/* coverage of small primitives, one wavefront per primitive                              */
/* ------------------------------------------------------------------------------------ */
/* A chart of many small polygons pays in k_raster for every (item, tile) pair: header, orientation, per-edge constants, the
 * flush's fixed parts and the blend, with two or three edges per pair that matter (C2: 55.8 M pairs, 650 VALU instructions
 * each).  Coverage does not depend on tiles: libcairo's scan converter accumulates (cover, area) per pixel row and cell, and a
 * tile needs of a primitive's row only those cells: the ones inside it, and the sum of the covers left of it.  So for every
 * primitive that is small enough (at most SMR_CV_MAXSLOTS edge slots, 126 px wide) ONE wavefront runs the scan converter over
 * the primitive's own rows, four at a time with k_raster's own crossing loop (raster_cross.inc: lanes = the 60 sub-rows), and
 * turns the cells into alpha right there (prefix sum of cover, libcairo's area-to-alpha, the primitive's pixel extents): per strip
 * one word per column with the four rows' alpha.  k_raster's work for such a primitive in a tile is to read its column's word and
 * to blend.
 * Anything that does not fit (lists, arenas, cell range) leaves the primitive to k_raster's own path: never an error. */
struct cover_args_t
{
   smr_prim_t *prims;
   uint32_t n_prims;
   const smr_edge_t *edges;
   const uint32_t *mask;
   uint32_t *alpha;           /* per strip of four rows and column: the rows' alpha, one byte each */
   uint32_t alpha_cap;
   uint2 *drows;              /* per (primitive, strip): first alpha word, smr_strip_pack(first column, words) */
   uint32_t drow_cap;
   uint32_t *ctr;             /* [0] next primitive [1] alpha words taken [2] strips taken [3] primitives done [4] primitives left to k_raster */
   int band_y0, band_h;
   uint32_t chunk;            /* primitives a wavefront takes from the queue with one atomic: SMR_CV_CHUNK on a full chart, fewer when the rule is small */
   uint32_t linear;           /* fills that pass the star test take the crossing-by-crossing path (SMR_HIP_LINEAR=0: none does) */
   uint32_t split;            /* a rule of few primitives: every primitive's strips in `split` parts, a wavefront each (a wavefront alone on its
                                 SIMD needs 200 us and more for a polygon of 100 rows, and the pass waits for the slowest); the strip table's
                                 entries were handed out by k_prim_setup then (drow_off), and the part that finishes last sets SMR_PF_DELTA */
};

struct cover_lds_t
{
   int cell[SMR_TH][SMR_CV_WC];
   uint16_t bucket[SMR_NSUB][SMR_BKS];
   int redge[SMR_TH][SMR_CV_SL][5];
   int gy[SMR_CV_MAXGROUPS][2];      /* sub-row extent of each group of 64 edge slots */
   int own[64];
};

/* the linear path of k_cover keeps the cells of a whole chunk of rows where the strip path keeps four rows, its lists and
 * its row edges: cell, bucket and redge are contiguous (asserted), SMR_CV_POOL ints; the last SMR_CV_AUX of them hold, per row of
 * the chunk, the number of edges spanning it when it is an exact row and the first SMR_CV_C0 of their positions */
#define SMR_CV_POOL ((int) ((sizeof(int) * SMR_TH * SMR_CV_WC + sizeof(uint16_t) * SMR_NSUB * SMR_BKS + sizeof(int) * SMR_TH * SMR_CV_SL * 5) / sizeof(int)))
#define SMR_CV_C0 6
#define SMR_CV_ROWS 64
#define SMR_CV_AUX (SMR_CV_ROWS * (1 + SMR_CV_C0))
static_assert(offsetof(cover_lds_t, bucket) == sizeof(int) * SMR_TH * SMR_CV_WC && offsetof(cover_lds_t, redge) % 4 == 0 &&
              offsetof(cover_lds_t, gy) == sizeof(int) * SMR_CV_POOL, "the linear path's pool is cell + bucket + redge");

__device__ __forceinline__ int smr_wave_scan_max(int v)
{
   v = max(v, __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x111, 0xf, 0xf, false));
   v = max(v, __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x112, 0xf, 0xf, false));
   v = max(v, __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x114, 0xf, 0xf, false));
   v = max(v, __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x118, 0xf, 0xf, false));
   v = max(v, __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x142, 0xa, 0xf, false));
   v = max(v, __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x143, 0xc, 0xf, false));
   return v;
}

__global__ __launch_bounds__(64 * SMR_CV_WAVES) void k_cover(cover_args_t A)
{
   __shared__ cover_lds_t LL[SMR_CV_WAVES];
   const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   cover_lds_t &L = LL[wave];
   const int lane = threadIdx.x & 63;
   const unsigned long long lt_mask = (1ull << lane) - 1ull;
   const int my_row = lane / SMR_GRID_Y;
   smr_bucket_t<false> bk;
   bk.p = L.bucket[lane < SMR_NSUB ? lane : 0];
   const int BK = SMR_BK;
#pragma unroll
   for (int r = 0; r < SMR_TH; r++) { L.cell[r][lane] = 0; L.cell[r][lane + 64] = 0; }
   SMR_WAVE_SYNC();
   uint32_t dpos = 0, dleft = 0, rpos = 0, rleft = 0;      /* this wavefront's blocks of the two arenas */
   uint32_t n_done = 0, n_left = 0, n_lin = 0;
#ifdef SMR_CV_PROFILE
   unsigned long long cvp[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, cvt = __builtin_amdgcn_s_memtime();
#define CVTICK(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); cvp[i] += t_ - cvt; cvt = t_; } while (0)
#else
#define CVTICK(i) do { } while (0)
#endif

   for (;;)
   {
   uint32_t p0 = 0;
   if (lane == 0) p0 = atomicAdd(&A.ctr[0], A.chunk);
   p0 = __builtin_amdgcn_readfirstlane(p0);
   if (p0 >= A.n_prims * A.split) break;
   const uint32_t p1 = min(p0 + A.chunk, A.n_prims * A.split);
   for (uint32_t wi = p0; wi < p1; wi++)
   {
      const uint32_t p = wi / A.split, part = wi - p * A.split;
      const smr_prim_t P = A.prims[p];
      if (!(P.flags & SMR_PF_VALID) || P.edge_cnt == 0) continue;
      const int cx0 = (P.minx >> 8) - 1;                   /* cell 0 is a guard: the exact rows reach half a sub-row beyond an edge's ends */
      const int wp = (P.maxx >> 8) - cx0 + 2;
      const int dr0 = max(P.row0, A.band_y0), dr1 = min(P.row0 + P.nrows, A.band_y0 + A.band_h);
      if (dr1 <= dr0) continue;
      if (P.edge_cnt > SMR_CV_MAXSLOTS || wp > SMR_CV_WC) { if (part == 0) n_left++; continue; }
#ifdef SMR_CV_ONLY
      if (((P.flags & SMR_PF_PIECES) != 0) != (SMR_CV_ONLY == 2)) { n_left++; continue; }
#endif
      const int32_t X0 = cx0 * 256, X1 = X0 + SMR_CV_WC * 256;
      const smr_edge_t *E = A.edges + P.edge_off;
      const uint32_t ngroups = (P.edge_cnt + 63u) >> 6;
      const uint32_t wmask = (P.flags & SMR_PF_EVENODD) ? 1u : ~0u;
      const bool pieces = (P.flags & SMR_PF_PIECES) != 0;
      bool bad = false;                                    /* uniform */

      /* the primitive's strips in the strip table */
      const int cr_first = A.band_y0 + ((dr0 - A.band_y0) & ~(SMR_TH - 1));
      const uint32_t nstrips = (uint32_t) ((dr1 - 1 - cr_first) >> 2) + 1u;
      /* this wavefront's part of them: rows [part_r0, part_r1) */
      const int part_r0 = cr_first + SMR_TH * (int) ((uint64_t) nstrips * part / A.split);
      const int part_r1 = min(dr1, cr_first + SMR_TH * (int) ((uint64_t) nstrips * (part + 1u) / A.split));
      uint32_t rbase;
      if (A.split > 1u) rbase = P.drow_off;
      else if (nstrips > rleft)
      {
         const uint32_t take = nstrips > SMR_CV_RBLOCK / 4 ? nstrips : SMR_CV_RBLOCK;
         uint32_t b = 0;
         if (lane == 0) b = atomicAdd(&A.ctr[2], take);
         b = __builtin_amdgcn_readfirstlane(b);
         if ((uint64_t) b + take > A.drow_cap) { n_left++; continue; }
         if (take == nstrips) rbase = b;
         else { rpos = b; rleft = take; rbase = rpos; rpos += nstrips; rleft -= nstrips; }
      }
      else { rbase = rpos; rpos += nstrips; rleft -= nstrips; }

      /* one group of 64 slots stays in registers; of several, each group's sub-row extent decides which strips read it */
      smr_edge_t e0; e0.x1 = e0.y1 = e0.x2 = e0.y2 = 0;
      if ((uint32_t) lane < P.edge_cnt) e0 = E[lane];
      if (ngroups > 1)
      {
         for (uint32_t g = 0; g < ngroups; g++)
         {
            const uint32_t idx = g * 64u + lane;
            smr_edge_t e; e.x1 = e.y1 = e.x2 = e.y2 = 0;
            if (g == 0) e = e0; else if (idx < P.edge_cnt) e = E[idx];
            const smr_oedge_t o = smr_orient(e);
            const int lo = smr_wave_min(o.dir != 0 ? o.ytop : INT32_MAX), hi = smr_wave_max(o.dir != 0 ? o.ybot : INT32_MIN);
            if (lane == 0) { L.gy[g][0] = lo; L.gy[g][1] = hi; }
         }
         SMR_WAVE_SYNC();
      }

      /* ---- cells -> alpha (k_raster's flush: prefix sum of cover, area, libcairo's extents), the four rows of a column in
       * one word; lane = column, two sweeps for a primitive wider than 64 cells; then the strip's words and its table entry.
       * cells = the strip's first row, rstride ints from row to row, ncell of them per row ---- */
      auto emit_strip = [&](const int crow, int *const cells, const int rstride, const int ncell, const uint32_t rows)
      {
         uint32_t a4lo = 0u, a4hi = 0u;
#pragma unroll
         for (int r = 0; r < SMR_TH; r++)
         {
            if (!((rows >> r) & 1u)) continue;
            int *const crowp = cells + r * rstride;
            int carry = 0;
            auto sweep = [&](const int i, uint32_t &a4)
            {
               const int v = i < ncell ? crowp[i] : 0;
               if (!__ballot(v != 0) && carry == 0) return;
               if (v) crowp[i] = 0;
               int c = smr_delta_cover(v);
               const int ua = smr_delta_area(v);
               c = (__ballot(c != 0) ? smr_wave_scan(c) : 0) + carry;
               carry = __builtin_amdgcn_readlane(c, 63);
               const int area = c * 512 - ua;
               int a = (area + (area << 4) + 256) >> 9;
               a = a < 0 ? 0 : a > 255 ? 255 : a;
               if (i < 1 || i > wp - 2) a = 0;           /* libcairo composites inside the primitive's pixel extents only */
               a4 |= (uint32_t) a << (8 * r);
            };
            sweep(lane, a4lo);
            if (wp > 64) sweep(lane + 64, a4hi);
         }
         {
            const unsigned long long m0 = __ballot(a4lo != 0), m1 = __ballot(a4hi != 0);
            uint2 entry = make_uint2(0u, 0u);
            if ((m0 | m1) && !bad)
            {
               const int xlo = m0 ? __ffsll((long long) m0) - 1 : 63 + __ffsll((long long) m1);
               const int xhi = m1 ? 127 - __clzll((long long) m1) : 63 - __clzll((long long) m0);
               const uint32_t n = (uint32_t) (xhi - xlo + 1);
               uint32_t abase = 0;
               if (n > dleft)
               {
                  uint32_t b = 0;
                  if (lane == 0) b = atomicAdd(&A.ctr[1], SMR_CV_DBLOCK);
                  b = __builtin_amdgcn_readfirstlane(b);
                  if ((uint64_t) b + SMR_CV_DBLOCK > A.alpha_cap) bad = true;
                  else { dpos = b; dleft = SMR_CV_DBLOCK; }
               }
               if (!bad)
               {
                  abase = dpos; dpos += n; dleft -= n;
                  if (lane >= xlo && lane <= xhi) A.alpha[abase + (uint32_t) (lane - xlo)] = a4lo;
                  if (lane + 64 >= xlo && lane + 64 <= xhi) A.alpha[abase + (uint32_t) (lane + 64 - xlo)] = a4hi;
                  entry = make_uint2(abase, smr_strip_pack(cx0 + xlo, (int) n));
               }
            }
            if (!bad && lane == 0) A.drows[rbase + (uint32_t) ((crow - cr_first) >> 2)] = entry;
         }
      };


      /* ---- fills that are simple polygons: every crossing counts by itself ----
       * The scan converter's spans are a function of the winding number along a sample line.  Where that number is only ever 0
       * or s (one sign for the whole polygon: a simple polygon), a crossing whose direction is s starts a span and any other
       * ends one, whatever lies left of it: the sorted lists, the walk and the per-row order of the exact rows fall away, and
       * (edge, row) pairs on the lanes add their crossings straight to the cells -- the sub-rows of a pair one after the other
       * in its lane, x by the scan converter's own DDA (quotient and remainder of one sub-row's step).  Libcairo's two rules
       * that look at neighbours hold by themselves: spans that touch merge (their two crossings cancel in one cell), and an
       * exact row's span that ends where the next begins continues -- which the sums do not reproduce, so a row with more than
       * two spanning edges must have them at different positions, else the primitive takes the strip path below.
       * Simple is tested as star-shaped: every edge turns the same way about the box's centre and the boundary goes round it
       * once (two edges straddle the centre's row).  Polygons that are not (an L, a river) take the strip path; so do
       * primitives of more than 64 edge slots (the cells' packed fields hold 64 crossings of one cell and sub-row). */
      bool linear_done = false;
      if (A.linear && !pieces && ngroups == 1)
      {
         const int ccx = (int) (((int64_t) P.minx + P.maxx) >> 1), ccy = (int) (((int64_t) P.miny + P.maxy) >> 1);
         const int ax = e0.x1 - ccx, ay = e0.y1 - ccy, bx = e0.x2 - ccx, by = e0.y2 - ccy;
         const bool real = (uint32_t) lane < P.edge_cnt && (e0.x1 != e0.x2 || e0.y1 != e0.y2);
         const bool inr = ((uint32_t) (ax + 32768) | (uint32_t) (ay + 32768) | (uint32_t) (bx + 32768) | (uint32_t) (by + 32768)) < 65536u;
         const int crs = ax * by - bx * ay;                  /* |.| < 2^31 whenever inr */
         const unsigned long long m_real = __ballot(real), m_pos = __ballot(real && crs > 0), m_neg = __ballot(real && crs < 0);
         const unsigned long long m_out = __ballot(real && !inr);
         const unsigned long long m_str = __ballot(real && ((ay <= 0 && by > 0) || (by <= 0 && ay > 0)));
         const bool star = m_real && !m_out && (m_pos == m_real || m_neg == m_real) && __popcll(m_str) == 2;
         if (star)
         {
            const int sg = m_pos == m_real ? -1 : 1;         /* a crossing of direction sg opens a span */
            const smr_oedge_t o = smr_orient(e0);
            int *const pool = &L.cell[0][0];
            int *const rowcnt = pool + (SMR_CV_POOL - SMR_CV_AUX), *const c0l = rowcnt + SMR_CV_ROWS;
            const int stride = wp;
            int R = ((SMR_CV_POOL - SMR_CV_AUX) / stride) & ~(SMR_TH - 1);
            if (R > SMR_CV_ROWS) R = SMR_CV_ROWS;
            /* one sub-row's step of x: floor(1024 dxs / D) and the remainder */
            int qs = 0, rs = 0;
            if (o.dir != 0 && o.dxs != 0)
            {
               const int n = 1024 * o.dxs;
               qs = (int) floor((double) n * smr_rcp((double) o.D));
               rs = n - qs * o.D;
               if (rs < 0) { qs--; rs += o.D; } else if (rs >= o.D) { qs++; rs -= o.D; }
            }
            /* the row mask, a word per lane (a star inside +-128 px has at most 8) */
            uint32_t mw = 0;
            if ((uint32_t) lane < (((uint32_t) P.nrows + 31u) >> 5)) mw = A.mask[P.mask_off + lane];
            bool fail = false;
            for (int c0r = part_r0; c0r < part_r1 && !bad && !fail; c0r += R)
            {
               const int c1r = min(c0r + R, part_r1), nr = c1r - c0r;
               for (int i = 2 * lane; i < nr * stride; i += 128) *reinterpret_cast<int2 *>(pool + i) = make_int2(0, 0);
               rowcnt[lane] = 0;
               SMR_WAVE_SYNC();
               const int ys_lo = max(o.ytop, max(c0r, dr0) * SMR_GRID_Y), ys_hi = min(o.ybot, c1r * SMR_GRID_Y);
               const int ra = ys_lo / SMR_GRID_Y;
               const int np = (o.dir != 0 && ys_hi > ys_lo) ? (ys_hi - 1) / SMR_GRID_Y - ra + 1 : 0;
               const int incl = smr_wave_scan(np), excl = incl - np;
               const int total = __builtin_amdgcn_readlane(incl, 63);
               bool oob = false;
               for (int base = 0; base < total; base += 64)
               {
                  L.own[lane] = INT32_MIN;
                  SMR_WAVE_SYNC();
                  if (np > 0)
                  {
                     const int first = excl - base;
                     if (first >= 0 && first < 64) L.own[first] = lane;
                     else if (first < 0 && incl > base) L.own[0] = lane;
                  }
                  SMR_WAVE_SYNC();
                  const int own = smr_wave_scan_max(L.own[lane]);
                  const bool valid = base + lane < total;
                  const int ad = (valid ? own : lane) << 2;
                  const int s_p1x = __builtin_amdgcn_ds_bpermute(ad, o.p1x), s_p1y = __builtin_amdgcn_ds_bpermute(ad, o.p1y);
                  const int s_dxs = __builtin_amdgcn_ds_bpermute(ad, o.dxs), s_D = __builtin_amdgcn_ds_bpermute(ad, o.D);
                  const int s_ytop = __builtin_amdgcn_ds_bpermute(ad, o.ytop), s_ybot = __builtin_amdgcn_ds_bpermute(ad, o.ybot);
                  const int s_qs = __builtin_amdgcn_ds_bpermute(ad, qs), s_rs = __builtin_amdgcn_ds_bpermute(ad, rs);
                  const int sgn = __builtin_amdgcn_ds_bpermute(ad, o.dir) * sg;
                  const int row = __builtin_amdgcn_ds_bpermute(ad, ra) + (base + lane - __builtin_amdgcn_ds_bpermute(ad, excl));
                  const int rr = row - P.row0, rs0 = row * SMR_GRID_Y;
                  const uint32_t mword = (uint32_t) __builtin_amdgcn_ds_bpermute(valid ? (rr >> 5) << 2 : 0, (int) mw);
                  const bool sub = (mword >> (rr & 31)) & 1u;
                  int *const crow = pool + (row - c0r) * stride;
                  if (valid && sub)
                  {
                     const int s0 = max(s_ytop, rs0), s1 = min(s_ybot, rs0 + SMR_GRID_Y);
                     int32_t q, rem;
                     smr_edge_qr(256 - 30ll * s_p1y, 2 * s_dxs, s_D, smr_rcp((double) s_D), s0, q, rem);
                     const int half = s_D >> 1;
                     for (int ys = s0; ys < s1; ys++)
                     {
                        const int rel = s_p1x + q + (rem >= half && s_dxs != 0 ? 1 : 0) - X0;
                        atomicAdd(&crow[rel >> 8], smr_delta_pack(sgn, sgn * 2 * (rel & 255)));
                        q += s_qs; rem += s_rs;
                        if (rem >= s_D) { q++; rem -= s_D; }
                     }
                  }
                  else if (valid && s_ytop <= rs0 && s_ybot >= rs0 + SMR_GRID_Y)
                  {
                     /* an exact row: libcairo's cell_list_render_edge for this edge, as in the strip path below */
                     int32_t q1, r1, q2, r2;
                     const int64_t ec = 256 - 30ll * s_p1y;
                     const double inv = smr_rcp((double) s_D);
                     smr_edge_qr(ec, 2 * s_dxs, s_D, inv, rs0, q1, r1);
                     smr_edge_qr(ec, 2 * s_dxs, s_D, inv, rs0 + SMR_GRID_Y, q2, r2);
                     const int64_t nn = (int64_t) s_dxs * 1024;
                     int64_t dq = (int64_t) trunc((double) nn * inv);
                     int64_t dr = nn - dq * s_D;
                     if (nn >= 0) { if (dr < 0) { dq--; dr += s_D; } else if (dr >= s_D) { dq++; dr -= s_D; } }
                     else { if (dr > 0) { dq++; dr -= s_D; } else if (dr <= -(int64_t) s_D) { dq--; dr += s_D; } }
                     const int32_t hq = (int32_t) (dq / 2), hr = (int32_t) (dr / 2);
                     q1 += s_p1x; q2 += s_p1x;
                     const int c0 = q1 + (r1 >= (s_D >> 1) ? 1 : 0);
                     q1 -= hq; r1 -= hr;
                     if (r1 < 0) { --q1; r1 += s_D; } else if (r1 >= s_D) { ++q1; r1 -= s_D; }
                     q2 -= hq; r2 -= hr;
                     if (r2 < 0) { --q2; r2 += s_D; } else if (r2 >= s_D) { ++q2; r2 -= s_D; }
                     q1 -= X0; q2 -= X0;
                     const int pos = atomicAdd(&rowcnt[row - c0r], 1);
                     if (pos < SMR_CV_C0) c0l[(row - c0r) * SMR_CV_C0 + pos] = c0;
                     int ix1 = q1 >> 8, ix2 = q2 >> 8, fx1 = q1 & 255, fx2 = q2 & 255;
                     if (min(ix1, ix2) < 0 || max(ix1, ix2) >= stride) oob = true;
                     else if (ix1 == ix2) atomicAdd(&crow[ix1], smr_delta_pack(sgn * SMR_GRID_Y, sgn * (fx1 + fx2) * SMR_GRID_Y));
                     else
                     {
                        const bool sw = ix2 < ix1;
                        const int32_t qa = sw ? q2 : q1, rra = sw ? r2 : r1, qb = sw ? q1 : q2, rrb = sw ? r1 : r2;
                        if (sw) { int t = ix1; ix1 = ix2; ix2 = t; t = fx1; fx1 = fx2; fx2 = t; }
                        const int64_t dx = (int64_t) (qb - qa) * s_D + (rrb - rra);
                        const int64_t tmp0 = ((int64_t) (256 - fx1) * s_D - rra) * SMR_GRID_Y;
                        const int64_t step = (int64_t) (SMR_GRID_Y * 256) * s_D;
                        const double idx = smr_rcp(smr_i64_to_f64(dx));
                        int ylast = 0;
                        for (int k = ix1; k <= ix2; k++)
                        {
                           int yhi = SMR_GRID_Y;
                           if (k != ix2)
                           {
                              const int64_t n = tmp0 + (int64_t) (k - ix1) * step;
                              yhi = (int) (smr_i64_to_f64(n) * idx);
                              const int64_t rm = n - (int64_t) yhi * dx;
                              if (rm < 0) yhi--; else if (rm >= dx) yhi++;
                           }
                           const int cov = yhi - ylast;
                           const int fac = k == ix1 ? 256 + fx1 : k == ix2 ? fx2 : 256;
                           if (cov) atomicAdd(&crow[k], smr_delta_pack(sgn * cov, sgn * cov * fac));
                           ylast = yhi;
                        }
                     }
                  }
               }
               SMR_WAVE_SYNC();
               /* exact rows: two spanning edges are a span by themselves; more must all stand at different positions */
               {
                  bool dup = false;
                  const int cnt = lane < nr ? rowcnt[lane] : 0;
                  if (cnt > SMR_CV_C0 || (cnt & 1)) dup = true;
                  else if (cnt > 2)
                     for (int i = 1; i < cnt; i++)
                        for (int j = 0; j < i; j++)
                           if (c0l[lane * SMR_CV_C0 + i] == c0l[lane * SMR_CV_C0 + j]) dup = true;
                  if (__ballot(dup || oob)) fail = true;
               }
               if (fail)
               {
                  for (int i = 2 * lane; i < nr * stride; i += 128) *reinterpret_cast<int2 *>(pool + i) = make_int2(0, 0);
                  SMR_WAVE_SYNC();
                  break;
               }
               for (int crow = c0r; crow < c1r && !bad; crow += SMR_TH)
               {
                  uint32_t rows = 0;
#pragma unroll
                  for (int r = 0; r < SMR_TH; r++) if (crow + r >= dr0 && crow + r < dr1) rows |= 1u << r;
                  emit_strip(crow, pool + (crow - c0r) * stride, stride, stride, rows);
               }
               SMR_WAVE_SYNC();
            }
            linear_done = !fail;
            if (linear_done && !bad && part == 0) n_lin++;         /* (a tally of primitives: the first part speaks for the others) */
         }
      }

      CVTICK(0);
      for (int crow = part_r0; crow < part_r1 && !bad && !linear_done; crow += SMR_TH)
      {
         /* the strip's rows: sub-sampled (bit of the row mask set), exact (clear), or not the primitive's */
         uint32_t partial, full_bits;
         {
            const int row = crow + lane;
            const bool inside = lane < SMR_TH && row >= dr0 && row < dr1;
            bool sub = false;
            if (inside) { const int rr = row - P.row0; sub = (A.mask[P.mask_off + ((uint32_t) rr >> 5)] >> (rr & 31)) & 1u; }
            partial = (uint32_t) __ballot(inside && sub); full_bits = (uint32_t) __ballot(inside && !sub);
         }
         const int ysbase = crow * SMR_GRID_Y;
         const int my_ys = ysbase + lane;
         uint32_t rowcnt[SMR_TH] = { 0, 0, 0, 0 };
         int cnt = 0, lw = 0, cl = 1, cr = 0;
         bool ovf = false, touched = false;
         auto push_interval = [&](int l, int r)
         {
            for (int i = 0; i < cnt;)
            {
               int ia = bk.get(2 * i), ib = bk.get(2 * i + 1);
               if (ia <= r && l <= ib)
               {
                  l = min(l, ia); r = max(r, ib);
                  cnt--;
                  bk.set(2 * i, bk.get(2 * cnt)); bk.set(2 * i + 1, bk.get(2 * cnt + 1));
                  i = 0;
               }
               else i++;
            }
            if (2 * cnt + 2 <= BK) { bk.set(2 * cnt, l); bk.set(2 * cnt + 1, r); cnt++; }
            else ovf = true;
         };
         /* the groups that meet the strip */
         uint32_t gm = 1u;
         if (ngroups > 1)
         {
            bool meets = false;
            if ((uint32_t) lane < ngroups) meets = L.gy[lane][0] < ysbase + SMR_NSUB && L.gy[lane][1] > ysbase;
            gm = (uint32_t) __ballot(meets);
         }
         CVTICK(1);
         struct { uint32_t edge_off; } it; it.edge_off = 0;
         const int tx = 0;
         (void) tx; (void) it; (void) lw; (void) touched;
         while (gm)
         {
            const uint32_t g = (uint32_t) __ffs((int) gm) - 1u;
            gm &= gm - 1u;
            smr_edge_t my_edge = e0;
            if (g) { my_edge.x1 = my_edge.y1 = my_edge.x2 = my_edge.y2 = 0; const uint32_t idx = g * 64u + lane; if (idx < P.edge_cnt) my_edge = E[idx]; }
            const smr_oedge_t o = smr_orient(my_edge);
            const int32_t emaxx = o.p1x + (o.dxs > 0 ? o.dxs : 0);
            const bool inband = o.dir != 0 && o.ytop < ysbase + SMR_NSUB && o.ybot > ysbase;
            it.edge_off = g * 64u;
            /* exact rows: who spans them */
            uint32_t fb = full_bits;
#ifdef SMR_CV_KO
            if (SMR_CV_KO & 16) fb = 0;
#endif
            while (fb)
            {
               const int r = __ffs((int) fb) - 1;
               fb &= fb - 1;
               const int rs0 = ysbase + r * SMR_GRID_Y;
               const bool act = o.dir != 0 && o.ytop <= rs0 && o.ybot >= rs0 + SMR_GRID_Y;
               const unsigned long long am = __ballot(act);
               if (!am) continue;
               const uint32_t cntr = r == 0 ? rowcnt[0] : r == 1 ? rowcnt[1] : r == 2 ? rowcnt[2] : rowcnt[3];
               const uint32_t pos = cntr + __popcll(am & lt_mask);
               if (act && pos < SMR_CV_SL) { int *re = L.redge[r][pos]; re[0] = o.p1x; re[1] = o.p1y; re[2] = o.dxs; re[3] = o.D; re[4] = o.dir; }
               if (cntr + __popcll(am) > SMR_CV_SL) bad = true;
#pragma unroll
               for (int k = 0; k < SMR_TH; k++) rowcnt[k] += k == r ? (uint32_t) __popcll(am) : 0u;
            }
            CVTICK(2);
#ifdef SMR_CV_KO
            if (SMR_CV_KO & 1) continue;
#endif
#define XTICK(i) do { } while (0)
#define XCOUNT(i) do { } while (0)
#define XCOUNTN(i, n) do { } while (0)
#include "raster_cross.inc"
#undef XTICK
#undef XCOUNT
#undef XCOUNTN
            CVTICK(3);
         }

         /* ---- sub-sampled rows: crossings -> spans -> cells; lane = sub-row (k_raster's flush) ---- */
#ifdef SMR_CV_KO
         if (SMR_CV_KO & 2) cnt = 0, cl = 1, cr = 0;
         if (SMR_CV_KO & 4) full_bits = 0;
#endif
         if (partial)
         {
            const bool mine = lane < SMR_NSUB && ((partial >> my_row) & 1u);
            auto add_span = [&](int x1, int x2)
            {
               if (x1 == x2) return;
               const int ix1 = x1 >> 8, ix2 = x2 >> 8, fx1 = x1 & 255, fx2 = x2 & 255;
               if (ix2 >= SMR_CV_WC) { ovf = true; return; }
               if (ix1 != ix2)
               {
                  atomicAdd(&L.cell[my_row][ix1], smr_delta_pack(1, 2 * fx1));
                  atomicAdd(&L.cell[my_row][ix2], smr_delta_pack(-1, -2 * fx2));
               }
               else atomicAdd(&L.cell[my_row][ix1], smr_delta_pack(0, 2 * (fx1 - fx2)));
            };
            if (mine && pieces)
            {
               if (cl <= cr)
               {
                  for (int i = 0; i < cnt;)
                  {
                     const int ia = bk.get(2 * i), ib = bk.get(2 * i + 1);
                     if (ia <= cr && cl <= ib)
                     {
                        cl = min(cl, ia); cr = max(cr, ib);
                        cnt--;
                        bk.set(2 * i, bk.get(2 * cnt)); bk.set(2 * i + 1, bk.get(2 * cnt + 1));
                        i = 0;
                     }
                     else i++;
                  }
                  add_span(cl, cr);
               }
               for (int i = 0; i < cnt; i++) add_span(bk.get(2 * i), bk.get(2 * i + 1));
            }
            else if (mine)
            {
               for (int i = 1; i < cnt; i++)
               {
                  const int v = bk.get(i);
                  int k = i - 1;
                  while (k >= 0 && (bk.get(k) >> 1) > (v >> 1)) { bk.set(k + 1, bk.get(k)); k--; }
                  bk.set(k + 1, v);
               }
               int winding = 0, xstart = -1;
               for (int i = 0; i < cnt; i++)
               {
                  const int bv = bk.get(i);
                  const int xend = bv >> 1;
                  winding += (bv & 1) ? 1 : -1;
                  if (((uint32_t) winding & wmask) == 0)
                  {
                     if (i + 1 < cnt && (bk.get(i + 1) >> 1) == xend) continue;
                     if (xstart >= 0) add_span(xstart, xend);
                     xstart = -1;
                  }
                  else if (xstart < 0) xstart = xend;
               }
            }
         }
         if (__ballot(ovf)) bad = true;
         SMR_WAVE_SYNC();
         CVTICK(4);

         /* ---- exact rows: lane = (row, edge slot), all four rows at once (k_raster's form of libcairo's cell_list_render_edge) ---- */
         if (full_bits && !bad)
         {
            const int grp = lane >> 4, es = lane & 15, row = grp;
            const uint32_t my_n = grp == 0 ? rowcnt[0] : grp == 1 ? rowcnt[1] : grp == 2 ? rowcnt[2] : rowcnt[3];
            uint32_t nmax = 0;
#pragma unroll
            for (int k = 0; k < SMR_TH; k++) if ((full_bits >> k) & 1u) nmax = max(nmax, rowcnt[k]);
            const bool onr = ((full_bits >> row) & 1u) && (uint32_t) es < my_n;
            const int rs0 = ysbase + row * SMR_GRID_Y;
            int32_t c0 = 0, c1 = 0, dir = 0, D = 1, q1 = 0, r1 = 0, q2 = 0, r2 = 0, dxs = 0;
            smr_oedge_t o; o.p1x = o.p1y = o.dxs = 0; o.D = 1; o.dir = 0;
            if (onr) { const int *re = L.redge[row][es]; o.p1x = re[0]; o.p1y = re[1]; o.dxs = re[2]; o.D = re[3]; o.dir = re[4]; }
            /* short edges (|dx|, dy < 16 px: every edge of a chart of small polygons) evaluate in 32 bits, as in the crossing loop */
            const bool allshort = !__ballot(onr && !(abs(o.dxs) < 4096 && o.D <= 60 * 4096));
            if (onr)
            {
               dxs = o.dxs; D = o.D; dir = o.dir;
               int32_t hq, hr;
               if (allshort)
               {
                  const float rf = __builtin_amdgcn_rcpf((float) o.D);
                  /* floor(n / D) and the remainder: truncating fp32 estimate within 1 of the floor, exact correction */
                  auto qr32 = [&](int n) -> int2
                  {
                     const int qe = (int) fmaf((float) n, rf, -0.5f);
                     const int rr = __mul24(qe, -o.D) + n;
                     const int adj = (rr >= o.D ? 1 : 0) + (rr >> 31);
                     return make_int2(qe + adj, rr - __mul24(adj, o.D));
                  };
                  const int c32 = (int) ((uint32_t) rs0 * 512u + 256u - 30u * (uint32_t) o.p1y);
                  const int n1 = __mul24(2 * o.dxs, c32);
                  const int2 A1 = qr32(n1), A2 = qr32(n1 + __mul24(2 * o.dxs, 512 * SMR_GRID_Y));
                  q1 = A1.x; r1 = A1.y; q2 = A2.x; r2 = A2.y;
                  const int nn = o.dxs * 1024;
                  const int2 A3 = qr32(abs(nn));
                  hq = nn < 0 ? -(A3.x >> 1) : (A3.x >> 1); hr = nn < 0 ? -(A3.y >> 1) : (A3.y >> 1);
               }
               else
               {
                  const int64_t ec = 256 - 30ll * o.p1y;
                  const double inv = smr_rcp((double) D);
                  smr_edge_qr(ec, 2 * dxs, D, inv, rs0, q1, r1);
                  smr_edge_qr(ec, 2 * dxs, D, inv, rs0 + SMR_GRID_Y, q2, r2);
                  const int64_t nn = (int64_t) dxs * 1024;
                  int64_t dq = (int64_t) trunc((double) nn * inv);
                  int64_t dr = nn - dq * D;
                  if (nn >= 0) { if (dr < 0) { dq--; dr += D; } else if (dr >= D) { dq++; dr -= D; } }
                  else { if (dr > 0) { dq++; dr -= D; } else if (dr <= -(int64_t) D) { dq--; dr += D; } }
                  hq = (int32_t) (dq / 2); hr = (int32_t) (dr / 2);
               }
               q1 += o.p1x; q2 += o.p1x;
               c0 = q1 + (r1 >= (D >> 1) ? 1 : 0); c1 = q2 + (r2 >= (D >> 1) ? 1 : 0);
               /* step back from the sample location (half a sub-row) to the row boundary */
               q1 -= hq; r1 -= hr;
               if (r1 < 0) { --q1; r1 += D; } else if (r1 >= D) { ++q1; r1 -= D; }
               q2 -= hq; r2 -= hr;
               if (r2 < 0) { --q2; r2 += D; } else if (r2 >= D) { ++q2; r2 -= D; }
               q1 -= X0; q2 -= X0;
            }
            /* position in the row's (cell, next cell) order, winding before me, neighbours' cells */
            int wsum = 0, pred_c0 = INT32_MIN, succ_c0 = INT32_MAX;
            bool has_pred = false, has_succ = false;
            for (uint32_t j = 0; j < nmax; j++)
            {
               const int src = (lane & 48) + (int) j;
               const int kc0 = __shfl(c0, src), kc1 = __shfl(c1, src), kdir = __shfl(dir, src);
               if (j < my_n && (int) j != es)
               {
                  const bool before = kc0 < c0 || (kc0 == c0 && (kc1 < c1 || (kc1 == c1 && (int) j < es)));
                  if (before) { wsum += kdir; pred_c0 = max(pred_c0, kc0); has_pred = true; }
                  else { succ_c0 = min(succ_c0, kc0); has_succ = true; }
               }
            }
            const int wincl = wsum + dir;
            /* every edge of the primitive is here, so libcairo's own rule applies: a span starts at the first edge after the
             * last span's end and ends where the winding returns to zero and the next edge lies in another cell, at the
             * row's last edge at the latest */
            const bool is_start = onr && (!has_pred || (((uint32_t) wsum & wmask) == 0 && pred_c0 != c0));
            const bool is_end = onr && !is_start && (!has_succ || (((uint32_t) wincl & wmask) == 0 && succ_c0 != c0));
            /* the cells every span-boundary edge crosses: one lane per (edge, cell) (an edge crossing many cells kept its whole
             * wavefront waiting while it walked them) */
            bool oob = false;
            int ncell = 0, sgn = 0, ix1 = 0, ix2 = 0, fx1 = 0, fx2 = 0;
            int64_t dx = 1, tmp0 = 0;
            if (is_start || is_end)
            {
               sgn = is_start ? 1 : -1;
               ix1 = (int) (q1 >> 8); ix2 = (int) (q2 >> 8); fx1 = q1 & 255; fx2 = q2 & 255;
               const bool sw = ix2 < ix1;
               const int32_t qa = sw ? q2 : q1, ra = sw ? r2 : r1, qb = sw ? q1 : q2, rb = sw ? r1 : r2;
               if (sw) { int t = ix1; ix1 = ix2; ix2 = t; t = fx1; fx1 = fx2; fx2 = t; }
               if (ix1 < 0 || ix2 >= SMR_CV_WC) oob = true;
               else
               {
                  ncell = ix2 - ix1 + 1;
                  if (ncell > 1) { dx = (int64_t) (qb - qa) * D + (rb - ra); tmp0 = ((int64_t) (256 - fx1) * D - ra) * SMR_GRID_Y; }
               }
            }
            if (__ballot(oob)) bad = true;
            const int incl = smr_wave_scan(ncell), excl = incl - ncell;
            const int total = bad ? 0 : __builtin_amdgcn_readlane(incl, 63);
            for (int base = 0; base < total; base += 64)
            {
               L.own[lane] = INT32_MIN;
               SMR_WAVE_SYNC();
               if (ncell > 0)
               {
                  const int first = excl - base;
                  if (first >= 0 && first < 64) L.own[first] = lane;
                  else if (first < 0 && incl > base) L.own[0] = lane;
               }
               SMR_WAVE_SYNC();
               const int own = smr_wave_scan_max(L.own[lane]);
               const bool valid = base + lane < total;
               const int src = valid ? own : lane, ad = src << 2;
               const int o_ix1 = __builtin_amdgcn_ds_bpermute(ad, ix1), o_ix2 = __builtin_amdgcn_ds_bpermute(ad, ix2);
               const int o_fx1 = __builtin_amdgcn_ds_bpermute(ad, fx1), o_fx2 = __builtin_amdgcn_ds_bpermute(ad, fx2);
               const int o_sgn = __builtin_amdgcn_ds_bpermute(ad, sgn), o_D = __builtin_amdgcn_ds_bpermute(ad, D);
               const int o_excl = __builtin_amdgcn_ds_bpermute(ad, excl);
               const int64_t o_dx = ((int64_t) __builtin_amdgcn_ds_bpermute(ad, (int) (dx >> 32)) << 32) | (uint32_t) __builtin_amdgcn_ds_bpermute(ad, (int) (uint32_t) dx);
               const int64_t o_tmp0 = ((int64_t) __builtin_amdgcn_ds_bpermute(ad, (int) (tmp0 >> 32)) << 32) | (uint32_t) __builtin_amdgcn_ds_bpermute(ad, (int) (uint32_t) tmp0);
               const int o_row = src >> 4;
               const int k = o_ix1 + (base + lane - o_excl);
               const bool multi = valid && o_ix1 != o_ix2;
               const int64_t step = (int64_t) (SMR_GRID_Y * 256) * o_D;
               const double inv = smr_rcp(smr_i64_to_f64(multi ? o_dx : 1));
               /* y (in sub-rows) at which the edge crosses the right boundary of cell `cell` */
               auto Y = [&](int cell) -> int
               {
                  const int64_t n = o_tmp0 + (int64_t) (cell - o_ix1) * step;
                  int y = (int) (smr_i64_to_f64(n) * inv);
                  int64_t rem = n - (int64_t) y * o_dx;
                  if (rem < 0) y--; else if (rem >= o_dx) y++;
                  return y;
               };
               const int yhi = (!multi || k == o_ix2) ? SMR_GRID_Y : Y(k);
               int ylo = __shfl_up(yhi, 1);                 /* the lane before holds the cell before, except at a batch's start */
               if (multi && k > o_ix1 && lane == 0) ylo = Y(k - 1);
               if (!multi || k == o_ix1) ylo = 0;
               if (valid)
               {
                  const int cov = yhi - ylo;
                  const int fac = !multi ? o_fx1 + o_fx2 : k == o_ix1 ? 256 + o_fx1 : k == o_ix2 ? o_fx2 : 256;
                  if (cov) atomicAdd(&L.cell[o_row][k], smr_delta_pack(o_sgn * cov, o_sgn * cov * fac));
               }
            }
         }
         SMR_WAVE_SYNC();

         CVTICK(5);
         emit_strip(crow, &L.cell[0][0], SMR_CV_WC, SMR_CV_WC, partial | full_bits);
         SMR_WAVE_SYNC();
         CVTICK(6);
      }
      if (bad)
      {
         /* cells of the strip that failed may be left */
#pragma unroll
         for (int r = 0; r < SMR_TH; r++) { L.cell[r][lane] = 0; L.cell[r][lane + 64] = 0; }
         SMR_WAVE_SYNC();
         n_left++;
         continue;
      }
      if (bad) { n_left++; continue; }
      if (A.split > 1u)
      {
         /* the parts that came through are counted in the primitive's spare word; the one that completes the count sets the flag
          * (a part that did not come through adds nothing: the primitive stays k_raster's) */
         uint32_t before = 0;
         if (lane == 0) before = atomicAdd(&A.prims[p].pad_[0], 1u);
         if ((uint32_t) __builtin_amdgcn_readfirstlane(before) + 1u != A.split) continue;
         if (lane == 0) A.prims[p].flags = P.flags | SMR_PF_DELTA;
      }
      else if (lane == 0) { A.prims[p].flags = P.flags | SMR_PF_DELTA; A.prims[p].drow_off = rbase; }
      n_done++;
   }
   }
   if (lane == 0) { if (n_done) atomicAdd(&A.ctr[3], n_done); if (n_left) atomicAdd(&A.ctr[4], n_left); if (n_lin) atomicAdd(&A.ctr[5], n_lin); }
#ifdef SMR_CV_PROFILE
   if (lane == 0) for (int i = 0; i < 8; i++) atomicAdd((unsigned long long *) (A.ctr + 6) + i, cvp[i]);
#endif
}

/* ------------------------------------------------------------------------------------ */
/* host side                                                                             */
/* ------------------------------------------------------------------------------------ */
static uint32_t host_color_pixel(uint32_t c) { return smr_color_pixel(c); }

/* 0xTTRRGGBB -> the premultiplied ARGB32 pixel libcairo makes of cairo_set_source_rgba (src/smrules_cairo.c:271-274) */
uint32_t smr_color_pixel(uint32_t c)
{
   double r = (double) ((c >> 16) & 0xff) / 255.0, g = (double) ((c >> 8) & 0xff) / 255.0, b = (double) (c & 0xff) / 255.0;
   double a = 1.0 - (double) (((c & 0x7f000000u) >> 23) & 0xff) / 255.0;
   auto d2s = [](double d) { return (uint32_t) (uint16_t) (d * 65535.0 + 0.5); };      /* _cairo_color_double_to_short of libcairo 1.16 */
   return ((d2s(a) >> 8) << 24) | ((d2s(r * a) >> 8) << 16) | ((d2s(g * a) >> 8) << 8) | (d2s(b * a) >> 8);
}

__global__ void k_fill_u32(uint32_t *p, size_t n, uint32_t v)
{
   size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t) gridDim.x * blockDim.x;
   for (; i < n; i += stride) p[i] = v;
}

extern "C" int smr_hip_canvas_create(int W, int H, int dpi, uint32_t bgcolor, int band_y0, int band_h, smr_canvas_t **out)
{
   SMR_NEED_INIT();
   if (!out || W <= 0 || H <= 0 || dpi <= 0 || band_y0 < 0 || band_h <= 0 || band_y0 + band_h > H || W > 65536 || H > 65536)
      return smr_fail(SMR_E_INVAL, "canvas_create: bad geometry %dx%d band %d+%d dpi %d", W, H, band_y0, band_h, dpi);
   smr_canvas_t *c = new smr_canvas_t();
   c->W = W; c->H = H; c->dpi = dpi; c->band_y0 = band_y0; c->band_h = band_h; c->bg = bgcolor;
   size_t n = (size_t) W * band_h;
   if (hipMalloc(&c->pix, n * 4) != hipSuccess) { delete c; return smr_fail(SMR_E_NOMEM, "canvas_create: hipMalloc %zu bytes", n * 4); }
   /* cairo_paint of the background onto the clear image == SOURCE */
   k_fill_u32<<<2048, 256, 0, smr_g.stream>>>(c->pix, n, host_color_pixel(bgcolor));
   SMR_HIP(hipGetLastError());
   *out = c;
   return SMR_OK;
}

extern "C" int smr_hip_canvas_clone(smr_canvas_t *c, smr_canvas_t **out)
{
   SMR_NEED_INIT();
   if (!c || !out) return smr_fail(SMR_E_INVAL, "canvas_clone: NULL");
   smr_canvas_t *n = new smr_canvas_t();
   n->W = c->W; n->H = c->H; n->dpi = c->dpi; n->band_y0 = c->band_y0; n->band_h = c->band_h; n->bg = c->bg;
   n->own_y0 = c->own_y0; n->own_h = c->own_h;
   size_t bytes = (size_t) c->W * c->band_h * 4;
   if (hipMalloc(&n->pix, bytes) != hipSuccess) { delete n; return smr_fail(SMR_E_NOMEM, "canvas_clone: hipMalloc %zu bytes", bytes); }
   if (hipMemcpyAsync(n->pix, c->pix, bytes, hipMemcpyDeviceToDevice, smr_g.stream) != hipSuccess)
   {
      smr_hip_canvas_destroy(n);
      return smr_fail(SMR_E_HIP, "canvas_clone: copy of %zu bytes failed", bytes);
   }
   *out = n;
   return SMR_OK;
}

extern "C" int smr_hip_canvas_clear(smr_canvas_t *c)
{
   SMR_NEED_INIT();
   if (!c) return smr_fail(SMR_E_INVAL, "canvas_clear: NULL");
   k_fill_u32<<<2048, 256, 0, smr_g.stream>>>(c->pix, (size_t) c->W * c->band_h, host_color_pixel(c->bg));
   SMR_HIP(hipGetLastError());
   return SMR_OK;
}

extern "C" int smr_hip_canvas_destroy(smr_canvas_t *c)
{
   if (!c) return SMR_OK;
   hipStreamSynchronize(smr_g.stream);
   if (c->pix) hipFree(c->pix);
   if (c->d_win_shift) hipFree(c->d_win_shift);
   if (c->d_win_box) hipFree(c->d_win_box);
   if (c->d_win_clip) hipFree(c->d_win_clip);
   if (c->d_win_lim) hipFree(c->d_win_lim);
   if (c->d_grid_off) hipFree(c->d_grid_off);
   if (c->d_grid_win) hipFree(c->d_grid_win);
   if (c->halo_above) hipFree(c->halo_above);
   if (c->halo_below) hipFree(c->halo_below);
   for (auto &b : c->buf) if (b.p) hipFree(b.p);
   delete c;
   return SMR_OK;
}

extern "C" int smr_hip_canvas_trim(smr_canvas_t *c)
{
   SMR_NEED_INIT();
   if (!c) return smr_fail(SMR_E_INVAL, "canvas_trim: NULL");
   SMR_HIP(hipStreamSynchronize(smr_g.stream));
   for (auto &b : c->buf) { if (b.p) hipFree(b.p); b.p = nullptr; b.cap = 0; }
   return SMR_OK;
}

extern "C" int smr_hip_canvas_read(smr_canvas_t *c, int x, int y, int w, int h, uint32_t *host_dst)
{
   SMR_NEED_INIT();
   if (!c || !host_dst || x < 0 || y < 0 || w <= 0 || h <= 0 || x + w > c->W || y + h > c->band_h)
      return smr_fail(SMR_E_INVAL, "canvas_read: rectangle outside the band");
   SMR_HIP(hipMemcpy2DAsync(host_dst, (size_t) w * 4, c->pix + (size_t) y * c->W + x, (size_t) c->W * 4,
                            (size_t) w * 4, h, hipMemcpyDeviceToHost, smr_g.stream));
   SMR_HIP(hipStreamSynchronize(smr_g.stream));
   return SMR_OK;
}

extern "C" int smr_hip_canvas_set_halo(smr_canvas_t *c, const uint32_t *above, int n_above, const uint32_t *below, int n_below)
{
   SMR_NEED_INIT();
   if (!c || n_above < 0 || n_below < 0 || (n_above && !above) || (n_below && !below) || n_above > c->band_y0 ||
       n_below > c->H - c->band_y0 - c->band_h)
      return smr_fail(SMR_E_INVAL, "canvas_set_halo: bad strip (above %d, below %d rows)", n_above, n_below);
   SMR_HIP(hipStreamSynchronize(smr_g.stream));
   if (c->halo_above) { hipFree(c->halo_above); c->halo_above = nullptr; }
   if (c->halo_below) { hipFree(c->halo_below); c->halo_below = nullptr; }
   c->n_above = c->n_below = 0;
   if (n_above)
   {
      size_t bytes = (size_t) n_above * c->W * 4;
      if (hipMalloc(&c->halo_above, bytes) != hipSuccess) return smr_fail(SMR_E_NOMEM, "canvas_set_halo: hipMalloc %zu bytes", bytes);
      SMR_HIP(hipMemcpy(c->halo_above, above, bytes, hipMemcpyHostToDevice));
      c->n_above = n_above;
   }
   if (n_below)
   {
      size_t bytes = (size_t) n_below * c->W * 4;
      if (hipMalloc(&c->halo_below, bytes) != hipSuccess) return smr_fail(SMR_E_NOMEM, "canvas_set_halo: hipMalloc %zu bytes", bytes);
      SMR_HIP(hipMemcpy(c->halo_below, below, bytes, hipMemcpyHostToDevice));
      c->n_below = n_below;
   }
   return SMR_OK;
}

/* the same with the strips already in device memory (what a device-to-device exchange over RCCL delivers): no host round trip */
extern "C" int smr_hip_canvas_set_halo_device(smr_canvas_t *c, const uint32_t *d_above, int n_above, const uint32_t *d_below, int n_below)
{
   SMR_NEED_INIT();
   if (!c || n_above < 0 || n_below < 0 || (n_above && !d_above) || (n_below && !d_below) || n_above > c->band_y0 ||
       n_below > c->H - c->band_y0 - c->band_h)
      return smr_fail(SMR_E_INVAL, "canvas_set_halo_device: bad strip (above %d, below %d rows)", n_above, n_below);
   hipStream_t s = smr_g.stream;
   SMR_HIP(hipStreamSynchronize(s));
   if (c->halo_above) { hipFree(c->halo_above); c->halo_above = nullptr; }
   if (c->halo_below) { hipFree(c->halo_below); c->halo_below = nullptr; }
   c->n_above = c->n_below = 0;
   if (n_above)
   {
      const size_t bytes = (size_t) n_above * c->W * 4;
      if (hipMalloc(&c->halo_above, bytes) != hipSuccess) return smr_fail(SMR_E_NOMEM, "canvas_set_halo_device: hipMalloc %zu bytes", bytes);
      SMR_HIP(hipMemcpyAsync(c->halo_above, d_above, bytes, hipMemcpyDeviceToDevice, s));
      c->n_above = n_above;
   }
   if (n_below)
   {
      const size_t bytes = (size_t) n_below * c->W * 4;
      if (hipMalloc(&c->halo_below, bytes) != hipSuccess) return smr_fail(SMR_E_NOMEM, "canvas_set_halo_device: hipMalloc %zu bytes", bytes);
      SMR_HIP(hipMemcpyAsync(c->halo_below, d_below, bytes, hipMemcpyDeviceToDevice, s));
      c->n_below = n_below;
   }
   SMR_HIP(hipStreamSynchronize(s));      /* the caller may reuse its strips */
   return SMR_OK;
}

/* band + halo strips as ONE canvas band [band_y0 - n_above, band_y0 + band_h + n_below): the rule-private background copy of a
 * banded caption rule (act_cap's auto_sfc, src/smrules_cairo.c:2455-2460).  Rectangles of finished captions (:2376-2382), the
 * neighbours' included, are drawn onto it like onto any band, halo rows and all; smr_hip_caprot on it scans the captions
 * whose anchor lies in the ORIGINAL band's rows, which is what its halo check then looks at. */
extern "C" int smr_hip_canvas_extend(smr_canvas_t *c, smr_canvas_t **out)
{
   SMR_NEED_INIT();
   if (!c || !out) return smr_fail(SMR_E_INVAL, "canvas_extend: NULL");
   smr_canvas_t *n = new smr_canvas_t();
   n->W = c->W; n->H = c->H; n->dpi = c->dpi; n->bg = c->bg;
   n->band_y0 = c->band_y0 - c->n_above; n->band_h = c->band_h + c->n_above + c->n_below;
   n->own_y0 = c->own_y0 >= 0 ? c->own_y0 : c->band_y0; n->own_h = c->own_y0 >= 0 ? c->own_h : c->band_h;
   const size_t row = (size_t) c->W * 4;
   if (hipMalloc(&n->pix, row * n->band_h) != hipSuccess) { delete n; return smr_fail(SMR_E_NOMEM, "canvas_extend: hipMalloc %zu bytes", row * n->band_h); }
   hipStream_t s = smr_g.stream;
   bool ok = true;
   if (c->n_above) ok = hipMemcpyAsync(n->pix, c->halo_above, row * c->n_above, hipMemcpyDeviceToDevice, s) == hipSuccess;
   ok = ok && hipMemcpyAsync(n->pix + (size_t) c->n_above * c->W, c->pix, row * c->band_h, hipMemcpyDeviceToDevice, s) == hipSuccess;
   if (ok && c->n_below) ok = hipMemcpyAsync(n->pix + (size_t) (c->n_above + c->band_h) * c->W, c->halo_below, row * c->n_below, hipMemcpyDeviceToDevice, s) == hipSuccess;
   if (!ok)
   {
      smr_hip_canvas_destroy(n);
      return smr_fail(SMR_E_HIP, "canvas_extend: copy of the band and its halo strips failed");
   }
   *out = n;
   return SMR_OK;
}

/* one image OVER the band: lane = pixel of the image */
__global__ void k_blit_over(uint32_t *__restrict__ canvas, int W, int band_y0, int band_h, const uint32_t *__restrict__ img,
                            int x0, int y0, int w, int h)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= w * h) return;
   const int px = x0 + i % w, py = y0 + i / w - band_y0;
   if (px < 0 || px >= W || py < 0 || py >= band_h) return;
   const uint32_t s = img[i];
   if (s == 0u) return;
   uint32_t *d = canvas + (size_t) py * W + px;
   *d = smr_blend_over(s, *d);
}

extern "C" int smr_hip_canvas_blit(smr_canvas_t *c, const uint32_t *pixels, uint64_t n_pixels, const smr_blit_t *items, int n_items)
{
   SMR_NEED_INIT();
   if (!c || n_items < 0 || (n_items && (!pixels || !items))) return smr_fail(SMR_E_INVAL, "canvas_blit: NULL argument");
   for (int k = 0; k < n_items; k++)
      if (items[k].w < 0 || items[k].h < 0 || (uint64_t) items[k].w * (uint64_t) items[k].h > n_pixels ||
          items[k].offset > n_pixels - (uint64_t) items[k].w * (uint64_t) items[k].h)
         return smr_fail(SMR_E_INVAL, "canvas_blit: item %d (%d x %d at offset %llu) lies outside the %llu pixels handed over", k,
                         items[k].w, items[k].h, (unsigned long long) items[k].offset, (unsigned long long) n_pixels);
   if (n_items == 0 || n_pixels == 0) return SMR_OK;
   hipStream_t s = smr_g.stream;
   uint32_t *d_img;
   SMR_TRY(c->ensure(SMR_BUF_BLIT, (size_t) n_pixels * 4, (void **) &d_img));
   SMR_HIP(hipMemcpyAsync(d_img, pixels, (size_t) n_pixels * 4, hipMemcpyHostToDevice, s));
   for (int k = 0; k < n_items; k++)
   {
      const smr_blit_t &it = items[k];
      const int n = it.w * it.h;
      if (n == 0) continue;
      k_blit_over<<<(n + 255) / 256, 256, 0, s>>>(c->pix, c->W, c->band_y0, c->band_h, d_img + it.offset, it.x, it.y, it.w, it.h);
   }
   SMR_HIP(hipGetLastError());
   SMR_HIP(hipStreamSynchronize(s));      /* the caller's pixels may go */
   return SMR_OK;
}

/* smr_hip_canvas_blit for the windows of an atlas: every image lands in every window it meets, at the window's offset rounded to
 * whole pixels (the images are rasterised on the chart's pixel grid: their sub-pixel shift is not redone) */
extern "C" int smr_hip_blit_windows(smr_canvas_t *c, const uint32_t *pixels, uint64_t n_pixels, const smr_blit_t *items, int n_items)
{
   SMR_NEED_INIT();
   if (!c || c->win_n <= 0) return smr_fail(SMR_E_INVAL, "blit_windows: the canvas is not an atlas of windows");
   if (n_items < 0 || (n_items && (!pixels || !items))) return smr_fail(SMR_E_INVAL, "blit_windows: NULL argument");
   for (int k = 0; k < n_items; k++)
      if (items[k].w < 0 || items[k].h < 0 || (uint64_t) items[k].w * (uint64_t) items[k].h > n_pixels ||
          items[k].offset > n_pixels - (uint64_t) items[k].w * (uint64_t) items[k].h)
         return smr_fail(SMR_E_INVAL, "blit_windows: item %d lies outside the %llu pixels handed over", k, (unsigned long long) n_pixels);
   if (n_items == 0 || n_pixels == 0) return SMR_OK;
   hipStream_t s = smr_g.stream;
   uint32_t *d_img = nullptr;
   bool up = false;
   for (int k = 0; k < n_items; k++)
   {
      const smr_blit_t &it = items[k];
      if (it.w == 0 || it.h == 0) continue;
      for (int q = 0; q < c->win_n; q++)
      {
         const int sz = c->win_sizes[(size_t) q];
         const int ox = (int) lrint(c->win_ox[(size_t) q]), oy = (int) lrint(c->win_oy[(size_t) q]);
         if (it.x + it.w <= ox || it.x >= ox + sz || it.y + it.h <= oy || it.y >= oy + sz) continue;
         if (!up)
         {
            SMR_TRY(c->ensure(SMR_BUF_BLIT, (size_t) n_pixels * 4, (void **) &d_img));
            SMR_HIP(hipMemcpyAsync(d_img, pixels, (size_t) n_pixels * 4, hipMemcpyHostToDevice, s));
            up = true;
         }
         const int n = it.w * it.h;
         k_blit_over<<<(n + 255) / 256, 256, 0, s>>>(c->pix + (size_t) q * c->win_pitch * c->W, c->W, 0, sz, d_img + it.offset, it.x - ox, it.y - oy, it.w, it.h);
      }
   }
   SMR_HIP(hipGetLastError());
   SMR_HIP(hipStreamSynchronize(s));
   return SMR_OK;
}

extern "C" int smr_hip_canvas_device(smr_canvas_t *c, uint32_t **d_pix, int *stride_px)
{
   if (!c || !d_pix || !stride_px) return smr_fail(SMR_E_INVAL, "canvas_device: NULL argument");
   *d_pix = c->pix; *stride_px = c->W;
   return SMR_OK;
}

extern "C" int smr_hip_debug_edges(smr_canvas_t *c, int32_t *host_xyxy, uint64_t max_edges, uint64_t *n_edges)
{
   SMR_NEED_INIT();
   if (!c || !n_edges) return smr_fail(SMR_E_INVAL, "debug_edges: NULL");
   *n_edges = c->last_n_edges;
   if (host_xyxy && c->last_n_edges)
   {
      SMR_HIP(hipStreamSynchronize(smr_g.stream));
      const uint64_t n = c->last_n_edges < max_edges ? c->last_n_edges : max_edges;
      SMR_HIP(hipMemcpy(host_xyxy, c->buf[SMR_BUF_EDGES].p, (size_t) n * sizeof(smr_edge_t), hipMemcpyDeviceToHost));
   }
   return SMR_OK;
}

static double mm2wu(double mm, int dpi) { return mm == 0.0 ? 1.0 * 72 / dpi : mm * 72 / 25.4; }

typedef smr_win_ctx_t win_ctx_t;
static int draw_pass(smr_canvas_t *c, const smr_draw_style_t *st, const smr_draw_batch_t *b, int mode, const win_ctx_t *wc = nullptr);

__global__ void k_max_u32(const uint32_t *__restrict__ v, int64_t n, uint32_t *__restrict__ keep)
{
   const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) keep[i] = max(keep[i], v[i]);
}

/* in-place exclusive scan of buf[0, n) on the stream, buf[n] = total, which also comes back to the host */
static int count_finish(uint32_t *buf, uint32_t n, uint32_t *sums, uint32_t *total_dev, uint32_t *total, hipStream_t s)
{
   const uint32_t nb = (n + SCAN_B - 1) / SCAN_B;
   k_scan_block<<<nb, 256, 0, s>>>(buf, buf, sums, n, nullptr);
   k_scan_sums<<<1, 1024, 0, s>>>(sums, nb, total_dev);
   k_scan_add<<<nb, 256, 0, s>>>(buf, sums, n);
   k_set_u32<<<1, 1, 0, s>>>(&buf[n], total_dev);
   SMR_HIP(hipGetLastError());
   SMR_HIP(hipMemcpyAsync(total, total_dev, 4, hipMemcpyDeviceToHost, s));
   SMR_HIP(hipStreamSynchronize(s));
   return SMR_OK;
}

/* cairo_smr_dash (:898-923) + parse_dash_style (src/smrparse.c:1077-1102): the dash array of one stroke in pt */
struct dash_spec_t { int n; double d[4]; double sum; };

static int dash_spec(int style, const double *mult, int nmult, double bwidth_mm, int dpi, dash_spec_t *o)
{
   static const double def_dashed[2] = { 7, 3 }, def_dotted[1] = { 1 }, def_round[2] = { 0, 2 };
   o->n = 0; o->sum = 0; o->d[0] = o->d[1] = o->d[2] = o->d[3] = 0;
   if (nmult < 0 || nmult > 4) return smr_fail(SMR_E_INVAL, "draw: %d dash lengths (at most 4)", nmult);
   if (nmult == 0)
   {
      if (style == SMR_STYLE_DASHED || style == SMR_STYLE_PIPE) { mult = def_dashed; nmult = 2; }
      else if (style == SMR_STYLE_DOTTED) { mult = def_dotted; nmult = 1; }
      else if (style == SMR_STYLE_ROUNDDOT) { mult = def_round; nmult = 2; }
      else return SMR_OK;
   }
   for (int i = 0; i < nmult; i++)
   {
      o->d[i] = mm2wu(bwidth_mm, dpi) * mult[i];
      if (!(o->d[i] >= 0) || !std::isfinite(o->d[i])) return smr_fail(SMR_E_INVAL, "draw: dash length %g (libcairo rejects the array; the rule would draw nothing)", mult[i]);
      o->sum += o->d[i];
   }
   o->n = nmult;
   if (!(o->sum * dpi / 72 >= 1.0 / 16)) return smr_fail(SMR_E_INVAL, "draw: dash period of %g px", o->sum * dpi / 72);
   return SMR_OK;
}

/* the stroker's pen (_cairo_pen_init, tolerance 0.1): on the host, with the libm libcairo itself would use */
static int host_pen(double hw_pt, int dpi, std::vector<int2> &out)
{
   const double s = (double) dpi / 72, tol = 0.1;
   const double major = hw_pt * sqrt(s * s);
   int n;
   if (tol >= 4 * major) n = 1;
   else if (tol >= major) n = 4;
   else
   {
      n = (int) ceil(2 * M_PI / acos(1 - tol / major));
      if (n % 2) n++;
      if (n < 4) n = 4;
   }
   if (n > 256) return smr_fail(SMR_E_CAPACITY, "draw: round caps of a %g px wide stroke need a pen of %d vertices (limit 256)", 2 * hw_pt * s, n);
   for (int i = 0; i < n; i++)
   {
      const double th = 2 * M_PI * i / (double) n;
      out.push_back(make_int2((int32_t) nearbyint(hw_pt * cos(th) * s * 256.0), (int32_t) nearbyint(hw_pt * sin(th) * s * 256.0)));
   }
   return SMR_OK;
}

/* SMR_HIP_LIMITS=0: no cuts at the page's sides (measurements; the oracle's ORC_NO_LIMITS) */
static int limits_off(void)
{
   const char *e = getenv("SMR_HIP_LIMITS");
   return e && e[0] == '0';
}

/* SMR_HIP_STROKE=polygon: every solid stroke as the stroker's polygon (measurements; the oracle's ORC_STROKE_OUTLINE=1) */
static int stroke_polygon_always(void)
{
   const char *e = getenv("SMR_HIP_STROKE");
   return e && strcmp(e, "polygon") == 0;
}

extern "C" uint32_t smr_hip_draw_status(void) { return smr_g.draw_status; }

int smr_scan_u32(uint32_t *buf, uint32_t n, uint32_t *sums, uint32_t *total_dev, uint32_t *total)
{
   return count_finish(buf, n, sums, total_dev, total, smr_g.stream);
}

extern "C" int smr_hip_draw(smr_canvas_t *c, const smr_draw_style_t *st, const smr_draw_batch_t *b) { return smr_draw_instances(c, st, b, nullptr); }

/* render_poly_line's image fill (src/smrules_cairo.c:989-997, img_ini :2691-2704, img_fill :2742-2751): the CLOSED ways of the batch
 * filled with an image repeated over the page -- CAIRO_EXTEND_REPEAT, pattern matrix scale(dpi / 72): one image pixel per chart
 * pixel, the image's top-left pixel on the chart's -- as one group of its own.  Of the style only directional and evenodd count */
extern "C" int smr_hip_draw_pattern(smr_canvas_t *c, const smr_draw_style_t *st, const smr_draw_batch_t *b, const uint32_t *pattern, int32_t pat_w, int32_t pat_h)
{
   smr_g.draw_status = 0;
   SMR_NEED_INIT();
   if (!c || !st || !b || !pattern) return smr_fail(SMR_E_INVAL, "draw_pattern: NULL argument");
   if (pat_w <= 0 || pat_h <= 0 || (int64_t) pat_w * pat_h > (1 << 26)) return smr_fail(SMR_E_INVAL, "draw_pattern: image of %d x %d pixels", pat_w, pat_h);
   if (c->win_n) return smr_fail(SMR_E_UNSUPPORTED, "draw_pattern: not on an atlas of windows (the image is laid out from the chart's origin)");
   uint32_t *d_pat;
   SMR_TRY(c->ensure(SMR_BUF_PATTERN, (size_t) pat_w * pat_h * 4, (void **) &d_pat));
   SMR_HIP(hipMemcpyAsync(d_pat, pattern, (size_t) pat_w * pat_h * 4, hipMemcpyHostToDevice, smr_g.stream));
   SMR_HIP(hipStreamSynchronize(smr_g.stream));             /* the caller's pixels may go */
   smr_draw_style_t s2;
   memset(&s2, 0, sizeof(s2));
   s2.fill_used = 1; s2.fill_col = 0; s2.ignore_open = 1; s2.directional = st->directional; s2.evenodd = st->evenodd;
   s2.curve = st->curve; s2.curve_factor = st->curve_factor;
   c->pat = d_pat; c->pat_w = pat_w; c->pat_h = pat_h;
   const int rc = smr_draw_instances(c, &s2, b, nullptr);
   c->pat = nullptr; c->pat_w = c->pat_h = 0;
   return rc;
}

int smr_draw_instances(smr_canvas_t *c, const smr_draw_style_t *st, const smr_draw_batch_t *b, const smr_win_ctx_t *wc)
{
   smr_g.draw_status = 0;
   SMR_NEED_INIT();
   if (!c || !st || !b) return smr_fail(SMR_E_INVAL, "draw: NULL argument");
   if (st->fill_style < 0 || st->fill_style > 5 || st->border_style < 0 || st->border_style > 5)
      return smr_fail(SMR_E_INVAL, "draw: unknown style %d / %d", st->fill_style, st->border_style);
   if (st->curve < 0 || st->curve > 2) return smr_fail(SMR_E_INVAL, "draw: curve mode %d", st->curve);
   if (st->curve == 1 && !(st->curve_factor > 0 && st->curve_factor < 100)) return smr_fail(SMR_E_INVAL, "draw: curve_factor %g", st->curve_factor);
   if (st->directional && st->fill_used)
   {
      /* act_draw_ini disables the border in directional mode (:674-680); open ways are drawn at once into the
       * rule's group (act_draw_main :1030-1036), the closed ones as one compound fill in a second group
       * (act_draw_fini :1084-1140) */
      smr_draw_style_t s2 = *st;
      s2.border_used = 0; s2.directional = 0;
      SMR_TRY(draw_pass(c, &s2, b, SMR_PASS_OPEN, wc));
      return draw_pass(c, &s2, b, SMR_PASS_COMPOUND, wc);
   }
   return draw_pass(c, st, b, SMR_PASS_NORMAL, wc);      /* directional without a fill is not directional (:681-685) */
}

/* most edge slots of one tile the second raster pass is given lists for; SMR_HIP_DEBUG_BIG_SLOTS lowers it so that a test can
 * reach the pass's clean failure (a real chart would need a 64x4-px tile with 16 384 entries) */
/* SMR_HIP_COVER=0 leaves every primitive to k_raster's own scan conversion (diagnostics, A/B timing) */
static bool smr_cover_enabled()
{
   const char *e = getenv("SMR_HIP_COVER");
   return !(e && *e == '0');
}

/* SMR_HIP_OVERLAP=0 runs a pass's binning after k_cover instead of beside it (A/B timing) */
static bool overlap_enabled()
{
   const char *e = getenv("SMR_HIP_OVERLAP");
   return !(e && *e == '0');
}

/* SMR_HIP_BLEND=0 sends the tiles that hold alpha words only through k_raster too (A/B timing) */
/* SMR_HIP_LINEAR=0: no fill takes k_cover's crossing-by-crossing path (A/B timing, parity of the two paths) */
static bool linear_enabled()
{
   const char *e = getenv("SMR_HIP_LINEAR");
   return !(e && *e == '0');
}

/* SMR_HIP_SPLIT=0: a primitive's strips are never shared out among wavefronts (A/B timing, parity) */
static bool split_enabled()
{
   const char *e = getenv("SMR_HIP_SPLIT");
   return !(e && *e == '0');
}

static bool blend_enabled()
{
   const char *e = getenv("SMR_HIP_BLEND");
   return !(e && *e == '0');
}

static uint64_t big_slots_max()
{
   const char *e = getenv("SMR_HIP_DEBUG_BIG_SLOTS");
   return e && *e ? strtoull(e, nullptr, 10) : (uint64_t) SMR_BIG_SLOTS_MAX;
}

static int draw_pass(smr_canvas_t *c, const smr_draw_style_t *st, const smr_draw_batch_t *b, int mode, const win_ctx_t *wc)
{
   if (b->n_ways < 0 || b->n_refs < 0 || b->n_ways > 0x7ffffff0ll / 2 || b->n_refs > 0x7fffffffll)
      return smr_fail(SMR_E_INVAL, "draw: batch too large (n_ways %lld, n_refs %lld)", (long long) b->n_ways, (long long) b->n_refs);
   if (b->n_ways == 0 || b->n_refs == 0 || (!st->fill_used && !st->border_used)) return SMR_OK;
   if (!b->d_x || !b->d_y || !b->d_way_off) return smr_fail(SMR_E_INVAL, "draw: missing device arrays");
   if (b->n_nodes <= 0 || (!b->d_ref && b->n_refs > b->n_nodes))
      return smr_fail(SMR_E_INVAL, "draw: n_nodes %lld does not cover the refs (n_refs %lld%s)", (long long) b->n_nodes, (long long) b->n_refs, b->d_ref ? "" : ", identity refs");

   hipStream_t s = smr_g.stream;
   /* dash arrays in pt: cairo_smr_dash (:898-923) scales the multipliers by the BORDER width, also for the fill */
   dash_spec_t dsb, dso;
   SMR_TRY(dash_spec(st->border_used ? st->border_style : 0, st->border_dash, st->border_ndash, st->border_width_mm, c->dpi, &dsb));
   SMR_TRY(dash_spec(st->fill_used ? st->fill_style : 0, st->fill_dash, st->fill_ndash, st->border_width_mm, c->dpi, &dso));
   const bool pipe = st->border_used && st->border_style == SMR_STYLE_PIPE;
   const bool round_border = st->border_used && (st->border_style == SMR_STYLE_ROUNDDOT || pipe);
   /* the fill stroke of an open way with style=pipe is the dots alone: cairo_smr_dash(DRAW_PIPE) sets {0, sum} and round caps (:915-920);
    * with style=pipe|rounddot it switches the rule's context to round caps for good (:909-921), and the border strokes of every
    * LATER way have them */
   const bool round_fill = st->fill_used && (st->fill_style == SMR_STYLE_PIPE || st->fill_style == SMR_STYLE_ROUNDDOT);
   if (st->fill_used && st->fill_style == SMR_STYLE_PIPE) { dso.n = 2; dso.d[0] = 0; dso.d[1] = dso.sum; dso.d[2] = dso.d[3] = 0; }
   const int K = pipe ? 3 : 2;
   if (pipe && mode == SMR_PASS_COMPOUND) return smr_fail(SMR_E_INVAL, "draw: internal: compound pass with a border");
   const int64_t n_ways = b->n_ways, n_prims = K * n_ways;
   int64_t n_refs = b->n_refs;
   const int64_t *way_off = b->d_way_off;      /* curve=1 swaps in the decomposed ways below */
   int64_t n_items = 2 * ((n_refs >> 3) + n_ways);           /* id space of k_prim_setup, sparse */
   const int TX = (c->W + SMR_TW - 1) / SMR_TW, TY = (c->band_h + SMR_TH - 1) / SMR_TH;
   const uint32_t n_tiles = (uint32_t) TX * TY;
   smr_timer_t T(smr_g.profiling);
   T.mark(s);                                                   /* 0 */

   int2 *pts; int32_t *refway; uint8_t *wayflags; uint32_t *small; smr_prim_t *prims; smr_item_t *items; uint32_t *way_pix = nullptr;
   SMR_TRY(c->ensure(SMR_BUF_PTS, (size_t) n_refs * sizeof(int2), (void **) &pts));
   SMR_TRY(c->ensure(SMR_BUF_REFWAY, (size_t) n_refs * 4, (void **) &refway));
   SMR_TRY(c->ensure(SMR_BUF_WAYFLAGS, (size_t) n_ways + 4, (void **) &wayflags));      /* k_stroke_risk sets a byte through its aligned word */
   SMR_TRY(c->ensure(SMR_BUF_SMALL, 2048, (void **) &small));     /* [0] any_open_fill [1] first_prim [2] status [3] entries [4] mask words */
   SMR_TRY(c->ensure(SMR_BUF_PRIMS, ((size_t) n_prims + 1) * sizeof(smr_prim_t), (void **) &prims));
   const uint32_t compound = mode == SMR_PASS_COMPOUND ? (uint32_t) n_prims : 0xffffffffu;
   const uint32_t cwords = mode == SMR_PASS_COMPOUND ? (uint32_t) (c->H + 31) / 32 + 1 : 0;
   SMR_TRY(c->ensure(SMR_BUF_ITEMS, (size_t) n_items * sizeof(smr_item_t), (void **) &items));
   uint32_t init_small[8] = { 0, 0xffffffffu, 0, 0, 0, 0xffffffffu, 0, 0 };     /* [5] first stroked way, [6] scan totals */
   SMR_HIP(hipMemsetAsync(small, 0, 2048, s));                  /* [64 + 8 * xcd]: k_raster's tile queues, [128 + 8 * xcd]: k_prim_setup's, [256 + 8 * xcd]: k_blend's */
   SMR_HIP(hipMemcpyAsync(small, init_small, sizeof(init_small), hipMemcpyHostToDevice, s));
   SMR_HIP(hipMemsetAsync(&small[14], 0xff, 4, s));             /* [14] first open way with a fill stroke */
   uint32_t *win_first = nullptr;
   if (wc && c->win_n > 0 && c->win_pitch > 0)
   {
      SMR_TRY(c->ensure(SMR_BUF_WIN_FIRST, (size_t) c->win_n * 4, (void **) &win_first));
      SMR_HIP(hipMemsetAsync(win_first, 0xff, (size_t) c->win_n * 4, s));
   }

   const int TB = 256;
   auto blocks = [](int64_t n, int tb) { return (unsigned) ((n + tb - 1) / tb); };
   int32_t sfix = (int32_t) nearbyint((double) c->dpi / 72 * 256.0);
   {
      unsigned nb = blocks(n_refs, TB); if (nb > 65536u * 4) nb = 65536u * 4;
      /* (a window of an atlas holds chart coordinates: its nodes are far from the atlas's own rows, the shift takes them there) */
      const double far_x = (double) (c->win_n ? c->chart_w : c->W) + SMR_FAR_PX, far_y = (double) (c->win_n ? c->chart_h : c->H) + SMR_FAR_PX;
      k_dev_points<<<nb, TB, 0, s>>>(b->d_x, b->d_y, b->d_ref, n_refs, b->n_nodes, c->dpi, sfix, far_x, far_y, b->d_way_off, n_ways, pts, refway, &small[2],
                                     wc ? wc->node_win : nullptr, wc ? wc->win_shift : nullptr);
   }
   k_way_info<<<blocks(n_ways, TB), TB, 0, s>>>(b->d_way_off, b->d_ref, n_ways, n_refs, st->border_used, st->fill_used,
                                                st->ignore_open, mode, wayflags, &small[0], &small[5], &small[14], &small[2]);
   uint32_t h_small[16];         /* [8] tiles left to k_raster<true>, [9] most entries of one of them, [10] that kernel's queue */
   SMR_HIP(hipMemcpyAsync(h_small, small, sizeof(h_small), hipMemcpyDeviceToHost, s));
   SMR_HIP(hipStreamSynchronize(s));
   const bool open_fill = h_small[0] != 0;
   /* the batch itself is checked on the device, by the two kernels that read it first; nothing has been drawn yet */
   if (h_small[2] & SMR_ST_BADBATCH)
      return smr_fail(SMR_E_INVAL, "draw: bad batch: d_way_off must start at 0, ascend and end at n_refs (%lld), every ref must lie in [0, n_nodes = %lld)",
                      (long long) b->n_refs, (long long) b->n_nodes);
   if (h_small[2] & SMR_ST_FARAWAY)
      return smr_fail(SMR_E_CAPACITY, "draw: a node of the batch lies more than %d px beyond the chart (or is not a number); the rule was not drawn: clip such ways first", (int) SMR_FAR_PX);

   double hw_nofill = mm2wu(st->border_width_mm, c->dpi) / 2;
   double hw_open = (mm2wu(2 * st->border_width_mm, c->dpi) + mm2wu(st->fill_width_mm, c->dpi)) / 2;
   double hw_closed = mm2wu(2 * st->border_width_mm, c->dpi) / 2;
   double hw_fill = mm2wu(st->fill_width_mm, c->dpi) / 2;
   /* a SOLID stroke with round caps: the dasher with one endless dash (quad per segment, wedge per join, a fan per cap; oracle:
    * stroke_styled).  Fill strokes after a round border: all of them (a way's border comes before its fill stroke).  Border strokes
    * after a round fill stroke: those of the ways after the first open way; the others stay k_stroke_edges' */
   const bool solid_open = open_fill && round_border && dso.n == 0;
   const bool solid_border = st->border_used && dsb.n == 0 && round_fill && open_fill;
   const dash_spec_t endless = { 1, { SMR_DASH_ENDLESS, 0, 0, 0 }, SMR_DASH_ENDLESS };
   if (solid_open) dso = endless;
   if (solid_border) dsb = endless;
   const int cap_b = st->border_style == SMR_STYLE_ROUNDDOT ? 1 : pipe ? 2 : round_fill && open_fill ? 3 : 0;
   const int cap_o = round_border || round_fill ? 1 : 0;

   /* dashed strokes: count the pieces of every way, place them with a scan */
   const bool dash_b = st->border_used && dsb.n > 0, dash_o = open_fill && dso.n > 0;
   int2 *tin = nullptr, *tout = nullptr;
   uint8_t *fskip = nullptr, *pflag = nullptr, *bskip = nullptr;
   if (st->curve != 0)
   {
      /* dashed strokes along curves: k_dash walks the decomposition like libcairo's dashing stroker (faces along the chords, round
       * joins, tangent faces where a curve_to ends).  A SOLID stroke of a curve whose caps an earlier stroke of the rule has made round
       * stays the polygon stroker's: its tangent pieces and a half disc at either end (k_dash's solid branch).  (A dashed fill stroke beside a solid border: the two strokers decompose under different limits; the pass's
       * one decomposition follows the dashing stroker and the border skips what its own limits leave out, k_curve's bskip.) */
      uint32_t *ccnt, *sums; int2 *fpts; int32_t *frefway; int64_t *fway_off;
      SMR_TRY(c->ensure(SMR_BUF_CURVE_CNT, ((size_t) n_refs + 1) * 4, (void **) &ccnt));
      SMR_TRY(c->ensure(SMR_BUF_SUMS, ((size_t) n_refs / SCAN_B + 2) * 4, (void **) &sums));
      curve_args_t CA;
      CA.x = b->d_x; CA.y = b->d_y; CA.ref = b->d_ref; CA.pts = pts; CA.way_off = way_off; CA.refway = refway; CA.wayflags = wayflags;
      CA.n_refs = n_refs; CA.n_ways = n_ways; CA.dpi = c->dpi; CA.sfix = sfix; CA.f = st->curve_factor; CA.knots = st->curve == 2;
      CA.node_win = wc ? wc->node_win : nullptr; CA.win_shift = wc ? wc->win_shift : nullptr; CA.way_clip = wc ? wc->way_clip : nullptr; CA.way_lim = wc ? wc->way_lim : nullptr;
      CA.border_used = st->border_used; CA.fill_used = st->fill_used;
      CA.W256 = c->W * 256; CA.H256 = c->H * 256;
      /* _cairo_stroke_style_max_distance_from_path: miter joins, limit 10 -> sqrt(2) * 10 line widths, in device px */
      const double wpt[4] = { 2 * hw_nofill, 2 * hw_open, 2 * hw_closed, 2 * hw_fill };
      for (int i = 0; i < 4; i++)
      {
         const double r = M_SQRT2 * 10.0 * wpt[i] * hypot((double) c->dpi / 72, 0.0);
         if (!(r * 256.0 < 1e9)) return smr_fail(SMR_E_INVAL, "draw: stroke width of %g pt", wpt[i]);
         CA.reach[i] = (int32_t) nearbyint(r * 256.0);
      }
      CA.cnt = ccnt; CA.off = nullptr; CA.fpts = nullptr; CA.tin = CA.tout = nullptr; CA.frefway = nullptr; CA.fway_off = nullptr; CA.fskip = nullptr;
      CA.pflag = nullptr; CA.bskip = nullptr; CA.border_dashed = dash_b && !solid_border ? 1 : 0; CA.fill_dashed = dash_o && !solid_open ? 1 : 0;      /* (a solid stroke with round caps is the polygon stroker's) */
      CA.status = &small[2];
      uint32_t *fixbuf;
      SMR_TRY(c->ensure(SMR_BUF_CURVEFIX, 16 + (size_t) SMR_CURVEFIX_CAP * sizeof(smr_curvefix_t), (void **) &fixbuf));
      CA.fix_n = st->curve == 1 ? fixbuf : nullptr; CA.fix = reinterpret_cast<smr_curvefix_t *>(fixbuf + 4); CA.fix_cnt = 0;
      SMR_HIP(hipMemsetAsync(fixbuf, 0, 16, s));
      k_curve<false><<<blocks(n_refs, 64), 64, 0, s>>>(CA);
      uint32_t total = 0;
      SMR_TRY(count_finish(ccnt, (uint32_t) n_refs, sums, &small[6], &total, s));
      if (st->curve == 1)
      {
         /* control points the device cannot reproduce to the bit (smr_control_points): from the host, then count again */
         uint32_t nfix = 0;
         SMR_HIP(hipMemcpyAsync(&nfix, fixbuf, 4, hipMemcpyDeviceToHost, s));
         SMR_HIP(hipStreamSynchronize(s));
         if (nfix > SMR_CURVEFIX_CAP)
            return smr_fail(SMR_E_CAPACITY, "draw: curve=1 with %u segments that run straight back along their neighbours (at most %u per rule)", nfix, SMR_CURVEFIX_CAP);
         if (nfix)
         {
            std::vector<smr_curvefix_t> fx(nfix);
            SMR_HIP(hipMemcpyAsync(fx.data(), CA.fix, (size_t) nfix * sizeof(smr_curvefix_t), hipMemcpyDeviceToHost, s));
            SMR_HIP(hipStreamSynchronize(s));
            for (auto &e : fx) { double o[4]; smr_control_points_host(e.v, st->curve_factor, o); e.v[0] = o[0]; e.v[1] = o[1]; e.v[2] = o[2]; e.v[3] = o[3]; }
            std::sort(fx.begin(), fx.end(), [](const smr_curvefix_t &a, const smr_curvefix_t &b) { return a.g < b.g; });
            SMR_HIP(hipMemcpyAsync(CA.fix, fx.data(), (size_t) nfix * sizeof(smr_curvefix_t), hipMemcpyHostToDevice, s));
            SMR_HIP(hipStreamSynchronize(s));      /* fx leaves scope */
            CA.fix_cnt = nfix;
            k_curve<false><<<blocks(n_refs, 64), 64, 0, s>>>(CA);
            SMR_TRY(count_finish(ccnt, (uint32_t) n_refs, sums, &small[6], &total, s));
         }
      }
      if (total == 0)
      {
         uint32_t stw = 0;
         SMR_HIP(hipMemcpyAsync(&stw, &small[2], 4, hipMemcpyDeviceToHost, s));
         SMR_HIP(hipStreamSynchronize(s));
         if (stw & SMR_ST_KNOTS_CLOSED)
            return smr_fail(SMR_E_UNSUPPORTED, "draw: a closed way of explicit Bezier knots (curve=2) must be p0, (c1, c2, p) * m, p0 (the way was not drawn)");
         return SMR_OK;
      }
      SMR_TRY(c->ensure(SMR_BUF_FPTS, (size_t) total * sizeof(int2), (void **) &fpts));
      SMR_TRY(c->ensure(SMR_BUF_FTIN, (size_t) total * sizeof(int2), (void **) &tin));
      SMR_TRY(c->ensure(SMR_BUF_FTOUT, (size_t) total * sizeof(int2), (void **) &tout));
      SMR_TRY(c->ensure(SMR_BUF_FREFWAY, (size_t) total * 4, (void **) &frefway));
      SMR_TRY(c->ensure(SMR_BUF_FWAYOFF, ((size_t) n_ways + 1) * 8, (void **) &fway_off));
      SMR_TRY(c->ensure(SMR_BUF_FSKIP, (size_t) total, (void **) &fskip));
      if (dash_b || dash_o) SMR_TRY(c->ensure(SMR_BUF_PFLAG, (size_t) total, (void **) &pflag));
      CA.pflag = pflag;
      if (dash_o && !solid_open && st->border_used && (!dash_b || solid_border)) SMR_TRY(c->ensure(SMR_BUF_BSKIP, (size_t) total, (void **) &bskip));
      CA.bskip = bskip;
      CA.cnt = nullptr; CA.off = ccnt; CA.fpts = fpts; CA.tin = tin; CA.tout = tout; CA.frefway = frefway; CA.fway_off = fway_off; CA.fskip = fskip;
      k_curve_finish<<<blocks(n_ways + 1, TB), TB, 0, s>>>(fway_off, way_off, ccnt, wayflags, n_ways, total);
      k_curve<true><<<blocks(n_refs, 64), 64, 0, s>>>(CA);
      SMR_HIP(hipGetLastError());
      pts = fpts; refway = frefway; way_off = fway_off; n_refs = total;
      n_items = 2 * ((n_refs >> 3) + n_ways);
      SMR_TRY(c->ensure(SMR_BUF_ITEMS, (size_t) n_items * sizeof(smr_item_t), (void **) &items));
   }
   uint32_t *po_border = nullptr, *po_dots = nullptr, *po_open = nullptr, *id_off = nullptr;
   uint32_t tot_b = 0, tot_d = 0, tot_o = 0;
   dash_args_t DA[3];
   if (dash_b || dash_o)
   {
      const size_t stride = (size_t) n_ways + 1;
      uint32_t *dbuf, *sums;
      SMR_TRY(c->ensure(SMR_BUF_DASH, 5 * stride * 4, (void **) &dbuf));
      SMR_TRY(c->ensure(SMR_BUF_SUMS, ((size_t) n_ways / SCAN_B + 2) * 4, (void **) &sums));
      /* pens of the three border widths, of the dots' and of the fill stroke: on the host, where libm is libcairo's */
      std::vector<int2> pens;
      int pen_off[7], pen_n[7];
      const double pw[7] = { hw_nofill, hw_open, hw_closed, hw_nofill * 2.5, hw_open * 2.5, hw_closed * 2.5, hw_fill };
      for (int i = 0; i < 7; i++)
      {
         pen_off[i] = (int) pens.size(); pen_n[i] = 0;
         /* (along a curve the round joins between its segments take the pen's vertices whatever the caps) */
         const bool need = i < 3 ? (dash_b && (cap_b != 0 || pflag)) : i < 6 ? pipe : (dash_o && (cap_o != 0 || pflag));
         if (need) { SMR_TRY(host_pen(pw[i], c->dpi, pens)); pen_n[i] = (int) pens.size() - pen_off[i]; }
      }
      int2 *d_pen;
      SMR_TRY(c->ensure(SMR_BUF_PEN, (pens.size() + 1) * sizeof(int2), (void **) &d_pen));
      if (!pens.empty()) SMR_HIP(hipMemcpyAsync(d_pen, pens.data(), pens.size() * sizeof(int2), hipMemcpyHostToDevice, s));
      auto fill_args = [&](dash_args_t &A, int dmode, const dash_spec_t &ds, int cap_mode, int pen0, double scale)
      {
         A.pts = pts; A.way_off = way_off; A.wayflags = wayflags; A.n_ways = n_ways; A.mode = dmode;
         A.dpi = c->dpi; A.fill_used = st->fill_used;
         A.hw_nofill = hw_nofill * scale; A.hw_open = hw_open * scale; A.hw_closed = hw_closed * scale; A.hw_fill = hw_fill;
         for (int i = 0; i < 4; i++) A.dash[i] = ds.d[i];
         A.ndash = ds.n; A.cap_mode = cap_mode; A.first_way = cap_mode == 3 ? &small[14] : &small[5]; A.pen = d_pen; A.round_only = 0; A.solid = 0;
         for (int i = 0; i < 3; i++) { A.pen_off[i] = pen_off[dmode == 1 ? 6 : pen0 + i]; A.pen_n[i] = pen_n[dmode == 1 ? 6 : pen0 + i]; }
         A.piece_cnt = nullptr; A.piece_off = nullptr; A.edges = nullptr;
         A.tin = pflag ? tin : nullptr; A.tout = pflag ? tout : nullptr; A.pflag = pflag;
         A.skip = dmode == 0 ? bskip : dmode == 1 ? fskip : nullptr;
      };
      /* the group counts need the per-way piece COUNTS: keep them beside the offsets */
      uint32_t *cnt_keep = dbuf + 4 * stride;
      SMR_HIP(hipMemsetAsync(cnt_keep, 0, stride * 4, s));
      if (dash_b)
      {
         dash_spec_t d1 = dsb;
         fill_args(DA[0], 0, d1, cap_b, 0, 1.0);
         DA[0].round_only = solid_border ? 1 : 0; DA[0].solid = solid_border ? 1 : 0;
         po_border = dbuf;
         DA[0].piece_cnt = po_border;
         k_dash<false><<<blocks(n_ways, 64), 64, 0, s>>>(DA[0]);
         k_max_u32<<<blocks(n_ways, TB), TB, 0, s>>>(po_border, n_ways, cnt_keep);
         SMR_TRY(count_finish(po_border, (uint32_t) n_ways, sums, &small[6], &tot_b, s));
         DA[0].piece_cnt = nullptr; DA[0].piece_off = po_border;
         if (pipe)
         {
            /* the dots: dash {0, sum} with round caps at 2.5 times the width (:979-986, PIPE_DOT_SCALE) */
            dash_spec_t d2; d2.n = 2; d2.d[0] = 0; d2.d[1] = dsb.sum; d2.d[2] = d2.d[3] = 0; d2.sum = dsb.sum;
            fill_args(DA[1], 2, d2, 1, 3, 2.5);
            po_dots = dbuf + stride;
            DA[1].piece_cnt = po_dots;
            k_dash<false><<<blocks(n_ways, 64), 64, 0, s>>>(DA[1]);
            k_max_u32<<<blocks(n_ways, TB), TB, 0, s>>>(po_dots, n_ways, cnt_keep);
            SMR_TRY(count_finish(po_dots, (uint32_t) n_ways, sums, &small[6], &tot_d, s));
            DA[1].piece_cnt = nullptr; DA[1].piece_off = po_dots;
         }
      }
      if (dash_o)
      {
         fill_args(DA[2], 1, dso, cap_o, 0, 1.0);
         DA[2].solid = solid_open ? 1 : 0;
         po_open = dbuf + 2 * stride;
         DA[2].piece_cnt = po_open;
         k_dash<false><<<blocks(n_ways, 64), 64, 0, s>>>(DA[2]);
         k_max_u32<<<blocks(n_ways, TB), TB, 0, s>>>(po_open, n_ways, cnt_keep);
         SMR_TRY(count_finish(po_open, (uint32_t) n_ways, sums, &small[6], &tot_o, s));
         DA[2].piece_cnt = nullptr; DA[2].piece_off = po_open;
      }
      /* id room per way: its refs or half its pieces (8-slot groups), whichever is more */
      id_off = dbuf + 3 * stride;
      k_id_groups<<<blocks(n_ways, TB), TB, 0, s>>>(way_off, n_ways, cnt_keep, nullptr, nullptr, id_off);
      uint32_t tot_groups = 0;
      SMR_TRY(count_finish(id_off, (uint32_t) n_ways, sums, &small[6], &tot_groups, s));
      n_items = (int64_t) K * (((int64_t) tot_groups >> 3) + n_ways);
      SMR_TRY(c->ensure(SMR_BUF_ITEMS, (size_t) n_items * sizeof(smr_item_t), (void **) &items));
   }

   /* edge arena: [border strokes: 8/ref or 4/piece][the pipe's dots][fill polygons: 1/ref][open-way fill strokes] */
   uint64_t off_border = 0;
   /* (solid_border: the plain layout of every way, then the piece lists of the ways with round caps) */
   const uint64_t off_border_po = solid_border ? (uint64_t) SMR_SLOTS * n_refs : 0;
   uint64_t off_dots = !st->border_used ? 0 : dash_b ? off_border_po + 4ull * tot_b : (uint64_t) SMR_SLOTS * n_refs;
   uint64_t off_fill = off_dots + 4ull * tot_d;
   uint64_t off_open = off_fill + (st->fill_used ? (uint64_t) n_refs : 0);
   uint64_t n_edges = off_open + (!open_fill ? 0 : dash_o ? 4ull * tot_o : (uint64_t) SMR_SLOTS * n_refs);
   if (n_edges >= 0xffffff00ull) return smr_fail(SMR_E_CAPACITY, "draw: %llu edge slots in one rule (limit 2^32)", (unsigned long long) n_edges);
   smr_edge_t *edges;
   SMR_TRY(c->ensure(SMR_BUF_EDGES, (size_t) n_edges * sizeof(smr_edge_t), (void **) &edges));
   c->last_n_edges = n_edges;

   if (dash_b)
   {
      DA[0].edges = edges + off_border + off_border_po;
      k_dash<true><<<blocks(n_ways, 64), 64, 0, s>>>(DA[0]);
      if (pipe) { DA[1].edges = edges + off_dots; k_dash<true><<<blocks(n_ways, 64), 64, 0, s>>>(DA[1]); }
   }
   if (st->border_used && (!dash_b || solid_border))
   {
      if (tin)
      {
         /* libcairo's spline_to adds the two corners of the face at every point of a decomposed curve to its contours: the polygon
          * is as simple to emit as for straight ways, but which curved ways NEED it (a quad that crosses itself, §7) is not worked out:
          * on request only */
         if (stroke_polygon_always())
            k_stroke_risk<<<blocks(n_refs, TB), TB, 0, s>>>(pts, way_off, refway, n_refs, wayflags, 0, c->dpi, hw_nofill, hw_open, hw_closed, hw_fill, st->fill_used, 1);
         k_stroke_edges<true><<<blocks(n_refs, TB), TB, 0, s>>>(pts, way_off, refway, n_refs, wayflags, 0, c->dpi, hw_nofill,
                                                                hw_open, hw_closed, hw_fill, st->fill_used, edges + off_border, tin, tout, bskip);
      }
      else
      {
         k_stroke_risk<<<blocks(n_refs, TB), TB, 0, s>>>(pts, way_off, refway, n_refs, wayflags, 0, c->dpi, hw_nofill, hw_open, hw_closed, hw_fill, st->fill_used, stroke_polygon_always());
         k_stroke_edges<false><<<blocks(n_refs, TB), TB, 0, s>>>(pts, way_off, refway, n_refs, wayflags, 0, c->dpi, hw_nofill,
                                                                 hw_open, hw_closed, hw_fill, st->fill_used, edges + off_border, nullptr, nullptr, nullptr);
      }
   }
   if (st->fill_used)
      k_fill_edges<<<blocks(n_refs, TB), TB, 0, s>>>(pts, way_off, refway, n_refs, wayflags, edges + off_fill, fskip);
   if (dash_o) { DA[2].edges = edges + off_open; k_dash<true><<<blocks(n_ways, 64), 64, 0, s>>>(DA[2]); }
   else if (open_fill)
   {
      if (tin)
      {
         if (stroke_polygon_always())
            k_stroke_risk<<<blocks(n_refs, TB), TB, 0, s>>>(pts, way_off, refway, n_refs, wayflags, 1, c->dpi, hw_nofill, hw_open, hw_closed, hw_fill, st->fill_used, 1);
         k_stroke_edges<true><<<blocks(n_refs, TB), TB, 0, s>>>(pts, way_off, refway, n_refs, wayflags, 1, c->dpi, hw_nofill,
                                                                hw_open, hw_closed, hw_fill, st->fill_used, edges + off_open, tin, tout, fskip);
      }
      else
      {
         k_stroke_risk<<<blocks(n_refs, TB), TB, 0, s>>>(pts, way_off, refway, n_refs, wayflags, 1, c->dpi, hw_nofill, hw_open, hw_closed, hw_fill, st->fill_used, stroke_polygon_always());
         k_stroke_edges<false><<<blocks(n_refs, TB), TB, 0, s>>>(pts, way_off, refway, n_refs, wayflags, 1, c->dpi, hw_nofill,
                                                                 hw_open, hw_closed, hw_fill, st->fill_used, edges + off_open, nullptr, nullptr, nullptr);
      }
   }
   uint32_t *wb_pix = nullptr, *wf_pix = nullptr;
   if (b->d_way_border_col || b->d_way_fill_col)
   {
      SMR_TRY(c->ensure(SMR_BUF_WAYPIX, (size_t) n_ways * 8, (void **) &way_pix));
      if (b->d_way_border_col) { wb_pix = way_pix; k_color_pixels<<<blocks(n_ways, TB), TB, 0, s>>>(b->d_way_border_col, n_ways, wb_pix); }
      if (b->d_way_fill_col) { wf_pix = way_pix + n_ways; k_color_pixels<<<blocks(n_ways, TB), TB, 0, s>>>(b->d_way_fill_col, n_ways, wf_pix); }
   }
   SMR_HIP(hipGetLastError());
   T.mark(s);                                                   /* 1: edges done */

   /* a rule of few primitives: k_cover takes every primitive's strips in parts (cover_args_t::split), up to eight, as long as the
    * parts are fewer than the wavefronts the chip holds of it */
   const uint32_t cv_split = mode != SMR_PASS_COMPOUND && !c->strict_rows && smr_cover_enabled() && split_enabled() ?
                             (uint32_t) std::max<int64_t>(1, std::min<int64_t>(8, 8192 / std::max<int64_t>(1, n_prims))) : 1u;
   /* primitive set-up (items, boxes, row masks); the mask arena grows on demand */
   uint32_t *maskbuf;
   /* a multiple of 8, and at least 8: every XCD's share of the primitives needs a workgroup (k_prim_setup) */
   const unsigned setup_wg = ((unsigned) std::min<int64_t>((n_prims + SMR_SETUP_WAVES - 1) / SMR_SETUP_WAVES, 256 * 24 / SMR_SETUP_WAVES) + 7u) & ~7u;
   size_t mask_cap = c->buf[SMR_BUF_MASK].cap / 4;
   {
      size_t want = (size_t) n_prims * 2 + (size_t) setup_wg * SMR_SETUP_WAVES * SMR_MASK_BLOCK + 65536 + cwords;
      if (mask_cap < want) mask_cap = want;
   }
   for (int attempt = 0;; attempt++)
   {
      SMR_TRY(c->ensure(SMR_BUF_MASK, mask_cap * 4, (void **) &maskbuf));
      setup_args_t SA;
      SA.edges = edges; SA.way_off = way_off; SA.wayflags = wayflags; SA.n_ways = n_ways;
      SA.off_border = off_border; SA.off_fill = off_fill; SA.off_open = off_open;
      SA.border_pix = host_color_pixel(st->border_col); SA.fill_pix = host_color_pixel(st->fill_col);
      SA.way_border_pix = wb_pix; SA.way_fill_pix = wf_pix; SA.W = c->W; SA.H = c->H; SA.way_clip = wc ? wc->way_clip : nullptr;
      SA.win_first = win_first; SA.win_pitch = c->win_pitch;
      SA.prims = prims; SA.items = items; SA.mask = maskbuf; SA.mask_cap = (uint32_t) std::min<size_t>(mask_cap, 0xffffffffu);
      SA.small = small; SA.compound = compound; SA.evenodd = st->evenodd != 0;
      {
         /* the recording surface's rectangle is the page in WHOLE pt (cairo_recording_surface_create floors / ceils the doubles of
          * cairo_smr_init_main_image, src/smrules_cairo.c:308-330); the replay's 24.8 scale takes it to device units.  A window of
          * an atlas is a replay at an offset: its ways come with the sides of the image libcairo makes of the source there
          * (way_lim, windows_setup); the page's own sides are rarely inside it, and are left out there */
         const int32_t sc = (int32_t) nearbyint((double) c->dpi / 72 * 256.0);
         const int chart_w = c->win_n ? c->chart_w : c->W;
         SA.lim_l = 0; SA.lim_r = (int32_t) (((int64_t) ((int32_t) ceil((double) chart_w * 72 / c->dpi) * 256) * sc) >> 8);
         SA.way_lim = wc ? wc->way_lim : nullptr;
         SA.lim_on = (wc == nullptr || wc->way_lim != nullptr) && !limits_off();
      }
      SA.K = K; SA.off_dots = off_dots; SA.po_border = po_border; SA.po_dots = po_dots; SA.po_open = po_open; SA.id_off = id_off;
      SA.solid_border = solid_border; SA.solid_open = solid_open; SA.mix_border = solid_border; SA.off_border_po = off_border + off_border_po;
      uint32_t *biglist;
      SMR_TRY(c->ensure(SMR_BUF_BIGLIST, ((size_t) n_prims * 2 + 2) * 4, (void **) &biglist));
      SMR_HIP(hipMemsetAsync(biglist, 0, 4, s));
      SA.band_y0 = c->band_y0; SA.band_h = c->band_h;
      SA.cv_split = cv_split > 1;
      SA.biglist = biglist; SA.scancap = c->strict_rows ? SMR_ROWCAP : SMR_SCANCAP;
      /* a long primitive has at least SMR_BIG_SLOTS edge slots */
      smr_bigacc_t *bigacc;
      SMR_TRY(c->ensure(SMR_BUF_BIGACC, ((size_t) (n_edges / SMR_BIG_SLOTS) + 2) * sizeof(smr_bigacc_t), (void **) &bigacc));
      SA.bigacc = bigacc;
      if (mode == SMR_PASS_COMPOUND)
      {
         /* the compound's own row mask sits at the start of the arena; the members' words follow */
         uint32_t *covered;
         SMR_TRY(c->ensure(SMR_BUF_COVERED, (size_t) cwords * 4, (void **) &covered));
         SMR_HIP(hipMemsetAsync(maskbuf, 0, (size_t) cwords * 4, s));
         SMR_HIP(hipMemsetAsync(covered, 0, (size_t) cwords * 4, s));
         SMR_HIP(hipMemcpyAsync(&small[4], &cwords, 4, hipMemcpyHostToDevice, s));
         k_compound_init<<<1, 1, 0, s>>>(prims, compound, off_fill, (uint32_t) n_refs, host_color_pixel(st->fill_col), c->H, st->evenodd != 0);
      }
      int mask_words = (c->H + 31) / 32 + 1;
      /* at least 8 takes per wavefront before the chunks grow: few primitives (a chart's 50 contour lines) go one by one */
      const int setup_chunk = (int) std::max<int64_t>(1, std::min<int64_t>(SMR_SETUP_CHUNK, n_prims / ((int64_t) setup_wg * SMR_SETUP_WAVES * 8)));
      SMR_HIP(hipMemsetAsync(small + 128, 0, 256, s));           /* the set-up's queues start at zero on every attempt */
      SMR_HIP(hipMemsetAsync(small + 11, 0, 12, s));             /* and so does its tally of what k_cover will need */
      SMR_HIP(hipMemsetAsync(small + 15, 0, 4, s));              /* and the strip-table entries it hands out (cv_split) */
      k_prim_setup<<<setup_wg, 64 * SMR_SETUP_WAVES, (size_t) mask_words * 4 * SMR_SETUP_WAVES, s>>>(SA, mask_words, setup_chunk);
      SMR_HIP(hipGetLastError());
      uint32_t nbig = 0;
      SMR_HIP(hipMemcpyAsync(h_small, small, sizeof(h_small), hipMemcpyDeviceToHost, s));
      SMR_HIP(hipMemcpyAsync(&nbig, biglist, 4, hipMemcpyDeviceToHost, s));
      SMR_HIP(hipStreamSynchronize(s));
      if (!(h_small[2] & SMR_ST_MASK_ARENA))
      {
         /* long primitives: items and box, rows and mask words, vertex rows, then the vertex-free rows window by window, each step on
          * as many wavefronts as it has items or windows */
         if (nbig)
         {
            k_big_items<<<256 * 4, 256, 0, s>>>(prims, edges, items, biglist, bigacc);
            k_big_finish<<<(nbig + 63) / 64, 64, 0, s>>>(prims, biglist, bigacc, maskbuf, SA.mask_cap, small, c->W, c->H, SA.win_first, SA.win_pitch);
            k_big_vrows<<<256 * 4, 256, 0, s>>>(prims, edges, items, biglist, bigacc, maskbuf);
            k_big_rows<<<256 * 16 / SMR_SETUP_WAVES, 64 * SMR_SETUP_WAVES, 0, s>>>(prims, edges, items, biglist, maskbuf, &small[2], c->strict_rows ? SMR_ROWCAP : SMR_SCANCAP, bigacc);
            SMR_HIP(hipGetLastError());
            SMR_HIP(hipMemcpyAsync(h_small, small, sizeof(h_small), hipMemcpyDeviceToHost, s));
            SMR_HIP(hipStreamSynchronize(s));
         }
      }
      if (!(h_small[2] & SMR_ST_MASK_ARENA))
      {
         if (mode == SMR_PASS_COMPOUND)
         {
            uint32_t *covered = (uint32_t *) c->buf[SMR_BUF_COVERED].p;
            k_compound_merge<<<blocks(n_ways, TB), TB, 0, s>>>(prims, n_ways, compound, maskbuf, covered);
            k_compound_rows<<<(unsigned) c->H, 64, 0, s>>>(edges, items, n_items, compound, maskbuf, covered, c->H, &small[2],
                                                           SA.lim_on, SA.lim_l, SA.lim_r, wc && wc->way_lim ? c->d_win_lim : nullptr, c->win_pitch);
            SMR_HIP(hipGetLastError());
         }
         break;
      }
      if (attempt) return smr_fail(SMR_E_CAPACITY, "draw: row-mask arena of %zu words still too small", mask_cap);
      mask_cap = (size_t) h_small[4] + 65536;
      init_small[0] = h_small[0];
      init_small[2] = h_small[2] & ~SMR_ST_MASK_ARENA;           /* what the kernels before the set-up reported stays */
      SMR_HIP(hipMemcpyAsync(small, init_small, sizeof(init_small), hipMemcpyHostToDevice, s));
   }
   if (h_small[2] & SMR_ST_ROWCAP)
      return smr_fail(SMR_E_CAPACITY, "draw: more than %d edges of one primitive span a vertex-free pixel row", SMR_ROWCAP);
   T.mark(s);                                                   /* 2: primitives set up */

   /* coverage of the small primitives, primitive by primitive (k_cover): what it manages carries SMR_PF_DELTA from here on.
    * Not for a compound (its members' items carry the compound's id), not on the second attempt of a pass */
   uint32_t *alpha = nullptr;
   uint2 *drows = nullptr;
   uint32_t *cvctr = &small[40];
   /* The binning below needs the items, which are final, and nothing of k_cover's; k_cover is arithmetic, the binning atomics:
    * with k_cover in the pass the binning runs beside it on the second stream (sb) and joins before the tiles are sorted.
    * SMR_HIP_OVERLAP=0: one after the other (A/B timing). */
   hipStream_t sb = s;
   if (mode != SMR_PASS_COMPOUND && !c->strict_rows && smr_cover_enabled() && h_small[11])
   {
      if (overlap_enabled())
      {
         sb = smr_g.stream2;
         SMR_HIP(hipEventRecord(smr_g.ev_fork, s));
         SMR_HIP(hipStreamWaitEvent(sb, smr_g.ev_fork, 0));
      }
      /* k_prim_setup has added up what the eligible primitives need at most (h_small[11] strips, h_small[12..13] alpha words);
       * on top of it the blocks the wavefronts take and do not use up.  An arena that runs out (beyond 2^32 words) leaves the
       * remaining primitives to k_raster's own path */
      /* a small rule is spread over as many wavefronts as it has primitives (a wavefront needs ~30 us per primitive) */
      const unsigned max_wg = 256 * 20 / SMR_CV_WAVES;
      const int64_t n_work = n_prims * cv_split;
      const uint32_t cv_chunk = (uint32_t) std::max<int64_t>(1, std::min<int64_t>(SMR_CV_CHUNK, n_work / ((int64_t) max_wg * SMR_CV_WAVES)));
      const unsigned nwg = (unsigned) std::min<int64_t>((n_work + cv_chunk * SMR_CV_WAVES - 1) / (cv_chunk * SMR_CV_WAVES), max_wg);
      uint64_t need_a = (uint64_t) h_small[12] | ((uint64_t) h_small[13] << 32);
      uint64_t acap = need_a + (uint64_t) nwg * SMR_CV_WAVES * SMR_CV_DBLOCK + 65536;
      uint64_t rcap = (uint64_t) h_small[11] + (uint64_t) nwg * SMR_CV_WAVES * SMR_CV_RBLOCK + 65536;
      if (acap > 0xfffffff0ull) acap = 0xfffffff0ull;
      if (rcap > 0xfffffff0ull) rcap = 0xfffffff0ull;
      SMR_TRY(c->ensure(SMR_BUF_DELTAS, (size_t) acap * 4, (void **) &alpha));
      SMR_TRY(c->ensure(SMR_BUF_DROWS, (size_t) rcap * sizeof(uint2), (void **) &drows));
      cover_args_t CV;
      CV.prims = prims; CV.n_prims = (uint32_t) n_prims; CV.edges = edges; CV.mask = maskbuf;
      CV.alpha = alpha; CV.alpha_cap = (uint32_t) acap; CV.drows = drows; CV.drow_cap = (uint32_t) rcap;
      CV.ctr = cvctr; CV.band_y0 = c->band_y0; CV.band_h = c->band_h; CV.chunk = cv_chunk; CV.linear = linear_enabled() ? 1u : 0u;
      CV.split = cv_split;
      k_cover<<<nwg, 64 * SMR_CV_WAVES, 0, s>>>(CV);
      SMR_HIP(hipGetLastError());
   }
   T.mark(s);                                                   /* 3: small primitives covered */

   /* binning */
   uint32_t *counts, *tile_off, *sums;
   const uint32_t nb_scan = (n_tiles + SCAN_B - 1) / SCAN_B;
   SMR_TRY(c->ensure(SMR_BUF_COUNTS, (size_t) n_tiles * 4, (void **) &counts));
   SMR_TRY(c->ensure(SMR_BUF_TILEOFF, ((size_t) n_tiles + 1) * 4, (void **) &tile_off));
   SMR_TRY(c->ensure(SMR_BUF_SUMS, (size_t) nb_scan * 4, (void **) &sums));
   SMR_HIP(hipMemsetAsync(counts, 0, (size_t) n_tiles * 4, sb));
   k_bin<false><<<blocks(n_items, TB), TB, 0, sb>>>(items, n_items, c->W, c->band_y0, c->band_h, TX, counts, nullptr, nullptr);
   k_scan_block<<<nb_scan, 256, 0, sb>>>(counts, tile_off, sums, n_tiles, &small[7]);
   k_scan_sums<<<1, 256, 0, sb>>>(sums, nb_scan, &small[3]);
   k_scan_add<<<nb_scan, 256, 0, sb>>>(tile_off, sums, n_tiles);
   k_set_u32<<<1, 1, 0, sb>>>(&tile_off[n_tiles], &small[3]);
   SMR_HIP(hipMemcpyAsync(h_small, small, sizeof(h_small), hipMemcpyDeviceToHost, sb));      /* (k_cover's counters are words 40 and up) */
   SMR_HIP(hipStreamSynchronize(sb));
   const uint32_t n_entries = h_small[3];
   /* what the geometry kernels refused is reported before a pixel is touched */
   if (h_small[2] & SMR_ST_CURVE_DEPTH)
      return smr_fail(SMR_E_CAPACITY, "draw: a curve needs more than %d levels of subdivision", SMR_CURVE_STACK);
   if (h_small[2] & SMR_ST_KNOTS_CLOSED)
      return smr_fail(SMR_E_UNSUPPORTED, "draw: a closed way of explicit Bezier knots (curve=2) must be p0, (c1, c2, p) * m, p0 (the way was not drawn)");
   uint32_t *entries;
   SMR_TRY(c->ensure(SMR_BUF_ENTRIES, ((size_t) n_entries + 1) * 4, (void **) &entries));
   SMR_HIP(hipMemsetAsync(counts, 0, (size_t) n_tiles * 4, sb));
   k_bin<true><<<blocks(n_items, TB), TB, 0, sb>>>(items, n_items, c->W, c->band_y0, c->band_h, TX, counts, tile_off, entries);
   SMR_HIP(hipGetLastError());
   if (sb != s)
   {
      /* joined at once: whatever follows on the first stream, an error path's too, comes after the binning */
      SMR_HIP(hipEventRecord(smr_g.ev_join, sb));
      SMR_HIP(hipStreamWaitEvent(s, smr_g.ev_join, 0));
   }
   smr_thdr_t *hdrs;
   uint32_t *tile_cnt, *pure_list = nullptr;
   SMR_TRY(c->ensure(SMR_BUF_TILECNT, (size_t) n_tiles * 4, (void **) &tile_cnt));
   SMR_TRY(c->ensure(SMR_BUF_HDRS, ((size_t) n_entries + 1) * sizeof(smr_thdr_t), (void **) &hdrs));
   {
      sort_args_t SO;
      SO.tile_off = tile_off; SO.entries = entries; SO.items = items; SO.prims = prims; SO.mask = maskbuf; SO.first_prim = &small[1]; SO.win_first = win_first; SO.win_pitch = c->win_pitch;
      SO.hdrs = hdrs; SO.n_tiles = n_tiles; SO.TX = TX; SO.band_y0 = c->band_y0; SO.band_h = c->band_h; SO.drows = drows; SO.tile_cnt = tile_cnt;
      uint32_t cap = 64;
      while (cap < h_small[7] && cap < SORT_LDS) cap <<= 1;       /* h_small[7]: entries of the fullest tile (k_scan_block) */
      SO.lds_cap = cap;
      SO.vec = (uint64_t) n_entries * 16ull < n_tiles ? 1 : 0;
      SO.pure_bit = alpha && blend_enabled() ? SMR_TILE_PURE : 0u;
      SO.pure_list = nullptr; SO.pure_n = &small[320];
      /* a rule of few entries: k_blend goes by a list of its tiles (an atomic per tile here; a chart full of tiles finds them faster by
       * looking, and a wavefront that finds 64 neighbouring tiles of one big polygon in its share blends them one after the other) */
      if ((SO.vec || n_entries <= 16384u) && SO.pure_bit)
         SMR_TRY(c->ensure(SMR_BUF_PURELIST, ((size_t) std::min<uint64_t>(n_entries, n_tiles) + 1) * 4, (void **) &SO.pure_list));
      pure_list = SO.pure_list;
      const uint32_t nwork = SO.vec ? (n_tiles + 63u) / 64u : n_tiles;
      k_tile_sort<<<(nwork + SORT_WAVES - 1) / SORT_WAVES, 64 * SORT_WAVES, (size_t) SORT_WAVES * cap * 4, s>>>(SO);
   }
   SMR_HIP(hipGetLastError());
   T.mark(s);                                                   /* 4: binned + sorted */

   /* A copy of the canvas is taken when the pass may have to be undone: rows of strokes with more spanning edges than one
    * tile's exact path holds (below), and a tile with so many entries that its lists might not fit the second
    * instantiation's memory (h_small[7]: entries of the fullest tile, 64 edge slots each) */
   const bool risk = ((h_small[2] & SMR_ST_ROWRISK) && !c->strict_rows) || (uint64_t) h_small[7] * 64u > big_slots_max();
   uint32_t *snapshot = nullptr;
   uint32_t *ovf_list;
   SMR_TRY(c->ensure(SMR_BUF_OVFLIST, (size_t) n_tiles * 4, (void **) &ovf_list));
   raster_args_t A;
   A.prims = prims; A.hdrs = hdrs; A.edges = edges; A.tile_off = tile_off; A.tile_cnt = tile_cnt;
   A.tile_ctr = &small[64];       /* zero since the memset at the top of the pass */
   A.blend_ctr = &small[256];
   A.blend_list = pure_list; A.blend_list_n = &small[320];
   A.canvas = c->pix; A.status = &small[2]; A.W = c->W; A.band_y0 = c->band_y0; A.band_h = c->band_h; A.TX = TX;
   A.n_tiles = n_tiles;
   A.chunk = n_entries >= 8ull * n_tiles ? SMR_TILE_CHUNK : n_entries >= 2ull * n_tiles ? 2 * SMR_TILE_CHUNK : n_entries * 16ull >= n_tiles ? 4 * SMR_TILE_CHUNK : 64;
   A.direct = st->direct != 0;
   A.ovf_list = ovf_list; A.ovf_cnt = &small[8]; A.ovf_n = 0; A.big_bucket = nullptr; A.big_cap = 0; A.alpha = alpha;
   A.pat = c->pat; A.pat_w = c->pat_w; A.pat_h = c->pat_h;
   {
      /* persistent grid: enough workgroups to fill every CU at the kernel's LDS/register occupancy */
      unsigned nwg = std::min<unsigned>((n_tiles + SMR_RASTER_WAVES - 1) / SMR_RASTER_WAVES, 256u * 8u);
      nwg = (nwg + 7u) & ~7u;            /* a multiple of the 8 XCDs (k_raster's tile order) */
      /* rows of strokes with more spanning edges than one tile's exact path holds: should a tile get more than its share of
       * them after all (SMR_ST_ROWCAP), the pass runs again from this copy with such rows sub-sampled */
      if (risk)
      {
         SMR_TRY(c->ensure(SMR_BUF_SNAPSHOT, (size_t) c->W * c->band_h * 4, (void **) &snapshot));
         SMR_HIP(hipMemcpyAsync(snapshot, c->pix, (size_t) c->W * c->band_h * 4, hipMemcpyDeviceToDevice, s));
      }
      /* the tiles that hold alpha words only (they and k_raster's are disjoint: the order of the two launches is free; side by
       * side on two streams they took longer than one after the other, measured).  k_raster then finds its tiles without the
       * queue: what is left to it is little, and the queue's atomics on eight addresses would be most of its time */
      A.strided = 0;
      if (alpha && blend_enabled())
      {
         unsigned bwg = std::min<unsigned>((unsigned) ((std::min<uint64_t>(pure_list ? n_entries : n_tiles, n_tiles) + SMR_BLEND_WAVES - 1) / SMR_BLEND_WAVES), 256u * 8u);
         k_blend<<<(bwg + 7u) & ~7u, 64 * SMR_BLEND_WAVES, 0, s>>>(A);
         A.strided = 1;
      }
      k_raster<false><<<nwg, 64 * SMR_RASTER_WAVES, 0, s>>>(A);
   }
   SMR_HIP(hipGetLastError());
   SMR_HIP(hipMemcpyAsync(h_small, small, sizeof(h_small), hipMemcpyDeviceToHost, s));
   SMR_HIP(hipStreamSynchronize(s));
   if (h_small[8])
   {
      /* tiles with more crossings on a sub-row than the fast instantiation's LDS lists hold (one-pixel dots, scribbles): again,
       * with lists in global memory sized for every edge slot binned to the fullest of them, so that this run cannot overflow */
      const uint64_t cap = (uint64_t) h_small[9] * 64u;
      if (cap > big_slots_max())
      {
         SMR_HIP(hipMemcpyAsync(c->pix, snapshot, (size_t) c->W * c->band_h * 4, hipMemcpyDeviceToDevice, s));      /* taken above: h_small[7] >= h_small[9] */
         SMR_HIP(hipStreamSynchronize(s));
         return smr_fail(SMR_E_CAPACITY, "draw: a tile with %u entries overflowed its crossing lists (the second pass holds %u edge slots per tile); "
                         "the canvas is as before the rule", h_small[9], (unsigned) big_slots_max());
      }
      const size_t per_wave = (size_t) cap * 64 * 4;
      size_t nwave = std::min<size_t>({ (size_t) 512, (size_t) h_small[8], std::max<size_t>(1, ((size_t) 256 << 20) / per_wave) });
      const unsigned nwg = (unsigned) ((nwave + SMR_RASTER_WAVES - 1) / SMR_RASTER_WAVES);
      uint32_t *big;
      SMR_TRY(c->ensure(SMR_BUF_BIGBUCKET, per_wave * nwg * SMR_RASTER_WAVES, (void **) &big));
      smr_g.draw_status |= SMR_DRAW_SECOND_PASS;
      A.ovf_n = h_small[8]; A.big_bucket = big; A.big_cap = (uint32_t) cap; A.tile_ctr = &small[10];
      k_raster<true><<<nwg, 64 * SMR_RASTER_WAVES, 0, s>>>(A);
      SMR_HIP(hipGetLastError());
      SMR_HIP(hipMemcpyAsync(h_small, small, sizeof(h_small), hipMemcpyDeviceToHost, s));
      SMR_HIP(hipStreamSynchronize(s));
   }
   T.mark(s);                                                   /* 5: rastered */
#ifdef SMR_COUNT_PROFILE
   {
      unsigned long long pr[24];
      hipMemcpy(pr, small + 192, sizeof(pr), hipMemcpyDeviceToHost);
      fprintf(stderr, "k_raster counts:");
      for (int i = 0; i < 18; i++) fprintf(stderr, " %d:%.3fM", i, (double) pr[i] / 1e6);
      fprintf(stderr, "\n");
   }
#endif
#ifdef SMR_SECTION_PROFILE
   {
      unsigned long long pr[12]; double tot = 0;
      hipMemcpy(pr, small + 16, sizeof(pr), hipMemcpyDeviceToHost);
      for (int i = 0; i < 12; i++) tot += (double) pr[i];
      fprintf(stderr, "k_raster sections:");
      for (int i = 0; i < 9; i++) fprintf(stderr, " %d:%.1f%%", i, 100.0 * (double) pr[i] / tot);
      fprintf(stderr, "\n");
   }
#endif
   if (smr_g.profiling)
   {
      smr_timings_t &t = smr_g.last;
      t.edges_ms = T.ms(0, 1); t.prims_ms = T.ms(1, 2); t.cover_ms = T.ms(2, 3); t.bin_ms = T.ms(3, 4); t.raster_ms = T.ms(4, 5); t.total_ms = T.ms(0, 5);
      t.n_prims = (uint64_t) n_prims; t.n_edges = n_edges; t.n_tile_refs = n_entries; t.n_tiles = n_tiles;
      uint32_t cv[6] = { 0, 0, 0, 0, 0, 0 };
      if (alpha) hipMemcpy(cv, cvctr, sizeof(cv), hipMemcpyDeviceToHost);
      t.n_deltas = cv[1]; t.n_cover_prims = cv[3]; t.n_cover_left = cv[4]; t.n_cover_linear = cv[5];
#ifdef SMR_CV_PROFILE
      {
         unsigned long long pr[8]; double tot = 0;
         hipMemcpy(pr, cvctr + 6, sizeof(pr), hipMemcpyDeviceToHost);
         for (int i = 0; i < 8; i++) tot += (double) pr[i];
         fprintf(stderr, "k_cover sections (prim head, strip head, group, crossing, spans, exact rows, alpha):");
         for (int i = 0; i < 7; i++) fprintf(stderr, " %d:%.1f%%", i, 100.0 * (double) pr[i] / tot);
         fprintf(stderr, "\n");
      }
#endif
   }
   if (h_small[2] & (SMR_ST_ROWCAP | SMR_ST_BUCKET_OVERFLOW))
   {
      /* a tile's exact whole-row path met more spanning edges than it holds (only where the set-up reported the risk, so the
       * copy exists): the canvas goes back to what it was, and the pass runs again with such rows sub-sampled */
      if (snapshot)
      {
         SMR_HIP(hipMemcpyAsync(c->pix, snapshot, (size_t) c->W * c->band_h * 4, hipMemcpyDeviceToDevice, s));
         SMR_HIP(hipStreamSynchronize(s));
      }
      if ((h_small[2] & SMR_ST_ROWCAP) && !(h_small[2] & SMR_ST_BUCKET_OVERFLOW) && snapshot && !c->strict_rows)
      {
         c->strict_rows = 1;
         const int rc = draw_pass(c, st, b, mode);
         c->strict_rows = 0;
         return rc;
      }
      if (h_small[2] & SMR_ST_ROWCAP)
         return smr_fail(SMR_E_CAPACITY, "draw: more than %d edges of one primitive span a vertex-free pixel row inside one tile%s", SMR_ROWCAP,
                         snapshot ? "; the canvas is as before the rule" : "");
      /* cannot happen: the second pass's lists hold every edge slot of the tile */
      return smr_fail(SMR_E_HIP, "draw: internal: the crossing lists of the second raster pass overflowed%s", snapshot ? "; the canvas is as before the rule" : "");
   }
   if (h_small[2] & SMR_ST_DEGRADED) smr_g.draw_status |= SMR_DRAW_DEGRADED_ROWS;
   return SMR_OK;
}
