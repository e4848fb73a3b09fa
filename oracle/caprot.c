/* caprot.c -- TEST INFRASTRUCTURE (oracle).  Caption auto-rotation scan, restated from
 * src/smrules_cairo.c:1536-2019 (cairo_smr_plane, cairo_smr_color_bw_stretch, cairo_smr_dist,
 * cairo_smr_diff, dv_sample, dv_weight, dv_quantize, dp_get, find_angle) on top of a
 * restatement of what libcairo 1.16 + pixman 0.40 do for the rotated paint at :1746-1759:
 * pattern matrix -> 16.16 pixman transform (with cairo's centre fix-up), bilinear fetch
 * with 7-bit weights and REPEAT_NONE, OVER combine.  Pinned against the real libraries by
 * tests/test_oracle_cairo.py (cd_rot_paint).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "smr_oracle.h"

#define M_2PI (2.0 * M_PI)
#define TRANSPIX 0x7fffffffu
#define DP_LIMIT 0.95
#define COL_STRETCH_F 1.25

static int32_t fixed16_from_double(double d) { return (int32_t) nearbyint(d * 65536.0); }

/* pixman_transform_point_3d on (x, y, 1): rows 0 and 1 of a 16.16 matrix */
static void xform_point(const int32_t m[2][3], int32_t vx, int32_t vy, int64_t *ox, int64_t *oy)
{
   for (int i = 0; i < 2; i++)
   {
      int64_t hi = (int64_t) m[i][0] * (vx >> 16) + (int64_t) m[i][1] * (vy >> 16) + (int64_t) m[i][2] * 1;
      int64_t lo = (int64_t) m[i][0] * (vx & 0xffff) + (int64_t) m[i][1] * (vy & 0xffff) + 0;
      int64_t r = hi + ((lo + 0x8000) >> 16);
      if (i == 0) *ox = r; else *oy = r;
   }
}

/* cairo 1.16 _cairo_matrix_to_pixman_matrix_offset + _cairo_matrix_to_pixman_matrix:
 * (1) part of the translation is moved into integer source offsets (xoff,yoff) chosen to
 * minimise the max-norm ("evenly spread the bits"), (2) every entry is rounded to 16.16,
 * (3) the translation is nudged so that pixman agrees with cairo at (xc,yc). */
#include <float.h>
void orc_rot_matrix(int dw, int dh, int sw, int sh, double a, int32_t m[2][3], int *xoff, int *yoff)
{
   /* pattern(dev) = R(-a) (dev - cd) + cs */
   double c = cos(a), s = sin(a);
   double cdx = dw / 2.0, cdy = dh / 2.0, csx = sw / 2.0, csy = sh / 2.0;
   double xx = c, xy = s, yx = -s, yy = c;
   double x0 = csx - (xx * cdx + xy * cdy), y0 = csy - (yx * cdx + yy * cdy);
   *xoff = 0; *yoff = 0;
   if (x0 != 0.0 || y0 != 0.0)
   {
      double tx = x0, ty = y0, norm = fmax(fabs(tx), fabs(ty));
      for (int i = -1; i < 2; i += 2)
         for (int j = -1; j < 2; j += 2)
         {
            double den = (xx + i) * (yy + j) - xy * yx;
            if (fabs(den) < DBL_EPSILON) continue;
            double x = y0 * xy - x0 * (yy + j);
            double y = x0 * yx - y0 * (xx + i);
            den = 1 / den;
            x *= den; y *= den;
            double nn = fmax(fabs(x), fabs(y));
            if (norm > nn) { norm = nn; tx = x; ty = y; }
         }
      tx = floor(tx); ty = floor(ty);
      *xoff = (int) -tx; *yoff = (int) -ty;
      /* cairo_matrix_translate (&m, tx, ty) */
      x0 = x0 + tx * xx + ty * xy;
      y0 = y0 + tx * yx + ty * yy;
   }
   m[0][0] = fixed16_from_double(xx); m[0][1] = fixed16_from_double(xy); m[0][2] = fixed16_from_double(x0);
   m[1][0] = fixed16_from_double(yx); m[1][1] = fixed16_from_double(yy); m[1][2] = fixed16_from_double(y0);
   int unity = (xx == 1.0 || xx == -1.0 || xx == 0.0) && (xy == 1.0 || xy == -1.0 || xy == 0.0) &&
               (yx == 1.0 || yx == -1.0 || yx == 0.0) && (yy == 1.0 || yy == -1.0 || yy == 0.0);
   if (unity) return;
   double det = xx * yy - xy * yx;
   double ixx = yy / det, ixy = -xy / det, iyx = -yx / det, iyy = xx / det;
   double ix0 = -(ixx * x0 + ixy * y0), iy0 = -(iyx * x0 + iyy * y0);
   double xc = dw / 2.0, yc = dh / 2.0;
   for (int it = 0; it < 5; it++)
   {
      int64_t vx, vy;
      xform_point(m, fixed16_from_double(xc), fixed16_from_double(yc), &vx, &vy);
      double x = vx / 65536.0, y = vy / 65536.0;
      double tx = ixx * x + ixy * y + ix0, ty = iyx * x + iyy * y + iy0;
      tx -= xc; ty -= yc;
      double dx = xx * tx + xy * ty, dy = yx * tx + yy * ty;
      int32_t fdx = fixed16_from_double(dx), fdy = fixed16_from_double(dy);
      m[0][2] -= fdx; m[1][2] -= fdy;
      if (fdx == 0 && fdy == 0) break;
   }
}

static inline uint32_t mul_80(uint32_t a, uint32_t b) { uint32_t t = a * b + 0x80; return (t + (t >> 8)) >> 8; }

static uint32_t bilinear(uint32_t tl, uint32_t tr, uint32_t bl, uint32_t br, int distx, int disty)
{
   distx <<= 1; disty <<= 1;                 /* 7-bit weights -> 8-bit scale */
   uint32_t wxy = distx * disty, wxiy = distx * (256 - disty), wixy = (256 - distx) * disty, wixiy = (256 - distx) * (256 - disty);
   uint32_t o = 0;
   for (int sh = 0; sh < 32; sh += 8)
   {
      uint64_t f = (uint64_t) ((tl >> sh) & 0xff) * wixiy + (uint64_t) ((tr >> sh) & 0xff) * wxiy +
                   (uint64_t) ((bl >> sh) & 0xff) * wixy + (uint64_t) ((br >> sh) & 0xff) * wxy;
      o |= (uint32_t) ((f >> 16) & 0xff) << sh;
   }
   return o;
}

static uint32_t fetch(const uint32_t *src, int sw, int sh, int x, int y)
{
   if (x < 0 || y < 0 || x >= sw || y >= sh) return 0;
   return src[(size_t) y * sw + x];
}

/* cairo_smr_diff: paint src rotated by a about the centre of dst, OVER, bilinear */
void orc_rot_paint(uint32_t *dst, int dw, int dh, const uint32_t *src, int sw, int sh, double a)
{
   int32_t m[2][3];
   int xoff, yoff;
   orc_rot_matrix(dw, dh, sw, sh, a, m, &xoff, &yoff);
   for (int line = 0; line < dh; line++)
   {
      int64_t x, y;
      xform_point(m, xoff * 65536 + 0x8000, (line + yoff) * 65536 + 0x8000, &x, &y);
      for (int i = 0; i < dw; i++, x += m[0][0], y += m[1][0])
      {
         int64_t x1 = x - 0x8000, y1 = y - 0x8000;
         int distx = (int) ((x1 >> 9) & 0x7f), disty = (int) ((y1 >> 9) & 0x7f);
         int ix = (int) (x1 >> 16), iy = (int) (y1 >> 16);
         uint32_t s;
         if (ix >= sw || ix + 1 < 0 || iy >= sh || iy + 1 < 0) s = 0;
         else s = bilinear(fetch(src, sw, sh, ix, iy), fetch(src, sw, sh, ix + 1, iy),
                           fetch(src, sw, sh, ix, iy + 1), fetch(src, sw, sh, ix + 1, iy + 1), distx, disty);
         uint32_t *d = &dst[(size_t) line * dw + i];
         uint32_t al = s >> 24;
         if (al == 0xff) *d = s;
         else if (s)
         {
            uint32_t ia = al ^ 0xff, o = 0, dv = *d;
            for (int shf = 0; shf < 32; shf += 8)
            {
               uint32_t v = ((s >> shf) & 0xff) + mul_80((dv >> shf) & 0xff, ia);
               o |= (v > 255 ? 255 : v) << shf;
            }
            *d = o;
         }
      }
   }
}

/* ---- src/smrules_cairo.c:1574-1619 ---- */
typedef struct { double x, y, z; } cart_t;
static void rot_y(cart_t *c, double a) { double x = c->x * cos(a) + c->z * sin(a); double z = -c->x * sin(a) + c->z * cos(a); c->x = x; c->z = z; }
static void rot_z(cart_t *c, double a) { double x = c->x * cos(a) - c->y * sin(a); double y = c->x * sin(a) + c->y * cos(a); c->x = x; c->y = y; }
#define COL_D(x) ((int) round((x) * 255))

uint32_t orc_bw_stretch(uint32_t col)
{
   cart_t c;
   c.x = (double) ((col >> 16) & 0xff) / 255.0;
   c.y = (double) ((col >> 8) & 0xff) / 255.0;
   c.z = (double) (col & 0xff) / 255.0;
   rot_z(&c, -M_PI_4);
   rot_y(&c, -acos(1 / sqrt(3)));
   c.x /= COL_STRETCH_F;
   c.y /= COL_STRETCH_F;
   rot_y(&c, acos(1 / sqrt(3)));
   rot_z(&c, M_PI_4);
   return (col & 0xff000000) | (uint32_t) (COL_D(c.x) << 16) | (uint32_t) (COL_D(c.y) << 8) | (uint32_t) COL_D(c.z);
}

static double alphad(uint32_t p) { return 1.0 - (double) (((p & 0x7f000000) >> 23) & 0xff) / 255.0; }
static double lum(uint32_t c)
{
   return 0.2125 * ((double) ((c >> 16) & 0xff) / 255.0) + 0.7154 * ((double) ((c >> 8) & 0xff) / 255.0) + 0.0721 * ((double) (c & 0xff) / 255.0);
}
static uint32_t to_gray(double a)
{
   if (a > 1.0) a = 1;
   if (a < 0.0) a = 0.0;
   int c = round(a * 255.0);
   return c | c << 8 | c << 16 | 0xff000000;
}

/* cairo_smr_dist, src/smrules_cairo.c:1670-1743 */
double orc_dist(uint32_t *dst, const uint32_t *fg, int w, int h, double *v)
{
   double avg = 0, var = 0, dist;
   int cnt = 0;
   for (int i = 0; i < w * h; i++)
   {
      if (alphad(dst[i]) > 0.2 || alphad(fg[i]) > 0.2) { dst[i] = TRANSPIX; continue; }
      uint32_t dc = orc_bw_stretch(dst[i]), sc = orc_bw_stretch(fg[i]);
      dist = fabs(lum(dc) - lum(sc));
      dst[i] = to_gray(dist);
      avg += dist;
      var += dist * dist;
      cnt++;
   }
   if (cnt) avg /= cnt;
   if (v) *v = var - avg * avg;
   return avg;
}

/* the same pass with the per-pixel difference kept as the exact integer
 * |2125 dR + 7154 dG + 721 dB| (= dist * 2 550 000): what the GPU kernel sums */
void orc_dist_int(uint32_t *dst, const uint32_t *fg, int w, int h, int64_t *ksum, int *cnt_out)
{
   int64_t k = 0; int cnt = 0;
   for (int i = 0; i < w * h; i++)
   {
      if (alphad(dst[i]) > 0.2 || alphad(fg[i]) > 0.2) { dst[i] = TRANSPIX; continue; }
      uint32_t dc = orc_bw_stretch(dst[i]), sc = orc_bw_stretch(fg[i]);
      int64_t d = 2125ll * ((int) ((dc >> 16) & 0xff) - (int) ((sc >> 16) & 0xff)) +
                  7154ll * ((int) ((dc >> 8) & 0xff) - (int) ((sc >> 8) & 0xff)) + 721ll * ((int) (dc & 0xff) - (int) (sc & 0xff));
      k += d < 0 ? -d : d;
      dst[i] = to_gray(fabs(lum(dc) - lum(sc)));
      cnt++;
   }
   *ksum = k; *cnt_out = cnt;
}

/* find_angle :1975-1978: r in px and the number of trial angles */
int orc_num_steps(int fg_w, int fg_h, int dpi, int *r_px)
{
   double r_pt = hypot(fg_w, fg_h) * 72 / dpi;                 /* rdata_px_unit(.., U_PT) */
   int num_steps = (int) round(r_pt * M_PI * 1.0 * 25.4 / 72.0);
   if (r_px) *r_px = (int) round(r_pt * ((double) dpi / 72.0));   /* cairo_smr_cut_out :1516 */
   return num_steps;
}

/* cairo_smr_plane :1536-1558 in device px: columns [x_px, w_px) opaque in the caption colour */
void orc_make_plane(uint32_t *fg, int w_px, int h_px, int x_px, uint32_t smr_col)
{
   uint32_t p = orc_color_pixel(smr_col);
   for (int y = 0; y < h_px; y++)
      for (int x = 0; x < w_px; x++)
         fg[(size_t) y * w_px + x] = x >= x_px ? p : 0;
}

typedef struct { double start, end; } peak_t;
static int cmp_dp(const void *a, const void *b)
{
   const peak_t *s = a, *d = b;
   if (s->end - s->start > d->end - d->start) return -1;
   if (s->end - s->start < d->end - d->start) return 1;
   return 0;
}

/* weight/quantize/peaks of find_angle (:1997-2013, dv_weight :1813, dv_quantize :1868, dp_get :1900) */
double orc_pick_angle(const double *dv_diff_in, int n, double phase_deg, double weight)
{
   double *diff = malloc(sizeof(double) * n), *ang = malloc(sizeof(double) * n), *q = malloc(sizeof(double) * n);
   peak_t *dp = malloc(sizeof(peak_t) * (n + 1));
   double a = 0;
   int i;
   for (a = 0, i = 0; i < n; a += M_2PI / n, i++) { ang[i] = a; diff[i] = dv_diff_in[i]; }
   double phase = phase_deg * M_PI / 180.0;
   for (i = 0; i < n; i++)
   {
      if (weight < 0.0) diff[i] = 1.0 - diff[i];
      diff[i] *= 1.0 - (1.0 - fabs(weight)) * (1.0 - cos(ang[i] * 2.0 + phase)) / 2.0;
   }
   double min = 1.0, max = 0.0;
   for (i = 0; i < n; i++) { if (diff[i] > max) max = diff[i]; if (diff[i] < min) min = diff[i]; }
   for (i = 0; i < n; i++) { q[i] = (diff[i] - min) / (max - min); if (isnan(q[i])) q[i] = 1.0; }
   int peak = q[0] >= DP_LIMIT ? 1 : 0, cnt = 0, last;
   for (i = 1, last = n; i <= last; i++)
   {
      if (peak && q[i % n] < DP_LIMIT)
      {
         peak = 0;
         if (cnt)
         {
            if (ang[i % n] > ang[(i - 1) % n]) dp[cnt - 1].end = (ang[i % n] + ang[(i - 1) % n]) / 2;
            else dp[cnt - 1].end = (ang[i % n] + ang[(i - 1) % n] + M_2PI) / 2;
            if (dp[cnt - 1].end < dp[cnt - 1].start) dp[cnt - 1].end += M_2PI;
         }
         continue;
      }
      if (!peak && q[i % n] >= DP_LIMIT)
      {
         peak = 1;
         if (!cnt) last = i + n - 1;
         cnt++;
         if (ang[i % n] > ang[(i - 1) % n]) dp[cnt - 1].start = (ang[i % n] + ang[(i - 1) % n]) / 2;
         else dp[cnt - 1].start = (ang[i % n] + ang[(i - 1) % n] + M_2PI) / 2;
      }
   }
   if (cnt)
   {
      qsort(dp, cnt, sizeof(peak_t), cmp_dp);
      a = M_2PI - (dp[0].end + dp[0].start) / 2;
   }
   else a = 0;
   free(diff); free(ang); free(q); free(dp);
   return a;
}

double orc_find_angle(const uint32_t *cut, int r, const uint32_t *fg, int w, int h, int num_steps,
                      double phase_deg, double weight, double *dv_out)
{
   uint32_t *dst = calloc((size_t) w * h, 4);
   double *dv = malloc(sizeof(double) * num_steps);
   double a; int i;
   for (a = 0, i = 0; i < num_steps; a += M_2PI / num_steps, i++)
   {
      orc_rot_paint(dst, w, h, cut, r, r, a);
      dv[i] = orc_dist(dst, fg, w, h, NULL);
      if (dv_out) dv_out[i] = dv[i];
   }
   a = orc_pick_angle(dv, num_steps, phase_deg, weight);
   free(dst); free(dv);
   return a;
}
