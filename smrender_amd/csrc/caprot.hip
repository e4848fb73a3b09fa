/* caprot.hip -- caption auto-rotation scan of libsmr_hip: one wavefront per caption.
 *
 * Replaces find_angle (src/smrules_cairo.c:1959-2019) and what it calls per caption:
 * dv_sample :1828-1861 (loop over num_steps angles), cairo_smr_diff :1746-1759 (paint the
 * cut-out rotated, OVER, bilinear -- libcairo/pixman arithmetic restated and pinned in
 * oracle/caprot.c), cairo_smr_dist :1670-1743 (transparency skip, bw-stretch :1598-1619,
 * luminosity difference, gray write-back), dv_weight :1813-1821, dv_quantize :1868-1890,
 * dp_get :1900-1956 and the widest-peak pick :2007-2013.
 *
 * Difference to the reference, by design: the r x r cut-out is a crop of the
 * device canvas at the rounded centre instead of a re-rasterisation of the whole chart at a
 * sub-pixel offset (cairo_smr_cut_out :1511-1533).
 */
#include "smr_dev.h"
#include "smr_host.h"
#include <float.h>
#include <vector>

#define M_2PI (2.0 * M_PI)
#ifndef SMR_CAPROT_SYNC_EVERY
#define SMR_CAPROT_SYNC_EVERY 1      /* angles between two workgroup barriers (they order nothing: they keep a caption's wavefronts on one band of the cut-out) */
#endif
#define SMR_TRANSPIX 0x7fffffffu
#define SMR_DP_LIMIT 0.95

extern "C" int smr_hip_caprot_steps(int fg_w, int fg_h, int dpi, int *num_steps, int *r_px)
{
   if (fg_w <= 0 || fg_h <= 0 || dpi <= 0) return smr_fail(SMR_E_INVAL, "caprot_steps: bad size");
   double r_pt = hypot((double) fg_w, (double) fg_h) * 72 / dpi;            /* find_angle :1975 */
   if (num_steps) *num_steps = (int) round(r_pt * M_PI * 1.0 * 25.4 / 72.0); /* :1978 */
   if (r_px) *r_px = (int) round(r_pt * ((double) dpi / 72.0));              /* cairo_smr_cut_out :1516 */
   return SMR_OK;
}

/* ---- host: the per-angle pixman transform.  It depends only on the plane size, the cut-out
 * size and the angle, never on the caption, so it is computed once per batch with the host
 * libm (the same one libcairo uses) and handed to the kernel as a table. ---- */
struct smr_rotmat_t { int32_t m[2][3]; int32_t xoff, yoff; };

static int32_t fixed16(double d) { return (int32_t) nearbyint(d * 65536.0); }

static void xform_point_h(const int32_t m[2][3], int32_t vx, int32_t vy, int64_t *ox, int64_t *oy)
{
   for (int i = 0; i < 2; i++)
   {
      int64_t hi = (int64_t) m[i][0] * (vx >> 16) + (int64_t) m[i][1] * (vy >> 16) + (int64_t) m[i][2];
      int64_t lo = (int64_t) m[i][0] * (vx & 0xffff) + (int64_t) m[i][1] * (vy & 0xffff);
      int64_t r = hi + ((lo + 0x8000) >> 16);
      if (i == 0) *ox = r; else *oy = r;
   }
}

/* cairo 1.16: _cairo_matrix_to_pixman_matrix_offset + _cairo_matrix_to_pixman_matrix for
 * pattern(dev) = R(-a)(dev - dst centre) + src centre (translate, rotate, set_source_surface
 * of cairo_smr_diff) */
static void rot_matrix(int dw, int dh, int sw, int sh, double a, smr_rotmat_t *out)
{
   double c = cos(a), s = sin(a);
   double cdx = dw / 2.0, cdy = dh / 2.0, csx = sw / 2.0, csy = sh / 2.0;
   double xx = c, xy = s, yx = -s, yy = c;
   double x0 = csx - (xx * cdx + xy * cdy), y0 = csy - (yx * cdx + yy * cdy);
   int32_t (*m)[3] = out->m;
   out->xoff = 0; out->yoff = 0;
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
      out->xoff = (int) -tx; out->yoff = (int) -ty;
      x0 = x0 + tx * xx + ty * xy;
      y0 = y0 + tx * yx + ty * yy;
   }
   m[0][0] = fixed16(xx); m[0][1] = fixed16(xy); m[0][2] = fixed16(x0);
   m[1][0] = fixed16(yx); m[1][1] = fixed16(yy); m[1][2] = fixed16(y0);
   bool unity = (xx == 1.0 || xx == -1.0 || xx == 0.0) && (xy == 1.0 || xy == -1.0 || xy == 0.0) &&
                (yx == 1.0 || yx == -1.0 || yx == 0.0) && (yy == 1.0 || yy == -1.0 || yy == 0.0);
   if (unity) return;
   double det = xx * yy - xy * yx;
   double ixx = yy / det, ixy = -xy / det, iyx = -yx / det, iyy = xx / det;
   double ix0 = -(ixx * x0 + ixy * y0), iy0 = -(iyx * x0 + iyy * y0);
   double xc = dw / 2.0, yc = dh / 2.0;
   for (int it = 0; it < 5; it++)
   {
      int64_t vx, vy;
      xform_point_h(m, fixed16(xc), fixed16(yc), &vx, &vy);
      double x = vx / 65536.0, y = vy / 65536.0;
      double tx = ixx * x + ixy * y + ix0, ty = iyx * x + iyy * y + iy0;
      tx -= xc; ty -= yc;
      double dx = xx * tx + xy * ty, dy = yx * tx + yy * ty;
      int32_t fdx = fixed16(dx), fdy = fixed16(dy);
      m[0][2] -= fdx; m[1][2] -= fdy;
      if (fdx == 0 && fdy == 0) break;
   }
}

/* ---- device ---- */
struct caprot_args_t
{
   const uint32_t *canvas;           /* band pixels */
   const uint32_t *halo_above, *halo_below;   /* neighbour bands' rows (multi-GPU), may be NULL */
   int n_above, n_below;
   int W, H, band_y0, band_h;
   const double *cx, *cy;
   int64_t n;
   int fg_w, fg_h, fg_x, r, num_steps;
   uint32_t fg_pix;
   const uint32_t *fg;               /* optional explicit plane pixels */
   double weight;
   const smr_rotmat_t *mats;         /* [num_steps] */
   const double *ang, *wfac;         /* [num_steps] */
   double *out_angle, *out_dv;
};

/* ALPHAD(p) > 0.2 (cairo_smr_dist :1683-1690) with ALPHAD(p) = 1 - 2 * ((p >> 24) & 0x7f) / 255 in doubles: true up to
 * 2 * 101 = 202 (1 - 202/255 = 0.2078), false from 204 on (1 - 204/255 rounds to 0.19999999999999996 < 0.2) */
__device__ __forceinline__ bool smr_transp(uint32_t p) { return ((p >> 24) & 0x7fu) < 102u; }

/* cairo_smr_color_bw_stretch: the rotate / divide by 1.25 / rotate back chain is the affine
 * map c' = m + (c - m) / 1.25 about the gray value m; in 8-bit units round((S + 12 c) / 15),
 * which has no ties (odd denominator), so integers reproduce the double chain exactly */
/* n / 30 for 0 <= n < 7700 (here n <= 2 * (765 + 3060) + 15): one 24-bit multiply and a shift, checked exhaustively
 * (8739 * 30 = 2^18 + 26, and 7699 * 26 / (30 * 2^18) < 1/30) */
__device__ __forceinline__ int smr_div30(int n) { return (int) (__umul24((uint32_t) n, 8739u) >> 18); }

__device__ __forceinline__ void smr_bw_stretch(uint32_t col, int &r, int &g, int &b)
{
   /* smr_div30(S + 24 c) = ((S + 24 c) * 8739) >> 18 with the common part multiplied once (all below 2^27) */
   const uint32_t R = (col >> 16) & 0xff, G = (col >> 8) & 0xff, B = col & 0xff, S8 = __umul24(2 * (R + G + B) + 15, 8739u);
   r = (int) ((__umul24(R, 24u * 8739u) + S8) >> 18); g = (int) ((__umul24(G, 24u * 8739u) + S8) >> 18); b = (int) ((__umul24(B, 24u * 8739u) + S8) >> 18);
}

/* T[v] = (double) v / 255.0, the IEEE quotient itself, from a table in LDS: three fp64 divisions per pixel and angle were a
 * third of the kernel's instructions */
__device__ __forceinline__ double smr_lum(const double *T, int r, int g, int b)
{
   return 0.2125 * T[r & 255] + 0.7154 * T[g & 255] + 0.0721 * T[b & 255];
}

struct __attribute__((packed, aligned(4))) smr_px2_t { uint32_t a, b; };      /* two neighbouring pixels in one 8-byte load (4-byte aligned) */

__device__ __forceinline__ uint32_t smr_cut_fetch(const caprot_args_t &A, int ox, int oy, int x, int y)
{
   if (x < 0 || y < 0 || x >= A.r || y >= A.r) return 0;      /* REPEAT_NONE on the r x r cut-out */
   int gx = ox + x, gy = oy + y - A.band_y0;
   if (gx < 0 || gx >= A.W) return 0;
   if (gy >= 0 && gy < A.band_h) return A.canvas[(size_t) gy * A.W + gx];
   /* outside the band: the neighbour's rows if they were handed over, clear beyond the chart */
   if (gy < 0 && gy >= -A.n_above) return A.halo_above[(size_t) (A.n_above + gy) * A.W + gx];
   if (gy >= A.band_h && gy < A.band_h + A.n_below) return A.halo_below[(size_t) (gy - A.band_h) * A.W + gx];
   return 0;
}

/* ONE WORKGROUP per caption, up to 16 wavefronts (A.nwaves).  Every thread owns fixed pixels of the caption plane through all
 * angles: what an angle leaves in the plane (cairo_smr_dist's gray write-back, which the next angle's OVER paint meets) is state
 * of that pixel alone, so it sits in the thread's own LDS slots and no barrier orders it.  The per-angle sums are integers
 * (|2125 dR + 7154 dG + 721 dB| and the pixel count), reduced per wavefront and added to LDS by atomics: any order gives the
 * same total.  Why a workgroup and not a wavefront per caption: the scan of one caption reads a disc of r/2 px around its
 * anchor (230 KB at C4's shape), angle by angle a band of it; with a wavefront per caption some 4 000 captions are in flight
 * and their discs (900 MB) stream from HBM once per angle (3.5 MB per caption measured); with a workgroup per caption 256 to
 * 512 are, all wavefronts of a caption on the same angle, and the band stays in the XCD's L2 from one angle to the next. */
#ifndef SMR_CAPROT_MAXWAVES
#define SMR_CAPROT_MAXWAVES 16
#endif
#ifdef SMR_CAPROT_SGPR
__attribute__((amdgpu_num_sgpr(SMR_CAPROT_SGPR)))
#endif
__global__ __launch_bounds__(64 * SMR_CAPROT_MAXWAVES) void k_caprot(caprot_args_t A)
{
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
   double *dv = (double *) smem;                              /* [num_steps] */
   double *T = dv + A.num_steps;                              /* [256] */
   unsigned long long *ksums = (unsigned long long *) (T + 256);      /* [num_steps] */
   uint32_t *cnts = (uint32_t *) (ksums + A.num_steps);       /* [num_steps] */
   uint32_t *dst = (uint32_t *) (smem + (((size_t) A.num_steps * 20 + 2048 + 15) & ~(size_t) 15));   /* the plane */
   const int64_t cap = blockIdx.x;
   const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nthr = blockDim.x, nwaves = nthr >> 6;
   /* dst holds the plane's opaque columns only (all of them with explicit plane pixels), indexed like the walk below */
   const int npix = (A.fg_w - (A.fg ? 0 : A.fg_x)) * A.fg_h;
   const int ox = (int) round(A.cx[cap] - A.r / 2.0), oy = (int) round(A.cy[cap] - A.r / 2.0);
   const bool fg_skip = smr_transp(A.fg_pix);
   int sr, sg, sb;
   smr_bw_stretch(A.fg_pix, sr, sg, sb);
   for (int v = tid; v < 256; v += nthr) T[v] = (double) v / 255.0;
   for (int p = tid; p < npix; p += nthr) dst[p] = 0;
   for (int i = tid; i < A.num_steps; i += nthr) { ksums[i] = 0; cnts[i] = 0; }
   __syncthreads();
   const double lum_s = smr_lum(T, sr, sg, sb);
   /* without explicit plane pixels only the columns from fg_x on are opaque (cairo_smr_plane: the left half of an
    * E/W-aligned caption's plane is clear): the threads walk those alone ... */
   /* ... in blocks of 16 columns x 4 lines per wavefront: rotated onto the canvas a block is a compact patch of a few cache lines,
    * where 64 pixels of one plane line are a 64-px segment that crosses up to 64 of them */
   const int c0 = A.fg ? 0 : A.fg_x, cw = A.fg_w - c0, nbc = (cw + 15) >> 4, nblk = nbc * ((A.fg_h + 3) >> 2);
   const bool small = A.r <= 4096;           /* every coordinate of the transform fits 32 bits (16.16 of less than 2^13 px) */

   /* What an angle leaves in a plane pixel is cairo_smr_dist's gray write-back (or TRANSPIX), and all the next angle does with
    * it is paint the rotated cut-out OVER it: under an opaque sample (nearly all of a chart) it is never looked at.  So the
    * plane keeps the pixel the metric saw, one bit per owned pixel says so, and the gray (three table look-ups and the fp64
    * luminance) is worked out only when a sample that is not opaque needs it. */
   uint32_t rawmask = 0;                      /* bit k: the slot of this thread's k-th block holds the composited pixel, not the gray */
   const bool wide = (nblk + nwaves - 1) / nwaves > 16;       /* more than 16 pixels a lane: the wavefront's sum of an angle needs 64 bits */
   const bool lazy = (nblk + nwaves - 1) / nwaves <= 32;      /* one bit per block of a wavefront: a plane of more blocks writes the gray back at once */
   for (int i = 0; i < A.num_steps; i++)
   {
      const smr_rotmat_t M = A.mats[i];
      uint32_t ksum = 0;                      /* this lane's pixels, at most 2.55 M each */
      int cnt = 0;
      if (!fg_skip || A.fg)
      {
         uint32_t kbit = 1;
         /* the transform of the lane's place in its block, once per angle: a block then adds one uniform term (below) */
         const int Xl = __mul24(M.m[0][0], lane & 15) + __mul24(M.m[0][1], lane >> 4);
         const int Yl = __mul24(M.m[1][0], lane & 15) + __mul24(M.m[1][1], lane >> 4);
         /* (block row, block column) of this wavefront's blocks: uniform, kept by running sums instead of a division per block */
         int brow = wave / nbc, bcol = wave - brow * nbc;
         const int drow = nwaves / nbc, dcol = nwaves - drow * nbc;
         for (int blk = wave; blk < nblk; blk += nwaves, kbit <<= 1, brow += drow, bcol += dcol)
         {
            if (bcol >= nbc) { bcol -= nbc; brow++; }
            const int line = 4 * brow + (lane >> 4), col = 16 * bcol + (lane & 15);
            if (col >= cw || line >= A.fg_h) continue;
            const int q = __mul24(line, cw) + col;                 /* (plane sizes are far below 2^23: a full 32-bit multiply is four passes) */
            const int p = __mul24(line, A.fg_w) + c0 + col;
            int psr = sr, psg = sg, psb = sb;
            int pk = __mul24(2125, sr) + __mul24(7154, sg) + __mul24(721, sb);      /* the plane pixel's side of the metric's difference */
            double plum = lum_s;
            if (A.fg)
            {
               uint32_t fp = A.fg[p];
               if (smr_transp(fp)) continue;            /* transparent plane pixel: never counted */
               smr_bw_stretch(fp, psr, psg, psb);
               pk = __mul24(2125, psr) + __mul24(7154, psg) + __mul24(721, psb);
               plum = smr_lum(T, psr, psg, psb);
            }
            /* pixman: transform the scanline start (pixel centre), then step by the first matrix column.  The start is
             * v = (xoff + 1/2, line + yoff + 1/2) in 16.16, so the low halves of both coordinates are 0x8000 and the rounded
             * low part of the product, (m0 * 0x8000 + m1 * 0x8000 + 0x8000) >> 16, is (m0 + m1 + 1) >> 1 */
            int ix, iy;
            uint32_t distx, disty;
            if (small)
            {
               /* 24-bit multiplies: |m| <= 2^16 (16.16 of a sine), the pixel indices are below 2^13 */
               /* (the products are exact in 32 bits, so the block's uniform part, which the scalar unit works out, and the lane's
                * part add up to the same X and Y) */
               const int ucol = M.xoff + c0 + 16 * bcol, uline = 4 * brow + M.yoff;
               const int x1 = Xl + (M.m[0][0] * ucol + M.m[0][1] * uline + (M.m[0][2] + ((M.m[0][0] + M.m[0][1] + 1) >> 1)) - 0x8000);
               const int y1 = Yl + (M.m[1][0] * ucol + M.m[1][1] * uline + (M.m[1][2] + ((M.m[1][0] + M.m[1][1] + 1) >> 1)) - 0x8000);
               distx = (uint32_t) ((x1 >> 9) & 0x7f) << 1; disty = (uint32_t) ((y1 >> 9) & 0x7f) << 1;
               ix = x1 >> 16; iy = y1 >> 16;
            }
            else
            {
               int32_t vx = M.xoff * 65536 + 0x8000, vy = (line + M.yoff) * 65536 + 0x8000;
               int64_t X = (int64_t) M.m[0][0] * (vx >> 16) + (int64_t) M.m[0][1] * (vy >> 16) + M.m[0][2] +
                           (((int64_t) M.m[0][0] * (vx & 0xffff) + (int64_t) M.m[0][1] * (vy & 0xffff) + 0x8000) >> 16);
               int64_t Y = (int64_t) M.m[1][0] * (vx >> 16) + (int64_t) M.m[1][1] * (vy >> 16) + M.m[1][2] +
                           (((int64_t) M.m[1][0] * (vx & 0xffff) + (int64_t) M.m[1][1] * (vy & 0xffff) + 0x8000) >> 16);
               X += (int64_t) (c0 + col) * M.m[0][0]; Y += (int64_t) (c0 + col) * M.m[1][0];
               int64_t x1 = X - 0x8000, y1 = Y - 0x8000;
               distx = (uint32_t) ((x1 >> 9) & 0x7f) << 1; disty = (uint32_t) ((y1 >> 9) & 0x7f) << 1;
               ix = (int) (x1 >> 16); iy = (int) (y1 >> 16);
            }
            uint32_t s = 0;
            const int gx = ox + ix, gy = oy + iy - A.band_y0;
            /* all four inside the cut-out, the chart and this band (0 <= v && v + 1 < n as one unsigned compare each) */
            const bool inside4 = (uint32_t) ix < (uint32_t) (A.r - 1) && (uint32_t) iy < (uint32_t) (A.r - 1) &&
                                 (uint32_t) gx < (uint32_t) (A.W - 1) && (uint32_t) gy < (uint32_t) (A.band_h - 1);
            if (inside4 || !(ix >= A.r || ix + 1 < 0 || iy >= A.r || iy + 1 < 0))
            {
               uint32_t tl, tr, bl, br;
               if (inside4)
               {
                  /* all four inside the cut-out, the chart and this band (nearly every sample): one address */
                  const uint32_t *q0 = A.canvas + (__umul24((uint32_t) gy, (uint32_t) A.W) + (uint32_t) gx);      /* gy < 2^16, W <= 2^16 */
                  const smr_px2_t t = *reinterpret_cast<const smr_px2_t *>(q0), u = *reinterpret_cast<const smr_px2_t *>(q0 + A.W);
                  tl = t.a; tr = t.b; bl = u.a; br = u.b;
               }
               else
               {
                  tl = smr_cut_fetch(A, ox, oy, ix, iy); tr = smr_cut_fetch(A, ox, oy, ix + 1, iy);
                  bl = smr_cut_fetch(A, ox, oy, ix, iy + 1); br = smr_cut_fetch(A, ox, oy, ix + 1, iy + 1);
               }
               if (tl == tr && tl == bl && tl == br) s = tl;      /* the weights sum to 2^16: four equal pixels give themselves */
               else
               {
                  /* pixman's f = tl (256-dx)(256-dy) + tr dx (256-dy) + bl (256-dx) dy + br dx dy per channel is, in integers,
                   * (256-dy) (tl (256-dx) + tr dx) + dy (bl (256-dx) + br dx): the inner sums are below 2^16 (the two weights add
                   * up to 256), so they are worked out two channels at a time in the packed 16-bit ALU; the outer one in 32 bits */
                  const smr_u16x2 wx = smr_pk(distx * 0x10001u), wix = smr_pk((256u - distx) * 0x10001u);
                  const uint32_t t_rb = smr_unpk(smr_pk_rb(tl) * wix + smr_pk_rb(tr) * wx), t_ag = smr_unpk(smr_pk_ag(tl) * wix + smr_pk_ag(tr) * wx);
                  const uint32_t b_rb = smr_unpk(smr_pk_rb(bl) * wix + smr_pk_rb(br) * wx), b_ag = smr_unpk(smr_pk_ag(bl) * wix + smr_pk_ag(br) * wx);
                  const uint32_t wiy = 256u - disty;
                  const uint32_t fb = (t_rb & 0xffffu) * wiy + (b_rb & 0xffffu) * disty, fr = (t_rb >> 16) * wiy + (b_rb >> 16) * disty;
                  const uint32_t fg = (t_ag & 0xffffu) * wiy + (b_ag & 0xffffu) * disty, fa = (t_ag >> 16) * wiy + (b_ag >> 16) * disty;
                  s = ((fb >> 16) & 0xffu) | ((fg >> 8) & 0xff00u) | (fr & 0xff0000u) | ((fa << 8) & 0xff000000u);
               }
            }
            /* OVER onto what the previous angle left in dst */
            uint32_t d;
            if ((s >> 24) == 0xff) d = s;
            else
            {
               d = dst[q];
               if (rawmask & kbit)
               {
                  /* the write-back of the previous angle, now that it is needed */
                  int qr, qg, qb;
                  smr_bw_stretch(d, qr, qg, qb);
                  double dist = fabs(smr_lum(T, qr, qg, qb) - plum);
                  if (dist > 1.0) dist = 1;
                  if (dist < 0.0) dist = 0.0;
                  /* round() of a non-negative value: truncate, then the exact half test */
                  const double d255 = dist * 255.0, dtr = trunc(d255);
                  const uint32_t c = (uint32_t) (int) dtr + (d255 - dtr >= 0.5 ? 1u : 0u);
                  d = c | c << 8 | c << 16 | 0xff000000u;
               }
               if (s) d = smr_blend_over(s, d);
            }
            /* cairo_smr_dist */
            if (smr_transp(d)) { dst[q] = SMR_TRANSPIX; rawmask &= ~kbit; continue; }
            int dr, dg, db;
            smr_bw_stretch(d, dr, dg, db);
            const int k = __mul24(2125, dr) + __mul24(7154, dg) + __mul24(721, db) - pk;
            ksum += (uint32_t) (k < 0 ? -k : k);
            cnt++;
            if (lazy) { dst[q] = d; rawmask |= kbit; }
            else
            {
               double dist = fabs(smr_lum(T, dr, dg, db) - plum);
               if (dist > 1.0) dist = 1;
               if (dist < 0.0) dist = 0.0;
               const double d255 = dist * 255.0, dtr = trunc(d255);
               const uint32_t c = (uint32_t) (int) dtr + (d255 - dtr >= 0.5 ? 1u : 0u);
               dst[q] = c | c << 8 | c << 16 | 0xff000000u;
            }
         }
      }
      /* a lane's sum fits 32 bits (at most 32 pixels of at most 2.55 M each), the wavefront's does while a lane has at most 16 */
      unsigned long long ksum64 = ksum;
      if (wide)
      {
#pragma unroll
         for (int d = 32; d > 0; d >>= 1) { ksum64 += (unsigned long long) __shfl_down((long long) ksum64, d); cnt += __shfl_down(cnt, d); }
      }
      else
      {
#pragma unroll
         for (int d = 32; d > 0; d >>= 1) { ksum += (uint32_t) __shfl_down((int) ksum, d); cnt += __shfl_down(cnt, d); }
         ksum64 = ksum;
      }
      if (lane == 0 && cnt) { atomicAdd(&ksums[i], ksum64); atomicAdd(&cnts[i], (uint32_t) cnt); }
      /* the wavefronts of a caption stay on one angle: no data depends on it, the band they sample stays in cache */
      if (nwaves > 1 && (i % SMR_CAPROT_SYNC_EVERY) == SMR_CAPROT_SYNC_EVERY - 1) __syncthreads();
   }
   __syncthreads();
   for (int i = tid; i < A.num_steps; i += nthr)
   {
      double avg = (double) (long long) ksums[i] / 2550000.0;
      if (cnts[i]) avg /= (int) cnts[i];
      dv[i] = avg;
      if (A.out_dv) A.out_dv[cap * A.num_steps + i] = avg;
   }
   __syncthreads();

   /* dv_weight, dv_quantize, dp_get, widest peak: a few hundred scalar operations */
   if (tid == 0)
   {
      const int n = A.num_steps;
      double mn = 1.0, mx = 0.0;
      for (int i = 0; i < n; i++)
      {
         double v = dv[i];
         if (A.weight < 0.0) v = 1.0 - v;
         v *= A.wfac[i];
         dv[i] = v;
         if (v > mx) mx = v;
         if (v < mn) mn = v;
      }
      for (int i = 0; i < n; i++)
      {
         double q = (dv[i] - mn) / (mx - mn);
         if (isnan(q)) q = 1.0;
         dv[i] = q;
      }
      int peak = dv[0] >= SMR_DP_LIMIT ? 1 : 0, pcnt = 0, last = n;
      double best_w = -1.0, best_s = 0, best_e = 0, cur_s = 0;
      for (int i = 1; i <= last; i++)
      {
         int a = i % n, b = (i - 1) % n;
         if (peak && dv[a] < SMR_DP_LIMIT)
         {
            peak = 0;
            if (pcnt)
            {
               double e;
               if (A.ang[a] > A.ang[b]) e = (A.ang[a] + A.ang[b]) / 2;
               else e = (A.ang[a] + A.ang[b] + M_2PI) / 2;
               if (e < cur_s) e += M_2PI;
               /* qsort by width, descending; glibc's merge sort is stable, so the first of equals wins */
               if (e - cur_s > best_w) { best_w = e - cur_s; best_s = cur_s; best_e = e; }
            }
            continue;
         }
         if (!peak && dv[a] >= SMR_DP_LIMIT)
         {
            peak = 1;
            if (!pcnt) last = i + n - 1;
            pcnt++;
            if (A.ang[a] > A.ang[b]) cur_s = (A.ang[a] + A.ang[b]) / 2;
            else cur_s = (A.ang[a] + A.ang[b] + M_2PI) / 2;
         }
      }
      A.out_angle[cap] = best_w >= 0.0 ? M_2PI - (best_e + best_s) / 2 : 0.0;
   }
}

static uint32_t cap_color_pixel(uint32_t c)
{
   double r = (double) ((c >> 16) & 0xff) / 255.0, g = (double) ((c >> 8) & 0xff) / 255.0, b = (double) (c & 0xff) / 255.0;
   double a = 1.0 - (double) (((c & 0x7f000000u) >> 23) & 0xff) / 255.0;
   auto d2s = [](double d) { return (uint32_t) (uint16_t) (d * 65535.0 + 0.5); };      /* _cairo_color_double_to_short of libcairo 1.16 */
   return ((d2s(a) >> 8) << 24) | ((d2s(r * a) >> 8) << 16) | ((d2s(g * a) >> 8) << 8) | (d2s(b * a) >> 8);
}

/* what both entry points check and derive from a batch */
static int caprot_geometry(smr_canvas_t *c, const smr_cap_batch_t *b, double *d_angle, int *n_steps, int *r, size_t *lds)
{
   if (!c || !b || !d_angle) return smr_fail(SMR_E_INVAL, "caprot: NULL argument");
   if (b->n < 0 || b->fg_w <= 0 || b->fg_h <= 0 || b->fg_x < 0 || b->fg_x > b->fg_w)
      return smr_fail(SMR_E_INVAL, "caprot: bad plane geometry");
   if (b->n > 0 && (!b->d_cx || !b->d_cy)) return smr_fail(SMR_E_INVAL, "caprot: missing caption coordinates");
   SMR_TRY(smr_hip_caprot_steps(b->fg_w, b->fg_h, c->dpi, n_steps, r));
   if (*n_steps < 1) return smr_fail(SMR_E_INVAL, "caprot: plane too small (num_steps = %d)", *n_steps);
   /* dv, the sums and the counts per angle (20 B), the 256-entry table, the plane's opaque pixels */
   *lds = (((size_t) *n_steps * 20 + 2048 + 15) & ~(size_t) 15) + (size_t) (b->fg_w - (b->d_fg ? 0 : b->fg_x)) * b->fg_h * 4;
   if (*lds > 64 * 1024)
      return smr_fail(SMR_E_CAPACITY, "caprot: plane of %dx%d px needs %zu bytes of LDS (limit 65536)", b->fg_w, b->fg_h, *lds);
   return SMR_OK;
}

/* per-batch tables with the host libm: angles accumulate like dv_sample's loop (:1842) */
static void caprot_tables(const smr_cap_batch_t *b, int n_steps, int r, smr_rotmat_t *mats, double *ang, double *wfac)
{
   double a = 0, phase = b->phase_deg * M_PI / 180.0;
   for (int i = 0; i < n_steps; a += M_2PI / n_steps, i++)
   {
      ang[i] = a;
      rot_matrix(b->fg_w, b->fg_h, r, r, a, &mats[i]);
      wfac[i] = 1.0 - (1.0 - fabs(b->weight)) * (1.0 - cos(a * 2.0 + phase)) / 2.0;      /* dv_weight :1819 */
   }
}

/* the launch itself; d_m / d_ang / d_wfac: the per-batch tables in device memory, already on their way on the stream */
static int caprot_launch(smr_canvas_t *c, const smr_cap_batch_t *b, double *d_angle, double *d_dv, int n_steps, int r, size_t lds,
                         const smr_rotmat_t *d_m, const double *d_ang, const double *d_wfac, bool timed)
{
   hipStream_t s = smr_g.stream;
   caprot_args_t A;
   A.canvas = c->pix; A.W = c->W; A.H = c->H; A.band_y0 = c->band_y0; A.band_h = c->band_h;
   A.halo_above = c->halo_above; A.halo_below = c->halo_below; A.n_above = c->n_above; A.n_below = c->n_below;
   /* a caption owned by this band (centre inside its rows; on a canvas made by smr_hip_canvas_extend: inside the rows of the
    * band it was made from) reads at most r/2 + 1 rows beyond them */
   {
      const int own_y0 = c->own_y0 >= 0 ? c->own_y0 : c->band_y0, own_h = c->own_y0 >= 0 ? c->own_h : c->band_h;
      const int need = r / 2 + 2;
      const int need_above = need < own_y0 ? need : own_y0;
      const int rest = c->H - own_y0 - own_h, need_below = need < rest ? need : rest;
      const int have_above = own_y0 - c->band_y0 + c->n_above, have_below = c->band_y0 + c->band_h - own_y0 - own_h + c->n_below;
      if (have_above < need_above || have_below < need_below)
         return smr_fail(SMR_E_INVAL, "caprot: this band needs %d halo rows above and %d below for a cut-out of %d px "
                         "(smr_hip_canvas_set_halo)", need_above, need_below, r);
   }
   A.cx = b->d_cx; A.cy = b->d_cy; A.n = b->n;
   A.fg_w = b->fg_w; A.fg_h = b->fg_h; A.fg_x = b->fg_x; A.r = r; A.num_steps = n_steps;
   A.fg_pix = cap_color_pixel(b->col); A.fg = b->d_fg; A.weight = b->weight;
   A.mats = d_m; A.ang = d_ang; A.wfac = d_wfac;
   A.out_angle = d_angle; A.out_dv = d_dv;
   smr_timer_t T(timed && smr_g.profiling);
   T.mark(s);
   /* wavefronts per caption: one per 4 x 16-px block of the plane, at most 16 */
   const int cw = b->fg_w - (b->d_fg ? 0 : b->fg_x), nblk = ((cw + 15) >> 4) * ((b->fg_h + 3) >> 2);
   /* four measured best at C4's shape (45 blocks: 14.1 ms per 20 000 captions against 16.4 with 2, 15.4 with 8, 23.3 with 16); more
    * for planes of more than 128 blocks, so that a wavefront's blocks stay within the 32 bits of the kernel's rawmask */
   int nwaves = (nblk + 31) / 32 < 4 ? 4 : (nblk + 31) / 32;
   if (nwaves > nblk) nwaves = nblk < 1 ? 1 : nblk;
   if (nwaves > SMR_CAPROT_MAXWAVES) nwaves = SMR_CAPROT_MAXWAVES;
   k_caprot<<<(unsigned) b->n, 64 * nwaves, lds, s>>>(A);
   SMR_HIP(hipGetLastError());
   T.mark(s);
   if (timed && smr_g.profiling) smr_g.last.caprot_ms = T.ms(0, 1);
   return SMR_OK;
}

extern "C" int smr_hip_caprot(smr_canvas_t *c, const smr_cap_batch_t *b, double *d_angle, double *d_dv)
{
   SMR_NEED_INIT();
   int n_steps, r; size_t lds;
   SMR_TRY(caprot_geometry(c, b, d_angle, &n_steps, &r, &lds));
   if (b->n == 0) return SMR_OK;
   std::vector<smr_rotmat_t> mats(n_steps);
   std::vector<double> ang(n_steps), wfac(n_steps);
   caprot_tables(b, n_steps, r, mats.data(), ang.data(), wfac.data());
   hipStream_t s = smr_g.stream;
   smr_rotmat_t *d_m; double *d_t;
   SMR_TRY(c->ensure(SMR_BUF_CAP0, sizeof(smr_rotmat_t) * n_steps, (void **) &d_m));
   SMR_TRY(c->ensure(SMR_BUF_CAP1, sizeof(double) * 2 * n_steps, (void **) &d_t));
   SMR_HIP(hipMemcpyAsync(d_m, mats.data(), sizeof(smr_rotmat_t) * n_steps, hipMemcpyHostToDevice, s));
   SMR_HIP(hipMemcpyAsync(d_t, ang.data(), sizeof(double) * n_steps, hipMemcpyHostToDevice, s));
   SMR_HIP(hipMemcpyAsync(d_t + n_steps, wfac.data(), sizeof(double) * n_steps, hipMemcpyHostToDevice, s));
   SMR_HIP(hipStreamSynchronize(s));      /* the host vectors go out of scope below */
   return caprot_launch(c, b, d_angle, d_dv, n_steps, r, lds, d_m, d_t, d_t + n_steps, true);
}

/* ------------------------------------------------------------------------------------ */
/* smr_hip_caprot_noalloc: the same scan for callers that cannot allocate (tight serving   */
/* loops, harnesses that forbid hipMalloc; not capture-safe: it syncs).  The scan's only    */
/* scratch, the per-batch tables (rotation matrices, angles, weight factors), lives in a    */
/* file-scope device array instead of the canvas's grow-only buffers; the host copies are   */
/* file-scope too.  Each GPU holds its own copy of the array at its own address: the        */
/* address is asked for once per device (smr_caprot_scratch_setup, called by smr_hip_init    */
/* with that device current) and only that pointer is ever handed to a kernel or a copy.     */
/* ------------------------------------------------------------------------------------ */
struct smr_caprot_tables_t
{
   smr_rotmat_t mats[SMR_CAPROT_NOALLOC_MAX_STEPS];
   double ang[SMR_CAPROT_NOALLOC_MAX_STEPS], wfac[SMR_CAPROT_NOALLOC_MAX_STEPS];
};
static __device__ __align__(256) unsigned char scratch[sizeof(smr_caprot_tables_t)];      /* 48 KiB per GPU, until the process ends */
#define SMR_MAX_DEVICES 64
static smr_caprot_tables_t *scratch_ptr[SMR_MAX_DEVICES];      /* per device: the array's address ON that device */
static smr_caprot_tables_t scratch_host;                       /* the tables on their way to the device */

int smr_caprot_scratch_setup(int device)
{
   if (device < 0 || device >= SMR_MAX_DEVICES) return smr_fail(SMR_E_INVAL, "caprot_noalloc: device %d (at most %d)", device, SMR_MAX_DEVICES);
   if (scratch_ptr[device]) return SMR_OK;
   void *p = nullptr;
   SMR_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
   if (!p) return smr_fail(SMR_E_HIP, "caprot_noalloc: hipGetSymbolAddress returned NULL");
   scratch_ptr[device] = static_cast<smr_caprot_tables_t *>(p);
   return SMR_OK;
}

extern "C" int smr_hip_caprot_noalloc(smr_canvas_t *c, const smr_cap_batch_t *b, double *d_angle, double *d_dv)
{
   SMR_NEED_INIT();
   int n_steps, r; size_t lds;
   SMR_TRY(caprot_geometry(c, b, d_angle, &n_steps, &r, &lds));
   if (n_steps > SMR_CAPROT_NOALLOC_MAX_STEPS)
      return smr_fail(SMR_E_CAPACITY, "caprot_noalloc: %d trial angles, the static tables hold %d (use smr_hip_caprot)", n_steps,
                      SMR_CAPROT_NOALLOC_MAX_STEPS);
   if (b->n == 0) return SMR_OK;
   smr_caprot_tables_t *d = smr_g.device >= 0 && smr_g.device < SMR_MAX_DEVICES ? scratch_ptr[smr_g.device] : nullptr;
   if (!d) return smr_fail(SMR_E_NODEVICE, "caprot_noalloc: no scratch address for device %d (smr_hip_init)", smr_g.device);
   hipStream_t s = smr_g.stream;
   /* the previous call's copies have left scratch_host: every call ends with the kernel enqueued behind them, and syncs here first */
   SMR_HIP(hipStreamSynchronize(s));
   caprot_tables(b, n_steps, r, scratch_host.mats, scratch_host.ang, scratch_host.wfac);
   /* every word the kernel reads ([0, n_steps) of each table) is rewritten: nothing of the last call's contents is used */
   SMR_HIP(hipMemcpyAsync(d->mats, scratch_host.mats, sizeof(smr_rotmat_t) * n_steps, hipMemcpyHostToDevice, s));
   SMR_HIP(hipMemcpyAsync(d->ang, scratch_host.ang, sizeof(double) * n_steps, hipMemcpyHostToDevice, s));
   SMR_HIP(hipMemcpyAsync(d->wfac, scratch_host.wfac, sizeof(double) * n_steps, hipMemcpyHostToDevice, s));
   SMR_HIP(hipStreamSynchronize(s));      /* pageable source: the copies are staged before scratch_host may change */
   return caprot_launch(c, b, d_angle, d_dv, n_steps, r, lds, d->mats, d->ang, d->wfac, false);
}
