/* windows.hip -- the caption scan's cut-outs the reference's way: re-rasterised at a sub-pixel offset.
 *
 * cairo_smr_cut_out (src/smrules_cairo.c:1511-1533) replays the chart's recording into an r x r image under
 * cairo_scale(dpi/72) with the translation (-x + r/2, -y + r/2) pt, x, y the caption's anchor: whatever fraction of a pixel
 * the anchor has, the cut-out's pixel grid is shifted by it against the chart's.  libcairo 1.16 transforms the recorded
 * 24.8 path as p' = fixed_mul(scale, p) + fixed(translation * scale) (_cairo_path_fixed_offset_and_scale; the translation is
 * quantised by itself), builds strokes on the shifted path and limits every polygon to the r x r image.  So a cut-out is the
 * chart's own device geometry plus ONE constant 24.8 vector per caption, clipped to the window -- which is what this file
 * renders, for many captions at once:
 *
 *   atlas      an ordinary canvas of r columns holding the windows one below the other (pitch = r rounded up to the tile
 *              height), created transparent; the chart's background is drawn into it as the first "rule" (one rectangle)
 *   instances  for every rule drawn so far (style + batch, kept by the caller) k_win_pairs finds the (window, way) pairs whose
 *              boxes meet, window-major and in way order inside a window (the painter's order), k_win_fill copies the ways'
 *              nodes into an instance batch: per node its window, per way the atlas rows of its window
 *   draw       the instance batch goes through the same draw pass as any rule: k_dev_points adds the window's shift to the
 *              device points, k_prim_setup limits a way's primitives to its window's rows (the columns are the canvas's)
 *   scan       smr_hip_caprot on the atlas with the anchors smr_hip_windows_anchors gives (the middle of each window)
 *
 * Pinned against libcairo's own cut-outs: tests/golden/cutout_subpixel.npz (tests/test_gpu_windows.py). */
#include <string.h>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "smr_host.h"
#include "smr_dev.h"

#define SMR_WIN_WAYS_PER_BLOCK 1024

struct win_box_t { double x0, y0, x1, y1; };

/* one wavefront per way, lanes = refs (coalesced reads of the way's nodes): its box in chart pixels */
__global__ __launch_bounds__(256) void k_win_bbox(const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ way_off,
                                                  const int32_t *__restrict__ ref, int64_t n_ways, int64_t n_nodes, double curve_f, win_box_t *__restrict__ wb)
{
   const int lane = threadIdx.x & 63;
   const int64_t w = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
   if (w >= n_ways) return;
   double x0 = 1e300, y0 = 1e300, x1 = -1e300, y1 = -1e300, lmax = 0;
   const int64_t a = way_off[w], e = way_off[w + 1];
   if (e - a >= 2)
      for (int64_t g0 = a; g0 < e; g0 += 64)
      {
         const int64_t g = g0 + lane;
         double vx = 0, vy = 0; bool ok = false;
         if (g < e)
         {
            const int64_t node = ref ? (int64_t) ref[g] : g;
            if (node >= 0 && node < n_nodes) { vx = x[node]; vy = y[node]; ok = true; }          /* (a bad ref is reported by the draw pass itself) */
         }
         if (ok) { x0 = fmin(x0, vx); x1 = fmax(x1, vx); y0 = fmin(y0, vy); y1 = fmax(y1, vy); }
         if (curve_f > 0)
         {
            /* the longest segment (between consecutive refs; a missing node's neighbours are compared across it by the serial
             * walk the reference does, here the two segments touching it are simply left out: the bound only has to be safe for
             * ways the draw pass accepts) */
            const double nx = __shfl_down(vx, 1), ny = __shfl_down(vy, 1);
            const int nok = __shfl_down((int) ok, 1);
            double px = nx, py = ny; int pok = nok;
            if (lane == 63)
            {
               pok = 0;
               if (g + 1 < e) { const int64_t node = ref ? (int64_t) ref[g + 1] : g + 1; if (node >= 0 && node < n_nodes) { px = x[node]; py = y[node]; pok = 1; } }
            }
            if (ok && pok && g + 1 < e) lmax = fmax(lmax, hypot(vx - px, vy - py));
         }
      }
#pragma unroll
   for (int d = 32; d >= 1; d >>= 1)
   {
      x0 = fmin(x0, __shfl_xor(x0, d)); y0 = fmin(y0, __shfl_xor(y0, d)); x1 = fmax(x1, __shfl_xor(x1, d)); y1 = fmax(y1, __shfl_xor(y1, d));
      lmax = fmax(lmax, __shfl_xor(lmax, d));
   }
   if (lane) return;
   win_box_t b = { x0, y0, x1, y1 };
   /* curve=1: a Bezier's control points lie curve_factor times the length of their segment beyond its ends (control_points,
    * src/bspline_ctrl.c:64-105: lgt * f along the tangent), and the curve inside the hull of its four knots: the way's box grows
    * by the longest segment's share (a long segment's curve can pass a window that its nodes' box misses) */
   if (curve_f > 0 && b.x0 <= b.x1) { const double m = curve_f * lmax + 0.25; b.x0 -= m; b.y0 -= m; b.x1 += m; b.y1 += m; }
   wb[w] = b;
}

/* The (window, way) pairs of a rule: one thread per way looks its box (grown by the rule's reach) up in the atlas's grid of
 * windows -- a way meets a handful of cells, a cell holds a handful of windows -- instead of every way being compared with
 * every window (240 M box tests per atlas of a C2 chart, twice, were a third of the scan's time).  A window lies in up to
 * four cells: the pair counts in the cell that holds the upper left corner of the two boxes' intersection.  Pairs leave as
 * keys window << 32 | way through one atomic counter; sorted, they are window-major and in way order inside a window: the
 * painter's order of the chart in every window. */
struct win_grid_t { const uint32_t *off, *win; int nx, ny, cell; };

__global__ __launch_bounds__(256) void k_win_pairs(const win_box_t *__restrict__ wb, int64_t n_ways, const win_box_t *__restrict__ win,
                                                   win_grid_t G, double reach, int k0, int k1, unsigned long long *__restrict__ keys,
                                                   uint32_t cap, uint32_t *__restrict__ ctr)
{
   __shared__ uint32_t sc[256];
   __shared__ uint32_t sbase;
   const int t = threadIdx.x;
   const int64_t w = (int64_t) blockIdx.x * blockDim.x + t;
   win_box_t b = { 1.0, 1.0, 0.0, 0.0 };
   if (w < n_ways) b = wb[w];
   const bool have = b.x0 <= b.x1;
   b.x0 -= reach; b.y0 -= reach; b.x1 += reach; b.y1 += reach;
   const double rc = 1.0 / G.cell;
   auto cellof = [&](double v, int n) { const double c = floor(v * rc) + 1.0; return c < 0 ? 0 : c > n - 1 ? n - 1 : (int) c; };
   const int cx0 = cellof(b.x0, G.nx), cx1 = cellof(b.x1, G.nx), cy0 = cellof(b.y0, G.ny), cy1 = cellof(b.y1, G.ny);
   /* the same walk twice: count, then (one atomic per block: a hundred thousand on one address took longer than everything else)
    * write at the block's base plus the thread's rank */
   uint32_t n = 0, pos = 0;
   for (int pass = 0; pass < 2; pass++)
   {
      if (have)
         for (int cy = cy0; cy <= cy1; cy++)
            for (int cx = cx0; cx <= cx1; cx++)
            {
               const uint32_t c = (uint32_t) cy * G.nx + cx;
               for (uint32_t i = G.off[c]; i < G.off[c + 1]; i++)
               {
                  const int k = (int) G.win[i];
                  if (k < k0 || k >= k1) continue;
                  const win_box_t W = win[k];
                  if (!(b.x1 >= W.x0 && b.x0 <= W.x1 && b.y1 >= W.y0 && b.y0 <= W.y1)) continue;
                  if (cellof(fmax(b.x0, W.x0), G.nx) != cx || cellof(fmax(b.y0, W.y0), G.ny) != cy) continue;      /* counted in another cell */
                  if (pass == 0) n++;
                  else { if (pos < cap) keys[pos] = ((unsigned long long) (uint32_t) k << 32) | (uint32_t) w; pos++; }
               }
            }
      if (pass == 1) break;
      sc[t] = n;
      __syncthreads();
      for (int d = 1; d < 256; d <<= 1)
      {
         const uint32_t v = t >= d ? sc[t - d] : 0;
         __syncthreads();
         sc[t] += v;
         __syncthreads();
      }
      if (t == 255) sbase = sc[255] ? atomicAdd(ctr, sc[255]) : 0u;
      __syncthreads();
      if (sc[255] == 0) return;
      pos = sbase + sc[t] - n;
   }
}

__global__ void k_win_inst(const unsigned long long *__restrict__ keys, uint32_t n_inst, const int64_t *__restrict__ way_off,
                           uint32_t *__restrict__ inst_way, uint32_t *__restrict__ inst_win, uint32_t *__restrict__ inst_nref)
{
   const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n_inst) return;
   const unsigned long long k = keys[i];
   const uint32_t w = (uint32_t) k;
   inst_way[i] = w; inst_win[i] = (uint32_t) (k >> 32); inst_nref[i] = (uint32_t) (way_off[w + 1] - way_off[w]);
}

struct win_fill_args_t
{
   const double *x, *y; const int64_t *way_off; const int32_t *ref; int64_t n_nodes;
   const uint32_t *way_fill_col, *way_border_col;
   const uint32_t *inst_way, *inst_win, *inst_off;      /* inst_off[n_inst] = all refs */
   uint32_t n_inst, n_out;
   const int2 *win_clip; const int4 *win_lim;
   double *ox, *oy; int32_t *oref, *onode_win; int64_t *oway_off; int2 *oway_clip; int4 *oway_lim; uint32_t *ofill_col, *oborder_col;
};

/* one thread per ref of the instance batch */
__global__ void k_win_fill(win_fill_args_t A)
{
   const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
   if (g <= A.n_inst)
   {
      A.oway_off[g] = (int64_t) A.inst_off[g];
      if (g < A.n_inst)
      {
         A.oway_clip[g] = A.win_clip[A.inst_win[g]];
         A.oway_lim[g] = A.win_lim[A.inst_win[g]];
         if (A.ofill_col) A.ofill_col[g] = A.way_fill_col[A.inst_way[g]];
         if (A.oborder_col) A.oborder_col[g] = A.way_border_col[A.inst_way[g]];
      }
   }
   if (g >= A.n_out) return;
   uint32_t lo = 0, hi = A.n_inst;          /* inst_off[lo] <= g < inst_off[hi] */
   while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (A.inst_off[mid] <= g) lo = mid; else hi = mid; }
   const uint32_t i = lo, t = g - A.inst_off[i], n = A.inst_off[i + 1] - A.inst_off[i];
   const int64_t w = A.inst_way[i], a = A.way_off[w];
   int64_t node = A.ref ? (int64_t) A.ref[a + t] : a + t;
   if (node < 0 || node >= A.n_nodes) node = 0;             /* the unshifted draw of the same batch has reported it */
   A.ox[g] = A.x[node]; A.oy[g] = A.y[node];
   A.onode_win[g] = (int32_t) A.inst_win[i];
   /* is_closed_poly compares the first and the last REF: the instance's last ref names its first node when the way's does */
   int32_t r = (int32_t) g;
   if (t == n - 1 && n >= 2)
   {
      const int64_t first = A.ref ? (int64_t) A.ref[a] : a, last = A.ref ? (int64_t) A.ref[a + n - 1] : a + n - 1;
      if (first == last) r = (int32_t) A.inst_off[i];
   }
   A.oref[g] = r;
}

/* the reach of a rule's strokes beyond its ways' nodes, in px: what _cairo_stroke_style_max_distance_from_path allows for
 * (miter limit 10), for the widest stroke the style can make, plus slack for curve=1 control points and the AA pixel */
static double style_reach_px(const smr_draw_style_t *st, int dpi)
{
   const double px_per_mm = dpi / 25.4;
   double wmax = 0;
   if (st->border_used) wmax = fmax(wmax, (2 * st->border_width_mm + st->fill_width_mm) * px_per_mm);
   if (st->fill_used) wmax = fmax(wmax, st->fill_width_mm * px_per_mm);
   wmax = fmax(wmax, 1.0);
   return wmax * M_SQRT2 * 10.0 * 0.5 + 3.0;
}

/* The chart's background in every window, without a draw pass: the background is ONE rectangle, the chart's, and what the scan
 * converter makes of an axis-parallel rectangle has a closed form.  A sub-row ys counts when ytop <= ys < ybot (the edges' grid
 * rows, smr_to_grid_y); on every counted sub-row the span runs from the left side's x to the right side's (vertical edges sit on
 * their own x, no rounding), and a vertex-free row's exact path gives 15 times the same numbers; so a pixel's area is
 * (counted sub-rows) * 2 * (its share of the span in 1/256 px), its alpha libcairo's (17 area + 256) >> 9, and the pixel what
 * k_raster's direct blend makes of a clear canvas.  The four numbers per window are the rectangle's device coordinates, made
 * exactly as k_dev_points makes them (windows_setup).  SMR_HIP_WINBG=draw sends the rectangle through the draw pass instead
 * (the two must agree: tests/test_gpu_windows.py). */
struct win_bgrect_t { int32_t x0, y0, x1, y1; };      /* 24.8, atlas space (the window's rows included) */

__global__ void k_win_bgfill(uint32_t *__restrict__ pix, int W, int pitch, const win_bgrect_t *__restrict__ rect, const int2 *__restrict__ clip,
                             uint32_t spix)
{
   const int k = blockIdx.y;
   const win_bgrect_t q = rect[k];
   const int r0 = clip[k].x, rows = clip[k].y - clip[k].x;
   const int ytop = smr_to_grid_y(q.y0), ybot = smr_to_grid_y(q.y1);
   const bool lerp = (spix >> 24) == 255u;
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * W; i += gridDim.x * blockDim.x)
   {
      const int x = i % W, row = r0 + i / W;
      const int s0 = max(ytop, row * SMR_GRID_Y), s1 = min(ybot, (row + 1) * SMR_GRID_Y);
      const int xl = min(max(q.x0 - x * 256, 0), 256), xr = min(max(q.x1 - x * 256, 0), 256);
      uint32_t out = 0u;
      /* libcairo composites inside the polygon's pixel extents only: floor of the least, ceil of the greatest coordinate */
      if (s1 > s0 && xr > xl && row >= (q.y0 >> 8) && row < ((q.y1 + 255) >> 8))
      {
         const int area = (s1 - s0) * 2 * (xr - xl);
         int a = (area + (area << 4) + 256) >> 9;
         a = a > 255 ? 255 : a;
         if (a) out = lerp ? smr_blend_lerp(spix, (uint32_t) a, 0u) : smr_blend_over_mask(spix, (uint32_t) a, 0u);
      }
      pix[(size_t) row * W + x] = out;
   }
}

/* the atlas's windows: tables, grid, clear pixels, the chart's background in every window */
static int windows_setup(smr_canvas_t *c, const smr_windows_t *w, int maxsize)
{
   const int pitch = c->win_pitch, dpi = c->dpi, chart_w = c->chart_w, chart_h = c->chart_h;
   SMR_HIP(hipMemsetAsync(c->pix, 0, (size_t) c->W * (size_t) c->band_h * 4, smr_g.stream));      /* a clear image: transparent */
   c->win_n = (int) w->n; c->win_size = maxsize;
   c->win_sizes.assign(w->size, w->size + w->n);
   c->win_ox.clear(); c->win_oy.clear();
   std::vector<int2> shift((size_t) w->n);
   std::vector<win_box_t> box((size_t) w->n);
   const double s = (double) dpi / 72;
   /* find_angle :1974-1975, cairo_smr_cut_out :1516-1528 with r = the window's size in pt */
   std::vector<int2> clip((size_t) w->n);
   /* What libcairo limits a polygon of the replay to (cuts at the sides add vertices, which decide the rows its scan converter
    * sub-samples).  The image compositor does not replay the recording onto the cut-out's image: it makes the source's pixels first
    * (_pixman_image_for_recording of cairo-image-source.c) in an image of its own that covers the SAMPLED area of the source, whole
    * units of the source, i.e. whole pt (_cairo_pattern_sampled_area: the centres of the corner pixels through the pattern's
    * matrix, 0.495 of a unit further out for a bilinear filter, floor and ceil), taken back to device pixels with floor and ceil
    * again: at 300 dpi its sides lie 2 to 6 px beyond the cut-out's own.  (oracle: orc_canvas_set_shift.  Top and bottom change no
    * row inside the cut-out, but they belong to the image a decomposed curve is tested against, k_curve; near the page's sides the sampled area is the recording's extents at most: left out.) */
   std::vector<int4> lim((size_t) w->n);
   /* the grid: cells as large as the largest window, one ring of cells around the chart for windows that hang over its edge */
   const int cell = std::max(maxsize, 64);
   const int nx = (chart_w + cell - 1) / cell + 2, ny = (chart_h + cell - 1) / cell + 2;
   auto cellof = [&](double v, int n) { const double q = floor(v / cell) + 1.0; return q < 0 ? 0 : q > n - 1 ? n - 1 : (int) q; };
   std::vector<uint32_t> goff((size_t) nx * ny + 1, 0u), gwin;
   for (int64_t k = 0; k < w->n; k++)
   {
      const double r_pt = (w->r_px && w->r_px[k] > 0 ? w->r_px[k] : (double) w->size[k]) * 72 / dpi;
      clip[(size_t) k] = make_int2((int) k * pitch, (int) k * pitch + w->size[k]);
      const double x_pt = w->cx[k] * 72 / dpi, y_pt = w->cy[k] * 72 / dpi;
      const double tx = -x_pt + r_pt / 2, ty = -y_pt + r_pt / 2;
      shift[(size_t) k] = make_int2((int32_t) nearbyint(tx * s * 256.0), (int32_t) nearbyint(ty * s * 256.0) + (int32_t) k * pitch * 256);
      {
         const double X = -tx, Y = -ty;                             /* the source coordinates of the cut-out's left side and top */
         const double sx1 = floor(0.5 / s + X - 0.495), sx2 = floor(((double) w->size[k] - 0.5) / s + X + 0.495) + 1.0;
         const double sy1 = floor(0.5 / s + Y - 0.495), sy2 = floor(((double) w->size[k] - 0.5) / s + Y + 0.495) + 1.0;
         lim[(size_t) k] = make_int4((int32_t) floor((sx1 - X) * s) * 256, (int32_t) ceil((sx2 - X) * s) * 256,
                                     ((int32_t) floor((sy1 - Y) * s) + (int32_t) k * pitch) * 256, ((int32_t) ceil((sy2 - Y) * s) + (int32_t) k * pitch) * 256);
      }
      /* the window on the chart, in chart px (what lies further than a rule's reach from it cannot touch it) */
      box[(size_t) k] = { -tx * s, -ty * s, -tx * s + c->W, -ty * s + w->size[k] };
      c->win_ox.push_back(-tx * s); c->win_oy.push_back(-ty * s);
      const win_box_t &b = box[(size_t) k];
      for (int cy = cellof(b.y0, ny); cy <= cellof(b.y1, ny); cy++)
         for (int cx = cellof(b.x0, nx); cx <= cellof(b.x1, nx); cx++) goff[(size_t) cy * nx + cx + 1]++;
   }
   for (size_t i = 1; i < goff.size(); i++) goff[i] += goff[i - 1];
   gwin.resize(goff.back() ? goff.back() : 1);
   {
      std::vector<uint32_t> fill(goff.begin(), goff.end() - 1);
      for (int64_t k = 0; k < w->n; k++)
      {
         const win_box_t &b = box[(size_t) k];
         for (int cy = cellof(b.y0, ny); cy <= cellof(b.y1, ny); cy++)
            for (int cx = cellof(b.x0, nx); cx <= cellof(b.x1, nx); cx++) gwin[fill[(size_t) cy * nx + cx]++] = (uint32_t) k;
      }
   }
   if (c->grid_nx != nx || c->grid_ny != ny || !c->d_grid_off)
   {
      if (c->d_grid_off) { hipStreamSynchronize(smr_g.stream); hipFree(c->d_grid_off); c->d_grid_off = nullptr; }
      if (hipMalloc((void **) &c->d_grid_off, goff.size() * 4) != hipSuccess) return smr_fail(SMR_E_NOMEM, "windows: grid");
   }
   if (c->grid_win_cap < gwin.size())
   {
      if (c->d_grid_win) { hipStreamSynchronize(smr_g.stream); hipFree(c->d_grid_win); c->d_grid_win = nullptr; }
      if (hipMalloc((void **) &c->d_grid_win, gwin.size() * 4 * 2) != hipSuccess) return smr_fail(SMR_E_NOMEM, "windows: grid lists");
      c->grid_win_cap = gwin.size() * 2;
   }
   c->grid_nx = nx; c->grid_ny = ny; c->grid_cell = cell;
   SMR_HIP(hipMemcpyAsync(c->d_grid_off, goff.data(), goff.size() * 4, hipMemcpyHostToDevice, smr_g.stream));
   SMR_HIP(hipMemcpyAsync(c->d_grid_win, gwin.data(), gwin.size() * 4, hipMemcpyHostToDevice, smr_g.stream));
   SMR_HIP(hipMemcpyAsync(c->d_win_shift, shift.data(), (size_t) w->n * sizeof(int2), hipMemcpyHostToDevice, smr_g.stream));
   SMR_HIP(hipMemcpyAsync(c->d_win_clip, clip.data(), (size_t) w->n * sizeof(int2), hipMemcpyHostToDevice, smr_g.stream));
   SMR_HIP(hipMemcpyAsync(c->d_win_lim, lim.data(), (size_t) w->n * sizeof(int4), hipMemcpyHostToDevice, smr_g.stream));
   SMR_HIP(hipMemcpyAsync(c->d_win_box, box.data(), (size_t) w->n * sizeof(win_box_t), hipMemcpyHostToDevice, smr_g.stream));
   SMR_HIP(hipStreamSynchronize(smr_g.stream));            /* the host vectors go out of scope */
   /* the chart's background: cairo_smr_recording_surface_from_bg paints it onto the recording, whose extents bound the paint;
    * in a cut-out that is the chart's rectangle at the window's offset (k_win_bgfill; below: the same through the draw pass) */
   {
      /* the rectangle's corners in every window's device space: px -> pt -> 24.8 -> the replay's fixed-point scale, plus the window's
       * shift (k_dev_points, raster.hip) */
      const int32_t sfix = (int32_t) nearbyint((double) dpi / 72 * 256.0);
      auto dev = [&](double v_px) { const int32_t f = (int32_t) rint(v_px * 72 / dpi * 256.0); return (int32_t) (((int64_t) f * sfix) >> 8); };
      std::vector<win_bgrect_t> rect((size_t) w->n);
      for (int64_t k = 0; k < w->n; k++)
         rect[(size_t) k] = { dev(0.0) + shift[(size_t) k].x, dev(0.0) + shift[(size_t) k].y, dev((double) chart_w) + shift[(size_t) k].x, dev((double) chart_h) + shift[(size_t) k].y };
      const char *e = getenv("SMR_HIP_WINBG");
      if (!(e && !strcmp(e, "draw")))
      {
         win_bgrect_t *d_rect;
         SMR_TRY(c->ensure(SMR_BUF_WIN_TMP, std::max<size_t>(256, rect.size() * sizeof(win_bgrect_t)), (void **) &d_rect));
         SMR_HIP(hipMemcpyAsync(d_rect, rect.data(), rect.size() * sizeof(win_bgrect_t), hipMemcpyHostToDevice, smr_g.stream));
         k_win_bgfill<<<dim3(64, (unsigned) w->n), 256, 0, smr_g.stream>>>(c->pix, c->W, pitch, d_rect, c->d_win_clip, smr_color_pixel(c->bg));
         SMR_HIP(hipGetLastError());
         SMR_HIP(hipStreamSynchronize(smr_g.stream));         /* rect goes out of scope */
         return SMR_OK;
      }
   }
   double rx[4] = { 0, (double) chart_w, (double) chart_w, 0 }, ry[4] = { 0, 0, (double) chart_h, (double) chart_h };
   int64_t off[2] = { 0, 5 };
   int32_t ref[5] = { 0, 1, 2, 3, 0 };
   uint8_t *d;                                             /* x[4] y[4] off[2] ref[5]: one small buffer of the canvas's own */
   SMR_TRY(c->ensure(SMR_BUF_WIN_TMP, 256, (void **) &d));
   SMR_HIP(hipMemcpyAsync(d, rx, 32, hipMemcpyHostToDevice, smr_g.stream));
   SMR_HIP(hipMemcpyAsync(d + 32, ry, 32, hipMemcpyHostToDevice, smr_g.stream));
   SMR_HIP(hipMemcpyAsync(d + 64, off, 16, hipMemcpyHostToDevice, smr_g.stream));
   SMR_HIP(hipMemcpyAsync(d + 80, ref, 20, hipMemcpyHostToDevice, smr_g.stream));
   SMR_HIP(hipStreamSynchronize(smr_g.stream));
   smr_draw_style_t st;
   memset(&st, 0, sizeof(st));
   st.fill_used = 1; st.fill_col = c->bg; st.direct = 1;
   smr_draw_batch_t b;
   memset(&b, 0, sizeof(b));
   b.d_x = (const double *) d; b.d_y = (const double *) (d + 32); b.n_nodes = 4; b.d_way_off = (const int64_t *) (d + 64); b.d_ref = (const int32_t *) (d + 80);
   b.n_ways = 1; b.n_refs = 5;
   const uint32_t keep_bg = c->bg;
   int rc = smr_hip_draw_windows(c, &st, &b);
   c->bg = keep_bg;
   SMR_HIP(hipStreamSynchronize(smr_g.stream));            /* (the batch sits in a work buffer the next draw may take) */
   return rc;
}

static int windows_check(const smr_windows_t *w, int *maxsize_out)
{
   if (!w || w->n <= 0 || !w->size || !w->cx || !w->cy) return smr_fail(SMR_E_INVAL, "windows: no windows");
   int maxsize = 0;
   for (int64_t k = 0; k < w->n; k++)
   {
      if (w->size[k] <= 0 || w->size[k] > 65536) return smr_fail(SMR_E_INVAL, "windows: window %lld of %d px", (long long) k, w->size[k]);
      maxsize = std::max(maxsize, w->size[k]);
   }
   *maxsize_out = maxsize;
   return SMR_OK;
}

extern "C" int smr_hip_windows_create(int chart_w, int chart_h, int dpi, uint32_t bgcolor, const smr_windows_t *w, smr_canvas_t **out)
{
   SMR_NEED_INIT();
   if (!out) return smr_fail(SMR_E_INVAL, "windows_create: NULL");
   *out = nullptr;
   int maxsize = 0;
   SMR_TRY(windows_check(w, &maxsize));
   if (chart_w <= 0 || chart_h <= 0 || dpi <= 0) return smr_fail(SMR_E_INVAL, "windows_create: bad chart %d x %d @ %d dpi", chart_w, chart_h, dpi);
   const int pitch = (maxsize + SMR_TH - 1) & ~(SMR_TH - 1);
   if ((int64_t) w->n * pitch > 65536)
      return smr_fail(SMR_E_CAPACITY, "windows_create: %lld windows of %d px do not fit one atlas (65 536 rows): at most %d per call",
                      (long long) w->n, maxsize, 65536 / pitch);
   smr_canvas_t *c = nullptr;
   SMR_TRY(smr_hip_canvas_create(maxsize, (int) w->n * pitch, dpi, 0, 0, (int) w->n * pitch, &c));
   c->win_pitch = pitch; c->chart_w = chart_w; c->chart_h = chart_h; c->win_cap = (int) w->n; c->bg = bgcolor;
   int rc = SMR_OK;
   if (hipMalloc((void **) &c->d_win_shift, (size_t) w->n * sizeof(int2)) != hipSuccess || hipMalloc((void **) &c->d_win_clip, (size_t) w->n * sizeof(int2)) != hipSuccess || hipMalloc((void **) &c->d_win_lim, (size_t) w->n * sizeof(int4)) != hipSuccess || hipMalloc((void **) &c->d_win_box, (size_t) w->n * sizeof(win_box_t)) != hipSuccess)
      rc = smr_fail(SMR_E_NOMEM, "windows_create: window tables");
   if (rc == SMR_OK) rc = windows_setup(c, w, maxsize);
   if (rc != SMR_OK) { smr_hip_canvas_destroy(c); return rc; }      /* (one way out for every failure after the canvas exists) */
   *out = c;
   return SMR_OK;
}

/* The atlas for the next batch of captions: the same chart, windows no larger and no more than it was created for.  Pixels
 * and tables are set up again; the canvas, its tables' memory and its work buffers stay (a C2 chart's scan makes and drops 84
 * atlases of 71 MB otherwise, each with its dozens of hipMalloc and hipFree).  No counterpart in the reference. */
extern "C" int smr_hip_windows_reset(smr_canvas_t *atlas, const smr_windows_t *w)
{
   SMR_NEED_INIT();
   if (!atlas || atlas->win_cap <= 0) return smr_fail(SMR_E_INVAL, "windows_reset: not an atlas");
   int maxsize = 0;
   SMR_TRY(windows_check(w, &maxsize));
   if (w->n > atlas->win_cap || maxsize > atlas->W || ((maxsize + SMR_TH - 1) & ~(SMR_TH - 1)) != atlas->win_pitch)
      return smr_fail(SMR_E_INVAL, "windows_reset: %lld windows of %d px into an atlas of %d windows of %d px (pitch %d)", (long long) w->n, maxsize,
                      atlas->win_cap, atlas->W, atlas->win_pitch);
   if (maxsize != atlas->W) return smr_fail(SMR_E_INVAL, "windows_reset: the largest window must be as wide as the atlas (%d px), not %d", atlas->W, maxsize);
   return windows_setup(atlas, w, maxsize);
}

/* the recording the reference replays has the chart's extents: nothing is drawn beyond them.  Pixels of a window whose centres lie
 * outside the chart's rectangle are cleared (whole pixels: the background rectangle's own edge is anti-aliased) */
__global__ void k_win_outside(uint32_t *__restrict__ pix, int W, int pitch, int n_win, const win_box_t *__restrict__ box, const int2 *__restrict__ clip,
                              int chart_w, int chart_h)
{
   const int k = blockIdx.y;
   const win_box_t b = box[k];
   const int rows = clip[k].y - clip[k].x;
   /* only windows that hang over an edge have anything to clear */
   if (b.x0 >= 0 && b.y0 >= 0 && b.x0 + W <= chart_w && b.y0 + rows <= chart_h) return;
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * W; i += gridDim.x * blockDim.x)
   {
      const int x = i % W, y = i / W;
      const double px = b.x0 + x + 0.5, py = b.y0 + y + 0.5;
      if (px < 0 || py < 0 || px > chart_w || py > chart_h) pix[(size_t) (k * pitch + y) * W + x] = 0u;
   }
}

extern "C" int smr_hip_windows_anchors(smr_canvas_t *atlas, double *cx, double *cy)
{
   if (!atlas || atlas->win_n <= 0 || !cx || !cy) return smr_fail(SMR_E_INVAL, "windows_anchors: not an atlas");
   k_win_outside<<<dim3(16, (unsigned) atlas->win_n), 256, 0, smr_g.stream>>>(atlas->pix, atlas->W, atlas->win_pitch, atlas->win_n,
                                                                              (const win_box_t *) atlas->d_win_box, atlas->d_win_clip, atlas->chart_w, atlas->chart_h);
   SMR_HIP(hipGetLastError());
   /* smr_hip_caprot crops at the anchor minus r/2, rounded: r/2 itself puts the crop on the window */
   for (int k = 0; k < atlas->win_n; k++) { cx[k] = atlas->win_sizes[(size_t) k] / 2.0; cy[k] = (double) k * atlas->win_pitch + atlas->win_sizes[(size_t) k] / 2.0; }
   return SMR_OK;
}

/* the rule into the windows k0 .. k1 - 1 of the atlas */
static int draw_windows_range(smr_canvas_t *c, const smr_draw_style_t *st, const smr_draw_batch_t *b, int k0, int k1)
{
   hipStream_t s = smr_g.stream;
   const int64_t n_ways = b->n_ways;
   const double reach = style_reach_px(st, c->dpi);
   const double curve_f = st->curve == 1 ? (st->curve_factor > 0 ? st->curve_factor : 0.2) : 0.0;      /* (curve=2: the knots are nodes themselves) */
   win_box_t *wb;
   SMR_TRY(c->ensure(SMR_BUF_WIN_WB, (size_t) n_ways * sizeof(win_box_t), (void **) &wb));
   k_win_bbox<<<(unsigned) ((n_ways + 3) / 4), 256, 0, s>>>(b->d_x, b->d_y, b->d_way_off, b->d_ref, n_ways, b->n_nodes, curve_f, wb);
   uint32_t *sums, *tot;
   SMR_TRY(c->ensure(SMR_BUF_WIN_TOT, 64, (void **) &tot));
   win_grid_t G;
   G.off = c->d_grid_off; G.win = c->d_grid_win; G.nx = c->grid_nx; G.ny = c->grid_ny; G.cell = c->grid_cell;
   uint32_t n_inst = 0;
   unsigned long long *keys = nullptr, *keys2 = nullptr;
   size_t cap = std::max<size_t>(c->buf[SMR_BUF_WIN_KEYS].cap / 8, (size_t) 1 << 16);
   for (int attempt = 0;; attempt++)
   {
      SMR_TRY(c->ensure(SMR_BUF_WIN_KEYS, cap * 8, (void **) &keys));
      SMR_HIP(hipMemsetAsync(tot, 0, 4, s));
      k_win_pairs<<<(unsigned) ((n_ways + 255) / 256), 256, 0, s>>>(wb, n_ways, (const win_box_t *) c->d_win_box, G, reach, k0, k1, keys,
                                                                   (uint32_t) std::min<size_t>(cap, 0xffffffffu), tot);
      SMR_HIP(hipGetLastError());
      SMR_HIP(hipMemcpyAsync(&n_inst, tot, 4, hipMemcpyDeviceToHost, s));
      SMR_HIP(hipStreamSynchronize(s));
      if (n_inst <= cap) break;
      if (attempt) return smr_fail(SMR_E_HIP, "draw_windows: internal: the pair count changed between two runs");
      cap = (size_t) n_inst + n_inst / 8;
   }
   if (n_inst == 0) return SMR_OK;
   if (n_inst > 0x7ffffff0u) return smr_fail(SMR_E_CAPACITY, "draw_windows: %u (window, way) pairs in one rule", n_inst);
   {
      /* window-major, ways ascending inside a window: 17 bits of window above 32 of way */
      SMR_TRY(c->ensure(SMR_BUF_WIN_KEYS2, (size_t) n_inst * 8, (void **) &keys2));
      size_t tb = 0;
      SMR_HIP(rocprim::radix_sort_keys(nullptr, tb, keys, keys2, (size_t) n_inst, 0, 49, s));
      void *tmp;
      SMR_TRY(c->ensure(SMR_BUF_WIN_CNT, tb + 256, &tmp));
      SMR_HIP(rocprim::radix_sort_keys(tmp, tb, keys, keys2, (size_t) n_inst, 0, 49, s));
   }
   uint32_t *inst_way, *inst_win, *inst_off;
   SMR_TRY(c->ensure(SMR_BUF_WIN_IWAY, (size_t) n_inst * 4, (void **) &inst_way));
   SMR_TRY(c->ensure(SMR_BUF_WIN_IWIN, (size_t) n_inst * 4, (void **) &inst_win));
   SMR_TRY(c->ensure(SMR_BUF_WIN_IOFF, ((size_t) n_inst + 1) * 4, (void **) &inst_off));
   SMR_TRY(c->ensure(SMR_BUF_WIN_SUMS, (size_t) (n_inst / 1024 + 2) * 4 + 64, (void **) &sums));
   k_win_inst<<<(n_inst + 255) / 256, 256, 0, s>>>(keys2, n_inst, b->d_way_off, inst_way, inst_win, inst_off);
   SMR_HIP(hipGetLastError());
   uint32_t n_out = 0;
   {
      /* refs of all instances; more than 2^31 would not fit one draw pass */
      SMR_TRY(smr_scan_u32(inst_off, n_inst, sums, tot, &n_out));
      if (n_out == 0) return SMR_OK;
      if (n_out > 0x7ffffff0u) return smr_fail(SMR_E_CAPACITY, "draw_windows: %u refs in the windows' instances of one rule", n_out);
   }
   double *ox, *oy; int32_t *oref, *onw; int64_t *owo; int2 *oclip; int4 *olim; uint32_t *ofc = nullptr, *obc = nullptr;
   SMR_TRY(c->ensure(SMR_BUF_WIN_X, (size_t) n_out * 8, (void **) &ox));
   SMR_TRY(c->ensure(SMR_BUF_WIN_Y, (size_t) n_out * 8, (void **) &oy));
   SMR_TRY(c->ensure(SMR_BUF_WIN_REF, (size_t) n_out * 4, (void **) &oref));
   SMR_TRY(c->ensure(SMR_BUF_WIN_NODEWIN, (size_t) n_out * 4, (void **) &onw));
   SMR_TRY(c->ensure(SMR_BUF_WIN_WAYOFF, ((size_t) n_inst + 1) * 8, (void **) &owo));
   SMR_TRY(c->ensure(SMR_BUF_WIN_CLIP, (size_t) n_inst * sizeof(int2), (void **) &oclip));
   SMR_TRY(c->ensure(SMR_BUF_WIN_LIM, (size_t) n_inst * sizeof(int4), (void **) &olim));
   if (b->d_way_fill_col) SMR_TRY(c->ensure(SMR_BUF_WIN_FCOL, (size_t) n_inst * 4, (void **) &ofc));
   if (b->d_way_border_col) SMR_TRY(c->ensure(SMR_BUF_WIN_BCOL, (size_t) n_inst * 4, (void **) &obc));
   win_fill_args_t F;
   F.x = b->d_x; F.y = b->d_y; F.way_off = b->d_way_off; F.ref = b->d_ref; F.n_nodes = b->n_nodes;
   F.way_fill_col = b->d_way_fill_col; F.way_border_col = b->d_way_border_col;
   F.inst_way = inst_way; F.inst_win = inst_win; F.inst_off = inst_off; F.n_inst = n_inst; F.n_out = n_out;
   F.win_clip = c->d_win_clip; F.win_lim = c->d_win_lim;
   F.ox = ox; F.oy = oy; F.oref = oref; F.onode_win = onw; F.oway_off = owo; F.oway_clip = oclip; F.oway_lim = olim; F.ofill_col = ofc; F.oborder_col = obc;
   k_win_fill<<<(std::max(n_out, n_inst + 1) + 255) / 256, 256, 0, s>>>(F);
   SMR_HIP(hipGetLastError());

   smr_draw_batch_t ib;
   memset(&ib, 0, sizeof(ib));
   ib.d_x = ox; ib.d_y = oy; ib.n_nodes = n_out; ib.d_way_off = owo; ib.d_ref = oref; ib.n_ways = n_inst; ib.n_refs = n_out;
   ib.d_way_fill_col = ofc; ib.d_way_border_col = obc;
   smr_win_ctx_t wc;
   wc.node_win = onw; wc.win_shift = c->d_win_shift; wc.way_clip = oclip; wc.way_lim = olim;
   return smr_draw_instances(c, st, &ib, &wc);
}

extern "C" int smr_hip_draw_windows(smr_canvas_t *c, const smr_draw_style_t *st, const smr_draw_batch_t *b)
{
   SMR_NEED_INIT();
   if (!c || !st || !b) return smr_fail(SMR_E_INVAL, "draw_windows: NULL argument");
   if (c->win_n <= 0) return smr_fail(SMR_E_INVAL, "draw_windows: the canvas is not an atlas of windows (smr_hip_windows_create)");
   if (b->n_ways <= 0 || b->n_refs <= 0 || (!st->fill_used && !st->border_used)) return SMR_OK;
   if (!b->d_x || !b->d_y || !b->d_way_off) return smr_fail(SMR_E_INVAL, "draw_windows: missing device arrays");
   if (b->n_ways > 0x7ffffff0ll / 2 || b->n_refs > 0x7fffffffll) return smr_fail(SMR_E_INVAL, "draw_windows: batch too large");
   if (!(st->directional && st->fill_used)) return draw_windows_range(c, st, b, 0, c->win_n);
   /* directional=1: the closed ways of a rule are ONE non-zero fill -- one per window, so the windows are drawn one by one */
   for (int k = 0; k < c->win_n; k++) SMR_TRY(draw_windows_range(c, st, b, k, k + 1));
   return SMR_OK;
}
