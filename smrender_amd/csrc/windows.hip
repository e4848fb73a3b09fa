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
#include "smr_host.h"
#include "smr_dev.h"

#define SMR_WIN_WAYS_PER_BLOCK 1024

struct win_box_t { double x0, y0, x1, y1; };

/* one thread per way: its box in chart pixels */
__global__ void k_win_bbox(const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ way_off,
                           const int32_t *__restrict__ ref, int64_t n_ways, int64_t n_nodes, double curve_f, win_box_t *__restrict__ wb)
{
   const int64_t w = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (w >= n_ways) return;
   win_box_t b = { 1e300, 1e300, -1e300, -1e300 };
   const int64_t a = way_off[w], e = way_off[w + 1];
   double lmax = 0, px = 0, py = 0;
   bool have = false;
   if (e - a >= 2)
      for (int64_t g = a; g < e; g++)
      {
         int64_t node = ref ? (int64_t) ref[g] : g;
         if (node < 0 || node >= n_nodes) continue;          /* reported by the draw pass itself */
         const double vx = x[node], vy = y[node];
         b.x0 = fmin(b.x0, vx); b.x1 = fmax(b.x1, vx); b.y0 = fmin(b.y0, vy); b.y1 = fmax(b.y1, vy);
         if (have) lmax = fmax(lmax, hypot(vx - px, vy - py));
         px = vx; py = vy; have = true;
      }
   /* curve=1: a Bezier's control points lie curve_factor times the length of their segment beyond its ends (control_points,
    * src/bspline_ctrl.c:64-105: lgt * f along the tangent), and the curve inside the hull of its four knots: the way's box grows
    * by the longest segment's share (a long segment's curve can pass a window that its nodes' box misses) */
   if (curve_f > 0 && b.x0 <= b.x1) { const double m = curve_f * lmax + 0.25; b.x0 -= m; b.y0 -= m; b.x1 += m; b.y1 += m; }
   wb[w] = b;
}

/* grid (blocks over ways, windows): the ways of the block that meet window blockIdx.y; COUNT -> cnt[k * nblk + blk];
 * EMIT -> the pairs at off[k * nblk + blk] + rank, i.e. window-major and ascending ways inside a window */
template <bool EMIT>
__global__ __launch_bounds__(256) void k_win_pairs(const win_box_t *__restrict__ wb, int64_t n_ways, const win_box_t *__restrict__ win,
                                                   const int64_t *__restrict__ way_off, uint32_t *__restrict__ cnt,
                                                   const uint32_t *__restrict__ off, uint32_t *__restrict__ inst_way,
                                                   uint32_t *__restrict__ inst_win, uint32_t *__restrict__ inst_nref, int k0)
{
   __shared__ uint32_t s[256];
   const int k = k0 + (int) blockIdx.y, t = threadIdx.x;
   const win_box_t W = win[k];
   const int64_t w0 = (int64_t) blockIdx.x * SMR_WIN_WAYS_PER_BLOCK + t * 4;
   uint32_t hit = 0, n = 0;
#pragma unroll
   for (int j = 0; j < 4; j++)
   {
      const int64_t w = w0 + j;
      if (w >= n_ways) continue;
      const win_box_t b = wb[w];
      if (b.x0 <= b.x1 && b.x1 >= W.x0 && b.x0 <= W.x1 && b.y1 >= W.y0 && b.y0 <= W.y1) { hit |= 1u << j; n++; }
   }
   s[t] = n;
   __syncthreads();
   for (int d = 1; d < 256; d <<= 1)
   {
      const uint32_t v = t >= d ? s[t - d] : 0;
      __syncthreads();
      s[t] += v;
      __syncthreads();
   }
   const size_t slot = (size_t) blockIdx.y * gridDim.x + blockIdx.x;
   if (!EMIT) { if (t == 255) cnt[slot] = s[255]; return; }
   uint32_t pos = off[slot] + s[t] - n;
#pragma unroll
   for (int j = 0; j < 4; j++)
      if (hit & (1u << j))
      {
         const int64_t w = w0 + j;
         inst_way[pos] = (uint32_t) w; inst_win[pos] = (uint32_t) k; inst_nref[pos] = (uint32_t) (way_off[w + 1] - way_off[w]);
         pos++;
      }
}

struct win_fill_args_t
{
   const double *x, *y; const int64_t *way_off; const int32_t *ref; int64_t n_nodes;
   const uint32_t *way_fill_col, *way_border_col;
   const uint32_t *inst_way, *inst_win, *inst_off;      /* inst_off[n_inst] = all refs */
   uint32_t n_inst, n_out;
   const int2 *win_clip;
   double *ox, *oy; int32_t *oref, *onode_win; int64_t *oway_off; int2 *oway_clip; uint32_t *ofill_col, *oborder_col;
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

extern "C" int smr_hip_windows_create(int chart_w, int chart_h, int dpi, uint32_t bgcolor, const smr_windows_t *w, smr_canvas_t **out)
{
   SMR_NEED_INIT();
   if (!w || !out || w->n <= 0 || !w->size || !w->cx || !w->cy) return smr_fail(SMR_E_INVAL, "windows_create: no windows");
   int maxsize = 0;
   for (int64_t k = 0; k < w->n; k++)
   {
      if (w->size[k] <= 0 || w->size[k] > 65536) return smr_fail(SMR_E_INVAL, "windows_create: window %lld of %d px", (long long) k, w->size[k]);
      maxsize = std::max(maxsize, w->size[k]);
   }
   if (chart_w <= 0 || chart_h <= 0 || dpi <= 0) return smr_fail(SMR_E_INVAL, "windows_create: bad chart %d x %d @ %d dpi", chart_w, chart_h, dpi);
   const int pitch = (maxsize + SMR_TH - 1) & ~(SMR_TH - 1);
   if ((int64_t) w->n * pitch > 65536)
      return smr_fail(SMR_E_CAPACITY, "windows_create: %lld windows of %d px do not fit one atlas (65 536 rows): at most %d per call",
                      (long long) w->n, maxsize, 65536 / pitch);
   smr_canvas_t *c = nullptr;
   SMR_TRY(smr_hip_canvas_create(maxsize, (int) w->n * pitch, dpi, 0, 0, (int) w->n * pitch, &c));
   SMR_HIP(hipMemsetAsync(c->pix, 0, (size_t) maxsize * (size_t) ((int) w->n * pitch) * 4, smr_g.stream));      /* a clear image: transparent */
   c->win_n = (int) w->n; c->win_size = maxsize; c->win_pitch = pitch; c->chart_w = chart_w; c->chart_h = chart_h;
   c->win_sizes.assign(w->size, w->size + w->n);
   std::vector<int2> shift((size_t) w->n);
   std::vector<win_box_t> box((size_t) w->n);
   const double s = (double) dpi / 72;
   /* find_angle :1974-1975, cairo_smr_cut_out :1516-1528 with r = the window's size in pt */
   std::vector<int2> clip((size_t) w->n);
   for (int64_t k = 0; k < w->n; k++)
   {
      const double r_pt = (w->r_px && w->r_px[k] > 0 ? w->r_px[k] : (double) w->size[k]) * 72 / dpi;
      clip[(size_t) k] = make_int2((int) k * pitch, (int) k * pitch + w->size[k]);
      const double x_pt = w->cx[k] * 72 / dpi, y_pt = w->cy[k] * 72 / dpi;
      const double tx = -x_pt + r_pt / 2, ty = -y_pt + r_pt / 2;
      shift[(size_t) k] = make_int2((int32_t) nearbyint(tx * s * 256.0), (int32_t) nearbyint(ty * s * 256.0) + (int32_t) k * pitch * 256);
      /* the window on the chart, in chart px (what lies further than a rule's reach from it cannot touch it) */
      box[(size_t) k] = { -tx * s, -ty * s, -tx * s + maxsize, -ty * s + w->size[k] };
      c->win_ox.push_back(-tx * s); c->win_oy.push_back(-ty * s);
   }
   if (hipMalloc((void **) &c->d_win_shift, (size_t) w->n * sizeof(int2)) != hipSuccess || hipMalloc((void **) &c->d_win_clip, (size_t) w->n * sizeof(int2)) != hipSuccess || hipMalloc((void **) &c->d_win_box, (size_t) w->n * sizeof(win_box_t)) != hipSuccess)
   {
      smr_hip_canvas_destroy(c);
      return smr_fail(SMR_E_NOMEM, "windows_create: window tables");
   }
   SMR_HIP(hipMemcpyAsync(c->d_win_shift, shift.data(), (size_t) w->n * sizeof(int2), hipMemcpyHostToDevice, smr_g.stream));
   SMR_HIP(hipMemcpyAsync(c->d_win_clip, clip.data(), (size_t) w->n * sizeof(int2), hipMemcpyHostToDevice, smr_g.stream));
   SMR_HIP(hipMemcpyAsync(c->d_win_box, box.data(), (size_t) w->n * sizeof(win_box_t), hipMemcpyHostToDevice, smr_g.stream));
   SMR_HIP(hipStreamSynchronize(smr_g.stream));
   *out = c;
   /* the chart's background: cairo_smr_recording_surface_from_bg paints it onto the recording, whose extents bound the paint;
    * in a cut-out that is the chart's rectangle at the window's offset */
   double rx[4] = { 0, (double) chart_w, (double) chart_w, 0 }, ry[4] = { 0, 0, (double) chart_h, (double) chart_h };
   int64_t off[2] = { 0, 5 };
   int32_t ref[5] = { 0, 1, 2, 3, 0 };
   void *dx = nullptr, *dy = nullptr, *doff = nullptr, *dref = nullptr;
   int rc = SMR_OK;
   if (hipMalloc(&dx, 32) != hipSuccess || hipMalloc(&dy, 32) != hipSuccess || hipMalloc(&doff, 16) != hipSuccess || hipMalloc(&dref, 20) != hipSuccess)
      rc = smr_fail(SMR_E_NOMEM, "windows_create: background batch");
   if (rc == SMR_OK)
   {
      hipMemcpy(dx, rx, 32, hipMemcpyHostToDevice); hipMemcpy(dy, ry, 32, hipMemcpyHostToDevice);
      hipMemcpy(doff, off, 16, hipMemcpyHostToDevice); hipMemcpy(dref, ref, 20, hipMemcpyHostToDevice);
      smr_draw_style_t st;
      memset(&st, 0, sizeof(st));
      st.fill_used = 1; st.fill_col = bgcolor; st.direct = 1;
      smr_draw_batch_t b;
      memset(&b, 0, sizeof(b));
      b.d_x = (const double *) dx; b.d_y = (const double *) dy; b.n_nodes = 4; b.d_way_off = (const int64_t *) doff; b.d_ref = (const int32_t *) dref;
      b.n_ways = 1; b.n_refs = 5;
      rc = smr_hip_draw_windows(c, &st, &b);
      hipStreamSynchronize(smr_g.stream);
   }
   if (dx) hipFree(dx); if (dy) hipFree(dy); if (doff) hipFree(doff); if (dref) hipFree(dref);
   if (rc != SMR_OK) { smr_hip_canvas_destroy(c); *out = nullptr; }
   return rc;
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
   const int n_win = k1 - k0;
   /* the windows grown by the rule's reach */
   const double reach = style_reach_px(st, c->dpi);
   const double curve_f = st->curve == 1 ? (st->curve_factor > 0 ? st->curve_factor : 0.2) : 0.0;      /* (curve=2: the knots are nodes themselves) */
   win_box_t *wb, *win;
   SMR_TRY(c->ensure(SMR_BUF_WIN_WB, (size_t) n_ways * sizeof(win_box_t), (void **) &wb));
   SMR_TRY(c->ensure(SMR_BUF_WIN_BOX, (size_t) c->win_n * sizeof(win_box_t), (void **) &win));
   {
      std::vector<win_box_t> h((size_t) c->win_n);
      SMR_HIP(hipMemcpyAsync(h.data(), c->d_win_box, (size_t) c->win_n * sizeof(win_box_t), hipMemcpyDeviceToHost, s));
      SMR_HIP(hipStreamSynchronize(s));
      for (auto &q : h) { q.x0 -= reach; q.y0 -= reach; q.x1 += reach; q.y1 += reach; }
      SMR_HIP(hipMemcpyAsync(win, h.data(), (size_t) c->win_n * sizeof(win_box_t), hipMemcpyHostToDevice, s));
      SMR_HIP(hipStreamSynchronize(s));          /* h goes out of scope */
   }
   k_win_bbox<<<(unsigned) ((n_ways + 255) / 256), 256, 0, s>>>(b->d_x, b->d_y, b->d_way_off, b->d_ref, n_ways, b->n_nodes, curve_f, wb);
   const unsigned nblk = (unsigned) ((n_ways + SMR_WIN_WAYS_PER_BLOCK - 1) / SMR_WIN_WAYS_PER_BLOCK);
   const uint64_t nslots = (uint64_t) nblk * n_win;
   if (nslots > 0x3fffffffull) return smr_fail(SMR_E_CAPACITY, "draw_windows: %lld ways x %d windows", (long long) n_ways, n_win);
   uint32_t *cnt, *sums, *tot;
   SMR_TRY(c->ensure(SMR_BUF_WIN_CNT, (size_t) (nslots + 1) * 4, (void **) &cnt));
   SMR_TRY(c->ensure(SMR_BUF_WIN_SUMS, (size_t) (nslots / 1024 + 2) * 4 + 64, (void **) &sums));
   SMR_TRY(c->ensure(SMR_BUF_WIN_TOT, 64, (void **) &tot));
   k_win_pairs<false><<<dim3(nblk, (unsigned) n_win), 256, 0, s>>>(wb, n_ways, win, b->d_way_off, cnt, nullptr, nullptr, nullptr, nullptr, k0);
   SMR_HIP(hipGetLastError());
   uint32_t n_inst = 0;
   SMR_TRY(smr_scan_u32(cnt, (uint32_t) nslots, sums, tot, &n_inst));
   if (n_inst == 0) return SMR_OK;
   uint32_t *inst_way, *inst_win, *inst_off;
   SMR_TRY(c->ensure(SMR_BUF_WIN_IWAY, (size_t) n_inst * 4, (void **) &inst_way));
   SMR_TRY(c->ensure(SMR_BUF_WIN_IWIN, (size_t) n_inst * 4, (void **) &inst_win));
   SMR_TRY(c->ensure(SMR_BUF_WIN_IOFF, ((size_t) n_inst + 1) * 4, (void **) &inst_off));
   SMR_TRY(c->ensure(SMR_BUF_WIN_SUMS, (size_t) (std::max<uint64_t>(nslots, n_inst) / 1024 + 2) * 4 + 64, (void **) &sums));
   k_win_pairs<true><<<dim3(nblk, (unsigned) n_win), 256, 0, s>>>(wb, n_ways, win, b->d_way_off, nullptr, cnt, inst_way, inst_win, inst_off, k0);
   SMR_HIP(hipGetLastError());
   uint32_t n_out = 0;
   {
      /* refs of all instances; more than 2^31 would not fit one draw pass */
      SMR_TRY(smr_scan_u32(inst_off, n_inst, sums, tot, &n_out));
      if (n_out == 0) return SMR_OK;
      if (n_out > 0x7ffffff0u) return smr_fail(SMR_E_CAPACITY, "draw_windows: %u refs in the windows' instances of one rule", n_out);
   }
   double *ox, *oy; int32_t *oref, *onw; int64_t *owo; int2 *oclip; uint32_t *ofc = nullptr, *obc = nullptr;
   SMR_TRY(c->ensure(SMR_BUF_WIN_X, (size_t) n_out * 8, (void **) &ox));
   SMR_TRY(c->ensure(SMR_BUF_WIN_Y, (size_t) n_out * 8, (void **) &oy));
   SMR_TRY(c->ensure(SMR_BUF_WIN_REF, (size_t) n_out * 4, (void **) &oref));
   SMR_TRY(c->ensure(SMR_BUF_WIN_NODEWIN, (size_t) n_out * 4, (void **) &onw));
   SMR_TRY(c->ensure(SMR_BUF_WIN_WAYOFF, ((size_t) n_inst + 1) * 8, (void **) &owo));
   SMR_TRY(c->ensure(SMR_BUF_WIN_CLIP, (size_t) n_inst * sizeof(int2), (void **) &oclip));
   if (b->d_way_fill_col) SMR_TRY(c->ensure(SMR_BUF_WIN_FCOL, (size_t) n_inst * 4, (void **) &ofc));
   if (b->d_way_border_col) SMR_TRY(c->ensure(SMR_BUF_WIN_BCOL, (size_t) n_inst * 4, (void **) &obc));
   win_fill_args_t F;
   F.x = b->d_x; F.y = b->d_y; F.way_off = b->d_way_off; F.ref = b->d_ref; F.n_nodes = b->n_nodes;
   F.way_fill_col = b->d_way_fill_col; F.way_border_col = b->d_way_border_col;
   F.inst_way = inst_way; F.inst_win = inst_win; F.inst_off = inst_off; F.n_inst = n_inst; F.n_out = n_out;
   F.win_clip = c->d_win_clip;
   F.ox = ox; F.oy = oy; F.oref = oref; F.onode_win = onw; F.oway_off = owo; F.oway_clip = oclip; F.ofill_col = ofc; F.oborder_col = obc;
   k_win_fill<<<(std::max(n_out, n_inst + 1) + 255) / 256, 256, 0, s>>>(F);
   SMR_HIP(hipGetLastError());

   smr_draw_batch_t ib;
   memset(&ib, 0, sizeof(ib));
   ib.d_x = ox; ib.d_y = oy; ib.n_nodes = n_out; ib.d_way_off = owo; ib.d_ref = oref; ib.n_ways = n_inst; ib.n_refs = n_out;
   ib.d_way_fill_col = ofc; ib.d_way_border_col = obc;
   smr_win_ctx_t wc;
   wc.node_win = onw; wc.win_shift = c->d_win_shift; wc.way_clip = oclip;
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
