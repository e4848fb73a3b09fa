/* smr_host.h -- host-side internals shared by the translation units of libsmr_hip. */
#ifndef SMR_HOST_H
#define SMR_HOST_H

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#include "../../include/smr_hip.h"

struct smr_global_t
{
   bool init = false;
   int device = -1;
   hipStream_t stream = nullptr;
   /* the binning of a draw pass (atomics, no arithmetic) runs beside k_cover (arithmetic, no memory traffic to speak of): a second
    * stream, forked from and joined to the first by two events */
   hipStream_t stream2 = nullptr;
   hipEvent_t ev_fork = nullptr, ev_join = nullptr;
   smr_proj_t proj;
   bool have_proj = false;
   bool profiling = false;
   smr_timings_t last;
   uint32_t draw_status = 0;
   char err[512] = "";
};
extern smr_global_t smr_g;

int smr_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

/* caprot.hip: the address, on the CURRENT device, of the static scratch array of smr_hip_caprot_noalloc (once per device) */
int smr_caprot_scratch_setup(int device);

#define SMR_NEED_INIT() do { if (!smr_g.init) return smr_fail(SMR_E_NODEVICE, "libsmr_hip: smr_hip_init() has not succeeded (no GPU, no work: there is no CPU fallback)"); } while (0)
#define SMR_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return smr_fail(SMR_E_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); } while (0)
#define SMR_TRY(call) do { int r_ = (call); if (r_ != SMR_OK) return r_; } while (0)

enum { SMR_BUF_PTS, SMR_BUF_WAYFLAGS, SMR_BUF_SMALL, SMR_BUF_PRIMS, SMR_BUF_EDGES, SMR_BUF_WAYPIX, SMR_BUF_COUNTS,
       SMR_BUF_TILEOFF, SMR_BUF_SUMS, SMR_BUF_ENTRIES, SMR_BUF_CAP0, SMR_BUF_CAP1, SMR_BUF_REFWAY, SMR_BUF_ITEMS, SMR_BUF_MASK, SMR_BUF_HDRS, SMR_BUF_COVERED, SMR_BUF_DASH, SMR_BUF_PEN,
       SMR_BUF_CURVE_CNT, SMR_BUF_FPTS, SMR_BUF_FTIN, SMR_BUF_FTOUT, SMR_BUF_FREFWAY, SMR_BUF_FWAYOFF, SMR_BUF_FSKIP, SMR_BUF_BIGLIST, SMR_BUF_BIGACC, SMR_BUF_CURVEFIX, SMR_BUF_SNAPSHOT, SMR_BUF_OVFLIST, SMR_BUF_BIGBUCKET, SMR_BUF_DELTAS, SMR_BUF_DROWS, SMR_BUF_TILECNT, SMR_BUF_BLIT,
       SMR_BUF_WIN_WB, SMR_BUF_WIN_BOX, SMR_BUF_WIN_CNT, SMR_BUF_WIN_SUMS, SMR_BUF_WIN_TOT, SMR_BUF_WIN_IWAY, SMR_BUF_WIN_IWIN, SMR_BUF_WIN_IOFF, SMR_BUF_WIN_X, SMR_BUF_WIN_Y,
       SMR_BUF_WIN_REF, SMR_BUF_WIN_NODEWIN, SMR_BUF_WIN_WAYOFF, SMR_BUF_WIN_CLIP, SMR_BUF_WIN_LIM, SMR_BUF_WIN_FIRST, SMR_BUF_WIN_FCOL, SMR_BUF_WIN_BCOL, SMR_BUF_WIN_KEYS, SMR_BUF_WIN_KEYS2, SMR_BUF_WIN_TMP, SMR_BUF_PATTERN, SMR_BUF_PURELIST, SMR_BUF_PFLAG, SMR_BUF_BSKIP, SMR_BUF_MAX };

struct smr_devbuf_t { void *p = nullptr; size_t cap = 0; };

struct smr_canvas
{
   int W = 0, H = 0, dpi = 0, band_y0 = 0, band_h = 0;
   uint32_t bg = 0;
   uint32_t *pix = nullptr;
   uint32_t *halo_above = nullptr, *halo_below = nullptr;     /* neighbour rows for the caption scan */
   int n_above = 0, n_below = 0;
   /* an atlas of caption cut-outs (windows.hip): win_n windows of win_size px, one every win_pitch rows; per window the 24.8 shift of
    * its replay (atlas placement included) and its box on the chart in px */
   int win_n = 0, win_size = 0, win_pitch = 0, chart_w = 0, chart_h = 0;
   int2 *d_win_shift = nullptr, *d_win_clip = nullptr; int4 *d_win_lim = nullptr; void *d_win_box = nullptr;
   int win_cap = 0;                         /* windows the atlas has rows and tables for (smr_hip_windows_reset) */
   /* a uniform grid over the chart (cells of grid_cell px from (-grid_cell, -grid_cell) on): per cell the windows whose box meets it */
   uint32_t *d_grid_off = nullptr, *d_grid_win = nullptr; int grid_nx = 0, grid_ny = 0, grid_cell = 0; size_t grid_win_cap = 0;
   std::vector<int> win_sizes;
   std::vector<double> win_ox, win_oy;      /* each window's corner on the chart, px */
   int own_y0 = -1, own_h = 0;         /* smr_hip_canvas_extend: the rows whose captions this canvas scans (-1: the band itself) */
   smr_devbuf_t buf[SMR_BUF_MAX];      /* grow-only work buffers, reused across draw calls */
   int strict_rows = 0;                /* draw_pass's second attempt: rows beyond SMR_ROWCAP spanning edges are sub-sampled (and reported) */
   const uint32_t *pat = nullptr; int pat_w = 0, pat_h = 0;      /* set for the passes of smr_hip_draw_pattern: the primitives' source is this image (device, premultiplied ARGB32) */
   uint64_t last_n_edges = 0;          /* edge slots of the last draw pass (smr_hip_debug_edges) */

   int ensure(int id, size_t bytes, void **out)
   {
      smr_devbuf_t &b = buf[id];
      if (bytes < 256) bytes = 256;
      if (b.cap < bytes)
      {
         if (b.p) { hipStreamSynchronize(smr_g.stream); hipFree(b.p); b.p = nullptr; b.cap = 0; }
         size_t want = bytes + bytes / 8;
         if (hipMalloc(&b.p, want) != hipSuccess) { b.p = nullptr; return smr_fail(SMR_E_NOMEM, "hipMalloc of %zu bytes failed", want); }
         b.cap = want;
      }
      *out = b.p;
      return SMR_OK;
   }
};

/* a batch whose ways are instances for the windows of an atlas (smr_hip_draw_windows): per node its window, per window the 24.8
 * shift of its replay, per way the atlas rows of its window */
struct smr_win_ctx_t { const int32_t *node_win; const int2 *win_shift; const int2 *way_clip; const int4 *way_lim; };      /* way_lim: left, right, top, bottom of what libcairo limits a way's polygons to, 24.8 (windows_setup) */
/* raster.hip: smr_hip_draw with the window context; the in-place exclusive scan of the draw pass on the library's stream
 * (buf[n] = total, which also comes back) */
int smr_draw_instances(smr_canvas *c, const smr_draw_style_t *st, const smr_draw_batch_t *b, const smr_win_ctx_t *wc);
int smr_scan_u32(uint32_t *buf, uint32_t n, uint32_t *sums, uint32_t *total_dev, uint32_t *total);
uint32_t smr_color_pixel(uint32_t smr_col);      /* raster.hip: 0xTTRRGGBB -> premultiplied ARGB32 source pixel */

/* HIP-event stage timer on the library stream (only active when profiling is on) */
struct smr_timer_t
{
   bool on; int n = 0; hipEvent_t ev[12];
   explicit smr_timer_t(bool enable) : on(enable) {}
   ~smr_timer_t() { for (int i = 0; i < n; i++) hipEventDestroy(ev[i]); }
   void mark(hipStream_t s) { if (!on || n >= 12) return; hipEventCreate(&ev[n]); hipEventRecord(ev[n], s); n++; }
   float ms(int a, int b) { float t = 0; if (!on || b >= n) return 0; hipEventSynchronize(ev[b]); hipEventElapsedTime(&t, ev[a], ev[b]); return t; }
};

#endif
