/* core.hip -- life cycle, projection kernel and memory helpers of libsmr_hip (gfx950).
 *
 * Projection replaces geo2pxf (Mercator branch, src/rdata.c:221-225), geo2pxi's round()
 * (:248-255), pxf2geo (:68-72) and is_on_page (:482-487) with one batched kernel over SoA
 * node arrays; the constants are init_bbox_mll's (src/smrender.c:278-311).
 */
#include "smr_dev.h"
#include "smr_host.h"
#include <string.h>

smr_global_t smr_g;

int smr_fail(int code, const char *fmt, ...)
{
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(smr_g.err, sizeof(smr_g.err), fmt, ap);
   va_end(ap);
   return code;
}

extern "C" const char *smr_hip_last_error(void) { return smr_g.err; }

extern "C" int smr_hip_init(int device)
{
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
      return smr_fail(SMR_E_NODEVICE, "smr_hip_init: no HIP device visible");
   if (device < 0 || device >= n) return smr_fail(SMR_E_INVAL, "smr_hip_init: device %d of %d", device, n);
   SMR_HIP(hipSetDevice(device));
   if (!smr_g.init || smr_g.device != device)
   {
      if (smr_g.stream) { hipStreamDestroy(smr_g.stream); smr_g.stream = nullptr; }
      if (smr_g.stream2) { hipStreamDestroy(smr_g.stream2); smr_g.stream2 = nullptr; }
      if (smr_g.ev_fork) { hipEventDestroy(smr_g.ev_fork); smr_g.ev_fork = nullptr; }
      if (smr_g.ev_join) { hipEventDestroy(smr_g.ev_join); smr_g.ev_join = nullptr; }
      SMR_HIP(hipStreamCreateWithFlags(&smr_g.stream, hipStreamNonBlocking));
      SMR_HIP(hipStreamCreateWithFlags(&smr_g.stream2, hipStreamNonBlocking));
      SMR_HIP(hipEventCreateWithFlags(&smr_g.ev_fork, hipEventDisableTiming));
      SMR_HIP(hipEventCreateWithFlags(&smr_g.ev_join, hipEventDisableTiming));
   }
   SMR_TRY(smr_caprot_scratch_setup(device));      /* with `device` current: every GPU has its own copy of the array */
   smr_g.device = device;
   smr_g.init = true;
   memset(&smr_g.last, 0, sizeof(smr_g.last));
   return SMR_OK;
}

extern "C" int smr_hip_shutdown(void)
{
   if (smr_g.init)
   {
      hipStreamSynchronize(smr_g.stream);
      hipStreamSynchronize(smr_g.stream2);
      hipStreamDestroy(smr_g.stream);
      hipStreamDestroy(smr_g.stream2);
      hipEventDestroy(smr_g.ev_fork); hipEventDestroy(smr_g.ev_join);
      smr_g.stream = smr_g.stream2 = nullptr;
      smr_g.ev_fork = smr_g.ev_join = nullptr;
      smr_g.init = false;
   }
   return SMR_OK;
}

extern "C" int smr_hip_device_name(char *buf, size_t len)
{
   SMR_NEED_INIT();
   hipDeviceProp_t p;
   SMR_HIP(hipGetDeviceProperties(&p, smr_g.device));
   snprintf(buf, len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
   return SMR_OK;
}

extern "C" int smr_hip_sync(void) { SMR_NEED_INIT(); SMR_HIP(hipStreamSynchronize(smr_g.stream)); return SMR_OK; }
extern "C" void *smr_hip_stream(void) { return (void *) smr_g.stream; }
extern "C" int smr_hip_set_profiling(int on) { smr_g.profiling = on != 0; return SMR_OK; }
extern "C" int smr_hip_timings(smr_timings_t *t) { if (!t) return smr_fail(SMR_E_INVAL, "timings: NULL"); *t = smr_g.last; return SMR_OK; }

extern "C" int smr_hip_malloc(void **d, size_t bytes) { SMR_NEED_INIT(); if (hipMalloc(d, bytes ? bytes : 1) != hipSuccess) return smr_fail(SMR_E_NOMEM, "hipMalloc(%zu)", bytes); return SMR_OK; }
extern "C" int smr_hip_free(void *d) { SMR_NEED_INIT(); hipStreamSynchronize(smr_g.stream); SMR_HIP(hipFree(d)); return SMR_OK; }
extern "C" int smr_hip_memcpy_h2d(void *d, const void *h, size_t n)
{ SMR_NEED_INIT(); SMR_HIP(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, smr_g.stream)); SMR_HIP(hipStreamSynchronize(smr_g.stream)); return SMR_OK; }
extern "C" int smr_hip_memcpy_d2h(void *h, const void *d, size_t n)
{ SMR_NEED_INIT(); SMR_HIP(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, smr_g.stream)); SMR_HIP(hipStreamSynchronize(smr_g.stream)); return SMR_OK; }

/* ---- projection constants on the host: init_bbox_mll, src/smrender.c:278-311 ---- */
static void host_pxf2geo(const smr_proj_t *p, double x, double y, double *lon, double *lat)
{
   *lon = x * p->wc / p->w + p->lon0;
   *lat = atan(sinh(p->lath_len * (0.5 - y / p->h) + p->lath)) * 180.0 / M_PI;
}

extern "C" int smr_hip_proj_from_window(smr_proj_t *p, double mean_lat, double mean_lon, double scale,
                                        double w_px, double h_px, int dpi)
{
   if (!p || w_px <= 0 || h_px <= 0 || dpi <= 0 || scale <= 0) return smr_fail(SMR_E_INVAL, "proj_from_window: bad argument");
   double lat, lon;
   double mean_lat_len = scale * (w_px / (double) dpi) * 2.54 / (60.0 * 1852 * 100);   /* smrender.c:697 */
   memset(p, 0, sizeof(*p));
   p->w = w_px; p->h = h_px; p->dpi = dpi;
   p->wc = mean_lat_len / cos(mean_lat * M_PI / 180);
   p->lon0 = mean_lon - p->wc / 2;
   p->lon_ru = mean_lon + p->wc / 2;
   double hc = mean_lat_len * p->h / p->w;
   p->lat_ru = mean_lat + hc / 2.0;
   p->lat_ll = mean_lat - hc / 2.0;
   for (int i = 0; i < 3; i++)
   {
      p->lath = asinh(tan(mean_lat * M_PI / 180.0));
      p->lath_len = asinh(tan(p->lat_ru * M_PI / 180.0)) - asinh(tan(p->lat_ll * M_PI / 180.0));
      host_pxf2geo(p, 0.0, 0.0, &lon, &lat); p->lat_ru = lat;
      host_pxf2geo(p, 0.0, p->h, &lon, &lat); p->lat_ll = lat;
   }
   return SMR_OK;
}

extern "C" int smr_hip_set_projection(const smr_proj_t *p)
{
   if (!p || !(p->w > 0) || !(p->h > 0) || p->wc == 0 || p->lath_len == 0 || p->dpi <= 0)
      return smr_fail(SMR_E_INVAL, "set_projection: degenerate projection");
   smr_g.proj = *p;
   smr_g.have_proj = true;
   return SMR_OK;
}

/* HBM-bound stream: 16 B in, 16 B out (+ optional 9 B) per node, two nodes per lane so
 * that every load/store is a 16-byte access */
__global__ __launch_bounds__(256) void k_project(const double *__restrict__ lat, const double *__restrict__ lon, int64_t n,
                                                 smr_proj_t P, double *__restrict__ x, double *__restrict__ y,
                                                 int32_t *__restrict__ ix, int32_t *__restrict__ iy, uint8_t *__restrict__ on)
{
   const int64_t npair = n >> 1;
   int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   const int64_t stride = (int64_t) gridDim.x * blockDim.x;
   for (; i < npair; i += stride)
   {
      double2 la = reinterpret_cast<const double2 *>(lat)[i], lo = reinterpret_cast<const double2 *>(lon)[i];
      double2 xo, yo;
      xo.x = (lo.x - P.lon0) * P.w / P.wc;
      xo.y = (lo.y - P.lon0) * P.w / P.wc;
      yo.x = P.h * (0.5 - (asinh(tan(la.x * M_PI / 180.0)) - P.lath) / P.lath_len);
      yo.y = P.h * (0.5 - (asinh(tan(la.y * M_PI / 180.0)) - P.lath) / P.lath_len);
      reinterpret_cast<double2 *>(x)[i] = xo;
      reinterpret_cast<double2 *>(y)[i] = yo;
      if (ix) reinterpret_cast<int2 *>(ix)[i] = make_int2((int) round(xo.x), (int) round(xo.y));
      if (iy) reinterpret_cast<int2 *>(iy)[i] = make_int2((int) round(yo.x), (int) round(yo.y));
      if (on)
      {
         uchar2 o;
         o.x = !(lo.x < P.lon0 || lo.x > P.lon_ru || la.x < P.lat_ll || la.x > P.lat_ru);
         o.y = !(lo.y < P.lon0 || lo.y > P.lon_ru || la.y < P.lat_ll || la.y > P.lat_ru);
         reinterpret_cast<uchar2 *>(on)[i] = o;
      }
   }
   if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
   {
      int64_t k = n - 1;
      double la = lat[k], lo = lon[k];
      double xo = (lo - P.lon0) * P.w / P.wc;
      double yo = P.h * (0.5 - (asinh(tan(la * M_PI / 180.0)) - P.lath) / P.lath_len);
      x[k] = xo; y[k] = yo;
      if (ix) ix[k] = (int) round(xo);
      if (iy) iy[k] = (int) round(yo);
      if (on) on[k] = !(lo < P.lon0 || lo > P.lon_ru || la < P.lat_ll || la > P.lat_ru);
   }
}

__global__ __launch_bounds__(256) void k_unproject(const double *__restrict__ x, const double *__restrict__ y, int64_t n,
                                                   smr_proj_t P, double *__restrict__ lon, double *__restrict__ lat)
{
   int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   const int64_t stride = (int64_t) gridDim.x * blockDim.x;
   for (; i < n; i += stride)
   {
      lon[i] = x[i] * P.wc / P.w + P.lon0;
      lat[i] = atan(sinh(P.lath_len * (0.5 - y[i] / P.h) + P.lath)) * 180.0 / M_PI;
   }
}

static unsigned stream_grid(int64_t work, int tb)
{
   int64_t nb = (work + tb - 1) / tb;
   if (nb > 256 * 8) nb = 256 * 8;     /* 8 blocks per CU, grid-stride the rest */
   if (nb < 1) nb = 1;
   return (unsigned) nb;
}

extern "C" int smr_hip_project(const double *d_lat, const double *d_lon, int64_t n,
                               double *d_x, double *d_y, int32_t *d_ix, int32_t *d_iy, uint8_t *d_onpage)
{
   SMR_NEED_INIT();
   if (!smr_g.have_proj) return smr_fail(SMR_E_INVAL, "project: smr_hip_set_projection() not called");
   if (n < 0 || (n > 0 && (!d_lat || !d_lon || !d_x || !d_y))) return smr_fail(SMR_E_INVAL, "project: NULL array");
   if (n == 0) return SMR_OK;
   if ((((uintptr_t) d_lat | (uintptr_t) d_lon | (uintptr_t) d_x | (uintptr_t) d_y) & 15) ||
       ((uintptr_t) d_ix & 7) || ((uintptr_t) d_iy & 7) || ((uintptr_t) d_onpage & 1))
      return smr_fail(SMR_E_INVAL, "project: arrays must be 16-byte aligned (ix/iy 8, onpage 2)");
   smr_timer_t T(smr_g.profiling);
   T.mark(smr_g.stream);
   k_project<<<stream_grid((n + 1) / 2, 256), 256, 0, smr_g.stream>>>(d_lat, d_lon, n, smr_g.proj, d_x, d_y, d_ix, d_iy, d_onpage);
   SMR_HIP(hipGetLastError());
   T.mark(smr_g.stream);
   if (smr_g.profiling) smr_g.last.proj_ms = T.ms(0, 1);
   return SMR_OK;
}

extern "C" int smr_hip_unproject(const double *d_x, const double *d_y, int64_t n, double *d_lon, double *d_lat)
{
   SMR_NEED_INIT();
   if (!smr_g.have_proj) return smr_fail(SMR_E_INVAL, "unproject: smr_hip_set_projection() not called");
   if (n < 0 || (n > 0 && (!d_x || !d_y || !d_lon || !d_lat))) return smr_fail(SMR_E_INVAL, "unproject: NULL array");
   if (n == 0) return SMR_OK;
   k_unproject<<<stream_grid(n, 256), 256, 0, smr_g.stream>>>(d_x, d_y, n, smr_g.proj, d_lon, d_lat);
   SMR_HIP(hipGetLastError());
   return SMR_OK;
}

/* ---- poly_area (src/smfunc.c:377-432): signed area in square nautical miles (> 0: counterclockwise) and centroid of every closed
 * way, the sums taken in the reference's order (one lane walks one way from its first node on).  What act_draw_fini sorts the
 * ways of a directional rule by (:1087-1099) and cap_way places a caption with (:2395-2444).  An open way (is_closed_poly,
 * src/smcoast.c:115-125: at least 4 refs, first == last) gets area 0 and a NaN centroid, where the reference returns -1. ---- */
__global__ void k_poly_area(const double *__restrict__ lat, const double *__restrict__ lon, int64_t n_nodes, const int64_t *__restrict__ way_off,
                            const int32_t *__restrict__ ref, int64_t n_ways, double *__restrict__ area, double *__restrict__ clat, double *__restrict__ clon)
{
   const int64_t w = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (w >= n_ways) return;
   const int64_t a = way_off[w], b = way_off[w + 1];
   auto node = [&](int64_t g) { int64_t v = ref ? (int64_t) ref[g] : g; return v < 0 || v >= n_nodes ? (int64_t) -1 : v; };
   double ar = 0, cla = 0, clo = 0;
   bool ok = b - a >= 4 && node(a) >= 0 && node(a) == node(b - 1);
   if (ok)
   {
      int64_t n1 = node(a);
      for (int64_t g = a; g + 1 < b; g++)
      {
         const int64_t n0 = n1;
         n1 = node(g + 1);
         if (n1 < 0) { ok = false; break; }
         const double x0 = lon[n0] * cos(lat[n0] * M_PI / 180.0), x1 = lon[n1] * cos(lat[n1] * M_PI / 180.0);
         const double f = x0 * lat[n1] - x1 * lat[n0];
         clo += (x0 + x1) * f;
         cla += (lat[n0] + lat[n1]) * f;
         ar += f;
      }
   }
   if (!ok) { area[w] = 0; if (clat) clat[w] = NAN; if (clon) clon[w] = NAN; return; }
   cla /= 3.0 * ar;
   clo /= 3.0 * ar * cos(cla * M_PI / 180.0);
   area[w] = ar * 1800.0;
   if (clat) clat[w] = cla;
   if (clon) clon[w] = clo;
}

extern "C" int smr_hip_poly_area(const double *d_lat, const double *d_lon, int64_t n_nodes, const int64_t *d_way_off, const int32_t *d_ref,
                                 int64_t n_ways, double *d_area, double *d_clat, double *d_clon)
{
   SMR_NEED_INIT();
   if (!d_lat || !d_lon || !d_way_off || !d_area || n_ways < 0 || n_nodes < 0) return smr_fail(SMR_E_INVAL, "poly_area: NULL argument");
   if (n_ways == 0) return SMR_OK;
   k_poly_area<<<(unsigned) ((n_ways + 127) / 128), 128, 0, smr_g.stream>>>(d_lat, d_lon, n_nodes, d_way_off, d_ref, n_ways, d_area, d_clat, d_clon);
   SMR_HIP(hipGetLastError());
   return SMR_OK;
}
