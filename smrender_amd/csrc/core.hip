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

/* asinh(tan(phi)), the Mercator ordinate of src/rdata.c:223, without the C library's tan and asinh (some 400 fp64 instructions
 * per node, which made the kernel ALU-bound at 3 TB/s): with h = phi / 2,
 *    asinh(tan phi) = ln tan(pi/4 + h) = ln((cos h + sin h) / (cos h - sin h)),
 * sin h and cos h on |h| <= pi/4 need no range reduction (fdlibm's kernel polynomials, < 1 ulp there), and the logarithm's own
 * reduction ln(S / C) = e ln 2 + 2 atanh(t), t = (S - C 2^e) / (S + C 2^e) with |t| <= 0.172, takes the ONE division of the whole
 * evaluation (fdlibm's log: 2 atanh(t) = 2 t + t R(t^2), seven coefficients).  Accurate to a few ulp of the result for
 * |lat| < 89.9 degrees (the difference cos h - sin h cancels towards the poles, where the projection itself is singular and
 * the reference's own libm is no better); beyond that the C library's functions are called as before. */
__device__ __forceinline__ double smr_rcp_div(double a, double b, double rb)
{
   /* a / b, correctly rounded, from rb = RN(1 / b) computed once (Markstein): one multiply and two FMAs instead of the ~30
    * instructions of the IEEE division sequence */
   const double q = a * rb;
   return fma(fma(-q, b, a), rb, q);
}

__device__ __attribute__((noinline)) double smr_mercator_libm(double lat_deg) { return asinh(tan(lat_deg * M_PI / 180.0)); }

__device__ __forceinline__ double smr_mercator(double lat_deg)
{
   if (!(fabs(lat_deg) < 89.9)) return smr_mercator_libm(lat_deg);
   const double h = lat_deg * (M_PI / 360.0), z = h * h;
   /* (explicit FMAs: the file is built with -ffp-contract=off, and Horner steps as separate multiplies and adds are twice the work) */
   double sp = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
   sp = fma(z, sp, 2.75573137070700676789e-06); sp = fma(z, sp, -1.98412698298579493134e-04);
   sp = fma(z, sp, 8.33333333332248946124e-03); sp = fma(z, sp, -1.66666666666666324348e-01);
   const double sh = fma(h * z, sp, h);
   double cp = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
   cp = fma(z, cp, -2.75573143513906633035e-07); cp = fma(z, cp, 2.48015872894767294178e-05);
   cp = fma(z, cp, -1.38888888888741095749e-03); cp = fma(z, cp, 4.16666666666666019037e-02);
   cp *= z;
   const double hz = 0.5 * z, w = 1.0 - hz;
   const double ch = w + fma(z, cp, (1.0 - w) - hz);
   const double S = ch + sh;
   double C = ch - sh;
   int e = __builtin_amdgcn_frexp_exp(S) - __builtin_amdgcn_frexp_exp(C);
   C = ldexp(C, e);                                   /* S / C in (1/2, 2) */
   if (S > C * M_SQRT2) { e++; C *= 2.0; }
   else if (S * M_SQRT2 < C) { e--; C *= 0.5; }
   /* (near the equator e == 0, and S - C = 2 sin h would cancel to 1e-16 absolute: taken from sin h and cos h themselves) */
   const double num = e == 0 ? 2.0 * sh : S - C;
   const double den = e == 0 ? 2.0 * ch : S + C, rd = __builtin_amdgcn_rcp(den);
   double r = fma(fma(-den, rd, 1.0), rd, rd);
   r = fma(fma(-den, r, 1.0), r, r);
   const double t = smr_rcp_div(num, den, r), t2 = t * t;
   double R = fma(t2, 1.479819860511658591e-01, 1.531383769920937332e-01);
   R = fma(t2, R, 1.818357216161805012e-01); R = fma(t2, R, 2.222219843214978396e-01); R = fma(t2, R, 2.857142874366239149e-01);
   R = fma(t2, R, 3.999999999940941908e-01); R = fma(t2, R, 6.666666666666735130e-01);
   R *= t2;
   const double de = (double) e;
   return fma(de, 6.93147180369123816490e-01, fma(t, R, fma(de, 1.90821492927058770002e-10, 2.0 * t)));
}

/* HBM-bound stream: 16 B in, 16 B out (+ optional 9 B) per node, two nodes per lane so that every load/store is a 16-byte
 * access.  x = (lon - lon0) * w / wc and the division by lath_len keep the reference's operations (src/rdata.c:221-225) and their
 * IEEE results, through reciprocals computed once on the host */
struct proj_rcp_t { double r_wc, r_len; };

__global__ __launch_bounds__(256) void k_project(const double *__restrict__ lat, const double *__restrict__ lon, int64_t n,
                                                 smr_proj_t P, proj_rcp_t Q, double *__restrict__ x, double *__restrict__ y,
                                                 int32_t *__restrict__ ix, int32_t *__restrict__ iy, uint8_t *__restrict__ on)
{
   const int64_t npair = n >> 1;
   int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   const int64_t stride = (int64_t) gridDim.x * blockDim.x;
   auto px = [&](double lo) { return smr_rcp_div((lo - P.lon0) * P.w, P.wc, Q.r_wc); };
   auto py = [&](double la) { return P.h * (0.5 - smr_rcp_div(smr_mercator(la) - P.lath, P.lath_len, Q.r_len)); };
   for (; i < npair; i += stride)
   {
      /* streamed once: non-temporal, so that 1 GB of coordinates does not push the canvas and the edge arena out of the L2s */
      typedef double smr_d2 __attribute__((ext_vector_type(2)));
      const smr_d2 la_ = __builtin_nontemporal_load(reinterpret_cast<const smr_d2 *>(lat) + i);
      const smr_d2 lo_ = __builtin_nontemporal_load(reinterpret_cast<const smr_d2 *>(lon) + i);
      double2 la, lo, xo, yo;
      la.x = la_.x; la.y = la_.y; lo.x = lo_.x; lo.y = lo_.y;
      xo.x = px(lo.x); xo.y = px(lo.y);
      yo.x = py(la.x); yo.y = py(la.y);
      smr_d2 xo_, yo_;
      xo_.x = xo.x; xo_.y = xo.y; yo_.x = yo.x; yo_.y = yo.y;
      __builtin_nontemporal_store(xo_, reinterpret_cast<smr_d2 *>(x) + i);
      __builtin_nontemporal_store(yo_, reinterpret_cast<smr_d2 *>(y) + i);
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
      double xo = px(lo);
      double yo = py(la);
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
   proj_rcp_t Q;
   Q.r_wc = 1.0 / smr_g.proj.wc; Q.r_len = 1.0 / smr_g.proj.lath_len;
   k_project<<<stream_grid((n + 1) / 2, 256), 256, 0, smr_g.stream>>>(d_lat, d_lon, n, smr_g.proj, Q, d_x, d_y, d_ix, d_iy, d_onpage);
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
