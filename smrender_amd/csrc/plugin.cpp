/* plugin.cpp -- the drop-in boundary: smrender's action API on top of the C-ABI of libsmr_hip.
 *
 * smrender binds these symbols with dlsym() when a rule says
 *    _action_ = draw@libsmr_hip.so:color=...      (src/smrparse.c:286-300, :408-448)
 * and then drives them as  _ini (at rule-parse time, before any OSM data is loaded:
 * src/smrparse.c:464), _main once per matching object in bxtree id order
 * (src/smcore.c:248-312) and _fini (src/smcore.c:324-379).  This file mirrors, for the hot path,
 *    act_draw_ini/_main/_fini   src/smrules_cairo.c:583-694, :1019-1073, :1076-1148
 *    act_cap_ini/_main/_fini    src/smrules_cairo.c:1243-1383, :2447-2484, :2487-2509  (angle=auto)
 *    cap_coord                  src/smrules_cairo.c:2202-2392  (extents, plane, flip, tags)
 * with the same parameter names, return codes and logging.  _main only appends to SoA
 * batches; _fini uploads, projects and launches.  Host functions (get_param, get_object,
 * rdata_get, log_msg, ...) are looked up at run time in the global scope: the smrender
 * executable is linked -rdynamic and libsmrender.so is a global library (configure.ac:98-101).
 *
 * Host Cairo is used for one thing only: text extents of the caption font (and the tiny
 * caption-colour plane), exactly the calls of cap_coord :2250-2252 and cairo_smr_plane
 * :1536-1558; libcairo is dlopen()ed for that.  No pixel of the chart is produced by it.
 */
#include <dlfcn.h>
#include <errno.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>
#include "smr_abi.h"
#include "../../include/smr_hip.h"

/* ---------------------------------------------------------------------------------------- */
/* host API                                                                                 */
/* ---------------------------------------------------------------------------------------- */
struct host_api_t
{
   bool resolved = false;
   char *(*get_param)(const char *, double *, const smr_action_t *) = nullptr;
   int (*get_param_bool)(const char *, const smr_action_t *) = nullptr;
   void *(*get_object)(int, int64_t) = nullptr;
   int (*match_attr)(const smr_osm_obj_t *, const char *, const char *) = nullptr;
   int (*realloc_tags)(smr_osm_obj_t *, int) = nullptr;
   int (*log_msg)(int, const char *, ...) = nullptr;
   struct smr_rdata *(*rdata_get)(void) = nullptr;
   int (*parse_color)(const char *) = nullptr;
   int (*parse_style)(const char *) = nullptr;
   int (*parse_alignment)(const smr_action_t *) = nullptr;
   int (*poly_area)(const smr_osm_way_t *, struct smr_coord *, double *) = nullptr;
   double (*rdata_square_mm)(void) = nullptr;
   double (*rdata_square_nm)(void) = nullptr;
   const smr_osm_way_t *(*page_way)(void) = nullptr;
   int (*parse_length_mm_array)(const char *, double *, int) = nullptr;
   /* objects the rules add to the host's data (cap fontbox=1, angle=auto mkarea=1): libsmrender/osm_inplace.h:132-139, smrender.h:81 */
   smr_osm_node_t *(*malloc_node)(short) = nullptr;
   smr_osm_way_t *(*malloc_way)(short, int) = nullptr;
   void (*osm_node_default)(smr_osm_node_t *) = nullptr;
   void (*osm_way_default)(smr_osm_way_t *) = nullptr;
   int (*put_object)(smr_osm_obj_t *) = nullptr;
   bool can_make_objects() const { return malloc_node && malloc_way && osm_node_default && osm_way_default && put_object; }
};
static host_api_t H;

template <typename F> static void host_sym(F &f, const char *name) { f = (F) dlsym(RTLD_DEFAULT, name); }

static void host_resolve(void)
{
   if (H.resolved) return;
   host_sym(H.get_param, "get_param"); host_sym(H.get_param_bool, "get_param_bool");
   host_sym(H.get_object, "get_object"); host_sym(H.match_attr, "match_attr");
   host_sym(H.realloc_tags, "realloc_tags"); host_sym(H.log_msg, "log_msg");
   host_sym(H.rdata_get, "rdata_get"); host_sym(H.parse_color, "parse_color");
   host_sym(H.parse_style, "parse_style"); host_sym(H.parse_alignment, "parse_alignment");
   host_sym(H.poly_area, "poly_area"); host_sym(H.rdata_square_mm, "rdata_square_mm");
   host_sym(H.rdata_square_nm, "rdata_square_nm"); host_sym(H.page_way, "page_way");
   host_sym(H.parse_length_mm_array, "parse_length_mm_array");
   host_sym(H.malloc_node, "malloc_node"); host_sym(H.malloc_way, "malloc_way"); host_sym(H.osm_node_default, "osm_node_default");
   host_sym(H.osm_way_default, "osm_way_default"); host_sym(H.put_object, "put_object");
   H.resolved = true;
}

static void plog(int level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static void plog(int level, const char *fmt, ...)
{
   char buf[1024];
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(buf, sizeof(buf), fmt, ap);
   va_end(ap);
   if (H.log_msg) H.log_msg(level, "libsmr_hip: %s", buf);
   else fprintf(stderr, "libsmr_hip[%d]: %s\n", level, buf);
}

/* angle=majoraxis is this value in the reference's struct actCaption (src/smrender_dev.h:76, parse_auto_rot src/smrparse.c:1064-1066) */
#define SMR_MAJORAXIS 720.0

/* ---------------------------------------------------------------------------------------- */
/* the chart: device, projection, canvas (one per process)                                  */
/* ---------------------------------------------------------------------------------------- */
struct plugin_t
{
   bool ready = false;
   smr_canvas_t *canvas = nullptr;
   smr_proj_t proj;
   int W = 0, H = 0, dpi = 0, band_y0 = 0, band_h = 0;
   uint32_t bg = 0xffffff;
   /* the guard box (chart px and degrees): a way with a node beyond it is clipped to it before it goes to the device, whose edge
    * arithmetic is exact for edges of less than 139 000 px of height only (far_clip below) */
   double gx0 = 0, gy0 = 0, gx1 = 0, gy1 = 0, glon0 = 0, glon1 = 0, glat0 = 0, glat1 = 0;
};
static plugin_t P;

/* What has been drawn so far, in order: the caption scan cuts its windows out of the chart the reference's way, by rendering the
 * rules again at each caption's sub-pixel offset (smr_hip_draw_windows, windows.hip), so every draw rule's style and device
 * arrays and every composited image stay until the chart is reset.  SMR_HIP_CUTOUT=crop keeps nothing and scans crops of the
 * canvas at the rounded anchors instead (faster; 11-13 % of the %.1f angle tags then differ from the reference's). */
struct kept_rule_t
{
   bool is_blit = false;
   smr_draw_style_t st;
   smr_draw_batch_t b;                                   /* device arrays, owned */
   std::vector<uint32_t> pixels; std::vector<smr_blit_t> items;
};
static std::vector<kept_rule_t> kept_;

static bool cutout_replay(void)
{
   const char *e = getenv("SMR_HIP_CUTOUT");
   return !(e && !strcasecmp(e, "crop"));
}

static void kept_release(void)
{
   for (kept_rule_t &k : kept_)
      if (!k.is_blit)
      {
         void *p[6] = { (void *) k.b.d_x, (void *) k.b.d_y, (void *) k.b.d_way_off, (void *) k.b.d_ref, (void *) k.b.d_way_fill_col, (void *) k.b.d_way_border_col };
         for (void *q : p) if (q) smr_hip_free(q);
      }
   kept_.clear();
}

static void kept_blit(const std::vector<uint32_t> &pixels, const std::vector<smr_blit_t> &items)
{
   if (!cutout_replay() || items.empty()) return;
   kept_.emplace_back();
   kept_.back().is_blit = true; kept_.back().pixels = pixels; kept_.back().items = items;
}

extern "C" smr_canvas_t *smr_hip_plugin_canvas(void) { return P.canvas; }

static void pool_release(void);

/* where the plugin's draw rules spend their time (diagnostics: tools/plugin_bench.py): seconds in _main (the host's get_object per
 * ref), in _fini before the device is asked for anything, in the uploads + projection, in smr_hip_draw, in the wait for it */
static double pstat_[8];
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec; }
extern "C" int smr_hip_plugin_stats(double *out, int n, int reset)
{
   for (int i = 0; i < n && i < 8; i++) out[i] = pstat_[i];
   if (reset) memset(pstat_, 0, sizeof(pstat_));
   return SMR_OK;
}

extern "C" int smr_hip_plugin_reset(void)
{
   if (P.canvas) smr_hip_canvas_destroy(P.canvas);
   kept_release();
   pool_release();
   P = plugin_t();
   return SMR_OK;
}

static int env_int(const char *name, int def)
{
   const char *s = getenv(name);
   return s && *s ? atoi(s) : def;
}

/* called from every _ini.  rdata is complete at that time (init_rendering_window runs before
 * the rules are parsed: src/smrender.c:1134 vs :1190). */
/* geo2pxf's Mercator branch and pxf2geo (src/rdata.c:221-225, :68-72) with the host's libm, for the few ways far_clip touches */
static void host_geo2pxf(double lon, double lat, double *x, double *y)
{
   *x = (lon - P.proj.lon0) * P.proj.w / P.proj.wc;
   *y = P.proj.h * (0.5 - (asinh(tan(lat * M_PI / 180.0)) - P.proj.lath) / P.proj.lath_len);
}
static void host_pxf2geo(double x, double y, double *lon, double *lat)
{
   *lon = x * P.proj.wc / P.proj.w + P.proj.lon0;
   *lat = atan(sinh(P.proj.lath_len * (0.5 - y / P.proj.h) + P.proj.lath)) * 180.0 / M_PI;
}

static int plugin_setup(void)
{
   host_resolve();
   if (P.ready) return 0;
   if (!H.get_param || !H.get_param_bool || !H.get_object || !H.rdata_get)
   {
      plog(SMR_LOG_ERR, "host symbols get_param/get_param_bool/get_object/rdata_get not found: not running inside smrender?");
      return -1;
   }
   const struct smr_rdata *rd = H.rdata_get();
   if (rd->proj != 0 || rd->polygon_window)
   {
      plog(SMR_LOG_ERR, "only the Mercator projection without polygon window is implemented on the GPU path");
      return -1;
   }
   if (smr_hip_init(env_int("SMR_HIP_DEVICE", 0)) != SMR_OK)
   {
      plog(SMR_LOG_ERR, "%s", smr_hip_last_error());
      return -1;
   }
   memset(&P.proj, 0, sizeof(P.proj));
   P.proj.lon0 = rd->bb.ll.lon; P.proj.w = rd->w; P.proj.h = rd->h; P.proj.wc = rd->wc;
   P.proj.lath = rd->lath; P.proj.lath_len = rd->lath_len;
   P.proj.lat_ll = rd->bb.ll.lat; P.proj.lon_ru = rd->bb.ru.lon; P.proj.lat_ru = rd->bb.ru.lat;
   P.proj.dpi = rd->dpi;
   if (smr_hip_set_projection(&P.proj) != SMR_OK)
   {
      plog(SMR_LOG_ERR, "%s", smr_hip_last_error());
      return -1;
   }
   /* the pixel grid of cairo_smr_image_surface_from_bg, src/smrules_cairo.c:344 */
   P.W = (int) round(rd->w); P.H = (int) round(rd->h); P.dpi = rd->dpi;
   {
      const double G = 32768.0 - 8.0;   /* a little inside SMR_FAR_PX (smr_dev.h: what the device accepts; a cut comes back from pxf2geo and geo2pxf
                                           to within 1e-9 px of the box's side): nothing that far is ever seen, and the device's edge arithmetic stays exact */
      P.gx0 = -G; P.gy0 = -G; P.gx1 = rd->w + G; P.gy1 = rd->h + G;
      double la, lo;
      host_pxf2geo(P.gx0, P.gy0, &lo, &la); P.glon0 = lo; P.glat1 = la;
      host_pxf2geo(P.gx1, P.gy1, &lo, &la); P.glon1 = lo; P.glat0 = la;
   }
   P.band_y0 = env_int("SMR_HIP_BAND_Y0", 0);
   P.band_h = env_int("SMR_HIP_BAND_H", P.H - P.band_y0);
   uint32_t bg = 0xffffff;
   if (H.parse_color) bg = (uint32_t) H.parse_color("bgcolor");     /* src/smrules_cairo.c:324 */
   P.bg = bg;
   if (smr_hip_canvas_create(P.W, P.H, P.dpi, bg, P.band_y0, P.band_h, &P.canvas) != SMR_OK)
   {
      plog(SMR_LOG_ERR, "%s", smr_hip_last_error());
      return -1;
   }
   char name[256] = "";
   smr_hip_device_name(name, sizeof(name));
   plog(SMR_LOG_NOTICE, "device canvas %dx%d px @ %d dpi, rows %d..%d on %s", P.W, P.H, P.dpi, P.band_y0,
        P.band_y0 + P.band_h - 1, name);
   P.ready = true;
   return 0;
}

/* device buffer with the life time of a scope */
struct dbuf_t
{
   void *p = nullptr;
   ~dbuf_t() { if (p) smr_hip_free(p); }
   int put(const void *h, size_t bytes)
   {
      if (smr_hip_malloc(&p, bytes ? bytes : 16) != SMR_OK) return -1;
      if (bytes && smr_hip_memcpy_h2d(p, h, bytes) != SMR_OK) return -1;
      return 0;
   }
   int alloc(size_t bytes) { return smr_hip_malloc(&p, bytes ? bytes : 16) == SMR_OK ? 0 : -1; }
};

/* the draw rules' device arrays (lat, lon, x, y, way table, refs, per-way colours): grow-only and kept from rule to rule -- a
 * chart has hundreds of draw rules, and six hipMalloc/hipFree pairs per rule cost more than the drawing of a small one */
struct pool_t { void *p = nullptr; size_t cap = 0; };
static pool_t pool_[8];
static void *pool_get(int i, size_t bytes)
{
   if (bytes < 16) bytes = 16;
   if (pool_[i].cap < bytes)
   {
      if (pool_[i].p) smr_hip_free(pool_[i].p);
      pool_[i].p = nullptr; pool_[i].cap = 0;
      const size_t want = bytes + bytes / 4 + 256;
      if (smr_hip_malloc(&pool_[i].p, want) != SMR_OK) return nullptr;
      pool_[i].cap = want;
   }
   return pool_[i].p;
}
static void *pool_put(int i, const void *h, size_t bytes)
{
   void *p = pool_get(i, bytes);
   if (p && bytes && smr_hip_memcpy_h2d(p, h, bytes) != SMR_OK) return nullptr;
   return p;
}
static void pool_release(void)
{
   for (pool_t &b : pool_) { if (b.p) smr_hip_free(b.p); b.p = nullptr; b.cap = 0; }
}

/* struct col_spec, src/smrender_dev.h:176-180 + parse_col_spec src/smrparse.c:261-267 */
struct col_spec_t { uint32_t col = 0; const char *key = nullptr; };

static void parse_col_spec(const char *s, col_spec_t *cs)
{
   if (*s == '%') cs->key = s + 1;
   else cs->col = H.parse_color ? (uint32_t) H.parse_color(s) : 0;
}

/* color_by_cs, src/smrules_cairo.c:2042-2076 */
static uint32_t color_by_cs(const smr_osm_obj_t *o, const col_spec_t *cs)
{
   char buf[32];
   int n;
   if (cs->key == nullptr || o == nullptr || !H.match_attr) return cs->col;
   if ((n = H.match_attr(o, cs->key, nullptr)) == -1)
   {
      plog(SMR_LOG_WARN, "object %d/%lld has no color key '%s'", o->type, (long long) o->id, cs->key);
      return 0;
   }
   if (o->otag[n].v.len > (int) sizeof(buf) - 1) return 0;
   memcpy(buf, o->otag[n].v.buf, o->otag[n].v.len);
   buf[o->otag[n].v.len] = '\0';
   return H.parse_color ? (uint32_t) H.parse_color(buf) : 0;
}

/* ---------------------------------------------------------------------------------------- */
/* draw                                                                                     */
/* ---------------------------------------------------------------------------------------- */
struct draw_rule_t
{
   smr_draw_style_t st;
   col_spec_t fill_cs, border_cs;
   std::vector<double> lat, lon;          /* one entry per ref (the closing ref of a closed way shares the first) */
   std::vector<int64_t> way_off;
   std::vector<int32_t> ref;
   std::vector<uint32_t> fill_col, border_col;
   /* directional=1: the closed way with the largest |area| decides about the page rectangle (:1091-1109) */
   double max_area = -1; bool max_cw = false; uint32_t max_col = 0;
   double min_area = 1e300; uint32_t min_col = 0;      /* the smallest closed way: the LAST after act_draw_fini's qsort by area (:1099) */
   double wavy_deg = 0;                   /* wavy=1: wavy_length in degrees (:632-642) */
   /* file=: the image the rule's closed ways are filled with below its colours (img_ini :2512-2712 as act_draw_ini :662-671 calls
    * it; render_poly_line :989-997), as img_ini leaves it: scaled, transparency applied, premultiplied ARGB32 */
   std::vector<uint32_t> pat; int pat_w = 0, pat_h = 0;
};

/* the image of a rule's file= as a pattern's pixels (defined with img below): 0, > 0 to leave the rule to the built-in action, < 0 */
static int rule_pattern(smr_smrule_t *r, const char *who, std::vector<uint32_t> &pix, int *w, int *h);

/* parse_dash_style, src/smrparse.c:1077-1102: "a:b:c:d" in mm (the host's parser knows the other units); no or
 * no valid array = 0 lengths, for which the library takes the style's default */
static int parse_dashes(const char *s, double *val)
{
   if (s == nullptr) return 0;
   int n;
   if (H.parse_length_mm_array) n = H.parse_length_mm_array(s, val, 4);
   else
   {
      for (n = 0; s != nullptr && n < 4; n++)
      {
         char *e;
         val[n] = strtod(s, &e);
         if (e == s) return 0;
         while (*e == ' ') e++;
         if (*e != '\0' && *e != ':' && strncmp(e, "mm", 2) != 0)
         {
            plog(SMR_LOG_ERR, "draw: unit in dash array '%s' needs the host's parse_length_mm_array()", s);
            return 0;
         }
         if ((s = strchr(s, ':')) != nullptr) s++;
      }
   }
   return n > 0 ? n : 0;
}

static int draw_ini(smr_smrule_t *r)
{
   int e;
   host_resolve();
   if (r->oo->type != SMR_OSM_WAY && r->oo->type != SMR_OSM_REL)
   {
      plog(SMR_LOG_WARN, "'draw' may be applied to ways or relations only");
      return 1;
   }
   if ((e = plugin_setup()) != 0) return e;

   draw_rule_t *d = new draw_rule_t();
   memset(&d->st, 0, sizeof(d->st));
   char *s;
   if ((s = H.get_param("color", nullptr, r->act)) != nullptr) { parse_col_spec(s, &d->fill_cs); d->st.fill_used = 1; }
   if (H.get_param("width", &d->st.fill_width_mm, r->act) == nullptr) d->st.fill_width_mm = 0;
   d->st.fill_style = H.parse_style ? H.parse_style(H.get_param("style", nullptr, r->act)) : 0;
   if ((s = H.get_param("bcolor", nullptr, r->act)) != nullptr) { parse_col_spec(s, &d->border_cs); d->st.border_used = 1; }
   if (H.get_param("bwidth", &d->st.border_width_mm, r->act) == nullptr) d->st.border_width_mm = 0;
   d->st.border_style = H.parse_style ? H.parse_style(H.get_param("bstyle", nullptr, r->act)) : 0;
   d->st.fill_col = d->fill_cs.col; d->st.border_col = d->border_cs.col;
   /* curve / wavy, :625-642: wavy wins when both are given; curve_factor defaults to DIV_PART (src/smrender_dev.h:114) */
   d->st.curve = 0;
   if (H.get_param_bool("curve", r->act))
   {
      d->st.curve = 1;
      if (H.get_param("curve_factor", &d->st.curve_factor, r->act) == nullptr) d->st.curve_factor = 0.2;
   }
   if (H.get_param_bool("wavy", r->act))
   {
      /* :632-642: wavy_length is a length on paper (default unit mm) turned into degrees, rdata_unit(&v, U_DEG) =
       * px * mean_lat_len / w (src/rdata.c:296-327); without the parameter WAVY_LENGTH 0.0015 degrees (:113) */
      d->st.curve = 2;
      d->wavy_deg = 0.0015;
      if ((s = H.get_param("wavy_length", nullptr, r->act)) != nullptr)
      {
         double mm = 0;
         if (H.parse_length_mm_array ? H.parse_length_mm_array(s, &mm, 1) != 1 : (mm = strtod(s, nullptr), false))
         {
            plog(SMR_LOG_ERR, "draw: cannot parse wavy_length=%s", s);
            delete d;
            return 1;
         }
         const struct smr_rdata *rd = H.rdata_get();
         d->wavy_deg = mm / 25.4 * rd->dpi * rd->mean_lat_len / rd->w;
      }
      if (!(d->wavy_deg > 0)) { plog(SMR_LOG_ERR, "draw: wavy_length must be positive"); delete d; return 1; }
   }
   d->st.directional = H.get_param_bool("directional", r->act);
   d->st.ignore_open = H.get_param_bool("ignore_open", r->act);
   /* not a parameter of the reference's draw (it fills with libcairo's default rule, :1012, :1134-1138): an extension of this back end */
   d->st.evenodd = H.get_param_bool("evenodd", r->act);
   /* DRAW_TRANSPARENT parses but draws solid (cairo_smr_dash has no case for it) */
   if (d->st.fill_style == 3) d->st.fill_style = 0;
   if (d->st.border_style == 3) d->st.border_style = 0;

   d->st.fill_ndash = parse_dashes(H.get_param("dash", nullptr, r->act), d->st.fill_dash);
   d->st.border_ndash = parse_dashes(H.get_param("bdash", nullptr, r->act), d->st.border_dash);

   /* dash styles along curves are drawn too (k_dash walks the decomposed curve as libcairo's dashing stroker does; a solid stroke whose
    * caps the rule's other stroke has made round keeps the polygon stroker's tangent pieces): nothing of draw's styles is refused here */
   /* file= (:662-671): the closed ways are filled with the image, too */
   if (H.get_param("file", nullptr, r->act) != nullptr)
   {
      const int e2 = rule_pattern(r, "draw", d->pat, &d->pat_w, &d->pat_h);
      if (e2 > 0) { delete d; return e2; }
      if (e2 < 0) { plog(SMR_LOG_WARN, "not using image in fill operation"); d->pat.clear(); }      /* as act_draw_ini goes on without it */
   }
   /* check if parameter combination makes sense (:673-686) */
   if (d->st.directional)
   {
      if ((d->st.fill_used || !d->pat.empty()) && d->st.border_used)
      {
         plog(SMR_LOG_WARN, "in directional mode, filling and border drawing is not supported at the same time -> disabling border");
         d->st.border_used = 0;
      }
      if (!d->st.fill_used && d->pat.empty() && d->st.border_used)
      {
         plog(SMR_LOG_NOTICE, "directional mode makes only sense with a fill operation -> disabling directional mode");
         d->st.directional = 0;
      }
   }
   d->way_off.push_back(0);
   r->data = d;
   return 0;
}

static void far_clip(struct draw_rule_t *d, size_t node0, size_t ref0, bool closed, const smr_osm_obj_t *obj);      /* below, before draw_main */

/* coord_diffp, src/smath.c:34-52: bearing (degrees, 0 = north, clockwise, in [0, 360)) and orthodrome distance (degrees) */
static void wavy_diff(double slat, double slon, double tlat, double tlon, double *bearing, double *dist)
{
   /* DEG2RAD(x) is ((x) * M_PI / 180.0), RAD2DEG(x) ((x) * 180.0 / M_PI) (libsmrender/smrender.h:48-49): written out, their rounding is part of the path */
   const double dlat = tlat - slat;
   const double dlon = (tlon - slon) * cos(((slat + tlat) / 2.0) * M_PI / 180.0);
   double b = atan2(dlon, dlat) * 180.0 / M_PI;
   *dist = acos(sin(slat * M_PI / 180.0) * sin(tlat * M_PI / 180.0) +
                cos(slat * M_PI / 180.0) * cos(tlat * M_PI / 180.0) * cos((tlon - slon) * M_PI / 180.0)) * 180.0 / M_PI;
   b = fmod(b, 360.0);
   *bearing = b < 0 ? b + 360.0 : b;
}

/* wavy=1 (cairo_smr_wavy, src/smrules_cairo.c:777-840, and wavy, :752-774): the way is resampled every wavy_length degrees,
 * and every step becomes one curve_to with its control points at -45 and +45 degrees of the step's bearing.  The resampling
 * is a chain in which every step starts where the last one ended and compares rounded distances: it runs here, on the host,
 * with the C library's sin/cos/acos/atan2, which are the ones the reference calls (a device libm may differ in the last bit,
 * and a flipped comparison is a wave more or less).  What goes to the GPU are the knots p0, (c1, c2, p) * n as nodes of a way
 * of their own (smr_draw_style_t.curve = 2); projection, decomposition and stroking happen there. */
static int draw_way_wavy(draw_rule_t *d, const smr_osm_way_t *w)
{
   /* is_closed_poly, src/smcoast.c:115-125.  A closed way is filled (:1009-1013): cairo_fill closes the wavy path, which ends at the
    * last whole wave before the way's end, with a straight line; the library learns it from one more ref equal to the first */
   const bool closed = w->ref_cnt >= 4 && w->ref[0] == w->ref[w->ref_cnt - 1];
   if (!closed && d->st.ignore_open) return 0;
   if (w->ref_cnt < 2)
   {
      plog(SMR_LOG_WARN, "ignoring short way %lld (len = %d)", (long long) w->obj.id, w->ref_cnt);
      return 0;
   }
   const size_t first = d->lat.size();
   auto put = [&](double lat, double lon) { d->ref.push_back((int32_t) d->lat.size()); d->lat.push_back(lat); d->lon.push_back(lon); };
   auto give_up = [&](int pos)
   {
      plog(SMR_LOG_ERR, "node %lld of way %lld das not exit", (long long) w->ref[pos], (long long) w->obj.id);      /* :793, :811 */
      d->lat.resize(first); d->lon.resize(first); d->ref.resize((size_t) d->way_off.back());
      return 0;
   };
   const smr_osm_node_t *n = (const smr_osm_node_t *) H.get_object(SMR_OSM_NODE, w->ref[0]);
   if (n == nullptr) return give_up(0);
   double sc_lat = n->lat, sc_lon = n->lon;
   put(sc_lat, sc_lon);
   /* directional=1: poly_area's shoelace sum over the WAY's nodes (src/smfunc.c:377-432), as in draw_way */
   double ar = 0, plat = n->lat, plon = n->lon;
   const double dist = d->wavy_deg;
   double bearing = 0, pdist = 0;
   for (int i = 1;;)
   {
      if (pdist <= 0)
      {
         if (i >= w->ref_cnt) break;
         if ((n = (const smr_osm_node_t *) H.get_object(SMR_OSM_NODE, w->ref[i])) == nullptr) return give_up(i);
         ar += plon * cos(plat * M_PI / 180.0) * n->lat - n->lon * cos(n->lat * M_PI / 180.0) * plat;
         plat = n->lat; plon = n->lon;
         i++;
         const double rest = pdist;
         wavy_diff(sc_lat, sc_lon, n->lat, n->lon, &bearing, &pdist);
         pdist += rest;
      }
      if (pdist > dist)
      {
         const double ic_lat = sc_lat + dist * cos(bearing * M_PI / 180.0);
         const double ic_lon = sc_lon + dist * sin(bearing * M_PI / 180.0) / cos(((ic_lat + sc_lat) / 2) * M_PI / 180.0);
         double wb, wd;
         wavy_diff(sc_lat, sc_lon, ic_lat, ic_lon, &wb, &wd);
         for (int side = 0; side < 2; side++)
         {
            wb += side == 0 ? -45.0 : 90.0;
            const double la = sc_lat + wd * M_SQRT1_2 * cos(wb * M_PI / 180.0);
            const double lo = sc_lon + wd * M_SQRT1_2 * sin(wb * M_PI / 180.0) / cos(((la + sc_lat) / 2) * M_PI / 180.0);
            put(la, lo);
         }
         put(ic_lat, ic_lon);
         sc_lat = ic_lat; sc_lon = ic_lon;
      }
      pdist -= dist;
   }
   if (d->lat.size() > 0x7fffff00u) { plog(SMR_LOG_ERR, "draw: more than 2^31 nodes in one rule"); return -1; }
   if (d->lat.size() - first < 4)
   {
      /* a way shorter than one wave: cairo_move_to alone, nothing to stroke */
      d->lat.resize(first); d->lon.resize(first); d->ref.resize((size_t) d->way_off.back());
      return 0;
   }
   if (closed) d->ref.push_back((int32_t) first);
   d->way_off.push_back((int64_t) d->ref.size());
   if (d->fill_cs.key) d->fill_col.push_back(color_by_cs(&w->obj, &d->fill_cs));
   if (d->border_cs.key) d->border_col.push_back(color_by_cs(&w->obj, &d->border_cs));
   if (closed && d->st.directional && fabs(ar) > d->max_area)
   {
      d->max_area = fabs(ar); d->max_cw = ar < 0;
      d->max_col = d->fill_cs.key ? d->fill_col.back() : d->fill_cs.col;
   }
   if (closed && d->st.directional && fabs(ar) <= d->min_area)
   {
      d->min_area = fabs(ar);
      d->min_col = d->fill_cs.key ? d->fill_col.back() : d->fill_cs.col;
   }
   return 0;
}

static int draw_way(draw_rule_t *d, const smr_osm_way_t *w)
{
   if (d->st.curve == 2) return draw_way_wavy(d, w);
   /* is_closed_poly, src/smcoast.c:115-125 */
   const bool closed = w->ref_cnt >= 4 && w->ref[0] == w->ref[w->ref_cnt - 1];
   if (!closed && d->st.ignore_open) return 0;
   if (w->ref_cnt < 2)
   {
      plog(SMR_LOG_WARN, "ignoring short way %lld (len = %d)", (long long) w->obj.id, w->ref_cnt);
      return 0;
   }
   const int32_t first = (int32_t) d->lat.size();
   int emitted = 0;
   double ar = 0, plat = 0, plon = 0;       /* poly_area's shoelace sum in lon * cos(lat), src/smfunc.c:377-432 */
   for (int i = 0; i < w->ref_cnt; i++)
   {
      if (closed && i == w->ref_cnt - 1 && emitted)
      {
         ar += plon * cos(plat * M_PI / 180.0) * d->lat[first] - d->lon[first] * cos(d->lat[first] * M_PI / 180.0) * plat;
         d->ref.push_back(first); emitted++;
         break;
      }
      const smr_osm_node_t *n = (const smr_osm_node_t *) H.get_object(SMR_OSM_NODE, w->ref[i]);
      if (n == nullptr && d->st.curve == 1)
      {
         /* cairo_smr_poly_curve gives the way up (:718-723): nothing of it is drawn */
         plog(SMR_LOG_EMERG, "node %lld of way %lld at pos %d does not exist", (long long) w->ref[i], (long long) w->obj.id, i);
         d->lat.resize(first); d->lon.resize(first); d->ref.resize((size_t) d->way_off.back());
         return 0;
      }
      if (n == nullptr)      /* cairo_smr_poly_line skips missing nodes, src/smrules_cairo.c:853-857 */
      {
         plog(SMR_LOG_WARN, "node %lld of way %lld at pos %d does not exist", (long long) w->ref[i], (long long) w->obj.id, i);
         continue;
      }
      if (emitted) ar += plon * cos(plat * M_PI / 180.0) * n->lat - n->lon * cos(n->lat * M_PI / 180.0) * plat;
      plat = n->lat; plon = n->lon;
      d->ref.push_back((int32_t) d->lat.size());
      d->lat.push_back(n->lat); d->lon.push_back(n->lon);
      emitted++;
   }
   if (d->lat.size() > 0x7fffff00u) { plog(SMR_LOG_ERR, "draw: more than 2^31 nodes in one rule"); return -1; }
   bool far = false;
   if (d->st.curve == 0)
      for (size_t k = (size_t) first; k < d->lat.size() && !far; k++)
         far = !(d->lon[k] >= P.glon0 && d->lon[k] <= P.glon1 && d->lat[k] >= P.glat0 && d->lat[k] <= P.glat1);
   if (far)
   {
      plog(SMR_LOG_DEBUG, "way %lld reaches more than 32768 px beyond the chart: clipped to that box", (long long) w->obj.id);
      far_clip(d, (size_t) first, (size_t) d->way_off.back(), closed && emitted >= 4, &w->obj);
   }
   else
   {
   d->way_off.push_back((int64_t) d->ref.size());
   if (d->fill_cs.key) d->fill_col.push_back(color_by_cs(&w->obj, &d->fill_cs));
   if (d->border_cs.key) d->border_col.push_back(color_by_cs(&w->obj, &d->border_cs));
   }
   if (closed && d->st.directional && fabs(ar) > d->max_area)
   {
      d->max_area = fabs(ar); d->max_cw = ar < 0;
      d->max_col = d->fill_cs.key ? d->fill_col.back() : d->fill_cs.col;
   }
   if (closed && d->st.directional && fabs(ar) <= d->min_area)
   {
      d->min_area = fabs(ar);
      d->min_col = d->fill_cs.key ? d->fill_col.back() : d->fill_cs.col;
   }
   return 0;
}

/* A way with a node 32 768 px or more beyond the chart (a coastline, a border or a ferry line that runs on for hundreds of
 * kilometres past a large-scale chart) cannot go to the device as it is: its edge arithmetic is exact for edges of less than
 * 139 000 px of height, and smr_hip_draw refuses the whole rule otherwise.  The reference draws such a way (libcairo's 24.8
 * holds +-35 M px at 300 dpi -- though its scan converter's own 64-bit products overflow for an edge of more than ~130 000 px in
 * both directions, and what it then paints is not the line), so the plugin clips it here, in chart px with the host's libm: a closed way as a polygon against the guard box
 * (Sutherland-Hodgman: what lies on the box's sides is 32 768 px from the chart), an open way into the runs of
 * its segments that lie inside the box.  Vertices that stay keep their coordinates; the cuts are new nodes (pxf2geo of the cut,
 * which geo2pxf takes back to within 1e-9 px).  What is seen on the chart is the same line: a segment that is cut keeps its
 * direction to within the 24.8 rounding of its new end point.  Straight ways only: a curve's control points depend on the
 * neighbouring nodes (a far node of a curve=1 way still ends in smr_hip_draw's refusal of the rule). */
struct far_pt_t { double x, y, lat, lon; bool orig; };

static void far_clip(draw_rule_t *d, size_t node0, size_t ref0, bool closed, const smr_osm_obj_t *obj)
{
   std::vector<far_pt_t> v;
   const size_t nref = d->ref.size() - ref0;
   for (size_t i = 0; i < nref - (closed ? 1 : 0); i++)
   {
      const int32_t k = d->ref[ref0 + i];
      far_pt_t q; q.lat = d->lat[(size_t) k]; q.lon = d->lon[(size_t) k]; q.orig = true;
      host_geo2pxf(q.lon, q.lat, &q.x, &q.y);
      v.push_back(q);
   }
   d->lat.resize(node0); d->lon.resize(node0); d->ref.resize(ref0);
   auto emit = [&](const std::vector<far_pt_t> &u, bool as_closed)
   {
      const int32_t first = (int32_t) d->lat.size();
      for (const far_pt_t &q : u)
      {
         double la = q.lat, lo = q.lon;
         if (!q.orig) host_pxf2geo(q.x, q.y, &lo, &la);
         d->ref.push_back((int32_t) d->lat.size());
         d->lat.push_back(la); d->lon.push_back(lo);
      }
      if (as_closed) d->ref.push_back(first);
      d->way_off.push_back((int64_t) d->ref.size());
      if (d->fill_cs.key) d->fill_col.push_back(color_by_cs(obj, &d->fill_cs));
      if (d->border_cs.key) d->border_col.push_back(color_by_cs(obj, &d->border_cs));
   };
   auto cut = [](const far_pt_t &a, const far_pt_t &b, double t) { far_pt_t q; q.x = a.x + t * (b.x - a.x); q.y = a.y + t * (b.y - a.y); q.lat = q.lon = 0; q.orig = false; return q; };
   if (closed)
   {
      /* one side of the box after the other: inside(p), and where the segment meets the side */
      for (int side = 0; side < 4 && !v.empty(); side++)
      {
         auto val = [&](const far_pt_t &q) { return side == 0 ? q.x - P.gx0 : side == 1 ? P.gx1 - q.x : side == 2 ? q.y - P.gy0 : P.gy1 - q.y; };
         std::vector<far_pt_t> u;
         for (size_t i = 0; i < v.size(); i++)
         {
            const far_pt_t &a = v[i], &b = v[(i + 1) % v.size()];
            const double fa = val(a), fb = val(b);
            if (fa >= 0) u.push_back(a);
            if ((fa >= 0) != (fb >= 0))
            {
               far_pt_t q = cut(a, b, fa / (fa - fb));
               if (side < 2) q.x = side == 0 ? P.gx0 : P.gx1; else q.y = side == 2 ? P.gy0 : P.gy1;      /* on the side, exactly */
               u.push_back(q);
            }
         }
         v.swap(u);
      }
      if (v.size() >= 3) emit(v, true);
      return;
   }
   std::vector<far_pt_t> run;
   for (size_t i = 0; i + 1 < v.size(); i++)
   {
      /* Liang-Barsky: the part t0 .. t1 of the segment inside the box */
      const far_pt_t &a = v[i], &b = v[i + 1];
      double t0 = 0, t1 = 1;
      const double dx = b.x - a.x, dy = b.y - a.y;
      const double pp[4] = { -dx, dx, -dy, dy }, qq[4] = { a.x - P.gx0, P.gx1 - a.x, a.y - P.gy0, P.gy1 - a.y };
      bool out = false;
      for (int k = 0; k < 4 && !out; k++)
      {
         if (pp[k] == 0) { if (qq[k] < 0) out = true; continue; }
         const double t = qq[k] / pp[k];
         if (pp[k] < 0) { if (t > t1) out = true; else if (t > t0) t0 = t; }
         else { if (t < t0) out = true; else if (t < t1) t1 = t; }
      }
      if (out || !(t0 < t1)) { if (run.size() >= 2) emit(run, false); run.clear(); continue; }
      if (t0 > 0 || run.empty())
      {
         if (run.size() >= 2) emit(run, false);
         run.clear();
         run.push_back(t0 > 0 ? cut(a, b, t0) : a);
      }
      run.push_back(t1 < 1 ? cut(a, b, t1) : b);
      if (t1 < 1) { emit(run, false); run.clear(); }
   }
   if (run.size() >= 2) emit(run, false);
}

static int draw_main(smr_smrule_t *r, smr_osm_obj_t *o)
{
   draw_rule_t *d = (draw_rule_t *) r->data;
   if (d == nullptr) return -1;
   if (o->type == SMR_OSM_WAY) { const double t0 = now_s(); const int e = draw_way(d, (const smr_osm_way_t *) o); pstat_[0] += now_s() - t0; return e; }
   if (o->type == SMR_OSM_REL)
   {
      const smr_osm_rel_t *rel = (const smr_osm_rel_t *) o;
      for (int i = 0; i < rel->mem_cnt; i++)
      {
         if (rel->mem[i].type != SMR_OSM_WAY) continue;
         const smr_osm_way_t *w = (const smr_osm_way_t *) H.get_object(SMR_OSM_WAY, rel->mem[i].id);
         if (w == nullptr) continue;
         int e = draw_way(d, w);
         if (e < 0) return e;
      }
      return 0;
   }
   plog(SMR_LOG_WARN, "draw() may not be applied to object type %d", o->type);
   return 1;
}

/* everything act_draw_fini does with the ways a rule has collected (also for img on ways: a rule of nothing but an image fill) */
static int draw_finish(draw_rule_t *d)
{
   int ret = 0;
   if (d->st.directional && d->max_area >= 0)
   {
      /* the largest polygon is clockwise: a page-sized polygon goes first, so that it becomes a hole
       * (:1103-1109); the fill colour is the one of that first way unless it is the hole (:1116-1129) */
      const bool cw = d->max_cw;                        /* of the chart's own ways: the page rectangle below goes through draw_way too */
      const uint32_t max_col = d->max_col, min_col = d->min_col;
      if (d->max_cw)
      {
         const smr_osm_way_t *pw = H.page_way ? H.page_way() : nullptr;
         if (pw == nullptr) plog(SMR_LOG_ERR, "NULL pointer to caught");      /* render_poly_line's message for page_way() == NULL */
         else { const char *k1 = d->fill_cs.key, *k2 = d->border_cs.key; d->fill_cs.key = d->border_cs.key = nullptr;
                draw_way(d, pw); d->fill_cs.key = k1; d->border_cs.key = k2; if (k1) d->fill_col.push_back(0); }
      }
      /* key colouring with a clockwise largest way: act_draw_fini sets no colour (:1117-1125), the fill takes the source that
       * render_poly_line (:1002) left for the last way of the sorted list, the smallest one */
      d->st.fill_col = cw ? (d->fill_cs.key ? min_col : d->fill_cs.col) : max_col;
      /* with key colouring every way carries its own colour; the closed ways are ONE fill in the colour decided above */
      if (d->fill_cs.key)
         for (size_t w = 0; w + 1 < d->way_off.size() && w < d->fill_col.size(); w++)
         {
            const int64_t a = d->way_off[w], b = d->way_off[w + 1];
            if (b - a >= 4 && d->ref[(size_t) a] == d->ref[(size_t) b - 1]) d->fill_col[w] = d->st.fill_col;
         }
   }
   const int64_t n_ways = (int64_t) d->way_off.size() - 1, n_refs = (int64_t) d->ref.size();
   if (n_ways > 0 && n_refs > 0)
   {
      const double t_up = now_s();
      size_t n = d->lat.size();
      if (n & 1) { d->lat.push_back(d->lat.back()); d->lon.push_back(d->lon.back()); n++; }
      /* a rule that is kept for the caption scan's windows owns its arrays; otherwise they come from the grow-only pool */
      /* (an image fill is not replayed into the cut-outs of later caption scans: they see the rule's colours only) */
      const bool keep = cutout_replay() && (d->st.fill_used || d->st.border_used);
      auto own_put = [&](int i, const void *h, size_t bytes) -> void *
      {
         if (!keep) return h ? pool_put(i, h, bytes) : pool_get(i, bytes);
         void *q = nullptr;
         if (smr_hip_malloc(&q, bytes ? bytes : 16) != SMR_OK) return nullptr;
         if (h && bytes && smr_hip_memcpy_h2d(q, h, bytes) != SMR_OK) { smr_hip_free(q); return nullptr; }
         return q;
      };
      void *lat = pool_put(0, d->lat.data(), n * 8), *lon = pool_put(1, d->lon.data(), n * 8), *x = own_put(2, nullptr, n * 8), *y = own_put(3, nullptr, n * 8);
      void *off = own_put(4, d->way_off.data(), d->way_off.size() * 8), *ref = own_put(5, d->ref.data(), d->ref.size() * 4);
      void *fc = nullptr, *bc = nullptr;
      bool ok = lat && lon && x && y && off && ref;
      if (ok && d->fill_cs.key) ok = (fc = own_put(6, d->fill_col.data(), d->fill_col.size() * 4)) != nullptr;
      if (ok && d->border_cs.key) ok = (bc = own_put(7, d->border_col.data(), d->border_col.size() * 4)) != nullptr;
      if (ok) ok = smr_hip_project((const double *) lat, (const double *) lon, (int64_t) n, (double *) x, (double *) y,
                                   nullptr, nullptr, nullptr) == SMR_OK;
      pstat_[2] += now_s() - t_up;
      const double t_dr = now_s();
      if (ok)
      {
         smr_draw_batch_t b;
         memset(&b, 0, sizeof(b));
         b.d_x = (const double *) x; b.d_y = (const double *) y; b.n_nodes = (int64_t) n;
         b.d_way_off = (const int64_t *) off; b.d_ref = (const int32_t *) ref; b.n_ways = n_ways; b.n_refs = n_refs;
         b.d_way_fill_col = (const uint32_t *) fc;
         b.d_way_border_col = (const uint32_t *) bc;
         /* the image's group is popped before the colours' (act_draw_fini :1081-1082) */
         if (!d->pat.empty()) ok = smr_hip_draw_pattern(P.canvas, &d->st, &b, d->pat.data(), d->pat_w, d->pat_h) == SMR_OK;
         if (ok && (d->st.fill_used || d->st.border_used)) ok = smr_hip_draw(P.canvas, &d->st, &b) == SMR_OK;
         if (keep && ok)
         {
            /* only what is on the chart may appear in the cut-outs: a rule whose draw failed is not replayed */
            kept_.emplace_back();
            kept_.back().st = d->st; kept_.back().b = b;        /* freed by kept_release */
         }
      }
      if (keep && !ok) { void *q[6] = { x, y, off, ref, fc, bc }; for (void *v : q) if (v) smr_hip_free(v); }
      pstat_[3] += now_s() - t_dr;
      const double t_sy = now_s();
      if (ok) ok = smr_hip_sync() == SMR_OK;
      pstat_[4] += now_s() - t_sy;
      if (!ok) { plog(SMR_LOG_ERR, "draw: %s", smr_hip_last_error()); ret = -1; }
      else if (smr_hip_draw_status() & SMR_DRAW_DEGRADED_ROWS)
         plog(SMR_LOG_WARN, "draw: some pixel rows with more than 64 crossing edges were sub-sampled (anti-aliasing may differ slightly)");
      else plog(SMR_LOG_INFO, "draw: %lld ways, %lld refs rendered on the GPU%s", (long long) n_ways, (long long) n_refs, d->pat.empty() ? "" : ", closed ways filled with the rule's image");
   }
   return ret;
}

static int draw_fini(smr_smrule_t *r)
{
   draw_rule_t *d = (draw_rule_t *) r->data;
   if (d == nullptr) return 0;
   const int ret = draw_finish(d);
   delete d;
   r->data = nullptr;
   return ret;
}

/* ---------------------------------------------------------------------------------------- */
/* host libcairo for text extents (glyph outlines stay on the host, BASELINE north_star)    */
/* ---------------------------------------------------------------------------------------- */
typedef struct { double x_bearing, y_bearing, width, height, x_advance, y_advance; } cr_text_extents_t;
typedef struct { double ascent, descent, height, max_x_advance, max_y_advance; } cr_font_extents_t;
typedef struct { double x, y, width, height; } cr_rectangle_t;

struct cairo_api_t
{
   void *lib = nullptr, *fc = nullptr;
   void *(*recording_surface_create)(int, const cr_rectangle_t *);
   void *(*image_surface_create)(int, int, int);
   void *(*create)(void *);
   void (*destroy)(void *);
   void (*surface_destroy)(void *);
   void (*surface_flush)(void *);
   unsigned char *(*image_surface_get_data)(void *);
   int (*image_surface_get_stride)(void *);
   void *(*font_options_create)(void);
   void (*font_options_set_hint_metrics)(void *, int);
   void (*set_font_options)(void *, const void *);
   void (*font_options_destroy)(void *);
   void (*select_font_face)(void *, const char *, int, int);
   void (*set_font_size)(void *, double);
   void (*font_extents)(void *, cr_font_extents_t *);
   void (*text_extents)(void *, const char *, cr_text_extents_t *);
   void (*scale)(void *, double, double);
   void (*set_source_rgba)(void *, double, double, double, double);
   void (*rectangle)(void *, double, double, double, double);
   void (*fill)(void *);
   void (*translate)(void *, double, double);
   void (*rotate)(void *, double);
   void (*move_to)(void *, double, double);
   void (*show_text)(void *, const char *);
   void (*set_source_surface)(void *, void *, double, double);
   void (*paint)(void *);
   void (*save)(void *);
   void (*restore)(void *);
   void (*set_operator)(void *, int);
   void *(*image_surface_create_from_png)(const char *);
   void (*surface_mark_dirty)(void *);
   int (*surface_status)(void *);
   int (*image_surface_get_width)(void *);
   int (*image_surface_get_height)(void *);
   /* optional: fontconfig face like the reference's CAIRO_HAS_FC_FONT branch (:1345-1354) */
   void *(*ft_font_face_create_for_pattern)(void *);
   void (*set_font_face)(void *, void *);
   void (*font_face_destroy)(void *);
   void *(*FcNameParse)(const unsigned char *);
   void (*FcPatternDestroy)(void *);
};
static cairo_api_t CR;

#define CR_LOAD(field, name) do { *(void **) (&CR.field) = dlsym(CR.lib, name); if (!CR.field) { plog(SMR_LOG_ERR, "libcairo lacks %s", name); return -1; } } while (0)

static int cairo_load(void)
{
   if (CR.lib) return 0;
   const char *cand[] = { getenv("SMR_HIP_LIBCAIRO"), "libcairo.so.2", "/opt/conda/lib/libcairo.so.2", nullptr };
   for (int i = 0; i < 4 && !CR.lib; i++)
      if (cand[i] && *cand[i]) CR.lib = dlopen(cand[i], RTLD_NOW | RTLD_GLOBAL);
   if (!CR.lib)
   {
      plog(SMR_LOG_ERR, "cap: libcairo.so.2 (text extents) cannot be loaded: %s", dlerror());
      return -1;
   }
   CR_LOAD(recording_surface_create, "cairo_recording_surface_create"); CR_LOAD(image_surface_create, "cairo_image_surface_create");
   CR_LOAD(create, "cairo_create"); CR_LOAD(destroy, "cairo_destroy"); CR_LOAD(surface_destroy, "cairo_surface_destroy");
   CR_LOAD(surface_flush, "cairo_surface_flush"); CR_LOAD(image_surface_get_data, "cairo_image_surface_get_data");
   CR_LOAD(image_surface_get_stride, "cairo_image_surface_get_stride"); CR_LOAD(font_options_create, "cairo_font_options_create");
   CR_LOAD(font_options_set_hint_metrics, "cairo_font_options_set_hint_metrics"); CR_LOAD(set_font_options, "cairo_set_font_options");
   CR_LOAD(font_options_destroy, "cairo_font_options_destroy"); CR_LOAD(select_font_face, "cairo_select_font_face");
   CR_LOAD(set_font_size, "cairo_set_font_size"); CR_LOAD(font_extents, "cairo_font_extents"); CR_LOAD(text_extents, "cairo_text_extents");
   CR_LOAD(scale, "cairo_scale"); CR_LOAD(set_source_rgba, "cairo_set_source_rgba"); CR_LOAD(rectangle, "cairo_rectangle");
   CR_LOAD(fill, "cairo_fill");
   CR_LOAD(translate, "cairo_translate"); CR_LOAD(rotate, "cairo_rotate"); CR_LOAD(move_to, "cairo_move_to"); CR_LOAD(show_text, "cairo_show_text");
   CR_LOAD(set_source_surface, "cairo_set_source_surface"); CR_LOAD(paint, "cairo_paint");
   CR_LOAD(save, "cairo_save"); CR_LOAD(restore, "cairo_restore");
   CR_LOAD(set_operator, "cairo_set_operator"); CR_LOAD(image_surface_create_from_png, "cairo_image_surface_create_from_png");
   CR_LOAD(surface_mark_dirty, "cairo_surface_mark_dirty");
   CR_LOAD(surface_status, "cairo_surface_status"); CR_LOAD(image_surface_get_width, "cairo_image_surface_get_width");
   CR_LOAD(image_surface_get_height, "cairo_image_surface_get_height");
   *(void **) (&CR.ft_font_face_create_for_pattern) = dlsym(CR.lib, "cairo_ft_font_face_create_for_pattern");
   *(void **) (&CR.set_font_face) = dlsym(CR.lib, "cairo_set_font_face");
   *(void **) (&CR.font_face_destroy) = dlsym(CR.lib, "cairo_font_face_destroy");
   *(void **) (&CR.FcNameParse) = dlsym(RTLD_DEFAULT, "FcNameParse");
   *(void **) (&CR.FcPatternDestroy) = dlsym(RTLD_DEFAULT, "FcPatternDestroy");
   return 0;
}

/* ---------------------------------------------------------------------------------------- */
/* cap (angle=auto)                                                                         */
/* ---------------------------------------------------------------------------------------- */
struct cap_item_t
{
   smr_osm_obj_t *o;
   double lat, lon;
   int pw, ph, px;            /* plane size / first opaque column in pt, ints like cairo_smr_plane's arguments */
   uint32_t col;
   short pos;
   double tw, hgt, ascent;    /* tx.width + tx.x_bearing, max(tx.height, ascent + descent), fe.ascent: the auto_ctx rectangle */
   double size_mm, descent;   /* for the glyphs */
   double fixed_a;            /* fixed-angle rules: the angle of cap_coord :2353-2356 */
   uint32_t fillcol;          /* fillcolor= box behind the text (cap_rect :2188-2199) */
   std::string text;
};

struct cap_rule_t
{
   short pos = 0;
   col_spec_t cs;
   const char *font = "serif", *key = nullptr;
   std::vector<std::string> klist;
   double size = 0, xoff = 0, yoff = 0, phase = 0, weight = 1;
   double min_auto_size = 0.5, max_auto_size = 12.0, min_area_size = 8.0, auto_scale = 0.2;   /* src/smrender_dev.h:107-110 */
   const char *halignkey = nullptr, *valignkey = nullptr;
   int hide = 0;
   bool autorot = false;      /* angle=auto: the scan; otherwise angle= / anglekey= */
   double angle = 0, bgbox_scale = 1.2;
   const char *akey = nullptr;
   bool fill_used = false;
   bool fontbox = false, mkarea = false;      /* objects for the host's data: cap_geo_rect (:2114-2176), dv_mkarea (:1772-1801) */
   col_spec_t fill_cs;
   bool have_auto_col = false; uint32_t auto_col = 0;      /* auto_ctx's source: the colour of the FIRST object act_cap_main sees, matched or not (:2455-2459) */
   void *sfc = nullptr, *ctx = nullptr;
   std::vector<cap_item_t> items;
};

static double mm2pt(double mm) { return mm * 72 / 25.4; }

/* parse_alignment_str, src/smrparse.c:667-701 */
static int alignment_str(const char *s)
{
   static const struct { const char *n; int p; } tab[] = {
      { "east", SMR_POS_E }, { "west", SMR_POS_W }, { "north", SMR_POS_N }, { "south", SMR_POS_S },
      { "northeast", SMR_POS_E | SMR_POS_N }, { "northwest", SMR_POS_W | SMR_POS_N }, { "southeast", SMR_POS_E | SMR_POS_S },
      { "southwest", SMR_POS_W | SMR_POS_S }, { "center", 0 }, { "middle", 0 } };
   if (s == nullptr || *s == '\0') return 0;
   for (auto &t : tab) if (!strcasecmp(s, t.n)) return t.p;
   plog(SMR_LOG_WARN, "unknown alignment '%s'", s);
   return 0;
}

static int alignment(const smr_action_t *act)      /* parse_alignment, src/smrparse.c:704-720 */
{
   if (H.parse_alignment) return H.parse_alignment(act);
   char *s; int pos = 0;
   if ((s = H.get_param("align", nullptr, act)) != nullptr) return alignment_str(s);
   if ((s = H.get_param("halign", nullptr, act)) != nullptr) pos |= alignment_str(s) & (SMR_POS_E | SMR_POS_W);
   if ((s = H.get_param("valign", nullptr, act)) != nullptr) pos |= alignment_str(s) & (SMR_POS_N | SMR_POS_S);
   return pos;
}

/* the font of a caption context: hint metrics off, a fontconfig face when libcairo has it (act_cap_ini :1335-1357) */
static void cap_font(void *ctx, const char *font)
{
   void *fo = CR.font_options_create();
   CR.font_options_set_hint_metrics(fo, 1 /* CAIRO_HINT_METRICS_OFF */);
   CR.set_font_options(ctx, fo);
   CR.font_options_destroy(fo);
   void *pat = nullptr;
   if (CR.FcNameParse && CR.FcPatternDestroy && CR.ft_font_face_create_for_pattern && CR.set_font_face && CR.font_face_destroy &&
       (pat = CR.FcNameParse((const unsigned char *) font)) != nullptr)
   {
      void *cfc = CR.ft_font_face_create_for_pattern(pat);
      CR.FcPatternDestroy(pat);
      CR.set_font_face(ctx, cfc);
      CR.font_face_destroy(cfc);
   }
   else CR.select_font_face(ctx, font, 0, 0);
}

static int cap_ini(smr_smrule_t *r)
{
   int e;
   host_resolve();
   if ((e = plugin_setup()) != 0) return e;
   char *s; double angle = 0;
   /* parse_auto_rot, src/smrparse.c:1033-1074; anglekey overrides angle=auto (act_cap_ini :1294-1299) */
   bool autorot = false;
   if ((s = H.get_param("angle", &angle, r->act)) != nullptr)
   {
      if (!strcasecmp(s, "auto")) { autorot = true; angle = 0; }
      else if (!strcasecmp(s, "majoraxis")) angle = SMR_MAJORAXIS;          /* a closed way's caption runs along its longest diagonal (cap_way) */
      else angle = fmod(angle, 360.0);
   }
   if (angle == SMR_MAJORAXIS && !H.get_object)
   {
      plog(SMR_LOG_ERR, "cap: angle=majoraxis needs the host symbol get_object");
      return -1;
   }
   const char *akey = H.get_param("anglekey", nullptr, r->act);
   if (akey != nullptr && autorot)
   {
      plog(SMR_LOG_NOTICE, "anglekey=%s overrides angle=auto", akey);
      autorot = false; angle = 0;
   }
   /* fontbox=1 (act_cap_ini :1291) and mkarea=1 (parse_auto_rot, src/smrparse.c:1062: read with angle=auto only) add nodes and ways
    * to the host's data through the host's own malloc_node / malloc_way / put_object */
   const bool fontbox = H.get_param_bool("fontbox", r->act) != 0;
   const bool mkarea = autorot && H.get_param_bool("mkarea", r->act) != 0;
   if ((fontbox || mkarea) && !H.can_make_objects())
   {
      plog(SMR_LOG_ERR, "cap: fontbox=1 / mkarea=1 need the host symbols malloc_node, malloc_way, osm_node_default, osm_way_default, put_object");
      return -1;
   }
   if (!H.match_attr || !H.realloc_tags)
   {
      plog(SMR_LOG_ERR, "cap: host symbols match_attr/realloc_tags not found");
      return -1;
   }
   if (autorot && !cutout_replay() && (P.band_y0 != 0 || P.band_h != P.H))
   {
      /* a crop of the canvas near the band's edge reaches into the neighbouring bands; the plugin is one process and exchanges no
       * halo rows.  (The default, cut-outs rendered again from the kept rules, does not look at the canvas: every rank of a banded
       * chart scans all captions from the geometry it holds anyway and composites the glyphs that fall into its rows.) */
      plog(SMR_LOG_ERR, "cap: the canvas is a row band (SMR_HIP_BAND_Y0/_H) and SMR_HIP_CUTOUT=crop: crops need the neighbours' rows, which only a "
                        "driver of the C-ABI exchanges (smr_hip_canvas_set_halo, smrender_amd/dist.py)");
      return 1;
   }
   if (cairo_load() != 0) return -1;

   cap_rule_t *cap = new cap_rule_t();
   cap->autorot = autorot; cap->angle = angle; cap->akey = akey; cap->fontbox = fontbox; cap->mkarea = mkarea;
   (void) H.get_param("bgbox_scale", &cap->bgbox_scale, r->act);
   if ((s = H.get_param("fillcolor", nullptr, r->act)) != nullptr) { parse_col_spec(s, &cap->fill_cs); cap->fill_used = true; }
   if ((s = H.get_param("font", nullptr, r->act)) != nullptr) cap->font = s;
   else plog(SMR_LOG_NOTICE, "parameter 'font' missing, defaulting to \"%s\"", cap->font);
   if (H.get_param("size", &cap->size, r->act) == nullptr && r->oo->type == SMR_OSM_NODE)
   {
      cap->size = 4.0;      /* DEFAULT_CAP_SIZE */
      plog(SMR_LOG_NOTICE, "parameter 'size' missing, defaulting to %.1f", cap->size);
   }
   cap->xoff = cap->yoff = mm2pt(cap->size) / 2;
   if ((cap->key = H.get_param("key", nullptr, r->act)) == nullptr)
   {
      plog(SMR_LOG_WARN, "parameter 'key' missing");
      delete cap;
      return 1;
   }
   if ((s = H.get_param("color", nullptr, r->act)) != nullptr) parse_col_spec(s, &cap->cs);
   (void) H.get_param("min_size", &cap->min_auto_size, r->act);
   (void) H.get_param("max_size", &cap->max_auto_size, r->act);
   (void) H.get_param("min_area", &cap->min_area_size, r->act);
   (void) H.get_param("auto_scale", &cap->auto_scale, r->act);
   (void) H.get_param("xoff", &cap->xoff, r->act);
   (void) H.get_param("yoff", &cap->yoff, r->act);
   if (H.get_param("weight", &cap->weight, r->act) == nullptr) cap->weight = 1.0;
   if (cap->weight > 1.0) cap->weight = 1.0;
   else if (cap->weight < -1.0) cap->weight = -1.0;
   (void) H.get_param("phase", &cap->phase, r->act);
   cap->hide = H.get_param_bool("hide", r->act);
   cap->pos = (short) alignment(r->act);
   if ((cap->halignkey = H.get_param("alignkey", nullptr, r->act)) != nullptr)
   {
      cap->valignkey = cap->halignkey;
      cap->pos &= ~(SMR_POS_N | SMR_POS_S | SMR_POS_E | SMR_POS_W);
   }
   else
   {
      if ((cap->halignkey = H.get_param("halignkey", nullptr, r->act)) != nullptr) cap->pos &= ~(SMR_POS_E | SMR_POS_W);
      if ((cap->valignkey = H.get_param("valignkey", nullptr, r->act)) != nullptr) cap->pos &= ~(SMR_POS_N | SMR_POS_S);
   }
   if (*cap->key == '*') { cap->key++; cap->pos |= SMR_POS_UC; }
   /* parse_keylist, src/smrparse.c:859-...: "(a|b|c)" is a list of alternatives */
   {
      std::string k(cap->key);
      if (k.size() > 1 && k.front() == '(' && k.back() == ')')
      {
         size_t a = 1;
         while (a < k.size())
         {
            size_t b = k.find('|', a);
            if (b == std::string::npos) b = k.size() - 1;
            cap->klist.push_back(k.substr(a, b - a));
            a = b + 1;
         }
      }
      else cap->klist.push_back(k);
   }
   /* the measuring context: like cap.ctx on the recording surface, hint metrics off (:1335-1357) */
   cr_rectangle_t ext = { 0, 0, P.proj.w * 72 / P.dpi, P.proj.h * 72 / P.dpi };
   cap->sfc = CR.recording_surface_create(0x3000, &ext);
   cap->ctx = CR.create(cap->sfc);
   cap_font(cap->ctx, cap->font);
   r->data = cap;
   return 0;
}

static void strupper(char *s) { for (; *s; s++) if (*s >= 'a' && *s <= 'z') *s -= 32; }

/* fmod2, src/smath.c:78-82 */
static double fmod2(double a, double n) { a = fmod(a, n); return a < 0 ? a + n : a; }

/* coord_diffp, src/smath.c:34-52: orthodrome distance and bearing (degrees, 0 = north, clockwise) from src to dst */
static void coord_diff_deg(double slat, double slon, double dlat_, double dlon_, double *dist, double *bearing)
{
   const double r = M_PI / 180.0;
   const double dlat = dlat_ - slat, dlon = (dlon_ - slon) * cos((slat + dlat_) / 2.0 * r);
   *bearing = fmod2(atan2(dlon, dlat) / r, 360.0);
   *dist = acos(sin(slat * r) * sin(dlat_ * r) + cos(slat * r) * cos(dlat_ * r) * cos((dlon_ - slon) * r)) / r;
}

/* farthest_node, src/smrules_cairo.c:1151-1180: the index of the way's node farthest from (lat, lon); -1 if none is farther than 0 */
static int farthest_node(double lat, double lon, const smr_osm_way_t *w, double *dist, double *bearing)
{
   int ref = -1;
   *dist = 0; *bearing = 0;
   for (int i = 0; i < w->ref_cnt; i++)
   {
      const smr_osm_node_t *n = (const smr_osm_node_t *) H.get_object(SMR_OSM_NODE, w->ref[i]);
      if (n == nullptr) continue;
      double d, b;
      coord_diff_deg(lat, lon, n->lat, n->lon, &d, &b);
      if (d > *dist) { *dist = d; *bearing = b; ref = i; }
   }
   return ref;
}

/* area_axis, src/smrules_cairo.c:1183-1240: walks from node 0 to the node farthest from it, from there to the node farthest
 * from that one, ... until the distance stops growing; *a = the bearing of the longest hop found */
static int area_axis(const smr_osm_way_t *w, double *a)
{
   if (w->ref_cnt < 2) return -1;
   double fdist = 0, fbearing = 0;
   int last = 0;
   for (int guard = 0; guard <= w->ref_cnt; guard++)        /* the distance grows strictly with every hop: at most ref_cnt of them */
   {
      const smr_osm_node_t *n = (const smr_osm_node_t *) H.get_object(SMR_OSM_NODE, w->ref[last]);
      if (n == nullptr) return -1;                          /* the reference spins here for ever */
      double dist, bearing;
      const int nref = farthest_node(n->lat, n->lon, w, &dist, &bearing);
      if (nref == 0) break;
      if (nref == -1) return -1;
      if (dist <= fdist) break;
      last = nref; fdist = dist; fbearing = bearing;
   }
   *a = fbearing;
   return 0;
}

/* the part of cap_coord (:2202-2281) that runs before find_angle: extents -> plane geometry */
static int cap_collect(cap_rule_t *cap, double size_mm, double angle_deg, double lat, double lon, const smr_bstring_t *str, smr_osm_obj_t *o)
{
   if (size_mm == 0.0) return 0;
   std::string buf(str->buf, (size_t) str->len);
   if (cap->pos & SMR_POS_UC) strupper(&buf[0]);
   cr_text_extents_t tx; cr_font_extents_t fe;
   CR.set_font_size(cap->ctx, mm2pt(size_mm));
   CR.font_extents(cap->ctx, &fe);
   CR.text_extents(cap->ctx, buf.c_str(), &tx);
   double width = tx.width + tx.x_bearing + cap->xoff, height = fe.ascent;
   cap_item_t it;
   it.o = o; it.lat = lat; it.lon = lon;
   it.fixed_a = 0; it.fillcol = cap->fill_used ? color_by_cs(o, &cap->fill_cs) : 0;
   if (cap->autorot) it.pos = (cap->pos & 0xc) ? (short) ((cap->pos & 0xfff0) | SMR_POS_E) : cap->pos;
   else
   {
      /* alignment from the object's tags (:2233-2248, retr_align_key_pos :2088-2111), angle from anglekey (:2353-2356) */
      auto key_pos = [&](const char *key) -> int
      {
         const int m = H.match_attr(o, key, nullptr);
         if (m < 0 || o->otag[m].v.len > 9) return 0;
         return alignment_str(std::string(o->otag[m].v.buf, (size_t) o->otag[m].v.len).c_str());
      };
      short pos = cap->pos;
      if (cap->halignkey != nullptr || cap->valignkey != nullptr)
      {
         if (cap->halignkey == cap->valignkey) pos = (short) key_pos(cap->halignkey);
         else
         {
            if (cap->halignkey != nullptr) pos = (short) ((pos & ~(SMR_POS_E | SMR_POS_W)) | (key_pos(cap->halignkey) & (SMR_POS_E | SMR_POS_W)));
            if (cap->valignkey != nullptr) pos = (short) ((pos & ~(SMR_POS_N | SMR_POS_S)) | (key_pos(cap->valignkey) & (SMR_POS_N | SMR_POS_S)));
         }
      }
      it.pos = pos;
      int m;
      if (cap->akey != nullptr && (m = H.match_attr(o, cap->akey, nullptr)) >= 0)
         it.fixed_a = strtod(std::string(o->otag[m].v.buf, (size_t) o->otag[m].v.len).c_str(), nullptr) * M_PI / 180.0;
      it.fixed_a += (360 - angle_deg) * M_PI / 180.0;
   }
   it.col = color_by_cs(o, &cap->cs);
   it.tw = tx.width + tx.x_bearing; it.hgt = fmax(tx.height, fe.ascent + fe.descent); it.ascent = fe.ascent;
   it.size_mm = size_mm; it.descent = fe.descent; it.text = buf;
   /* cairo_smr_plane(int w, int h, int x, int col): the double arguments are truncated to int pt (:1536, :2272, :2277) */
   if (cap->pos & 0xc) { it.pw = (int) (width * 2); it.ph = (int) height; it.px = (int) width; }
   else { it.pw = (int) width; it.ph = (int) height; it.px = 0; }
   cap->items.push_back(it);
   return 0;
}

static int cap_main(smr_smrule_t *r, smr_osm_obj_t *o)
{
   cap_rule_t *cap = (cap_rule_t *) r->data;
   if (cap == nullptr) return -1;
   if (!cap->have_auto_col) { cap->auto_col = color_by_cs(o, &cap->cs); cap->have_auto_col = true; }
   int n = -1;
   for (size_t i = 0; i < cap->klist.size(); i++)
      if ((n = H.match_attr(o, cap->klist[i].c_str(), nullptr)) >= 0) break;
   if (n == -1) return 0;
   if (o->type == SMR_OSM_NODE)
   {
      const smr_osm_node_t *nd = (const smr_osm_node_t *) o;
      return cap_collect(cap, cap->size, cap->angle, nd->lat, nd->lon, &o->otag[n].v, o);
   }
   if (o->type == SMR_OSM_WAY)
   {
      /* cap_way, :2395-2444: closed ways only, caption at the centroid, size from the area */
      const smr_osm_way_t *w = (const smr_osm_way_t *) o;
      if (!(w->ref_cnt >= 4 && w->ref[0] == w->ref[w->ref_cnt - 1])) return 0;
      if (!H.poly_area) { plog(SMR_LOG_WARN, "cap: host symbol poly_area not found, way %lld skipped", (long long) o->id); return 0; }
      struct smr_coord c; double ar;
      if (H.poly_area(w, &c, &ar)) return 0;
      double size = cap->size;
      if (size == 0.0)
      {
         if (!H.rdata_square_mm || !H.rdata_square_nm) return 0;
         double area_mm2 = fabs(ar) * H.rdata_square_mm() / H.rdata_square_nm();
         size = cap->auto_scale * sqrt(area_mm2);
         if (cap->max_auto_size != 0.0 && size > cap->max_auto_size) size = cap->max_auto_size;
         if (cap->min_auto_size != 0.0 && size < cap->min_auto_size)
            size = area_mm2 < cap->min_area_size ? 0.0 : cap->min_auto_size;
      }
      double angle = cap->angle;
      if (angle == SMR_MAJORAXIS)
      {
         /* :2431-2441: the bearing of the way's longest diagonal as a math angle, turned to read upright.  A way whose axis
          * cannot be found keeps 720 here, as the reference's tmp_cap.angle does, and so ends at 90 */
         (void) area_axis(w, &angle);
         angle = fmod2(90 - angle, 360);
         if (angle > 90 && angle <= 270) angle -= 180.0;
      }
      return cap_collect(cap, size, angle, c.lat, c.lon, &o->otag[n].v, o);
   }
   return 1;
}

/* cairo_smr_plane, :1536-1558, with the real libcairo: a w x h pt image at dpi/72, the
 * rectangle [x,w] x [0,h] filled in the caption colour */
static int make_plane(int w_pt, int h_pt, int x_pt, uint32_t col, int dpi, std::vector<uint32_t> &pix, int *w_px, int *h_px)
{
   double s = (double) dpi / 72.0;
   int W = (int) round(w_pt * s), Hh = (int) round(h_pt * s);
   *w_px = W; *h_px = Hh;
   if (W <= 0 || Hh <= 0) return -1;
   void *sfc = CR.image_surface_create(0 /* ARGB32 */, W, Hh);
   void *ctx = CR.create(sfc);
   CR.scale(ctx, s, s);
   CR.set_source_rgba(ctx, (double) ((col >> 16) & 0xff) / 255.0, (double) ((col >> 8) & 0xff) / 255.0, (double) (col & 0xff) / 255.0,
                      1.0 - (double) (((col & 0x7f000000u) >> 23) & 0xff) / 255.0);
   CR.rectangle(ctx, x_pt, 0, w_pt - x_pt, h_pt);
   CR.fill(ctx);
   CR.destroy(ctx);
   CR.surface_flush(sfc);
   const unsigned char *d = CR.image_surface_get_data(sfc);
   int stride = CR.image_surface_get_stride(sfc);
   pix.resize((size_t) W * Hh);
   for (int y = 0; y < Hh; y++) memcpy(&pix[(size_t) y * W], d + (size_t) y * stride, (size_t) W * 4);
   CR.surface_destroy(sfc);
   return 0;
}

static void add_tag(smr_osm_obj_t *o, const char *key, const char *sub, const char *val)     /* :2299-2348 */
{
   if (H.realloc_tags(o, o->tag_cnt + 1) == -1) return;
   smr_otag *t = &o->otag[o->tag_cnt - 1];
   size_t kl = strlen(key) + strlen(sub) + 2, vl = strlen(val) + 1;
   t->k.buf = (char *) malloc(kl); t->v.buf = (char *) malloc(vl < 8 ? 8 : vl);
   if (!t->k.buf || !t->v.buf) { free(t->k.buf); free(t->v.buf); o->tag_cnt--; return; }
   t->k.len = snprintf(t->k.buf, kl, "%s:%s", key, sub);
   t->v.len = snprintf(t->v.buf, vl < 8 ? 8 : vl, "%s", val);
}

/* pos_offset, src/smrules_cairo.c:1478-1508 */
static void pos_offset(int pos, double width, double height, double xoff, double yoff, double *ox, double *oy)
{
   switch (pos & 0x3)
   {
      case SMR_POS_N: *oy = 0 - yoff; break;
      case SMR_POS_S: *oy = height + yoff; break;
      default: *oy = height / 2;
   }
   switch (pos & 0xc)
   {
      case SMR_POS_E: *ox = 0 + xoff; break;
      case SMR_POS_W: *ox = -width - xoff; break;
      default: *ox = -width / 2;
   }
}

/* The reference scans the captions of one rule one after the other, and each finished caption
 * leaves a filled rectangle on the rule's private background copy (auto_sfc, :2376-2382) that later
 * captions of the same rule see.  Captions whose cut-outs cannot reach each other's rectangles are
 * independent: they get the same LEVEL and are scanned in one GPU batch; levels run in order. */
static void cap_levels(const std::vector<double> &cx, const std::vector<double> &cy, const std::vector<double> &reach,
                       std::vector<int> &level, int &n_levels)
{
   const size_t n = cx.size();
   level.assign(n, 0);
   n_levels = n ? 1 : 0;
   double rmax = 1;
   for (size_t i = 0; i < n; i++) rmax = std::max(rmax, reach[i]);
   const double cell = 2 * rmax;
   std::map<std::pair<long, long>, std::vector<size_t>> grid;
   for (size_t i = 0; i < n; i++)
   {
      const long gx = (long) floor(cx[i] / cell), gy = (long) floor(cy[i] / cell);
      int lv = 0;
      for (long dy = -1; dy <= 1; dy++)
         for (long dx = -1; dx <= 1; dx++)
         {
            auto f = grid.find({ gx + dx, gy + dy });
            if (f == grid.end()) continue;
            for (size_t j : f->second)
               if (hypot(cx[i] - cx[j], cy[i] - cy[j]) <= reach[i] + reach[j]) lv = std::max(lv, level[j] + 1);
         }
      level[i] = lv;
      n_levels = std::max(n_levels, lv + 1);
      grid[{ gx, gy }].push_back(i);
   }
}

/* ---------------------------------------------------------------------------------------- */
/* find_angle over the reference's own cut-outs: atlases of windows (windows.hip)             */
/* ---------------------------------------------------------------------------------------- */
struct scan_item_t
{
   size_t id;                          /* the caller's index: angle[id] is the result */
   double cx, cy;                      /* anchor, chart px */
   int fw, fh;                         /* plane, px */
   std::vector<int64_t> key;           /* captions with equal keys share plane pixels (one caprot batch) */
   int rpx = 0; double r = 0;          /* find_angle's r: rounded (the window) and as hypot(fw, fh) */
};

/* the rectangles the rule's earlier captions left on auto_ctx (:2376-2382), as a batch in chart px */
struct rect_batch_t { std::vector<double> x, y; std::vector<int64_t> off = std::vector<int64_t>(1, 0); std::vector<int32_t> ref; uint32_t col = 0; };

/* The atlases of one rule's captions.  The chart's rules are rendered ONCE into every window (build); the rule's dependency
 * levels then take turns on the same atlases: scan(level) runs find_angle for the level's captions, add_rects() draws the
 * rectangles they left (auto_ctx, :2376-2382) into every window they reach, where the later levels' scans see them. */
struct atlas_set_t
{
   struct atlas_t { smr_canvas_t *c = nullptr; size_t i0 = 0, i1 = 0; std::vector<double> ax, ay; };
   std::vector<scan_item_t> items;          /* sorted by window size, largest first */
   std::vector<int> level;                  /* per item */
   std::vector<atlas_t> atlases;

   ~atlas_set_t() { for (atlas_t &a : atlases) if (a.c) smr_hip_canvas_destroy(a.c); }

   int build(std::vector<scan_item_t> in, const std::vector<int> &lev)
   {
      std::vector<size_t> order(in.size());
      for (size_t k = 0; k < in.size(); k++)
      {
         order[k] = k;
         int steps = 0;
         in[k].rpx = 0;
         if (in[k].fw > 0 && in[k].fh > 0) smr_hip_caprot_steps(in[k].fw, in[k].fh, P.dpi, &steps, &in[k].rpx);
         in[k].r = hypot((double) in[k].fw, (double) in[k].fh);
      }
      std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return in[a].rpx > in[b].rpx; });
      for (size_t k : order) if (in[k].rpx > 0) { items.push_back(in[k]); level.push_back(lev.empty() ? 0 : lev[k]); }      /* an empty plane: angle 0 */
      for (size_t i0 = 0; i0 < items.size();)
      {
         /* one atlas: windows of similar size (a smaller window is limited to the atlas's columns, include/smr_hip.h) */
         const int maxsize = items[i0].rpx, pitch = (maxsize + 3) & ~3;
         size_t i1 = i0;
         while (i1 < items.size() && (int) (i1 - i0 + 1) * pitch <= 65536 && items[i1].rpx * 4 >= maxsize * 3) i1++;
         const size_t n = i1 - i0;
         std::vector<int32_t> size(n); std::vector<double> rp(n), cx(n), cy(n);
         for (size_t k = 0; k < n; k++) { const scan_item_t &it = items[i0 + k]; size[k] = it.rpx; rp[k] = it.r; cx[k] = it.cx; cy[k] = it.cy; }
         smr_windows_t w;
         w.n = (int64_t) n; w.size = size.data(); w.r_px = rp.data(); w.cx = cx.data(); w.cy = cy.data();
         atlases.emplace_back();
         atlas_t &a = atlases.back();
         a.i0 = i0; a.i1 = i1; a.ax.resize(n); a.ay.resize(n);
         if (smr_hip_windows_create(P.W, P.H, P.dpi, P.bg, &w, &a.c) != SMR_OK) return -1;
         for (const kept_rule_t &k : kept_)
         {
            const int rc = k.is_blit ? smr_hip_blit_windows(a.c, k.pixels.data(), k.pixels.size(), k.items.data(), (int) k.items.size())
                                     : smr_hip_draw_windows(a.c, &k.st, &k.b);
            if (rc != SMR_OK) return -1;
         }
         /* the work buffers of the replay (edge arenas, bins) are not needed while the atlas waits for its levels */
         if (smr_hip_canvas_trim(a.c) != SMR_OK) return -1;
         i0 = i1;
      }
      return 0;
   }

   /* dv (optional): per caption the raw per-angle differences of its scan (dv_diff before dv_weight), for mkarea */
   template <typename PlaneFn>
   int scan(int lv, PlaneFn plane_of, double phase, double weight, std::vector<double> &angle, std::vector<std::vector<double>> *dv = nullptr)
   {
      for (atlas_t &a : atlases)
      {
         std::map<std::vector<int64_t>, std::vector<size_t>> groups;
         for (size_t k = a.i0; k < a.i1; k++) if (level[k] == lv) groups[items[k].key].push_back(k);
         if (groups.empty()) continue;
         /* clears what lies beyond the chart's extents, too (rectangles of earlier levels included) */
         if (smr_hip_windows_anchors(a.c, a.ax.data(), a.ay.data()) != SMR_OK) return -1;
         for (auto &g : groups)
         {
            const std::vector<size_t> &idx = g.second;
            std::vector<uint32_t> plane; int w_px = 0, h_px = 0;
            if (plane_of(items[idx[0]].id, plane, &w_px, &h_px) != 0) continue;
            const size_t m = idx.size();
            std::vector<double> gx(m), gy(m), out(m);
            for (size_t k = 0; k < m; k++) { gx[k] = a.ax[idx[k] - a.i0]; gy[k] = a.ay[idx[k] - a.i0]; }
            dbuf_t dx, dy, dfg, dang, ddv;
            int steps = 0, rr = 0;
            smr_hip_caprot_steps(w_px, h_px, P.dpi, &steps, &rr);
            bool ok = dx.put(gx.data(), m * 8) == 0 && dy.put(gy.data(), m * 8) == 0 && dfg.put(plane.data(), plane.size() * 4) == 0 && dang.alloc(m * 8) == 0;
            if (ok && dv) ok = ddv.alloc(m * (size_t) steps * 8) == 0;
            if (ok)
            {
               smr_cap_batch_t b;
               memset(&b, 0, sizeof(b));
               b.d_cx = (const double *) dx.p; b.d_cy = (const double *) dy.p; b.n = (int64_t) m;
               b.fg_w = w_px; b.fg_h = h_px; b.fg_x = 0; b.phase_deg = phase; b.weight = weight; b.d_fg = (const uint32_t *) dfg.p;
               ok = smr_hip_caprot(a.c, &b, (double *) dang.p, dv ? (double *) ddv.p : nullptr) == SMR_OK;
            }
            if (ok) ok = smr_hip_memcpy_d2h(out.data(), dang.p, m * 8) == SMR_OK;
            if (ok && dv)
            {
               std::vector<double> all(m * (size_t) steps);
               ok = smr_hip_memcpy_d2h(all.data(), ddv.p, all.size() * 8) == SMR_OK;
               for (size_t k = 0; k < m && ok; k++) (*dv)[items[idx[k]].id].assign(all.begin() + k * steps, all.begin() + (k + 1) * steps);
            }
            if (!ok) return -1;
            for (size_t k = 0; k < m; k++) angle[items[idx[k]].id] = out[k];
         }
      }
      return 0;
   }

   int add_rects(const rect_batch_t &r)
   {
      if (r.ref.empty()) return 0;
      std::vector<double> rx(r.x), ry(r.y);
      if (rx.size() & 1) { rx.push_back(0); ry.push_back(0); }
      dbuf_t dx, dy, doff, dref;
      if (dx.put(rx.data(), rx.size() * 8) || dy.put(ry.data(), ry.size() * 8) || doff.put(r.off.data(), r.off.size() * 8) || dref.put(r.ref.data(), r.ref.size() * 4))
         return -1;
      smr_draw_style_t st;
      memset(&st, 0, sizeof(st));
      st.fill_used = 1; st.fill_col = r.col; st.direct = 1;
      smr_draw_batch_t b;
      memset(&b, 0, sizeof(b));
      b.d_x = (const double *) dx.p; b.d_y = (const double *) dy.p; b.n_nodes = (int64_t) rx.size();
      b.d_way_off = (const int64_t *) doff.p; b.d_ref = (const int32_t *) dref.p;
      b.n_ways = (int64_t) r.off.size() - 1; b.n_refs = (int64_t) r.ref.size();
      for (atlas_t &a : atlases)
         if (smr_hip_draw_windows(a.c, &st, &b) != SMR_OK || smr_hip_sync() != SMR_OK) return -1;
      return 0;
   }
};

/* cairo_show_text of every caption (cap_coord :2363-2388: translate to the anchor, rotate, move_to the aligned origin, show_text),
 * rasterised by the host's libcairo + FreeType into a small image per caption on the chart's own pixel grid and painted OVER
 * the device canvas in object order (smr_hip_canvas_blit): glyph outlines stay on the host, the pixels land on the device */
static int cap_draw_glyphs(const cap_rule_t *cap, const std::vector<double> &cx, const std::vector<double> &cy,
                           const std::vector<double> &angle, const std::vector<short> &final_pos)
{
   std::vector<uint32_t> pixels;
   std::vector<smr_blit_t> items;
   const double sc = (double) P.dpi / 72.0;
   const int W = (int) round(P.proj.w), Hh = (int) round(P.proj.h);
   auto flush = [&]() -> int
   {
      if (items.empty()) return SMR_OK;
      const int rc = smr_hip_canvas_blit(P.canvas, pixels.data(), pixels.size(), items.data(), (int) items.size());
      if (rc == SMR_OK) kept_blit(pixels, items);
      pixels.clear(); items.clear();
      return rc;
   };
   for (size_t i = 0; i < cap->items.size(); i++)
   {
      const cap_item_t &it = cap->items[i];
      if (it.size_mm == 0.0 || it.text.empty()) continue;
      const double a = angle[i], ca = cos(a), sa = sin(a);
      double ox, oy;
      pos_offset(final_pos[i], it.tw, it.hgt, cap->xoff, cap->yoff, &ox, &oy);
      /* the text's box about its origin (a font size of slack for overhangs), rotated, in chart pixels */
      const double pad = mm2pt(it.size_mm), bx = cap->fill_used ? fabs(cap->bgbox_scale - 1) * it.tw * .5 : 0, by = cap->fill_used ? fabs(cap->bgbox_scale - 1) * it.hgt * .5 : 0;
      const double ux[2] = { ox - pad - bx, ox + it.tw + pad + bx }, uy[2] = { oy - it.hgt - pad - by, oy + it.descent + pad + by };
      double mnx = 1e300, mny = 1e300, mxx = -1e300, mxy = -1e300;
      for (int k = 0; k < 4; k++)
      {
         const double X = (ca * ux[k & 1] - sa * uy[k >> 1]) * sc + cx[i], Y = (sa * ux[k & 1] + ca * uy[k >> 1]) * sc + cy[i];
         mnx = fmin(mnx, X); mxx = fmax(mxx, X); mny = fmin(mny, Y); mxy = fmax(mxy, Y);
      }
      const int ix0 = (int) fmax(floor(mnx), 0.0), iy0 = (int) fmax(floor(mny), 0.0);
      const int ix1 = (int) fmin(ceil(mxx), (double) W), iy1 = (int) fmin(ceil(mxy), (double) Hh);
      if (ix1 <= ix0 || iy1 <= iy0) continue;
      const int w = ix1 - ix0, h = iy1 - iy0;
      /* recorded in pt on a recording surface and replayed at dpi/72, as the reference's chart is: libcairo rasterises the glyphs
       * of a replayed recording differently from glyphs shown on an image directly (oracle/cairo_drive.c:cd_caption_direct
       * against cd_caption_recorded: 60 % of the text pixels differ, against none) */
      cr_rectangle_t ext = { 0, 0, P.proj.w * 72 / P.dpi, P.proj.h * 72 / P.dpi };
      void *rec = CR.recording_surface_create(0x3000, &ext);
      void *ctx = CR.create(rec);
      CR.translate(ctx, cx[i] * 72 / P.dpi, cy[i] * 72 / P.dpi);
      CR.rotate(ctx, a);
      cap_font(ctx, cap->font);
      CR.set_font_size(ctx, mm2pt(it.size_mm));
      CR.set_source_rgba(ctx, (double) ((it.col >> 16) & 0xff) / 255.0, (double) ((it.col >> 8) & 0xff) / 255.0, (double) (it.col & 0xff) / 255.0,
                         1.0 - (double) (((it.col & 0x7f000000u) >> 23) & 0xff) / 255.0);
      if (cap->fill_used)
      {
         /* cap_rect (:2188-2199): the box behind the text, scaled by bgbox_scale about the text's middle */
         const double bw = it.tw, bh = -it.hgt, ws = bw * cap->bgbox_scale, hs = bh * cap->bgbox_scale;
         CR.save(ctx);
         CR.translate(ctx, (bw - ws) * .5, it.descent + (bh - hs) * .5);
         CR.set_source_rgba(ctx, (double) ((it.fillcol >> 16) & 0xff) / 255.0, (double) ((it.fillcol >> 8) & 0xff) / 255.0, (double) (it.fillcol & 0xff) / 255.0,
                            1.0 - (double) (((it.fillcol & 0x7f000000u) >> 23) & 0xff) / 255.0);
         CR.rectangle(ctx, ox, oy, ws, hs);
         CR.fill(ctx);
         CR.restore(ctx);
         CR.set_source_rgba(ctx, (double) ((it.col >> 16) & 0xff) / 255.0, (double) ((it.col >> 8) & 0xff) / 255.0, (double) (it.col & 0xff) / 255.0,
                            1.0 - (double) (((it.col & 0x7f000000u) >> 23) & 0xff) / 255.0);
      }
      if (!cap->hide)
      {
         CR.move_to(ctx, ox, oy);
         CR.show_text(ctx, it.text.c_str());
      }
      CR.destroy(ctx);
      void *sfc = CR.image_surface_create(0 /* ARGB32 */, w, h);
      ctx = CR.create(sfc);
      CR.translate(ctx, -ix0, -iy0);
      CR.scale(ctx, sc, sc);
      CR.set_source_surface(ctx, rec, 0, 0);
      CR.paint(ctx);
      CR.destroy(ctx);
      CR.surface_destroy(rec);
      CR.surface_flush(sfc);
      const unsigned char *d = CR.image_surface_get_data(sfc);
      const int stride = CR.image_surface_get_stride(sfc);
      smr_blit_t b; b.x = ix0; b.y = iy0; b.w = w; b.h = h; b.offset = pixels.size();
      pixels.resize(pixels.size() + (size_t) w * h);
      for (int y = 0; y < h; y++) memcpy(&pixels[b.offset + (size_t) y * w], d + (size_t) y * stride, (size_t) w * 4);
      CR.surface_destroy(sfc);
      items.push_back(b);
      if (pixels.size() > ((size_t) 16 << 20)) { const int rc = flush(); if (rc != SMR_OK) return rc; }
   }
   return flush();
}

/* cap_geo_rect, src/smrules_cairo.c:2114-2176 (fontbox=1): the caption's box, NOT rotated, as a closed way of four new nodes in the
 * host's data; way and nodes carry the object's tags and smrender:fontbox=yes.  x, y: the text's origin on the page in pt */
static int cap_geo_rect(const cap_rule_t *cap, double x, double y, double width, double height, double desc, const smr_osm_obj_t *o)
{
   const double ws = width * cap->bgbox_scale, hs = height * cap->bgbox_scale;
   smr_osm_way_t *w = H.malloc_way((short) (o->tag_cnt + 2), 5);
   if (w == nullptr) return -1;
   H.osm_way_default(w);
   /* the object's tags behind generator=smrender, or over it when the object has that tag itself */
   const int i0 = H.match_attr(o, "generator", "smrender") == -1;
   memcpy(&w->obj.otag[i0], o->otag, sizeof(*o->otag) * (size_t) o->tag_cnt);
   w->obj.tag_cnt = (short) (o->tag_cnt + i0);
   static char k_fontbox[] = "smrender:fontbox", v_yes[] = "yes";
   struct smr_otag &t = w->obj.otag[w->obj.tag_cnt];
   t.k.buf = k_fontbox; t.k.len = (int) strlen(k_fontbox); t.v.buf = v_yes; t.v.len = (int) strlen(v_yes);
   w->obj.tag_cnt++;
   for (int i = 0; i < 4; i++)
   {
      smr_osm_node_t *n = H.malloc_node(w->obj.tag_cnt);
      if (n == nullptr) return -1;
      H.osm_node_default(n);
      memcpy(n->obj.otag, w->obj.otag, sizeof(*w->obj.otag) * (size_t) w->obj.tag_cnt);
      n->obj.tag_cnt = w->obj.tag_cnt;
      double x0 = x + (width - ws) * .5, y0 = y + (height - hs) * .5 + desc;
      if (i == 1 || i == 2) x0 += ws;
      if (i == 2 || i == 3) y0 += hs;
      host_pxf2geo(x0 * P.dpi / 72, y0 * P.dpi / 72, &n->lon, &n->lat);      /* rdata_unit_px(.., U_PT) */
      H.put_object(&n->obj);
      w->ref[i] = n->obj.id;
   }
   w->ref[4] = w->ref[0];
   H.put_object(&w->obj);
   return 0;
}

/* dv_mkarea, src/smrules_cairo.c:1772-1801 (angle=auto;mkarea=1): the scan's weighted, quantised differences as a closed way around
 * the caption's anchor, one new node per trial angle at r * dv_quant from it (r in pt taken as px, as the reference does), each
 * tagged smrender:autorot:angle="<bearing>;<percent>".  dv: the raw differences of smr_hip_caprot; dv_weight (:1813-1821) and
 * dv_quantize (:1868-1890) are applied here as find_angle does before it calls dv_mkarea (:1997-2000) */
static int cap_mkarea(const cap_rule_t *cap, const cap_item_t &, double cx_px, double cy_px, double r, const std::vector<double> &dv)
{
   const int cnt = (int) dv.size();
   std::vector<double> diff(dv), ang((size_t) cnt), quant((size_t) cnt);
   const double phase = cap->phase * M_PI / 180.0;
   double a = 0;
   for (int i = 0; i < cnt; i++, a += 2 * M_PI / cnt)       /* dv_sample's running sum (:1842) */
   {
      ang[(size_t) i] = a;
      if (cap->weight < 0.0) diff[(size_t) i] = 1.0 - diff[(size_t) i];
      diff[(size_t) i] *= 1.0 - (1.0 - fabs(cap->weight)) * (1.0 - cos(a * 2.0 + phase)) / 2.0;
   }
   double mn = 1.0, mx = 0.0;
   for (double d : diff) { if (d > mx) mx = d; if (d < mn) mn = d; }
   for (int i = 0; i < cnt; i++)
   {
      quant[(size_t) i] = (diff[(size_t) i] - mn) / (mx - mn);
      if (std::isnan(quant[(size_t) i])) quant[(size_t) i] = 1.0;
   }
   smr_osm_way_t *w = H.malloc_way(1, cnt + 1);
   if (w == nullptr) return -1;
   H.osm_way_default(w);
   static char k_angle[] = "smrender:autorot:angle";
   for (int i = 0; i < cnt; i++)
   {
      smr_osm_node_t *n = H.malloc_node(2);
      if (n == nullptr) return -1;
      H.osm_node_default(n);
      w->ref[i] = n->obj.id;
      host_pxf2geo(cx_px + r * quant[(size_t) i] * cos(2 * M_PI - ang[(size_t) i]), cy_px + r * quant[(size_t) i] * sin(2 * M_PI - ang[(size_t) i]), &n->lon, &n->lat);
      char buf[32];
      snprintf(buf, sizeof(buf), "%.1f;%.1f", fmod2((M_PI_2 - ang[(size_t) i]) * 180.0 / M_PI, 360), quant[(size_t) i] * 100);
      char *v = strdup(buf);                               /* lives as long as the node, like the reference's */
      n->obj.otag[1].k.buf = k_angle; n->obj.otag[1].k.len = (int) strlen(k_angle);
      n->obj.otag[1].v.buf = v; n->obj.otag[1].v.len = (int) strlen(v);
      H.put_object(&n->obj);
   }
   w->ref[cnt] = w->ref[0];
   H.put_object(&w->obj);
   return 0;
}

static int cap_fini(smr_smrule_t *r)
{
   cap_rule_t *cap = (cap_rule_t *) r->data;
   if (cap == nullptr) return 0;
   int ret = 0;
   const size_t n_all = cap->items.size();
   std::vector<double> angle(n_all, 0.0), cx(n_all), cy(n_all), reach(n_all, 1.0);
   std::vector<int> fw(n_all), fh(n_all), level;
   std::vector<short> final_pos(n_all);
   std::vector<std::vector<double>> dvraw(cap->mkarea ? n_all : 0);
   int n_levels = 0;
   smr_canvas_t *autoc = nullptr;
   size_t n_groups = 0, n_atlases = 0;
   const double sc = (double) P.dpi / 72.0;
   /* the reference's cut-outs (windows of the chart rendered again at each anchor's sub-pixel offset) unless told to crop */
   const bool replay = cutout_replay();
   atlas_set_t atl;
   if (n_all)
   {
      /* all anchors -> px in one projection call */
      const size_t np = (n_all + 1) & ~(size_t) 1;
      std::vector<double> lat(np), lon(np);
      for (size_t k = 0; k < n_all; k++) { lat[k] = cap->items[k].lat; lon[k] = cap->items[k].lon; }
      if (np > n_all) { lat[n_all] = lat[n_all - 1]; lon[n_all] = lon[n_all - 1]; }
      dbuf_t dlat, dlon, dx, dy;
      bool ok = dlat.put(lat.data(), np * 8) == 0 && dlon.put(lon.data(), np * 8) == 0 && dx.alloc(np * 8) == 0 && dy.alloc(np * 8) == 0;
      if (ok) ok = smr_hip_project((const double *) dlat.p, (const double *) dlon.p, (int64_t) np, (double *) dx.p, (double *) dy.p,
                                   nullptr, nullptr, nullptr) == SMR_OK;
      std::vector<double> hx(np), hy(np);
      if (ok) ok = smr_hip_memcpy_d2h(hx.data(), dx.p, np * 8) == SMR_OK && smr_hip_memcpy_d2h(hy.data(), dy.p, np * 8) == SMR_OK;
      if (!ok) { plog(SMR_LOG_ERR, "cap: %s", smr_hip_last_error()); ret = -1; }
      for (size_t k = 0; k < n_all && ok; k++)
      {
         const cap_item_t &it = cap->items[k];
         cx[k] = hx[k]; cy[k] = hy[k];
         fw[k] = (int) round(it.pw * sc); fh[k] = (int) round(it.ph * sc);
         int steps = 0, rpx = 0;
         if (fw[k] > 0 && fh[k] > 0) smr_hip_caprot_steps(fw[k], fh[k], P.dpi, &steps, &rpx);
         /* what a caption can see (half the cut-out's diagonal) or paint (its rectangle): both within r_px */
         reach[k] = std::max(1.0, (double) rpx);
      }
      if (!cap->autorot)
         for (size_t k = 0; k < n_all; k++) { angle[k] = cap->items[k].fixed_a; final_pos[k] = cap->items[k].pos; }
      if (ret == 0 && cap->autorot) cap_levels(cx, cy, reach, level, n_levels);
      if (ret == 0 && cap->autorot && !replay && smr_hip_canvas_clone(P.canvas, &autoc) != SMR_OK) { plog(SMR_LOG_ERR, "cap: %s", smr_hip_last_error()); ret = -1; }
   }
   for (int lv = 0; lv < n_levels && ret == 0; lv++)
   {
      if (replay)
      {
         auto plane_of = [&](size_t id, std::vector<uint32_t> &plane, int *w_px, int *h_px)
         {
            const cap_item_t &it = cap->items[id];
            return make_plane(it.pw, it.ph, it.px, it.col, P.dpi, plane, w_px, h_px);
         };
         if (lv == 0)
         {
            std::vector<scan_item_t> todo(n_all);
            for (size_t i = 0; i < n_all; i++)
            {
               const cap_item_t &it = cap->items[i];
               scan_item_t &q = todo[i];
               q.id = i; q.cx = cx[i]; q.cy = cy[i]; q.fw = fw[i]; q.fh = fh[i]; q.key = { it.pw, it.ph, it.px, (int64_t) it.col };
            }
            if (atl.build(todo, level) != 0) { plog(SMR_LOG_ERR, "cap: %s", smr_hip_last_error()); ret = -1; }
            n_atlases = atl.atlases.size();
         }
         if (ret == 0 && atl.scan(lv, plane_of, cap->phase, cap->weight, angle, cap->mkarea ? &dvraw : nullptr) != 0) { plog(SMR_LOG_ERR, "cap: %s", smr_hip_last_error()); ret = -1; }
      }
      else
      {
         /* one GPU batch per distinct caption plane of this level */
         std::map<std::vector<int64_t>, std::vector<size_t>> groups;
         for (size_t i = 0; i < n_all; i++)
         {
            if (level[i] != lv) continue;
            const cap_item_t &it = cap->items[i];
            groups[{ it.pw, it.ph, it.px, (int64_t) it.col }].push_back(i);
         }
         n_groups += groups.size();
         for (auto &g : groups)
         {
            const std::vector<size_t> &idx = g.second;
            const cap_item_t &first = cap->items[idx[0]];
            std::vector<uint32_t> plane; int w_px, h_px;
            if (make_plane(first.pw, first.ph, first.px, first.col, P.dpi, plane, &w_px, &h_px) != 0) continue;   /* empty plane: angle 0 */
            const size_t n = idx.size();
            std::vector<double> gx(n), gy(n), out(n);
            for (size_t k = 0; k < n; k++) { gx[k] = cx[idx[k]]; gy[k] = cy[idx[k]]; }
            dbuf_t dx, dy, dfg, dang, ddv;
            int steps = 0, rr = 0;
            smr_hip_caprot_steps(w_px, h_px, P.dpi, &steps, &rr);
            bool ok = dx.put(gx.data(), n * 8) == 0 && dy.put(gy.data(), n * 8) == 0 && dfg.put(plane.data(), plane.size() * 4) == 0 &&
                      dang.alloc(n * 8) == 0;
            if (ok && cap->mkarea) ok = ddv.alloc(n * (size_t) steps * 8) == 0;
            if (ok)
            {
               smr_cap_batch_t b;
               memset(&b, 0, sizeof(b));
               b.d_cx = (const double *) dx.p; b.d_cy = (const double *) dy.p; b.n = (int64_t) n;
               b.fg_w = w_px; b.fg_h = h_px; b.fg_x = 0; b.col = first.col; b.phase_deg = cap->phase; b.weight = cap->weight;
               b.d_fg = (const uint32_t *) dfg.p;
               ok = smr_hip_caprot(autoc, &b, (double *) dang.p, cap->mkarea ? (double *) ddv.p : nullptr) == SMR_OK;
            }
            if (ok) ok = smr_hip_memcpy_d2h(out.data(), dang.p, n * 8) == SMR_OK;
            if (ok && cap->mkarea)
            {
               std::vector<double> all(n * (size_t) steps);
               ok = smr_hip_memcpy_d2h(all.data(), ddv.p, all.size() * 8) == SMR_OK;
               for (size_t k = 0; k < n && ok; k++) dvraw[idx[k]].assign(all.begin() + k * steps, all.begin() + (k + 1) * steps);
            }
            if (!ok) { plog(SMR_LOG_ERR, "cap: %s", smr_hip_last_error()); ret = -1; break; }
            for (size_t k = 0; k < n; k++) angle[idx[k]] = out[k];
         }
      }
      if (ret != 0) break;
      /* flip (:2284-2292), then this level's rectangles onto the private background (:2376-2382) */
      std::vector<double> rx, ry;
      std::vector<int64_t> roff(1, 0);
      std::vector<int32_t> rref;
      for (size_t i = 0; i < n_all; i++)
      {
         if (level[i] != lv) continue;
         const cap_item_t &it = cap->items[i];
         double a = angle[i];
         short pos = it.pos;
         if (a > M_PI_2 && a < 3 * M_PI_2)
         {
            a -= M_PI;
            if (pos & SMR_POS_E) pos = (short) ((cap->pos & 0xfff0) | SMR_POS_W);
         }
         angle[i] = a; final_pos[i] = pos;
         if (lv + 1 == n_levels) continue;                   /* nobody is left to see the last level's rectangles */
         double ox, oy;
         pos_offset(pos, it.tw, it.hgt, cap->xoff, cap->yoff, &ox, &oy);
         const double wr = it.tw + cap->xoff, hr = -it.ascent, ca = cos(a), sa = sin(a);
         const double xpt = cx[i] * 72 / P.dpi, ypt = cy[i] * 72 / P.dpi;
         const double ux[4] = { ox, ox + wr, ox + wr, ox }, uy[4] = { oy, oy, oy + hr, oy + hr };
         const int32_t base = (int32_t) rx.size();
         for (int k = 0; k < 4; k++)
         {
            const double X = (ca * ux[k] + -sa * uy[k]) + xpt, Y = (sa * ux[k] + ca * uy[k]) + ypt;
            rx.push_back(X * P.dpi / 72); ry.push_back(Y * P.dpi / 72);
            rref.push_back(base + k);
         }
         rref.push_back(base);
         roff.push_back((int64_t) rref.size());
      }
      if (!rref.empty() && replay)
      {
         /* this level's rectangles into every window they reach: the later levels' scans see them */
         rect_batch_t rb;
         rb.x = rx; rb.y = ry; rb.off = roff; rb.ref = rref;
         rb.col = cap->auto_col;                               /* auto_ctx's source: the first object's colour (:2459) */
         if (atl.add_rects(rb) != 0) { plog(SMR_LOG_ERR, "cap: %s", smr_hip_last_error()); ret = -1; }
      }
      else if (!rref.empty())
      {
         if (rx.size() & 1) { rx.push_back(0); ry.push_back(0); }
         dbuf_t dx, dy, doff, dref;
         bool ok = dx.put(rx.data(), rx.size() * 8) == 0 && dy.put(ry.data(), ry.size() * 8) == 0 &&
                   doff.put(roff.data(), roff.size() * 8) == 0 && dref.put(rref.data(), rref.size() * 4) == 0;
         if (ok)
         {
            smr_draw_style_t st;
            memset(&st, 0, sizeof(st));
            st.fill_used = 1; st.fill_col = cap->auto_col; st.direct = 1;       /* auto_ctx's source: the first object's colour (:2459) */
            smr_draw_batch_t b;
            memset(&b, 0, sizeof(b));
            b.d_x = (const double *) dx.p; b.d_y = (const double *) dy.p; b.n_nodes = (int64_t) rx.size();
            b.d_way_off = (const int64_t *) doff.p; b.d_ref = (const int32_t *) dref.p;
            b.n_ways = (int64_t) roff.size() - 1; b.n_refs = (int64_t) rref.size();
            ok = smr_hip_draw(autoc, &st, &b) == SMR_OK && smr_hip_sync() == SMR_OK;
         }
         if (!ok) { plog(SMR_LOG_ERR, "cap: %s", smr_hip_last_error()); ret = -1; }
      }
   }
   if (autoc) smr_hip_canvas_destroy(autoc);
   if (ret == 0 && cap->autorot)
      for (size_t i = 0; i < n_all; i++)
      {
         cap_item_t &it = cap->items[i];
         char val[32];
         snprintf(val, sizeof(val), "%.1f", angle[i] * 180.0 / M_PI);
         add_tag(it.o, cap->key, "autoangle", val);
         add_tag(it.o, cap->key, "autoalign", (final_pos[i] & SMR_POS_E) ? "east" : (final_pos[i] & SMR_POS_W) ? "west" : "center");
      }
   /* what the rule adds to the host's data, caption by caption in object order as cap_coord does (find_angle's area first) */
   if (ret == 0 && (cap->mkarea || cap->fontbox))
      for (size_t i = 0; i < n_all; i++)
      {
         const cap_item_t &it = cap->items[i];
         if (cap->mkarea && !dvraw[i].empty()) cap_mkarea(cap, it, cx[i], cy[i], hypot((double) fw[i], (double) fh[i]) * 72 / P.dpi, dvraw[i]);
         if (cap->fontbox)
         {
            double ox, oy;
            pos_offset(final_pos[i], it.tw, it.hgt, cap->xoff, cap->yoff, &ox, &oy);
            cap_geo_rect(cap, cx[i] * 72 / P.dpi + ox, cy[i] * 72 / P.dpi + oy, it.tw, -it.hgt, it.descent, it.o);
         }
      }
   if (ret == 0 && (!cap->hide || cap->fill_used) && n_all && cap_draw_glyphs(cap, cx, cy, angle, final_pos) != SMR_OK)
   {
      plog(SMR_LOG_ERR, "cap: %s", smr_hip_last_error());
      ret = -1;
   }
   if (cap->autorot && replay)
      plog(SMR_LOG_INFO, "cap: %zu captions scanned on the GPU in %d dependency level(s): %zu atlas(es) of cut-outs, each with the %zu rule(s) drawn so far%s",
           n_all, n_levels, n_atlases, kept_.size(), cap->hide ? "" : "; glyphs composited onto the device canvas");
   else if (cap->autorot)
      plog(SMR_LOG_INFO, "cap: %zu captions scanned on the GPU in %d dependency level(s), %zu batch(es), crops of the canvas%s", n_all, n_levels, n_groups,
           cap->hide ? "" : ", glyphs composited onto the device canvas");
   else plog(SMR_LOG_INFO, "cap: %zu captions at their given angles%s", n_all, cap->hide ? "" : ", glyphs composited onto the device canvas");
   CR.destroy(cap->ctx);
   CR.surface_destroy(cap->sfc);
   delete cap;
   r->data = nullptr;
   return ret;
}

/* ---------------------------------------------------------------------------------------- */
/* img: symbols on nodes (act_img_ini/_main/_fini, src/smrules_cairo.c:2512-2871)                    */
/* ---------------------------------------------------------------------------------------- */
/* The symbol itself is decoded / rendered by the host's libraries, as in the reference (PNG: libcairo; SVG: librsvg into a
 * recording surface); every placement is recorded with the reference's call sequence (img_place :2754-2826), replayed at
 * dpi/72 into an image on the chart's pixel grid and painted OVER the device canvas in object order (smr_hip_canvas_blit).
 * angle=auto is find_angle with the symbol as the plane (:2767-2794): one smr_hip_caprot batch for all nodes of the rule
 * (img has no auto_ctx: its symbols do not see each other, :2790 scans sfc_).  img on ways (img_fill :2742-2751) is an
 * image fill of the closed ways on the device (draw_finish, smr_hip_draw_pattern); JPEG files are decoded by gdk-pixbuf
 * (the reference has a decoder of its own, src/cairo_jpg.c).  Not here: SVG and rotated patterns, the rule stays built-in. */
struct rsvg_api_t
{
   void *lib = nullptr, *gobj = nullptr;
   void *(*handle_new_from_file)(const char *, void **) = nullptr;
   void (*handle_set_dpi)(void *, double) = nullptr;
   int (*handle_get_intrinsic_size_in_pixels)(void *, double *, double *) = nullptr;
   void (*handle_get_dimensions)(void *, void *) = nullptr;
   int (*handle_render_document)(void *, void *, const void *, void **) = nullptr;
   int (*handle_render_cairo)(void *, void *) = nullptr;
   void (*object_unref)(void *) = nullptr;
};
static rsvg_api_t RS;

static int rsvg_load(void)
{
   if (RS.lib) return 0;
   const char *cand[] = { getenv("SMR_HIP_LIBRSVG"), "librsvg-2.so.2", nullptr };
   for (int i = 0; i < 3 && !RS.lib; i++)
      if (cand[i] && *cand[i]) RS.lib = dlopen(cand[i], RTLD_NOW | RTLD_GLOBAL);
   if (!RS.lib) { plog(SMR_LOG_WARN, "img: librsvg-2.so.2 cannot be loaded: %s", dlerror()); return -1; }
   *(void **) (&RS.handle_new_from_file) = dlsym(RS.lib, "rsvg_handle_new_from_file");
   *(void **) (&RS.handle_set_dpi) = dlsym(RS.lib, "rsvg_handle_set_dpi");
   *(void **) (&RS.handle_get_intrinsic_size_in_pixels) = dlsym(RS.lib, "rsvg_handle_get_intrinsic_size_in_pixels");
   *(void **) (&RS.handle_get_dimensions) = dlsym(RS.lib, "rsvg_handle_get_dimensions");
   *(void **) (&RS.handle_render_document) = dlsym(RS.lib, "rsvg_handle_render_document");
   *(void **) (&RS.handle_render_cairo) = dlsym(RS.lib, "rsvg_handle_render_cairo");
   *(void **) (&RS.object_unref) = dlsym(RTLD_DEFAULT, "g_object_unref");
   if (!RS.handle_new_from_file || !RS.handle_set_dpi || !RS.object_unref || (!RS.handle_render_document && !RS.handle_render_cairo) ||
       (!RS.handle_get_intrinsic_size_in_pixels && !RS.handle_get_dimensions))
   {
      plog(SMR_LOG_WARN, "img: librsvg lacks the entry points the reference uses");
      dlclose(RS.lib); RS.lib = nullptr;
      return -1;
   }
   return 0;
}

struct img_item_t { double lat, lon, a; };

struct img_rule_t
{
   void *img = nullptr;                /* the scaled symbol: an image surface (PNG) or a recording surface (SVG) */
   bool vector = false;
   double w = 0, h = 0, scale = 1, trans = 0, angle = 0, phase = 0, weight = 1;
   bool autorot = false;
   const char *akey = nullptr, *alignkey = nullptr;
   std::vector<img_item_t> items;
   draw_rule_t *ways = nullptr;        /* img on ways (img_fill :2742-2751): a rule of nothing but an image fill of its closed ways */
   ~img_rule_t() { delete ways; }
};

/* a JPEG through gdk-pixbuf (the reference decodes it with libjpeg itself, src/cairo_jpg.c:243-290: the same library underneath;
 * jpeglib.h is not in this image, so its structs are out of reach) into a cairo ARGB32 image surface */
static void *jpeg_surface(const char *name)
{
   static void *lib = nullptr;
   static void *(*new_from_file)(const char *, void **) = nullptr;
   static int (*get_i[4])(const void *) = { nullptr, nullptr, nullptr, nullptr };
   static const unsigned char *(*get_pixels)(const void *) = nullptr;
   static void (*unref)(void *) = nullptr;
   if (!lib)
   {
      const char *cand[] = { getenv("SMR_HIP_LIBGDKPIXBUF"), "libgdk_pixbuf-2.0.so.0", nullptr };
      for (int i = 0; i < 3 && !lib; i++) if (cand[i] && *cand[i]) lib = dlopen(cand[i], RTLD_NOW | RTLD_GLOBAL);
      if (!lib) { plog(SMR_LOG_WARN, "img: libgdk_pixbuf-2.0.so.0 (JPEG decoding) cannot be loaded: %s", dlerror()); return nullptr; }
      *(void **) (&new_from_file) = dlsym(lib, "gdk_pixbuf_new_from_file");
      const char *names[4] = { "gdk_pixbuf_get_width", "gdk_pixbuf_get_height", "gdk_pixbuf_get_rowstride", "gdk_pixbuf_get_n_channels" };
      for (int i = 0; i < 4; i++) *(void **) (&get_i[i]) = dlsym(lib, names[i]);
      *(void **) (&get_pixels) = dlsym(lib, "gdk_pixbuf_get_pixels");
      *(void **) (&unref) = dlsym(RTLD_DEFAULT, "g_object_unref");
      if (!new_from_file || !get_i[0] || !get_i[1] || !get_i[2] || !get_i[3] || !get_pixels || !unref) { lib = nullptr; return nullptr; }
   }
   void *pb = new_from_file(name, nullptr);
   if (!pb) return nullptr;
   const int w = get_i[0](pb), h = get_i[1](pb), rs = get_i[2](pb), nc = get_i[3](pb);
   const unsigned char *px = get_pixels(pb);
   void *sfc = CR.image_surface_create(0 /* ARGB32 */, w, h);
   if (CR.surface_status(sfc) != 0 || nc < 3) { unref(pb); return sfc; }
   CR.surface_flush(sfc);
   unsigned char *d = CR.image_surface_get_data(sfc);
   const int stride = CR.image_surface_get_stride(sfc);
   for (int y = 0; y < h; y++)
   {
      uint32_t *row = (uint32_t *) (d + (size_t) y * stride);
      const unsigned char *q = px + (size_t) y * rs;
      for (int x = 0; x < w; x++, q += nc) row[x] = 0xff000000u | ((uint32_t) q[0] << 16) | ((uint32_t) q[1] << 8) | q[2];      /* a JPEG has no alpha */
   }
   CR.surface_mark_dirty(sfc);
   unref(pb);
   return sfc;
}

/* the part of img_ini (:2512-2674) that makes the image: file=, scale= (times the chart's img_scale), transparency=; SVG through
 * librsvg into a recording surface, PNG / JPEG into an ARGB32 image of w x h pixels.  0, or what img_ini is to return */
static int img_load(smr_smrule_t *r, img_rule_t *g)
{
   char *name;
   if ((name = H.get_param("file", nullptr, r->act)) == nullptr)
   {
      plog(SMR_LOG_WARN, "parameter 'file' missing");
      return -1;
   }
   if (cairo_load() != 0) return -1;
   const size_t nl = strlen(name);
   if (H.get_param("scale", &g->scale, r->act) == nullptr) g->scale = 1;
   g->scale *= H.rdata_get()->img_scale;
   if (H.get_param("transparency", &g->trans, r->act) != nullptr) g->trans = fmin(fmax(g->trans, 0.0), 1.0);
   void *ctx;
   if (nl >= 4 && !strcasecmp(name + nl - 4, ".svg"))
   {
      if (rsvg_load() != 0) return 1;
      void *rh = RS.handle_new_from_file(name, nullptr);
      if (rh == nullptr) { plog(SMR_LOG_ERR, "error opening file %s", name); return -1; }
      RS.handle_set_dpi(rh, (double) P.dpi);
      double dw = 32, dh = 32;
      if (RS.handle_get_intrinsic_size_in_pixels && RS.handle_render_document)
      {
         if (!RS.handle_get_intrinsic_size_in_pixels(rh, &dw, &dh))
         {
            plog(SMR_LOG_WARN, "rsvg_handle_get_intrinsic_size_in_pixels() failed");
            dw = dh = 32;
         }
      }
      else
      {
         struct { int width, height; double em, ex; } dd;
         RS.handle_get_dimensions(rh, &dd);
         dw = dd.width; dh = dd.height;
      }
      g->w = dw * g->scale; g->h = dh * g->scale;
      cr_rectangle_t rect = { 0, 0, g->w, g->h };
      g->img = CR.recording_surface_create(0x3000, &rect);
      g->vector = true;
      ctx = CR.create(g->img);
      CR.scale(ctx, g->scale, g->scale);
      int ok;
      if (RS.handle_get_intrinsic_size_in_pixels && RS.handle_render_document)
      {
         const double vp[4] = { 0, 0, dw, dh };
         ok = RS.handle_render_document(rh, ctx, vp, nullptr);
      }
      else ok = RS.handle_render_cairo(rh, ctx);
      RS.object_unref(rh);
      if (!ok)
      {
         plog(SMR_LOG_ERR, "rendering %s failed", name);
         CR.destroy(ctx); CR.surface_destroy(g->img); g->img = nullptr;
         return -1;
      }
   }
   else
   {
      const bool jpg = nl >= 4 && !strcasecmp(name + nl - 4, ".jpg");
      void *sfc = jpg ? jpeg_surface(name) : CR.image_surface_create_from_png(name);
      if (sfc == nullptr || CR.surface_status(sfc) != 0)
      {
         plog(SMR_LOG_ERR, "cannot open file %s (cairo status %d)", name, sfc ? CR.surface_status(sfc) : -1);
         if (sfc) CR.surface_destroy(sfc);
         return -1;
      }
      g->w = CR.image_surface_get_width(sfc) * g->scale; g->h = CR.image_surface_get_height(sfc) * g->scale;
      g->img = CR.image_surface_create(0 /* ARGB32 */, (int) g->w, (int) g->h);
      if (CR.surface_status(g->img) != 0)
      {
         plog(SMR_LOG_ERR, "cannot create the %g x %g image of %s", g->w, g->h, name);
         CR.surface_destroy(sfc); CR.surface_destroy(g->img); g->img = nullptr;
         return -1;
      }
      ctx = CR.create(g->img);
      CR.scale(ctx, g->scale, g->scale);
      CR.set_source_surface(ctx, sfc, 0, 0);
      CR.paint(ctx);
      CR.surface_destroy(sfc);
   }
   /* add transparency (:2598-2601, :2653-2656) */
   CR.set_source_rgba(ctx, 0, 0, 0, g->trans);
   CR.set_operator(ctx, 9 /* CAIRO_OPERATOR_DEST_OUT */);
   CR.paint(ctx);
   CR.destroy(ctx);
   return 0;
}

/* the image of a way rule as the pixels of its pattern (img_ini :2689-2704: cairo_pattern_create_for_surface, matrix
 * scale(1 / PX2PT_SCALE) . rotate(angle), CAIRO_EXTEND_REPEAT): at angle 0 one image pixel per chart pixel from the chart's origin,
 * which smr_hip_draw_pattern lays out; a rotated or a vector (SVG) pattern is left to the built-in action */
static int rule_pattern(smr_smrule_t *r, const char *who, std::vector<uint32_t> &pix, int *w, int *h)
{
   img_rule_t g;
   const int e = img_load(r, &g);
   if (e != 0) return e;
   double angle = 0;
   const char *val = H.get_param("angle", &angle, r->act);
   if (val != nullptr && (!strcasecmp(val, "auto") || !strcasecmp(val, "majoraxis"))) angle = 0;      /* "ignoring angle=auto" (:2685-2689); 720 degrees */
   angle = fmod(angle, 360.0);
   int ret = 0;
   if (g.vector) { plog(SMR_LOG_ERR, "%s: an SVG as the pattern of a way fill is not implemented on the GPU path; use the built-in %s", who, who); ret = 1; }
   else if (angle != 0.0) { plog(SMR_LOG_ERR, "%s: a rotated pattern (angle=%g) is not implemented on the GPU path; use the built-in %s", who, angle, who); ret = 1; }
   else if ((int) g.w < 1 || (int) g.h < 1) { plog(SMR_LOG_ERR, "%s: the scaled image is empty", who); ret = -1; }
   else
   {
      *w = (int) g.w; *h = (int) g.h;
      CR.surface_flush(g.img);
      const unsigned char *d = CR.image_surface_get_data(g.img);
      const int stride = CR.image_surface_get_stride(g.img);
      pix.resize((size_t) *w * *h);
      for (int y = 0; y < *h; y++) memcpy(&pix[(size_t) y * *w], d + (size_t) y * stride, (size_t) *w * 4);
   }
   if (g.img) CR.surface_destroy(g.img);
   g.img = nullptr;
   return ret;
}

static int img_ini(smr_smrule_t *r)
{
   int e;
   host_resolve();
   if ((e = plugin_setup()) != 0) return e;
   if (r->oo->type != SMR_OSM_NODE && r->oo->type != SMR_OSM_WAY)
   {
      plog(SMR_LOG_WARN, "img() only applicable to nodes and ways");
      return -1;
   }
   if (r->oo->type == SMR_OSM_WAY)
   {
      /* a pattern fill of the rule's closed ways (img_ini :2683-2704, act_img_main -> img_fill :2742-2751) */
      img_rule_t *gw = new img_rule_t();
      gw->ways = new draw_rule_t();
      memset(&gw->ways->st, 0, sizeof(gw->ways->st));
      gw->ways->st.ignore_open = 1;                       /* img_fill passes open ways over */
      gw->ways->way_off.push_back(0);
      const int e2 = rule_pattern(r, "img", gw->ways->pat, &gw->ways->pat_w, &gw->ways->pat_h);
      if (e2 != 0) { delete gw; return e2; }
      r->data = gw;
      return 0;
   }
   img_rule_t *g = new img_rule_t();
   if ((e = img_load(r, g)) != 0) { delete g; return e; }

   /* parse_auto_rot (src/smrparse.c:1033-1074) */
   char *val;
   if ((val = H.get_param("angle", &g->angle, r->act)) != nullptr)
   {
      if (!strcasecmp("auto", val))
      {
         g->autorot = true;
         if (H.get_param("weight", &g->weight, r->act) == nullptr) g->weight = 1.0;
         g->weight = fmin(fmax(g->weight, -1.0), 1.0);
         (void) H.get_param("phase", &g->phase, r->act);
      }
      /* only cap_way looks at MAJORAXIS (:2431): img_place turns by 360 - 720 degrees, a full turn */
      else if (!strcasecmp("majoraxis", val)) g->angle = SMR_MAJORAXIS;
      else g->angle = fmod(g->angle, 360.0);
   }
   g->akey = H.get_param("anglekey", nullptr, r->act);
   if (g->akey != nullptr && g->autorot)
   {
      plog(SMR_LOG_NOTICE, "ignoring angle=auto");
      g->autorot = false; g->angle = 0;
   }
   g->alignkey = H.get_param("alignkey", nullptr, r->act);
   if (g->autorot && !cutout_replay() && (P.band_y0 != 0 || P.band_h != P.H))
   {
      plog(SMR_LOG_ERR, "img: angle=auto on a row band needs the neighbours' rows (see cap)");
      CR.surface_destroy(g->img); delete g;
      return 1;
   }
   if (!H.match_attr && (g->akey || g->alignkey))
   {
      plog(SMR_LOG_ERR, "img: host symbol match_attr not found");
      CR.surface_destroy(g->img); delete g;
      return -1;
   }
   r->data = g;
   return 0;
}

static int img_main(smr_smrule_t *r, smr_osm_obj_t *o)
{
   img_rule_t *g = (img_rule_t *) r->data;
   if (g == nullptr) { plog(SMR_LOG_WARN, "r->data == NULL"); return 1; }
   if (g->ways != nullptr) return o->type == SMR_OSM_WAY ? draw_way(g->ways, (const smr_osm_way_t *) o) : 1;      /* act_img_main :2835-2836 */
   if (o->type != SMR_OSM_NODE) return 1;
   const smr_osm_node_t *n = (const smr_osm_node_t *) o;
   img_item_t it; it.lat = n->lat; it.lon = n->lon; it.a = 0;
   if (!g->autorot)
   {
      /* img_place :2796-2817 */
      int m = -1;
      if (g->akey != nullptr && (m = H.match_attr(o, g->akey, nullptr)) >= 0)
         it.a = strtod(std::string(o->otag[m].v.buf, (size_t) o->otag[m].v.len).c_str(), nullptr) * M_PI / 180.0;
      if (g->alignkey != nullptr && (m = H.match_attr(o, g->alignkey, nullptr)) >= 0)
         if (alignment_str(std::string(o->otag[m].v.buf, (size_t) o->otag[m].v.len).c_str()) & SMR_POS_W) it.a += M_PI_2;
      it.a += (360 - g->angle) * M_PI / 180.0;
   }
   g->items.push_back(it);
   return 0;
}

/* anchors -> chart pixels through the device projection (geo2pxf) */
static int project_anchors(const std::vector<double> &lat_in, const std::vector<double> &lon_in, std::vector<double> &hx, std::vector<double> &hy)
{
   const size_t n = lat_in.size(), np = (n + 1) & ~(size_t) 1;
   std::vector<double> lat(lat_in), lon(lon_in);
   lat.resize(np, n ? lat_in[n - 1] : 0.0); lon.resize(np, n ? lon_in[n - 1] : 0.0);
   hx.assign(np, 0.0); hy.assign(np, 0.0);
   if (n == 0) return 0;
   dbuf_t dlat, dlon, dx, dy;
   bool ok = dlat.put(lat.data(), np * 8) == 0 && dlon.put(lon.data(), np * 8) == 0 && dx.alloc(np * 8) == 0 && dy.alloc(np * 8) == 0;
   if (ok) ok = smr_hip_project((const double *) dlat.p, (const double *) dlon.p, (int64_t) np, (double *) dx.p, (double *) dy.p,
                                nullptr, nullptr, nullptr) == SMR_OK;
   if (ok) ok = smr_hip_memcpy_d2h(hx.data(), dx.p, np * 8) == SMR_OK && smr_hip_memcpy_d2h(hy.data(), dy.p, np * 8) == SMR_OK;
   return ok ? 0 : -1;
}

static int img_fini(smr_smrule_t *r)
{
   img_rule_t *g = (img_rule_t *) r->data;
   if (g == nullptr) { plog(SMR_LOG_WARN, "r->data == NULL"); return 1; }
   if (g->ways != nullptr)
   {
      const int e = draw_finish(g->ways);
      delete g;
      r->data = nullptr;
      return e;
   }
   int ret = 0;
   const size_t n = g->items.size();
   std::vector<double> lat(n), lon(n), cx, cy;
   for (size_t k = 0; k < n; k++) { lat[k] = g->items[k].lat; lon[k] = g->items[k].lon; }
   if (n && project_anchors(lat, lon, cx, cy) != 0) { plog(SMR_LOG_ERR, "img: %s", smr_hip_last_error()); ret = -1; }
   const int iw = (int) g->w, ih = (int) g->h;
   if (ret == 0 && n && g->autorot && iw > 0 && ih > 0)
   {
      /* the plane of find_angle: the symbol as an ARGB32 image (:2773-2788) */
      void *fg = CR.image_surface_create(0, iw, ih);
      void *fx = CR.create(fg);
      CR.set_source_surface(fx, g->img, 0, 0);
      CR.paint(fx);
      CR.destroy(fx);
      CR.surface_flush(fg);
      std::vector<uint32_t> plane((size_t) iw * ih);
      const unsigned char *d = CR.image_surface_get_data(fg);
      const int stride = CR.image_surface_get_stride(fg);
      for (int y = 0; y < ih; y++) memcpy(&plane[(size_t) y * iw], d + (size_t) y * stride, (size_t) iw * 4);
      CR.surface_destroy(fg);
      if (cutout_replay())
      {
         std::vector<scan_item_t> todo(n);
         for (size_t k = 0; k < n; k++) { todo[k].id = k; todo[k].cx = cx[k]; todo[k].cy = cy[k]; todo[k].fw = iw; todo[k].fh = ih; todo[k].key = { 0 }; }
         auto plane_of = [&](size_t, std::vector<uint32_t> &pl, int *w_px, int *h_px) { pl = plane; *w_px = iw; *h_px = ih; return 0; };
         std::vector<double> out(n, 0.0);
         atlas_set_t atl;
         if (atl.build(todo, std::vector<int>()) != 0 || atl.scan(0, plane_of, g->phase, g->weight, out) != 0) { plog(SMR_LOG_ERR, "img: %s", smr_hip_last_error()); ret = -1; }
         else for (size_t k = 0; k < n; k++) g->items[k].a = out[k];
      }
      else
      {
         dbuf_t dx, dy, dfg, dang;
         std::vector<double> out(n);
         bool ok = dx.put(cx.data(), n * 8) == 0 && dy.put(cy.data(), n * 8) == 0 && dfg.put(plane.data(), plane.size() * 4) == 0 && dang.alloc(n * 8) == 0;
         if (ok)
         {
            smr_cap_batch_t b;
            memset(&b, 0, sizeof(b));
            b.d_cx = (const double *) dx.p; b.d_cy = (const double *) dy.p; b.n = (int64_t) n;
            b.fg_w = iw; b.fg_h = ih; b.phase_deg = g->phase; b.weight = g->weight; b.d_fg = (const uint32_t *) dfg.p;
            ok = smr_hip_caprot(P.canvas, &b, (double *) dang.p, nullptr) == SMR_OK;
         }
         if (ok) ok = smr_hip_memcpy_d2h(out.data(), dang.p, n * 8) == SMR_OK;
         if (!ok) { plog(SMR_LOG_ERR, "img: %s", smr_hip_last_error()); ret = -1; }
         for (size_t k = 0; k < n && ok; k++) g->items[k].a = out[k];
      }
   }
   if (ret == 0 && n)
   {
      std::vector<uint32_t> pixels;
      std::vector<smr_blit_t> items;
      const double sc = (double) P.dpi / 72.0;
      const int W = (int) round(P.proj.w), Hh = (int) round(P.proj.h);
      auto flush = [&]() -> int
      {
         if (items.empty()) return SMR_OK;
         const int rc = smr_hip_canvas_blit(P.canvas, pixels.data(), pixels.size(), items.data(), (int) items.size());
         if (rc == SMR_OK) kept_blit(pixels, items);
         pixels.clear(); items.clear();
         return rc;
      };
      for (size_t i = 0; i < n && ret == 0; i++)
      {
         const double a = g->items[i].a, ca = cos(a), sa = sin(a);
         double mnx = 1e300, mny = 1e300, mxx = -1e300, mxy = -1e300;
         for (int k = 0; k < 4; k++)
         {
            const double ux = (k & 1 ? 0.5 : -0.5) * g->w, uy = (k >> 1 ? 0.5 : -0.5) * g->h;
            const double X = ca * ux - sa * uy + cx[i], Y = sa * ux + ca * uy + cy[i];
            mnx = fmin(mnx, X); mxx = fmax(mxx, X); mny = fmin(mny, Y); mxy = fmax(mxy, Y);
         }
         const int ix0 = (int) fmax(floor(mnx) - 2, 0.0), iy0 = (int) fmax(floor(mny) - 2, 0.0);
         const int ix1 = (int) fmin(ceil(mxx) + 2, (double) W), iy1 = (int) fmin(ceil(mxy) + 2, (double) Hh);
         if (ix1 <= ix0 || iy1 <= iy0) continue;
         const int w = ix1 - ix0, h = iy1 - iy0;
         /* img_ini :2680-2683 (the context on the chart, in px) + img_place :2763-2823, recorded in pt like the chart */
         cr_rectangle_t ext = { 0, 0, P.proj.w * 72 / P.dpi, P.proj.h * 72 / P.dpi };
         void *rec = CR.recording_surface_create(0x3000, &ext);
         void *ctx = CR.create(rec);
         CR.scale(ctx, 72.0 / P.dpi, 72.0 / P.dpi);
         CR.translate(ctx, cx[i], cy[i]);
         CR.rotate(ctx, a);
         CR.set_source_surface(ctx, g->img, g->w / -2.0, g->h / -2.0);
         CR.paint(ctx);
         CR.destroy(ctx);
         void *sfc = CR.image_surface_create(0, w, h);
         ctx = CR.create(sfc);
         CR.translate(ctx, -ix0, -iy0);
         CR.scale(ctx, sc, sc);
         CR.set_source_surface(ctx, rec, 0, 0);
         CR.paint(ctx);
         CR.destroy(ctx);
         CR.surface_destroy(rec);
         CR.surface_flush(sfc);
         const unsigned char *d = CR.image_surface_get_data(sfc);
         const int stride = CR.image_surface_get_stride(sfc);
         smr_blit_t b; b.x = ix0; b.y = iy0; b.w = w; b.h = h; b.offset = pixels.size();
         pixels.resize(pixels.size() + (size_t) w * h);
         for (int y = 0; y < h; y++) memcpy(&pixels[b.offset + (size_t) y * w], d + (size_t) y * stride, (size_t) w * 4);
         CR.surface_destroy(sfc);
         items.push_back(b);
         if (pixels.size() > ((size_t) 16 << 20) && flush() != SMR_OK) ret = -1;
      }
      if (ret == 0 && flush() != SMR_OK) ret = -1;
      if (ret != 0) plog(SMR_LOG_ERR, "img: %s", smr_hip_last_error());
   }
   plog(SMR_LOG_INFO, "img: %zu symbols (%s, %.1f x %.1f px%s) composited onto the device canvas", n, g->vector ? "SVG" : "PNG", g->w, g->h,
        g->autorot ? ", angle=auto scanned on the GPU" : "");
   CR.surface_destroy(g->img);
   delete g;
   r->data = nullptr;
   return ret;
}

/* ---------------------------------------------------------------------------------------- */
/* hipsave: the device canvas -> PNG (stands in for save_main_image, src/smrules_cairo.c:388-412)   */
/* ---------------------------------------------------------------------------------------- */
static int hipsave_ini(smr_smrule_t *r)
{
   int e;
   host_resolve();
   if ((e = plugin_setup()) != 0) return e;
   if (H.get_param("file", nullptr, r->act) == nullptr)
   {
      plog(SMR_LOG_WARN, "hipsave: parameter 'file' missing");
      return 1;
   }
   return 0;
}

static int hipsave_main(smr_smrule_t *, smr_osm_obj_t *) { return 0; }

/* the text gen_kap_header (src/smkap.c:143-167) puts before "OST/1": chart name and scale, size, the corners' coordinates
 * (pxf2geo through smr_hip_unproject).  The first line names this library where the reference prints its PACKAGE_STRING */
static int kap_header(std::string &out)
{
   const struct smr_rdata *rd = H.rdata_get();
   const double px[7] = { 0, rd->w, 0, 0, rd->w, rd->w, 0 }, py[7] = { 0, rd->h, rd->h, 0, 0, rd->h, rd->h };
   double lon[7], lat[7];
   void *d = nullptr;
   if (smr_hip_malloc(&d, 4 * 7 * sizeof(double)) != SMR_OK) return -1;
   double *dd = (double *) d;
   int e = smr_hip_memcpy_h2d(dd, px, sizeof(px));
   if (e == SMR_OK) e = smr_hip_memcpy_h2d(dd + 7, py, sizeof(py));
   if (e == SMR_OK) e = smr_hip_unproject(dd, dd + 7, 7, dd + 14, dd + 21);
   if (e == SMR_OK) e = smr_hip_memcpy_d2h(lon, dd + 14, sizeof(lon));
   if (e == SMR_OK) e = smr_hip_memcpy_d2h(lat, dd + 21, sizeof(lat));
   smr_hip_free(d);
   if (e != SMR_OK) return -1;
   std::string title = rd->title ? rd->title : "";
   for (char &ch : title) if (ch == ',') ch = ' ';
   const double mppx = rd->mean_lat_len / rd->w * 1852.0 * 60.0;
   char buf[512];
   out = "! generated by smrender (libsmr_hip)\r\n";
   snprintf(buf, sizeof(buf), "VER/2.0\r\nBSB/NA=%s %d\r\n    NU=,RA=%d,%d,DU=%d\r\n", title.c_str(), (int) round(rd->scale), (int) round(rd->w), (int) round(rd->h), rd->dpi);
   out += buf;
   snprintf(buf, sizeof(buf), "KNP/SC=%d,GD=WGS84,PR=MERCATOR\r\n    PP=%f,PI=UNKNOWN,SP=UNKNOWN,SK=0.0,TA=90.0\r\n", (int) round(rd->scale), rd->mean_lat);
   out += buf;
   snprintf(buf, sizeof(buf), "    UN=METRES,SD=MLWS,DX=%.2f,DY=%.2f\r\n", mppx, mppx);
   out += buf;
   for (int k = 0; k < 3; k++) { snprintf(buf, sizeof(buf), "REF/%d,%.0f,%.0f,%.8f,%.8f\r\n", k + 1, px[k], py[k], lat[k], lon[k]); out += buf; }
   for (int k = 0; k < 4; k++) { snprintf(buf, sizeof(buf), "PLY/%d,%.8f,%.8f\r\n", k + 1, lat[3 + k], lon[3 + k]); out += buf; }
   out += "DTM/0.0,0.0\r\nCPH/0.0\r\n";
   return 0;
}

static int hipsave_fini(smr_smrule_t *r)
{
   const char *file = H.get_param("file", nullptr, r->act);
   if (!file || !P.canvas) return -1;
   if (smr_hip_sync() != SMR_OK) return -1;
   const size_t fl = strlen(file);
   if (fl > 4 && strcasecmp(file + fl - 4, ".kap") == 0)
   {
      /* a KAP chart (what `smrender -k` writes through save_kap, src/smkap.c:196-245): colour reduction, indices and rows on the device (kap.hip) */
      std::string header;
      smr_kap_stats_t ks;
      if (kap_header(header) != 0 || smr_hip_canvas_write_kap(P.canvas, header.c_str(), file, &ks) != SMR_OK)
      {
         plog(SMR_LOG_ERR, "hipsave: %s", smr_hip_last_error());
         return -1;
      }
      plog(SMR_LOG_NOTICE, "hipsave: KAP chart of %dx%d px, %d colours, written to %s: %.1f MB in %.0f ms", P.W, P.H, ks.n_colors, file, ks.file_bytes / 1e6,
           ks.reduce_ms + ks.index_ms + ks.rows_ms + ks.write_ms);
      return 0;
   }
   /* un-premultiply, filter choice, filtering and deflate on the device, the compressed bytes cross PCIe (png.hip);
    * SMR_HIP_PNG_LEVEL=-1, 0..9: zlib at that level on the host's cores instead (smaller files, ten times the time) */
   smr_png_stats_t st;
   const int level = env_int("SMR_HIP_PNG_LEVEL", SMR_PNG_DEVICE_DEFLATE);
   int e = smr_hip_canvas_write_png(P.canvas, 0, 0, P.W, P.band_h, file, level, env_int("SMR_HIP_PNG_THREADS", 0), &st);
   if (e != SMR_OK) { plog(SMR_LOG_ERR, "hipsave: %s", smr_hip_last_error()); return -1; }
   plog(SMR_LOG_NOTICE, "hipsave: %dx%d px (rows %d..%d of the chart, %s) written to %s: %.1f MB in %.0f ms (%s, %d host threads)", P.W, P.band_h,
        P.band_y0, P.band_y0 + P.band_h - 1, st.n_opaque ? "RGB" : "RGBA", file, st.bytes_written / 1e6, st.total_ms,
        level == SMR_PNG_DEVICE_DEFLATE ? "deflate on the device" : "zlib", st.threads);
   return e;
}

/* ---- the exported symbols (include/smr_hip.h): void* in the public header, host types here ---- */
#define SMR_ACTION(name) \
   extern "C" int act_##name##_ini(void *r) { return name##_ini((smr_smrule_t *) r); } \
   extern "C" int act_##name##_main(void *r, void *o) { return name##_main((smr_smrule_t *) r, (smr_osm_obj_t *) o); } \
   extern "C" int act_##name##_fini(void *r) { return name##_fini((smr_smrule_t *) r); }
SMR_ACTION(draw)
SMR_ACTION(cap)
SMR_ACTION(img)
SMR_ACTION(hipsave)
