/* smr_abi.h -- our own re-declaration of the part of smrender's plugin ABI that libsmr_hip
 * touches (64-bit Linux).  Nothing here is copied code: it restates the memory layout of
 *   smrule_t            libsmrender/smrender.h:73-78
 *   action_t, fparam_t  libsmrender/smaction.h:67-73, :97-123 (fields before stag[] only)
 *   bstring_t, otag     libsmrender/bstring.h:31-35, libsmrender/osm_inplace.h:49-52
 *   osm_obj_t (2024 layout), osm_node_t, osm_way_t, osm_rel_t, rmember   osm_inplace.h:71-118
 *   struct rdata        src/rdata.h:99-158 (with bbox_t :76-79, dstats :81-92, MAX_ITER 64)
 *   struct specialTag, struct stag   libsmrender/smaction.h:75-89 (the array behind action_t: objects.hip reads it)
 * and lists the host symbols the plugin calls.
 */
#ifndef SMR_ABI_H
#define SMR_ABI_H
#include <stdint.h>
#include <time.h>
#include <regex.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SMR_OSM_NA, SMR_OSM_NODE, SMR_OSM_WAY, SMR_OSM_REL };          /* osm_inplace.h:38 */
enum { SMR_POS_N = 1, SMR_POS_S = 2, SMR_POS_E = 4, SMR_POS_W = 8, SMR_POS_UC = 16 };   /* smrender_dev.h:55-71 */
enum { SMR_LOG_EMERG = 0, SMR_LOG_ERR = 3, SMR_LOG_WARN = 4, SMR_LOG_NOTICE = 5, SMR_LOG_INFO = 6, SMR_LOG_DEBUG = 7 };   /* syslog levels */

typedef struct { int len; char *buf; } smr_bstring_t;
struct smr_otag { smr_bstring_t k, v; };

typedef struct
{
   int8_t type, vis;
   short tag_cnt;
   int ver, cs, uid;
   int64_t id;
   time_t tim;
   struct smr_otag *otag;
} smr_osm_obj_t;

typedef struct { smr_osm_obj_t obj; double lat, lon; } smr_osm_node_t;
typedef struct { smr_osm_obj_t obj; int ref_cnt; int64_t *ref; } smr_osm_way_t;
struct smr_rmember { short type, role; int64_t id; };
typedef struct { smr_osm_obj_t obj; short mem_cnt; struct smr_rmember *mem; } smr_osm_rel_t;

typedef struct { char *attr, *val; double dval; int conv_error; } smr_fparam_t;

typedef struct smr_action
{
   void *ini, *main, *fini;      /* unions of function pointer / void* */
   void *libhandle;
   char *func_name, *parm;
   smr_fparam_t **fp;
   short flags, finished, way_type, tag_cnt;
   /* struct stag stag[] follows; never touched by the plugin */
} smr_action_t;

typedef struct { smr_osm_obj_t *oo; void *data; smr_action_t *act; } smr_smrule_t;

/* how a rule's match tag compares (parse_matchtype, src/smrparse.c:65-139, fills it): libsmrender/smaction.h:33-39, :52-54 */
enum { SMR_SPECIAL_DIRECT = 0, SMR_SPECIAL_REGEX = 1, SMR_SPECIAL_GT = 2, SMR_SPECIAL_LT = 3, SMR_SPECIAL_MASK = 0xff,
       SMR_SPECIAL_NOT = 0x4000, SMR_SPECIAL_INVERT = 0x8000 };
enum { SMR_ACTION_OPEN_WAY = 1 << 5, SMR_ACTION_CLOSED_WAY = 1 << 6 };
struct smr_special_tag { short type; union { regex_t re; double val; }; };
struct smr_stag { struct smr_special_tag stk, stv; };

struct smr_coord { double lat, lon; };
typedef struct { struct smr_coord ll, ru; } smr_bbox_t;
struct smr_dstats
{
   smr_bbox_t bb;
   long cnt[4];
   int64_t min_id[4], max_id[4];
   int id_bits[4];
   int64_t id_mask[4];
   const void *lo_addr, *hi_addr;
   int ver_cnt;
   int ver[64];
};
struct smr_rdata
{
   void *rules;
   int need_index;
   void *index;
   smr_bbox_t bb;
   int polygon_window;
   struct smr_coord pw[4];
   double rot, wc, hc, mean_lat, mean_lat_len, transversal_lat, mean_lon, lath, lath_len, w, h, pgw, pgh;
   int dpi;
   double scale;
   struct smr_dstats ds;
   char *cmdline, *title;
   int flags;
   int64_t id_off;
   double img_scale;
   int proj;
   int nthreads;
};

/* Host symbols (libsmrender.so / the -rdynamic executable) are NOT linked: plugin.cpp looks
 * them up with dlsym(RTLD_DEFAULT, ...) at the first _ini, so the library also loads in
 * processes that are not smrender (tests, bench).  Prototypes it expects:
 *   char *get_param(const char*, double*, const action_t*)      libsmrender/smutil.c:500
 *   int get_param_bool(const char*, const action_t*)            libsmrender/smutil.c:591
 *   void *get_object(int, int64_t)                              libsmrender/smutil.c:236
 *   int match_attr(const osm_obj_t*, const char*, const char*)  libsmrender/smutil.c:362
 *   int realloc_tags(osm_obj_t*, int)                           libsmrender/osm_func.c
 *   int log_msg(int, const char*, ...)                          libsmrender/smlog.c:258
 *   struct rdata *rdata_get(void)                               src/rdata.c:41-47
 *   int parse_color(const char*), parse_style(const char*)      src/smrparse.c:222, :270
 *   int parse_alignment(const action_t*)                        src/smrparse.c:704
 *   int poly_area(const osm_way_t*, struct coord*, double*)     src/smfunc.c:377
 *   double rdata_square_mm(void), rdata_square_nm(void)         src/rdata.c
 */

#ifdef __cplusplus
}
#endif
#endif
