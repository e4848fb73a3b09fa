/* hostsim.c -- TEST INFRASTRUCTURE.  A miniature smrender host for exercising the plugin
 * boundary of libsmr_hip.so (act_draw_ and act_cap_ entry points) without the smrender executable.
 *
 * It is OUR code, compiled against the reference's own headers where they lie
 * (/root/reference/libsmrender/smrender.h, /root/reference/src/rdata.h: read at build time,
 * never copied) and linked with oracle/_ref/libsmrender_ref.so, which oracle/Makefile builds
 * from the reference's untouched libsmrender sources (bstring.c bxtree.c lists.c osm_func.c
 * smlog.c smutil.c; no config.h needed: every use is behind #ifdef HAVE_CONFIG_H; the two settings
 * bxtree.h takes from outside come on the compiler's command line as smrender's configure sets them,
 * configure.ac:21-22: bx_hash_t = int64_t, BX_RES = 4 -- object ids are 64 bits as in production).
 * So get_param, get_object, put_object, match_attr, realloc_tags, malloc_node/way, log_msg
 * are the REAL host functions and smrule_t/action_t/osm_*_t have the REAL layout; the
 * _Static_asserts below pin smrender_amd/csrc/smr_abi.h to them.
 *
 * What lives in the smrender executable (not buildable here: it needs the autoconf-generated
 * config.h and cairo headers) is restated minimally: rdata_get (src/rdata.c:41-47),
 * parse_color ('#' forms + a few names; src/smrparse.c:222-258), parse_style (:270-283),
 * the rule set-up of init_rule/parse_fparam (:365-467, :616-664) and the
 * _ini / traverse / _main / _fini protocol of apply_smrules (src/smcore.c:424-491).
 */
#define _GNU_SOURCE
#include <errno.h>
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "smrender.h"
#include "rdata.h"
#include "../smrender_amd/csrc/smr_abi.h"

/* ---- ABI pin: our restated layouts vs the reference's headers ---- */
#define SAME(T1, T2) _Static_assert(sizeof(T1) == sizeof(T2), "sizeof " #T1)
#define SAMEF(T1, T2, f) _Static_assert(offsetof(T1, f) == offsetof(T2, f), "offsetof " #T1 "." #f)
SAME(smr_osm_obj_t, osm_obj_t); SAMEF(smr_osm_obj_t, osm_obj_t, type); SAMEF(smr_osm_obj_t, osm_obj_t, vis);
SAMEF(smr_osm_obj_t, osm_obj_t, tag_cnt); SAMEF(smr_osm_obj_t, osm_obj_t, ver); SAMEF(smr_osm_obj_t, osm_obj_t, id);
SAMEF(smr_osm_obj_t, osm_obj_t, tim); SAMEF(smr_osm_obj_t, osm_obj_t, otag);
SAME(smr_osm_node_t, osm_node_t); SAMEF(smr_osm_node_t, osm_node_t, lat); SAMEF(smr_osm_node_t, osm_node_t, lon);
SAME(smr_osm_way_t, osm_way_t); SAMEF(smr_osm_way_t, osm_way_t, ref_cnt); SAMEF(smr_osm_way_t, osm_way_t, ref);
SAME(smr_osm_rel_t, osm_rel_t); SAMEF(smr_osm_rel_t, osm_rel_t, mem_cnt); SAMEF(smr_osm_rel_t, osm_rel_t, mem);
SAME(struct smr_rmember, struct rmember); SAMEF(struct smr_rmember, struct rmember, id); SAMEF(struct smr_rmember, struct rmember, role);
SAME(struct smr_otag, struct otag); SAME(smr_bstring_t, bstring_t); SAMEF(smr_bstring_t, bstring_t, buf);
SAME(smr_fparam_t, fparam_t); SAMEF(smr_fparam_t, fparam_t, val); SAMEF(smr_fparam_t, fparam_t, dval); SAMEF(smr_fparam_t, fparam_t, conv_error);
SAME(smr_smrule_t, smrule_t); SAMEF(smr_smrule_t, smrule_t, data); SAMEF(smr_smrule_t, smrule_t, act);
_Static_assert(sizeof(smr_action_t) == offsetof(action_t, stag), "action_t head");
SAMEF(smr_action_t, action_t, fp); SAMEF(smr_action_t, action_t, flags); SAMEF(smr_action_t, action_t, tag_cnt);
SAMEF(smr_action_t, action_t, libhandle); SAMEF(smr_action_t, action_t, parm);
SAME(struct smr_special_tag, struct specialTag); SAMEF(struct smr_special_tag, struct specialTag, val); SAMEF(struct smr_special_tag, struct specialTag, re);
SAME(struct smr_stag, struct stag); SAMEF(struct smr_stag, struct stag, stv);
_Static_assert(SMR_SPECIAL_REGEX == SPECIAL_REGEX && SMR_SPECIAL_GT == SPECIAL_GT && SMR_SPECIAL_LT == SPECIAL_LT && SMR_SPECIAL_MASK == SPECIAL_MASK
               && SMR_SPECIAL_NOT == SPECIAL_NOT && SMR_SPECIAL_INVERT == SPECIAL_INVERT && SMR_SPECIAL_DIRECT == SPECIAL_DIRECT, "match kinds");
_Static_assert(SMR_ACTION_OPEN_WAY == ACTION_OPEN_WAY && SMR_ACTION_CLOSED_WAY == ACTION_CLOSED_WAY, "way flags");
SAME(struct smr_rdata, struct rdata); SAMEF(struct smr_rdata, struct rdata, bb); SAMEF(struct smr_rdata, struct rdata, polygon_window);
SAMEF(struct smr_rdata, struct rdata, wc); SAMEF(struct smr_rdata, struct rdata, lath); SAMEF(struct smr_rdata, struct rdata, lath_len);
SAMEF(struct smr_rdata, struct rdata, w); SAMEF(struct smr_rdata, struct rdata, h); SAMEF(struct smr_rdata, struct rdata, dpi);
SAMEF(struct smr_rdata, struct rdata, proj); SAMEF(struct smr_rdata, struct rdata, ds); SAMEF(struct smr_rdata, struct rdata, mean_lat);
_Static_assert((int) SMR_OSM_NODE == (int) OSM_NODE && (int) SMR_OSM_WAY == (int) OSM_WAY && (int) SMR_OSM_REL == (int) OSM_REL, "object type enum");

int hs_abi_check(void) { return 0; }      /* compiling this file is the check */

/* ---- what the executable exports ---- */
static struct rdata rd_;
struct rdata *rdata_get(void) { return &rd_; }

static unsigned bgcolor_ = 0xffffff;

int parse_color(const char *s)
{
   static const struct { const char *n; int c; } names[] = {
      { "black", 0x000000 }, { "white", 0xffffff }, { "red", 0xff0000 }, { "green", 0x00ff00 }, { "blue", 0x0000ff },
      { "yellow", 0xffff00 }, { "grey", 0xbebebe }, { "gray", 0xbebebe }, { NULL, 0 } };
   if (s == NULL) return -1;
   if (*s == '#')
   {
      s++;
      int l = (int) strlen(s);
      if (l != 6 && l != 8) return 0;
      errno = 0;
      long long c = strtoll(s, NULL, 16);
      if (errno) return 0;
      return (int) (c & 0x7fffffff);
   }
   if (!strcmp(s, "bgcolor")) return (int) bgcolor_;
   for (int i = 0; names[i].n != NULL; i++)
      if (!strcmp(s, names[i].n)) return names[i].c;
   log_msg(LOG_WARN, "unknown color %s, defaulting to black", s);
   return 0;
}

int parse_style(const char *s)
{
   static const char *st[] = { "solid", "dashed", "dotted", "transparent", "pipe", "rounddot" };   /* enum order of src/smrender_dev.h */
   if (s == NULL) return 0;
   for (int i = 0; i < 6; i++) if (!strcmp(s, st[i])) return i;
   return 0;
}

/* chart geometry: the fields init_rendering_window/init_bbox_mll leave in struct rdata */
void hs_set_rdata(double lon0, double w, double h, double wc, double lath, double lath_len, double lat_ll, double lon_ru,
                  double lat_ru, int dpi, unsigned bgcolor)
{
   memset(&rd_, 0, sizeof(rd_));
   rd_.bb.ll.lon = lon0; rd_.bb.ll.lat = lat_ll; rd_.bb.ru.lon = lon_ru; rd_.bb.ru.lat = lat_ru;
   rd_.w = rd_.pgw = w; rd_.h = rd_.pgh = h; rd_.wc = wc; rd_.hc = lat_ru - lat_ll;
   rd_.lath = lath; rd_.lath_len = lath_len; rd_.dpi = dpi;
   rd_.mean_lat = (lat_ll + lat_ru) / 2; rd_.mean_lon = (lon0 + lon_ru) / 2;
   rd_.mean_lat_len = wc * cos(rd_.mean_lat * M_PI / 180);      /* init_bbox_mll, src/smrender.c:285: wc = mean_lat_len / cos(mean_lat) */
   rd_.proj = PROJ_MERC; rd_.img_scale = 1;
   bgcolor_ = bgcolor;
}

/* is_closed_poly (src/smcoast.c:115-125), poly_area (src/smfunc.c:377-432) and the chart's area in both units
 * (src/rdata.c:452-461): what act_cap's closed-way branch asks the executable for (cap_way, src/smrules_cairo.c:2395-2444).
 * Restated: the shoelace sum over (lon * cos(lat), lat), the centroid from the same terms, nautical square miles
 * (60 nm per degree: 3600 / 2 = 1800).  tests/test_plugin_host.py checks it against an independent shoelace in numpy. */
int is_closed_poly(const osm_way_t *w)
{
   return w->ref_cnt >= 4 && w->ref[0] == w->ref[w->ref_cnt - 1];
}

int poly_area(const osm_way_t *w, struct coord *center, double *area)
{
   if (center == NULL && area == NULL) return 0;
   if (!is_closed_poly(w)) return -1;
   const osm_node_t *a, *b = (const osm_node_t *) get_object(OSM_NODE, w->ref[0]);
   if (b == NULL) return -1;
   double sum = 0, clat = 0, clon = 0;
   for (int i = 1; i < w->ref_cnt; i++)
   {
      a = b;
      if ((b = (const osm_node_t *) get_object(OSM_NODE, w->ref[i])) == NULL) return -1;
      const double xa = a->lon * cos(a->lat * M_PI / 180), xb = b->lon * cos(b->lat * M_PI / 180);
      const double f = xa * b->lat - xb * a->lat;
      clon += (xa + xb) * f;
      clat += (a->lat + b->lat) * f;
      sum += f;
   }
   clat /= 3.0 * sum;
   clon /= 3.0 * sum * cos(clat * M_PI / 180);
   if (center != NULL) { center->lat = clat; center->lon = clon; }
   if (area != NULL) *area = sum * 1800.0;
   return 0;
}

/* test access: centroid and area of a way in the object tree */
int hs_poly_area(long long way_id, double *out)
{
   const osm_way_t *w = (const osm_way_t *) get_object(OSM_WAY, way_id);
   struct coord c; double ar;
   if (w == NULL || poly_area(w, &c, &ar)) return -1;
   out[0] = c.lat; out[1] = c.lon; out[2] = ar;
   return 0;
}

double rdata_square_mm(void) { return (rd_.w * 25.4 / rd_.dpi) * (rd_.h * 25.4 / rd_.dpi); }
double rdata_square_nm(void) { return rd_.mean_lat_len * rd_.hc * 3600; }

/* page_way(), src/smcoast.c:105-108: the way init_corner_brg (:804-841) builds from the corners of the
 * bounding box, refs = NW, SW, SE, NE, NW (counterclockwise on the chart) */
static osm_way_t *page_way_;
const osm_way_t *page_way(void) { return page_way_; }

int hs_make_page_way(void)
{
   const double n = rd_.bb.ru.lat, s = rd_.bb.ll.lat, e = rd_.bb.ru.lon, w = rd_.bb.ll.lon;
   const double lat[4] = { n, s, s, n }, lon[4] = { w, w, e, e };       /* NW, SW, SE, NE */
   osm_way_t *pw = malloc_way(2, 5);
   osm_way_default(pw);
   for (int k = 3; k >= 0; k--)           /* the reference creates the nodes NE first; ids only matter for uniqueness */
   {
      osm_node_t *nd = malloc_node(2);
      osm_node_default(nd);
      nd->lat = lat[k]; nd->lon = lon[k];
      put_object(&nd->obj);
      pw->ref[k] = nd->obj.id;
   }
   pw->ref[4] = pw->ref[0];
   pw->ref_cnt = 5;
   put_object(&pw->obj);
   page_way_ = pw;
   return 0;
}

/* ---- rules: what init_rule + parse_fparam build ---- */
static char *next_tok(char **p, const char *delim, char *found)
{
   char *s = *p;
   if (s == NULL || *s == '\0') return NULL;
   char *e = s + strcspn(s, delim);
   *found = *e;
   if (*e) { *e = '\0'; *p = e + 1; } else *p = e;
   return s;
}

smrule_t *hs_rule_new(int type, const char *func, const char *params)
{
   smrule_t *r = calloc(1, sizeof(*r));
   action_t *a = calloc(1, sizeof(*a));
   osm_node_t *oo = calloc(1, sizeof(osm_storage_t));
   oo->obj.type = (int8_t) type;
   oo->obj.id = 0x7f000001;
   r->oo = &oo->obj; r->act = a;
   a->func_name = strdup(func);
   a->parm = strdup(params ? params : "");
   char *buf = strdup(a->parm), *p = buf, c, *s;
   int n = 0;
   fparam_t **fp = calloc(1, sizeof(*fp));
   while ((s = next_tok(&p, "=;", &c)) != NULL)
   {
      fp = realloc(fp, sizeof(*fp) * (n + 2));
      fp[n] = calloc(1, sizeof(**fp));
      fp[n + 1] = NULL;
      fp[n]->attr = s;
      if (c == '=' && (fp[n]->val = next_tok(&p, ";", &c)) != NULL)
      {
         char *end;
         errno = 0;
         fp[n]->dval = strtod(fp[n]->val, &end);
         fp[n]->conv_error = end == fp[n]->val ? EDOM : errno;
      }
      n++;
   }
   a->fp = fp;
   return r;
}

/* ---- objects, stored in the real bxtree through the real put_object ---- */
static void set_tags(osm_obj_t *o, const char *tags)
{
   char *buf = strdup(tags ? tags : ""), *p = buf, c, *k;
   int n = 1;      /* tag 0 is "generator=smrender", set by osm_*_default (libsmrender/osm_func.c) */
   while ((k = next_tok(&p, "=", &c)) != NULL && n < o->tag_cnt)
   {
      char *v = next_tok(&p, ";", &c);
      o->otag[n].k.buf = k; o->otag[n].k.len = (int) strlen(k);
      o->otag[n].v.buf = v ? v : ""; o->otag[n].v.len = v ? (int) strlen(v) : 0;
      n++;
   }
   o->tag_cnt = (short) n;
}

static int count_tags(const char *tags)
{
   int n = 0;
   if (tags == NULL || *tags == '\0') return 0;
   for (n = 1; *tags; tags++) if (*tags == ';') n++;
   return n;
}

int hs_node(int64_t id, double lat, double lon, const char *tags)
{
   osm_node_t *n = malloc_node((short) (count_tags(tags) + 1));
   osm_node_default(n);
   n->obj.id = id; n->lat = lat; n->lon = lon;
   set_tags(&n->obj, tags);
   return put_object(&n->obj);
}

int hs_way(int64_t id, const int64_t *ref, int cnt, const char *tags)
{
   osm_way_t *w = malloc_way((short) (count_tags(tags) + 1), cnt);
   osm_way_default(w);
   w->obj.id = id;
   memcpy(w->ref, ref, sizeof(*ref) * cnt);
   set_tags(&w->obj, tags);
   return put_object(&w->obj);
}

int hs_rel_of_ways(int64_t id, const int64_t *way, int cnt)
{
   osm_rel_t *r = malloc_rel(0, (short) cnt);
   r->obj.type = OSM_REL; r->obj.id = id; r->obj.vis = 1;
   for (int i = 0; i < cnt; i++) { r->mem[i].type = OSM_WAY; r->mem[i].id = way[i]; r->mem[i].role = ROLE_EMPTY; }
   return put_object(&r->obj);
}

/* call_ini -> apply to the listed objects in the given order -> call_fini (src/smcore.c:382-421,
 * :248-312, :324-379): >0 from _ini disables the rule, non-zero from _main stops the traversal */
int hs_apply(smrule_t *r, int (*ini)(smrule_t *), int (*mainf)(smrule_t *, osm_obj_t *), int (*fini)(smrule_t *),
             int type, const int64_t *ids, int n)
{
   int e = ini ? ini(r) : 0;
   if (e < 0) return 1000 + -e;
   if (e > 0) return 2000 + e;
   for (int i = 0; i < n; i++)
   {
      osm_obj_t *o = get_object(type, ids[i]);
      if (o == NULL) return 3000;
      if ((e = mainf(r, o)) != 0) break;
   }
   int f = fini ? fini(r) : 0;
   return e ? 4000 + (e < 0 ? -e : e) : (f ? 5000 + (f < 0 ? -f : f) : 0);
}

/* value of a tag (for the autoangle / autoalign checks); NULL if absent */
const char *hs_get_tag(int type, int64_t id, const char *key)
{
   static char buf[256];
   osm_obj_t *o = get_object(type, id);
   if (o == NULL) return NULL;
   int n = match_attr(o, key, NULL);
   if (n < 0 || n >= o->tag_cnt) return NULL;
   int l = o->otag[n].v.len < 255 ? o->otag[n].v.len : 255;
   memcpy(buf, o->otag[n].v.buf, l);
   buf[l] = '\0';
   return buf;
}

/* refs of a way / position of a node / the ids the next new node and way will get (for the objects that cap fontbox=1 and
 * mkarea=1 add through malloc_node / malloc_way / put_object) */
int hs_way_refs(int64_t id, int64_t *out, int cap)
{
   osm_way_t *w = get_object(OSM_WAY, id);
   if (w == NULL) return -1;
   for (int i = 0; i < w->ref_cnt && i < cap; i++) out[i] = w->ref[i];
   return w->ref_cnt;
}

int hs_node_pos(int64_t id, double *latlon)
{
   osm_node_t *n = get_object(OSM_NODE, id);
   if (n == NULL) return -1;
   latlon[0] = n->lat; latlon[1] = n->lon;
   return n->obj.tag_cnt;
}

/* unique_node_id / unique_way_id hand out descending ids and have no peek: take one and say which comes next */
int64_t hs_next_node_id(void) { return unique_node_id() - 1; }
int64_t hs_next_way_id(void) { return unique_way_id() - 1; }

/* ---- match tags and the executor's filter (for smr_hip_objects_match) ----
 * A rule as init_rule leaves it (src/smrparse.c:365-467): oo->otag[] = the patterns with their markers stripped, act->stag[] =
 * their kinds, act->flags = ACTION_OPEN_WAY / ACTION_CLOSED_WAY from the rule way's own refs (check_way_type :340-351).
 * match_kind restates parse_matchtype (:65-139): "!..!" inverts, "~..~" negates, then "/../" is a regular expression
 * (regcomp, extended, no sub-matches), "]..[" greater than, "[..]" less than (strtod); anything else compares bytes. */
static int match_kind(bstring_t *b, struct specialTag *t)
{
   t->type = 0;
   if (b->len > 2 && b->buf[0] == b->buf[b->len - 1] && (b->buf[0] == '!' || b->buf[0] == '~'))
   {
      t->type |= b->buf[0] == '!' ? SPECIAL_INVERT : SPECIAL_NOT;
      b->buf[b->len - 1] = '\0'; b->buf++; b->len -= 2;
   }
   if (b->len <= 2) return 0;
   const char c0 = b->buf[0], c1 = b->buf[b->len - 1];
   if (!((c0 == '/' && c1 == '/') || (c0 == ']' && c1 == '[') || (c0 == '[' && c1 == ']'))) return 0;
   b->buf[b->len - 1] = '\0'; b->buf++; b->len -= 2;
   if (c0 == '/')
   {
      if (regcomp(&t->re, b->buf, REG_EXTENDED | REG_NOSUB)) return -1;
      t->type |= SPECIAL_REGEX;
      return 0;
   }
   errno = 0;
   t->val = strtod(b->buf, NULL);
   if (errno) return -2;
   t->type |= c0 == ']' ? SPECIAL_GT : SPECIAL_LT;
   return 0;
}

/* way_kind: 0 = the rule way has no refs, 1 = open (ACTION_OPEN_WAY), 2 = closed (ACTION_CLOSED_WAY) */
smrule_t *hs_match_rule(int type, const char *const *keys, const char *const *vals, int n, int way_kind)
{
   smrule_t *r = calloc(1, sizeof(*r));
   action_t *a = calloc(1, sizeof(*a) + sizeof(struct stag) * (n > 0 ? n : 1));
   osm_node_t *oo = calloc(1, sizeof(osm_storage_t));
   oo->obj.type = (int8_t) type;
   oo->obj.id = 0x7f000002;
   oo->obj.otag = calloc(n > 0 ? n : 1, sizeof(struct otag));
   oo->obj.tag_cnt = (short) n;
   r->oo = &oo->obj; r->act = a;
   a->tag_cnt = (short) n;
   if (type == OSM_WAY && way_kind) a->flags |= way_kind == 2 ? ACTION_CLOSED_WAY : ACTION_OPEN_WAY;
   for (int i = 0; i < n; i++)
   {
      struct otag *ot = &oo->obj.otag[i];
      ot->k.buf = strdup(keys[i]); ot->k.len = (int) strlen(keys[i]);
      ot->v.buf = strdup(vals[i]); ot->v.len = (int) strlen(vals[i]);
      if (match_kind(&ot->k, &a->stag[i].stk) < 0 || match_kind(&ot->v, &a->stag[i].stv) < 0) break;      /* init_rule :385-386 */
   }
   return r;
}

/* apply_rule (src/smcore.c:248-291) up to the call of `_main`, with the REAL bs_match_attr of libsmrender: out[i] = 1 where the
 * action would be called.  on_page: the is_on_page verdict per node (NULL = render_all_nodes_) */
int hs_filter_ptrs(const smrule_t *r, const void *const *objs, int64_t n, const uint8_t *on_page, uint8_t *out)
{
   for (int64_t k = 0; k < n; k++)
   {
      const osm_obj_t *o = objs[k];
      out[k] = 0;
      if (o == NULL) return -1;
      if (o->type == OSM_NODE && on_page != NULL && !on_page[k]) continue;
      if (r->oo->type == OSM_WAY)
      {
         const osm_way_t *w = (const osm_way_t *) o;
         if (sm_is_flag_set(r, ACTION_CLOSED_WAY)) { if (w->ref_cnt && w->ref[0] != w->ref[w->ref_cnt - 1]) continue; }
         else if (sm_is_flag_set(r, ACTION_OPEN_WAY)) { if (w->ref_cnt && w->ref[0] == w->ref[w->ref_cnt - 1]) continue; }
      }
      int i;
      for (i = 0; i < r->oo->tag_cnt; i++)
         if (bs_match_attr(o, &r->oo->otag[i], &r->act->stag[i]) == -1) break;
      if (i < r->oo->tag_cnt) continue;
      if (!o->vis) continue;
      out[k] = 1;
   }
   return 0;
}

int hs_filter(const smrule_t *r, int type, const int64_t *ids, int n, const uint8_t *on_page, uint8_t *out)
{
   const void **p = malloc(sizeof(*p) * (n > 0 ? n : 1));
   for (int k = 0; k < n; k++) p[k] = get_object(type, ids[k]);
   int e = hs_filter_ptrs(r, p, n, on_page, out);
   free(p);
   return e;
}

int hs_object_ptrs(int type, const int64_t *ids, int n, void **out)
{
   for (int k = 0; k < n; k++)
      if ((out[k] = get_object(type, ids[k])) == NULL) return -1;
   return 0;
}

int hs_set_visible(int type, int64_t id, int vis)
{
   osm_obj_t *o = get_object(type, id);
   if (o == NULL) return -1;
   o->vis = (int8_t) vis;
   return 0;
}

/* a large synthetic world for tools/objects_bench.py, allocated with the real malloc_node / malloc_way but kept OUT of the
 * bxtree (the export takes pointers): n_ways closed ways of refs_per_way nodes of their own, every way with tags_per_way tags
 * drawn from the given pairs plus a name of its own; one node in node_tag_every carries one tag */
static unsigned long long rng_(unsigned long long *s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }

int hs_synth_world(int64_t n_ways, int refs_per_way, const char *const *keys, const char *const *vals, int n_pairs, int tags_per_way,
                   int node_tag_every, unsigned seed, void **nodes_out, void **ways_out)
{
   unsigned long long st = 0x9e3779b97f4a7c15ull ^ seed;
   const int64_t per = refs_per_way - 1;
   char *names = malloc((size_t) n_ways * 12);
   for (int64_t w = 0; w < n_ways; w++)
   {
      for (int64_t k = 0; k < per; k++)
      {
         const int64_t i = w * per + k;
         const int tagged = node_tag_every > 0 && rng_(&st) % node_tag_every == 0;
         osm_node_t *n = malloc_node((short) tagged);
         n->obj.type = OSM_NODE; n->obj.vis = 1; n->obj.id = i + 1;
         n->lat = 44.0 + (double) (rng_(&st) % 2000000) * 1e-6; n->lon = 14.0 + (double) (rng_(&st) % 2000000) * 1e-6;
         if (tagged)
         {
            const int p = (int) (rng_(&st) % n_pairs);
            n->obj.otag[0].k.buf = (char *) keys[p]; n->obj.otag[0].k.len = (int) strlen(keys[p]);
            n->obj.otag[0].v.buf = (char *) vals[p]; n->obj.otag[0].v.len = (int) strlen(vals[p]);
         }
         nodes_out[i] = n;
      }
      osm_way_t *wy = malloc_way((short) (tags_per_way + 1), refs_per_way);
      wy->obj.type = OSM_WAY; wy->obj.vis = 1; wy->obj.id = w + 1;
      for (int64_t k = 0; k < per; k++) wy->ref[k] = w * per + k + 1;
      wy->ref[per] = wy->ref[0];
      for (int t = 0; t < tags_per_way; t++)
      {
         const int p = (int) (rng_(&st) % n_pairs);
         wy->obj.otag[t].k.buf = (char *) keys[p]; wy->obj.otag[t].k.len = (int) strlen(keys[p]);
         wy->obj.otag[t].v.buf = (char *) vals[p]; wy->obj.otag[t].v.len = (int) strlen(vals[p]);
      }
      char *nm = names + w * 12;
      wy->obj.otag[tags_per_way].k.buf = "name"; wy->obj.otag[tags_per_way].k.len = 4;
      wy->obj.otag[tags_per_way].v.buf = nm; wy->obj.otag[tags_per_way].v.len = snprintf(nm, 12, "w%lld", (long long) w);
      ways_out[w] = wy;
   }
   return 0;
}
