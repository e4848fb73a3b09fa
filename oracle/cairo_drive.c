/* cairo_drive.c -- TEST INFRASTRUCTURE (oracle side), not product code.
 *
 * Drives the real third-party rasterizer (libcairo 1.16.0 + pixman 0.40.0, the
 * un-vendored dependency of the reference: configure.ac:52) with the SAME call
 * sequence the reference issues for the draw path, restated from
 *   src/smrules_cairo.c:308-330  (main recording surface in pt, bg paint)
 *   src/smrules_cairo.c:653-660  (per-rule context + push_group)
 *   src/smrules_cairo.c:845-863  (new_sub_path + line_to per ref)
 *   src/smrules_cairo.c:953-1016 (border stroke, then fill / open stroke)
 *   src/smrules_cairo.c:1081-1082 (pop_group_to_source + paint)
 *   src/smrules_cairo.c:339-356  (replay: scale(dpi/72), set_source_surface, paint)
 *   src/smrules_cairo.c:1511-1558, 1746-1759 (caption cut-out / plane / rotated diff)
 * It is used (a) here, to generate the golden fixtures in tests/golden and to pin
 * the C restatement in smr_oracle.c, and (b) on the GPU box as the "reference
 * CPU render" timed beside the HIP path (bench.py cpu_baseline). Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * libcairo is dlopen()ed so the file builds without cairo headers.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct _cairo cairo_t;
typedef struct _cairo_surface cairo_surface_t;
typedef struct _cairo_pattern cairo_pattern_t;
typedef struct { double x, y, width, height; } cairo_rectangle_t;
enum { CAIRO_CONTENT_COLOR_ALPHA = 0x3000, CAIRO_FORMAT_ARGB32 = 0,
       CAIRO_LINE_CAP_BUTT = 0, CAIRO_LINE_CAP_ROUND = 1, CAIRO_OPERATOR_OVER = 2 };

#define FN(ret, name, args) static ret (*p_##name) args;
FN(cairo_surface_t*, cairo_recording_surface_create, (int, const cairo_rectangle_t*))
FN(cairo_surface_t*, cairo_image_surface_create, (int, int, int))
FN(cairo_surface_t*, cairo_image_surface_create_for_data, (unsigned char*, int, int, int, int))
FN(cairo_t*, cairo_create, (cairo_surface_t*))
FN(void, cairo_destroy, (cairo_t*))
FN(void, cairo_surface_destroy, (cairo_surface_t*))
FN(void, cairo_surface_flush, (cairo_surface_t*))
FN(void, cairo_surface_mark_dirty, (cairo_surface_t*))
FN(unsigned char*, cairo_image_surface_get_data, (cairo_surface_t*))
FN(int, cairo_image_surface_get_stride, (cairo_surface_t*))
FN(void, cairo_set_source_rgba, (cairo_t*, double, double, double, double))
FN(void, cairo_set_source_surface, (cairo_t*, cairo_surface_t*, double, double))
FN(void, cairo_paint, (cairo_t*))
FN(void, cairo_push_group, (cairo_t*))
FN(void, cairo_pop_group_to_source, (cairo_t*))
FN(void, cairo_new_sub_path, (cairo_t*))
FN(void, cairo_move_to, (cairo_t*, double, double))
FN(void, cairo_line_to, (cairo_t*, double, double))
FN(void, cairo_curve_to, (cairo_t*, double, double, double, double, double, double))
FN(void, cairo_rectangle, (cairo_t*, double, double, double, double))
FN(void, cairo_set_line_width, (cairo_t*, double))
FN(void, cairo_set_line_cap, (cairo_t*, int))
FN(void, cairo_set_dash, (cairo_t*, const double*, int, double))
FN(void, cairo_stroke, (cairo_t*))
FN(void, cairo_fill, (cairo_t*))
FN(void, cairo_set_fill_rule, (cairo_t*, int))
typedef struct { double xx, yx, xy, yy, x0, y0; } cd_matrix_t;       /* cairo_matrix_t */
FN(cairo_pattern_t*, cairo_pattern_create_for_surface, (cairo_surface_t*))
FN(void, cairo_pattern_set_matrix, (cairo_pattern_t*, const cd_matrix_t*))
FN(void, cairo_pattern_set_extend, (cairo_pattern_t*, int))
FN(void, cairo_pattern_destroy, (cairo_pattern_t*))
FN(void, cairo_set_source, (cairo_t*, cairo_pattern_t*))
FN(void, cairo_matrix_init_scale, (cd_matrix_t*, double, double))
FN(void, cairo_matrix_rotate, (cd_matrix_t*, double))
FN(void, cairo_scale, (cairo_t*, double, double))
FN(void, cairo_translate, (cairo_t*, double, double))
FN(void, cairo_rotate, (cairo_t*, double))
FN(void, cairo_save, (cairo_t*))
FN(void, cairo_restore, (cairo_t*))
FN(void, cairo_set_operator, (cairo_t*, int))
FN(int, cairo_status, (cairo_t*))
FN(int, cairo_surface_status, (cairo_surface_t*))
FN(const char*, cairo_version_string, (void))
FN(int, cairo_surface_write_to_png, (cairo_surface_t*, const char*))
typedef struct { double x_bearing, y_bearing, width, height, x_advance, y_advance; } cairo_text_extents_t;
typedef struct { double ascent, descent, height, max_x_advance, max_y_advance; } cairo_font_extents_t;
FN(void*, cairo_font_options_create, (void))
FN(void, cairo_font_options_set_hint_metrics, (void*, int))
FN(void, cairo_set_font_options, (cairo_t*, const void*))
FN(void, cairo_font_options_destroy, (void*))
FN(void, cairo_select_font_face, (cairo_t*, const char*, int, int))
FN(void, cairo_set_font_size, (cairo_t*, double))
FN(void, cairo_font_extents, (cairo_t*, cairo_font_extents_t*))
FN(void, cairo_text_extents, (cairo_t*, const char*, cairo_text_extents_t*))

static void *lib_;
#define LOAD(name) do { *(void**)(&p_##name) = dlsym(lib_, #name); if (!p_##name) { fprintf(stderr, "cairo_drive: missing %s\n", #name); return -2; } } while (0)

int cd_init(const char *path)
{
   if (lib_) return 0;
   const char *cand[] = { path, "libcairo.so.2", "/opt/conda/lib/libcairo.so.2",
                          "/usr/lib/x86_64-linux-gnu/libcairo.so.2", NULL };
   for (int i = 0; cand[i] != NULL || i == 0; i++)
   {
      if (cand[i] == NULL) continue;
      if ((lib_ = dlopen(cand[i], RTLD_NOW | RTLD_GLOBAL)) != NULL) break;
   }
   if (!lib_) return -1;
   LOAD(cairo_recording_surface_create); LOAD(cairo_image_surface_create);
   LOAD(cairo_image_surface_create_for_data); LOAD(cairo_create); LOAD(cairo_destroy);
   LOAD(cairo_surface_destroy); LOAD(cairo_surface_flush); LOAD(cairo_surface_mark_dirty);
   LOAD(cairo_image_surface_get_data); LOAD(cairo_image_surface_get_stride);
   LOAD(cairo_set_source_rgba); LOAD(cairo_set_source_surface); LOAD(cairo_paint);
   LOAD(cairo_push_group); LOAD(cairo_pop_group_to_source); LOAD(cairo_new_sub_path);
   LOAD(cairo_move_to); LOAD(cairo_line_to); LOAD(cairo_curve_to); LOAD(cairo_rectangle);
   LOAD(cairo_set_line_width); LOAD(cairo_set_line_cap); LOAD(cairo_set_dash);
   LOAD(cairo_stroke); LOAD(cairo_fill); LOAD(cairo_set_fill_rule); LOAD(cairo_scale); LOAD(cairo_translate);
   LOAD(cairo_rotate); LOAD(cairo_save); LOAD(cairo_restore); LOAD(cairo_set_operator);
   LOAD(cairo_status); LOAD(cairo_surface_status); LOAD(cairo_version_string); LOAD(cairo_surface_write_to_png);
   LOAD(cairo_font_options_create); LOAD(cairo_font_options_set_hint_metrics); LOAD(cairo_set_font_options);
   LOAD(cairo_font_options_destroy); LOAD(cairo_select_font_face); LOAD(cairo_set_font_size);
   LOAD(cairo_font_extents); LOAD(cairo_text_extents);
   LOAD(cairo_pattern_create_for_surface); LOAD(cairo_pattern_set_matrix); LOAD(cairo_pattern_set_extend); LOAD(cairo_pattern_destroy);
   LOAD(cairo_set_source); LOAD(cairo_matrix_init_scale); LOAD(cairo_matrix_rotate);
   return 0;
}

const char *cd_version(void) { return lib_ ? p_cairo_version_string() : "unloaded"; }

/* colour model of the reference: src/smrules_cairo.c:90-95, :271-274 */
static void set_color(cairo_t *ctx, uint32_t col)
{
   double r = (double) ((col >> 16) & 0xff) / 255.0;
   double g = (double) ((col >> 8) & 0xff) / 255.0;
   double b = (double) (col & 0xff) / 255.0;
   double a = 1.0 - (double) (((col & 0x7f000000) >> 23) & 0xff) / 255.0;
   p_cairo_set_source_rgba(ctx, r, g, b, a);
}

/* ---- chart state: one main recording surface (sfc_) + the current rule ctx ---- */
static cairo_surface_t *sfc_;
static cairo_rectangle_t ext_;
static cairo_t *rule_;

int cd_begin(double w_pt, double h_pt, uint32_t bgcolor)
{
   if (!lib_) return -1;
   if (sfc_) { p_cairo_surface_destroy(sfc_); sfc_ = NULL; }
   ext_.x = 0; ext_.y = 0; ext_.width = w_pt; ext_.height = h_pt;
   sfc_ = p_cairo_recording_surface_create(CAIRO_CONTENT_COLOR_ALPHA, &ext_);
   cairo_t *ctx = p_cairo_create(sfc_);
   set_color(ctx, bgcolor);
   p_cairo_paint(ctx);
   p_cairo_destroy(ctx);
   return p_cairo_surface_status(sfc_);
}

int cd_rule_begin(void)
{
   rule_ = p_cairo_create(sfc_);
   p_cairo_push_group(rule_);
   return p_cairo_status(rule_);
}

int cd_rule_end(void)
{
   p_cairo_pop_group_to_source(rule_);
   p_cairo_paint(rule_);
   int e = p_cairo_status(rule_);
   p_cairo_destroy(rule_);
   rule_ = NULL;
   return e;
}

/* new_sub_path + line_to per point (cairo_smr_poly_line) */
void cd_poly_line(const double *xy, int n)
{
   p_cairo_new_sub_path(rule_);
   for (int i = 0; i < n; i++)
      p_cairo_line_to(rule_, xy[2 * i], xy[2 * i + 1]);
}

/* move_to + n curve_to (cairo_smr_poly_curve); c = [c1x c1y c2x c2y x y] * n */
void cd_poly_curve(double x0, double y0, const double *c, int n)
{
   p_cairo_new_sub_path(rule_);
   p_cairo_move_to(rule_, x0, y0);
   for (int i = 0; i < n; i++, c += 6)
      p_cairo_curve_to(rule_, c[0], c[1], c[2], c[3], c[4], c[5]);
}

void cd_set_dash(const double *dash, int n, int round_cap)
{
   if (round_cap) p_cairo_set_line_cap(rule_, CAIRO_LINE_CAP_ROUND);
   p_cairo_set_dash(rule_, dash, n, 0);
}

void cd_stroke(uint32_t col, double width_pt)
{
   set_color(rule_, col);
   p_cairo_set_line_width(rule_, width_pt);
   p_cairo_stroke(rule_);
}

/* CAIRO_FILL_RULE_WINDING = 0 (libcairo's default, the only one the reference uses), CAIRO_FILL_RULE_EVEN_ODD = 1: on the current
 * rule's context, for the even-odd extension of the rasterizer */
void cd_set_fill_rule(int evenodd) { p_cairo_set_fill_rule(rule_, evenodd ? 1 : 0); }

/* img_ini's pattern (src/smrules_cairo.c:2691-2704) as the source of the current rule's context: the image (w x h premultiplied
 * ARGB32, copied) as a surface pattern with the matrix scale(1 / PX2PT_SCALE) . rotate(angle) and CAIRO_EXTEND_REPEAT;
 * cd_fill_source fills the current path with it (img_fill :2742-2751, render_poly_line :989-997) */
static cairo_surface_t *pat_sfc_;
int cd_set_source_pattern(const uint32_t *pix, int w, int h, double dpi, double angle_deg)
{
   if (pat_sfc_) { p_cairo_surface_destroy(pat_sfc_); pat_sfc_ = NULL; }
   pat_sfc_ = p_cairo_image_surface_create(CAIRO_FORMAT_ARGB32, w, h);
   if (p_cairo_surface_status(pat_sfc_)) return -1;
   p_cairo_surface_flush(pat_sfc_);
   unsigned char *d = p_cairo_image_surface_get_data(pat_sfc_);
   int stride = p_cairo_image_surface_get_stride(pat_sfc_);
   for (int y = 0; y < h; y++) memcpy(d + (size_t) y * stride, pix + (size_t) y * w, (size_t) w * 4);
   p_cairo_surface_mark_dirty(pat_sfc_);
   cairo_pattern_t *pat = p_cairo_pattern_create_for_surface(pat_sfc_);
   cd_matrix_t m;
   p_cairo_matrix_init_scale(&m, dpi / 72, dpi / 72);
   p_cairo_matrix_rotate(&m, angle_deg * M_PI / 180);
   p_cairo_pattern_set_matrix(pat, &m);
   p_cairo_pattern_set_extend(pat, 1 /* CAIRO_EXTEND_REPEAT */);
   p_cairo_set_source(rule_, pat);
   p_cairo_pattern_destroy(pat);
   return p_cairo_status(rule_);
}

void cd_fill_source(void) { p_cairo_fill(rule_); }

/* img_ini's image (:2637-2658): the decoded w x h ARGB32 picture painted at `scale` into a new image of (int) (w * scale) x
 * (int) (h * scale) pixels, then the transparency taken off with DEST_OUT.  out holds that many pixels; returns its width,
 * its height in *oh */
int cd_scaled_image(const uint32_t *pix, int w, int h, double scale, double trans, uint32_t *out, int cap, int *oh)
{
   cairo_surface_t *src = p_cairo_image_surface_create(CAIRO_FORMAT_ARGB32, w, h);
   p_cairo_surface_flush(src);
   unsigned char *d = p_cairo_image_surface_get_data(src);
   int stride = p_cairo_image_surface_get_stride(src);
   for (int y = 0; y < h; y++) memcpy(d + (size_t) y * stride, pix + (size_t) y * w, (size_t) w * 4);
   p_cairo_surface_mark_dirty(src);
   const int ow = (int) (w * scale), hh = (int) (h * scale);
   if (ow < 1 || hh < 1 || (long) ow * hh > cap) { p_cairo_surface_destroy(src); return -1; }
   cairo_surface_t *img = p_cairo_image_surface_create(CAIRO_FORMAT_ARGB32, ow, hh);
   cairo_t *ctx = p_cairo_create(img);
   p_cairo_scale(ctx, scale, scale);
   p_cairo_set_source_surface(ctx, src, 0, 0);
   p_cairo_paint(ctx);
   p_cairo_set_source_rgba(ctx, 0, 0, 0, trans);
   p_cairo_set_operator(ctx, 9 /* CAIRO_OPERATOR_DEST_OUT */);
   p_cairo_paint(ctx);
   p_cairo_destroy(ctx);
   p_cairo_surface_destroy(src);
   p_cairo_surface_flush(img);
   d = p_cairo_image_surface_get_data(img);
   stride = p_cairo_image_surface_get_stride(img);
   for (int y = 0; y < hh; y++) memcpy(out + (size_t) y * ow, d + (size_t) y * stride, (size_t) ow * 4);
   p_cairo_surface_destroy(img);
   *oh = hh;
   return ow;
}

void cd_fill(uint32_t col)
{
   set_color(rule_, col);
   p_cairo_fill(rule_);
}

/* replay the chart into a W x H ARGB32 window whose top-left pixel is (x0,y0)
 * of the full canvas (cairo_smr_image_surface_from_bg convention + integer tile origin) */
int cd_render(double dpi, int x0, int y0, int W, int H, uint32_t *out)
{
   cairo_surface_t *img = p_cairo_image_surface_create(CAIRO_FORMAT_ARGB32, W, H);
   if (p_cairo_surface_status(img)) return -1;
   cairo_t *dst = p_cairo_create(img);
   p_cairo_translate(dst, -x0, -y0);
   p_cairo_scale(dst, dpi / 72, dpi / 72);
   p_cairo_set_source_surface(dst, sfc_, 0, 0);
   p_cairo_paint(dst);
   int e = p_cairo_status(dst);
   p_cairo_destroy(dst);
   p_cairo_surface_flush(img);
   unsigned char *d = p_cairo_image_surface_get_data(img);
   int stride = p_cairo_image_surface_get_stride(img);
   for (int y = 0; y < H; y++)
      memcpy(out + (size_t) y * W, d + (size_t) y * stride, (size_t) W * 4);
   p_cairo_surface_destroy(img);
   return e;
}

/* save_main_image's PNG branch (src/smrules_cairo.c:388-412): replay the chart into a W x H ARGB32 image and write it as PNG
 * (timing anchor of tools/cpu_anchor.py; the page rotation of :405 is the identity for an unrotated chart) */
int cd_save_png(double dpi, int W, int H, const char *path)
{
   cairo_surface_t *img = p_cairo_image_surface_create(CAIRO_FORMAT_ARGB32, W, H);
   if (p_cairo_surface_status(img)) return -1;
   cairo_t *dst = p_cairo_create(img);
   p_cairo_scale(dst, dpi / 72, dpi / 72);
   p_cairo_set_source_surface(dst, sfc_, 0, 0);
   p_cairo_paint(dst);
   p_cairo_destroy(dst);
   int e = p_cairo_surface_write_to_png(img, path);
   p_cairo_surface_destroy(img);
   return e;
}

/* libcairo's own PNG writer on given ARGB32 pixels (what save_main_image :409 and save_image :467 call on the rendered image):
 * the pin of smr_hip_canvas_write_png / _write_tiles, whose files must DECODE to the same pixels */
int cd_write_png_data(const uint32_t *argb, int W, int H, const char *path)
{
   cairo_surface_t *img = p_cairo_image_surface_create(CAIRO_FORMAT_ARGB32, W, H);
   if (p_cairo_surface_status(img)) return -1;
   p_cairo_surface_flush(img);
   unsigned char *d = p_cairo_image_surface_get_data(img);
   int stride = p_cairo_image_surface_get_stride(img);
   for (int y = 0; y < H; y++) memcpy(d + (size_t) y * stride, argb + (size_t) y * W, (size_t) W * 4);
   p_cairo_surface_mark_dirty(img);
   int e = p_cairo_surface_write_to_png(img, path);
   p_cairo_surface_destroy(img);
   return e;
}

/* cairo_smr_cut_out (src/smrules_cairo.c:1511-1533): the r x r pt window of the chart around (x, y) pt as find_angle takes it --
 * the recording replayed at dpi/72 with the translation (-x + r/2, -y + r/2) pt, i.e. at a sub-pixel offset whenever the
 * anchor is not on the pixel grid.  out holds n x n pixels, n = round(r * dpi / 72) (returned). */
int cd_cut_out(double x_pt, double y_pt, double r_pt, double dpi, uint32_t *out, int cap)
{
   int n = (int) round(r_pt * dpi / 72);
   if (n <= 0 || n * n > cap) return -1;
   cairo_surface_t *img = p_cairo_image_surface_create(CAIRO_FORMAT_ARGB32, n, n);
   if (p_cairo_surface_status(img)) return -1;
   cairo_t *ctx = p_cairo_create(img);
   p_cairo_scale(ctx, dpi / 72, dpi / 72);
   p_cairo_set_source_surface(ctx, sfc_, -x_pt + r_pt / 2, -y_pt + r_pt / 2);
   p_cairo_paint(ctx);
   int e = p_cairo_status(ctx);
   p_cairo_destroy(ctx);
   p_cairo_surface_flush(img);
   unsigned char *d = p_cairo_image_surface_get_data(img);
   int stride = p_cairo_image_surface_get_stride(img);
   for (int y = 0; y < n; y++) memcpy(out + (size_t) y * n, d + (size_t) y * stride, (size_t) n * 4);
   p_cairo_surface_destroy(img);
   return e ? -1 : n;
}

/* direct (no recording surface) variant used for micro-probes: draw a polygon in
 * device px straight onto an image */
/* fill rule of the next cd_direct_fill calls (0 winding = libcairo's default, 1 even-odd) */
static int direct_rule_;
void cd_set_direct_fill_rule(int evenodd) { direct_rule_ = evenodd ? 1 : 0; }

int cd_direct_fill(const double *xy, int n, uint32_t col, uint32_t bg, int W, int H, uint32_t *out)
{
   cairo_surface_t *img = p_cairo_image_surface_create(CAIRO_FORMAT_ARGB32, W, H);
   cairo_t *c = p_cairo_create(img);
   set_color(c, bg); p_cairo_paint(c);
   p_cairo_new_sub_path(c);
   for (int i = 0; i < n; i++) p_cairo_line_to(c, xy[2 * i], xy[2 * i + 1]);
   p_cairo_set_fill_rule(c, direct_rule_);
   set_color(c, col); p_cairo_fill(c);
   p_cairo_destroy(c);
   p_cairo_surface_flush(img);
   unsigned char *d = p_cairo_image_surface_get_data(img);
   int stride = p_cairo_image_surface_get_stride(img);
   for (int y = 0; y < H; y++) memcpy(out + (size_t) y * W, d + (size_t) y * stride, (size_t) W * 4);
   p_cairo_surface_destroy(img);
   return 0;
}

/* micro-probe for the spline decomposition: one Bezier chain in device px (exact 24.8 values) filled straight
 * onto an image */
int cd_direct_curve_fill(double x0, double y0, const double *c, int n, uint32_t col, uint32_t bg, int W, int H, uint32_t *out)
{
   cairo_surface_t *img = p_cairo_image_surface_create(CAIRO_FORMAT_ARGB32, W, H);
   cairo_t *cr = p_cairo_create(img);
   set_color(cr, bg); p_cairo_paint(cr);
   p_cairo_new_sub_path(cr);
   p_cairo_move_to(cr, x0, y0);
   for (int i = 0; i < n; i++, c += 6) p_cairo_curve_to(cr, c[0], c[1], c[2], c[3], c[4], c[5]);
   set_color(cr, col); p_cairo_fill(cr);
   p_cairo_destroy(cr);
   p_cairo_surface_flush(img);
   unsigned char *d = p_cairo_image_surface_get_data(img);
   int stride = p_cairo_image_surface_get_stride(img);
   for (int y = 0; y < H; y++) memcpy(out + (size_t) y * W, d + (size_t) y * stride, (size_t) W * 4);
   p_cairo_surface_destroy(img);
   return 0;
}

/* ---- caption scan pieces (cairo_smr_diff): paint src rotated by a about the
 * centre of dst, OVER, default (bilinear) filter.  dst is updated in place. ---- */
int cd_rot_paint(uint32_t *dstpix, int dw, int dh, const uint32_t *srcpix, int sw, int sh, double a)
{
   cairo_surface_t *dst = p_cairo_image_surface_create_for_data((unsigned char*) dstpix, CAIRO_FORMAT_ARGB32, dw, dh, dw * 4);
   cairo_surface_t *src = p_cairo_image_surface_create_for_data((unsigned char*) srcpix, CAIRO_FORMAT_ARGB32, sw, sh, sw * 4);
   cairo_t *ctx = p_cairo_create(dst);
   double x = dw, y = dh;
   p_cairo_save(ctx);
   p_cairo_translate(ctx, x / 2.0, y / 2.0);
   p_cairo_rotate(ctx, a);
   p_cairo_set_operator(ctx, CAIRO_OPERATOR_OVER);
   p_cairo_set_source_surface(ctx, src, sw / -2.0, sh / -2.0);
   p_cairo_paint(ctx);
   p_cairo_restore(ctx);
   int e = p_cairo_status(ctx);
   p_cairo_destroy(ctx);
   p_cairo_surface_flush(dst);
   p_cairo_surface_destroy(dst);
   p_cairo_surface_destroy(src);
   return e;
}

/* ---- caption geometry: the part of cap_coord before find_angle (src/smrules_cairo.c:2250-2279)
 * with the font set-up of act_cap_ini (:1335-1357; fontconfig face when libcairo has it, like
 * the reference's CAIRO_HAS_FC_FONT branch, else the toy face of its #else branch).
 * out[0..2] = the int arguments (w, h, x in pt) cap_coord passes to cairo_smr_plane. ---- */
static double metrics_[4];
void cd_last_metrics(double *out) { memcpy(out, metrics_, sizeof(metrics_)); }

int cd_caption_geometry(const char *font, double size_mm, const char *text, double xoff_pt, double yoff_pt, int ew_aligned,
                        double w_pt, double h_pt, int *out)
{
   cairo_rectangle_t ext = { 0, 0, w_pt, h_pt };
   cairo_surface_t *sfc = p_cairo_recording_surface_create(CAIRO_CONTENT_COLOR_ALPHA, &ext);
   cairo_t *ctx = p_cairo_create(sfc);
   void *fo = p_cairo_font_options_create();
   p_cairo_font_options_set_hint_metrics(fo, 1);
   p_cairo_set_font_options(ctx, fo);
   p_cairo_font_options_destroy(fo);
   void *(*fc_parse)(const unsigned char*) = (void *(*)(const unsigned char*)) dlsym(RTLD_DEFAULT, "FcNameParse");
   void (*fc_destroy)(void*) = (void (*)(void*)) dlsym(RTLD_DEFAULT, "FcPatternDestroy");
   void *(*ft_face)(void*) = (void *(*)(void*)) dlsym(lib_, "cairo_ft_font_face_create_for_pattern");
   void (*set_face)(cairo_t*, void*) = (void (*)(cairo_t*, void*)) dlsym(lib_, "cairo_set_font_face");
   void (*face_destroy)(void*) = (void (*)(void*)) dlsym(lib_, "cairo_font_face_destroy");
   void *pat;
   if (fc_parse && fc_destroy && ft_face && set_face && face_destroy && (pat = fc_parse((const unsigned char*) font)) != NULL)
   {
      void *cfc = ft_face(pat);
      fc_destroy(pat);
      set_face(ctx, cfc);
      face_destroy(cfc);
   }
   else
      p_cairo_select_font_face(ctx, font, 0, 0);
   cairo_text_extents_t tx; cairo_font_extents_t fe;
   p_cairo_set_font_size(ctx, size_mm * 72 / 25.4);
   p_cairo_font_extents(ctx, &fe);
   p_cairo_text_extents(ctx, text, &tx);
   double width = tx.width + tx.x_bearing + xoff_pt, height = fe.ascent;
   (void) yoff_pt;
   if (ew_aligned) { out[0] = (int) (width * 2); out[1] = (int) height; out[2] = (int) width; }
   else { out[0] = (int) width; out[1] = (int) height; out[2] = 0; }
   /* the extents cap_coord uses for the auto_ctx rectangle (:2364-2379) */
   metrics_[0] = tx.width + tx.x_bearing; metrics_[1] = fe.ascent; metrics_[2] = fe.descent; metrics_[3] = tx.height;
   p_cairo_destroy(ctx);
   p_cairo_surface_destroy(sfc);
   return 0;
}

/* cap_coord's drawing part (src/smrules_cairo.c:2215-2217, :2363, :2384-2388) on the CURRENT RULE's context (between
 * cd_rule_begin and cd_rule_end, like cap.ctx inside its pushed group): save, translate to the anchor, rotate, move_to the
 * aligned origin, show_text, restore; font set-up as in cd_caption_geometry */
int cd_show_caption(const char *font, double size_mm, const char *text, double x_pt, double y_pt, double angle,
                    double ox_pt, double oy_pt, uint32_t col)
{
   if (!rule_) return -1;
   static void (*show_text)(cairo_t*, const char*);
   if (!show_text) show_text = (void (*)(cairo_t*, const char*)) dlsym(lib_, "cairo_show_text");
   if (!show_text) return -2;
   cairo_t *ctx = rule_;
   p_cairo_save(ctx);
   void *fo = p_cairo_font_options_create();
   p_cairo_font_options_set_hint_metrics(fo, 1);
   p_cairo_set_font_options(ctx, fo);
   p_cairo_font_options_destroy(fo);
   void *(*fc_parse)(const unsigned char*) = (void *(*)(const unsigned char*)) dlsym(RTLD_DEFAULT, "FcNameParse");
   void (*fc_destroy)(void*) = (void (*)(void*)) dlsym(RTLD_DEFAULT, "FcPatternDestroy");
   void *(*ft_face)(void*) = (void *(*)(void*)) dlsym(lib_, "cairo_ft_font_face_create_for_pattern");
   void (*set_face)(cairo_t*, void*) = (void (*)(cairo_t*, void*)) dlsym(lib_, "cairo_set_font_face");
   void (*face_destroy)(void*) = (void (*)(void*)) dlsym(lib_, "cairo_font_face_destroy");
   void *pat;
   if (fc_parse && fc_destroy && ft_face && set_face && face_destroy && (pat = fc_parse((const unsigned char*) font)) != NULL)
   {
      void *cfc = ft_face(pat);
      fc_destroy(pat);
      set_face(ctx, cfc);
      face_destroy(cfc);
   }
   else
      p_cairo_select_font_face(ctx, font, 0, 0);
   p_cairo_set_font_size(ctx, size_mm * 72 / 25.4);
   set_color(ctx, col);
   p_cairo_translate(ctx, x_pt, y_pt);
   p_cairo_rotate(ctx, angle);
   p_cairo_move_to(ctx, ox_pt, oy_pt);
   show_text(ctx, text);
   p_cairo_restore(ctx);
   return p_cairo_status(ctx);
}

/* the same caption drawn the way libsmr_hip's plugin does it: straight onto a w x h image whose pixel (0, 0) is chart pixel
 * (ix0, iy0), no recording surface in between (probe: does the recording + replay change the glyphs?) */
int cd_caption_direct(const char *font, double size_mm, const char *text, double x_pt, double y_pt, double angle,
                      double ox_pt, double oy_pt, uint32_t col, double dpi, int ix0, int iy0, int w, int h, uint32_t *out)
{
   cairo_surface_t *img = p_cairo_image_surface_create(CAIRO_FORMAT_ARGB32, w, h);
   cairo_t *save = rule_;
   rule_ = p_cairo_create(img);
   p_cairo_translate(rule_, -ix0, -iy0);
   p_cairo_scale(rule_, dpi / 72, dpi / 72);
   int e = cd_show_caption(font, size_mm, text, x_pt, y_pt, angle, ox_pt, oy_pt, col);
   p_cairo_destroy(rule_);
   rule_ = save;
   p_cairo_surface_flush(img);
   unsigned char *d = p_cairo_image_surface_get_data(img);
   int stride = p_cairo_image_surface_get_stride(img);
   for (int y = 0; y < h; y++) memcpy(out + (size_t) y * w, d + (size_t) y * stride, (size_t) w * 4);
   p_cairo_surface_destroy(img);
   return e;
}

/* ... and through a recording surface of its own (chart extents in pt, like the reference's), replayed at dpi/72 into the image */
int cd_caption_recorded(const char *font, double size_mm, const char *text, double x_pt, double y_pt, double angle,
                        double ox_pt, double oy_pt, uint32_t col, double dpi, int ix0, int iy0, int w, int h, uint32_t *out)
{
   cairo_surface_t *rec = p_cairo_recording_surface_create(CAIRO_CONTENT_COLOR_ALPHA, &ext_);
   cairo_t *save = rule_;
   rule_ = p_cairo_create(rec);
   int e = cd_show_caption(font, size_mm, text, x_pt, y_pt, angle, ox_pt, oy_pt, col);
   p_cairo_destroy(rule_);
   rule_ = save;
   cairo_surface_t *img = p_cairo_image_surface_create(CAIRO_FORMAT_ARGB32, w, h);
   cairo_t *dst = p_cairo_create(img);
   p_cairo_translate(dst, -ix0, -iy0);
   p_cairo_scale(dst, dpi / 72, dpi / 72);
   p_cairo_set_source_surface(dst, rec, 0, 0);
   p_cairo_paint(dst);
   p_cairo_destroy(dst);
   p_cairo_surface_flush(img);
   unsigned char *d = p_cairo_image_surface_get_data(img);
   int stride = p_cairo_image_surface_get_stride(img);
   for (int y = 0; y < h; y++) memcpy(out + (size_t) y * w, d + (size_t) y * stride, (size_t) w * 4);
   p_cairo_surface_destroy(img);
   p_cairo_surface_destroy(rec);
   return e;
}

/* cairo_smr_plane, src/smrules_cairo.c:1536-1558.  out must hold round(w*dpi/72) x round(h*dpi/72) pixels */
int cd_plane(int w_pt, int h_pt, int x_pt, uint32_t col, int dpi, uint32_t *out)
{
   double s = (double) dpi / 72.0;
   int W = (int) round(w_pt * s), H = (int) round(h_pt * s);
   cairo_surface_t *img = p_cairo_image_surface_create(CAIRO_FORMAT_ARGB32, W, H);
   if (p_cairo_surface_status(img)) return -1;
   cairo_t *c = p_cairo_create(img);
   p_cairo_scale(c, s, s);
   set_color(c, col);
   p_cairo_rectangle(c, x_pt, 0, w_pt - x_pt, h_pt);
   p_cairo_fill(c);
   p_cairo_destroy(c);
   p_cairo_surface_flush(img);
   unsigned char *d = p_cairo_image_surface_get_data(img);
   int stride = p_cairo_image_surface_get_stride(img);
   for (int y = 0; y < H; y++) memcpy(out + (size_t) y * W, d + (size_t) y * stride, (size_t) W * 4);
   p_cairo_surface_destroy(img);
   return 0;
}

/* ---- act_img on nodes (src/smrules_cairo.c:2512-2871), the reference's call sequence on the real libcairo ---------------- */
static cairo_surface_t *sym_;
static double sym_w_, sym_h_;

/* img_ini's PNG branch (:2627-2658): the file scaled into an ARGB32 image of (int) w x (int) h, transparency by DEST_OUT.
 * Returns 0 and the image's size in wh[2] (pixels, as doubles: img->w, img->h) */
int cd_img_load_png(const char *path, double scale, double trans, double *wh)
{
   static cairo_surface_t *(*from_png)(const char*);
   static int (*get_w)(cairo_surface_t*), (*get_h)(cairo_surface_t*);
   if (!from_png)
   {
      from_png = (cairo_surface_t *(*)(const char*)) dlsym(lib_, "cairo_image_surface_create_from_png");
      get_w = (int (*)(cairo_surface_t*)) dlsym(lib_, "cairo_image_surface_get_width");
      get_h = (int (*)(cairo_surface_t*)) dlsym(lib_, "cairo_image_surface_get_height");
   }
   if (!from_png || !get_w || !get_h) return -2;
   if (sym_) { p_cairo_surface_destroy(sym_); sym_ = NULL; }
   cairo_surface_t *sfc = from_png(path);
   if (p_cairo_surface_status(sfc)) { p_cairo_surface_destroy(sfc); return -1; }
   sym_w_ = get_w(sfc) * scale; sym_h_ = get_h(sfc) * scale;
   sym_ = p_cairo_image_surface_create(CAIRO_FORMAT_ARGB32, sym_w_, sym_h_);
   cairo_t *ctx = p_cairo_create(sym_);
   p_cairo_scale(ctx, scale, scale);
   p_cairo_set_source_surface(ctx, sfc, 0, 0);
   p_cairo_paint(ctx);
   p_cairo_set_source_rgba(ctx, 0, 0, 0, trans);
   p_cairo_set_operator(ctx, 9 /* CAIRO_OPERATOR_DEST_OUT */);
   p_cairo_paint(ctx);
   p_cairo_destroy(ctx);
   p_cairo_surface_destroy(sfc);
   wh[0] = sym_w_; wh[1] = sym_h_;
   return 0;
}

/* the symbol's pixels ((int) w x (int) h premultiplied ARGB32): find_angle's plane for angle=auto (:2773-2790) */
int cd_img_pixels(uint32_t *out)
{
   if (!sym_) return -1;
   p_cairo_surface_flush(sym_);
   unsigned char *d = p_cairo_image_surface_get_data(sym_);
   int stride = p_cairo_image_surface_get_stride(sym_), W = (int) sym_w_, H = (int) sym_h_;
   for (int y = 0; y < H; y++) memcpy(out + (size_t) y * W, d + (size_t) y * stride, (size_t) W * 4);
   return 0;
}

/* img_place (:2754-2826) on the current rule's context, which img_ini (:2680-2683) scaled to pixels: save, translate to the
 * node in px, rotate, the symbol as source centred on the node, paint, restore */
int cd_img_place(double x_px, double y_px, double angle, double dpi)
{
   if (!rule_ || !sym_) return -1;
   cairo_t *ctx = rule_;
   p_cairo_save(ctx);
   p_cairo_scale(ctx, 72 / dpi, 72 / dpi);
   p_cairo_translate(ctx, x_px, y_px);
   p_cairo_rotate(ctx, angle);
   p_cairo_set_source_surface(ctx, sym_, sym_w_ / -2.0, sym_h_ / -2.0);
   p_cairo_paint(ctx);
   p_cairo_restore(ctx);
   return p_cairo_status(ctx);
}

/* cap_coord's background box (fillcolor=, src/smrules_cairo.c:2367-2370 with cap_rect :2188-2199) on the current rule's context:
 * translate to the anchor, rotate, then the box of the text (w x h pt at the aligned origin, h negative: upwards) scaled by
 * `scale` about its middle */
int cd_caption_box(double x_pt, double y_pt, double angle, double ox_pt, double oy_pt, double w, double h, double desc, double scale, uint32_t col)
{
   if (!rule_) return -1;
   cairo_t *ctx = rule_;
   double ws = w * scale, hs = h * scale;
   p_cairo_save(ctx);
   p_cairo_translate(ctx, x_pt, y_pt);
   p_cairo_rotate(ctx, angle);
   p_cairo_save(ctx);
   p_cairo_translate(ctx, (w - ws) * .5, desc + (h - hs) * .5);
   set_color(ctx, col);
   p_cairo_rectangle(ctx, ox_pt, oy_pt, ws, hs);
   p_cairo_fill(ctx);
   p_cairo_restore(ctx);
   p_cairo_restore(ctx);
   return p_cairo_status(ctx);
}
