/* caprot.hip -- caption auto-rotation scan (placeholder until the kernel lands). */
#include "smr_dev.h"
#include "smr_host.h"

extern "C" int smr_hip_caprot_steps(int fg_w, int fg_h, int dpi, int *num_steps, int *r_px)
{
   if (fg_w <= 0 || fg_h <= 0 || dpi <= 0) return smr_fail(SMR_E_INVAL, "caprot_steps: bad size");
   double r_pt = hypot((double) fg_w, (double) fg_h) * 72 / dpi;            /* find_angle :1975 */
   if (num_steps) *num_steps = (int) round(r_pt * M_PI * 1.0 * 25.4 / 72.0); /* :1978 */
   if (r_px) *r_px = (int) round(r_pt * ((double) dpi / 72.0));              /* cairo_smr_cut_out :1516 */
   return SMR_OK;
}

extern "C" int smr_hip_caprot(smr_canvas_t *c, const smr_cap_batch_t *b, double *d_angle, double *d_dv)
{
   SMR_NEED_INIT();
   return smr_fail(SMR_E_UNSUPPORTED, "caprot: kernel not built yet");
}
