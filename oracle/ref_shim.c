/* ref_shim.c -- TEST INFRASTRUCTURE. Our own few lines, compiled TOGETHER WITH the
 * reference's untouched source src/bspline_ctrl.c (read from /root/reference at build
 * time, never copied) into oracle/_ref/libref_bspline.so by oracle/Makefile, so that
 * the real control_points() (src/bspline_ctrl.c:64-105) can be called on chosen inputs
 * to pin the restatement in smr_oracle.c.
 * (src/rdata.c cannot be built this way: it needs `image_t`, which only the
 * autoconf-generated config.h defines -- configure.ac:55-57 -- so the projection
 * restatement is pinned by identities only; see DESIGN.md.)
 */
#include "bspline.h"

void ref_control_points(const double *g, const double *l, double f, double *c1, double *c2)
{
   line_t lg, ll; point_t p1, p2;
   lg.A.x = g[0]; lg.A.y = g[1]; lg.B.x = g[2]; lg.B.y = g[3];
   ll.A.x = l[0]; ll.A.y = l[1]; ll.B.x = l[2]; ll.B.y = l[3];
   control_points(&lg, &ll, &p1, &p2, f);
   c1[0] = p1.x; c1[1] = p1.y; c2[0] = p2.x; c2[1] = p2.y;
}
