/* tor.h -- TEST INFRASTRUCTURE (oracle). CPU restatement of the scan converter that
 * produces the reference's pixels: libcairo 1.16.0 `cairo-tor-scan-converter.c`
 * (third-party, un-vendored: /root/reference/configure.ac:52 pulls "cairo" unpinned;
 * the image holds 1.16.0 + pixman 0.40.0).  The reference reaches it through
 * cairo_fill()/cairo_stroke() at src/smrules_cairo.c:976,984,1007,1012,1134,1138 and
 * the replay at :403-407.  Algorithm restated from its published description:
 * 24.8 fixed-point input, sample grid 256 x 15 per pixel, edges sampled at sub-row
 * centres on the unquantised line, non-zero / even-odd span extraction per sub-row,
 * exact "full row" trapezoid path when no edge starts/ends/crosses in a pixel row,
 * area->alpha = (17*a + 256) >> 9.   Pinned against libcairo by tests/test_oracle_cairo.py.
 */
#ifndef SMR_ORACLE_TOR_H
#define SMR_ORACLE_TOR_H
#include <stdint.h>

typedef struct { int32_t x1, y1, x2, y2; } tor_edge_t;   /* 24.8 fixed, direction = winding sign */

/* Rasterise `n` edges into an 8-bit coverage window: W x H pixels whose top-left
 * pixel is (x0,y0) in device space.  nonzero: 1 = winding rule, 0 = even-odd. */
void tor_set_path_order(int on);      /* the edges of the next tor_coverage calls are a fill's, in the path's order (tor.c) */
void tor_coverage(const tor_edge_t *e, int n, int nonzero, int x0, int y0, int W, int H, uint8_t *alpha);
void tor_set_limits(int on, int32_t x_left, int32_t x_right);

#endif
