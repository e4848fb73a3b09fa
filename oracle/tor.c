/* tor.c -- TEST INFRASTRUCTURE (oracle); see tor.h for what it restates. */
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include "tor.h"

#define GRID_X 256
#define GRID_Y 15
#define AREA_TO_ALPHA(c) (((c) + ((c) << 4) + 256) >> 9)

typedef struct { int64_t quo, rem; } quorem_t;

typedef struct
{
   int ytop, ybot;          /* sub-row range [ytop, ybot) */
   int dir;
   int vertical;
   int32_t p1x;
   int64_t p1y;             /* 24.8 */
   int64_t Ex, Ey;
   quorem_t dxdy, dxdy_full;
   /* per-row scratch */
   quorem_t x; int64_t cell; quorem_t xn; int64_t celln;
   int ncut, cut[2];
   int ytop_raw;               /* ytop before it is clamped to the first row */
   int32_t topx, topy;         /* the edge's upper end point */
} edge_t;

static int to_grid_y(int64_t y) { return (int) ((GRID_Y * y + 128) >> 8); }

static int lim_on_;
static int64_t lim_x0_, lim_x1_;
void tor_set_limits(int on, int32_t x_left, int32_t x_right) { lim_on_ = on; lim_x0_ = x_left; lim_x1_ = x_right; }

/* The edges come in the order in which the path added them to libcairo's polygon (a fill: one edge per segment of the path).  Then
 * two edges that START at a row's first sub-row in the same cell, the two sides of a vertex a hair above the row's first sample line,
 * stand in libcairo's active list in that order (merge_sorted_edges keeps ties as they come), and when it is right side before left
 * side their cells at the row's bottom come out descending: can_do_full_row says no, the row is sub-sampled.  Seen on libcairo 1.16
 * itself (6 000 triangles with such a vertex: tests/test_oracle_cairo.py); off for edges in an order of our own (strokes). */
static int path_order_;
void tor_set_path_order(int on) { path_order_ = on; }

static int64_t floor_div(int64_t a, int64_t b)
{
   int64_t q = a / b;
   if ((a % b != 0) && ((a < 0) != (b < 0))) q--;
   return q;
}
static int64_t y_for_x(int64_t p1x, int64_t p1y, int64_t p2x, int64_t p2y, int64_t x)
{
   if (x == p1x) return p1y;
   if (x == p2x) return p2y;
   int64_t dx = p2x - p1x;
   return dx ? p1y + floor_div((x - p1x) * (p2y - p1y), dx) : p1y;
}
static int64_t x_for_y(int64_t p1x, int64_t p1y, int64_t p2x, int64_t p2y, int64_t y)
{
   if (y == p1y) return p1x;
   if (y == p2y) return p2x;
   int64_t dy = p2y - p1y;
   return dy ? p1x + floor_div((y - p1y) * (p2x - p1x), dy) : p1x;
}
static int limit_cuts(int64_t p1x, int64_t p1y, int64_t p2x, int64_t p2y, int64_t *cut)
{
   int n = 0;
   if (!lim_on_) return 0;
   const int64_t L = lim_x0_, R = lim_x1_, pleft = p1x < p2x ? p1x : p2x, pright = p1x < p2x ? p2x : p1x;
   if ((L <= pleft && pright <= R) || pright <= L || R <= pleft) return 0;
   int64_t top_y = p1y, bot_y = p2y, left_y, right_y;
   if (p1x <= p2x)
   {
      if (pleft >= L) left_y = top_y;
      else { left_y = y_for_x(p1x, p1y, p2x, p2y, L); if (x_for_y(p1x, p1y, p2x, p2y, left_y) < L) left_y++; }
      if (left_y > bot_y) left_y = bot_y;
      if (top_y < left_y) { cut[n++] = left_y; top_y = left_y; }
      if (pright <= R) right_y = bot_y;
      else { right_y = y_for_x(p1x, p1y, p2x, p2y, R); if (x_for_y(p1x, p1y, p2x, p2y, right_y) > R) right_y--; }
      if (right_y < top_y) right_y = top_y;
      if (bot_y > right_y) cut[n++] = right_y;
   }
   else
   {
      if (pright <= R) right_y = top_y;
      else { right_y = y_for_x(p1x, p1y, p2x, p2y, R); if (x_for_y(p1x, p1y, p2x, p2y, right_y) > R) right_y++; }
      if (right_y > bot_y) right_y = bot_y;
      if (top_y < right_y) { cut[n++] = right_y; top_y = right_y; }
      if (pleft >= L) left_y = bot_y;
      else { left_y = y_for_x(p1x, p1y, p2x, p2y, L); if (x_for_y(p1x, p1y, p2x, p2y, left_y) < L) left_y--; }
      if (left_y < top_y) left_y = top_y;
      if (bot_y > left_y) cut[n++] = left_y;
   }
   return n;
}

/* x at the centre of sub-row ys, exact: quo + rem/Ey, 0 <= rem < Ey */
static quorem_t edge_x(const edge_t *e, int ys)
{
   quorem_t x;
   if (e->vertical) { x.quo = e->p1x; x.rem = 0; return x; }
   int64_t tmp = ((int64_t) (2 * (int64_t) ys + 1) << 8) - e->p1y * GRID_Y * 2;
   tmp *= e->Ex;
   x.quo = tmp / e->Ey;
   x.rem = tmp % e->Ey;
   x.quo += e->p1x;
   if (x.rem < 0) { x.quo--; x.rem += e->Ey; }
   else if (x.rem >= e->Ey) { x.quo++; x.rem -= e->Ey; }
   return x;
}

static int64_t cell_of(const edge_t *e, quorem_t x)
{
   return e->vertical ? x.quo : x.quo + (x.rem >= e->Ey / 2);
}

static void norm(quorem_t *x, int64_t dy)
{
   if (x->rem < 0) { x->quo--; x->rem += dy; }
   else if (x->rem >= dy) { x->quo++; x->rem -= dy; }
}

typedef struct { int W; int64_t left_cover; int32_t *cover, *uarea; } cells_t;

static void cell_add(cells_t *c, int64_t ix, int dcover, int duarea)
{
   if (ix < 0) { c->left_cover += dcover; return; }
   if (ix >= c->W) return;
   c->cover[ix] += dcover;
   c->uarea[ix] += duarea;
}

static void add_subspan(cells_t *c, int64_t x1, int64_t x2)
{
   if (x1 == x2) return;
   int64_t ix1 = x1 >> 8, ix2 = x2 >> 8;
   int fx1 = (int) (x1 & 255), fx2 = (int) (x2 & 255);
   if (ix1 != ix2)
   {
      cell_add(c, ix1, 1, 2 * fx1);
      cell_add(c, ix2, -1, -2 * fx2);
   }
   else
      cell_add(c, ix1, 0, 2 * (fx1 - fx2));
}

/* exact coverage of one edge across a full pixel row (sign +1 left edge, -1 right edge) */
static void render_edge(cells_t *c, const edge_t *e, quorem_t x1, quorem_t x2, int sign)
{
   if (!e->vertical)
   {
      x1.quo -= e->dxdy.quo / 2; x1.rem -= e->dxdy.rem / 2; norm(&x1, e->Ey);
      x2.quo -= e->dxdy.quo / 2; x2.rem -= e->dxdy.rem / 2; norm(&x2, e->Ey);
   }
   int64_t ix1 = x1.quo >> 8, ix2 = x2.quo >> 8;
   int fx1 = (int) (x1.quo & 255), fx2 = (int) (x2.quo & 255);
   if (ix1 == ix2)
   {
      cell_add(c, ix1, sign * GRID_Y, sign * (fx1 + fx2) * GRID_Y);
      return;
   }
   if (ix2 < ix1)
   {
      int64_t t = ix1; ix1 = ix2; ix2 = t;
      int f = fx1; fx1 = fx2; fx2 = f;
      quorem_t tx = x1; x1 = x2; x2 = tx;
   }
   int64_t dx = (x2.quo - x1.quo) * e->Ey + (x2.rem - x1.rem);
   int64_t tmp = (ix1 + 1) * GRID_X * e->Ey;
   tmp -= x1.quo * e->Ey + x1.rem;
   tmp *= GRID_Y;
   quorem_t y; y.quo = tmp / dx; y.rem = tmp % dx;
   cell_add(c, ix1, (int) (sign * y.quo), (int) (sign * y.quo * (GRID_X + fx1)));
   int64_t y_last = y.quo;
   if (ix1 + 1 < ix2)
   {
      quorem_t full;
      full.quo = GRID_Y * GRID_X * e->Ey / dx;
      full.rem = GRID_Y * GRID_X * e->Ey % dx;
      ++ix1;
      do
      {
         y.quo += full.quo; y.rem += full.rem;
         if (y.rem >= dx) { y.quo++; y.rem -= dx; }
         cell_add(c, ix1, (int) (sign * (y.quo - y_last)), (int) (sign * (y.quo - y_last) * GRID_X));
         y_last = y.quo;
         ++ix1;
      } while (ix1 != ix2);
   }
   cell_add(c, ix2, (int) (sign * (GRID_Y - y_last)), (int) (sign * (GRID_Y - y_last) * fx2));
}

static int cmp_ys0_;
static int cmp_row(const void *a, const void *b)
{
   const edge_t *p = *(edge_t * const *) a, *q = *(edge_t * const *) b;
   if (p->cell != q->cell) return p->cell < q->cell ? -1 : 1;
   /* the two sides of one vertex at the row's top: as the path added them (tor_set_path_order) */
   if (path_order_ && p->ytop_raw == cmp_ys0_ && q->ytop_raw == cmp_ys0_ && p->topx == q->topx && p->topy == q->topy) return p < q ? -1 : p > q;
   if (p->celln != q->celln) return p->celln < q->celln ? -1 : 1;
   return 0;
}

typedef struct { int64_t cell; int dir; } xing_t;
static int cmp_xing(const void *a, const void *b)
{
   const xing_t *p = a, *q = b;
   return p->cell < q->cell ? -1 : p->cell > q->cell;
}

void tor_coverage(const tor_edge_t *in, int n, int nonzero, int x0, int y0, int W, int H, uint8_t *alpha)
{
   edge_t *ed = malloc(sizeof(*ed) * (n ? n : 1));
   edge_t **act = malloc(sizeof(*act) * (n ? n : 1));
   xing_t *xs = malloc(sizeof(*xs) * (n ? n : 1));
   int32_t *cover = malloc(sizeof(int32_t) * W), *uarea = malloc(sizeof(int32_t) * W);
   int m = 0;
   unsigned mask = nonzero ? ~0u : 1u;
   memset(alpha, 0, (size_t) W * H);

   for (int i = 0; i < n; i++)
   {
      /* _cairo_polygon_add_edge: drop horizontal, orient downwards */
      int32_t ax = in[i].x1 - x0 * 256, ay = in[i].y1 - y0 * 256, bx = in[i].x2 - x0 * 256, by = in[i].y2 - y0 * 256;
      int dir = 1;
      if (ay == by) continue;
      if (ay > by) { int32_t t = ax; ax = bx; bx = t; t = ay; ay = by; by = t; dir = -1; }
      edge_t *e = &ed[m];
      e->ytop = to_grid_y(ay); e->ybot = to_grid_y(by);
      e->ytop_raw = e->ytop; e->topx = ax; e->topy = ay;
      if (e->ytop < 0) e->ytop = 0;
      if (e->ybot > H * GRID_Y) e->ybot = H * GRID_Y;
      if (e->ybot <= e->ytop) continue;
      e->dir = dir; e->p1x = ax; e->p1y = ay;
      {
         int64_t cy[2];
         e->ncut = limit_cuts((int64_t) ax + (int64_t) x0 * 256, (int64_t) ay + (int64_t) y0 * 256, (int64_t) bx + (int64_t) x0 * 256, (int64_t) by + (int64_t) y0 * 256, cy);
         for (int k = 0; k < e->ncut; k++) e->cut[k] = to_grid_y(cy[k] - (int64_t) y0 * 256);
      }
      if (ax == bx)
      {
         e->vertical = 1; e->Ex = 0; e->Ey = 0;
         e->dxdy.quo = e->dxdy.rem = e->dxdy_full.quo = e->dxdy_full.rem = 0;
      }
      else
      {
         e->vertical = 0;
         e->Ex = (int64_t) (bx - ax) * GRID_X;
         e->Ey = (int64_t) (by - ay) * GRID_Y * 512;
         e->dxdy.quo = e->Ex * 512 / e->Ey; e->dxdy.rem = e->Ex * 512 % e->Ey;
         int64_t tmp = e->Ex * (2 * GRID_Y << 8);
         e->dxdy_full.quo = tmp / e->Ey; e->dxdy_full.rem = tmp % e->Ey;
      }
      m++;
   }

   for (int row = 0; row < H; row++)
   {
      int ys0 = row * GRID_Y, ys1 = ys0 + GRID_Y, na = 0, partial = 0;
      for (int i = 0; i < m; i++)
      {
         edge_t *e = &ed[i];
         if (e->ytop >= ys1 || e->ybot <= ys0) continue;
         act[na++] = e;
         if (e->ytop > ys0 || e->ybot < ys1) partial = 1;       /* starts below the row's first sub-row, or ends inside the row */
         for (int k = 0; k < e->ncut; k++) if (e->cut[k] > ys0 && e->cut[k] < ys1) partial = 1;
      }
      if (!na) continue;
      cells_t c = { W, 0, cover, uarea };
      memset(cover, 0, sizeof(int32_t) * W); memset(uarea, 0, sizeof(int32_t) * W);

      int full = !partial;
      if (full)
      {
         for (int i = 0; i < na; i++)
         {
            edge_t *e = act[i];
            e->x = edge_x(e, ys0); e->cell = cell_of(e, e->x);
            e->xn = edge_x(e, ys1); e->celln = cell_of(e, e->xn);
            if (lim_on_ && getenv("ORC_NO_PROJECT") == NULL)
            {
               /* what lies beyond a side of the limits libcairo has projected onto that side (_add_clipped_edge: a vertical edge
                * there) before the scan converter looks: two edges that cross out there do not cross for it.  An edge is that
                * vertical piece in this row when it is beyond the side at the row's first AND last sample; one that reaches the side
                * at the row's bottom is still its own line here, and the cell libcairo steps it to lies beyond the side.  (An edge
                * that crosses a side inside this row has a cut here, and the row is sub-sampled anyway.) */
               const int64_t L = lim_x0_ - (int64_t) x0 * 256, R = lim_x1_ - (int64_t) x0 * 256;
               if (e->cell < L || e->cell > R)
               {
                  const int64_t last = cell_of(e, edge_x(e, ys1 - 1));
                  if (e->cell < L && last < L) e->cell = e->celln = L;
                  else if (e->cell > R && last > R) e->cell = e->celln = R;
               }
            }
         }
         cmp_ys0_ = ys0;
         qsort(act, na, sizeof(*act), cmp_row);
         int64_t prev = INT64_MIN;
         for (int i = 0; i < na && full; i++)
         {
            if (act[i]->celln < prev) full = 0;
            prev = act[i]->celln;
         }
      }
      if (full)
      {
         int i = 0;
         while (i < na)
         {
            edge_t *left = act[i];
            int winding = left->dir;
            int j = i + 1;
            for (; j < na; j++)
            {
               winding += act[j]->dir;
               if ((winding & mask) == 0 && (j + 1 >= na || act[j + 1]->cell != act[j]->cell)) break;
            }
            if (j >= na) j = na - 1;   /* unbalanced input: close at the last edge */
            render_edge(&c, left, left->x, left->xn, +1);
            if (j > i) render_edge(&c, act[j], act[j]->x, act[j]->xn, -1);
            i = j + 1;
         }
      }
      else
      {
         for (int ys = ys0; ys < ys1; ys++)
         {
            int k = 0;
            for (int i = 0; i < na; i++)
            {
               edge_t *e = act[i];
               if (ys < e->ytop || ys >= e->ybot) continue;
               xs[k].cell = cell_of(e, edge_x(e, ys)); xs[k].dir = e->dir; k++;
            }
            qsort(xs, k, sizeof(*xs), cmp_xing);
            int winding = 0; int64_t xstart = INT64_MIN;
            for (int i = 0; i < k; i++)
            {
               int64_t xend = xs[i].cell;
               winding += xs[i].dir;
               if ((winding & mask) == 0)
               {
                  if (i + 1 >= k || xs[i + 1].cell != xend)
                  {
                     if (xstart != INT64_MIN) add_subspan(&c, xstart, xend);
                     xstart = INT64_MIN;
                  }
               }
               else if (xstart == INT64_MIN)
                  xstart = xend;
            }
         }
      }

      /* blit_a8: cells -> alpha */
      int32_t cov = (int32_t) (c.left_cover * GRID_X * 2);
      uint8_t *out = alpha + (size_t) row * W;
      for (int x = 0; x < W; x++)
      {
         cov += cover[x] * GRID_X * 2;
         int32_t area = cov - uarea[x];
         int a = AREA_TO_ALPHA(area);
         out[x] = (uint8_t) (a < 0 ? 0 : a > 255 ? 255 : a);
      }
   }
   free(ed); free(act); free(xs); free(cover); free(uarea);
}
