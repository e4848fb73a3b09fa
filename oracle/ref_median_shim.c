/* ref_median_shim.c -- TEST INFRASTRUCTURE.  Our own few lines around the reference's untouched src/median_cut.c, which the
 * compiler reads from where it lies (-I/root/reference/src; never copied) WITHOUT HAVE_CAIRO: the colour reduction itself
 * (mc_median_cut :229-264, mc_nearest_block_index :159-172) needs no cairo, but its functions are static, so this file
 * includes the source text instead of linking an object.  oracle/Makefile builds it into oracle/_ref/libref_median.so
 * (log_msg comes from the reference's libsmrender, oracle/_ref/libsmrender_ref.so).  It pins oracle/kap.c. */
#include "median_cut.c"

/* points: num * 3 bytes (x[0], x[1], x[2] = blue, green, red as mc_cairo_color_to_point :277-283 fills them); they are
 * reordered in place, as there.  avg: [desired * 3], len: [desired].  Returns the number of blocks */
int ref_median_cut(unsigned char *points, unsigned num, unsigned desired, unsigned char *avg, int *len)
{
   mc_block_t *blk = mc_block_list(desired);
   if (blk == NULL) return -1;
   int n = mc_median_cut((mc_point_t *) points, num, desired, blk);
   for (int i = 0; i < n; i++)
   {
      for (int j = 0; j < NUM_DIMENSIONS; j++) avg[i * 3 + j] = blk[i].avg.x[j];
      len[i] = blk[i].len;
   }
   free(blk);
   return n;
}

int ref_nearest_block(const unsigned char *avg, int size, const unsigned char *pt)
{
   mc_block_t blk[size > 0 ? size : 1];
   mc_point_t p;
   memset(blk, 0, sizeof(blk));
   for (int i = 0; i < size; i++)
      for (int j = 0; j < NUM_DIMENSIONS; j++) blk[i].avg.x[j] = avg[i * 3 + j];
   for (int j = 0; j < NUM_DIMENSIONS; j++) p.x[j] = pt[j];
   return mc_nearest_block_index(blk, size, &p);
}
