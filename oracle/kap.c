/* kap.c -- TEST INFRASTRUCTURE: CPU restatement of the reference's KAP (BSB raster chart) writer for checking
 * smr_hip_canvas_write_kap.  Only tests/ and bench tools may use it.
 *
 *   orc_median_cut     the colour reduction of src/median_cut.c:229-264 (mc_median_cut) with its helpers :66-206: always the
 *                      block with the longest side is cut, at the median of its points sorted along that side; the sort is
 *                      glibc's qsort, which is a STABLE merge sort up to glibc 2.36 (the image here has 2.35) -- which of
 *                      the points that tie at the median go left decides both halves' boxes, so the order matters; the lower
 *                      half keeps the average it had as part of its parent (only the new block's is computed, :258).
 *                      Pinned against the reference's own file compiled here (oracle/_ref/libref_median.so).
 *   orc_nearest        mc_nearest_block_index :159-172 (L1 distance, the first of equals)
 *   orc_bsb_row        bsb_compress_row / bsb_compress_nb, src/smkap.c:53-114 (widthout == widthin, as save_kap calls it)
 *   orc_kap_body       save_kap :196-245 from "OST/1" on: palette lines, the binary marker, the rows, the offset table
 *   orc_bsb_decode_row the inverse of orc_bsb_row (our own; for round-trip properties at sizes the byte comparison skips)
 * smkap.c itself cannot be compiled here (smrender_dev.h wants the generated config.h): the row coder is restated only and
 * checked through the decoder's round trip -- "parity unpinned" for the bytes of a row beyond that.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <arpa/inet.h>

typedef struct { uint8_t lo[3], hi[3], avg[3]; uint8_t *p; int len; } blk_t;

static int side_index(const blk_t *b)
{
   int m = 0, mi = 0;
   for (int i = 0; i < 3; i++) { int d = b->hi[i] - b->lo[i]; if (d > m) { m = d; mi = i; } }
   return mi;
}
static int side_len(const blk_t *b) { int i = side_index(b); return b->hi[i] - b->lo[i]; }

static void shrink(blk_t *b)
{
   memcpy(b->lo, b->p, 3); memcpy(b->hi, b->p, 3);
   for (int i = 1; i < b->len; i++)
      for (int j = 0; j < 3; j++)
      {
         uint8_t v = b->p[i * 3 + j];
         if (v < b->lo[j]) b->lo[j] = v;
         if (v > b->hi[j]) b->hi[j] = v;
      }
}
static void average(blk_t *b)
{
   for (int j = 0; j < 3; j++)
   {
      long s = 0;
      for (int i = 0; i < b->len; i++) s += b->p[i * 3 + j];
      b->avg[j] = (uint8_t) (s / b->len);
   }
}

/* stable counting sort of n 3-byte points by byte d */
static void sort_points(uint8_t *p, int n, int d)
{
   size_t cnt[257] = { 0 };
   uint8_t *t = malloc((size_t) n * 3);
   for (int i = 0; i < n; i++) cnt[p[i * 3 + d] + 1]++;
   for (int i = 0; i < 256; i++) cnt[i + 1] += cnt[i];
   for (int i = 0; i < n; i++) memcpy(t + 3 * cnt[p[i * 3 + d]]++, p + i * 3, 3);
   memcpy(p, t, (size_t) n * 3);
   free(t);
}

int orc_median_cut(uint8_t *points, unsigned num, unsigned desired, uint8_t *avg, int *len)
{
   blk_t *b = calloc(desired, sizeof(*b));
   int size = 1;
   b[0].p = points; b[0].len = (int) num;
   shrink(&b[0]); average(&b[0]);
   while (size < (int) desired && b[0].len > 1)
   {
      sort_points(b[0].p, b[0].len, side_index(&b[0]));
      int half = (b[0].len + 1) / 2;
      b[size] = b[0];                       /* (the new block's stale fields are all overwritten below) */
      b[size].p = b[0].p + 3 * half; b[size].len = b[0].len - half;
      b[0].len = half;
      shrink(&b[0]); shrink(&b[size]); average(&b[size]);
      size++;
      /* stable, longest side first */
      for (int i = 1; i < size; i++)
      {
         blk_t k = b[i];
         int j = i - 1, kl = side_len(&k);
         while (j >= 0 && side_len(&b[j]) < kl) { b[j + 1] = b[j]; j--; }
         b[j + 1] = k;
      }
   }
   for (int i = 0; i < size; i++) { memcpy(avg + 3 * i, b[i].avg, 3); len[i] = b[i].len; }
   free(b);
   return size;
}

int orc_nearest(const uint8_t *avg, int size, const uint8_t *pt)
{
   int best = 0x1000000, idx = 0;
   for (int i = 0; i < size; i++)
   {
      int d = abs(avg[3 * i] - pt[0]) + abs(avg[3 * i + 1] - pt[1]) + abs(avg[3 * i + 2] - pt[2]);
      if (d < best) { best = d; idx = i; }
   }
   return idx;
}

static int put_nb(uint8_t *p, unsigned nb, unsigned pixel, unsigned max)
{
   int n = 0;
   if (nb > max)
   {
      n = put_nb(p, nb >> 7, pixel | 0x80, max);
      p[n++] = (uint8_t) ((nb & 0x7f) | (pixel & 0x80));
      return n;
   }
   pixel |= nb;
   if (!(pixel & 0xff)) p[n++] = 0x80;
   p[n++] = (uint8_t) pixel;
   return n;
}

/* one row of palette indices -> BSB run-length bytes; returns their number (out: at least 2 w + 8 bytes) */
int orc_bsb_row(const uint8_t *in, int w, int depth, int line, uint8_t *out)
{
   const int dec = 7 - depth, max = (1 << dec) - 1;
   int n = put_nb(out, (unsigned) line, 0, 0x7f), x = 0;
   while (x < w)
   {
      uint8_t c = in[x++];
      unsigned run = 0;
      while (x < w && in[x] == c) { x++; run++; }
      n += put_nb(out + n, run, (unsigned) ((c << dec) & 0xff), (unsigned) max);
   }
   out[n++] = 0;
   return n;
}

/* the inverse: returns the bytes consumed (0 on a malformed row); *line and w indices out */
int orc_bsb_decode_row(const uint8_t *in, int n_in, int w, int depth, int *line, uint8_t *out)
{
   const int dec = 7 - depth, max = (1 << dec) - 1;
   int i = 0, x = 0;
   unsigned v = 0;
   do { if (i >= n_in) return 0; v = (v << 7) | (in[i] & 0x7f); } while (in[i++] & 0x80);
   *line = (int) v;
   while (x < w)
   {
      if (i >= n_in) return 0;
      uint8_t b = in[i++];
      unsigned c = (b & 0x7f) >> dec, run = b & max;
      while (b & 0x80) { if (i >= n_in) return 0; b = in[i++]; run = (run << 7) | (b & 0x7f); }
      for (unsigned k = 0; k <= run && x < w; k++) out[x++] = (uint8_t) c;
   }
   if (i >= n_in || in[i++] != 0) return 0;
   return i;
}

static int depth_of(int n) { int d, m; for (d = 1, m = 2; m < n; m <<= 1, d++); return d; }

/* save_kap from "OST/1" on, for ARGB32 pixels (row-major, w * h): what follows a header of header_len bytes.
 * out must hold 64 + 127 * 24 + h * (2 w + 8) + 4 (h + 1) bytes.  palette: [127] 0xffRRGGBB.  Returns the byte count */
long orc_kap_body(const uint32_t *pix, int w, int h, long header_len, uint8_t *out, uint32_t *palette, int *n_colors)
{
   const size_t n = (size_t) w * h;
   uint8_t *pts = malloc(n * 3), avg[127 * 3];
   int len[127];
   for (size_t i = 0; i < n; i++) { pts[3 * i] = pix[i] & 0xff; pts[3 * i + 1] = (pix[i] >> 8) & 0xff; pts[3 * i + 2] = (pix[i] >> 16) & 0xff; }
   const int nc = orc_median_cut(pts, (unsigned) n, 127, avg, len);
   free(pts);
   for (int i = 0; i < nc; i++) palette[i] = 0xff000000u | avg[3 * i] | (uint32_t) avg[3 * i + 1] << 8 | (uint32_t) avg[3 * i + 2] << 16;
   *n_colors = nc;
   const int d = depth_of(nc);
   long o = 0;
   o += sprintf((char *) out + o, "OST/1\r\nIFM/%d\r\n", d);
   for (int i = 0; i < nc; i++) o += sprintf((char *) out + o, "RGB/%d,%d,%d,%d\r\n", i + 1, (palette[i] >> 16) & 0xff, (palette[i] >> 8) & 0xff, palette[i] & 0xff);
   out[o++] = 0x1a; out[o++] = 0; out[o++] = (uint8_t) d;
   int32_t *offp = malloc(sizeof(int32_t) * (h + 1));
   uint8_t *row = malloc(w);
   for (int y = 0; y < h; y++)
   {
      offp[y] = (int32_t) htonl((uint32_t) (header_len + o));
      for (int x = 0; x < w; x++)
      {
         /* the reduced image holds the nearest block's colour; palette_index (:177-184) then finds the FIRST palette entry of
          * that colour */
         uint32_t p = pix[(size_t) y * w + x];
         uint8_t pt[3] = { (uint8_t) (p & 0xff), (uint8_t) ((p >> 8) & 0xff), (uint8_t) ((p >> 16) & 0xff) };
         uint32_t col = palette[orc_nearest(avg, nc, pt)];
         int k = 0;
         while (palette[k] != col) k++;
         row[x] = (uint8_t) (k + 1);
      }
      o += orc_bsb_row(row, w, d, y, out + o);
   }
   offp[h] = (int32_t) htonl((uint32_t) (header_len + o));
   memcpy(out + o, offp, sizeof(int32_t) * (h + 1));
   o += (long) sizeof(int32_t) * (h + 1);
   free(offp); free(row);
   return o;
}
