/* pngread.c -- TEST INFRASTRUCTURE (oracle): the reconstruction side of PNG's row filters (PNG specification, section 9
 * "Filtering": None, Sub, Up, Average, Paeth) for 8-bit samples, so that tests can decode the files smr_hip_canvas_write_png
 * and libcairo's own writer produce and compare PIXELS.  The container and zlib are handled in pyoracle.png_decode. */
#include <stdint.h>
#include <stdlib.h>

static int paeth(int a, int b, int c)
{
   int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
   return pa <= pb && pa <= pc ? a : pb <= pc ? b : c;
}

/* in: h rows of 1 + rowbytes (filter byte first); out: h rows of rowbytes.  Returns 0, or -1 for an unknown filter type.
 * hist[5] (may be NULL) counts the rows per filter type. */
int orc_png_unfilter(const uint8_t *in, int64_t rowbytes, int64_t h, int bpp, uint8_t *out, int64_t *hist)
{
   for (int64_t y = 0; y < h; y++)
   {
      const uint8_t *s = in + y * (rowbytes + 1);
      uint8_t *d = out + y * rowbytes;
      const uint8_t *u = y ? d - rowbytes : NULL;
      const int f = s[0];
      if (f > 4) return -1;
      if (hist) hist[f]++;
      s++;
      for (int64_t i = 0; i < rowbytes; i++)
      {
         const int a = i >= bpp ? d[i - bpp] : 0, b = u ? u[i] : 0, c = u && i >= bpp ? u[i - bpp] : 0;
         int pred = 0;
         switch (f)
         {
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: pred = paeth(a, b, c); break;
         }
         d[i] = (uint8_t) (s[i] + pred);
      }
   }
   return 0;
}
