/* png.hip -- PNG files and slippy-map tiles straight from the device canvas.
 *
 * Stands in for the image branch of save_main_image (src/smrules_cairo.c:398-412: ARGB32 image ->
 * cairo_surface_write_to_png_stream) and for the tile writer (create_tiles src/smtile.c:150-228 with
 * save_image src/smrules_cairo.c:462-473).  What libcairo's PNG writer (third party, 1.16: cairo-png.c write_png)
 * does with an ARGB32 image is restated here: an image whose every alpha is 255 is written as 8-bit RGB, any
 * other as 8-bit RGBA with every colour un-premultiplied as (c*255 + a/2)/a, a == 0 giving four zero bytes.
 * The DECODED pixels of our file are those of cairo's file (tests/test_gpu_png.py); the compressed bytes are not
 * meant to be (libpng's filter heuristic and zlib's block splitting are not restated).
 *
 * Division of labour:
 *   k_png_opaque   one pass over the rectangle(s): is every alpha 255?  (cairo's _cairo_image_analyze_transparency)
 *   k_png_rows     one workgroup per (job, row): un-premultiply, try the five PNG filters on the row (minimum sum of
 *                  absolute residuals, the heuristic PNG's specification recommends), write filter byte + residuals.
 *                  HBM-bound byte work: 4 B in, 3 or 4 B out per pixel, the neighbours come from L2.
 *   host           the filtered rows cross PCIe ONCE, in strips of <= 64 MB through two pinned buffers, while worker
 *                  threads deflate the previous strip (zlib, independent raw-deflate chunks joined by sync flushes,
 *                  Adler-32 combined) or, for tiles, one whole PNG per worker at a time.
 * A job is a w x h rectangle of the band; a PNG of a region is one job cut into strips, a tile set is nx*ny jobs. */
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include <errno.h>
#include <string.h>
#include <sys/stat.h>
#include "smr_host.h"
#include "smr_dev.h"

#define SMR_PNG_STRIP_BYTES (64u << 20)

struct png_args_t
{
   const uint32_t *pix; int W;       /* the band: W pixels per row */
   int x0, y0;                       /* first job's corner, band-local */
   int w, h;                         /* every job's size */
   int nx;                           /* jobs per row of jobs */
   int row0, nrows;                  /* rows of EVERY job handled by this launch (a strip of a one-job image, or 0..h) */
   int job0, njobs;                  /* jobs of this launch */
   const uint32_t *opaque;           /* per job (index from 0 of the whole call) */
   unsigned char *out;               /* row (j - job0, r - row0) at ((j - job0) * nrows + r - row0) * stride + 3 */
   uint32_t stride;                  /* 4 + 4 * w: three pad bytes put the pixel bytes on a dword boundary */
};

__global__ __launch_bounds__(256) void k_png_opaque(const uint32_t *__restrict__ pix, int W, int x0, int y0, int w, int h, int nx,
                                                     int njobs, uint32_t *__restrict__ opaque)
{
   const int j = blockIdx.x / h, r = blockIdx.x % h;
   if (j >= njobs) return;
   const uint32_t *row = pix + (size_t) (y0 + (j / nx) * h + r) * W + x0 + (j % nx) * w;
   uint32_t all = 0xff000000u;
   for (int x = threadIdx.x; x < w; x += 256) all &= row[x];
   if (__any(all != 0xff000000u) && (threadIdx.x & 63) == 0) atomicAnd(&opaque[j], 0u);
}

/* cairo-png.c unpremultiply_data; the result as bytes R, G, B, A from the low byte up */
__device__ __forceinline__ uint32_t png_unpremultiply(uint32_t p)
{
   const uint32_t a = p >> 24;
   if (a == 255u) return (p & 0xff00ff00u) | ((p >> 16) & 0xffu) | ((p & 0xffu) << 16);
   if (a == 0u) return 0u;
   const uint32_t r = ((((p >> 16) & 0xffu) * 255u + a / 2u) / a) & 0xffu, g = ((((p >> 8) & 0xffu) * 255u + a / 2u) / a) & 0xffu,
                  b = (((p & 0xffu) * 255u + a / 2u) / a) & 0xffu;
   return r | g << 8 | b << 16 | a << 24;
}

__device__ __forceinline__ int png_paeth(int a, int b, int c)
{
   const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
   return pa <= pb && pa <= pc ? a : pb <= pc ? b : c;
}

/* residual of byte x (left a, up b, up-left c) under filter f */
__device__ __forceinline__ uint32_t png_residual(int f, int x, int a, int b, int c)
{
   int pred = 0;
   switch (f)
   {
      case 1: pred = a; break;
      case 2: pred = b; break;
      case 3: pred = (a + b) >> 1; break;
      case 4: pred = png_paeth(a, b, c); break;
   }
   return (uint32_t) (x - pred) & 0xffu;
}

__global__ __launch_bounds__(256) void k_png_rows(png_args_t A)
{
   __shared__ uint32_t part[4][5];
   __shared__ int chosen;
   const int jl = blockIdx.x / A.nrows, r = A.row0 + blockIdx.x % A.nrows;     /* job of this launch, row of the job */
   const int j = A.job0 + jl;
   const int bpp = A.opaque[j] ? 3 : 4;
   const uint32_t *cur = A.pix + (size_t) (A.y0 + (j / A.nx) * A.h + r) * A.W + A.x0 + (j % A.nx) * A.w;
   const uint32_t *up = cur - A.W;
   const bool has_up = r > 0;                     /* PNG: the row above the first one is all zero */
   unsigned char *dst = A.out + ((size_t) jl * A.nrows + (r - A.row0)) * A.stride + 3;
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

   /* pass 1: sum of |residual| (as signed bytes) of the five filters */
   uint32_t sum[5] = { 0, 0, 0, 0, 0 };
   for (int x = tid; x < A.w; x += 256)
   {
      const uint32_t X = png_unpremultiply(cur[x]), L = x ? png_unpremultiply(cur[x - 1]) : 0u,
                     U = has_up ? png_unpremultiply(up[x]) : 0u, UL = has_up && x ? png_unpremultiply(up[x - 1]) : 0u;
      for (int ch = 0; ch < bpp; ch++)
      {
         const int xv = (X >> (8 * ch)) & 0xff, a = (L >> (8 * ch)) & 0xff, b = (U >> (8 * ch)) & 0xff, c = (UL >> (8 * ch)) & 0xff;
#pragma unroll
         for (int f = 0; f < 5; f++)
         {
            const uint32_t v = png_residual(f, xv, a, b, c);
            sum[f] += v < 128u ? v : 256u - v;
         }
      }
   }
#pragma unroll
   for (int f = 0; f < 5; f++)
   {
      uint32_t s = sum[f];
      for (int o = 32; o; o >>= 1) s += __shfl_down(s, o);
      if (lane == 0) part[wave][f] = s;
   }
   __syncthreads();
   if (tid == 0)
   {
      int best = 0; uint32_t bs = 0xffffffffu;
      for (int f = 0; f < 5; f++)
      {
         const uint32_t s = part[0][f] + part[1][f] + part[2][f] + part[3][f];
         if (s < bs) { bs = s; best = f; }
      }
      chosen = best;
      dst[0] = (unsigned char) best;
   }
   __syncthreads();
   const int f = chosen;

   /* pass 2: the residuals of the chosen filter; four pixels per thread so that the stores are whole dwords */
   unsigned char *px = dst + 1;                    /* dword aligned: stride % 4 == 0, +3 +1 */
   for (int g = tid; g * 4 < A.w; g += 256)
   {
      const int xb = g * 4, n = min(4, A.w - xb);
      uint32_t res[4];
      uint32_t L = xb ? png_unpremultiply(cur[xb - 1]) : 0u, UL = has_up && xb ? png_unpremultiply(up[xb - 1]) : 0u;
      for (int k = 0; k < 4; k++)
      {
         uint32_t X = 0, U = 0;
         if (k < n) { X = png_unpremultiply(cur[xb + k]); U = has_up ? png_unpremultiply(up[xb + k]) : 0u; }
         uint32_t o = 0;
         for (int ch = 0; ch < bpp; ch++)
            o |= png_residual(f, (X >> (8 * ch)) & 0xff, (L >> (8 * ch)) & 0xff, (U >> (8 * ch)) & 0xff, (UL >> (8 * ch)) & 0xff) << (8 * ch);
         res[k] = o; L = X; UL = U;
      }
      if (n == 4)
      {
         uint32_t *o32 = (uint32_t *) (px + (size_t) xb * bpp);
         if (bpp == 4) { o32[0] = res[0]; o32[1] = res[1]; o32[2] = res[2]; o32[3] = res[3]; }
         else
         {
            o32[0] = (res[0] & 0xffffffu) | res[1] << 24;
            o32[1] = ((res[1] >> 8) & 0xffffu) | res[2] << 16;
            o32[2] = ((res[2] >> 16) & 0xffu) | res[3] << 8;
         }
      }
      else
         for (int k = 0; k < n; k++)
            for (int ch = 0; ch < bpp; ch++) px[(size_t) (xb + k) * bpp + ch] = (unsigned char) (res[k] >> (8 * ch));
   }
}

/* ---------------------------------------------------------------------------------------- */
/* deflate on the device (level SMR_PNG_DEVICE_DEFLATE)                                            */
/* ---------------------------------------------------------------------------------------- */
/* zlib on sixteen host threads takes nine times as long as the chart took to draw, and the filtered rows cross PCIe for it.
 * Here the rows stay on the device: every strip becomes ONE deflate block with a Huffman code of its own (RFC 1951 3.2.7).
 * The matcher is what PNG's filters leave to do on a chart: runs.  A filtered row of flat colour is zeros, so a lane walks its
 * segment of a row (SMR_DFL_SEG bytes; segments never straddle rows, whose filter byte ends a run anyway) and turns every run of
 * a byte into the byte and one match of distance 1 (3..258 bytes).  Three passes of the same walk: (0) histogram of the
 * literal/length symbols (LDS, then global) and the strip's Adler-32 sums; the host builds the length-limited code from the 286
 * counts; (1) bits per segment with that code, summed per workgroup; a scan; (2) every lane writes its bits at its bit offset:
 * whole words of its own by plain stores, its first and last word (shared with its neighbours) by atomicOr into zeroed memory.
 * Only the compressed bytes cross PCIe; the host adds the block header, the end-of-block code and the CRCs of the IDAT chunks. */
#define SMR_DFL_SEG 256u
#define SMR_DFL_NSYM 286

struct dfl_args_t
{
   const unsigned char *rows;        /* row r of job j (of this launch) at rows + (j * nrows + r) * stride, its 1 + rowbytes bytes from +3 on */
   uint32_t stride, w, nrows;        /* bytes from row to row; pixels per row; rows per job */
   uint32_t bpj;                     /* workgroups per job: 256 segments each, the last ones of a job partly idle */
   const uint32_t *opaque;           /* per job: 3 or 4 bytes per pixel */
   const uint32_t *code;             /* [job][288] bits << 16 | the code, bit-reversed (as it goes into the stream, low bit first) */
   uint32_t *hist;                   /* [job][288] */
   unsigned long long *adler;        /* [job][2]: sum of bytes, sum of byte * (bytes from it to the job's end), both mod 65521 per lane */
   uint32_t *bits;                   /* [workgroup][256] */
   uint32_t *blocksum;               /* [workgroup] */
   const uint64_t *blockstart;       /* [workgroup], from the job's first segment */
   const uint64_t *jobbit0;          /* [job] where the job's first segment's bits start in out (behind its block's header) */
   uint32_t *out;
};

/* length 3..258 -> symbol, number of extra bits, their value (RFC 1951 3.2.5) */
__device__ __forceinline__ void dfl_len_sym(uint32_t m, uint32_t &sym, uint32_t &eb, uint32_t &ev)
{
   if (m == 258u) { sym = 285u; eb = 0; ev = 0; return; }
   const uint32_t l = m - 3u;
   if (l < 8u) { sym = 257u + l; eb = 0; ev = 0; return; }
   const uint32_t e = 29u - (uint32_t) __clz(l);
   sym = 261u + 4u * e + ((l >> e) - 4u); eb = e; ev = l & ((1u << e) - 1u);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_dfl(dfl_args_t A)
{
   __shared__ uint32_t lh[288];
   __shared__ uint32_t wsum[4];
   const uint32_t job = blockIdx.x / A.bpj, sl = (blockIdx.x - job * A.bpj) * 256u + threadIdx.x;      /* the job, the segment of the job */
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   if (MODE == 0) { for (int i = threadIdx.x; i < 288; i += 256) lh[i] = 0; __syncthreads(); }
   const uint32_t rowlen = 1u + A.w * (A.opaque[job] ? 3u : 4u), spr = (rowlen + SMR_DFL_SEG - 1u) / SMR_DFL_SEG;
   const bool on = sl < A.nrows * spr;
   const uint32_t *const code = A.code + (size_t) job * 288u;
   uint32_t nbits = 0;
   unsigned long long acc = 0; uint32_t nb = 0, wpos = 0, first_w = 0, last_w = 0;
   unsigned long long sa = 0, sb = 0;

   if (MODE == 2)
   {
      /* the segment's bit offset: the job's, the workgroup's within the job, the segments before it in the workgroup */
      uint32_t b = on ? A.bits[blockIdx.x * 256u + threadIdx.x] : 0u;
      uint32_t incl = (uint32_t) smr_wave_scan((int) b);
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      uint32_t before = 0;
      for (int w = 0; w < wave; w++) before += wsum[w];
      const uint64_t B0 = A.jobbit0[job] + A.blockstart[blockIdx.x] + before + (incl - b);
      wpos = (uint32_t) (B0 >> 5); nb = (uint32_t) (B0 & 31u); first_w = wpos; last_w = (uint32_t) ((B0 + b - 1u) >> 5);
   }

   auto put = [&](uint32_t v, uint32_t n)         /* n <= 32 bits, low bit first */
   {
      if (MODE == 1) { nbits += n; return; }
      acc |= (unsigned long long) v << nb; nb += n;
      if (nb >= 32u)
      {
         const uint32_t w = (uint32_t) acc;
         if (wpos == first_w || wpos == last_w) atomicOr(&A.out[wpos], w); else A.out[wpos] = w;
         wpos++; acc >>= 32; nb -= 32u;
      }
   };
   auto lit = [&](uint32_t v)
   {
      if (MODE == 0) { atomicAdd(&lh[v], 1u); return; }
      const uint32_t c = code[v];
      put(c & 0xffffu, c >> 16);
   };
   auto flush = [&](uint32_t v, uint32_t run)     /* a run of `run` bytes v: the byte, then the rest as a match on the byte before */
   {
      lit(v);
      const uint32_t m = run - 1u;
      if (m >= 3u)
      {
         uint32_t sym, eb, ev;
         dfl_len_sym(m, sym, eb, ev);
         if (MODE == 0) { atomicAdd(&lh[sym], 1u); return; }
         const uint32_t c = code[sym];
         put((c & 0xffffu) | ev << (c >> 16), (c >> 16) + eb + 1u);      /* the code, its extra bits, distance code 0 (one bit) */
      }
      else for (uint32_t i = 0; i < m; i++) lit(v);
   };

   if (on)
   {
      const uint32_t row = sl / spr, k = sl - row * spr;
      const uint32_t n = min(SMR_DFL_SEG, rowlen - k * SMR_DFL_SEG);
      const uint32_t *p = reinterpret_cast<const uint32_t *>(A.rows + ((size_t) job * A.nrows + row) * A.stride + (size_t) k * SMR_DFL_SEG);      /* the bytes from p + 3 on */
      uint32_t w = p[0], prev = 0xffffffffu, run = 0;
      for (uint32_t j = 0; j < n; j++)
      {
         const uint32_t bo = 3u + j;
         if ((bo & 3u) == 0u) w = p[bo >> 2];
         const uint32_t b = (w >> (8u * (bo & 3u))) & 255u;
         if (MODE == 0) { sa += b; sb += (unsigned long long) b * (n - j); }
         if (b == prev && run < 259u) run++;
         else { if (run) flush(prev, run); prev = b; run = 1u; }
      }
      flush(prev, run);
      if (MODE == 2 && nb) atomicOr(&A.out[wpos], (uint32_t) acc);
      if (MODE == 0)
      {
         const unsigned long long after = (unsigned long long) A.nrows * rowlen - ((unsigned long long) row * rowlen + (unsigned long long) k * SMR_DFL_SEG + n);
         sb = (sb + sa * (after % 65521ull)) % 65521ull;
      }
   }
   if (MODE == 0)
   {
      for (int o = 32; o; o >>= 1) { sa += __shfl_down(sa, o); sb += __shfl_down(sb, o); }
      if (lane == 0) { atomicAdd(&A.adler[2u * job], sa); atomicAdd(&A.adler[2u * job + 1u], sb); }
      __syncthreads();
      for (int i = threadIdx.x; i < 288; i += 256) if (lh[i]) atomicAdd(&A.hist[(size_t) job * 288u + i], lh[i]);
   }
   if (MODE == 1)
   {
      A.bits[blockIdx.x * 256u + threadIdx.x] = nbits;
      uint32_t t = nbits;
      for (int o = 32; o; o >>= 1) t += __shfl_down(t, o);
      if (lane == 0) wsum[wave] = t;
      __syncthreads();
      if (threadIdx.x == 0) A.blocksum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
   }
}

/* per job (one workgroup each): exclusive scan of its workgroups' bit counts, and their total */
__global__ __launch_bounds__(256) void k_dfl_scan(const uint32_t *__restrict__ sum, uint64_t *__restrict__ start, uint64_t *__restrict__ total, uint32_t bpj)
{
   __shared__ unsigned long long part[256];
   const uint32_t base = blockIdx.x * bpj;
   const uint32_t per = (bpj + 255u) / 256u, i0 = min(bpj, threadIdx.x * per), i1 = min(bpj, i0 + per);
   unsigned long long t = 0;
   for (uint32_t i = i0; i < i1; i++) t += sum[base + i];
   part[threadIdx.x] = t;
   __syncthreads();
   if (threadIdx.x == 0)
   {
      unsigned long long run = 0;
      for (int i = 0; i < 256; i++) { const unsigned long long v = part[i]; part[i] = run; run += v; }
      total[blockIdx.x] = run;
   }
   __syncthreads();
   unsigned long long run = part[threadIdx.x];
   for (uint32_t i = i0; i < i1; i++) { start[base + i] = run; run += sum[base + i]; }
}

/* ---------------------------------------------------------------------------------------- */
/* host side                                                                                 */
/* ---------------------------------------------------------------------------------------- */
static void be32(unsigned char *o, uint32_t v) { o[0] = (unsigned char) (v >> 24); o[1] = (unsigned char) (v >> 16); o[2] = (unsigned char) (v >> 8); o[3] = (unsigned char) v; }

static void png_chunk(std::vector<unsigned char> &f, const char *type, const unsigned char *data, size_t len)
{
   unsigned char h[8];
   be32(h, (uint32_t) len); memcpy(h + 4, type, 4);
   f.insert(f.end(), h, h + 8);
   if (len) f.insert(f.end(), data, data + len);
   uLong c = crc32(0L, h + 4, 4);
   if (len) c = crc32(c, data, (uInt) len);
   unsigned char t[4]; be32(t, (uint32_t) c);
   f.insert(f.end(), t, t + 4);
}

static void png_head(std::vector<unsigned char> &f, int w, int h, bool opaque)
{
   static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
   f.insert(f.end(), sig, sig + 8);
   unsigned char ihdr[13];
   be32(ihdr, (uint32_t) w); be32(ihdr + 4, (uint32_t) h);
   ihdr[8] = 8; ihdr[9] = opaque ? 2 : 6; ihdr[10] = ihdr[11] = ihdr[12] = 0;
   png_chunk(f, "IHDR", ihdr, 13);
}

/* deflates rows [r0, r1) of a strip (row k at base + k * stride + 3, 1 + rowbytes bytes) into out.  wbits 15: a whole
 * zlib stream (tiles); wbits -15: a raw chunk that ends on a byte boundary (sync flush) unless `last`.  Returns the
 * Adler-32 of the chunk's input in *adler (raw chunks only). */
static int deflate_rows(const unsigned char *base, size_t stride, size_t rowbytes, int r0, int r1, int level, int wbits, bool last,
                        std::vector<unsigned char> &out, uLong *adler)
{
   z_stream z;
   memset(&z, 0, sizeof(z));
   if (deflateInit2(&z, level, Z_DEFLATED, wbits, 9, Z_FILTERED) != Z_OK) return -1;
   out.resize(std::max<size_t>(65536, (size_t) (r1 - r0) * (rowbytes + 1) / 8));
   size_t have = 0;
   uLong ad = adler32(0L, Z_NULL, 0);
   for (int r = r0; r <= r1; r++)
   {
      const bool fin = r == r1;
      if (!fin)
      {
         z.next_in = (Bytef *) (base + (size_t) r * stride + 3);
         z.avail_in = (uInt) (rowbytes + 1);
         if (adler) ad = adler32(ad, z.next_in, z.avail_in);
      }
      const int flush = !fin ? Z_NO_FLUSH : (last || wbits > 0) ? Z_FINISH : Z_SYNC_FLUSH;
      for (;;)
      {
         if (out.size() - have < 32768) out.resize(out.size() * 2);
         z.next_out = out.data() + have;
         z.avail_out = (uInt) std::min<size_t>(out.size() - have, 1u << 30);
         const uInt before = z.avail_out;
         const int e = deflate(&z, flush);
         have += before - z.avail_out;
         if (e == Z_STREAM_ERROR) { deflateEnd(&z); return -1; }
         if (flush == Z_FINISH ? e == Z_STREAM_END : (z.avail_in == 0 && z.avail_out != 0)) break;
      }
   }
   deflateEnd(&z);
   out.resize(have);
   if (adler) *adler = ad;
   return 0;
}

static int n_workers(int threads)
{
   if (threads > 0) return std::min(threads, 64);
   unsigned hc = std::thread::hardware_concurrency();
   return (int) std::max(1u, std::min(16u, hc ? hc : 1u));
}

struct png_staging_t
{
   unsigned char *d[2] = { nullptr, nullptr }, *h[2] = { nullptr, nullptr };
   hipEvent_t ev[2] = { nullptr, nullptr };
   uint32_t *d_opaque = nullptr;
   ~png_staging_t()
   {
      for (int k = 0; k < 2; k++) { if (d[k]) hipFree(d[k]); if (h[k]) hipHostFree(h[k]); if (ev[k]) hipEventDestroy(ev[k]); }
      if (d_opaque) hipFree(d_opaque);
   }
   int setup(size_t bytes, size_t njobs, bool host = true)
   {
      for (int k = 0; k < 2; k++)
      {
         if (hipMalloc((void **) &d[k], bytes) != hipSuccess || (host && hipHostMalloc((void **) &h[k], bytes, hipHostMallocDefault) != hipSuccess) ||
             hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess)
            return smr_fail(SMR_E_NOMEM, "png: staging buffers of %zu bytes", bytes);
      }
      if (hipMalloc((void **) &d_opaque, njobs * 4) != hipSuccess) return smr_fail(SMR_E_NOMEM, "png: flags of %zu jobs", njobs);
      return SMR_OK;
   }
};

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

/* the flags of all jobs: device array (filled) and host copy */
static int png_opaque(smr_canvas_t *c, int x, int y, int w, int h, int nx, int njobs, uint32_t *d_opaque, std::vector<uint32_t> &flags)
{
   hipStream_t s = smr_g.stream;
   flags.assign((size_t) njobs, 1u);
   SMR_HIP(hipMemcpyAsync(d_opaque, flags.data(), (size_t) njobs * 4, hipMemcpyHostToDevice, s));
   k_png_opaque<<<(unsigned) ((int64_t) njobs * h), 256, 0, s>>>(c->pix, c->W, x, y, w, h, nx, njobs, d_opaque);      /* njobs * h < 2^31: checked by the callers */
   SMR_HIP(hipGetLastError());
   SMR_HIP(hipMemcpyAsync(flags.data(), d_opaque, (size_t) njobs * 4, hipMemcpyDeviceToHost, s));
   SMR_HIP(hipStreamSynchronize(s));
   return SMR_OK;
}

static int png_check_rect(smr_canvas_t *c, const char *who, int x, int y, int64_t w, int64_t h)
{
   if (!c || x < 0 || y < 0 || w <= 0 || h <= 0 || x + w > c->W || y + h > c->band_h)
      return smr_fail(SMR_E_INVAL, "%s: the rectangle %lld x %lld at (%d, %d) is not inside the band (%d x %d)", who, (long long) w,
                      (long long) h, x, y, c ? c->W : 0, c ? c->band_h : 0);
   if (w > 0x1fffffff) return smr_fail(SMR_E_INVAL, "%s: rows of more than 2^29 pixels are not supported", who);
   return SMR_OK;
}

/* ---- the host's share of the device deflate: the strip's Huffman code and its block header ---- */
/* code lengths of at most 15 bits for the symbols with a count (Huffman's algorithm; should a code come out longer, the counts
 * are halved, keeping every used symbol, and it runs again: a flatter distribution gives a shallower tree) */
static void huff_lengths(const uint32_t *count, int n, unsigned char *len)
{
   std::vector<uint64_t> f((size_t) n);
   for (int i = 0; i < n; i++) f[(size_t) i] = count[i];
   for (;;)
   {
      struct node_t { uint64_t w; int l, r; };
      std::vector<node_t> t;
      std::vector<int> live;
      for (int i = 0; i < n; i++) { t.push_back({ f[(size_t) i], -1, -1 }); if (f[(size_t) i]) live.push_back(i); }
      memset(len, 0, (size_t) n);
      if (live.size() == 1) { len[live[0]] = 1; return; }
      auto cmp = [&](int a, int b) { return t[(size_t) a].w != t[(size_t) b].w ? t[(size_t) a].w > t[(size_t) b].w : a > b; };
      std::make_heap(live.begin(), live.end(), cmp);
      while (live.size() > 1)
      {
         std::pop_heap(live.begin(), live.end(), cmp); const int a = live.back(); live.pop_back();
         std::pop_heap(live.begin(), live.end(), cmp); const int b = live.back(); live.pop_back();
         t.push_back({ t[(size_t) a].w + t[(size_t) b].w, a, b });
         live.push_back((int) t.size() - 1);
         std::push_heap(live.begin(), live.end(), cmp);
      }
      int deepest = 0;
      std::vector<std::pair<int, int>> st;
      st.push_back({ live[0], 0 });
      while (!st.empty())
      {
         const auto [v, d] = st.back(); st.pop_back();
         if (t[(size_t) v].l < 0) { len[v] = (unsigned char) d; deepest = std::max(deepest, d); }
         else { st.push_back({ t[(size_t) v].l, d + 1 }); st.push_back({ t[(size_t) v].r, d + 1 }); }
      }
      if (deepest <= 15) return;
      for (int i = 0; i < n; i++) if (f[(size_t) i]) f[(size_t) i] = (f[(size_t) i] + 1) / 2;
   }
}

static uint32_t bit_reverse(uint32_t v, int n) { uint32_t r = 0; for (int i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1 - i); return r; }

/* canonical codes (RFC 1951 3.2.2), each as the device wants it: bits << 16 | the code with its first bit lowest */
static void huff_codes(const unsigned char *len, int n, uint32_t *code)
{
   uint32_t bl_count[16] = { 0 }, next[16] = { 0 };
   for (int i = 0; i < n; i++) bl_count[len[i]]++;
   bl_count[0] = 0;
   uint32_t c = 0;
   for (int b = 1; b <= 15; b++) { c = (c + bl_count[b - 1]) << 1; next[b] = c; }
   for (int i = 0; i < n; i++) code[i] = len[i] ? ((uint32_t) len[i] << 16 | bit_reverse(next[len[i]]++, len[i])) : 0u;
}

struct bitbuf_t
{
   std::vector<unsigned char> b; uint64_t n = 0;
   void put(uint32_t v, int bits) { for (int i = 0; i < bits; i++, n++) { if ((n >> 3) >= b.size()) b.push_back(0); b[(size_t) (n >> 3)] |= (unsigned char) (((v >> i) & 1u) << (n & 7u)); } }
};

/* BFINAL, BTYPE = 2, and the two codes' lengths (RFC 1951 3.2.7).  The lengths go out one by one under a code-length code that
 * gives each of 0..15 four bits (no repeat symbols: 150 bytes per strip do not matter); one distance code, of one bit */
static void dfl_block_header(bitbuf_t &h, bool last, const unsigned char *len)
{
   static const int order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
   h.put(last ? 1u : 0u, 1); h.put(2u, 2);
   h.put(SMR_DFL_NSYM - 257, 5); h.put(0u, 5); h.put(19 - 4, 4);
   for (int i = 0; i < 19; i++) h.put(order[i] >= 16 ? 0u : 4u, 3);
   for (int i = 0; i < SMR_DFL_NSYM; i++) h.put(bit_reverse(len[i], 4), 4);      /* symbol v of the code-length code is the 4-bit code v */
   h.put(bit_reverse(1u, 4), 4);                                                  /* distance code 0: length 1 */
}

struct dfl_dev_t
{
   uint32_t *code = nullptr, *hist = nullptr, *bits = nullptr, *blocksum = nullptr, *out = nullptr;
   uint64_t *blockstart = nullptr, *jobtotal = nullptr, *jobbit0 = nullptr;
   unsigned long long *adler = nullptr;
   unsigned char *h_out = nullptr;
   size_t out_cap = 0, nblk_cap = 0, njobs_cap = 0;
   /* per job of the last batch: where its block lies in h_out, how long it is, its input's Adler-32 */
   std::vector<size_t> off, size;
   std::vector<uint64_t> nbit;             /* the block's bits (size = those rounded up to bytes) */
   std::vector<uLong> ad;
   ~dfl_dev_t()
   {
      void *all[] = { code, hist, bits, blocksum, out, blockstart, jobtotal, jobbit0, adler };
      for (void *q : all) if (q) hipFree(q);
      if (h_out) hipHostFree(h_out);
   }
   int setup(size_t njobs, size_t bpj)
   {
      const size_t nblk = njobs * bpj;
      if (hipMalloc((void **) &code, njobs * 288 * 4) != hipSuccess || hipMalloc((void **) &hist, njobs * 288 * 4) != hipSuccess ||
          hipMalloc((void **) &adler, njobs * 16) != hipSuccess || hipMalloc((void **) &bits, nblk * 256 * 4) != hipSuccess ||
          hipMalloc((void **) &blocksum, nblk * 4) != hipSuccess || hipMalloc((void **) &blockstart, nblk * 8) != hipSuccess ||
          hipMalloc((void **) &jobtotal, njobs * 8) != hipSuccess || hipMalloc((void **) &jobbit0, njobs * 8) != hipSuccess)
         return smr_fail(SMR_E_NOMEM, "png: tables of the device deflate for %zu jobs of %zu workgroups", njobs, bpj);
      nblk_cap = nblk; njobs_cap = njobs;
      return SMR_OK;
   }
   int room(size_t bytes)
   {
      if (bytes <= out_cap) return SMR_OK;
      if (out) hipFree(out);
      if (h_out) hipHostFree(h_out);
      out = nullptr; h_out = nullptr; out_cap = 0;
      const size_t cap = bytes + bytes / 4 + 65536;
      if (hipMalloc((void **) &out, cap) != hipSuccess || hipHostMalloc((void **) &h_out, cap, hipHostMallocDefault) != hipSuccess)
         return smr_fail(SMR_E_NOMEM, "png: %zu bytes for a batch of compressed rows", cap);
      out_cap = cap;
      return SMR_OK;
   }
};

/* njobs jobs of nrows filtered rows each (on the device, job after job) -> one deflate block per job in D.h_out: job j's at
 * D.off[j], D.size[j] bytes that end on a byte boundary, BFINAL set if `last`; D.ad[j]: the Adler-32 of the job's rows by themselves.
 * pre: bytes left free in front of every block (a tile's zlib header).  T: host threads for the jobs' Huffman codes. */
static int dfl_batch(dfl_dev_t &D, const unsigned char *d_rows, uint32_t stride, int w, int nrows, int njobs, const uint32_t *d_opaque,
                     const uint32_t *h_opaque, bool last, size_t pre, int T)
{
   hipStream_t s = smr_g.stream;
   const uint32_t spr_max = (1u + 4u * (uint32_t) w + SMR_DFL_SEG - 1u) / SMR_DFL_SEG;
   const uint32_t bpj = (uint32_t) (((uint64_t) nrows * spr_max + 255u) / 256u);
   const size_t nblk = (size_t) njobs * bpj;
   if (nblk > D.nblk_cap || (size_t) njobs > D.njobs_cap || nblk > 0x7fffffffu) return smr_fail(SMR_E_HIP, "png: internal: a batch of %zu workgroups", nblk);
   dfl_args_t A;
   memset(&A, 0, sizeof(A));
   A.rows = d_rows; A.stride = stride; A.w = (uint32_t) w; A.nrows = (uint32_t) nrows; A.bpj = bpj; A.opaque = d_opaque;
   A.code = D.code; A.hist = D.hist; A.adler = D.adler; A.bits = D.bits; A.blocksum = D.blocksum; A.blockstart = D.blockstart; A.jobbit0 = D.jobbit0;
   SMR_HIP(hipMemsetAsync(D.hist, 0, (size_t) njobs * 288 * 4, s));
   SMR_HIP(hipMemsetAsync(D.adler, 0, (size_t) njobs * 16, s));
   k_dfl<0><<<(unsigned) nblk, 256, 0, s>>>(A);
   SMR_HIP(hipGetLastError());
   std::vector<uint32_t> h_hist((size_t) njobs * 288), code((size_t) njobs * 288, 0u);
   std::vector<unsigned long long> h_ad((size_t) njobs * 2);
   SMR_HIP(hipMemcpyAsync(h_hist.data(), D.hist, h_hist.size() * 4, hipMemcpyDeviceToHost, s));
   SMR_HIP(hipMemcpyAsync(h_ad.data(), D.adler, h_ad.size() * 8, hipMemcpyDeviceToHost, s));
   SMR_HIP(hipStreamSynchronize(s));
   std::vector<bitbuf_t> hdr((size_t) njobs);
   {
      std::atomic<int> next(0);
      auto work = [&]()
      {
         unsigned char len[288];
         for (;;)
         {
            const int j = next.fetch_add(1);
            if (j >= njobs) break;
            uint32_t *h = &h_hist[(size_t) j * 288];
            h[256] = 1;                                              /* the end-of-block code */
            memset(len, 0, sizeof(len));
            huff_lengths(h, SMR_DFL_NSYM, len);
            huff_codes(len, SMR_DFL_NSYM, &code[(size_t) j * 288]);
            dfl_block_header(hdr[(size_t) j], last, len);
         }
      };
      const int nt = std::max(1, std::min(T, njobs / 4));
      if (nt == 1) work();
      else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(work); for (auto &t : th) t.join(); }
   }
   SMR_HIP(hipMemcpyAsync(D.code, code.data(), code.size() * 4, hipMemcpyHostToDevice, s));
   k_dfl<1><<<(unsigned) nblk, 256, 0, s>>>(A);
   k_dfl_scan<<<(unsigned) njobs, 256, 0, s>>>(D.blocksum, D.blockstart, D.jobtotal, bpj);
   SMR_HIP(hipGetLastError());
   std::vector<uint64_t> total((size_t) njobs), bit0((size_t) njobs);
   SMR_HIP(hipMemcpyAsync(total.data(), D.jobtotal, (size_t) njobs * 8, hipMemcpyDeviceToHost, s));
   SMR_HIP(hipStreamSynchronize(s));
   /* the jobs' blocks one behind the other, each from a dword boundary on (no word of the stream belongs to two jobs) */
   D.off.assign((size_t) njobs, 0); D.size.assign((size_t) njobs, 0); D.ad.assign((size_t) njobs, 0); D.nbit.assign((size_t) njobs, 0);
   size_t at = 0;
   for (int j = 0; j < njobs; j++)
   {
      const uint32_t eob = code[(size_t) j * 288 + 256] >> 16;
      at = (at + pre + 3) & ~(size_t) 3;
      D.off[(size_t) j] = at;
      D.nbit[(size_t) j] = hdr[(size_t) j].n + total[(size_t) j] + eob;
      D.size[(size_t) j] = (size_t) ((D.nbit[(size_t) j] + 7) >> 3);
      bit0[(size_t) j] = (uint64_t) at * 8 + hdr[(size_t) j].n;
      at += D.size[(size_t) j] + 8;                                  /* room behind it: the stored block's four bytes, a tile's Adler-32 */
   }
   SMR_TRY(D.room(at + 16));
   A.out = D.out;
   SMR_HIP(hipMemcpyAsync(D.jobbit0, bit0.data(), (size_t) njobs * 8, hipMemcpyHostToDevice, s));
   SMR_HIP(hipMemsetAsync(D.out, 0, (at + 16 + 3) & ~(size_t) 3, s));
   k_dfl<2><<<(unsigned) nblk, 256, 0, s>>>(A);
   SMR_HIP(hipGetLastError());
   SMR_HIP(hipMemcpyAsync(D.h_out, D.out, at, hipMemcpyDeviceToHost, s));
   SMR_HIP(hipStreamSynchronize(s));
   const uLong M = 65521;
   for (int j = 0; j < njobs; j++)
   {
      unsigned char *o = D.h_out + D.off[(size_t) j];
      const bitbuf_t &h = hdr[(size_t) j];
      for (size_t i = 0; i < h.b.size(); i++) o[i] |= h.b[i];
      const uint64_t e0 = h.n + total[(size_t) j];                   /* the end-of-block code; what follows it in the byte are zeros */
      const uint32_t c = code[(size_t) j * 288 + 256];
      for (uint32_t i = 0; i < (c >> 16); i++) o[(size_t) ((e0 + i) >> 3)] |= (unsigned char) (((c >> i) & 1u) << ((e0 + i) & 7u));
      const uint64_t nbytes = (uint64_t) nrows * (1u + (uint64_t) w * (h_opaque[j] ? 3u : 4u));
      D.ad[(size_t) j] = (uLong) (((nbytes + h_ad[(size_t) j * 2 + 1]) % M) << 16 | ((1 + h_ad[(size_t) j * 2]) % M));
   }
   return SMR_OK;
}

extern "C" int smr_hip_canvas_write_png(smr_canvas_t *c, int x, int y, int w, int h, const char *path, int level, int threads,
                                        smr_png_stats_t *st)
{
   SMR_NEED_INIT();
   SMR_TRY(png_check_rect(c, "canvas_write_png", x, y, w, h));
   if (!path) return smr_fail(SMR_E_INVAL, "canvas_write_png: no path");
   const bool on_device = level == SMR_PNG_DEVICE_DEFLATE;
   if ((level < -1 || level > 9) && !on_device)
      return smr_fail(SMR_E_INVAL, "canvas_write_png: compression level %d (zlib: -1, 0..9; %d: deflate on the device)", level, SMR_PNG_DEVICE_DEFLATE);
   const double t0 = now_ms();
   hipStream_t s = smr_g.stream;
   const int T = n_workers(threads);
   const uint32_t stride = 4u + 4u * (uint32_t) w;
   const int strip_rows = (int) std::max<uint64_t>(1, std::min<uint64_t>((uint64_t) h, SMR_PNG_STRIP_BYTES / stride));
   png_staging_t S;
   SMR_TRY(S.setup((size_t) strip_rows * stride, 1, !on_device));
   std::vector<uint32_t> flags;
   SMR_TRY(png_opaque(c, x, y, w, h, 1, 1, S.d_opaque, flags));
   const bool opaque = flags[0] != 0;
   const size_t rowbytes = (size_t) w * (opaque ? 3 : 4);

   FILE *f = fopen(path, "wb");
   if (!f) return smr_fail(SMR_E_INVAL, "canvas_write_png: cannot open %s: %s", path, strerror(errno));
   std::vector<unsigned char> head;
   png_head(head, w, h, opaque);
   static const unsigned char zhdr[2] = { 0x78, 0x9c };
   png_chunk(head, "IDAT", zhdr, 2);
   bool io_ok = fwrite(head.data(), 1, head.size(), f) == head.size();
   uint64_t bytes_out = head.size(), bytes_in = 0;
   double deflate_ms = 0;

   auto launch = [&](int k, int row0) -> int
   {
      png_args_t A;
      A.pix = c->pix; A.W = c->W; A.x0 = x; A.y0 = y; A.w = w; A.h = h; A.nx = 1; A.row0 = row0; A.nrows = std::min(strip_rows, h - row0);
      A.job0 = 0; A.njobs = 1; A.opaque = S.d_opaque; A.out = S.d[k & 1]; A.stride = stride;
      k_png_rows<<<(unsigned) A.nrows, 256, 0, s>>>(A);
      SMR_HIP(hipGetLastError());
      if (!on_device) SMR_HIP(hipMemcpyAsync(S.h[k & 1], S.d[k & 1], (size_t) A.nrows * stride, hipMemcpyDeviceToHost, s));
      SMR_HIP(hipEventRecord(S.ev[k & 1], s));
      return SMR_OK;
   };

   const int nstrips = (h + strip_rows - 1) / strip_rows;
   int rc = launch(0, 0);
   uLong adler = adler32(0L, Z_NULL, 0);
   std::vector<std::vector<unsigned char>> outs((size_t) T);
   std::vector<uLong> ads((size_t) T);
   std::vector<int> ok((size_t) T);
   std::vector<unsigned char> chunked;
   dfl_dev_t D;
   if (on_device && rc == SMR_OK)
      rc = D.setup(1, ((size_t) strip_rows * ((1u + 4u * (size_t) w + SMR_DFL_SEG - 1) / SMR_DFL_SEG) + 255) / 256);
   for (int k = 0; k < nstrips && rc == SMR_OK && on_device; k++)
   {
      /* the strip's rows are filtered, deflated and copied out by the one stream; the next strip's filter pass is queued behind */
      const int row0 = k * strip_rows, nrows = std::min(strip_rows, h - row0);
      const bool last = k == nstrips - 1;
      const double td = now_ms();
      if ((rc = dfl_batch(D, S.d[k & 1], stride, w, nrows, 1, S.d_opaque, flags.data(), last, 0, 1)) != SMR_OK) break;
      if (k + 1 < nstrips && (rc = launch(k + 1, row0 + nrows)) != SMR_OK) break;
      deflate_ms += now_ms() - td;
      /* a strip that is not the last ends in an empty stored block: its three header bits are zeros left in the block's last byte, or
       * one more byte when fewer are left, then LEN and NLEN on the byte boundary */
      unsigned char *z = D.h_out + D.off[0];
      size_t n = D.size[0];
      if (!last)
      {
         if ((uint64_t) n * 8 - D.nbit[0] < 3) z[n++] = 0;
         z[n++] = 0; z[n++] = 0; z[n++] = 0xff; z[n++] = 0xff;
      }
      adler = adler32_combine(adler, D.ad[0], (z_off_t) ((int64_t) nrows * (int64_t) (rowbytes + 1)));
      bytes_in += (uint64_t) nrows * (rowbytes + 1);
      /* one IDAT chunk per worker: their CRCs are the host's share */
      const int nt = (int) std::max<size_t>(1, std::min<size_t>((size_t) T, n >> 16));
      std::vector<std::thread> th;
      for (int t = 0; t < nt; t++)
         th.emplace_back([&, t]()
         {
            const size_t a = n * (size_t) t / (size_t) nt, b = n * (size_t) (t + 1) / (size_t) nt;
            outs[(size_t) t].clear();
            png_chunk(outs[(size_t) t], "IDAT", z + a, b - a);
         });
      for (auto &t : th) t.join();
      for (int t = 0; t < nt; t++)
      {
         io_ok = io_ok && fwrite(outs[(size_t) t].data(), 1, outs[(size_t) t].size(), f) == outs[(size_t) t].size();
         bytes_out += outs[(size_t) t].size();
      }
   }
   for (int k = 0; k < nstrips && rc == SMR_OK && !on_device; k++)
   {
      const int row0 = k * strip_rows, nrows = std::min(strip_rows, h - row0);
      if (hipEventSynchronize(S.ev[k & 1]) != hipSuccess) { rc = smr_fail(SMR_E_HIP, "canvas_write_png: strip %d did not arrive", k); break; }
      if (k + 1 < nstrips && (rc = launch(k + 1, row0 + nrows)) != SMR_OK) break;     /* the GPU and PCIe work on the next strip meanwhile */
      const double td = now_ms();
      const int nt = std::max(1, std::min(T, nrows / 8));           /* chunks of at least a few rows */
      std::vector<std::thread> th;
      for (int t = 0; t < nt; t++)
      {
         const int r0 = (int) ((int64_t) nrows * t / nt), r1 = (int) ((int64_t) nrows * (t + 1) / nt);
         const bool last = k == nstrips - 1 && t == nt - 1;
         th.emplace_back([&, t, r0, r1, last]() { ok[t] = deflate_rows(S.h[k & 1], stride, rowbytes, r0, r1, level, -15, last, outs[t], &ads[t]); });
      }
      for (auto &t : th) t.join();
      deflate_ms += now_ms() - td;
      for (int t = 0; t < nt; t++)
      {
         if (ok[t] != 0) { rc = smr_fail(SMR_E_INVAL, "canvas_write_png: zlib refused the stream"); break; }
         const int r0 = (int) ((int64_t) nrows * t / nt), r1 = (int) ((int64_t) nrows * (t + 1) / nt);
         adler = adler32_combine(adler, ads[t], (z_off_t) ((int64_t) (r1 - r0) * (int64_t) (rowbytes + 1)));
         bytes_in += (uint64_t) (r1 - r0) * (rowbytes + 1);
         /* IDAT chunks hold at most 2^31 - 1 bytes */
         for (size_t off = 0; off < outs[t].size();)
         {
            const size_t n = std::min<size_t>(outs[t].size() - off, 0x40000000u);
            chunked.clear();
            png_chunk(chunked, "IDAT", outs[t].data() + off, n);
            io_ok = io_ok && fwrite(chunked.data(), 1, chunked.size(), f) == chunked.size();
            bytes_out += chunked.size();
            off += n;
         }
      }
   }
   hipStreamSynchronize(s);
   if (rc == SMR_OK)
   {
      unsigned char tail[4]; be32(tail, (uint32_t) adler);
      chunked.clear();
      png_chunk(chunked, "IDAT", tail, 4);
      png_chunk(chunked, "IEND", nullptr, 0);
      io_ok = io_ok && fwrite(chunked.data(), 1, chunked.size(), f) == chunked.size();
      bytes_out += chunked.size();
   }
   io_ok = (fclose(f) == 0) && io_ok;
   if (rc != SMR_OK) { remove(path); return rc; }
   if (!io_ok) { remove(path); return smr_fail(SMR_E_INVAL, "canvas_write_png: writing %s failed: %s", path, strerror(errno)); }
   if (st)
   {
      memset(st, 0, sizeof(*st));
      st->n_files = 1; st->n_opaque = opaque; st->bytes_filtered = bytes_in; st->bytes_written = bytes_out;
      st->deflate_ms = (float) deflate_ms; st->total_ms = (float) (now_ms() - t0); st->threads = T;
   }
   return SMR_OK;
}

static int make_dir(const std::string &d)
{
   struct stat sb;
   if (!stat(d.c_str(), &sb)) return 0;
   if (errno != ENOENT || (mkdir(d.c_str(), 0777) == -1 && errno != EEXIST)) return -1;
   return 0;
}

extern "C" int smr_hip_canvas_write_tiles(smr_canvas_t *c, int x, int y, int nx, int ny, int tile_px, const char *dir, int tx0, int ty0,
                                          int level, int threads, smr_png_stats_t *st)
{
   SMR_NEED_INIT();
   if (nx <= 0 || ny <= 0 || tile_px <= 0 || tile_px > 16384 || !dir) return smr_fail(SMR_E_INVAL, "canvas_write_tiles: bad tile grid");
   SMR_TRY(png_check_rect(c, "canvas_write_tiles", x, y, (int64_t) nx * tile_px, (int64_t) ny * tile_px));
   if ((int64_t) nx * ny > 0x7fffffff / tile_px) return smr_fail(SMR_E_INVAL, "canvas_write_tiles: %d x %d tiles are too many for one call", nx, ny);
   const bool on_device = level == SMR_PNG_DEVICE_DEFLATE;
   if ((level < -1 || level > 9) && !on_device)
      return smr_fail(SMR_E_INVAL, "canvas_write_tiles: compression level %d (zlib: -1, 0..9; %d: deflate on the device)", level, SMR_PNG_DEVICE_DEFLATE);
   const double t0 = now_ms();
   hipStream_t s = smr_g.stream;
   const int T = n_workers(threads), njobs = nx * ny;
   const uint32_t stride = 4u + 4u * (uint32_t) tile_px;
   const size_t tile_bytes = (size_t) tile_px * stride;
   const int per_batch = (int) std::max<size_t>(1, std::min<size_t>((size_t) njobs, SMR_PNG_STRIP_BYTES / tile_bytes));
   png_staging_t S;
   SMR_TRY(S.setup((size_t) per_batch * tile_bytes, (size_t) njobs, !on_device));
   dfl_dev_t D;
   if (on_device) SMR_TRY(D.setup((size_t) per_batch, ((size_t) tile_px * ((1u + 4u * (size_t) tile_px + SMR_DFL_SEG - 1) / SMR_DFL_SEG) + 255) / 256));
   std::vector<uint32_t> flags;
   SMR_TRY(png_opaque(c, x, y, tile_px, tile_px, nx, njobs, S.d_opaque, flags));
   /* <dir>/<x>/<y>.png, the layout of create_tiles (src/smtile.c:186-212) below its zoom directory */
   if (make_dir(dir)) return smr_fail(SMR_E_INVAL, "canvas_write_tiles: cannot create %s: %s", dir, strerror(errno));
   for (int i = 0; i < nx; i++)
      if (make_dir(std::string(dir) + "/" + std::to_string(tx0 + i)))
         return smr_fail(SMR_E_INVAL, "canvas_write_tiles: cannot create %s/%d: %s", dir, tx0 + i, strerror(errno));

   auto launch = [&](int k, int job0) -> int
   {
      png_args_t A;
      A.pix = c->pix; A.W = c->W; A.x0 = x; A.y0 = y; A.w = tile_px; A.h = tile_px; A.nx = nx; A.row0 = 0; A.nrows = tile_px;
      A.job0 = job0; A.njobs = std::min(per_batch, njobs - job0); A.opaque = S.d_opaque; A.out = S.d[k & 1]; A.stride = stride;
      k_png_rows<<<(unsigned) ((int64_t) A.njobs * tile_px), 256, 0, s>>>(A);
      SMR_HIP(hipGetLastError());
      if (!on_device) SMR_HIP(hipMemcpyAsync(S.h[k & 1], S.d[k & 1], (size_t) A.njobs * tile_bytes, hipMemcpyDeviceToHost, s));
      SMR_HIP(hipEventRecord(S.ev[k & 1], s));
      return SMR_OK;
   };

   const int nbatches = (njobs + per_batch - 1) / per_batch;
   int rc = launch(0, 0);
   std::atomic<uint64_t> bytes_out(0);
   std::atomic<int> failed(0);
   uint64_t n_opaque = 0, bytes_in = 0;
   double deflate_ms = 0;
   for (int k = 0; k < nbatches && rc == SMR_OK; k++)
   {
      const int job0 = k * per_batch, nj = std::min(per_batch, njobs - job0);
      if (hipEventSynchronize(S.ev[k & 1]) != hipSuccess) { rc = smr_fail(SMR_E_HIP, "canvas_write_tiles: batch %d did not arrive", k); break; }
      const double td = now_ms();
      /* on the device: every tile of the batch one deflate block (a zlib stream of its own: two bytes in front, its Adler-32 behind) */
      if (on_device && (rc = dfl_batch(D, S.d[k & 1], stride, tile_px, tile_px, nj, S.d_opaque + job0, &flags[(size_t) job0], true, 2, T)) != SMR_OK) break;
      if (k + 1 < nbatches && (rc = launch(k + 1, job0 + nj)) != SMR_OK) break;
      std::atomic<int> next(0);
      std::vector<std::thread> th;
      for (int t = 0; t < std::min(T, nj); t++)
         th.emplace_back([&]()
         {
            std::vector<unsigned char> z, file;
            for (;;)
            {
               const int jl = next.fetch_add(1);
               if (jl >= nj) break;
               const int j = job0 + jl;
               const bool opaque = flags[(size_t) j] != 0;
               const unsigned char *zp; size_t zn;
               if (on_device)
               {
                  unsigned char *o = D.h_out + D.off[(size_t) jl];
                  o[-2] = 0x78; o[-1] = 0x9c;
                  be32(o + D.size[(size_t) jl], (uint32_t) D.ad[(size_t) jl]);
                  zp = o - 2; zn = D.size[(size_t) jl] + 6;
               }
               else
               {
                  if (deflate_rows(S.h[k & 1] + (size_t) jl * tile_bytes, stride, (size_t) tile_px * (opaque ? 3 : 4), 0, tile_px, level, 15, true, z, nullptr))
                  { failed.store(1); continue; }
                  zp = z.data(); zn = z.size();
               }
               file.clear();
               png_head(file, tile_px, tile_px, opaque);
               png_chunk(file, "IDAT", zp, zn);
               png_chunk(file, "IEND", nullptr, 0);
               const std::string p = std::string(dir) + "/" + std::to_string(tx0 + j % nx) + "/" + std::to_string(ty0 + j / nx) + ".png";
               FILE *f = fopen(p.c_str(), "wb");
               if (!f || fwrite(file.data(), 1, file.size(), f) != file.size()) failed.store(2);
               if (f && fclose(f) != 0) failed.store(2);
               bytes_out.fetch_add(file.size());
            }
         });
      for (auto &t : th) t.join();
      deflate_ms += now_ms() - td;
      for (int jl = 0; jl < nj; jl++)
      {
         const bool opaque = flags[(size_t) (job0 + jl)] != 0;
         n_opaque += opaque;
         bytes_in += (uint64_t) tile_px * ((uint64_t) tile_px * (opaque ? 3 : 4) + 1);
      }
   }
   hipStreamSynchronize(s);
   if (rc != SMR_OK) return rc;
   if (failed.load()) return smr_fail(SMR_E_INVAL, failed.load() == 1 ? "canvas_write_tiles: zlib refused a stream" : "canvas_write_tiles: writing a tile below %s failed", dir);
   if (st)
   {
      memset(st, 0, sizeof(*st));
      st->n_files = (uint64_t) njobs; st->n_opaque = n_opaque; st->bytes_filtered = bytes_in; st->bytes_written = bytes_out.load();
      st->deflate_ms = (float) deflate_ms; st->total_ms = (float) (now_ms() - t0); st->threads = T;
   }
   return SMR_OK;
}
