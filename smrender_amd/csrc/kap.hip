/* kap.hip -- SURVEY 8f rank 4: the KAP (BSB raster chart) writer from the device canvas.
 *
 * save_kap (src/smkap.c:196-245) reduces the rendered image to at most 127 colours with a median cut
 * (cairo_smr_image_surface_color_reduce, src/median_cut.c:293-350), looks every pixel's colour up in the palette and writes
 * the rows run-length coded (bsb_compress_row, src/smkap.c:70-114) with a table of row offsets behind them.  Here:
 *   - the median cut runs as the reference's does, cut by cut, on the pixels themselves: the block with the longest side is
 *     sorted along that side -- a STABLE sort by one byte, as glibc's merge-sort qsort is (up to 2.36; the order of the points
 *     that tie at the median decides both halves' boxes, so it is part of the result) -- and split at the median; the host
 *     keeps the list of blocks (at most 127) and the device sorts a segment and reduces the two halves per cut;
 *   - nearest palette colour per pixel (mc_nearest_block_index :159-172: L1 distance, the first of equals; then
 *     palette_index, smkap.c:177-184: the first palette entry of that colour);
 *   - one wavefront per row codes the runs: a run is written by the lane of the pixel that ends it, places come from a scan of
 *     the byte counts; rows are counted first, placed by a scan, then written.
 * The header text before "OST/1" (gen_kap_header :143-167) is the caller's.
 */
#include "smr_host.h"
#include <arpa/inet.h>
#include <string.h>
#include <chrono>
#include <rocprim/rocprim.hpp>

namespace {

struct stats_t { uint32_t nlo[3], hi[3]; unsigned long long sum[3]; };      /* nlo = 255 - minimum: all-zero is the neutral start */

__global__ __launch_bounds__(256) void k_kap_points(const uint32_t *__restrict__ pix, int stride, int w, int h, uint32_t *__restrict__ pts)
{
   const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
   if (x < w) pts[(int64_t) y * w + x] = pix[(int64_t) y * stride + x] & 0xffffffu;
}

/* box and sums of a segment of points (mc_shrink :178-189, the sums of mc_avg_block :195-206) */
__global__ __launch_bounds__(256) void k_kap_stats(const uint32_t *__restrict__ pts, int64_t n, stats_t *__restrict__ out)
{
   __shared__ stats_t part[4];
   uint32_t lo[3] = { 255, 255, 255 }, hi[3] = { 0, 0, 0 };
   unsigned long long sum[3] = { 0, 0, 0 };
   for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256)
   {
      const uint32_t p = pts[i];
      for (int j = 0; j < 3; j++) { const uint32_t v = (p >> (8 * j)) & 0xff; lo[j] = min(lo[j], v); hi[j] = max(hi[j], v); sum[j] += v; }
   }
   for (int d = 32; d; d >>= 1)
      for (int j = 0; j < 3; j++)
      {
         lo[j] = min(lo[j], (uint32_t) __shfl_xor((int) lo[j], d)); hi[j] = max(hi[j], (uint32_t) __shfl_xor((int) hi[j], d));
         sum[j] += (unsigned long long) __shfl_xor((long long) sum[j], d);
      }
   if ((threadIdx.x & 63) == 0) for (int j = 0; j < 3; j++) { part[threadIdx.x >> 6].nlo[j] = 255 - lo[j]; part[threadIdx.x >> 6].hi[j] = hi[j]; part[threadIdx.x >> 6].sum[j] = sum[j]; }
   __syncthreads();
   if (threadIdx.x < 3)
   {
      const int j = threadIdx.x;
      uint32_t a = 0, b = 0; unsigned long long s = 0;
      for (int k = 0; k < 4; k++) { a = max(a, part[k].nlo[j]); b = max(b, part[k].hi[j]); s += part[k].sum[j]; }
      atomicMax(&out->nlo[j], a); atomicMax(&out->hi[j], b); atomicAdd(&out->sum[j], s);
   }
}

struct pal_t { int n; uint8_t c[127][4]; uint8_t first[128]; };      /* c: blue, green, red; first[i]: 1 + the first entry of entry i's colour */

__global__ __launch_bounds__(256) void k_kap_index(const uint32_t *__restrict__ pix, int stride, int w, int h, pal_t pal, uint8_t *__restrict__ idx)
{
   __shared__ pal_t sp;
   for (int i = threadIdx.x; i < (int) (sizeof(pal_t) / 4); i += 256) ((uint32_t *) &sp)[i] = ((const uint32_t *) &pal)[i];
   __syncthreads();
   const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
   if (x >= w) return;
   const uint32_t p = pix[(int64_t) y * stride + x];
   const int b = p & 0xff, g = (p >> 8) & 0xff, r = (p >> 16) & 0xff;
   int best = 0x1000000, at = 0;
   for (int i = 0; i < sp.n; i++)
   {
      const int d = abs(sp.c[i][0] - b) + abs(sp.c[i][1] - g) + abs(sp.c[i][2] - r);
      if (d < best) { best = d; at = i; }
   }
   idx[(int64_t) y * w + x] = sp.first[at];
}

/* bsb_compress_nb (src/smkap.c:53-67) without the recursion: the number of bytes, and the bytes */
__device__ __forceinline__ int nb_count(uint32_t nb, uint32_t pixel, uint32_t max)
{
   if (nb <= max) return ((pixel | nb) & 0xff) ? 1 : 2;
   int n = 1;
   while (nb > max) { nb >>= 7; n++; }
   return n;               /* below the first level `pixel` carries 0x80: never the 0x80 0x00 pair */
}
__device__ __forceinline__ void nb_put(uint8_t *p, uint32_t nb, uint32_t pixel, uint32_t max, int n)
{
   if (nb <= max)
   {
      const uint32_t v = (pixel | nb) & 0xff;
      if (!v) *p++ = 0x80;
      *p = (uint8_t) v;
      return;
   }
   p[n - 1] = (uint8_t) ((nb & 0x7f) | (pixel & 0x80));
   nb >>= 7;
   for (int k = n - 2; k >= 1; k--) { p[k] = (uint8_t) ((nb & 0x7f) | 0x80); nb >>= 7; }
   p[0] = (uint8_t) (pixel | 0x80 | nb);
}

/* bsb_compress_row (src/smkap.c:70-114, widthout == widthin): one wavefront per row.  EMIT = false counts the row's bytes */
template <bool EMIT>
__global__ __launch_bounds__(256) void k_kap_rows(const uint8_t *__restrict__ idx, int w, int h, int depth, const uint32_t *__restrict__ row_off,
                                                  uint32_t *__restrict__ row_len, uint8_t *__restrict__ bytes)
{
   const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
   if (row >= h) return;
   const uint8_t *in = idx + (int64_t) row * w;
   uint8_t *out = EMIT ? bytes + row_off[row] : nullptr;
   const uint32_t dec = 7 - depth, max = (1u << dec) - 1;
   int n = nb_count((uint32_t) row, 0, 0x7f);
   if (EMIT && lane == 0) nb_put(out, (uint32_t) row, 0, 0x7f, n);
   int run_start = 0;
   for (int x0 = 0; x0 <= w; x0 += 64)
   {
      const int x = x0 + lane;
      const int cur = x < w ? in[x] : -1;
      const int prev = x > 0 && x <= w ? in[x - 1] : -2;
      const bool ends = x > 0 && x <= w && cur != prev;          /* a run ends before pixel x (x == w: the row's last run) */
      const uint64_t mask = __ballot(ends);
      const uint64_t below = mask & ((1ull << lane) - 1);
      const int start = below ? x0 + 63 - __clzll(below) : run_start;
      const uint32_t nb = (uint32_t) (x - start - 1), pixel = ((uint32_t) prev << dec) & 0xff;
      const int mine = ends ? nb_count(nb, pixel, max) : 0;
      int incl = mine;
      for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
      if (EMIT && ends) nb_put(out + n + incl - mine, nb, pixel, max, mine);
      n += __shfl(incl, 63);
      if (mask) run_start = x0 + 63 - __clzll(mask);
   }
   if (EMIT && lane == 0) out[n] = 0;
   n++;
   if (!EMIT && lane == 0) row_len[row] = (uint32_t) n;
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct blk_t { int64_t off, len; int lo[3], hi[3], avg[3]; };
int side_index(const blk_t &b) { int m = 0, mi = 0; for (int i = 0; i < 3; i++) { const int d = b.hi[i] - b.lo[i]; if (d > m) { m = d; mi = i; } } return mi; }
int side_len(const blk_t &b) { const int i = side_index(b); return b.hi[i] - b.lo[i]; }

struct dmem_t
{
   void *p = nullptr;
   ~dmem_t() { if (p) hipFree(p); }
   int get(size_t bytes) { if (hipMalloc(&p, bytes < 256 ? 256 : bytes) != hipSuccess) { p = nullptr; return smr_fail(SMR_E_NOMEM, "write_kap: hipMalloc of %zu bytes failed", bytes); } return SMR_OK; }
   template <class T> T *as() const { return (T *) p; }
};

}  // namespace

extern "C" int smr_hip_canvas_write_kap(smr_canvas_t *c, const char *header, const char *path, smr_kap_stats_t *st)
{
   SMR_NEED_INIT();
   if (!c || !path) return smr_fail(SMR_E_INVAL, "write_kap: NULL argument");
   if (c->band_y0 != 0 || c->band_h != c->H) return smr_fail(SMR_E_UNSUPPORTED, "write_kap: the colour reduction looks at the whole chart: not on a band (rows %d..%d of %d)", c->band_y0, c->band_y0 + c->band_h, c->H);
   const int w = c->W, h = c->H;
   if (w > 65535 || h > 65535) return smr_fail(SMR_E_INVAL, "write_kap: %d x %d px (KAP rows and line numbers are 16 bit)", w, h);
   if ((int64_t) w * h >= (int64_t) INT32_MAX) return smr_fail(SMR_E_CAPACITY, "write_kap: %lld pixels (the reference's mc_median_cut counts them in an int)", (long long) w * h);
   hipStream_t s = smr_g.stream;
   const int64_t n = (int64_t) w * h;
   const double t0 = now_ms();

   /* ---- the median cut, src/median_cut.c:229-264 ---- */
   dmem_t pts, alt, tmp, dstat;
   SMR_TRY(pts.get(n * 4)); SMR_TRY(alt.get(n * 4)); SMR_TRY(dstat.get(2 * sizeof(stats_t)));
   size_t tmp_bytes = 0;
   SMR_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, pts.as<uint32_t>(), alt.as<uint32_t>(), (size_t) n, 0, 8, s));
   SMR_TRY(tmp.get(tmp_bytes));
   k_kap_points<<<dim3((w + 255) / 256, h), 256, 0, s>>>(c->pix, c->W, w, h, pts.as<uint32_t>());
   auto stats = [&](const blk_t *a, const blk_t *b, stats_t *host) -> int {
      SMR_HIP(hipMemsetAsync(dstat.p, 0, 2 * sizeof(stats_t), s));
      for (int k = 0; k < 2; k++)
      {
         const blk_t *q = k ? b : a;
         if (!q) continue;
         const unsigned nb = (unsigned) std::min<int64_t>((q->len + 255) / 256, 2048);
         k_kap_stats<<<nb, 256, 0, s>>>(pts.as<uint32_t>() + q->off, q->len, dstat.as<stats_t>() + k);
      }
      SMR_HIP(hipGetLastError());
      SMR_HIP(hipMemcpyAsync(host, dstat.p, 2 * sizeof(stats_t), hipMemcpyDeviceToHost, s));
      SMR_HIP(hipStreamSynchronize(s));
      return SMR_OK;
   };
   auto take_box = [](blk_t &b, const stats_t &t) { for (int j = 0; j < 3; j++) { b.lo[j] = 255 - (int) t.nlo[j]; b.hi[j] = (int) t.hi[j]; } };
   auto take_avg = [](blk_t &b, const stats_t &t) { for (int j = 0; j < 3; j++) b.avg[j] = (int) ((long) t.sum[j] / (long) b.len); };
   std::vector<blk_t> blk(1);
   stats_t hs[2];
   blk[0].off = 0; blk[0].len = n;
   SMR_TRY(stats(&blk[0], nullptr, hs));
   take_box(blk[0], hs[0]); take_avg(blk[0], hs[0]);
   int64_t sorted_points = 0;
   while ((int) blk.size() < 127 && blk[0].len > 1)
   {
      const int d = side_index(blk[0]);
      uint32_t *seg = pts.as<uint32_t>() + blk[0].off, *out = alt.as<uint32_t>() + blk[0].off;
      size_t tb = tmp_bytes;
      SMR_HIP(rocprim::radix_sort_keys(tmp.p, tb, seg, out, (size_t) blk[0].len, 8 * d, 8 * d + 8, s));
      SMR_HIP(hipMemcpyAsync(seg, out, blk[0].len * 4, hipMemcpyDeviceToDevice, s));
      sorted_points += blk[0].len;
      const int64_t half = (blk[0].len + 1) / 2;
      blk_t nb = blk[0];
      nb.off = blk[0].off + half; nb.len = blk[0].len - half;
      blk[0].len = half;                                   /* its average stays the parent's, as in the reference (:256-258) */
      SMR_TRY(stats(&blk[0], &nb, hs));
      take_box(blk[0], hs[0]); take_box(nb, hs[1]); take_avg(nb, hs[1]);
      blk.push_back(nb);
      std::stable_sort(blk.begin(), blk.end(), [](const blk_t &a, const blk_t &b) { return side_len(a) > side_len(b); });
   }
   const double t1 = now_ms();

   /* ---- palette and indices ---- */
   pal_t pal;
   memset(&pal, 0, sizeof(pal));
   pal.n = (int) blk.size();
   uint32_t palette[127];
   for (int i = 0; i < pal.n; i++)
   {
      for (int j = 0; j < 3; j++) pal.c[i][j] = (uint8_t) blk[i].avg[j];
      palette[i] = 0xff000000u | (uint32_t) blk[i].avg[0] | (uint32_t) blk[i].avg[1] << 8 | (uint32_t) blk[i].avg[2] << 16;
      int f = 0;
      while (palette[f] != palette[i]) f++;
      pal.first[i] = (uint8_t) (f + 1);
   }
   int depth = 1;
   for (int m = 2; m < pal.n; m <<= 1) depth++;            /* get_depth, src/smkap.c:187-193 */
   dmem_t idx, rlen, sums;
   SMR_TRY(idx.get(n)); SMR_TRY(rlen.get(((size_t) h + 1) * 4)); SMR_TRY(sums.get(((size_t) h / 1024 + 2) * 4 + 16));
   k_kap_index<<<dim3((w + 255) / 256, h), 256, 0, s>>>(c->pix, c->W, w, h, pal, idx.as<uint8_t>());
   SMR_HIP(hipGetLastError());
   SMR_HIP(hipStreamSynchronize(s));
   const double t2 = now_ms();

   /* ---- rows ---- */
   k_kap_rows<false><<<(h + 3) / 4, 256, 0, s>>>(idx.as<uint8_t>(), w, h, depth, nullptr, rlen.as<uint32_t>(), nullptr);
   uint32_t total = 0;
   SMR_TRY(smr_scan_u32(rlen.as<uint32_t>(), (uint32_t) h, sums.as<uint32_t>() + 4, sums.as<uint32_t>(), &total));
   dmem_t body;
   SMR_TRY(body.get(total));
   k_kap_rows<true><<<(h + 3) / 4, 256, 0, s>>>(idx.as<uint8_t>(), w, h, depth, rlen.as<uint32_t>(), nullptr, body.as<uint8_t>());
   SMR_HIP(hipGetLastError());
   std::vector<uint8_t> rows(total);
   std::vector<uint32_t> roff((size_t) h + 1);
   SMR_HIP(hipMemcpyAsync(rows.data(), body.p, total, hipMemcpyDeviceToHost, s));
   SMR_HIP(hipMemcpyAsync(roff.data(), rlen.p, ((size_t) h + 1) * 4, hipMemcpyDeviceToHost, s));
   SMR_HIP(hipStreamSynchronize(s));
   const double t3 = now_ms();

   /* ---- the file, save_kap :216-240 ---- */
   FILE *f = fopen(path, "wb");
   if (!f) return smr_fail(SMR_E_INVAL, "write_kap: cannot open %s", path);
   long off = 0;
   if (header) off += fprintf(f, "%s", header);
   off += fprintf(f, "OST/1\r\nIFM/%d\r\n", depth);
   for (int i = 0; i < pal.n; i++) off += fprintf(f, "RGB/%d,%d,%d,%d\r\n", i + 1, (palette[i] >> 16) & 0xff, (palette[i] >> 8) & 0xff, palette[i] & 0xff);
   off += fprintf(f, "\x1a%c%c", '\0', (char) depth);
   if ((int64_t) off + total > INT32_MAX) { fclose(f); return smr_fail(SMR_E_CAPACITY, "write_kap: %lld bytes of rows (the offset table is 32 bit)", (long long) total); }
   std::vector<int32_t> offp((size_t) h + 1);
   for (int y = 0; y <= h; y++) offp[y] = (int32_t) htonl((uint32_t) (off + roff[y]));
   bool ok = total == 0 || fwrite(rows.data(), total, 1, f) == 1;
   ok = ok && fwrite(offp.data(), sizeof(int32_t), (size_t) h + 1, f) == (size_t) h + 1;
   ok = (fclose(f) == 0) && ok;
   if (!ok) return smr_fail(SMR_E_INVAL, "write_kap: writing %s failed", path);
   if (st)
   {
      st->n_colors = pal.n; st->depth = depth;
      memset(st->palette, 0, sizeof(st->palette));
      memcpy(st->palette, palette, sizeof(uint32_t) * pal.n);
      st->row_bytes = total; st->file_bytes = off + (int64_t) total + 4 * ((int64_t) h + 1); st->sorted_points = sorted_points;
      st->reduce_ms = t1 - t0; st->index_ms = t2 - t1; st->rows_ms = t3 - t2; st->write_ms = now_ms() - t3;
   }
   return SMR_OK;
}
