/* smr_dev.h -- shared device-side definitions of libsmr_hip (gfx950 only). */
#ifndef SMR_DEV_H
#define SMR_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define SMR_TW 64            /* tile width in px  = one wavefront of columns */
#define SMR_TH 4             /* tile height in px = 60 sub-rows, one per lane */
#define SMR_GRID_Y 15        /* sub-rows per pixel row (libcairo tor scan converter) */
#define SMR_NSUB (SMR_TH * SMR_GRID_Y)
#define SMR_BK 30            /* crossings kept per sub-row inside one tile */
#define SMR_BKS 31           /* bucket stride in uint16 (odd dword count would be better; 31 halves keeps banks spread) */
#define SMR_ROWCAP 64        /* active edges per pixel row that the exact full-row path handles */
#define SMR_SLOTS 8          /* stroke edge slots per segment */

/* one polygon edge in device space, 24.8 fixed point; direction carries the winding sign.
 * y1 == y2 marks an empty slot (horizontal edges never contribute). */
struct __attribute__((aligned(16))) smr_edge_t { int32_t x1, y1, x2, y2; };

/* one primitive = one cairo_stroke()/cairo_fill() of the reference (render_poly_line,
 * src/smrules_cairo.c:953-1016): an edge range + a solid source pixel + the row mask */
struct __attribute__((aligned(16))) smr_prim_t
{
   uint64_t edge_off;        /* first edge slot in the edge arena */
   uint32_t edge_cnt;        /* number of slots (empty ones included) */
   uint32_t color;           /* premultiplied ARGB32 source pixel */
   int32_t minx, miny, maxx, maxy;   /* bbox of the emitted vertices, 24.8 */
   uint32_t flags;           /* SMR_PF_* */
   uint32_t mask_off;        /* first word of the row mask in the mask arena */
   int32_t row0, nrows;      /* chart rows the mask covers; bit set = the row is sub-sampled (an edge starts or
                                ends inside it, or edges cross in it), clear = libcairo's exact full-row path */
   uint32_t drow_off;        /* SMR_PF_DELTA: first entry of the primitive's strips in the strip table of k_cover */
   uint32_t pad_[3];
};
#define SMR_PF_VALID 1u
#define SMR_PF_EVENODD 2u
#define SMR_PF_PIECES 4u     /* every 4 aligned edge slots form one closed convex piece, all wound alike (strokes) */
#define SMR_PF_MULTI 8u      /* the primitive has more than one item */
#define SMR_PF_DELTA 16u     /* k_cover has worked out the primitive's coverage: per strip of four chart rows the alpha of every column
                                (the four rows of a column in one word); k_raster only composites */

/* k_cover accumulates per cell what libcairo's tor scan converter does: cover in sub-rows and area in 1/512 cell units, packed
 * as cover * 2^20 + area in one word for the LDS atomics: |cover| <= 15 * 30 crossings and |area| <= 15 * 30 * 510 by the
 * limits of k_cover's lists, so both fit and the sum unpacks without loss */
__device__ __forceinline__ int smr_delta_area(int ca) { return (ca << 12) >> 12; }
__device__ __forceinline__ int smr_delta_cover(int ca) { return (ca - smr_delta_area(ca)) >> 20; }
__device__ __forceinline__ int smr_delta_pack(int dcover, int darea) { return (int) (((uint32_t) dcover << 20) + (uint32_t) darea); }
/* one strip of a covered primitive: its alpha words start at column x0 (may be left of the chart) and there are n of them */
__device__ __forceinline__ uint32_t smr_strip_pack(int x0, int n) { return (uint32_t) (x0 + 256) | ((uint32_t) n << 20); }
__device__ __forceinline__ int smr_strip_x0(uint32_t info) { return (int) (info & 0xfffffu) - 256; }
__device__ __forceinline__ uint32_t smr_strip_n(uint32_t info) { return info >> 20; }

/* one item = up to 64 consecutive edge slots of one primitive: the unit that is binned to
 * tiles and that one wavefront loads with one 16-byte access per lane */
struct __attribute__((aligned(16))) smr_item_t
{
   uint32_t edge_off;        /* first slot (arena index) */
   uint32_t prim;            /* 2 * way + (0 border | 1 fill) */
   uint32_t cnt_flags;       /* slots in bits 0..7, SMR_PF_* << 8 */
   uint32_t color;
   int32_t minx, miny, maxx, maxy;   /* binning box, 24.8: maxx is the PRIMITIVE's for fill chunks (edges left of a
                                        tile carry winding into it) */
};

/* one entry of a tile's sorted work stream: everything the rasterizer needs about (item, tile) in
 * one sequential 32-byte record, so that the wave that rasterizes the tile never chases pointers */
struct __attribute__((aligned(16))) smr_thdr_t
{
   uint32_t edge_off;        /* first edge slot of the item */
   uint32_t prim;
   uint32_t cnt_flags;       /* slots in bits 0..7, SMR_PF_* << 8, bit 16: first primitive of the layer */
   uint32_t color;
   uint32_t rowbits;         /* bits 0..3: the tile's rows that are sub-sampled for this primitive (row mask);
                                bits 4..7: the tile's rows inside the primitive's pixel extents */
   int32_t colmin, colmax;   /* tile columns inside the primitive's pixel extents (may be outside 0..63) */
   uint32_t dcounts;         /* SMR_PF_DELTA: smr_strip_pack of the tile's strip of the primitive; edge_off is then its first alpha word */
};

/* status word bits reported back to the host */
#define SMR_ST_BUCKET_OVERFLOW 1u
#define SMR_ST_ROWCAP 2u
#define SMR_ST_MASK_ARENA 4u
#define SMR_ST_KNOTS_CLOSED 32u  /* a closed way of explicit Bezier knots (curve=2) that is not p0, (c1, c2, p) * m, p0 */
#define SMR_ST_CURVE_DEPTH 16u  /* a Bezier needed more levels of subdivision than the decomposition's stack holds */
#define SMR_ST_ROWRISK 64u      /* a stroke's vertex-free row with more than SMR_ROWCAP spanning edges was found free of crossings: it takes the
                                   exact whole-row path, which holds SMR_ROWCAP of them per tile (draw_pass keeps a copy of the canvas) */
#define SMR_ST_BADBATCH 128u    /* the way table does not start at 0 / ascend / end at n_refs, or a ref lies outside [0, n_nodes) */
#define SMR_ST_FARAWAY 256u     /* a node more than SMR_FAR_PX beyond the chart: beyond what the edge arithmetic holds exactly */
#define SMR_DASH_ENDLESS 1e12   /* pt: the one dash of a solid stroke that goes through the dasher for its round caps (oracle: ORC_DASH_ENDLESS) */
#define SMR_FAR_PX 32768.0      /* with charts of up to 65 536 px an edge then spans less than the 139 000 px whose 60 * dy (24.8) fit 31 bits */
#define SMR_ST_DEGRADED 8u      /* a compound row with more than SMR_ROWCAP spanning edges was sub-sampled */

__device__ __forceinline__ int smr_to_grid_y(int32_t y) { return (int) ((15ll * y + 128) >> 8); }

/* floor division of a 64-bit numerator by a positive 31-bit denominator through fp64 with
 * an exact integer correction */
__device__ __forceinline__ void smr_floordiv(int64_t n, int32_t d, int64_t &q, int32_t &r)
{
   double qd = floor((double) n / (double) d);
   q = (int64_t) qd;
   int64_t rr = n - q * (int64_t) d;
   if (rr < 0) { q--; rr += d; }
   else if (rr >= d) { q++; rr -= d; }
   if (rr < 0) { q--; rr += d; }
   else if (rr >= d) { q++; rr -= d; }
   r = (int32_t) rr;
}

/* oriented (downward) edge, derived form.  D = 60*dy: libcairo's Ey / 128 (all remainders
 * of the tor scan converter are exact multiples of 128, so the reduced form is bit-equal). */
struct smr_oedge_t
{
   int32_t p1x, p1y, dxs, D, ytop, ybot, dir;   /* dir == 0: empty */
};

__device__ __forceinline__ smr_oedge_t smr_orient(const smr_edge_t e)
{
   smr_oedge_t o;
   int32_t ax = e.x1, ay = e.y1, bx = e.x2, by = e.y2;
   o.dir = 1;
   if (ay > by) { int32_t t = ax; ax = bx; bx = t; t = ay; ay = by; by = t; o.dir = -1; }
   o.p1x = ax; o.p1y = ay; o.dxs = bx - ax; o.D = 60 * (by - ay);
   o.ytop = smr_to_grid_y(ay); o.ybot = smr_to_grid_y(by);
   if (ay == by || o.ytop >= o.ybot) o.dir = 0;
   return o;
}

/* x of the edge at the centre of sub-row ys: quo + rem/D, 0 <= rem < D */
__device__ __forceinline__ void smr_edge_x(int32_t p1x, int32_t p1y, int32_t dxs, int32_t D, int ys, int64_t &quo, int32_t &rem)
{
   if (dxs == 0) { quo = p1x; rem = 0; return; }
   int64_t diff = ((int64_t) (2 * (int64_t) ys + 1) << 8) - (int64_t) p1y * 30;
   int64_t q;
   smr_floordiv(2 * diff * (int64_t) dxs, D, q, rem);
   quo = q + p1x;
}

__device__ __forceinline__ int32_t smr_cell(int64_t quo, int32_t rem, int32_t D, int32_t dxs)
{
   return (int32_t) quo + ((dxs != 0 && rem >= (D >> 1)) ? 1 : 0);
}

/* Fast exact form of smr_edge_x + smr_cell for a lane that evaluates many sub-rows of one edge.
 * Per edge (computed once by the lane that owns the edge): c = 256 - 30*p1y, dxs2 = 2*dxs,
 * invD = 1/D.  x(ys) = p1x + floor(dxs2 * (512*ys + c) / D); the quotient fits 32 bits
 * (|q| <= |dxs|), so one fp64 multiply estimates it to +-1 and one exact integer remainder
 * corrects it.  Valid while |coordinates| < 2^20 px (the numerator stays below 2^63). */
__device__ __forceinline__ int32_t smr_edge_cell(int64_t c, int32_t dxs2, int32_t D, double invD, int32_t p1x, int ys)
{
   const int64_t t = (int64_t) ys * 512 + c;
   const int64_t n = t * (int64_t) dxs2;
   const double nd = fma((double) (int32_t) (n >> 32), 4294967296.0, (double) (uint32_t) n);
   int32_t q = (int32_t) floor(nd * invD);
   int64_t rem = n - (int64_t) q * D;
   if (rem < 0) { q--; rem += D; }
   else if (rem >= D) { q++; rem -= D; }
   return p1x + q + (rem >= (int64_t) (D >> 1) ? 1 : 0);
}

/* same evaluation, returning the quotient (without p1x) and the remainder: x = p1x + q + rem / D */
__device__ __forceinline__ void smr_edge_qr(int64_t c, int32_t dxs2, int32_t D, double invD, int ys, int32_t &q, int32_t &rem)
{
   const int64_t t = (int64_t) ys * 512 + c;
   const int64_t n = t * (int64_t) dxs2;
   const double nd = fma((double) (int32_t) (n >> 32), 4294967296.0, (double) (uint32_t) n);
   int32_t qq = (int32_t) floor(nd * invD);
   int64_t r = n - (int64_t) qq * D;
   if (r < 0) { qq--; r += D; }
   else if (r >= D) { qq++; r -= D; }
   q = qq; rem = (int32_t) r;
}

/* hypot as glibc 2.35 computes it (sysdeps/ieee754/dbl-64/e_hypot.c, the kernel without FMA, which is the one x86-64
 * builds run): libcairo normalises slopes with the C library's hypot, and a join of two nearly parallel faces turns the
 * last bit of that into miter-or-bevel.  Every operation below is a correctly rounded IEEE one on the device too, so the
 * result is bit-equal (checked against glibc on 2*10^7 inputs); the operands here are far from the
 * library's overflow / underflow rescaling branches. */
__device__ __forceinline__ double smr_hypot(double x, double y)
{
   x = fabs(x); y = fabs(y);
   const double ax = x < y ? y : x, ay = x < y ? x : y;
   if (ax >= ay / 0x1p-54) return ax + ay;
   double h = sqrt(ax * ax + ay * ay), t1, t2;
   if (h <= 2.0 * ay)
   {
      const double delta = h - ay;
      t1 = ax * (2.0 * delta - ax);
      t2 = (delta - 2.0 * (ax - ay)) * delta;
   }
   else
   {
      const double delta = h - ax;
      t1 = 2.0 * delta * (ax - 2.0 * ay);
      t2 = (4.0 * delta - ay) * ay + delta * delta;
   }
   h -= (t1 + t2) / (2.0 * h);
   return h;
}

/* 1/d for the quotient ESTIMATES of the edge evaluation (every one of them is corrected by its exact integer remainder):
 * v_rcp_f64 + two Newton steps, a few instructions instead of the ~35 of the IEEE division */
__device__ __forceinline__ double smr_rcp(double d)
{
   double r = __builtin_amdgcn_rcp(d);
   r = fma(fma(-d, r, 1.0), r, r);
   r = fma(fma(-d, r, 1.0), r, r);
   return r;
}

__device__ __forceinline__ double smr_i64_to_f64(int64_t n)
{
   return fma((double) (int32_t) (n >> 32), 4294967296.0, (double) (uint32_t) n);
}

/* inclusive prefix sum over the 64 lanes with DPP row shifts / broadcasts (VALU latency only; the
 * ds_bpermute form of __shfl_up costs an LDS round trip per step) */
__device__ __forceinline__ int smr_wave_scan(int v)
{
   v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      /* row_shr:1 */
   v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      /* row_shr:2 */
   v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      /* row_shr:4 */
   v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      /* row_shr:8 */
   v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      /* row_bcast:15 -> rows 1, 3 */
   v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      /* row_bcast:31 -> rows 2, 3 */
   return v;
}

/* wave-local ordering of LDS traffic: a wavefront's DS operations execute in program order, so
 * only the compiler has to be held back; no s_barrier (waves of one workgroup are independent) */
#define SMR_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                             __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

/* ---- 8-bit compositing arithmetic (libcairo image spans / pixman 0.40) ---- */
__device__ __forceinline__ uint32_t smr_mul7f(uint32_t a, uint32_t b) { uint32_t t = a * b + 0x7f; return (t + (t >> 8)) >> 8; }
__device__ __forceinline__ uint32_t smr_mul80(uint32_t a, uint32_t b) { uint32_t t = a * b + 0x80; return (t + (t >> 8)) >> 8; }
__device__ __forceinline__ uint32_t smr_sat(uint32_t v) { return v > 255u ? 255u : v; }

/* two 8-bit channels per 32-bit word (0x00XX00YY) in the packed 16-bit VALU (v_pk_mad_u16, v_pk_lshrrev_b16, v_pk_add_u16,
 * v_pk_min_u16): every intermediate of pixman's MUL_UN8 / ADD_UN8 stays below 2^16, so the two halves never meet and the
 * per-channel results are the ones of smr_mul7f / smr_mul80 / smr_sat bit for bit */
#define SMR_RB 0x00ff00ffu
typedef unsigned short smr_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ smr_u16x2 smr_pk(uint32_t x) { return __builtin_bit_cast(smr_u16x2, x); }
__device__ __forceinline__ uint32_t smr_unpk(smr_u16x2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ smr_u16x2 smr_pk_mul(smr_u16x2 x, smr_u16x2 a, unsigned short half)
{
   smr_u16x2 t = x * a + (smr_u16x2) (half);
   t = t + (t >> 8);
   return t >> 8;
}
__device__ __forceinline__ smr_u16x2 smr_pk_addsat(smr_u16x2 x, smr_u16x2 y) { return __builtin_elementwise_min((smr_u16x2) (x + y), (smr_u16x2) (255)); }
__device__ __forceinline__ smr_u16x2 smr_pk_rb(uint32_t p) { return smr_pk(p & SMR_RB); }
__device__ __forceinline__ smr_u16x2 smr_pk_ag(uint32_t p) { return smr_pk(p) >> 8; }
__device__ __forceinline__ uint32_t smr_pk_join(smr_u16x2 rb, smr_u16x2 ag) { return smr_unpk(rb) | (smr_unpk(ag) << 8); }

__device__ __forceinline__ uint32_t smr_blend_lerp(uint32_t s, uint32_t a, uint32_t d)
{
   if (a == 255u) return s;
   const smr_u16x2 a2 = smr_pk(a * 0x10001u), ia2 = smr_pk(SMR_RB - a * 0x10001u);
   const smr_u16x2 rb = smr_pk_addsat(smr_pk_mul(smr_pk_rb(s), a2, 0x7f), smr_pk_mul(smr_pk_rb(d), ia2, 0x7f));
   const smr_u16x2 ag = smr_pk_addsat(smr_pk_mul(smr_pk_ag(s), a2, 0x7f), smr_pk_mul(smr_pk_ag(d), ia2, 0x7f));
   return smr_pk_join(rb, ag);
}

/* the same two without their shortcuts, for code that runs them on whole wavefronts: pixman's MUL_UN8 gives x for x * 255 and
 * 0 for x * 0 (with either rounding constant), so a = 255 and a = 0 come out as s and d by the arithmetic itself */
__device__ __forceinline__ uint32_t smr_blend_lerp_all(uint32_t s, uint32_t a, uint32_t d)
{
   const smr_u16x2 a2 = smr_pk(a * 0x10001u), ia2 = smr_pk(SMR_RB - a * 0x10001u);
   const smr_u16x2 rb = smr_pk_addsat(smr_pk_mul(smr_pk_rb(s), a2, 0x7f), smr_pk_mul(smr_pk_rb(d), ia2, 0x7f));
   const smr_u16x2 ag = smr_pk_addsat(smr_pk_mul(smr_pk_ag(s), a2, 0x7f), smr_pk_mul(smr_pk_ag(d), ia2, 0x7f));
   return smr_pk_join(rb, ag);
}

__device__ __forceinline__ uint32_t smr_blend_over_mask_all(uint32_t s, uint32_t m, uint32_t d)
{
   const smr_u16x2 m2 = smr_pk(m * 0x10001u);
   const smr_u16x2 srb = smr_pk_mul(smr_pk_rb(s), m2, 0x80), sag = smr_pk_mul(smr_pk_ag(s), m2, 0x80);
   const smr_u16x2 ia2 = smr_pk((255u - (smr_unpk(sag) >> 16)) * 0x10001u);
   const smr_u16x2 rb = smr_pk_addsat(srb, smr_pk_mul(smr_pk_rb(d), ia2, 0x80));
   const smr_u16x2 ag = smr_pk_addsat(sag, smr_pk_mul(smr_pk_ag(d), ia2, 0x80));
   return smr_pk_join(rb, ag);
}

__device__ __forceinline__ uint32_t smr_blend_over_mask(uint32_t s, uint32_t m, uint32_t d)
{
   smr_u16x2 srb = smr_pk_rb(s), sag = smr_pk_ag(s);
   if (m != 255u)
   {
      const smr_u16x2 m2 = smr_pk(m * 0x10001u);
      srb = smr_pk_mul(srb, m2, 0x80); sag = smr_pk_mul(sag, m2, 0x80);
   }
   const smr_u16x2 ia2 = smr_pk((255u - (smr_unpk(sag) >> 16)) * 0x10001u);
   const smr_u16x2 rb = smr_pk_addsat(srb, smr_pk_mul(smr_pk_rb(d), ia2, 0x80));
   const smr_u16x2 ag = smr_pk_addsat(sag, smr_pk_mul(smr_pk_ag(d), ia2, 0x80));
   return smr_pk_join(rb, ag);
}

__device__ __forceinline__ uint32_t smr_blend_over(uint32_t s, uint32_t d)
{
   const uint32_t a = s >> 24;
   if (a == 255u) return s;
   if (s == 0u) return d;
   const smr_u16x2 ia2 = smr_pk((255u - a) * 0x10001u);
   const smr_u16x2 rb = smr_pk_addsat(smr_pk_rb(s), smr_pk_mul(smr_pk_rb(d), ia2, 0x80));
   const smr_u16x2 ag = smr_pk_addsat(smr_pk_ag(s), smr_pk_mul(smr_pk_ag(d), ia2, 0x80));
   return smr_pk_join(rb, ag);
}

/* XCD-aware, bijective blockIdx -> work item map: blocks b and b+8 share an XCD (and its
 * L2), so give every XCD one contiguous range of work items */
__device__ __forceinline__ uint32_t smr_xcd_remap(uint32_t b, uint32_t n)
{
   uint32_t q = n >> 3, r = n & 7u, xcd = b & 7u, k = b >> 3;
   uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
   return base + k;
}

#endif
