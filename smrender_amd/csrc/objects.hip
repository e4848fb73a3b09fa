/* objects.hip -- SURVEY 8f rank 3: the host's object tree exported ONCE as structure-of-arrays, and the rule executor's
 * filter (apply_rule, src/smcore.c:248-312, with bs_match_attr / bs_match, libsmrender/smutil.c:282-356) evaluated for all
 * objects and many rules at once on the device, so that no `_main` has to be called per object to find the few a rule is for.
 *
 * Layout.  Tags are dictionary encoded: every distinct byte string (key or value) gets an id; an object's tags are pairs of
 * ids.  What a rule's match tag asks of a string (equal to a pattern, matched by a regular expression, greater or less than a
 * number) then depends on the id alone: equality is a comparison of ids (the pattern is looked up in the dictionary once),
 * the number bs_tod() reads from a string is computed once per string on the device, and a regular expression runs once per
 * distinct string on the host -- regexec() on the rule's own compiled regex_t, the call the reference makes per object -- and
 * comes to the device as one bit per string.  The per-object work on the device is gathers and integer compares.
 *
 * Way refs are node ids; the device turns them into node indices by binary search in the id-sorted node table.
 */
#include "smr_host.h"
#include "smr_abi.h"
#include <string.h>
#include <string>
#include <chrono>
#include <thread>
#include <algorithm>
#include <rocprim/rocprim.hpp>

namespace {

/* ------------------------------------------------------------------------------------------------------------ dictionary */
inline uint64_t mix64(uint64_t x) { x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; return x; }

inline uint64_t hash_bytes(const char *s, int len)
{
   uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t) len;
   int i = 0;
   for (; i + 8 <= len; i += 8) { uint64_t w; memcpy(&w, s + i, 8); h = mix64(h ^ w); }
   if (i < len) { uint64_t w = 0; memcpy(&w, s + i, len - i); h = mix64(h ^ w); }
   return mix64(h);
}

struct dict_t
{
   std::vector<char> pool;             /* strings one after the other, each followed by a NUL (for regexec) */
   std::vector<uint64_t> off { 0 };    /* [n + 1]: string i = pool[off[i], off[i+1] - 1) */
   std::vector<uint64_t> hash;
   std::vector<uint8_t> role;          /* bit 0: seen as a key, bit 1: seen as a value */
   std::vector<uint32_t> slot;         /* open addressing, id + 1 */
   uint64_t mask = 0;

   dict_t() { slot.assign(1024, 0); mask = 1023; }
   size_t size() const { return hash.size(); }
   int len(uint32_t id) const { return (int) (off[id + 1] - off[id] - 1); }
   const char *str(uint32_t id) const { return pool.data() + off[id]; }

   void grow()
   {
      std::vector<uint32_t> ns(slot.size() * 2, 0);
      const uint64_t nm = ns.size() - 1;
      for (uint32_t id = 0; id < hash.size(); id++)
      {
         uint64_t p = hash[id] & nm;
         while (ns[p]) p = (p + 1) & nm;
         ns[p] = id + 1;
      }
      slot.swap(ns); mask = nm;
   }
   uint32_t find(const char *s, int n, uint64_t h) const
   {
      for (uint64_t p = h & mask; slot[p]; p = (p + 1) & mask)
      {
         const uint32_t id = slot[p] - 1;
         if (hash[id] == h && len(id) == n && memcmp(str(id), s, n) == 0) return id;
      }
      return UINT32_MAX;
   }
   uint32_t intern(const char *s, int n, uint64_t h, uint8_t r)
   {
      uint64_t p = h & mask;
      for (; slot[p]; p = (p + 1) & mask)
      {
         const uint32_t id = slot[p] - 1;
         if (hash[id] == h && len(id) == n && memcmp(str(id), s, n) == 0) { role[id] |= r; return id; }
      }
      const uint32_t id = (uint32_t) hash.size();
      slot[p] = id + 1;
      hash.push_back(h); role.push_back(r);
      pool.insert(pool.end(), s, s + n); pool.push_back('\0');
      off.push_back(pool.size());
      if (hash.size() * 2 > slot.size()) grow();
      return id;
   }
};

/* what one thread gathers from its share of the objects (ids of ITS dictionary until the merge) */
struct part_t
{
   dict_t d;
   std::vector<uint32_t> ntag_cnt, wtag_cnt, ref_cnt;
   std::vector<int32_t> nk, nv, wk, wv;
   std::vector<int64_t> refs;
   std::vector<uint32_t> remap;
   bool ascending = true;
};

inline void take_tags(const smr_osm_obj_t *o, dict_t &d, std::vector<int32_t> &k, std::vector<int32_t> &v, std::vector<uint32_t> &cnt)
{
   const int n = o->tag_cnt > 0 && o->otag ? o->tag_cnt : 0;
   for (int i = 0; i < n; i++)
   {
      const smr_bstring_t &bk = o->otag[i].k, &bv = o->otag[i].v;
      const int kl = bk.len > 0 && bk.buf ? bk.len : 0, vl = bv.len > 0 && bv.buf ? bv.len : 0;
      k.push_back((int32_t) d.intern(kl ? bk.buf : "", kl, hash_bytes(kl ? bk.buf : "", kl), 1));
      v.push_back((int32_t) d.intern(vl ? bv.buf : "", vl, hash_bytes(vl ? bv.buf : "", vl), 2));
   }
   cnt.push_back((uint32_t) n);
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

/* ------------------------------------------------------------------------------------------------------------ kernels */
/* bs_tod (libsmrender/bstring.c:156-188), operation for operation: the digits accumulate by *10 and +, a first '.' starts the
 * count of decimals (a second one counts as a decimal place, as there), the first other byte ends the number, then one
 * division by 10 per decimal */
__global__ __launch_bounds__(256) void k_dict_tod(const uint8_t *__restrict__ pool, const uint64_t *__restrict__ off, int64_t n, double *__restrict__ tod)
{
   const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   const uint8_t *s = pool + off[i];
   int64_t len = (int64_t) (off[i + 1] - off[i]) - 1;
   int neg = 0, e = -1;
   double d = 0.0;
   if (len && *s == '-') { s++; len--; neg = 1; }
   for (; len; s++, len--)
   {
      const uint8_t c = *s;
      if (c == '.') { e++; continue; }
      if (c < '0' || c > '9') break;
      if (e >= 0) e++;
      d *= 10.0;
      d += (double) (c - '0');
   }
   for (; e > 0; e--) d /= 10.0;
   tod[i] = neg ? -d : d;
}

/* node index of every ref: binary search in the ascending ids (sorted_idx NULL: the nodes themselves are in id order) */
__global__ __launch_bounds__(256) void k_ref_index(const int64_t *__restrict__ ref_id, int64_t n_refs, const int64_t *__restrict__ ids,
                                                   const int32_t *__restrict__ sorted_idx, int64_t n_nodes, int32_t *__restrict__ ref, uint32_t *__restrict__ missing)
{
   const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n_refs) return;
   const int64_t want = ref_id[i];
   int64_t lo = 0, hi = n_nodes;
   while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ids[mid] < want) lo = mid + 1; else hi = mid; }
   int32_t r = -1;
   if (lo < n_nodes && ids[lo] == want) r = sorted_idx ? sorted_idx[lo] : (int32_t) lo;
   ref[i] = r;
   if (r < 0) atomicAdd(missing, 1u);
}

__global__ __launch_bounds__(256) void k_iota(int32_t *v, int64_t n)
{
   const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) v[i] = (int32_t) i;
}

/* one side (key or value) of a rule's match tag: struct specialTag (libsmrender/smaction.h:75-83) with the pattern resolved */
struct rside_t { int32_t has, type, pat, pad_; double val; };
struct rtag_t { rside_t k, v; };
struct rdesc_t { int32_t tag0, ntags, wayflag, empty_ok; int64_t row; };      /* empty_ok: an object without tags passes (every match tag is a ~not~) */
enum { SP_DIRECT = SMR_SPECIAL_DIRECT, SP_REGEX = SMR_SPECIAL_REGEX, SP_GT = SMR_SPECIAL_GT, SP_LT = SMR_SPECIAL_LT, SP_MASK = SMR_SPECIAL_MASK,
       SP_NOT = SMR_SPECIAL_NOT, SP_INVERT = SMR_SPECIAL_INVERT };

struct match_args_t
{
   const int64_t *tag_off; const int32_t *kid, *vid; const uint8_t *vis, *extra;   /* extra: nodes = on the page (may be NULL), ways = 0 none / 1 open / 2 closed */
   int64_t n_obj;
   const rdesc_t *rules; const rtag_t *rtags; int n_rules, is_way;
   const double *tod; const uint32_t *rbits; int64_t rwords;
   uint64_t *bitmap; unsigned long long *counts;      /* counts: k_row_count's */
};

/* bs_match (libsmrender/smutil.c:282-318) on the id of the object's string */
__device__ __forceinline__ int side_match(const rside_t &s, int32_t id, const match_args_t &a)
{
   int r = 1;
   switch (s.type & SP_MASK)
   {
      case SP_DIRECT: r = id != s.pat; break;
      case SP_REGEX: r = !((a.rbits[(int64_t) s.pat * a.rwords + (id >> 5)] >> (id & 31)) & 1u); break;
      case SP_GT: r = !(a.tod[id] > s.val); break;
      case SP_LT: r = !(a.tod[id] < s.val); break;
   }
   return (s.type & SP_INVERT) ? r != 0 : r == 0;
}

#define MATCH_RULES_PER_BLOCK 64
#define MATCH_LDS_TAGS 256
#define MATCH_OBJ_ITERS 4          /* a block takes 4 x 256 objects, one after the other */

/* apply_rule's filter (src/smcore.c:248-291) for 1024 objects and MATCH_RULES_PER_BLOCK rules per block: page test for nodes,
 * open/closed for ways, every match tag through bs_match_attr (libsmrender/smutil.c:330-356), visibility.
 * Most nodes carry no tag, and what a rule says to a tagless object does not depend on the object (rdesc_t.empty_ok): their
 * verdicts are a few bitwise operations on 64-object words.  The objects that do have tags are gathered into a list and take
 * the lanes densely, one object per lane; the rule records wait in LDS (all lanes read the same one).  A rule's 16 words of
 * the block leave as one 128-byte piece of its bitmap row */
__global__ __launch_bounds__(256) void k_match(match_args_t a)
{
   constexpr int WPB = MATCH_OBJ_ITERS * 4;
   __shared__ rdesc_t s_rule[MATCH_RULES_PER_BLOCK];
   __shared__ rtag_t s_tag[MATCH_LDS_TAGS];
   __shared__ unsigned long long s_word[MATCH_RULES_PER_BLOCK][WPB];
   __shared__ uint64_t s_base[WPB], s_tagged[WPB], s_open[WPB], s_closed[WPB];
   __shared__ uint16_t s_list[256 * MATCH_OBJ_ITERS];
   __shared__ int s_n;
   const int rb = blockIdx.y * MATCH_RULES_PER_BLOCK, nr = min(a.n_rules - rb, MATCH_RULES_PER_BLOCK);
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   if ((int) threadIdx.x < nr) s_rule[threadIdx.x] = a.rules[rb + threadIdx.x];
   if (threadIdx.x == 0) s_n = 0;
   __syncthreads();
   const int tag_base = s_rule[0].tag0, n_tags = s_rule[nr - 1].tag0 + s_rule[nr - 1].ntags - tag_base;
   const bool in_lds = n_tags <= MATCH_LDS_TAGS;
   if (in_lds) for (int t = threadIdx.x; t < n_tags; t += 256) s_tag[t] = a.rtags[tag_base + t];

   /* what does not depend on the rule, 64 objects per word, and the list of the objects with tags */
   const int64_t obj0 = (int64_t) blockIdx.x * (256 * MATCH_OBJ_ITERS);
   for (int it = 0; it < MATCH_OBJ_ITERS; it++)
   {
      const int64_t i = obj0 + it * 256 + threadIdx.x;
      bool base = false, tagged = false;
      int kind = 0;
      if (i < a.n_obj)
      {
         const int extra = a.extra ? a.extra[i] : 1;
         base = a.vis[i] != 0 && (a.is_way || extra != 0);          /* invisible, or a node outside the page (ERULE_OUTOFBBOX) */
         tagged = base && a.tag_off[i + 1] > a.tag_off[i];
         kind = a.is_way ? extra : 0;
      }
      const uint64_t wb = __ballot(base), wt = __ballot(tagged), wo = __ballot(kind == 1), wc = __ballot(kind == 2);
      int at = 0;
      if (lane == 0)
      {
         s_base[it * 4 + wave] = wb; s_tagged[it * 4 + wave] = wt; s_open[it * 4 + wave] = wo; s_closed[it * 4 + wave] = wc;
         if (wt) at = atomicAdd(&s_n, __popcll(wt));
      }
      at = __shfl(at, 0);
      if (tagged) s_list[at + __popcll(wt & ((1ull << lane) - 1))] = (uint16_t) (it * 256 + threadIdx.x);
   }
   __syncthreads();
   const int n_list = s_n;
   for (int k = threadIdx.x; k < nr * WPB; k += 256)
   {
      const rdesc_t R = s_rule[k / WPB];
      const int w = k % WPB;
      const uint64_t excl = R.wayflag == 2 ? s_open[w] : R.wayflag == 1 ? s_closed[w] : 0;       /* ACTION_CLOSED_WAY meets an open way, ACTION_OPEN_WAY a closed one */
      s_word[k / WPB][w] = R.empty_ok ? (s_base[w] & ~s_tagged[w] & ~excl) : 0;
   }
   __syncthreads();

   for (int k0 = 0; k0 < n_list; k0 += 256)
   {
      const int k = k0 + threadIdx.x;
      const bool live = k < n_list;
      const int o = live ? s_list[k] : 0;
      int64_t t0 = 0, t1 = 0;
      int kind = 0;
      if (live)
      {
         t0 = a.tag_off[obj0 + o]; t1 = a.tag_off[obj0 + o + 1];
         kind = a.is_way ? a.extra[obj0 + o] : 0;
      }
      if (!__any(live)) continue;
      for (int r = 0; r < nr; r++)
      {
         const rdesc_t R = s_rule[r];
         bool ok = live && !(R.wayflag == 2 && kind == 1) && !(R.wayflag == 1 && kind == 2);
         for (int t = 0; t < R.ntags && __any(ok); t++)
         {
            const rtag_t T = in_lds ? s_tag[R.tag0 - tag_base + t] : a.rtags[R.tag0 + t];
            if (!ok) continue;
            int res = -1;
            bool done = false;
            for (int64_t j = t0; j < t1; j++)
            {
               const int km = T.k.has ? side_match(T.k, a.kid[j], a) : 1;
               const int vm = T.v.has ? side_match(T.v, a.vid[j], a) : 1;
               if (km && (T.k.type & SP_NOT)) { res = -1; done = true; break; }
               if (vm && (T.v.type & SP_NOT)) { res = -1; done = true; break; }
               if (km && vm) { res = (int) (j - t0); done = true; break; }
            }
            if (!done && ((T.k.type & SP_NOT) || (T.v.type & SP_NOT))) res = 0x7fffffff;
            if (res == -1) ok = false;
         }
         if (ok) atomicOr(&s_word[r][o >> 6], 1ull << (o & 63));
      }
   }
   __syncthreads();
   const int64_t n_words = (a.n_obj + 63) / 64;
   for (int k = threadIdx.x; k < nr * WPB; k += 256)
   {
      const int64_t w = (int64_t) blockIdx.x * WPB + k % WPB;
      if (w < n_words) a.bitmap[s_rule[k / WPB].row + w] = s_word[k / WPB][k % WPB];
   }
}

/* the 1s of every rule's row (counted from the finished rows: a rule that most objects pass would otherwise send one atomic
 * per wavefront to one address) */
__global__ __launch_bounds__(256) void k_row_count(const uint64_t *__restrict__ bitmap, const rdesc_t *__restrict__ rules, int64_t n_words, unsigned long long *__restrict__ counts)
{
   __shared__ uint32_t part[4];
   const uint64_t *row = bitmap + rules[blockIdx.y].row;
   uint32_t c = 0;
   for (int64_t i = (int64_t) blockIdx.x * 4096 + threadIdx.x, e = min(n_words, ((int64_t) blockIdx.x + 1) * 4096); i < e; i += 256) c += (uint32_t) __popcll(row[i]);
   for (int d = 32; d; d >>= 1) c += __shfl_xor(c, d);
   if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
   __syncthreads();
   if (threadIdx.x == 0 && part[0] + part[1] + part[2] + part[3]) atomicAdd(&counts[blockIdx.y], (unsigned long long) part[0] + part[1] + part[2] + part[3]);
}

/* ------------------------------------------------------------------------------------------------------------ selection */
__global__ __launch_bounds__(256) void k_word_pop(const uint64_t *__restrict__ row, int64_t n_words, uint32_t *__restrict__ pop)
{
   const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n_words) pop[i] = (uint32_t) __popcll(row[i]);
}

/* rank of every selected object among the selected; for ways also the number of refs whose node exists (poly_line passes
 * over the others, src/smrules_cairo.c:853-857) */
__global__ __launch_bounds__(256) void k_sel_rank(const uint64_t *__restrict__ row, const uint32_t *__restrict__ word_rank, int64_t n_obj,
                                                  const int64_t *__restrict__ way_off, const int32_t *__restrict__ ref,
                                                  int32_t *__restrict__ index, uint32_t *__restrict__ cnt)
{
   const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n_obj) return;
   const uint64_t w = row[i >> 6];
   if (!((w >> (i & 63)) & 1)) return;
   const uint32_t rank = word_rank[i >> 6] + (uint32_t) __popcll(w & ((1ull << (i & 63)) - 1));
   index[rank] = (int32_t) i;
   if (cnt)
   {
      uint32_t c = 0;
      for (int64_t j = way_off[i]; j < way_off[i + 1]; j++) c += ref[j] >= 0;
      cnt[rank] = c;
   }
}

/* the selected ways' refs, one wavefront per way */
__global__ __launch_bounds__(256) void k_sel_copy(const int32_t *__restrict__ index, const uint32_t *__restrict__ new_off, int64_t n_sel,
                                                  const int64_t *__restrict__ way_off, const int32_t *__restrict__ ref,
                                                  int64_t *__restrict__ out_off, int32_t *__restrict__ out_ref)
{
   const int64_t s = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   const int lane = threadIdx.x & 63;
   if (s > n_sel) return;
   if (s == n_sel) { if (lane == 0) out_off[s] = new_off[s]; return; }
   const int64_t w = index[s], a = way_off[w], b = way_off[w + 1];
   uint32_t o = new_off[s];
   if (lane == 0) out_off[s] = o;
   for (int64_t j = a; j < b; j += 64)
   {
      const int32_t v = j + lane < b ? ref[j + lane] : -1;
      const uint64_t m = __ballot(v >= 0);
      if (v >= 0) out_ref[o + __popcll(m & ((1ull << lane) - 1))] = v;
      o += (uint32_t) __popcll(m);
   }
}

struct dbuf_t
{
   void *p = nullptr; size_t cap = 0;
   int need(size_t bytes)
   {
      if (bytes < 256) bytes = 256;
      if (cap >= bytes) return SMR_OK;
      if (p) { hipStreamSynchronize(smr_g.stream); hipFree(p); p = nullptr; cap = 0; }
      const size_t want = bytes + bytes / 8;
      if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return smr_fail(SMR_E_NOMEM, "objects: hipMalloc of %zu bytes failed", want); }
      cap = want;
      return SMR_OK;
   }
   void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
   template <class T> T *as() const { return (T *) p; }
};

template <class T> int upload(dbuf_t &b, const std::vector<T> &v)
{
   SMR_TRY(b.need(v.size() * sizeof(T)));
   if (!v.empty()) SMR_HIP(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, smr_g.stream));
   return SMR_OK;
}

inline unsigned blocks(int64_t n, int per = 256) { return (unsigned) ((n + per - 1) / per); }

}  // namespace

struct smr_objects
{
   int64_t n_nodes = 0, n_ways = 0, n_refs = 0, n_ntags = 0, n_wtags = 0, missing = 0;
   dict_t dict;
   dbuf_t node_id, lat, lon, nvis, ntag_off, nk, nv;
   dbuf_t way_id, way_off, ref_id, ref, wvis, wkind, wtag_off, wk, wv;
   dbuf_t pool, str_off, tod, sorted_id, sorted_idx, sort_tmp, misc;
   dbuf_t rules, rtags, rbits, counts;                   /* per match call */
   dbuf_t pop, sums, sel_index, sel_cnt, sel_off, sel_ref;        /* per selection */
   double extract_ms = 0, upload_ms = 0, index_ms = 0, match_host_ms = 0, match_device_ms = 0;
   bool nodes_ascending = true;

   void release()
   {
      for (dbuf_t *b : { &node_id, &lat, &lon, &nvis, &ntag_off, &nk, &nv, &way_id, &way_off, &ref_id, &ref, &wvis, &wkind, &wtag_off, &wk, &wv,
                         &pool, &str_off, &tod, &sorted_id, &sorted_idx, &sort_tmp, &misc, &rules, &rtags, &rbits, &counts, &pop, &sums, &sel_index, &sel_cnt, &sel_off, &sel_ref })
         b->release();
   }
};

static int objects_build(smr_objects *o, const void *const *nodes, int64_t n_nodes, const void *const *ways, int64_t n_ways, int threads)
{
   const double t_begin = now_ms();
   if (threads <= 0) threads = (int) std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
   if ((int64_t) threads > std::max<int64_t>(1, (n_nodes + n_ways) / 4096)) threads = (int) std::max<int64_t>(1, (n_nodes + n_ways) / 4096);
   if (n_nodes >= INT32_MAX || n_ways >= INT32_MAX) return smr_fail(SMR_E_CAPACITY, "objects: %lld nodes / %lld ways (indices are 32 bit)", (long long) n_nodes, (long long) n_ways);

   /* pass 1, one share of the nodes and of the ways per thread: geometry into the final arrays, tags into the thread's dictionary */
   std::vector<int64_t> node_id(n_nodes), way_id(n_ways);
   std::vector<double> lat(n_nodes), lon(n_nodes);
   std::vector<uint8_t> nvis(n_nodes), wvis(n_ways), wkind(n_ways);
   std::vector<part_t> part(threads);
   auto share = [&](int64_t n, int t) { return n * t / threads; };
   {
      std::vector<std::thread> th;
      for (int t = 0; t < threads; t++) th.emplace_back([&, t]() {
         part_t &p = part[t];
         for (int64_t i = share(n_nodes, t); i < share(n_nodes, t + 1); i++)
         {
            const smr_osm_node_t *n = (const smr_osm_node_t *) nodes[i];
            node_id[i] = n->obj.id; lat[i] = n->lat; lon[i] = n->lon; nvis[i] = n->obj.vis != 0;
            if (i > share(n_nodes, t) && node_id[i] <= node_id[i - 1]) p.ascending = false;
            take_tags(&n->obj, p.d, p.nk, p.nv, p.ntag_cnt);
         }
         for (int64_t i = share(n_ways, t); i < share(n_ways, t + 1); i++)
         {
            const smr_osm_way_t *w = (const smr_osm_way_t *) ways[i];
            const int rc = w->ref_cnt > 0 && w->ref ? w->ref_cnt : 0;
            way_id[i] = w->obj.id; wvis[i] = w->obj.vis != 0;
            wkind[i] = rc == 0 ? 0 : (w->ref[0] == w->ref[rc - 1] ? 2 : 1);      /* the test of apply_rule, src/smcore.c:264-279 */
            p.ref_cnt.push_back((uint32_t) rc);
            p.refs.insert(p.refs.end(), w->ref, w->ref + rc);
            take_tags(&w->obj, p.d, p.wk, p.wv, p.wtag_cnt);
         }
      });
      for (auto &t : th) t.join();
   }
   for (int64_t i = 0; i < n_nodes; i++)
      if (((const smr_osm_obj_t *) nodes[i])->type != SMR_OSM_NODE) return smr_fail(SMR_E_INVAL, "objects: nodes[%lld] is not a node", (long long) i);
   for (int64_t i = 0; i < n_ways; i++)
      if (((const smr_osm_obj_t *) ways[i])->type != SMR_OSM_WAY) return smr_fail(SMR_E_INVAL, "objects: ways[%lld] is not a way", (long long) i);

   /* merge the dictionaries (serial: one probe per distinct string of every thread) */
   dict_t &D = o->dict;
   for (part_t &p : part)
   {
      p.remap.resize(p.d.size());
      for (uint32_t id = 0; id < p.d.size(); id++) p.remap[id] = D.intern(p.d.str(id), p.d.len(id), p.d.hash[id], p.d.role[id]);
   }
   if (D.size() >= (size_t) INT32_MAX) return smr_fail(SMR_E_CAPACITY, "objects: %zu distinct strings", D.size());
   o->nodes_ascending = true;
   for (int t = 0; t < threads; t++)
   {
      if (!part[t].ascending) o->nodes_ascending = false;
      const int64_t b = share(n_nodes, t);
      if (t && b > 0 && b < n_nodes && share(n_nodes, t + 1) > b && node_id[b] <= node_id[b - 1]) o->nodes_ascending = false;
   }

   /* pass 2: offsets, then every thread writes its tags (global ids) and refs to their places */
   std::vector<int64_t> ntag_off(n_nodes + 1), wtag_off(n_ways + 1), way_off(n_ways + 1);
   std::vector<int64_t> nt0(threads + 1, 0), wt0(threads + 1, 0), rf0(threads + 1, 0);
   for (int t = 0; t < threads; t++)
   {
      nt0[t + 1] = nt0[t] + (int64_t) part[t].nk.size();
      wt0[t + 1] = wt0[t] + (int64_t) part[t].wk.size();
      rf0[t + 1] = rf0[t] + (int64_t) part[t].refs.size();
   }
   std::vector<int32_t> nk(nt0[threads]), nv(nt0[threads]), wk(wt0[threads]), wv(wt0[threads]);
   std::vector<int64_t> refs(rf0[threads]);
   {
      std::vector<std::thread> th;
      for (int t = 0; t < threads; t++) th.emplace_back([&, t]() {
         part_t &p = part[t];
         int64_t at = nt0[t], k = 0;
         for (int64_t i = share(n_nodes, t); i < share(n_nodes, t + 1); i++, k++) { ntag_off[i] = at; at += p.ntag_cnt[k]; }
         for (size_t j = 0; j < p.nk.size(); j++) { nk[nt0[t] + j] = (int32_t) p.remap[p.nk[j]]; nv[nt0[t] + j] = (int32_t) p.remap[p.nv[j]]; }
         at = wt0[t]; k = 0;
         int64_t rat = rf0[t];
         for (int64_t i = share(n_ways, t); i < share(n_ways, t + 1); i++, k++) { wtag_off[i] = at; at += p.wtag_cnt[k]; way_off[i] = rat; rat += p.ref_cnt[k]; }
         for (size_t j = 0; j < p.wk.size(); j++) { wk[wt0[t] + j] = (int32_t) p.remap[p.wk[j]]; wv[wt0[t] + j] = (int32_t) p.remap[p.wv[j]]; }
         if (!p.refs.empty()) memcpy(refs.data() + rf0[t], p.refs.data(), p.refs.size() * sizeof(int64_t));
      });
      for (auto &t : th) t.join();
   }
   ntag_off[n_nodes] = nt0[threads]; wtag_off[n_ways] = wt0[threads]; way_off[n_ways] = rf0[threads];
   part.clear();
   if (rf0[threads] >= (int64_t) UINT32_MAX) return smr_fail(SMR_E_CAPACITY, "objects: %lld refs", (long long) rf0[threads]);
   o->n_nodes = n_nodes; o->n_ways = n_ways; o->n_refs = rf0[threads]; o->n_ntags = nt0[threads]; o->n_wtags = wt0[threads];
   const double t_extracted = now_ms();

   hipStream_t s = smr_g.stream;
   SMR_TRY(upload(o->node_id, node_id)); SMR_TRY(upload(o->lat, lat)); SMR_TRY(upload(o->lon, lon)); SMR_TRY(upload(o->nvis, nvis));
   SMR_TRY(upload(o->ntag_off, ntag_off)); SMR_TRY(upload(o->nk, nk)); SMR_TRY(upload(o->nv, nv));
   SMR_TRY(upload(o->way_id, way_id)); SMR_TRY(upload(o->way_off, way_off)); SMR_TRY(upload(o->ref_id, refs));
   SMR_TRY(upload(o->wvis, wvis)); SMR_TRY(upload(o->wkind, wkind)); SMR_TRY(upload(o->wtag_off, wtag_off)); SMR_TRY(upload(o->wk, wk)); SMR_TRY(upload(o->wv, wv));
   SMR_TRY(upload(o->pool, D.pool)); SMR_TRY(upload(o->str_off, D.off));
   SMR_HIP(hipStreamSynchronize(s));          /* the vectors above go out of use at return */
   const double t_uploaded = now_ms();

   /* the number every string reads as, once */
   SMR_TRY(o->tod.need(D.size() * sizeof(double)));
   if (D.size()) k_dict_tod<<<blocks((int64_t) D.size()), 256, 0, s>>>(o->pool.as<uint8_t>(), o->str_off.as<uint64_t>(), (int64_t) D.size(), o->tod.as<double>());

   /* refs -> node indices */
   const int64_t *ids = o->node_id.as<int64_t>();
   const int32_t *sidx = nullptr;
   if (!o->nodes_ascending && n_nodes > 1)
   {
      SMR_TRY(o->sorted_id.need(n_nodes * sizeof(int64_t))); SMR_TRY(o->sorted_idx.need(n_nodes * sizeof(int32_t)));
      SMR_TRY(o->misc.need(n_nodes * sizeof(int32_t)));
      k_iota<<<blocks(n_nodes), 256, 0, s>>>(o->misc.as<int32_t>(), n_nodes);
      size_t tmp = 0;
      SMR_HIP(rocprim::radix_sort_pairs(nullptr, tmp, o->node_id.as<int64_t>(), o->sorted_id.as<int64_t>(), o->misc.as<int32_t>(), o->sorted_idx.as<int32_t>(),
                                        (size_t) n_nodes, 0, 64, s));
      SMR_TRY(o->sort_tmp.need(tmp));
      SMR_HIP(rocprim::radix_sort_pairs(o->sort_tmp.p, tmp, o->node_id.as<int64_t>(), o->sorted_id.as<int64_t>(), o->misc.as<int32_t>(), o->sorted_idx.as<int32_t>(),
                                        (size_t) n_nodes, 0, 64, s));
      ids = o->sorted_id.as<int64_t>(); sidx = o->sorted_idx.as<int32_t>();
   }
   SMR_TRY(o->ref.need(std::max<int64_t>(1, o->n_refs) * sizeof(int32_t)));
   SMR_TRY(o->counts.need(256));
   SMR_HIP(hipMemsetAsync(o->counts.p, 0, 8, s));
   if (o->n_refs)
      k_ref_index<<<blocks(o->n_refs), 256, 0, s>>>(o->ref_id.as<int64_t>(), o->n_refs, ids, sidx, n_nodes, o->ref.as<int32_t>(), o->counts.as<uint32_t>());
   SMR_HIP(hipGetLastError());
   uint32_t missing = 0;
   SMR_HIP(hipMemcpyAsync(&missing, o->counts.p, 4, hipMemcpyDeviceToHost, s));
   SMR_HIP(hipStreamSynchronize(s));
   o->missing = missing;
   o->sort_tmp.release(); o->misc.release();
   o->extract_ms = t_extracted - t_begin; o->upload_ms = t_uploaded - t_extracted; o->index_ms = now_ms() - t_uploaded;
   return SMR_OK;
}

extern "C" int smr_hip_objects_create(const void *const *nodes, int64_t n_nodes, const void *const *ways, int64_t n_ways, int threads, smr_objects_t **out)
{
   SMR_NEED_INIT();
   if (!out || n_nodes < 0 || n_ways < 0 || (n_nodes && !nodes) || (n_ways && !ways)) return smr_fail(SMR_E_INVAL, "objects_create: bad argument");
   smr_objects *o = new smr_objects();
   const int e = objects_build(o, nodes, n_nodes, ways, n_ways, threads);
   if (e != SMR_OK) { o->release(); delete o; return e; }
   *out = o;
   return SMR_OK;
}

extern "C" int smr_hip_objects_destroy(smr_objects_t *o)
{
   if (!o) return SMR_OK;
   if (smr_g.init) hipStreamSynchronize(smr_g.stream);
   o->release();
   delete o;
   return SMR_OK;
}

extern "C" int smr_hip_objects_info(const smr_objects_t *o, smr_objects_info_t *info)
{
   if (!o || !info) return smr_fail(SMR_E_INVAL, "objects_info: NULL argument");
   info->n_nodes = o->n_nodes; info->n_ways = o->n_ways; info->n_refs = o->n_refs; info->n_node_tags = o->n_ntags; info->n_way_tags = o->n_wtags;
   info->n_strings = (int64_t) o->dict.size(); info->pool_bytes = (int64_t) o->dict.pool.size(); info->missing_refs = o->missing;
   info->d_node_id = o->node_id.as<int64_t>(); info->d_lat = o->lat.as<double>(); info->d_lon = o->lon.as<double>();
   info->d_way_id = o->way_id.as<int64_t>(); info->d_way_off = o->way_off.as<int64_t>(); info->d_ref = o->ref.as<int32_t>();
   info->extract_ms = o->extract_ms; info->upload_ms = o->upload_ms; info->index_ms = o->index_ms;
   info->match_host_ms = o->match_host_ms; info->match_device_ms = o->match_device_ms;
   return SMR_OK;
}

/* one side of a match tag as parse_matchtype (src/smrparse.c:65-139) left it: pattern bytes in the rule's otag, kind in stag */
static int resolve_side(smr_objects *o, const smr_bstring_t &pat, const smr_special_tag &st, rside_t *out, std::vector<uint32_t> &rbits, int64_t rwords, int *n_regex)
{
   out->has = pat.len != 0; out->type = (uint16_t) st.type; out->pat = -1; out->pad_ = 0; out->val = 0;
   if (!out->has) return SMR_OK;
   if (pat.len < 0 || !pat.buf) return smr_fail(SMR_E_INVAL, "objects_match: a match tag of length %d", pat.len);
   switch (st.type & SP_MASK)
   {
      case SP_DIRECT:
      {
         const uint32_t id = o->dict.find(pat.buf, pat.len, hash_bytes(pat.buf, pat.len));
         out->pat = id == UINT32_MAX ? -1 : (int32_t) id;
         break;
      }
      case SP_REGEX:      /* a row of bits, one per string; the caller queues regexec for it */
         out->pat = (*n_regex)++;
         rbits.resize((size_t) (*n_regex) * rwords, 0);
         break;
      case SP_GT: case SP_LT: out->val = st.val; break;
      default: break;        /* bs_match leaves r = 1: never a match (always one with SPECIAL_INVERT) */
   }
   return SMR_OK;
}

extern "C" int smr_hip_objects_match(smr_objects_t *o, const void *const *rules, int n_rules, const uint8_t *d_node_on_page,
                                     uint64_t *d_bitmap, int64_t words_per_rule, int64_t *counts)
{
   SMR_NEED_INIT();
   if (!o || !rules || n_rules <= 0 || !d_bitmap || !counts) return smr_fail(SMR_E_INVAL, "objects_match: bad argument");
   const int64_t need_words = (std::max(o->n_nodes, o->n_ways) + 63) / 64;
   if (words_per_rule < need_words) return smr_fail(SMR_E_INVAL, "objects_match: %lld words per rule, %lld objects need %lld", (long long) words_per_rule,
                                                    (long long) std::max(o->n_nodes, o->n_ways), (long long) need_words);
   hipStream_t s = smr_g.stream;
   const double t_begin = now_ms();
   const int64_t n_str = (int64_t) o->dict.size(), rwords = (n_str + 31) / 32;
   /* rules in two groups, node rules and way rules, each keeping the caller's numbering through `row` and `slot` */
   std::vector<rdesc_t> desc[2];
   std::vector<int> slot[2];
   std::vector<rtag_t> rtags;
   std::vector<uint32_t> rbits;
   struct regex_job_t { const regex_t *re; int row; uint8_t role; std::string pat; };
   std::vector<regex_job_t> jobs;
   int n_regex = 0;
   for (int r = 0; r < n_rules; r++)
   {
      const smr_smrule_t *R = (const smr_smrule_t *) rules[r];
      counts[r] = 0;
      if (!R || !R->oo || !R->act) return smr_fail(SMR_E_INVAL, "objects_match: rule %d is incomplete", r);
      if (R->oo->type != SMR_OSM_NODE && R->oo->type != SMR_OSM_WAY) { counts[r] = -1; continue; }      /* relations stay with the host's executor */
      const int g = R->oo->type == SMR_OSM_WAY;
      const smr_stag *st = (const smr_stag *) ((const char *) R->act + sizeof(smr_action_t));
      rdesc_t d;
      d.tag0 = (int32_t) rtags.size(); d.ntags = R->oo->tag_cnt > 0 ? R->oo->tag_cnt : 0; d.empty_ok = 1;
      d.wayflag = (R->act->flags & SMR_ACTION_CLOSED_WAY) ? 2 : (R->act->flags & SMR_ACTION_OPEN_WAY) ? 1 : 0;
      d.row = (int64_t) r * words_per_rule;
      for (int t = 0; t < d.ntags; t++)
      {
         rtag_t T;
         const int before = n_regex;
         SMR_TRY(resolve_side(o, R->oo->otag[t].k, st[t].stk, &T.k, rbits, rwords, &n_regex));
         if (n_regex > before) jobs.push_back({ &st[t].stk.re, T.k.pat, 1, std::string(R->oo->otag[t].k.buf, R->oo->otag[t].k.len) });
         const int mid = n_regex;
         SMR_TRY(resolve_side(o, R->oo->otag[t].v, st[t].stv, &T.v, rbits, rwords, &n_regex));
         if (n_regex > mid) jobs.push_back({ &st[t].stv.re, T.v.pat, 2, std::string(R->oo->otag[t].v.buf, R->oo->otag[t].v.len) });
         rtags.push_back(T);
         if (!((T.k.type | T.v.type) & SP_NOT)) d.empty_ok = 0;
      }
      desc[g].push_back(d); slot[g].push_back(r);
   }
   /* the regular expressions: every distinct string seen in the role the expression is asked about (glibc's regexec takes a
    * lock per compiled expression, so the jobs, not the strings, are what threads share) */
   if (!jobs.empty())
   {
      /* glibc's regexec takes a lock per compiled expression, so threads that share the strings each need expressions of their
       * own: compiled from the rule's pattern text with the flags parse_matchtype uses (thread 0, and any thread whose copy
       * does not compile, runs the rule's own regex_t).  A thread's strings start at a multiple of 32: no two share a word */
      const int nth = (int64_t) jobs.size() * n_str > 100000 ? 16 : 1;
      std::vector<std::thread> th;
      for (int t = 0; t < nth; t++) th.emplace_back([&, t]() {
         const int64_t lo = rwords * t / nth * 32, hi = std::min<int64_t>(n_str, rwords * (t + 1) / nth * 32);
         for (size_t j = 0; j < jobs.size(); j++)
         {
            regex_t own;
            const bool have = t > 0 && regcomp(&own, jobs[j].pat.c_str(), REG_EXTENDED | REG_NOSUB) == 0;
            const regex_t *re = have ? &own : jobs[j].re;
            uint32_t *bits = rbits.data() + (size_t) jobs[j].row * rwords;
            for (int64_t id = lo; id < hi; id++)
               if ((o->dict.role[id] & jobs[j].role) && regexec(re, o->dict.str((uint32_t) id), 0, nullptr, 0) == 0) bits[id >> 5] |= 1u << (id & 31);
            if (have) regfree(&own);
         }
      });
      for (auto &t : th) t.join();
   }
   const double t_host = now_ms();
   std::vector<rdesc_t> all(desc[0]);
   all.insert(all.end(), desc[1].begin(), desc[1].end());
   SMR_TRY(upload(o->rules, all)); SMR_TRY(upload(o->rtags, rtags)); SMR_TRY(upload(o->rbits, rbits));
   SMR_TRY(o->counts.need(std::max<size_t>(1, all.size()) * 8));
   SMR_HIP(hipMemsetAsync(o->counts.p, 0, std::max<size_t>(1, all.size()) * 8, s));
   SMR_HIP(hipMemsetAsync(d_bitmap, 0, (size_t) n_rules * words_per_rule * 8, s));
   for (int g = 0; g < 2; g++)
   {
      if (desc[g].empty()) continue;
      match_args_t a;
      a.is_way = g; a.n_obj = g ? o->n_ways : o->n_nodes;
      if (a.n_obj == 0) continue;
      a.tag_off = g ? o->wtag_off.as<int64_t>() : o->ntag_off.as<int64_t>();
      a.kid = g ? o->wk.as<int32_t>() : o->nk.as<int32_t>(); a.vid = g ? o->wv.as<int32_t>() : o->nv.as<int32_t>();
      a.vis = g ? o->wvis.as<uint8_t>() : o->nvis.as<uint8_t>();
      a.extra = g ? o->wkind.as<uint8_t>() : d_node_on_page;
      const size_t first = g ? desc[0].size() : 0;
      a.rules = o->rules.as<rdesc_t>() + first; a.rtags = o->rtags.as<rtag_t>(); a.n_rules = (int) desc[g].size();
      a.tod = o->tod.as<double>(); a.rbits = o->rbits.as<uint32_t>(); a.rwords = rwords;
      a.bitmap = d_bitmap; a.counts = o->counts.as<unsigned long long>() + first;
      dim3 grid(blocks(a.n_obj, 256 * MATCH_OBJ_ITERS), (unsigned) ((desc[g].size() + MATCH_RULES_PER_BLOCK - 1) / MATCH_RULES_PER_BLOCK));
      k_match<<<grid, 256, 0, s>>>(a);
      const int64_t n_words = (a.n_obj + 63) / 64;
      k_row_count<<<dim3(blocks(n_words, 4096), (unsigned) desc[g].size()), 256, 0, s>>>(d_bitmap, a.rules, n_words, a.counts);
   }
   SMR_HIP(hipGetLastError());
   std::vector<unsigned long long> got(std::max<size_t>(1, all.size()));
   SMR_HIP(hipMemcpyAsync(got.data(), o->counts.p, all.size() * 8, hipMemcpyDeviceToHost, s));
   SMR_HIP(hipStreamSynchronize(s));
   o->match_host_ms = t_host - t_begin; o->match_device_ms = now_ms() - t_host;
   for (size_t i = 0; i < slot[0].size(); i++) counts[slot[0][i]] = (int64_t) got[i];
   for (size_t i = 0; i < slot[1].size(); i++) counts[slot[1][i]] = (int64_t) got[slot[0].size() + i];
   return SMR_OK;
}

static int select_rank(smr_objects *o, const uint64_t *d_row, int64_t n_obj, bool ways, uint32_t *n_sel)
{
   hipStream_t s = smr_g.stream;
   const int64_t n_words = (n_obj + 63) / 64;
   *n_sel = 0;
   if (n_obj == 0) return SMR_OK;
   SMR_TRY(o->pop.need((n_words + 1) * 4)); SMR_TRY(o->sums.need((n_words / 1024 + 2) * 4 + 16));
   k_word_pop<<<blocks(n_words), 256, 0, s>>>(d_row, n_words, o->pop.as<uint32_t>());
   uint32_t *sums = o->sums.as<uint32_t>();
   SMR_TRY(smr_scan_u32(o->pop.as<uint32_t>(), (uint32_t) n_words, sums + 4, sums, n_sel));
   SMR_TRY(o->sel_index.need(std::max<size_t>(1, *n_sel) * 4));
   if (ways) SMR_TRY(o->sel_cnt.need(((size_t) *n_sel + 1) * 4));
   if (*n_sel)
      k_sel_rank<<<blocks(n_obj), 256, 0, s>>>(d_row, o->pop.as<uint32_t>(), n_obj, o->way_off.as<int64_t>(), o->ref.as<int32_t>(), o->sel_index.as<int32_t>(),
                                              ways ? o->sel_cnt.as<uint32_t>() : nullptr);
   SMR_HIP(hipGetLastError());
   return SMR_OK;
}

extern "C" int smr_hip_objects_select_ways(smr_objects_t *o, const uint64_t *d_bitmap_row, smr_way_selection_t *sel)
{
   SMR_NEED_INIT();
   if (!o || !d_bitmap_row || !sel) return smr_fail(SMR_E_INVAL, "objects_select_ways: NULL argument");
   hipStream_t s = smr_g.stream;
   uint32_t n_sel = 0, n_refs = 0;
   SMR_TRY(select_rank(o, d_bitmap_row, o->n_ways, true, &n_sel));
   SMR_TRY(o->sel_off.need(((size_t) n_sel + 1) * 8));
   if (n_sel)
   {
      uint32_t *sums = o->sums.as<uint32_t>();
      SMR_TRY(o->sums.need((n_sel / 1024 + 2) * 4 + 16));
      sums = o->sums.as<uint32_t>();
      SMR_TRY(smr_scan_u32(o->sel_cnt.as<uint32_t>(), n_sel, sums + 4, sums, &n_refs));
      SMR_TRY(o->sel_ref.need(std::max<size_t>(1, n_refs) * 4));
      k_sel_copy<<<blocks(((int64_t) n_sel + 1) * 64), 256, 0, s>>>(o->sel_index.as<int32_t>(), o->sel_cnt.as<uint32_t>(), n_sel, o->way_off.as<int64_t>(),
                                                                   o->ref.as<int32_t>(), o->sel_off.as<int64_t>(), o->sel_ref.as<int32_t>());
      SMR_HIP(hipGetLastError());
   }
   else SMR_HIP(hipMemsetAsync(o->sel_off.p, 0, 8, s));
   sel->n_ways = n_sel; sel->n_refs = n_refs;
   sel->d_way_off = o->sel_off.as<int64_t>(); sel->d_ref = o->sel_ref.as<int32_t>(); sel->d_way_index = o->sel_index.as<int32_t>();
   return SMR_OK;
}

extern "C" int smr_hip_objects_select_nodes(smr_objects_t *o, const uint64_t *d_bitmap_row, const int32_t **d_index, int64_t *n)
{
   SMR_NEED_INIT();
   if (!o || !d_bitmap_row || !d_index || !n) return smr_fail(SMR_E_INVAL, "objects_select_nodes: NULL argument");
   uint32_t n_sel = 0;
   SMR_TRY(select_rank(o, d_bitmap_row, o->n_nodes, false, &n_sel));
   *d_index = o->sel_index.as<int32_t>(); *n = n_sel;
   return SMR_OK;
}
