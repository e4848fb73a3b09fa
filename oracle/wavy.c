/* oracle/wavy.c -- TEST INFRASTRUCTURE ONLY (see oracle/Makefile): plain-C restatement of the path that the reference records
 * for a way of a `wavy=1` draw rule.
 *
 *   orc_coord_diffp   src/smath.c:34-52   (bearing and orthodrome distance, both in degrees)
 *   orc_wavy_path     src/smrules_cairo.c:777-840 (cairo_smr_wavy: the way is resampled every `dist` degrees) and
 *                     :752-774 (wavy: one curve_to per step, control points at -45 and +45 degrees of the step's bearing,
 *                     dist / sqrt(2) from its start)
 *
 * Output: lat/lon of p0 and of (c1, c2, p) per curve_to, in path order -- what geo2pt then projects (:758, :765, :770, :799).
 * Pinned: orc_coord_diffp against the reference's own src/smath.c (oracle/_ref/libref_smath.so,
 * tests/test_oracle_golden.py::test_coord_diffp_equals_the_reference); the loop has no reference-held vector (parity unpinned
 * beyond that: the reference's function is static and draws into a cairo context). */
#include <math.h>

#define DEG2RAD(x) ((x) * M_PI / 180.0)      /* libsmrender/smrender.h:48-49 */
#define RAD2DEG(x) ((x) * 180.0 / M_PI)

static double fmod2_(double a, double n) { a = fmod(a, n); return a < 0 ? a + n : a; }      /* src/smath.c:78-82 */

void orc_coord_diffp(double slat, double slon, double dlat_, double dlon_, double *bearing, double *dist)
{
   double dlat = dlat_ - slat;
   double dlon = (dlon_ - slon) * cos(DEG2RAD((slat + dlat_) / 2.0));
   *bearing = RAD2DEG(atan2(dlon, dlat));
   *dist = RAD2DEG(acos(sin(DEG2RAD(slat)) * sin(DEG2RAD(dlat_)) + cos(DEG2RAD(slat)) * cos(DEG2RAD(dlat_)) * cos(DEG2RAD(dlon_ - slon))));
   *bearing = fmod2_(*bearing, 360.0);
}

/* lat[n], lon[n]: the way's nodes.  out_lat/out_lon[cap]: p0, then c1, c2, p per wave.  Returns the number of points written
 * (1 + 3 * waves), or -1 when cap is too small. */
int orc_wavy_path(const double *lat, const double *lon, int n, double dist, double *out_lat, double *out_lon, int cap)
{
   if (n < 1 || cap < 1) return -1;
   int m = 0;
   double sc_lat = lat[0], sc_lon = lon[0];
   out_lat[m] = sc_lat; out_lon[m] = sc_lon; m++;                       /* cairo_move_to :799-800 */
   double pc_bearing = 0, pc_dist = 0, d;
   for (int i = 1;;)
   {
      if (pc_dist <= 0)
      {
         if (i >= n) break;
         double dc_lat = lat[i], dc_lon = lon[i];
         i++;
         d = pc_dist;
         orc_coord_diffp(sc_lat, sc_lon, dc_lat, dc_lon, &pc_bearing, &pc_dist);
         pc_dist += d;
      }
      if (pc_dist > dist)
      {
         double ic_lat = sc_lat + dist * cos(DEG2RAD(pc_bearing));
         double ic_lon = sc_lon + dist * sin(DEG2RAD(pc_bearing)) / cos(DEG2RAD((ic_lat + sc_lat) / 2));
         /* wavy(&sc, &ic) */
         double wb, wd, la, lo;
         orc_coord_diffp(sc_lat, sc_lon, ic_lat, ic_lon, &wb, &wd);
         if (m + 3 > cap) return -1;
         wb -= 45.0;
         la = sc_lat + wd * M_SQRT1_2 * cos(DEG2RAD(wb));
         lo = sc_lon + wd * M_SQRT1_2 * sin(DEG2RAD(wb)) / cos(DEG2RAD((la + sc_lat) / 2));
         out_lat[m] = la; out_lon[m] = lo; m++;
         wb += 90.0;
         la = sc_lat + wd * M_SQRT1_2 * cos(DEG2RAD(wb));
         lo = sc_lon + wd * M_SQRT1_2 * sin(DEG2RAD(wb)) / cos(DEG2RAD((la + sc_lat) / 2));
         out_lat[m] = la; out_lon[m] = lo; m++;
         out_lat[m] = ic_lat; out_lon[m] = ic_lon; m++;
         sc_lat = ic_lat; sc_lon = ic_lon;
      }
      pc_dist -= dist;
   }
   return m;
}
