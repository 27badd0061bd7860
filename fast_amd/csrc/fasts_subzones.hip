// Host logic of the 'implicit_local' sub-zones; see fasts_subzones.h.
#include "fasts_subzones.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

// the schedule of Fast/FastC/FastC/Compute/skip_lu.for for nitmax = 3 .. 41, one row after the other: entry q of a row =
// sub-iterations a block may skip before sub-iteration q + 1 takes it up again
static const unsigned char SKIP_TAB[] = {
  0, 0, 0,
  0, 0, 1, 0,
  0, 0, 1, 2, 0,
  0, 0, 3, 1, 2, 0,
  0, 0, 3, 4, 1, 2, 0,
  0, 0, 5, 3, 4, 1, 2, 0,
  0, 0, 5, 3, 4, 6, 1, 2, 0,
  0, 0, 7, 3, 5, 6, 1, 2, 4, 0,
  0, 0, 8, 5, 3, 4, 6, 1, 7, 2, 0,
  0, 0, 9, 5, 8, 3, 4, 6, 1, 2, 7, 0,
  0, 0, 9, 5, 3, 8, 4, 10, 1, 6, 2, 7, 0,
  0, 0, 11, 5, 9, 3, 8, 4, 6, 1, 10, 2, 7, 0,
  0, 0, 11, 5, 9, 3, 8, 4, 12, 6, 1, 10, 2, 7, 0,
  0, 0, 11, 5, 9, 3, 8, 13, 4, 12, 6, 1, 10, 2, 7, 0,
  0, 0, 14, 8, 5, 13, 3, 11, 9, 4, 12, 6, 1, 10, 2, 7, 0,
  0, 0, 15, 10, 3, 13, 5, 9, 12, 6, 4, 8, 1, 11, 2, 14, 7, 0,
  0, 0, 14, 5, 9, 13, 3, 11, 8, 16, 4, 12, 6, 1, 10, 2, 15, 7, 0,
  0, 0, 14, 5, 9, 16, 3, 11, 8, 17, 4, 12, 6, 1, 10, 15, 2, 7, 13, 0,
  0, 0, 14, 5, 9, 16, 3, 11, 12, 8, 18, 4, 12, 6, 1, 10, 17, 2, 7, 15, 0,
  0, 0, 18, 5, 9, 16, 3, 14, 8, 13, 4, 19, 6, 12, 1, 10, 17, 2, 7, 15, 11, 0,
  0, 0, 18, 5, 9, 16, 3, 11, 14, 8, 13, 4, 20, 12, 6, 1, 19, 10, 17, 2, 7, 15, 0,
  0, 0, 20, 5, 14, 9, 16, 3, 11, 18, 8, 13, 4, 21, 12, 19, 1, 10, 6, 17, 2, 7, 15, 0,
  0, 0, 20, 5, 14, 9, 16, 11, 18, 8, 13, 22, 4, 12, 6, 19, 1, 10, 17, 2, 20, 7, 14, 3, 0,
  0, 0, 21, 14, 5, 9, 16, 3, 11, 18, 8, 13, 23, 12, 20, 6, 19, 1, 10, 17, 2, 22, 4, 7, 15, 0,
  0, 0, 23, 14, 5, 9, 16, 3, 11, 18, 8, 13, 24, 4, 20, 22, 6, 19, 1, 10, 17, 2, 12, 21, 7, 15, 0,
  0, 0, 24, 14, 5, 21, 9, 16, 11, 18, 20, 8, 4, 25, 12, 23, 6, 19, 1, 10, 17, 2, 22, 3, 13, 7, 15, 0,
  0, 0, 25, 14, 5, 21, 9, 16, 3, 11, 18, 8, 13, 26, 20, 12, 23, 6, 19, 1, 4, 24, 10, 17, 2, 22, 7, 15, 0,
  0, 0, 25, 14, 5, 21, 9, 16, 3, 11, 18, 24, 8, 13, 27, 4, 20, 12, 6, 19, 1, 17, 23, 10, 26, 2, 22, 7, 15, 0,
  0, 0, 26, 15, 6, 22, 9, 16, 3, 12, 20, 25, 7, 12, 28, 5, 21, 12, 7, 19, 3, 21, 25, 11, 27, 3, 23, 6, 17, 0, 0,
  0, 0, 25, 16, 7, 22, 8, 17, 4, 13, 18, 28, 7, 13, 29, 5, 21, 12, 7, 20, 3, 19, 25, 11, 28, 3, 24, 6, 16, 20, 22, 0,
  0, 0, 26, 16, 7, 22, 8, 17, 4, 14, 18, 29, 7, 13, 30, 6, 21, 12, 7, 20, 4, 20, 25, 11, 28, 4, 24, 6, 16, 20, 8, 23, 0,
  0, 0, 26, 16, 8, 22, 8, 17, 4, 14, 20, 30, 7, 13, 31, 6, 22, 12, 7, 20, 5, 20, 25, 11, 28, 4, 24, 7, 16, 20, 8, 15, 23, 0,
  0, 0, 26, 16, 8, 22, 9, 20, 5, 14, 20, 30, 7, 14, 32, 6, 22, 12, 8, 20, 5, 18, 25, 11, 28, 4, 24, 7, 16, 20, 8, 15, 30, 22, 0,
  0, 0, 27, 16, 8, 22, 9, 20, 5, 14, 20, 31, 7, 14, 33, 6, 22, 12, 8, 20, 5, 18, 25, 11, 28, 4, 24, 7, 16, 20, 8, 15, 30, 22, 12, 0,
  0, 0, 27, 16, 8, 22, 9, 20, 5, 14, 20, 31, 7, 14, 33, 6, 22, 12, 8, 20, 5, 18, 25, 11, 28, 4, 24, 7, 16, 34, 8, 15, 30, 22, 12, 20, 0,
  0, 0, 27, 16, 8, 22, 9, 20, 5, 14, 20, 31, 7, 14, 33, 6, 22, 12, 8, 20, 5, 18, 25, 11, 28, 4, 24, 7, 16, 34, 8, 15, 30, 22, 12, 20, 10, 0,
  0, 0, 27, 16, 8, 22, 9, 20, 5, 14, 20, 31, 7, 14, 33, 6, 22, 12, 8, 20, 5, 18, 25, 11, 28, 4, 24, 7, 16, 34, 8, 15, 30, 22, 12, 20, 7, 12, 0,
  0, 0, 27, 16, 8, 22, 9, 20, 5, 14, 20, 31, 7, 14, 33, 6, 22, 12, 8, 20, 5, 18, 25, 11, 28, 4, 24, 7, 16, 38, 8, 15, 35, 22, 12, 20, 7, 12, 28, 0,
  0, 0, 27, 16, 8, 22, 9, 20, 5, 14, 20, 31, 7, 14, 33, 6, 22, 12, 8, 20, 5, 18, 25, 11, 28, 4, 24, 7, 16, 34, 8, 15, 30, 22, 12, 20, 12, 8, 20, 30, 0,
};
int skip_row(int nitmax)   // first entry of the row of nitmax: rows have 3, 4, 5 ... entries
{
  return (nitmax * (nitmax - 1)) / 2 - 3;
}

}  // namespace

void fsz_skip_lu(int nitmax, int mx_ssiter, int32_t *iskip_lu)
{
  if (nitmax < 3 || nitmax > 41) {
    for (int q = 0; q < 2 * mx_ssiter; q++) iskip_lu[q] = 0;
    return;
  }
  int32_t *col = iskip_lu + (nitmax == mx_ssiter ? 0 : mx_ssiter);
  const unsigned char *row = SKIP_TAB + skip_row(nitmax);
  for (int q = 0; q < nitmax && q < mx_ssiter; q++) col[q] = row[q];
}

// Blocks are numbered b = i + ni * (j + nj * k) from 0 here, from 1 in no_lu (the reference's numbering).
int fsz_boxes(const int nblk[3], const int size[3], const int ijkv[3], int iseuil, const int32_t *it_target,
              int32_t *no_lu, int32_t *count, std::vector<int32_t> &ind_dm)
{
  const int ni = nblk[0], nj = nblk[1], nk = nblk[2], n = ni * nj * nk;
  const int inc[3] = {1, ni, ni * nj};
  std::vector<char> todo((size_t)n);
  std::vector<int> seeds;
  for (int b = 0; b < n; b++) {
    todo[b] = it_target[b] >= iseuil;
    if (todo[b]) seeds.push_back(b);
  }
  struct Box { int lo[3], hi[3]; };
  std::vector<Box> bx;
  auto any_todo = [&](const int lo[3], const int hi[3]) {
    for (int k = lo[2]; k <= hi[2]; k++)
      for (int j = lo[1]; j <= hi[1]; j++)
        for (int i = lo[0]; i <= hi[0]; i++)
          if (todo[i + ni * (j + nj * k)]) return true;
    return false;
  };
  for (int b : seeds) {
    if (!todo[b]) continue;   // swallowed by the box of an earlier seed
    const int c[3] = {b % ni, (b / ni) % nj, b / (ni * nj)};
    Box x;
    // the run of blocks to do through the seed along each axis
    for (int d = 0; d < 3; d++) {
      int lo = c[d], hi = c[d];
      while (lo > 0 && todo[b - (c[d] - lo + 1) * inc[d]]) lo--;
      while (hi < nblk[d] - 1 && todo[b + (hi - c[d] + 1) * inc[d]]) hi++;
      x.lo[d] = lo; x.hi[d] = hi;
    }
    // grow while a block to do touches the box (its faces, edges and corners), direction after direction, until a
    // whole round adds nothing
    bool again = true;
    while (again) {
      again = false;
      for (int d = 0; d < 3; d++)
        for (int side = 0; side < 2; side++) {
          for (;;) {
            const int at = side ? x.hi[d] + 1 : x.lo[d] - 1;
            if (at < 0 || at >= nblk[d]) break;
            int lo[3], hi[3];
            for (int e = 0; e < 3; e++) {
              lo[e] = std::max(0, x.lo[e] - 1);
              hi[e] = std::min(nblk[e] - 1, x.hi[e] + 1);
            }
            lo[d] = hi[d] = at;
            if (!any_todo(lo, hi)) break;
            if (side) x.hi[d] = at; else x.lo[d] = at;
            again = true;
          }
        }
    }
    for (int k = x.lo[2]; k <= x.hi[2]; k++)
      for (int j = x.lo[1]; j <= x.hi[1]; j++)
        for (int i = x.lo[0]; i <= x.hi[0]; i++) todo[i + ni * (j + nj * k)] = 0;
    bx.push_back(x);
  }
  // Boxes that touch or overlap become one.  The scan works in place on the list as the reference does: the absorbed
  // box leaves, the later ones move down one place and the scan goes on with the same two positions -- so when the
  // absorbed box stood before the absorbing one, the latter is compared from then on through its old place, which still
  // holds its values (and, past the end of the list, can absorb its own new copy: the reference then reports
  // 'WARNING MERGE' and loses the box; reproduced as is).
  {
    std::vector<Box> slot(bx);          // places 0 .. c1-1 are the list; places beyond keep what they last held
    int c1 = (int)slot.size();
    bool merged = true;
    while (merged && c1 != 1) {
      merged = false;
      int a = 0;
      while (a < c1 && c1 != 1) {
        int o = 0;
        while (o < c1) {
          bool touch = o != a && c1 != 1;
          for (int d = 0; touch && d < 3; d++)
            touch = slot[a].lo[d] - 1 <= slot[o].hi[d] && slot[a].hi[d] + 1 >= slot[o].lo[d];
          if (touch) {
            merged = true;
            for (int d = 0; d < 3; d++) {
              slot[a].lo[d] = std::min(slot[a].lo[d], slot[o].lo[d]);
              slot[a].hi[d] = std::max(slot[a].hi[d], slot[o].hi[d]);
            }
            for (int q = o; q < c1 - 1; q++) slot[q] = slot[q + 1];
            c1--;
          } else
            o++;
        }
        a++;
      }
    }
    bx.assign(slot.begin(), slot.begin() + c1);
  }
  int cnt = 0;
  for (const Box &x : bx) {
    for (int k = x.lo[2]; k <= x.hi[2]; k++)
      for (int j = x.lo[1]; j <= x.hi[1]; j++)
        for (int i = x.lo[0]; i <= x.hi[0]; i++) no_lu[cnt++] = 1 + i + ni * (j + nj * k);
    for (int d = 0; d < 3; d++) {
      ind_dm.push_back(x.lo[d] * size[d] + 1);
      ind_dm.push_back(x.hi[d] == nblk[d] - 1 ? ijkv[d] : (x.hi[d] + 1) * size[d]);
    }
  }
  *count = cnt;
  return (int)bx.size();
}

void fsz_init(FsSubZones &s, const int ijkv[3], const int size_ssdom[3], int nssiter, int neq, bool threed)
{
  for (int d = 0; d < 3; d++) {
    s.ijkv[d] = ijkv[d];
    s.size[d] = size_ssdom[d];
    s.nblk[d] = std::max(1, ijkv[d] / size_ssdom[d]);
  }
  s.n = s.nblk[0] * s.nblk[1] * s.nblk[2];
  s.nssiter = nssiter; s.neq = neq; s.threed = threed;
  // Fast/FastS/FastS/Metric/allocate_metric.cpp:441-471: counters and targets start at the number of sub-iterations
  s.it_lu.assign((size_t)s.n, nssiter);
  s.it_target.assign((size_t)s.n, nssiter);
  s.it_target_old.assign((size_t)s.n, 0);
  s.it_temp.assign((size_t)s.n, 0);
  s.no_lu.resize((size_t)s.n);
  for (int b = 0; b < s.n; b++) s.no_lu[b] = b + 1;
  s.nisdom_residu.assign((size_t)nssiter, 0);
  s.boxes.assign((size_t)nssiter, std::vector<int32_t>());
  s.it_bloc = 0;
  s.rdm.assign((size_t)nssiter * neq * 2 * s.n, 0.);
}

void fsz_reset(FsSubZones &s)
{
  s.it_bloc = 0;
  std::fill(s.it_lu.begin(), s.it_lu.end(), 0);
}

void fsz_evaluate(FsSubZones &s, int nstep, const int32_t *iskip_lu)
{
  const int nitmin = 3;   // fewest sub-iterations of a block
  const int iseuil = (nstep != s.nssiter ? iskip_lu[s.nssiter + nstep - 1] : iskip_lu[nstep - 1]) + nitmin;
  std::vector<int32_t> &b = s.boxes[(size_t)nstep - 1];
  b.clear();
  int32_t count = 0;
  fsz_boxes(s.nblk, s.size, s.ijkv, iseuil, s.it_target.data(), s.no_lu.data(), &count, b);
  s.nisdom_residu[(size_t)nstep - 1] = count;
}

void fsz_residual(FsSubZones &s, int nstep, double epsi, const double *norms, const int32_t *no_lu, int nis)
{
  const double cut0x = 1e-20;
  const double conv_loo = std::log10(epsi);
  const int nss = s.nssiter, neq = s.neq, it = s.it_bloc;
  auto rdm = [&](int itb, int ne, int c, int b) -> double & {   // rdm_sdm(itb, ne + 1, c + 1, b + 1)
    return s.rdm[(size_t)(itb - 1) + (size_t)nss * ((size_t)ne + (size_t)neq * ((size_t)c + 2 * (size_t)b))];
  };
  for (int b = 0; b < s.n; b++) {
    // the block counts as computed when it stands among the first `nis` entries of the list (the first entry is
    // looked at even when the list is empty)
    int q = 0;
    while (no_lu[q] != b + 1 && q + 1 < nis) q++;
    const bool computed = no_lu[q] == b + 1;
    if (computed) {
      s.it_lu[b] += 1;
      for (int ne = 0; ne < neq; ne++) {
        rdm(it, ne, 0, b) = norms[12 * (size_t)b + ne];
        rdm(it, ne, 1, b) = norms[12 * (size_t)b + 6 + ne];
      }
      double x[5];
      for (int ne = 0; ne < 5; ne++)
        x[ne] = std::log10(std::max(rdm(it, ne, 1, b), cut0x)) - std::log10(std::max(rdm(1, ne, 1, b), cut0x));
      // mean drop of the largest magnitudes, in decades (two-dimensional zones leave the third momentum out)
      const double rmax = s.threed ? (x[1] + x[2] + x[3] + x[0] + x[4]) * 0.2 : (x[1] + x[2] + x[0] + x[4]) * 0.25;
      if (rmax > conv_loo) {
        s.it_temp[b] = s.it_lu[b];
        // converged at the very last sub-iteration allowed: one more next time
        if (s.it_temp[b] == s.it_target[b] && nss == nstep) s.it_temp[b] += 1;
      }
    } else {
      for (int ne = 0; ne < neq; ne++) {
        rdm(it, ne, 0, b) = rdm(it - 1, ne, 0, b);
        rdm(it, ne, 1, b) = rdm(it - 1, ne, 1, b);
      }
    }
    if (nstep == nss) {
      s.it_target_old[b] = s.it_target[b];
      s.it_target[b] = std::max(3, s.it_temp[b] + 1);
    }
  }
}
