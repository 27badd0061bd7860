// Sub-zones of the 'implicit_local' scheme (host logic only): which blocks of SIZE_SSDOM cells of a zone still iterate
// at a Newton sub-iteration, the boxes they are merged into, and the per-block convergence bookkeeping that feeds the
// next evaluation.  Restates, on plain arrays:
//   schedule          Fast/FastC/FastC/Compute/skip_lu.for
//   boxes             Fast/FastC/FastC/Compute/init_ssiter_bloc.for:45-80,141, ssdom_lu_ijk.for
//   bookkeeping       Fast/FastS/FastS/Compute/cprdu3s1.for, cprdu3s1_incl.for
#pragma once
#include <cstdint>
#include <vector>

struct FsSubZones {
  int nblk[3] = {1, 1, 1};     // blocks per direction (ijkv_lu)
  int size[3] = {1, 1, 1};     // cells per block (SIZE_SSDOM); the last block of a direction takes the remainder
  int ijkv[3] = {1, 1, 1};
  int n = 1, nssiter = 1, neq = 5;
  bool threed = true;
  std::vector<int32_t> it_lu, it_target, it_target_old, it_temp, no_lu;   // n each
  std::vector<int32_t> nisdom_residu;                                     // per sub-iteration
  std::vector<std::vector<int32_t>> boxes;                                // per sub-iteration: 6 integers per box
  int it_bloc = 0;
  std::vector<double> rdm;   // rdm_sdm(nssiter, neq, 2, n), column-major: sums of squares and largest magnitudes
};

// iskip_lu(mx_ssiter, 2): column 1 is written when nitmax == mx_ssiter, else column 2; nitmax outside 3..41 zeroes both
void fsz_skip_lu(int nitmax, int mx_ssiter, int32_t *iskip_lu);
// ssdom_lu_ijk: boxes (in cells, appended to `ind_dm`, 6 integers each) of the blocks with it_target >= iseuil;
// no_lu receives the blocks of the boxes, *count their number.  Returns the number of boxes.
int fsz_boxes(const int nblk[3], const int size[3], const int ijkv[3], int iseuil, const int32_t *it_target,
              int32_t *no_lu, int32_t *count, std::vector<int32_t> &ind_dm);
void fsz_init(FsSubZones &s, const int ijkv[3], const int size_ssdom[3], int nssiter, int neq, bool threed);
// first sub-iteration of a verification step: counters back to zero (init_ssiter_bloc.for:45-54)
void fsz_reset(FsSubZones &s);
// boxes of sub-iteration nstep from the current targets (init_ssiter_bloc.for:58-80,141)
void fsz_evaluate(FsSubZones &s, int nstep, const int32_t *iskip_lu);
// one call of cprdu3s1 at sub-iteration nstep: norms = 12 doubles per block (sums of squares of 6 equations, then their
// largest magnitudes) of the Newton residual at this sub-iteration; no_lu / nis = the block list and its length as
// they stood when the sub-iteration ran.  s.it_bloc is the slot (1-based) the norms go to.
void fsz_residual(FsSubZones &s, int nstep, double epsi, const double *norms, const int32_t *no_lu, int nis);
