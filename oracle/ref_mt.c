/* TEST INFRASTRUCTURE ONLY (CPU timing baseline).  Runs one RHS + update sub-iteration of the compiled reference
 * (navier_stokes_struct_ of oracle/_ref/libfast_ref.so, handed over as a function pointer) on N OpenMP threads the
 * way the reference's own parallel region does: Fast/FastS/FastS/Compute/gsdr3.cpp:186-215 (one team, per-thread
 * topology / sub-domain / ijkv_sdm slots) and Fast/FastS/FastS/Compute/rhs.cpp:40-110 (sub-zone dealt to the
 * threads, one shared lock array `lok`, per-thread timer and CFL slots).  The thread sub-domains come from the
 * reference's own indice_boucle_lu_ (Fast/FastC/FastC/HPC_LAYER/indice_boucle_lu.for), the neighbour
 * synchronisation is the reference's lock protocol inside navier_stokes_struct_ (synchro_omp_scater.for). */
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef int32_t i32;
typedef void (*ibl_fn)(i32 *, i32 *, i32 *, i32 *, i32 *, i32 *, i32 *);
typedef void (*ns_fn)(i32 *, i32 *, i32 *, i32 *, i32 *, i32 *, i32 *, i32 *, i32 *, i32 *, i32 *, i32 *, i32 *, i32 *,
                      i32 *, i32 *, i32 *, i32 *, i32 *, double *, double *, i32 *, i32 *, i32 *, i32 *, i32 *, i32 *, i32 *,
                      double *, double *, double *, double *, double *, double *, double *, double *, double *, double *,
                      double *, double *, double *, double *, double *, double *, double *, double *, double *, double *,
                      double *, double *, double *, double *, double *, double *, double *, double *, double *, double *,
                      double *, double *, double *);

/* work: 64 * nthreads int32 (per-thread topology and ranges); lok: mx_synchro * nthreads int32, zeroed here;
 * cfl: 3 * nthreads; timer: 2 * nthreads */
int fo_ref_mt_subiter(void *ns, void *ibl, int nthreads, int nd, int nstep, int nssiter, int nitrun, int first_it,
                      i32 *pint, double *preal, double temps, i32 *ind_dm_zone, i32 *lok, int mx_synchro, i32 *work,
                      double *timer, double *cfl, double *alias, double *x, double *y, double *z, double *ro,
                      double *ro_m1, double *ro_p1, double *ro_ssiter, double *xmut, double *ti, double *tj, double *tk,
                      double *vol, double *wig, double *drodm, double *coe)
{
  ns_fn navier_stokes_struct = (ns_fn)ns;
  ibl_fn indice_boucle_lu = (ibl_fn)ibl;
  const int explicit_ = pint[29] == 2;
  memset(lok, 0, sizeof(i32) * (size_t)mx_synchro * nthreads);
  int used = 0;
#pragma omp parallel num_threads(nthreads)
  {
#ifdef _OPENMP
    i32 ithread = omp_get_thread_num() + 1, nth = omp_get_num_threads();
#else
    i32 ithread = 1, nth = 1;
#endif
    i32 *w = work + 64 * (ithread - 1);
    i32 *ijkv_sdm = w, *topo_socket = w + 3, *ind_dm_socket = w + 6, *topo_omp = w + 12, *ind_dm_omp = w + 15;
    i32 one = 1, zero = 0, lmin = explicit_ ? 4 : 10, ndo = nd, mxs = mx_synchro;
    i32 nitcfg = nstep, nss = nssiter, nitr = nitrun, fit = first_it;
    double norm_kry = 0., t = temps;
    indice_boucle_lu(&ndo, &one, &one, &lmin, ind_dm_zone, topo_socket, ind_dm_socket);
    indice_boucle_lu(&ndo, &ithread, &nth, &lmin, ind_dm_socket, topo_omp, ind_dm_omp);
    navier_stokes_struct(&ndo, &nth, &ithread, &ithread, &zero, &one, &one, &one, &mxs,
                         &zero, &zero, &one, &nitcfg, &nss, &nitr, &fit,
                         &zero, &zero,
                         pint, preal, &t,
                         ijkv_sdm, ind_dm_zone, ind_dm_socket, ind_dm_omp, topo_socket,
                         lok, topo_omp, timer + 2 * (ithread - 1),
                         alias, &norm_kry, cfl + 3 * (ithread - 1),
                         x, y, z, alias, alias,
                         ro, ro_m1, ro_p1, ro_ssiter,
                         xmut,
                         ti, tj, tk, vol,
                         ti, tj, tk, vol,
                         alias, alias, alias,
                         wig, wig, wig,
                         drodm, coe, alias, alias, alias);
#pragma omp master
    used = nth;
  }
  return used;
}
