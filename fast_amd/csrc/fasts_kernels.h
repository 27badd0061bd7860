// Launch interface between the C ABI (fasts_capi.hip) and the kernels (fasts_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "fasts_flux.h"

// one BC window: ghost cells lo..hi (interior-based indices, ghosts <= 0 or > n)
// device-side BC kinds: the tags of Fast/FastC/FastC/PyTree.py:30-82 after the dispatch of BCzone.cpp
// (tag 4 is a slip wall for Euler and a no-slip wall otherwise)
// FS_BC_ZERO: the extrapolation BC applied to the Newton residual (lrhs = 1): ghost = first interior cell * 0
// (BCExtrapolate_init.for:2-6)
enum { FS_BC_EXTRAPOLATE = 0, FS_BC_FARFIELD = 1, FS_BC_SUPERSONIC = 2, FS_BC_INFLOW = 13, FS_BC_WALL_SLIP = 3, FS_BC_WALL_NOSLIP = 6, FS_BC_OUTFLOW = 10, FS_BC_PERIODIC = 11, FS_BC_EXTRAP_NUT = 14, FS_BC_OUTPRES = 16, FS_BC_INJ1 = 17, FS_BC_INJMFR = 20, FS_BC_OUTMFR = 21, FS_BC_WALL_ISOTHERMAL = 32, FS_BC_ZERO = 100 };

struct FsBcWin {
  int type;          // FS_BC_*
  int idir;          // 1..6 = imin,imax,jmin,jmax,kmin,kmax
  int lo[3], hi[3];  // ghost cells of the window (interior-based indices)
  long long shift;   // periodic source offset (bvbs_periodique.for: incp)
  double st[5];      // conservative reference state (BCzone.cpp:26-38)
  double tcn;        // face metric along the normal (ti, tj or tk of the Cartesian zone)
  double nue;        // reference nu_tilde of the SA model (BCFarfield_init.for: state(6)/state(1));
                     // FS_BC_EXTRAP_NUT: the lower bound RONUTILDEINF (BCExtrapolateRansLes_init.for)
  // window data (device): one value per face cell of the BC window, value of face cell (a,b) at
  // (a - dlo[0]) + (b - dlo[1]) * dn, a = first transverse direction (BC/INCREMENT_BC.h, indbci)
  const double *data;
  int dlo[2], dn;
  int dsize;         // values per array when the window carries several arrays back to back
  int nit;           // FS_BC_INJ1: largest number of Newton iterations (NIT_INFLOW) and their tolerance (EPSI_INFLOW)
  double tol;
};

// depth = ghost layers covered besides the interior: 1 for the explicit scheme (navier_stokes_struct.for:223,
// range ind_grad), the whole ghost depth for the implicit one (:219, range ind_coe)
void fsk_invist(const FsZone &z, int depth, const double *rop, double *xmut, hipStream_t s);
// selective frequency damping of the interior (Compute/sfd.for): rof relaxes towards rop, then (pull != 0) rop is
// pulled towards rof; coe1 = dt/vol of every cell (coe(:,1) of the implicit scheme)
void fsk_sfd(const FsZone &z, double chi, double delta, int pull, double *rop, double *rof, const double *coe1, hipStream_t s);
// stage kernel of Cartesian zones, planes staged by LDS-DMA (fasts_stage_dma.hip): flux balance + Runge-Kutta update
// (or Newton residual); at the first stage it also evaluates and stores the sensor.  The zone is cut in nti x ntj x ntk
// tiles of ti x tj x kchunk cells, numbered bi + nti*(bj + ntj*bk); blist (device, nlist entries)
// restricts a launch to a subset of the tiles, NULL launches all of them.
struct FsTiling { int nti, ntj, ntk, kchunk, ti, tj, pace; };
FsTiling fsk_stage_dma_tiling(const FsZone &z);
// bdf != NULL: the kernel writes the Newton residual of the implicit scheme to drodm instead of the
// Runge-Kutta update (rop_out unused, coefH / coefW ignored)
// keep_balance = 0 (Runge-Kutta form only): the balance is not written to drodm; for the last stage of a step, whose
// balance no later stage, norm or correction reads
int fsk_stage_dma(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
                  double *drodm, const double *xmut, unsigned char *wig, const int *blist, int nlist, const FsBdf *bdf,
                  int keep_balance, hipStream_t s);
// stage kernel of curvilinear zones (zone type 0, fasts_stage_full.hip); sensors stored as bytes
// blist / nlist: a subset of the patches (numbered px + nti * (py + ntj * pz) in the grid of fsk_stage_full_tiling);
// NULL = all of them
FsTiling fsk_stage_full_tiling(const FsZone &z);
int fsk_stage_full(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
                   double *drodm, const double *xmut, unsigned char *wig, const FsBdf *bdf, const double *dist,
                   int keep_balance, const int *blist, int nlist, hipStream_t s);
void fsk_sa_coe6_full(const FsZone &z, const double *rop, const double *dist, double *coe, hipStream_t s);
// Spalart-Allmaras on Cartesian zones (fasts_stage_sa.hip): total viscosity on the whole zone, stage kernel
void fsk_vispalart(const FsZone &z, const double *rop, double *xmut, hipStream_t s);
int fsk_stage_sa(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
                 double *drodm, const double *xmut, const double *dist, unsigned char *wig, const FsBdf *bdf,
                 int keep_balance, hipStream_t s);
// coe(:,6), diagonal of the nu_tilde equation, interior + one ghost layer; needs coe(:,1) and coe(:,5)
void fsk_sa_coe6(const FsZone &z, const double *rop, const double *dist, double *coe, hipStream_t s);
// o3sc slope (SLOPE 6; 5 equations): the shock sensor of every face (3 doubles per cell), evaluated before every
// sub-iteration, and stage kernels of their own that read it (one thread per cell; the tiled kernels are not involved)
void fsk_sciacovelli(const FsZone &z, const double *rop, double *wsc, hipStream_t s);
int fsk_stage_o3sc_cart(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
                        double *drodm, const double *xmut, unsigned char *wig, const double *wsc, const FsBdf *bdf,
                        int keep_balance, hipStream_t s);
// flux: work array of 15 doubles per cell (the fluxes of the faces of the three directions)
int fsk_stage_o3sc_full(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
                        double *drodm, const double *xmut, unsigned char *wig, const double *wsc, double *flux,
                        const FsBdf *bdf, int keep_balance, hipStream_t s);
struct FsBox { int b[6]; };   // i1, i2, j1, j2, k1, k2
// Sub-zones: the flux loops of the reference add the flux of a face to the cell on its upper side and take it from the
// cell on its lower side, the cells just outside the range included (Compute/template_FluxAndBalance.for:192-271,
// assemble_drodm_plus_vec1.for, flu_send_scater_vec1.for).  For a box inside a zone the interior cells that touch it
// therefore see what they held of the balance change by the flux of the face they share with the box -- nothing the
// solution depends on, but the norms cprdu3s1 takes of blocks it counts as computed do.  One launch, after the stage
// kernel of the box: - F on the lower neighbours, + F on the upper ones.  _cart: Cartesian zones with 5 equations;
// _sa: Cartesian zones with the Spalart-Allmaras model; _full: the other zone types.
void fsk_box_faces_cart(const FsZone &z, const int *box, const double *rop, const double *xmut, const unsigned char *wig,
                        double *drodm, hipStream_t s);
void fsk_box_faces_sa(const FsZone &z, const int *box, const double *rop, const double *xmut, unsigned char *wig,
                      double *drodm, hipStream_t s);
void fsk_box_faces_full(const FsZone &z, const int *box, const double *rop, const double *xmut, unsigned char *wig,
                        double *drodm, hipStream_t s);
// Sub-zones: the source term of the nu_tilde equation is evaluated on the box and the layer of cells around it
// (ind_ssa of Fast/FastC/FastC/HPC_LAYER/indice_boucle_ssdom.for:105-160; SA/spsource_SA.for adds it to drodm(:,6)
// there), the balance is started from zero on the box alone: the interior cells of that layer keep what they held of
// the 6th equation plus the source.  box = i1, i2, j1, j2, k1, k2.  _full: curvilinear, k-homogeneous and 2-D zones.
void fsk_sa_ring_source(const FsZone &z, const int *box, const double *rop, const double *dist, double *drodm, hipStream_t s);
void fsk_sa_ring_source_full(const FsZone &z, const int *box, const double *rop, const double *dist, double *drodm,
                             hipStream_t s);
// kinds: bit 0 = the table holds periodic / extrapolation windows (copy kernel), bit 1 = walls / far field.
// The windows of one table must not overlap: the two kernels are not ordered cell by cell.
void fsk_bc_windows(const FsZone &z, int nwin, const FsBcWin *dev_wins, int kinds, long long max_cells, double *rop,
                    hipStream_t s);
// box != NULL: i1, i2, j1, j2, k1, k2 of the interior cells the fix is made for (a sub-zone); NULL = the zone
void fsk_bc_edges(const FsZone &z, double *rop, hipStream_t s, const int *box = nullptr);
void fsk_join_copy(int npts, int neq, long long nd_r, long long nd_d, const int *rcv, const int *dnr,
                   double *rop_r, const double *rop_d, hipStream_t s);
void fsk_halo_pack(int npts, int neq, long long nd, const int *pts, const double *rop, double *buf, hipStream_t s);
void fsk_halo_unpack(int npts, int neq, long long nd, const int *pts, double *rop, const double *buf, hipStream_t s);
// use_coe = 1: norms of dt/vol * drodm (explicit balance); 0: norms of drodm (Newton residual, already scaled);
// coe1 != NULL: the per-cell coe(:,1) of a local time step.  part: 12 doubles per (j,k) row of scratch;
// out[0..5] = sums of squares, out[6..11] = largest magnitudes (5 or 6 equations)
void fsk_residual(const FsZone &z, int use_coe, const double *drodm, double *part, double *out, hipStream_t s,
                  const double *coe1 = nullptr);
// norms of drodm per block of size[] cells (nblk[] blocks per direction): 12 doubles per block, as `out` above
void fsk_residual_blocks(const FsZone &z, const int nblk[3], const int size[3], const double *drodm, double *out,
                         hipStream_t s);
// out[3] = largest, smallest and mean local CFL number of the interior cells (cpcfl3.for); part = 3 doubles per
// (j,k) row of scratch; gam1 = gamma / Prandtl, gam2 = gamma * R
void fsk_cfl(const FsZone &z, double gam1, double gam2, const double *rop, const double *xmut, double *part, double *out,
             hipStream_t s);
// kinetic energy / enstrophy sums of a Cartesian zone added to out[3] = {enstrophy, kinetic energy, volume};
// part = 2 doubles per (j,k) row of scratch
// a[i] = b[i] + s * c[i] over n doubles, 16 bytes per lane: the streaming ceiling a memory-bound kernel is held against
void fsk_triad(long long n, double *a, const double *b, const double *c, double s, hipStream_t st);
void fsk_tgv_sums(const FsZone &z, const double *rop, double *part, double *out, hipStream_t s);
// implicit scheme (fasts_lusgs.hip)
void fsk_coe_implicit(const FsZone &z, int first_it, int dtloc, double cfl, double gamma, double prandtl, double rgp,
                      const double *rop, const double *xmut, double *coe, hipStream_t s);
// range[6] = cells swept (interior, plus one ghost layer on wall faces)
// apart = true: the blocks are boxes that do not cover `range` (sub-zones): nothing outside them is touched
// flow: state of the one-launch-per-sweep form of a zone (used for one sweep of the range; nullptr: one launch per
// hyperplane); fsk_luflow_gave_up: 1 once a wave of such a launch has stopped waiting for a predecessor (a defect)
struct FsLuFlow;
FsLuFlow *fsk_luflow_new();
void fsk_luflow_free(FsLuFlow *f);
void fsk_luflow_concurrency(FsLuFlow *f, int n);   // sweeps of n zones run side by side: fewer waves per launch
int fsk_luflow_gave_up(const FsLuFlow *f);
int fsk_lusgs(const FsZone &z, const int *range, int nblocks, const int *blocks, const int *blocks_dev, double gamma,
              const double *rop, double *rop_out, double *scratch, double *dw, const double *coe, hipStream_t s,
              bool apart = false, FsLuFlow *flow = nullptr);
int fsk_lussor(const FsZone &z, const int *range, const int *sdm, int nblocks, const int *blocks, const int *blocks_dev,
               int nb_relax, int overlap, double gamma, const double *rop, double *rop_out, double *scratch, const double *din,
               double *work, const double *coe, hipStream_t s, FsLuFlow *flow = nullptr);

// roe + walls (fasts_wallflux.hip): correction of the balance on the cells next to one wall window (glo/ghi = its
// ghost-cell window, tag = BC tag of the reference), then the end of the stage redone from the corrected balance
void fsk_roe_wall(const FsZone &z, int idir, const int *glo, const int *ghi, int tag, const double *rop, double *drodm,
                  hipStream_t s);
void fsk_finish_wall_rows(const FsZone &z, int idir, const int *glo, const int *ghi, double coefW, const double *rop_in,
                          double *rop_out, double *drodm, hipStream_t s);
void fsk_finish_bdf(const FsZone &z, const double *rop_in, double *drodm, const FsBdf &bdf, hipStream_t s);
void fsk_finish_rk_local(const FsZone &z, double coefW, const double *coe1, const double *rop_in, double *rop_out,
                         double *drodm, hipStream_t s);
