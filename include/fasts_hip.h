/*
 * fasts_hip.h -- C ABI of the MI355X (gfx950) FastS time-step library (libfasts_hip.so).
 *
 * This is the drop-in boundary for the structured-grid compressible Navier-Stokes time
 * step of IvanMary69/Fast: the entry points below are what a binding of the reference's
 * CPython extension modules `fasts` / `fastc` would call instead of its OpenMP/Fortran
 * path.  Each declaration cites the reference interface it replaces (paths relative to the
 * reference tree).  Plain pointers and sizes only; every function returns 0 on success or a
 * negative error code, never exits the process; fs_last_error() gives the message.
 *
 * Data contract (same as the reference, Fast/FastS/FastS/Compute/computePT.cpp:225-369):
 *   - Parameter_int  : int32 array, layout of Fast/FastS/FastS/param_solver.h:77-214 with the
 *                      BC table appended from Parameter_int[PT_BC]
 *                      (Fast/FastC/FastC/PyTree.py:2958-3014)
 *   - Parameter_real : float64 array, layout of param_solver.h:225-306
 *   - a time level   : NEQ contiguous fields of NDIMDX float64 (ghost cells included,
 *                      i fastest), i.e. the buffer FastC._compact builds
 *                      (Fast/FastC/FastC/PyTree.py:2256-2268)
 *   - metrics        : [ti | tj | tk | vol] as allocated by fasts.allocate_metric
 *                      (Fast/FastS/FastS/Metric/allocate_metric.cpp:281-284)
 * Host arrays stay owned by the caller (the pyTree); the library keeps device copies and
 * moves data only in fs_upload / fs_download.
 *
 * Threading: one host thread per context; contexts are independent (one per GPU / rank).
 */
#ifndef FASTS_HIP_H
#define FASTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fs_ctx fs_ctx;

/* error codes */
#define FS_OK 0
#define FS_ERR_ARG (-1)         /* bad argument / inconsistent sizes                     */
#define FS_ERR_UNSUPPORTED (-2) /* option outside what the device path implements       */
#define FS_ERR_HIP (-3)         /* HIP runtime error (message in fs_last_error)          */
#define FS_ERR_STATE (-4)       /* call order violated (e.g. compute before upload)      */

/* time-level selectors for fs_download / fs_fill_ghostcells / fs_apply_bc */
#define FS_LEVEL_N 1
#define FS_LEVEL_M1 2
#define FS_LEVEL_P1 4
#define FS_LEVEL_FILTERED 8   /* fs_download only: the filtered state of selective frequency damping */

/* Library version, build flags (for the "is the native path loaded" check). */
const char *fs_version(void);

/* Create / destroy a solver context bound to one HIP device.
 * Replaces the module-level state of the reference glue (FastC.HOOK work arrays created by
 * Fast/FastC/FastC/PyTree.py:2058-2224: drodm, coe, wiggle). */
int fs_create(fs_ctx **ctx, int device);
void fs_destroy(fs_ctx *ctx);
const char *fs_last_error(const fs_ctx *ctx);

/* Use an existing HIP stream (hipStream_t passed as void*) for every launch and copy of this
 * context; NULL selects the context's own stream. */
int fs_set_stream(fs_ctx *ctx, void *hip_stream);

/* Verification period: residual norms are evaluated on sub-iteration 1 of the time steps with
 * nitrun % modulo_verif == 0 or nitrun == 1, the rule the reference applies to its CFL /
 * residual diagnostics (Fast/FastS/FastS/Compute/computePT.cpp:100; modulo_verif is
 * .Solver#dtloc[1], Fast/FastC/FastC/PyTree.py:1186).  Default 200. */
int fs_set_verif(fs_ctx *ctx, int modulo_verif);

/* FIRST_IT of the reference (FastC.HOOK["FIRST_IT"], Fast/FastS/FastS/PyTree.py:235,
 * Fast/FastC/FastC/PyTree.py:2195-2222): 0 = no time step completed yet, the implicit scheme then uses
 * BDF1 and a unit diagonal shift (Fast/FastS/FastS/Compute/core3as2.for:37-45, tstb3c.for:67-68); 1 = BDF2.
 * A context starts at 0 and switches to 1 at the end of its first time step; a restart from a stored
 * M1 level sets 1 here. */
int fs_set_first_it(fs_ctx *ctx, int first_it);

/* 'implicit_local' (lssiter_loc = 1 in the reference's work dictionary, Fast/FastC/FastC/PyTree.py:2074-2076,2205, read by
 * Fast/FastS/FastS/Compute/computePT.cpp:125-127): every Newton sub-iteration works on the sub-zones -- boxes of blocks
 * of SIZE_SSDOM cells (Parameter_int[30..32]) -- whose residual had not dropped by EPSI_NEWTON at the last verification
 * step (Fast/FastC/FastC/Compute/souszones_list.cpp:31-90, init_ssiter_bloc.for, ssdom_lu_ijk.for;
 * Fast/FastS/FastS/Compute/cprdu3s1_incl.for:166-197).  0 (default) = every sub-iteration on the whole zones.  Implicit
 * zones only, before fs_upload; not with LU thread blocks, nb_relax > 1, lu_match = 1 or roe with walls. */
int fs_set_local_newton(fs_ctx *ctx, int lssiter_loc);

/* The boxes (i1, i2, j1, j2, k1, k2; at most max_boxes) that zone `zone` works on at sub-iteration nstep, as last
 * evaluated (ind_dm(6, :, nstep) and nidom_loc(nstep) of the reference's metrics[5] array,
 * Fast/FastS/FastS/Metric/allocate_metric.cpp:398-440).  Returns their number. */
int fs_subzone_list(fs_ctx *ctx, int zone, int nstep, int32_t *boxes, int max_boxes);
/* it_target_ssdom of zone `zone` (one entry per block, metrics[6] of the reference); returns the number of blocks. */
int fs_subzone_targets(fs_ctx *ctx, int zone, int32_t *it_target, int max_blocks);

/* The host logic of the sub-zones on plain arrays, no context and no device: the routines of the reference they
 * restate are called on the same inputs by the tests.
 * fs_subzone_skip_lu   = skip_lu (Fast/FastC/FastC/Compute/skip_lu.for): iskip_lu(mx_ssiter, 2)
 * fs_subzone_boxes     = ssdom_lu_ijk (Fast/FastC/FastC/Compute/ssdom_lu_ijk.for): the boxes (6 integers each, at most
 *                        max_boxes) of the blocks with it_target >= iseuil; no_lu / *count = the blocks they hold;
 *                        returns the number of boxes
 * fs_subzone_residual  = cprdu3s1 (Fast/FastS/FastS/Compute/cprdu3s1.for, cprdu3s1_incl.for) on block norms the caller
 *                        has reduced: norms = 12 doubles per block (sums of squares of 6 equations, their largest
 *                        magnitudes); rdm = rdm_sdm(nssiter, neq, 2, nblocks) */
int fs_subzone_skip_lu(int nitmax, int mx_ssiter, int32_t *iskip_lu);
int fs_subzone_boxes(const int32_t *ijkv, const int32_t *size_ssdom, int iseuil, const int32_t *it_target,
                     int32_t *no_lu, int32_t *count, int32_t *ind_dm, int max_boxes);
int fs_subzone_residual(const int32_t *ijkv, const int32_t *size_ssdom, int nssiter, int neq, int threed, int nstep,
                        int it_bloc, double epsi, const double *norms, const int32_t *no_lu, int nis, int32_t *it_lu,
                        int32_t *it_target, int32_t *it_target_old, int32_t *it_temp, double *rdm);

/* Register one zone.  Returns the zone index (>= 0) or an error code.
 * Replaces the per-zone pointer gathering of K_FASTS::_computePT
 * (Fast/FastS/FastS/Compute/computePT.cpp:225-369) and the metric list produced by
 * fasts.allocate_metric / fastc.init_metric
 * (Fast/FastS/FastS/Metric/allocate_metric.cpp:464-471).
 * ro / ro_m1 / ro_p1 are the host buffers of the three time levels (Density, Density_M1,
 * Density_P1 and the fields compacted behind them).
 * mtr is metrics[0] of the reference: [ti | tj | tk | vol].  Cartesian zone (ITYPZONE 2): four
 * scalars.  Curvilinear zone (ITYPZONE 0): ti,tj,tk(NDIMDX_MTR,3) and vol(NDIMDX_MTR), dimensioned
 * on the nodes with the strides of NIJK_MTR (Fast/FastC/FastC/Metric/nature_geom_dom.for:186-503,
 * cp_tijk.for:60-123, cp_vol.for:55-93); the array is copied to the device during this call.
 * Two-dimensional zone (ITYPZONE 3: one cell along k, NIJK(5) = 0): ti,tj(NDIMDX_MTR,2) and vol(NDIMDX_MTR)
 * on the nodes of one k-plane, no tk (nature_geom_dom.for:186-198, cp_tijk.for:376-402, cp_vol.for:177-212).
 * k-homogeneous zone (ITYPZONE 1, an extruded mesh): ti,tj(NDIMDX_MTR,2), tk(NDIMDX_MTR,1) and vol(NDIMDX_MTR) on the
 * nodes of one k-plane, no stride along k (nature_geom_dom.for:468-479, cp_tijk.for:196-289, cp_vol.for:115-146).
 * The BC table appended to param_int from PT_BC and the BC data appended to param_real
 * (Fast/FastC/FastC/PyTree.py:2958-3014) are read here: window data (pressure of a BCOutpres, wall temperature
 * of a BCWallViscousIsothermal, state arrays of a BCInflowSupersonic or BCInj1, the uniform state a far field or
 * inflow may carry, Fast/FastS/FastS/BC/BCzone.cpp:26-48,257,417-448,539-618) are copied to the device during
 * this call. */
int fs_zone_add(fs_ctx *ctx, const int32_t *param_int, int n_int, const double *param_real,
                int n_real, double *ro, double *ro_m1, double *ro_p1, const double *mtr,
                int n_mtr);

/* Wall distance of a Spalart-Allmaras zone (IFLOW 3): one value per cell, ghost cells included, i fastest
 * (the TurbulentDistance field, which the reference keeps in the second slot of its viscosity array,
 * Fast/FastS/FastS/Compute/src_term.for:93-96).  The pointer must stay valid until fs_upload. */
int fs_zone_set_distance(fs_ctx *ctx, int zone, const double *dist, int n);

/* Selective frequency damping (Parameter_int[SFD] = 1; implicit scheme): host buffer of the filtered state of a zone,
 * laid out like a time level (NEQ * NDIMDX doubles: Density_f, VelocityX_f ... compacted,
 * Fast/FastC/FastC/PyTree.py:357-361,678-687), uploaded by fs_upload and read back by fs_download(FS_LEVEL_FILTERED).
 * Once per time step, in its last sub-iteration that runs, the library applies sfd.for to the interior of the level
 * just written (Fast/FastS/FastS/Compute/lhs.cpp:181-185, Compute/sfd.for) with SFD_CHI, SFD_DELTA of Parameter_real
 * and, from time step SFD_INIT_IT on, pulls the state towards the filtered one.  The device exponential is not the
 * host's: results agree with the reference to 1e-12, not bit for bit.  Call before fs_upload. */
int fs_zone_set_filtered(fs_ctx *ctx, int zone, double *rof, long long n_doubles);

/* Blocks of the LU-SGS / LU-SSOR solve of a zone (implicit scheme, lu_match = 0).  The reference gives every OpenMP
 * thread a block of the zone (indice_boucle_lu, Fast/FastC/FastC/HPC_LAYER/indice_boucle_lu.for:94-335: cut along k
 * and j, along j and i on a two-dimensional zone, no block thinner than 10 cells) and each thread runs invlu on its own
 * block, skipping the neighbour terms across its borders (Fast/FastS/FastS/Compute/lhs.cpp:129-161,
 * LU/invlu_l.for:90-215, LU/invlu_u.for:93-263): the result of an implicit step depends on the number of threads.
 * Without this call the library gives the one-thread result (one lexicographic sweep of the zone).  With it, ind_sdm
 * holds nblocks ranges (imin, imax, jmin, jmax, kmin, kmax; interior cells numbered from 1, as ind_dm_thread of the
 * reference) that tile the interior, and the library gives the result of the reference run with those thread blocks:
 * the blocks are swept side by side, one hyperplane of every block per launch; a block that touches a face of the zone
 * carrying a wall takes the ghost layer there as the reference's test per thread does (BC/BCzone.cpp:99-127).  With
 * nb_relax > 1 every block is relaxed on its own as the threads do on work arrays of their own window
 * (Compute/ssor.cpp:52-70, LU/prep_lussor.h).  nblocks <= 1 restores the single sweep; refused with lu_match = 1 (there
 * the reference lets the ranges of the threads overlap, BC/BCzone.cpp:129-149).  Call before fs_upload. */
int fs_zone_set_lu_blocks(fs_ctx *ctx, int zone, int nblocks, const int32_t *ind_sdm);

/* The same for every zone of the context with the blocks the reference gives nthreads OpenMP threads (omp_mode 0:
 * every zone cut among all the threads, Fast/FastC/FastC/HPC_LAYER/distributeThreads.cpp:90-95, the cut of
 * indice_boucle_lu.for:94-335 with the 10-cell minimum of the implicit scheme, threads without cells dropped): what
 * __NUMTHREADS__ / OMP_NUM_THREADS is to the reference's result.  1 = one sweep per zone.  Call after the zones are
 * registered and before fs_upload. */
int fs_set_lu_threads(fs_ctx *ctx, int nthreads);

/* LU ranges of the blocks of a zone as the library holds them (6 integers each: the block, with the ghost layer on the
 * wall faces of the zone it touches), at most `capacity` of them copied to ind_lu; returns the number of blocks (0: one
 * sweep of the zone).  For checks. */
int fs_zone_lu_blocks(fs_ctx *ctx, int zone, int32_t *ind_lu, int capacity);

/* Match-join ghost transfer tables (one entry per join "rac"): receiver zone, donor zone and
 * the two index lists (0-based, ghost-inclusive linear cell indices i + j*ni + k*ni*nj), i.e.
 * the PointListDonor / PointList arrays of a 'direct' (type 0) ID join, as hand-built in
 * Fast/FastS/test/tgvMpi.py:141-195 and consumed by K_FASTC::setInterpTransfersIntra
 * (Fast/FastC/FastC/Com/setInterpTransfersFast.cpp:440-935).
 * donor_zone < 0 marks a join whose donor lives in another context (another GPU): its values
 * arrive through fs_halo_unpack, and -donor_zone-1 is the caller's peer tag. */
int fs_join_add(fs_ctx *ctx, int rcv_zone, int donor_zone, const int32_t *rcv_pts,
                const int32_t *donor_pts, int npts);

/* Send side of an inter-context join: donor cells of local zone `donor_zone` whose values go
 * to peer `peer_tag`.  Replaces the packing loop of K_FASTC::setInterpTransfersInter
 * (Fast/FastC/FastC/Com/setInterpTransfersFast.cpp:944-1582). */
int fs_send_add(fs_ctx *ctx, int donor_zone, int peer_tag, const int32_t *donor_pts, int npts);

/* All joins at once from the flat connectivity arrays tc/Parameter_int and tc/Parameter_real, in the layout
 * K_FASTC::setInterpTransfersIntra / Inter read (Fast/FastC/FastC/Com/setInterpTransfersFast.cpp:114-118,220,
 * 469-478,522-528,646-653,748-758): blocks whose destination is `rank` become local joins (fs_join_add), the others
 * sends to that rank (fs_send_add_remote; the receiver zone number in the arrays is the zone's index on the
 * destination rank, so every rank gives its zones their own index as global id).  Zone numbers are the indices
 * fs_zone_add returned.  Accepted: steady match joins (one donor per point, unit coefficient, 5 or 6 variables, no
 * rotation, no immersed boundary); anything else returns FS_ERR_UNSUPPORTED.  Call before fs_upload. */
int fs_joins_set(fs_ctx *ctx, const int32_t *pint_tc, int n_int, const double *preal_tc, int n_real, int rank);

/* Host -> device copy of the three time levels of every zone; allocates device storage on
 * first call. */
int fs_upload(fs_ctx *ctx);

/* Device -> host copy of the selected time levels (FS_LEVEL_* mask) into the buffers that
 * currently play those roles. */
int fs_download(fs_ctx *ctx, int levels_mask);

/* Advance nit_c time steps, each running sub-iterations nstep_deb..nstep_fin
 * (RHS -> update -> joins -> BCs -> edge/corner fix) and rotating the time levels at the end
 * of a step.  Replaces fasts._computePT(zones, metrics, nitrun, nstep_deb, nstep_fin,
 * layer_mode=1, nit_c, hook) (Fast/FastS/FastS/Compute/computePT.cpp:44-719) and
 * K_FASTS::gsdr3 (Fast/FastS/FastS/Compute/gsdr3.cpp:38-461).  *t_trans receives 0 (transfer
 * time is not separable on the device). Asynchronous: returns after enqueueing. */
int fs_compute(fs_ctx *ctx, int nitrun, int nstep_deb, int nstep_fin, int nit_c, double *t_trans);

/* Split form of one sub-iteration for callers that exchange halos between contexts:
 *   fs_stage          RHS + update of every local zone          (gsdr3.cpp:251-269)
 *   fs_halo_pack      gather the donor cells for peer_tag into dev_buf (device pointer,
 *                     neq*npts doubles, variable-major)          (setInterpTransfersFast.cpp:1200-1206)
 *   fs_halo_unpack    scatter dev_buf into the ghost cells of the joins tagged peer_tag
 *                                                                (setInterpTransfersFast.cpp:1652-1667)
 *   fs_local_joins    intra-context joins                        (setInterpTransfersFast.cpp:440-935)
 *   fs_bcs            BCs + edge/corner fix on the stage output  (gsdr3.cpp:403-422, bcs.cpp:30-41)
 *   fs_rotate         M1 <- N <- P1 <- M1                        (computePT.cpp:673-677)        */
int fs_stage(fs_ctx *ctx, int nitrun, int nstep);
/* fs_stage in two launches, so that halo messages overlap the bulk of the computation (the
 * reference runs its transfers after the whole RHS + update, Fast/FastS/FastS/Compute/gsdr3.cpp:251-349,
 * transfert_multiblock.cpp:11-45; the results are the same): part 0 = the tiles holding donor cells
 * of joins to other contexts, part 1 = everything else.  After part 0 the donor cells of this
 * sub-iteration are final and fs_halo_pack may run on another stream. */
int fs_stage_part(fs_ctx *ctx, int nitrun, int nstep, int part);
/* Stream (hipStream_t as void*) on which fs_halo_pack / fs_halo_unpack launch; NULL = the context's
 * stream.  The caller orders the two streams with events. */
int fs_set_comm_stream(fs_ctx *ctx, void *hip_stream);
int fs_halo_count(fs_ctx *ctx, int peer_tag, int is_send);
int fs_halo_pack(fs_ctx *ctx, int nstep, int peer_tag, double *dev_buf, int n_doubles);
int fs_halo_unpack(fs_ctx *ctx, int nstep, int peer_tag, const double *dev_buf, int n_doubles);
int fs_local_joins(fs_ctx *ctx, int nstep);
int fs_bcs(fs_ctx *ctx, int nstep);
int fs_rotate(fs_ctx *ctx);

/* ---- halo transport inside the library: RCCL send/recv between the contexts of a job (one per GPU) --------------
 * Replaces the MPI side of K_FASTC::setInterpTransfersInter / getTransfersInter
 * (Fast/FastC/FastC/Com/setInterpTransfersFast.cpp:944-1582 pack + isend, :1590-1719 recv + scatter, driven by
 * Fast/FastS/FastS/Mpi.py:117-233 once per sub-iteration).  With a communicator set up, fs_compute runs the whole
 * sub-iteration: part 0 of the stage -> pack / ncclSend+ncclRecv (one group) / unpack on a stream of its own, while
 * part 1 runs on the context's stream -> local joins -> BCs.  librccl is opened at run time by the first of these
 * calls; a job of one context never loads it.
 *
 *   fs_comm_unique_id    128 bytes identifying a new communicator; one rank creates them, the caller carries them to
 *                        the others (file, socket, MPI broadcast: the library opens no connection of its own)
 *   fs_comm_init         join the communicator as `rank` of `world` (collective, blocks until every rank has called)
 *   fs_zone_set_global_id  job-wide number of a local zone (the same numbering on every rank, e.g. the position of
 *                        the zone name in the sorted graph['procDict'], Fast/FastS/test/tgvMpi.py:198-256)
 *   fs_send_add_remote   fs_send_add + the receiver side of the join: zone (global id) and ghost cells on
 *                        `peer_rank` that take the values (PointListDonor); the reference ships that list with every
 *                        message (setInterpTransfersFast.cpp:1185-1206), here it travels once, in fs_comm_setup
 *   fs_comm_setup        after fs_upload, collective: exchanges the receiver lists, creates the ghost-cell joins fed
 *                        by each peer and the message buffers
 *   fs_halo_exchange     one round of messages outside fs_compute: nstep >= 1 on the output level of that
 *                        sub-iteration, nstep <= 0 on time level -nstep (ghost cells of the initial field,
 *                        Fast/FastS/FastS/Mpi.py:934)
 *   fs_comm_allreduce    n <= 64 host doubles reduced over the ranks in place, op 0 sum / 1 max / 2 min, after waiting
 *                        for the context's stream: the barrier and the max-over-ranks clock of a benchmark, the global
 *                        sums of diagnostics (the reference reduces over MPI the same way, Fast/FastS/FastS/Mpi.py:473,954)   */
int fs_comm_unique_id(void *id, int nbytes);
int fs_comm_init(fs_ctx *ctx, int rank, int world, const void *id, int nbytes);
int fs_zone_set_global_id(fs_ctx *ctx, int zone, int gid);
int fs_send_add_remote(fs_ctx *ctx, int donor_zone, int peer_rank, const int32_t *donor_pts, int rcv_gid,
                       const int32_t *rcv_pts, int npts);
int fs_comm_setup(fs_ctx *ctx);
int fs_halo_exchange(fs_ctx *ctx, int nstep);
int fs_comm_allreduce(fs_ctx *ctx, double *vals, int n, int op);

/* Joins then BCs then edge/corner fix on one time level: replaces
 * FastS.PyTree._fillGhostcells (Fast/FastS/FastS/PyTree.py:1153-1218) for local joins. */
int fs_fill_ghostcells(fs_ctx *ctx, int level);

/* BCs + edge/corner fix only: replaces fasts._applyBC(zones, metrics, hook, nstep, var)
 * (Fast/FastS/FastS/BC/applyBC.cpp:37-39). */
int fs_apply_bc(fs_ctx *ctx, int nstep, int level);

/* Residual norms of the first sub-iteration of the last verification step of one zone, per equation (neq = 5 or 6):
 * l2[e] = sqrt(sum r^2 / ncell)/dt, loo[e] = max|r|/dt, r = Newton residual (implicit scheme) or dt/vol * balance of
 * the first stage (explicit scheme, where the reference tabulates nothing)
 * (Fast/FastS/FastS/Compute/cprdu3s1_incl.for:47-71 +
 *  Fast/FastS/FastS/DISPLAY/display_ss_iteration.cpp:128-141).  Synchronises. */
int fs_residuals(fs_ctx *ctx, int zone, double *l2, double *loo, int neq);

/* Residual history of the last verification step of one zone: the norms above for every Newton sub-iteration
 * 1..NSSITER (the last slot is the verification pass on the final iterate), i.e. rdm(it_bloc, neq, 2) as
 * Fast/FastS/FastS/Compute/residus_navier.h:12-43 -> cprdu3s1.for fills it and
 * Fast/FastS/FastS/DISPLAY/display_ss_iteration.cpp:112-146 reduces it over the sub-blocks, the table
 * dpssiter.for:88-99 prints.  l2 / loo: [nslot][neq], sub-iteration major.  Returns the number of slots filled
 * (1 for the explicit scheme) or an error code.  Synchronises. */
int fs_residual_history(fs_ctx *ctx, int zone, double *l2, double *loo, int nslot, int neq);

/* Largest, smallest and mean local CFL number of one zone, cfl[3] = {max, min, mean}, evaluated at the first
 * sub-iteration of the last verification step on the state that step started from: what the reference stores
 * in .Solver#ownData/CFL_minmaxmoy (Fast/FastS/FastS/Compute/cptst3.for:114-120 -> cpcfl3.for,
 * Fast/FastS/FastS/Compute/computePT.cpp:519,610-643) and prints in
 * Fast/FastS/FastS/DISPLAY/display_ss_iteration.cpp:100-105.  Synchronises. */
int fs_cfl(fs_ctx *ctx, int zone, double *cfl);

/* Kinetic energy and enstrophy of the Taylor-Green diagnostics, summed over the (Cartesian) zones of the context on
 * the given time level: sums[3] = {sum 0.5 rho |curl u|^2 vol, sum 0.5 rho |u|^2 vol, sum vol}.  The reference returns
 * sums[0]/sums[2] and sums[1]/sums[2] (fasts.computePT_enstrophy, Fast/FastS/FastS/POST/computePT_enstrophy.cpp:176-227
 * -> POST/post.for, cp_gradu_3dcart.for with the 4-point face average, cptaylor.for; FastS.PyTree._computeEnstrophy,
 * Fast/FastS/FastS/PyTree.py:1869-1882); a multi-GPU caller adds the sums of its ranks first.  Synchronises. */
int fs_tgv_sums(fs_ctx *ctx, int level, double *sums);

/* fs_tgv_sums without any allocation, for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The three sums are written to dev_sums,
 * a DEVICE pointer to 3 doubles owned by the caller; the per-row partial sums go through a file-scope device array
 * of the library (4 MiB per GPU, held until the process ends) instead of the context's own scratch.  Limit: every
 * zone of the context at most 262144 (j,k) rows (jv * kv), else FS_ERR_ARG.  Asynchronous on the context's stream:
 * no synchronisation, no hipMalloc / hipFree.  All contexts of one GPU share the array: calls that overlap on one
 * GPU (different streams) race; calls on one stream are ordered. */
int fs_tgv_sums_static(fs_ctx *ctx, int level, double *dev_sums);

/* Page-locked host memory for the level buffers.  Replaces the allocation of the compacted time levels in
 * FastC._compact (Fast/FastC/FastC/PyTree.py:2256-2268: one numpy buffer per level): a buffer obtained here is
 * copied by fs_upload / fs_download at the rate of the host link (57 GB/s measured on an MI355X box) instead of
 * through the runtime's pageable staging copy (17 GB/s device to host).  device = the GPU of the context that will
 * move the buffer; nbytes > 0; *ptr is NULL on failure (no GPU,
 * no memory): the caller then uses ordinary memory.  fs_host_free releases what fs_host_alloc returned. */
int fs_host_alloc(int device, void **ptr, size_t nbytes);
int fs_host_free(void *ptr);

/* Block until everything enqueued on the context's stream has finished. */
int fs_sync(fs_ctx *ctx);

/* Measurement aid without a counterpart in the reference: streaming rate (GB/s) of the context's device on a triad
 * a = b + s*c over three arrays of n_doubles (even) doubles, HIP events over `reps` launches.  The ceiling the
 * memory-bound stage kernel is held against next to the 8 TB/s of the data sheet.  Allocates and frees 24 n bytes. */
int fs_stream_triad(fs_ctx *ctx, long long n_doubles, int reps, double *gbs);

/* Average device time in milliseconds of the stage kernel (RHS + update) launches since the
 * last call with reset != 0; measured with HIP events on the launch stream.  *n_launch
 * receives the number of launches averaged. */
int fs_stage_kernel_ms(fs_ctx *ctx, int reset, double *avg_ms, int *n_launch);

/* Copy one raw device work array of a zone to the host (tests / diagnostics):
 * which = 0 drodm (neq*ndimdx), 2 xmut (ndimdx). */
int fs_debug_fetch(fs_ctx *ctx, int zone, int which, double *host, int n_doubles);

#ifdef __cplusplus
}
#endif
#endif /* FASTS_HIP_H */
