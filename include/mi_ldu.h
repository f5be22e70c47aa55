/*
 * mi_ldu.h -- C ABI of the MI355X-native lduMatrix / fvMatrix compute engine.
 *
 * This is the drop-in boundary for the hot path of SimFlowCFD/RapidCFD-dev
 * (SURVEY.md section 8b).  The reference has no FFI: the path sits behind
 * OpenFOAM's run-time selection tables (lduMatrix::solver / preconditioner /
 * smoother, lduMatrix.H:141-185,297-341,438-460) and the lduMatrix member
 * functions.  A maintainer binds this ABI from a C++ shim that registers the
 * same run-time names (INTEGRATION.md shows the stub); the C++ host mirror of
 * those classes lives in rapidcfd-dev_amd/foam/.
 *
 * Conventions
 *   - scalar = double, label = int32_t (reference defaults: etc/bashrc:76,
 *     primitives/ints/label/label.H:57-66).
 *   - Pointers named *_dev are DEVICE pointers owned by the caller (the
 *     reference's gpuList<T> storage, gpuList.H:23-112); *_host are host
 *     pointers.  The engine never frees caller memory.
 *   - Vectors passed across this ABI are in the CALLER's cell/face order.
 *     The engine keeps its own tiled order internally; the
 *     *_engine entry points work on vectors already in engine order.
 *   - Every function returns MI_OK (0) or a negative error code; the message
 *     is available from mi_last_error().  The reference aborts the process
 *     instead (FatalError / CUDA_CALL, DeviceConfig.H:6-11); the shim turns a
 *     non-zero status into FatalError.
 *   - One context per GPU / per rank, not thread-safe (the reference is one
 *     host thread per MPI rank, argList.C:775-811).
 *   - There is no CPU fallback: every compute entry point needs a gfx950
 *     device and fails with MI_ERR_DEVICE otherwise.
 *
 * All paths in citations are relative to /root/reference/src/OpenFOAM/matrices/lduMatrix/
 * unless they start with src/.
 */
#ifndef MI_LDU_H
#define MI_LDU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_ctx_s *mi_ctx_t;
typedef struct mi_addr_s *mi_addr_t;
typedef struct mi_matrix_s *mi_matrix_t;
typedef struct mi_gamg_s *mi_gamg_t;
typedef struct mi_patch_s *mi_patch_t;
typedef struct mi_comm_s *mi_comm_t;

enum {
    MI_OK = 0,
    MI_ERR_ARG = -1,
    MI_ERR_DEVICE = -2,   /* no gfx950 device / HIP failure */
    MI_ERR_ALLOC = -3,
    MI_ERR_STATE = -4,    /* e.g. coefficients not bound */
    MI_ERR_LIMIT = -5,    /* mesh exceeds a tile-format limit */
    MI_ERR_UNSUPPORTED = -6 /* a combination this build does not provide (stated where it can occur) */
};

/* preconditioner names of lduMatrix::preconditioner::New
 * (lduMatrix/lduMatrixPreconditioner.C:67-158).  DIC/DILU resolve to AINV in
 * the reference (preconditioners/DICPreconditioner/DICPreconditioner.C:42-58). */
enum { MI_PRECOND_NONE = 0, MI_PRECOND_DIAGONAL = 1, MI_PRECOND_AINV = 2 };

/* solverPerformance (src/OpenFOAM/matrices/LduMatrix/LduMatrix/SolverPerformance.H) */
typedef struct {
    double initialResidual;
    double finalResidual;
    double normFactor;
    int32_t nIterations;
    int32_t converged;
    int32_t singular;
    int32_t reserved;
} mi_solver_perf;

/* lduMatrix::solver::readControls (lduMatrix/lduMatrixSolver.C:167-173) */
typedef struct {
    double tolerance; /* default 1e-6 */
    double relTol;    /* default 0    */
    int32_t maxIter;  /* default 1000 */
    int32_t minIter;  /* default 0    */
} mi_solver_controls;

/* ---- context (replaces src/OpenFOAM/device/DeviceConfig.{H,C}, DeviceStream) ---- */
/* stream: a hipStream_t created by the caller (e.g. torch's current stream) or NULL
 * for the engine's own stream.                                                     */
int mi_ctx_create(int device, void *hip_stream, mi_ctx_t *out);
int mi_ctx_destroy(mi_ctx_t ctx);
int mi_ctx_synchronize(mi_ctx_t ctx);
/* run-time switches of a context (the environment variable of the same meaning sets the value a new context starts with):
 *   "pcg_persist" 0 / 1          the persistent PCG kernel for matrices whose tiles fit the CUs' registers (MI_PCG_PERSIST)
 *   "pcg_fuse_rp" 0 / 1          PCG on one GPU: residual update of an iteration + direction update of the next as one
 *                                launch that keeps the preconditioned residual on the chip (MI_PCG_FUSE_RP; same bits)
 *   "pcg_fuse_test" 0..3         tests: workgroups of that launch leave its barrier at once (1: every third, 2: the first)
 *   "win_direct" 0 / 1 / 2       operators of attached matrices as ONE launch whose boundary tiles read the halo window
 *                                (MI_WIN_DIRECT; 2: also between ranks that share a device -- tests)
 *   "gamg_graph_attached" 0 / 1  hipGraph replay of the V-cycle of a decomposed case (MI_GAMG_GRAPH_ATTACHED)             */
int mi_ctx_set_option(mi_ctx_t ctx, const char *name, int32_t value);
/* which solver paths ran on this context (diagnostics, tests): launches of the persistent PCG kernel -- one per batch of
 * iterations -- on plain (MI_STAT_PERSIST_PCG) / communicator-attached (MI_STAT_PERSIST_DPCG) matrices; runs of the grid
 * barrier litmus that gates that kernel (MI_STAT_BARRIER_LITMUS); V-cycles of a decomposed case replayed as a hipGraph
 * (MI_STAT_GAMG_GRAPH_ATTACHED); launches of the fused residual / direction kernel of PCG (MI_STAT_PCG_FUSED_RP) */
#define MI_STAT_PERSIST_PCG 0
#define MI_STAT_PERSIST_DPCG 1
#define MI_STAT_BARRIER_LITMUS 2
#define MI_STAT_GAMG_GRAPH_ATTACHED 3
#define MI_STAT_PCG_FUSED_RP 4
int mi_ctx_stat(mi_ctx_t ctx, int32_t which, int64_t *out);
const char *mi_last_error(void);
/* 1 if a usable gfx950 device is visible to this process, else 0 */
int mi_device_available(void);

/* ---- addressing (replaces lduAddressing demand-driven tables,
 *      lduAddressing/lduAddressing.H:128-145, .C:169-344; K23 in SURVEY.md) ----
 * lower/upper: host mirrors of lowerAddr/upperAddr (fvMeshLduAddressing.H:165-181),
 *   faces in OpenFOAM upper-triangular order is NOT required, only lower<upper.
 * patches: coupled (processor) interfaces, patch p has patch_sizes[p] faces whose
 *   internal cells are patch_face_cells_host[p][...] (fvMeshLduAddressing.H:113-120).
 * Builds the tiled engine layout once per mesh.                                   */
int mi_addr_create(mi_ctx_t ctx, int32_t n_cells, int32_t n_faces,
                   const int32_t *lower_addr_host, const int32_t *upper_addr_host,
                   int32_t n_patches, const int32_t *patch_sizes,
                   const int32_t *const *patch_face_cells_host, mi_addr_t *out);
/* Same, with LOCAL coupled patches (cyclic, cyclicLduInterfaceField: lduAddressing/lduInterfaceFields/
 * cyclicLduInterfaceField, src/finiteVolume/fields/fvPatchFields/constraint/cyclic): patch_nbr_cells_host[p] != NULL
 * gives, face by face, the local cell on the other side of patch p (the neighbour patch's faceCells in matching
 * order); the update result[faceCells] -= coeffs*psi[nbrCells] then needs no exchange and every solver entry
 * point works on such a matrix.  NULL entries are processor patches (ext region).                              */
int mi_addr_create_coupled(mi_ctx_t ctx, int32_t n_cells, int32_t n_faces,
                           const int32_t *lower_addr_host, const int32_t *upper_addr_host,
                           int32_t n_patches, const int32_t *patch_sizes,
                           const int32_t *const *patch_face_cells_host,
                           const int32_t *const *patch_nbr_cells_host, mi_addr_t *out);
/* cyclicAMI patch (row f3; src/finiteVolume/fields/fvPatchFields/constraint/cyclicAMI/cyclicAMIFvPatchField.C:195-224,
 * src/meshTools/AMIInterpolation/GAMG/interfaceFields/cyclicAMIGAMGInterfaceField/cyclicAMIGAMGInterfaceField.C:97-130,
 * AMIInterpolation/AMIInterpolationF.H:62-105): `patch` -- created WITHOUT neighbour cells, i.e. with an ext region like a
 * processor patch -- takes its neighbour values from the cells of `nbr_patch` of the same addressing through the AMI
 * addressing and weights of its side (srcAddress/srcWeights on the owner patch, tgtAddress/tgtWeights on the other):
 *     pnf[i] = sum_{k in [start[i], start[i+1])} weights[k] * ( factor * psi[ faceCells(nbr_patch)[ address[k] ] ] )
 * one fused multiply-add per term in address order (multiplyWeightedOp<plusEqOp> under nvcc's contraction), factor =
 * mi_matrix_set_patch_transform; low_weight[i] != 0 (weightsSum[i] < lowWeightCorrection, decided by the caller) replaces
 * the sum by the face's own cell value, the default the reference passes (`pif`).  The two sides may differ in size.  Every
 * operator and solver that reads coupled-patch neighbour values interpolates them before its tile pass; no exchange.
 * start/address/weights all NULL: one face to one face with unit weight -- a cyclic (or processorCyclic-on-one-rank) patch
 * that needs its transformation factor (cyclicLduInterfaceField.C:45-62).
 * GAMG (mi_gamg_create on such an addressing; mergeLevels 1): every level carries the agglomerated AMI -- one coarse patch
 * face per distinct local coarse cell in order of first appearance (cyclicAMIGAMGInterface.C:47-165), addresses and
 * weights agglomerated with the face areas of mi_addr_set_ami_face_areas and renormalised (AMIInterpolation::agglomerate,
 * AMIInterpolation.C:279-540; no low-weight correction on coarse levels, :751).  The coarsest level is solved iteratively:
 * mi_gamg_solve wants directSolveCoarsest = 0 there, as the reference's direct coarsest solver casts every interface to a
 * cyclic one (LUscalarMatrix.C:246-251) and cannot run such a case either.                                               */
int mi_addr_set_ami_patch(mi_addr_t addr, int32_t patch, int32_t nbr_patch, const int32_t *start_host_or_null,
                          const int32_t *address_host_or_null, const double *weights_host_or_null,
                          const uint8_t *low_weight_host_or_null);
/* cyclicAMI whose PARTNER patch lives on another rank (round 4; the reference's distributed AMI: singlePatchProc_ == -1,
 * src/meshTools/AMIInterpolation/AMIInterpolation/AMIInterpolation.C:940-1091 calcProcMap + mapDistribute, :281-520 agglomerate
 * with targetMapPtr).  The partner's patch-internal field travels through an ordinary PROCESSOR patch of this addressing, the
 * transport patch: created like any processor patch (no neighbour cells), the SAME size on both ranks (the larger AMI side,
 * padded), its faceCells = this side's AMI faceCells (padding entries: any cell), its interface coefficients zero
 * (mi_matrix_set_interface_coeffs), so the matrix never sees it and every transport of the engine carries it as it is; in
 * mi_matrix_attach_comm it names the partner rank and the partner's transport patch.
 *     pnf[i] = sum_k weights[k] * ( factor * received[ address[k] ] ),  address[k] = face of the transport patch whose
 * received value is meant (finest level: the partner's face number; < n_partner_faces <= size of the transport patch).
 * The interpolation runs after the halo exchange of the operator (such a matrix exchanges first and runs all tiles in one
 * launch).  GAMG: as for a local cyclicAMI patch; the partner side's coarse faces follow from the coarse cells the transport
 * patch receives on every level, no further talk between the ranks.
 * mi_addr_set_ami_patch_remote_multi (round 6): the partner SIDE is itself split over several ranks -- what decomposePar makes of
 * any cyclicAMI patch larger than one rank's share; the reference: calcProcMap (AMIInterpolation.C:940-1091) builds a map that
 * brings every target face a rank's source faces overlap, from whichever rank holds it, into one list numbered rank by rank
 * (AMIInterpolationParallelOps.C).  Here: ONE transport patch per partner piece (as above, each between this rank and the rank
 * that holds the piece), and address[k] numbers the pieces' faces CONCATENATED in the order of transport_patches: piece q =
 * [n_partner_faces[0] + ... + n_partner_faces[q-1], + n_partner_faces[q]), its face j = face j of transport patch q.  A face's
 * weighted sum may take terms from several pieces; they are added in address order, as on one rank.  GAMG: every piece's coarse
 * faces are derived from what ITS transport patch receives, piece by piece (each rank agglomerates its own piece).          */
int mi_addr_set_ami_patch_remote(mi_addr_t addr, int32_t patch, int32_t transport_patch, int32_t n_partner_faces,
                                 const int32_t *start_host, const int32_t *address_host, const double *weights_host,
                                 const uint8_t *low_weight_host_or_null);
int mi_addr_set_ami_patch_remote_multi(mi_addr_t addr, int32_t patch, int32_t n_transports, const int32_t *transport_patches_host,
                                       const int32_t *n_partner_faces_host, const int32_t *start_host, const int32_t *address_host,
                                       const double *weights_host, const uint8_t *low_weight_host_or_null);
/* face areas |Sf| of a cyclicAMI patch's faces (AMIInterpolation::srcMagSf / tgtMagSf), read by the GAMG agglomeration only */
int mi_addr_set_ami_face_areas(mi_addr_t addr, int32_t patch, const double *mag_sf_host);
/* ORDERED addressing -- the caller's numbering is kept: engine order == caller order, mi_addr_cell_perm is the identity and the
 * caller-order operators (mi_amul, mi_tmul, mi_residual, mi_H, mi_sumA, mi_precondition, mi_jacobi_smooth) run straight on the
 * caller's arrays, with no permutation passes (only an n_cells copy of the input where an operator reads coupled-patch
 * neighbour values, which live behind the owned values of an engine vector).  This is how fields "live in engine order for
 * the life of the mesh": renumber the mesh ONCE with the cell order an ordinary mi_addr_create proposes (mi_addr_cell_perm =
 * new-to-old cell map, the `cellMap` renumberMesh's manual method reads; src/renumber/renumberMethods/manualRenumber), keep
 * mi_addr_tile_starts with it, and create the addressing of the renumbered mesh here.  tile_cell_start NULL: consecutive
 * cells are cut into tiles greedily (any numbering with locality, e.g. after a Cuthill-McKee renumberMesh).
 * lduAddressing itself is unchanged by this (lowerAddr/upperAddr of the renumbered mesh, upper-triangular as always).        */
int mi_addr_create_ordered(mi_ctx_t ctx, int32_t n_cells, int32_t n_faces, const int32_t *lower_host, const int32_t *upper_host,
                           int32_t n_patches, const int32_t *patch_sizes, const int32_t *const *patch_face_cells_host,
                           const int32_t *const *patch_nbr_cells_host_or_null, int32_t n_tiles, const int32_t *tile_cell_start_host_or_null,
                           mi_addr_t *out);
/* RENUMBER AT BIND (round 3): the two steps above as ONE call for a mesh that was never renumberMesh-ed -- the shim adopts the
 * engine's cell order for the life of the mesh.  Builds the clustered layout of the mesh as given, renumbers the cells into the
 * engine order and the faces into the upper-triangular order of that numbering (faces whose owner and neighbour swap are
 * flipped), and returns the ORDERED addressing of the renumbered mesh together with what polyMesh::renumber / mapPolyMesh need:
 *   cell_new_to_old [n_cells]   new cell i = old cell map[i]                 (the cellMap of renumberMesh's manual method)
 *   face_new_to_old [n_faces]   new internal face f = old face map[f]
 *   face_flipped    [n_faces]   1: its owner / neighbour (hence upper / lower, and the sign of a flux) swapped
 *   lower_out / upper_out       the addressing of the renumbered mesh (any of the five outputs may be NULL)
 * Fields are permuted ONCE when they are read (new[i] = old[cell_new_to_old[i]]) and back when they are written; from then on
 * every operator of this ABI runs on the caller's arrays without permutation passes (mi_amul 0.62-0.70 of the HBM roofline
 * instead of 0.52 through the permuting addressing).  Patch face cells (and local partner cells) are renumbered inside; their
 * face order is kept.  mi_layout_adopt_host: the host part alone (no device; tests).                                           */
int mi_addr_create_adopted(mi_ctx_t ctx, int32_t n_cells, int32_t n_faces, const int32_t *lower_host, const int32_t *upper_host,
                           int32_t n_patches, const int32_t *patch_sizes, const int32_t *const *patch_face_cells_host,
                           const int32_t *const *patch_nbr_cells_host_or_null, int32_t *cell_new_to_old_out, int32_t *face_new_to_old_out,
                           uint8_t *face_flipped_out, int32_t *lower_out, int32_t *upper_out, mi_addr_t *out);
int mi_layout_adopt_host(int32_t n_cells, int32_t n_faces, const int32_t *lower_host, const int32_t *upper_host, int32_t n_patches,
                         const int32_t *patch_sizes, const int32_t *const *patch_face_cells_host, const int32_t *const *patch_nbr_cells_host_or_null,
                         int32_t *cell_new_to_old_out, int32_t *face_new_to_old_out, uint8_t *face_flipped_out, int32_t *lower_out,
                         int32_t *upper_out, int32_t *n_tiles_out);
/* the tiles of a layout as ranges of ENGINE cells: n_tiles + 1 offsets (mi_addr_n_tiles) */
int mi_addr_tile_starts(mi_addr_t addr, int32_t *tile_cell_start_out_host);
/* 1 when engine order == caller order (mi_addr_create_ordered, or a numbering that happened to be tile-contiguous) */
int mi_addr_is_ordered(mi_addr_t addr);
int mi_addr_destroy(mi_addr_t addr);
int32_t mi_addr_n_cells(mi_addr_t addr);
int32_t mi_addr_n_faces(mi_addr_t addr);
int32_t mi_addr_n_tiles(mi_addr_t addr);
/* number of halo ("external") cells appended after the n_cells owned cells in
 * engine-order vectors: sum of patch sizes                                       */
int32_t mi_addr_n_ext(mi_addr_t addr);
/* engine->caller cell permutation (n_cells ints) */
int mi_addr_cell_perm(mi_addr_t addr, int32_t *engine_to_caller_host);
/* layout statistics: [0]=tiles [1]=slots total [2]=entries total (padded)
 * [3]=halo entries total [4]=max cells/tile [5]=max slots/tile [6]=max halo/tile
 * [7]=LDS bytes per workgroup (symmetric)                                        */
int mi_addr_stats(mi_addr_t addr, int64_t stats[8]);

/* ---- matrix (replaces lduMatrix storage + lowerSort()/upperSort() caches,
 *      lduMatrix/lduMatrix.C:221-472; K22 calcSortCoeffs) ---- */
int mi_matrix_create(mi_addr_t addr, mi_matrix_t *out);
mi_addr_t mi_matrix_addr(mi_matrix_t m);   /* the addressing a matrix was created on (borrowed handle) */
int mi_matrix_destroy(mi_matrix_t m);
/* Bind new coefficient values ("coefficients changed" epoch; the reference
 * invalidates lowerSortPtr_ in every mutator, lduMatrix.C:235,266).
 * lower_dev == NULL => symmetric (lower aliases upper, lduMatrix.C:328-345).
 * Copies into the engine's tiled layout; the caller arrays are not retained.    */
int mi_matrix_set_coeffs(mi_matrix_t m, const double *diag_dev, const double *upper_dev,
                         const double *lower_dev);
/* interfaceBouCoeffs / interfaceIntCoeffs of coupled patch p (device pointers,
 * patch_sizes[p] values each; int_coeffs_dev may be NULL for symmetric matrices). */
int mi_matrix_set_interface_coeffs(mi_matrix_t m, int32_t patch, const double *bou_coeffs_dev,
                                   const double *int_coeffs_dev);
/* transformCoupleField of a coupled patch (cyclicLduInterfaceField.C:45-62, processorGAMGInterfaceField.C:213,230,
 * cyclicAMILduInterfaceField.C:45-62): the neighbour values are multiplied by
 *     factor = pow(diag(forwardT).component(cmpt), rank)
 * before they enter result -= coeffs*pnf -- 1 for scalars (rank 0) and untransformed patches, the caller's number for the
 * component solves of a vector across a rotational cyclic / processorCyclic.  Processor patches: applied to the received
 * values; cyclicAMI and one-to-one patches declared with mi_addr_set_ami_patch: applied before the interpolation.  A plain
 * cyclic patch (created with neighbour cells) only accepts 1.  Held per matrix: Ux, Uy, Uz solves set their own.
 * Decomposed case: the FIRST factor of a matrix is set before mi_matrix_attach_comm -- the ranks agree there whether the case
 * has transformed patches anywhere (it decides the solver pipeline of EVERY rank); afterwards the values may change freely,
 * but a first factor != 1 on a case that was attached without any is refused (MI_ERR_STATE).                              */
int mi_matrix_set_patch_transform(mi_matrix_t m, int32_t patch, double factor);

/* ---- halo (replaces init/updateMatrixInterfaces + processorFvPatchField
 *      gather/scatter, lduMatrixUpdateMatrixInterfaces.C:30-276,
 *      src/finiteVolume/fields/fvPatchFields/constraint/processor/processorFvPatchScalarField.C:36-170) ----
 * pack: send_dev[patch_offset[p]+i] = x[patch_face_cells[p][i]] for all patches
 *       (K5 fvPatchTemplates.C:49-63), x in engine order.
 * The received neighbour values are written by the caller (RCCL recv / peer copy)
 * straight into the ext region x_engine[n_cells .. n_cells+n_ext).              */
int mi_halo_pack_engine(mi_addr_t addr, const double *x_engine_dev, double *send_dev);
int mi_addr_patch_offsets(mi_addr_t addr, int32_t *offsets_host /* n_patches+1 */);

/* ---- vector layout helpers ---- */
int mi_vec_to_engine(mi_addr_t addr, const double *x_caller_dev, double *x_engine_dev);
int mi_vec_from_engine(mi_addr_t addr, const double *x_engine_dev, double *x_caller_dev);

/* ---- SpMV family, caller order (lduMatrix::Amul/Tmul lduMatrixATmul.C:183-342,
 *      sumA :345-395, residual :397-496, H1 :533-554, H lduMatrixOperations.C:130-154,
 *      faceH lduMatrixTemplates.C:110-148) ----
 * Interface (halo) terms use ext values previously placed with mi_matrix_set_ext (single
 * process: none).  On an addressing that permutes (mi_addr_create) these run the tile pass straight on the caller's arrays:
 * x is gathered through the cell permutation while a tile is staged, y (and source) are addressed through it in the row
 * loop -- no separate permutation passes (mi_amul 173 us against 146 in engine order on the 216^3 box); with ordered
 * addressing there is no permutation at all.                                                                            */
int mi_amul(mi_matrix_t m, const double *psi_dev, double *Apsi_dev);
int mi_tmul(mi_matrix_t m, const double *psi_dev, double *Tpsi_dev);
int mi_sumA(mi_matrix_t m, double *sumA_dev);
int mi_residual(mi_matrix_t m, const double *psi_dev, const double *source_dev, double *rA_dev);
int mi_H(mi_matrix_t m, const double *psi_dev, double *H_dev);
int mi_H1(mi_matrix_t m, double *H1_dev);
int mi_faceH(mi_matrix_t m, const double *psi_dev, double *faceH_dev);
/* neighbour values for the coupled patches, caller patch order, n_ext doubles */
int mi_matrix_set_ext(mi_matrix_t m, const double *ext_values_dev);

/* ---- SpMV family, engine order (inner loops of the solvers; what bench.py times) ----
 * vectors are n_cells + n_ext long; which: 0 = all tiles, 1 = interior tiles only
 * (no ext reference), 2 = boundary tiles only (after the halo has arrived).       */
int mi_amul_engine(mi_matrix_t m, const double *psi_e, double *Apsi_e, int which);
int mi_tmul_engine(mi_matrix_t m, const double *psi_e, double *Tpsi_e, int which);

/* ---- engine-order primitives for a caller that keeps its own solver loop but lets its work vectors live in engine order for
 *      the duration of a solve (mi_vec_to_engine once, mi_vec_from_engine once): the same operators without the permutation
 *      passes of the caller-order entry points (measured on the 216^3 box: the reference's PCG::solve runs at 682 us/iteration
 *      on the caller-order primitives).  mi_amul_engine / mi_tmul_engine with which = 0 exchange the halo
 *      themselves when a communicator is attached; reductions (mi_sum*) do not care about the order.                      */
int mi_precondition_engine(mi_matrix_t m, int kind, int transpose, const double *rA_e, double *wA_e);
int mi_residual_engine(mi_matrix_t m, const double *psi_e, const double *source_e, double *rA_e);
int mi_jacobi_smooth_engine(mi_matrix_t m, double omega, double *psi_e, const double *source_e, int32_t n_sweeps);
int mi_norm_factor_engine(mi_matrix_t m, const double *psi_e, const double *source_e, const double *Apsi_e, double *out_host);

/* ---- preconditioners (lduMatrix::preconditioner::precondition / preconditionT,
 *      diagonalPreconditioner.C:45-89, AINVPreconditioner.C:49-120) ---- */
int mi_precondition(mi_matrix_t m, int kind, int transpose, const double *rA_dev, double *wA_dev);

/* ---- smoother (lduMatrix::smoother::smooth; JacobiSmoother.C:39-148; the
 *      reference's "GaussSeidel" is this Jacobi with omega 0.9) ---- */
int mi_jacobi_smooth(mi_matrix_t m, double omega, double *psi_dev, const double *source_dev,
                     int32_t n_sweeps);

/* ---- field reductions (gpuFieldCommonFunctions.C:351-367,420-440,492-511) ----
 * deterministic fixed-tree device reductions; result returned to the host.       */
int mi_sum(mi_ctx_t ctx, const double *a_dev, int64_t n, double *out_host);
int mi_sum_prod(mi_ctx_t ctx, const double *a_dev, const double *b_dev, int64_t n, double *out_host);
int mi_sum_mag(mi_ctx_t ctx, const double *a_dev, int64_t n, double *out_host);
/* lduMatrix::solver::normFactor (lduMatrixSolver.C:182-236):
 *   sum(|Apsi - avg(psi)*sumA| + |source - avg(psi)*sumA|) + small_ ; caller order; the same
 * bits as perf.normFactor of the whole solvers; global when a communicator is attached.  */
int mi_norm_factor(mi_matrix_t m, const double *psi_dev, const double *source_dev,
                   const double *Apsi_dev, double *out_host);

/* ---- whole solvers (lduMatrix::solver::solve; PCG.C:68-208, PBiCG.C:67-246,
 *      PBiCGStab.C:67-300, smoothSolver.C:77-196).  psi in/out, caller order.
 * residual_history_host (may be NULL) receives the normalised residual after
 * every iteration, [0] = initial; history_len entries at most.                    */
int mi_pcg_solve(mi_matrix_t m, double *psi_dev, const double *source_dev,
                 const mi_solver_controls *controls, int precond,
                 mi_solver_perf *perf_out, double *residual_history_host, int32_t history_len);
/* The same PCG as a session, for callers that want to overlap or time it
 * themselves (bench.py): begin = PCG.C:85-121 (A.psi, rA, normFactor, initial
 * residual); iterate = enqueue n_iters bodies of the do-loop (PCG.C:133-204) on
 * the context's stream WITHOUT synchronising -- bodies past convergence are
 * device-side no-ops, so the result equals the reference loop; end = fetch psi,
 * solverPerformance and the residual history.  If amul_ms_sum != NULL, every
 * Amul launch of this call is bracketed by HIP events on the stream and the sum
 * of their durations is returned (this synchronises).
 * One session per context: between mi_pcg_begin and mi_pcg_end the session owns the context's solver scratch
 * (device-side solver state, reduction partials) and the matrix's work vectors; until mi_pcg_end the reductions
 * (mi_sum*), mi_norm_factor*, every mi_*_solve and a mi_pcg_begin on another matrix of the same context return
 * MI_ERR_STATE instead of corrupting the running solve.  The operators (mi_amul ... mi_precondition) stay usable. */
int mi_pcg_begin(mi_matrix_t m, const double *psi0_dev, const double *source_dev,
                 const mi_solver_controls *controls, int precond, int32_t history_len);
int mi_pcg_iterate(mi_matrix_t m, int32_t n_iters, float *amul_ms_sum);
/* same, events around every event_stride-th Amul only; *amul_ms_sum = mean sampled duration x n_iters */
int mi_pcg_iterate_sampled(mi_matrix_t m, int32_t n_iters, int32_t event_stride, float *amul_ms_sum);
int mi_pcg_end(mi_matrix_t m, double *psi_out_dev, mi_solver_perf *perf_out,
               double *residual_history_host, int32_t history_len);

/* ---- GAMG (solvers/GAMG/GAMGSolver.C:47-249, GAMGSolverSolve.C:59-619, GAMGSolverScale.C:59-171,
 *      GAMGSolverAgglomerateMatrix.C:37-321, GAMGAgglomerations/.../pairGAMGAgglomerate.C:31-313,
 *      GAMGAgglomerateLduAddressing.C:245-461, GAMGAgglomerationTemplates.C:35-308) ----
 * mi_gamg_create builds the pair-agglomeration hierarchy once per mesh (host, cached like the
 * reference's GAMGAgglomeration MeshObject) and one engine layout per level.
 *   face_weights_host: [n_faces] agglomeration weights -- faceAreaPair passes
 *     |Sf/sqrt|Sf| o (1,1.01,1.02)| (src/finiteVolume/.../faceAreaPairGAMGAgglomeration.C:54-81),
 *     algebraicPair passes |upper|.
 *   n_cells_in_coarsest_level: the mandatory fvSolution key (GAMGAgglomeration.C:96-99).
 *   merge_levels: fvSolution's mergeLevels (pairGAMGAgglomerate.C:110-117): every level folds that many pair steps
 *     (GAMGAgglomeration::combineLevels, GAMGAgglomerateLduAddressing.C:606-760); 1 = one pair step per level.
 *   forward_init: the reference's process-wide static sweep direction (pairGAMGAgglomeration.C:33),
 *     true in a fresh process; mi_gamg_forward_out returns its value after the build.
 * mi_gamg_solve agglomerates the level matrices from the matrix's current coefficients (the
 * reference does this on every solver construction, GAMGSolver.C:88-97), then runs V-cycles.
 * Smoother: Jacobi omega (the reference's "GaussSeidel", JacobiSmoother.C:34-36).  The coarsest
 * level is solved directly (directSolveCoarsest, GAMGSolver.C:144-172) with a dense inverse kept
 * on the device.  scaleCorrection < 0 selects the reference default (= symmetric).            */
typedef struct {
    double tolerance, relTol;
    int32_t maxIter, minIter;
    int32_t nPreSweeps, preSweepsLevelMultiplier, maxPreSweeps;     /* 0, 1, 4 */
    int32_t nPostSweeps, postSweepsLevelMultiplier, maxPostSweeps;  /* 2, 1, 4 */
    int32_t nFinestSweeps, scaleCorrection;                         /* 2, -1   */
    double omega;                                                   /* 0.9     */
    int32_t directSolveCoarsest;  /* 1 (GAMGSolver.C:77): dense solve on the device; 0: ICCG / BICCG (PCG / PBiCG + AINV, */
    int32_t reserved;             /* GAMG's tolerance and relTol, zero start) on the coarsest level, GAMGSolverSolve.C:572-613 */
} mi_gamg_controls;
int mi_gamg_create(mi_addr_t fine_addr, const double *face_weights_host, int32_t n_cells_in_coarsest_level,
                   int32_t merge_levels, int forward_init, mi_gamg_t *out);
/* Same for a matrix with coupled patches.  Cyclic (local) patches need no communicator (pass NULLs; mi_gamg_create
 * does that).  Processor patches (processorGAMGInterface, solvers/GAMG/interfaces/processorGAMGInterface/
 * processorGAMGInterface.C:54-245; GAMGAgglomerateLduAddressing.C:464-520) need the communicators the matrix is
 * attached to: the ranks agree on when to stop (GAMGAgglomeration.C:72-81), exchange the restrict addressing of the
 * patch cells on every level, every level matrix exchanges its own halo, the scale factors are all-reduced
 * (GAMGSolverScale.C:104-107) and the coarsest level is the GLOBAL system (LUscalarMatrix.C:57-150) whose dense
 * inverse every rank holds the rows of.  All ranks call create/solve together.                                  */
/* agglomerator "dummy" (GAMGAgglomerations/dummyAgglomeration/dummyAgglomeration.C:45-90; the reference's debugging
 * agglomerator): n_levels levels whose restrict addressing is the identity -- every level is the fine mesh again, the
 * coarsest (= fine) system is solved directly, so it is for small meshes (<= 4096 cells with directSolveCoarsest). */
int mi_gamg_create_dummy(mi_addr_t fine_addr, int32_t n_levels, mi_gamg_t *out);
int mi_gamg_create_coupled(mi_addr_t fine_addr, const double *face_weights_host,
                           int32_t n_cells_in_coarsest_level, int32_t merge_levels, int forward_init, mi_comm_t reduce_or_null,
                           mi_comm_t halo_or_null, const int32_t *patch_rank,
                           const int32_t *patch_nbr_patch_or_null, mi_gamg_t *out);
int mi_gamg_destroy(mi_gamg_t g);
int32_t mi_gamg_n_levels(mi_gamg_t g);
int mi_gamg_forward_out(mi_gamg_t g);
/* out4 = {n_fine_cells, n_fine_faces, n_coarse_cells, n_coarse_faces} of coarse level `level` */
int mi_gamg_level_sizes(mi_gamg_t g, int32_t level, int32_t out4[4]);
int mi_gamg_solve(mi_gamg_t g, mi_matrix_t m, double *psi_dev, const double *source_dev,
                  const mi_gamg_controls *controls, mi_solver_perf *perf_out,
                  double *residual_history_host, int32_t history_len);
/* level operators in the caller order of each level (restrictField / prolongField,
 * GAMGAgglomerationTemplates.C:35-153,273-308) and the agglomerated level coefficients
 * (GAMGSolverAgglomerateMatrix.C:218-317); used by the parity tests                          */
int mi_gamg_restrict(mi_gamg_t g, int32_t level, const double *fine_dev, double *coarse_dev);
int mi_gamg_prolong(mi_gamg_t g, int32_t level, const double *coarse_dev, double *fine_dev);
/* The pieces of the V-cycle as operators, for a caller that keeps the reference's GAMGSolverSolve.C and swaps only the
 * primitives (tests/ref_dropin runs exactly that):
 *   mi_gamg_update        agglomerateMatrix of every level from the matrix's current coefficients + the coarsest direct
 *                         solver (the GAMGSolver constructor, GAMGSolver.C:88-172); mi_gamg_solve does it itself.
 *   mi_gamg_level_matrix  matrixLevels_[level] as a BORROWED handle (vectors in the hierarchy's coarse-cell numbering):
 *                         mi_amul / mi_jacobi_smooth / mi_residual ... work on it.
 *   mi_gamg_scale         GAMGSolver::scale (GAMGSolverScale.C:59-171) on the fine or a level matrix.
 *   mi_gamg_solve_coarsest  solveCoarsestLevel with directSolveCoarsest (GAMGSolverSolve.C:552-570), on the device.    */
int mi_gamg_update(mi_gamg_t g, mi_matrix_t m);
int mi_gamg_level_matrix(mi_gamg_t g, int32_t level, mi_matrix_t *level_matrix_out);
int mi_gamg_scale(mi_matrix_t m, double *field_dev, double *Acf_dev, const double *source_dev);
int mi_gamg_solve_coarsest(mi_gamg_t g, const double *source_dev, double *corr_dev);
int mi_gamg_level_coeffs(mi_gamg_t g, mi_matrix_t m, int32_t level, double *diag_out_dev,
                         double *upper_out_dev, double *lower_out_dev);

/* ---- fvMatrix assembly sweeps, scalar fields, caller-order device arrays ----
 * (src/finiteVolume/finiteVolume/convectionSchemes/gaussConvectionScheme/gaussConvectionScheme.C:74-115,
 *  .../laplacianSchemes/gaussLaplacianScheme/gaussLaplacianScheme.C:44-88,
 *  src/OpenFOAM/matrices/lduMatrix/lduMatrix/lduMatrixOperations.C:36-106,
 *  src/finiteVolume/fvMatrices/fvMatrix/fvMatrix.C:38-124,208-349,1087-1345,
 *  src/finiteVolume/finiteVolume/fvc/fvcSurfaceIntegrate.C:40-96,
 *  src/finiteVolume/interpolation/surfaceInterpolation/surfaceInterpolationScheme/surfaceInterpolationScheme.C:337-352)
 * Every scheme is one streaming face pass + one row pass whose own-face reads are staged through LDS
 * (the reference: 3-6 Thrust passes + temporaries).  They need the caller's faces owner-sorted
 * (OpenFOAM's upper-triangular order); face fields 16-byte aligned.
 * mi_row_face_op kind: 0 sumDiag, 1 negSumDiag, 2 sumMagOffDiag; lower_dev NULL => symmetric.
 * mi_patch_*: a boundary patch = its faceCells; mi_patch_add applies pf per unique cell in ascending
 * patch-face order (addToInternalField, K26): fn 0 add, 1 subtract, 2 add magnitudes.          */
int mi_row_face_op(mi_addr_t addr, int kind, const double *lower_dev, const double *upper_dev, double *inout_dev);
int mi_fvm_laplacian(mi_addr_t addr, const double *delta_coeffs_dev, const double *gamma_magsf_dev,
                     double *upper_out_dev, double *diag_out_dev);
int mi_fvm_div(mi_addr_t addr, const double *weights_dev, const double *face_flux_dev,
               double *lower_out_dev, double *upper_out_dev, double *diag_out_dev);
int mi_surface_integrate(mi_addr_t addr, const double *ssf_dev, const double *vol_dev_or_null, double *ivf_dev);
int mi_face_interpolate(mi_addr_t addr, const double *lambda_dev, const double *phi_dev, double *sf_dev);
/* ---- scheme front-end either side of the matrix ----
 * mi_fvm_ddt_euler: EulerDdtScheme fvmDdt (src/finiteVolume/finiteVolume/ddtSchemes/EulerDdtScheme/EulerDdtScheme.C):
 *   diag = rDeltaT*rho*V, source = rDeltaT*rho*psiOld*V (rho = 1 for the plain form).
 * mi_upwind_weights: pos(faceFlux) (limitedSchemes/upwind, limitedSurfaceInterpolationScheme.C:177-187).
 * mi_limited_linear_weights: limitedLinear(k) weights of a scalar field in one face pass -- r of
 *   limitedSchemes/LimitedScheme/NVDTVD.H, limiter of limitedSchemes/limitedLinear/limitedLinear.H:79-97, weights of
 *   limitedSurfaceInterpolationScheme.C:177-187; grad = cell gradient of phi (component arrays), c = cell centres;
 *   limiter_out may be NULL.  The result feeds mi_fvm_div / mi_face_interpolate.
 * mi_gauss_grad: fvc::grad with Gauss integration (finiteVolume/gradSchemes/gaussGrad/gaussGrad.C:27-90,143-250):
 *   g = (sum_own Sf*ssf - sum_nei Sf*ssf)/V over the internal faces, component arrays; boundary faces are added with
 *   mi_patch_add per component (vol = NULL here, then divide) exactly as the reference adds them per patch.
 * mi_vec_axpby: out = a*x + b*y (fvMatrix operator+=, -=, *= on coefficient arrays; out may alias x or y).        */
int mi_fvm_ddt_euler(mi_ctx_t ctx, int64_t n, double r_delta_t, double rho, const double *vol_dev,
                     const double *psi_old_dev, double *diag_out_dev, double *source_out_dev);
int mi_upwind_weights(mi_ctx_t ctx, int64_t n_faces, const double *face_flux_dev, double *weights_out_dev);
int mi_limited_linear_weights(mi_addr_t addr, double k, const double *cd_weights_dev, const double *face_flux_dev,
                              const double *phi_dev, const double *gradx_dev, const double *grady_dev,
                              const double *gradz_dev, const double *cx_dev, const double *cy_dev,
                              const double *cz_dev, double *weights_out_dev, double *limiter_out_dev_or_null);
int mi_gauss_grad(mi_addr_t addr, const double *sfx_dev, const double *sfy_dev, const double *sfz_dev,
                  const double *ssf_dev, const double *vol_dev_or_null, double *gx_dev, double *gy_dev, double *gz_dev);
int mi_vec_axpby(mi_ctx_t ctx, int64_t n, double a, const double *x_dev, double b, const double *y_dev, double *out_dev);
/* ---- the compressible operators of rhoPimpleFoam (BASELINE config 5; round 5) ----
 * mi_fvm_ddt_euler_rho: fvm::ddt(rho, vf) with a density FIELD -- EulerDdtScheme<Type>::fvmDdt(const volScalarField& rho, vf),
 *   src/finiteVolume/finiteVolume/ddtSchemes/EulerDdtScheme/EulerDdtScheme.C:403-440:
 *     diag = rDeltaT*rho*V, source = rDeltaT*rho.oldTime()*vf.oldTime()*V   (rhoPimpleFoam/UEqn.H:5, EEqn.H:6; with psi in rho's
 *   place fvm::ddt(psi, p), pEqn.H:38,62).
 * mi_fvm_su / mi_fvm_sp / mi_fvm_susp: src/finiteVolume/finiteVolume/fvm/fvmSup.C:34-54 (source -= V*su), :100-170 (diag += V*sp;
 *   sp_dev NULL => the dimensionedScalar form with sp_value), :190-214 (diag += V*max(susp,0); source -= V*min(susp,0)*vf) -- in
 *   place on the matrix's diagonal / source (fvOptions(rho, U), the explicit terms of EEqn.H).
 * mi_flux_div: phi = Sf & linear-interpolate([cell_scale *] V) [+ add_a [* add_b]] on the internal faces AND
 *   div = fvc::surfaceIntegrate(phi) [/ vol] in one call -- pEqn.H:49-71's
 *   phiHbyA = (fvc::interpolate(rho*HbyA) & mesh.Sf()) + rhorAUf*fvc::ddtCorr(rho, U, phi) followed by fvc::div(phiHbyA)
 *   (finiteVolume/fvc/fvcSurfaceIntegrate.C:40-96), which the reference computes with seven field passes: here one face pass
 *   (rho*HbyA, the three interpolates, the dot product, the product and the sum at once) + the row sum.  lambda = the
 *   interpolation weights (surfaceInterpolationScheme.C:275-280); boundary faces: mi_patch_add on div as the reference adds them.
 * mi_ddt_phi_corr: fvc::ddtCorr(rho, U, phi) on the internal faces, EulerDdtScheme<Type>::fvcDdtPhiCorr(rho, U, phi)
 *   (EulerDdtScheme.C:663-720, first branch; rho_old NULL: fvcDdtPhiCorr(U, phi) :523-551) with ddtScheme<Type>::fvcDdtPhiCoeff
 *   (ddtSchemes/ddtScheme/ddtScheme.C:139-174) in one face pass:
 *     phiCorr = phi0 - (Sf & interpolate(rho0*U0)); out = (1 - min(|phiCorr|/(|phi0| + SMALL), 1))*rDeltaT*phiCorr.          */
int mi_fvm_ddt_euler_rho(mi_ctx_t ctx, int64_t n, double r_delta_t, const double *rho_dev, const double *rho_old_dev,
                         const double *vol_dev, const double *psi_old_dev, double *diag_out_dev, double *source_out_dev);
int mi_fvm_su(mi_ctx_t ctx, int64_t n, const double *vol_dev, const double *su_dev, double *source_inout_dev);
int mi_fvm_sp(mi_ctx_t ctx, int64_t n, const double *vol_dev, const double *sp_dev_or_null, double sp_value, double *diag_inout_dev);
int mi_fvm_susp(mi_ctx_t ctx, int64_t n, const double *vol_dev, const double *susp_dev, const double *vf_dev,
                double *diag_inout_dev, double *source_inout_dev);
int mi_flux_div(mi_addr_t addr, const double *lambda_dev, const double *sfx_dev, const double *sfy_dev, const double *sfz_dev,
                const double *vx_dev, const double *vy_dev, const double *vz_dev, const double *cell_scale_dev_or_null,
                const double *add_a_dev_or_null, const double *add_b_dev_or_null, double *phi_out_dev,
                const double *vol_dev_or_null, double *div_out_dev);
int mi_ddt_phi_corr(mi_addr_t addr, double r_delta_t, const double *lambda_dev, const double *sfx_dev, const double *sfy_dev,
                    const double *sfz_dev, const double *ux_old_dev, const double *uy_old_dev, const double *uz_old_dev,
                    const double *rho_old_dev_or_null, const double *phi_old_dev, double *out_dev);
/* ---- fused matrix assembly (round 6) ----
 * mi_fvm_assemble: the matrix of   [fvm::ddt(rho, vf)] + [fvm::div(flux, vf)] - [fvm::laplacian(gamma, vf)] [+- fvm::Sp(sp, vf)] [+- su ...]
 * written ONCE by one row pass -- the expression every transport equation of simpleFoam / pisoFoam / rhoPimpleFoam has
 * (UEqn.H, EEqn.H, pEqn.H) and that the reference evaluates scheme by scheme (gaussConvectionScheme.C:74-115, gaussLaplacianScheme.C:44-88,
 * EulerDdtScheme.C:371-440, fvmSup.C:34-214) and then combines array by array with fvMatrix::operator+ / - / == (fvMatrix.C:1693-2030 ->
 * lduMatrix::operator+= / -=, lduMatrixOperations.C:235-396).  The outputs equal that sequence BIT FOR BIT, because every intermediate is
 * rounded where the sequence rounds it:
 *   per face    lB = -w*flux, uB = lB + flux (w = div_weights, or pos(flux) when div_weights_dev is NULL: upwind);  uL = deltaCoeffs*gammaMagSf
 *               lower = lB - uL, upper = uB - uL   (a term that is absent drops out of the expression; no convection: upper = -uL, the
 *               matrix stays symmetric and lower_out_dev may be NULL)
 *   per cell    sumB = negSumDiag of (lB, uB), sumL = negSumDiag of (uL, uL)   (lduMatrixOperations.C:61-83: own faces ascending, then losort order)
 *               diag = (((rDeltaT*rho)*V + sumB) - sumL) [+- V*sp]
 *               source[r] = ((rDeltaT*rho_old)*psi_old[r])*V, then  -= V*su[k][r] (su_sign[k] > 0: the term stands on the left, `+ su`)
 *               or += V*su[k][r] (su_sign[k] < 0: `== su`), k ascending      (fvMatrix.C:1850-1905 operator+/-(fvMatrix, volField), :1741 operator==)
 * rho_dev NULL: the constant rho_value (fvm::ddt(vf): 1).  su_dev holds n_su * n_rhs device pointers, su_dev[k * n_rhs + r]; su_sign
 * n_su host doubles.  n_rhs <= 4 right-hand sides share the coefficients (the components of a vector equation, fvMatrixSolve.C:103-225),
 * n_su <= 4.  sum_mag_off_diag_out_dev (may be NULL): lduMatrix::sumMagOffDiag of the FINAL coefficients from the same pass -- what
 * fvMatrix::relax needs (mi_relax_multi takes it).  Boundary coefficients stay with the caller (mi_patch_add), as in the reference.
 * Coefficient outputs must not alias inputs (faces cut by a block boundary are recomputed from the inputs).                          */
typedef struct mi_fvm_terms {
    int32_t ddt;                                   /* != 0: Euler time derivative present */
    double r_delta_t, rho_value;
    const double *rho_dev, *rho_old_dev;           /* both NULL (constant rho_value) or both given */
    const double *vol_dev;                         /* needed with ddt, sp or su */
    const double *div_flux_dev, *div_weights_dev;  /* flux NULL: no convection term */
    const double *lap_delta_coeffs_dev, *lap_gamma_magsf_dev;   /* delta NULL: no diffusion term */
    const double *sp_dev; double sp_sign;          /* sp NULL: none; sign > 0: diag += V*sp, < 0: diag -= V*sp */
    int32_t n_rhs; const double *const *psi_old_dev;   /* n_rhs old-time fields (read when ddt != 0) */
    int32_t n_su; const double *const *su_dev; const double *su_sign;
} mi_fvm_terms;
int mi_fvm_assemble(mi_addr_t addr, const mi_fvm_terms *terms, double *lower_out_dev_or_null, double *upper_out_dev, double *diag_out_dev,
                    double *const *source_out_dev, double *sum_mag_off_diag_out_dev_or_null);
/* fvMatrix::setReference (src/finiteVolume/fvMatrices/fvMatrix/fvMatrix.C:964-981; icoFoam.C:89, simpleFoam/pEqn.H:21: the pressure level of a
 * closed domain): source[celli] += diag[celli]*value; diag[celli] += diag[celli].  celli < 0 (the rank does not hold the cell): no-op. */
int mi_fvm_set_reference(mi_addr_t addr, int32_t celli, double value, double *diag_dev, double *source_dev);
/* fvMatrix::setValues (fvMatrix.C:454-656, functors :352-452; fvOptions constraints, wall-function cells): psi[cell] = value,
 * source[cell] = value*diag[cell]; the source of every OTHER row is corrected for its set neighbours; the coefficients that couple a
 * set row to its neighbours and the boundary coefficients of its patch faces are cleared.
 * upstream_semantics == 0, the REFERENCE's behaviour (it deviates from upstream OpenFOAM, SURVEY appendix B style): the correction uses
 * the transposed coefficient (lower[face] for an own face, upper[face] for a neighbour-side face), upper is cleared where the OWNER is
 * set and lower where the NEIGHBOUR is set, and a symmetric matrix comes out asymmetric (the reference's non-const lower() copies
 * upper) -- hence lower_out_dev is always written (lower_in_dev NULL: symmetric input).  != 0: upstream OpenFOAM's
 * (source[nei] -= lower*value, source[own] -= upper*value, both triangles cleared at every face of a set cell).  Outputs may alias inputs. */
int mi_fvm_set_values(mi_addr_t addr, int32_t n_set, const int32_t *cell_labels_dev, const double *values_dev, int32_t upstream_semantics,
                      double *psi_dev, const double *diag_dev, double *source_dev, const double *upper_in_dev, const double *lower_in_dev_or_null,
                      double *upper_out_dev, double *lower_out_dev, int32_t n_patches, const mi_patch_t *patches,
                      double *const *internal_coeffs_dev, double *const *boundary_coeffs_dev);
/* fvc::div(faceFlux, vf) -- gaussConvectionScheme<Type>::fvcDiv (src/finiteVolume/finiteVolume/convectionSchemes/gaussConvectionScheme/
 * gaussConvectionScheme.C:117-140): fvc::surfaceIntegrate(faceFlux*interpolate(faceFlux, vf)) on the internal faces; EEqn.H:9's fvc::div(phi, K).
 * weights NULL: upwind (pos(faceFlux)).  face_out receives faceFlux*interpolate (gaussConvectionScheme::flux, :62-70); div_out the row sums
 * [/ vol].  One face pass + one row pass (the reference: weights, interpolate, product, surfaceIntegrate = four field passes); boundary
 * faces: mi_patch_add on div_out as the reference adds them.                                                                           */
int mi_fvc_div(mi_addr_t addr, const double *face_flux_dev, const double *weights_dev_or_null, const double *vf_dev,
               const double *vol_dev_or_null, double *face_out_dev, double *div_out_dev);
/* out = x / y element-wise (fvMatrix::A = D/V, fvMatrix::H /= V; fvMatrix.C:1424-1506); out may alias x */
int mi_vec_div(mi_ctx_t ctx, int64_t n, const double *x_dev, const double *y_dev, double *out_dev);
/* Non-orthogonal correction of fvm::laplacian (row a22): gaussLaplacianScheme<Type, scalar>::fvmLaplacian with a `corrected`
 * snGrad scheme (laplacianSchemes/gaussLaplacianScheme/gaussLaplacianSchemes.C:64-90):
 *     source -= V * fvc::div( gammaMagSf * snGradScheme.correction(vf) ),
 * correction(vf) = nonOrthCorrectionVectors & linear.interpolate(grad(vf))  (snGradSchemes/correctedSnGrad/correctedSnGrad.C:45-65,
 * vectors of surfaceInterpolation.C:498-630: Sf/|Sf| - delta*nonOrthDeltaCoeffs on internal faces and coupled patches, zero on the others).
 * mi_sngrad_correction_flux is the face pass over the internal faces (the reference: an interpolated gradient field, a dot-product
 * field and a product field); mi_patch_sngrad_correction_flux the same on one COUPLED patch, with the patchNeighbourField of the gradient
 * (mi_matrix_patch_neighbour_field per component).  Then mi_surface_integrate(flux, NULL, div) + mi_patch_add(patch flux) + mi_vec_div(V)
 * is fvc::div and mi_vec_submul(V, div, source) the `source -= V*div`.  gamma_magsf NULL: the bare correction (snGrad, fvc::laplacian). */
int mi_sngrad_correction_flux(mi_addr_t addr, const double *corr_vec_x_dev, const double *corr_vec_y_dev, const double *corr_vec_z_dev,
                              const double *weights_dev, const double *grad_x_dev, const double *grad_y_dev, const double *grad_z_dev,
                              const double *gamma_magsf_dev_or_null, double *flux_out_dev);
int mi_patch_sngrad_correction_flux(mi_patch_t patch, const double *corr_vec_x_dev, const double *corr_vec_y_dev, const double *corr_vec_z_dev,
                                    const double *patch_weights_dev, const double *grad_x_dev, const double *grad_y_dev, const double *grad_z_dev,
                                    const double *nbr_grad_x_dev, const double *nbr_grad_y_dev, const double *nbr_grad_z_dev,
                                    const double *gamma_magsf_dev_or_null, double *flux_out_dev);
/* fvPatchField::patchInternalField: out[i] = psi[faceCells[i]] (zeroGradient boundary values for the Gauss gradient, fvMatrix::flux ...) */
int mi_patch_internal_field(mi_patch_t patch, const double *psi_dev, double *out_dev);
/* inout -= x*y, the product rounded before the subtraction (a temporary field, then operator-=) */
int mi_vec_submul(mi_ctx_t ctx, int64_t n, const double *x_dev, const double *y_dev, double *inout_dev);
int mi_patch_create(mi_ctx_t ctx, int32_t n_cells, int32_t n_patch_faces, const int32_t *face_cells_host, mi_patch_t *out);
int mi_patch_destroy(mi_patch_t patch);
int mi_patch_add(mi_patch_t patch, const double *pf_dev, double *intf_dev, int fn);
/* coupled part of fvMatrix::addBoundarySource (fvMatrix.C:318-346; used by fvMatrix::H, :1458-1506):
 * intf[faceCells[i]] +=(fn 0) / -=(fn 1) pf[i]*q[i], e.g. boundaryCoeffs * patchNeighbourField.            */
int mi_patch_add_product(mi_patch_t patch, const double *pf_dev, const double *q_dev, double *intf_dev, int fn);
/* boundary part of fvMatrix::flux (fvMatrix.C:1621-1653): flux[i] = internalCoeffs[i]*psi[faceCells[i]] -
 * boundaryCoeffs[i] * patchNeighbourField[i]  (coupled patch) | - boundaryCoeffs[i]  (NULL neighbour field: not coupled).
 * The internal-face part of flux() is mi_faceH.                                                                      */
int mi_patch_flux(mi_patch_t patch, const double *internal_coeffs_dev, const double *boundary_coeffs_dev,
                  const double *psi_dev, const double *patch_neighbour_field_dev_or_null, double *flux_dev);
/* fvMatrix<scalar>::relax(alpha) (fvMatrix.C:1087-1345, functors :983-1084): coupled[p] != 0 marks processor-like patches.
 * mi_relax_multi: the same for an equation with n_rhs sources / solution components that share the diagonal (fvMatrix<vector>::relax:
 * S += (D - D0)*psi per component); sum_mag_off_diag_dev_or_null: lduMatrix::sumMagOffDiag of the coefficients if the caller already
 * has it (mi_fvm_assemble's by-product; lower / upper may then be NULL) -- completed IN PLACE with the coupled patches' |boundaryCoeffs|. */
int mi_relax_multi(mi_addr_t addr, double alpha, double *diag_dev, const double *lower_dev, const double *upper_dev,
                   double *sum_mag_off_diag_dev_or_null, int32_t n_rhs, double *const *source_dev, const double *const *psi_dev,
                   int32_t n_patches, const mi_patch_t *patches, const double *const *internal_coeffs_dev,
                   const double *const *boundary_coeffs_dev, const int32_t *coupled);
int mi_relax(mi_addr_t addr, double alpha, double *diag_dev, const double *lower_dev, const double *upper_dev,
             double *source_dev, const double *psi_dev, int32_t n_patches, const mi_patch_t *patches,
             const double *const *internal_coeffs_dev, const double *const *boundary_coeffs_dev,
             const int32_t *coupled);

/* host-only inspection of the hierarchy builder (no device; CPU tests compare it with the
 * oracle's independent restatement).  name: restrictMap, faceRestrict, faceFlip (uint8), cLower,
 * cUpper, cellChildStart, cellChild, faceChildStart, faceChild, diagChildStart, diagChild.     */
int mi_gamg_host_build(int32_t n_cells, int32_t n_faces, const int32_t *lower_addr_host,
                       const int32_t *upper_addr_host, const double *face_weights_host,
                       int32_t n_cells_in_coarsest_level, int32_t merge_levels, int forward_init, void **hierarchy_out);
/* every domain of a DECOMPOSED case in one process (one thread per domain runs the per-rank builder, a barrier and a shared
 * table stand in for the communicator): patch p of domain d couples to patch patch_nbr_patch[d][p] of domain
 * patch_nbr_domain[d][p].  hierarchies_out[n_domains]; mi_gamg_host_patch_array: faceRestrict / faceCells / nbrCells of a
 * coupled patch on a level.  CPU tests compare with the oracle's multi-domain builder.                                  */
int mi_gamg_host_build_domains(int32_t n_domains, const int32_t *n_cells, const int32_t *n_faces,
                               const int32_t *const *lower_addr_host, const int32_t *const *upper_addr_host,
                               const double *const *face_weights_host, const int32_t *n_patches,
                               const int32_t *const *patch_sizes, const int32_t *const *const *patch_face_cells,
                               const int32_t *const *patch_nbr_domain, const int32_t *const *patch_nbr_patch,
                               int32_t n_cells_in_coarsest_level, int32_t merge_levels, int forward_init,
                               void **hierarchies_out);
/* one domain whose coupled patches are all cyclicAMI (arguments as mi_addr_set_ami_patch / mi_addr_set_ami_face_areas take
 * them, per patch); mi_gamg_host_patch_array then also answers "amiStart" / "amiAddr" (int32) and "amiW" / "amiMagSf" (double) */
int mi_gamg_host_build_ami(int32_t n_cells, int32_t n_faces, const int32_t *lower_addr_host, const int32_t *upper_addr_host,
                           const double *face_weights_host, int32_t n_cells_in_coarsest_level, int32_t n_patches,
                           const int32_t *patch_sizes, const int32_t *const *patch_face_cells_host, const int32_t *patch_nbr_patch,
                           const int32_t *const *ami_start_host, const int32_t *const *ami_addr_host, const double *const *ami_w_host,
                           const double *const *ami_mag_sf_host, void **hierarchy_out);
int mi_gamg_host_patch_array(void *hierarchy, int32_t level, int32_t patch, const char *name, const void **data,
                             int64_t *len);
int32_t mi_gamg_host_n_levels(void *hierarchy);
int mi_gamg_host_array(void *hierarchy, int32_t level, const char *name, const void **data,
                       int64_t *len, int32_t *elem_size);
int mi_gamg_host_free(void *hierarchy);

/* ---- distributed PCG (one rank per GPU).  The reference runs the same PCG on every
 * MPI rank and meets its neighbours in Foam::reduce (PCG.C:142,166,195 ->
 * src/Pstream/mpi/allReduceTemplates.C:195-208) and in the processor-patch exchange inside
 * Amul (lduMatrixUpdateMatrixInterfaces.C:30-276).  Here the device-resident pipeline is
 * cut at exactly those points: the caller owns the engine-order vectors (n_cells + n_ext
 * doubles each), an 8-double scalar block and the send buffer (n_ext doubles), runs the
 * collectives (RCCL through torch.distributed) between phases, and receives the halo
 * straight into pA[n_cells ..).  scal: [0] sum wA.rA  [1] sum|rA|  [2] sum wA.pA
 * [3] sum psi  [4] normFactor sum.  Phases:
 *    0  pack psi for the exchange            (then: exchange into psi ext)
 *    1  wA=A psi, rA, sumA, scal[3]          (then: allreduce scal[3]; avg = scal[3]/N_global)
 *    2  arg=avg: scal[4], scal[1], scal[0]   (then: allreduce scal[0..1], scal[4])
 *    3  normFactor, initial residual, first convergence test
 *   10  (test of it-1,) pA update, pack pA   (then: start exchange into pA ext)
 *   11  Amul interior tiles                  (overlaps the exchange)
 *   12  Amul boundary tiles, scal[2]         (after the exchange; then allreduce scal[2])
 *   13  psi, rA update, scal[1], scal[0]     (then: allreduce scal[0..1])
 *   14  convergence test of iteration `it`   (end of a batch)                           */
int mi_dpcg_set_buffers(mi_matrix_t m, double *psi_e, double *src_e, double *pA_e, double *wA_e,
                        double *rA_e, double *scal8, double *send_buf,
                        const mi_solver_controls *controls, int precond, int32_t history_len);
int mi_dpcg_phase(mi_matrix_t m, int phase, int32_t it, double arg);
int mi_dpcg_status(mi_matrix_t m, mi_solver_perf *perf_out, int32_t *done_out,
                   double *residual_history_host, int32_t history_len);
/* ---- RCCL communicator and the C++ host loop of the distributed PCG.  Replaces the reference's MPI layer
 * on this path (src/Pstream/mpi/UPstream.C, allReduceTemplates.C:195-208, UIPread.C/UOPwrite.C as used by
 * processorFvPatchScalarField.C:36-170): one rank per GPU, all-reduces on the engine's stream, halo
 * send/recv on a second stream so Amul's interior tiles overlap the exchange.  RCCL is bound at run time;
 * its absence is an error here (MI_ERR_DEVICE), never a fallback.
 *   id: 128 bytes from mi_comm_unique_id on rank 0, distributed to all ranks by the launcher (any channel).
 *   mi_dpcg_comm_begin: prologue up to the first convergence test (PCG.C:75-118); buffers and controls
 *     come from mi_dpcg_set_buffers.  reduce/halo may be the same communicator; two let the halo traffic
 *     and the scalar all-reduces progress independently.  patch_rank[p] = rank on the other side of
 *     processor patch p; patch_nbr_patch[p] = index of the matching patch on that rank (NULL = same index).
 *   mi_dpcg_comm_iterate: enqueue n_iters iterations without host synchronisation; poll with
 *     mi_dpcg_status.  record_amul_events = s > 0 brackets the Amul phases of every s-th enqueued iteration
 *     (k = 0, s, 2s, ...) with the events 2(k/s), 2(k/s)+1 of the matrix (mi_event_elapsed_ms).       */
int mi_comm_unique_id(void *id_out, int32_t len);
int mi_comm_create(mi_ctx_t ctx, int32_t n_ranks, int32_t rank, const void *id, mi_comm_t *out);
int mi_comm_destroy(mi_comm_t comm);
/* A communicator over the CALLER's transport instead of RCCL -- the reference's own is MPI through Pstream
 * (src/Pstream/mpi/UPstream.C:185-262 reduce, processorLduInterfaceTemplates.C:127-298 send/receive), so a shim inside a running
 * OpenFOAM application can hand its Pstream calls in here and needs no second launcher-level bootstrap.  Blocking by contract:
 * before every callback the engine synchronises the stream that produced the buffers; a callback returns (0 = ok) once its
 * results are in place in DEVICE memory.  allreduce: in-place sum over all ranks of n doubles.  exchange: one grouped
 * neighbour exchange (init/updateMatrixInterfaces): n_send messages (peer rank, tag, device pointer, count) and n_recv
 * messages; a message is matched by (peer, tag) -- the tag is the SENDER's patch index on both ends, so several patches per
 * peer and a rank that is its own neighbour work.  Everything the RCCL communicators do (attached operators and solvers,
 * mi_dpcg_comm_*, mi_gamg_create_coupled) runs over such a communicator too, without the compute/exchange overlap.          */
typedef int (*mi_comm_allreduce_fn)(void *user, double *buf_dev, int64_t n);
typedef int (*mi_comm_exchange_fn)(void *user, int32_t n_send, const int32_t *send_peer, const int32_t *send_tag,
                                   const double *const *send_dev, const int64_t *send_count, int32_t n_recv,
                                   const int32_t *recv_peer, const int32_t *recv_tag, double *const *recv_dev,
                                   const int64_t *recv_count);
int mi_comm_create_external(mi_ctx_t ctx, int32_t n_ranks, int32_t rank, mi_comm_allreduce_fn allreduce,
                            mi_comm_exchange_fn exchange, void *user, mi_comm_t *out);
int mi_comm_allreduce_sum(mi_comm_t comm, double *buf_dev, int64_t n);
/* Attach communicators to a matrix: from here on it behaves like the reference's lduMatrix on a decomposed
 * case -- mi_amul/tmul/residual/H/jacobi_smooth exchange the processor-patch values themselves
 * (init/updateMatrixInterfaces), every global sum inside the solvers is all-reduced, and
 * mi_pcg_solve / mi_pbicg_solve / mi_pbicgstab_solve / mi_smooth_solve solve the GLOBAL system; all ranks call
 * them together, like the MPI ranks of the reference.  (mi_pcg_solve: diagonal/none run the device-resident
 * phase pipeline, AINV a host-stepped loop.)  GAMG on such a matrix: mi_gamg_create_coupled.                 */
/* ONE-SHOT PEER ALL-REDUCE ("RCCL ncclAllReduce fallback / one-shot P2P all-to-all on the fully connected
 * node"; replaces the latency of src/Pstream/mpi/allReduceTemplates.C:195-208 for the 1-3 scalars of a Krylov iteration).
 * Opt-in, on any communicator (RCCL or external): every rank calls mi_comm_peer_window (allocates a window of device memory
 * -- fine-grained when the runtime exports that over IPC -- and returns its 64-byte hipIpcMemHandle), ships the handle to
 * all ranks with whatever transport it has, and calls mi_comm_peer_connect with the handles of ALL ranks in rank order;
 * after a barrier of the caller's, every all-reduce of <= 8 doubles on this communicator is one single-workgroup kernel
 * per rank: store the values + an epoch flag into the own slot of every rank's window, poll the own window until all
 * slots carry the epoch, add them in rank order (the same bits on every rank).  Larger all-reduces keep the
 * communicator's ordinary path.  mi_comm_peer_status: 0, or 1 once a wait has run out of polls (a rank that never
 * arrived: the sums are void); fine_grained_out says which memory the window got.  Exercised between processes that
 * share one device (tests/test_distributed.py); between devices it needs the fine-grained window -- unmeasured here. */
int mi_comm_peer_window(mi_comm_t comm, void *ipc_handle_out_64_bytes, int32_t len);
int mi_comm_peer_connect(mi_comm_t comm, const void *handles_rank_order, int32_t n_handles);
int mi_comm_peer_status(mi_comm_t comm, int32_t *status_out, int32_t *fine_grained_out_or_null);
/* The same set-up without help from the launcher, plus what makes it safe to be the DEFAULT (round 3):
 *   mi_comm_peer_auto     collective.  Allocates the window, all-gathers the IPC handles over the communicator's own transport
 *                         (RCCL or the external callbacks; the bytes travel as 16-bit integers through one all-reduce), maps the
 *                         peers, runs the coherence self-test and lets the ranks agree: *enabled_out = 1 on every rank, or 0 on
 *                         every rank (then nothing changes: RCCL / the external transport stay in charge).  It never fails because
 *                         windows are unavailable -- no IPC, a window that is not fine-grained while a peer sits on another device
 *                         (mi_comm_peer_connect refuses that with MI_ERR_UNSUPPORTED), a wait that runs out -- only on a broken
 *                         base transport.
 *   mi_comm_peer_selftest rounds all-reduces of known values against every peer with a short poll limit; *ok_out = 1 when every
 *                         sum is right.  All ranks together, after every rank has connected.
 *   mi_comm_peer_enable   switch the windows on / off after the ranks have agreed on the self-test's outcome.
 * A communicator in peer mode also moves the HALO of every matrix attached to it afterwards through windows
 * (mi_matrix_peer_halo_auto below), and the phase loop of the distributed PCG becomes three launches per iteration with no
 * collective call.  A wait that runs out of polls (a rank that died) raises a status word; the solver loops check it at their
 * host synchronisations and return MI_ERR_DEVICE (they used to carry on with void sums).                                      */
int mi_comm_peer_auto(mi_comm_t comm, int32_t *enabled_out);
int mi_comm_peer_selftest(mi_comm_t comm, int32_t rounds, int32_t *ok_out);
int mi_comm_peer_enable(mi_comm_t comm, int32_t on);
/* HALO WINDOWS (replaces lduMatrixUpdateMatrixInterfaces.C:30-276 / processorFvPatchScalarField.C:36-170 -- pack, host-staged
 * MPI_Isend/Irecv, unpack -- and this engine's own ncclSend/ncclRecv group): every attached matrix owns a window of
 * fine-grained device memory ([2 parities][n_ext] values + an epoch flag per patch and parity) that its neighbour ranks map
 * over hipIpc.  An exchange is then: one launch that gathers the patch-internal values and STORES them into the neighbours'
 * windows (k_halo_push; flags behind a system-scope release), the interior tiles meanwhile, one small launch that waits for the
 * own flags and copies the window into the operand's ext region (k_halo_pull; transformCoupleField factors of processorCyclic
 * patches applied there), the boundary tiles.  No second stream, no events, no collective call.  mi_matrix_attach_comm sets the
 * windows up by itself (collectively: every rank attaches its matching matrix at that point, GAMG level matrices included)
 * when the reduce communicator is in peer mode (MI_PEER_HALO=0 keeps the send/recv path); mi_matrix_peer_halo_auto does it
 * explicitly.  The set-up ends with a self-test against the communicator's ordinary exchange and an agreement of the ranks;
 * *enabled_out = 0 leaves the matrix on send/recv.  mi_matrix_peer_halo_status: whether windows are in use and whether a wait
 * has run out of polls.
 * With halo windows AND a peer-mode reduce communicator, mi_dpcg_comm_iterate (and mi_pcg_solve on the attached matrix,
 * diagonal / none) runs WITHOUT any collective call: k_dpcg_update_p (p-update + convergence test of the previous iteration +
 * deferred psi update; every block packs the patch cells of its own chunk into the neighbours' windows), tile_kernel_dist (all
 * tiles in one launch, interior first; block 0 raises the neighbours' flags, a boundary tile polls its own and reads the window
 * directly), a one-workgroup kernel that folds wA.pA and all-reduces it through the windows, k_dpcg_update_psi_r, a one-workgroup
 * kernel for sum|rA| and the next wA.rA: five launches, 51 us per iteration at 108^3 cells per rank against 81 us for the phase
 * loop over RCCL on the same box (profiles/r03_a_*; MI_DPCG_FUSED=0: the phase loop, 3 | 4: the all-reduces inside the passes,
 * measured slower).  The sums are formed in the order the separate reduction kernels use: same bits as the phase loop.
 * Not on this path: cyclicAMI / transformed patches (mi_dpcg_set_buffers refuses them: the boundary tiles read the window
 * without the interpolation / factor; mi_pcg_solve on such a matrix takes the tile-operator pipeline, whose k_halo_pull
 * applies the factors) and AINV (its preconditioner is itself a tile pass with a halo of its own).                              */
int mi_matrix_peer_halo_auto(mi_matrix_t m, int32_t *enabled_out);
int mi_matrix_peer_halo_status(mi_matrix_t m, int32_t *enabled_out, int32_t *status_out_or_null);
int mi_matrix_attach_comm(mi_matrix_t m, mi_comm_t reduce, mi_comm_t halo, const int32_t *patch_rank,
                          const int32_t *patch_nbr_patch_or_null, int64_t n_global_cells);
int mi_matrix_detach_comm(mi_matrix_t m);
/* coupledFvPatchField::patchNeighbourField (processorFvPatchField.C:107-135, cyclicFvPatchField.C:109-150): psi in the
 * cell across every interface face, n_ext values in caller patch order.  Cyclic patches read the partner cells;
 * processor patches exchange over the attached halo communicator (all ranks call together) -- what the coupled parts
 * of fvMatrix::flux / H consume (mi_patch_flux, mi_patch_add_product).                                           */
int mi_matrix_patch_neighbour_field(mi_matrix_t m, const double *psi_dev, double *nbr_out_dev);
int mi_dpcg_comm_begin(mi_matrix_t m, mi_comm_t reduce, mi_comm_t halo, const int32_t *patch_rank,
                       const int32_t *patch_nbr_patch_or_null, int64_t n_global_cells);
int mi_dpcg_comm_iterate(mi_matrix_t m, int32_t n_iters, int32_t record_amul_events);
/* HIP-event helpers on the context's stream (bench.py times the Amul phases with them) */
int mi_event_record(mi_matrix_t m, int32_t idx);
int mi_event_elapsed_ms(mi_matrix_t m, int32_t idx0, int32_t idx1, float *ms_out);

/* fvMatrix<vector>::solveSegregated (src/finiteVolume/fvMatrices/fvMatrix/fvMatrixSolve.C:103-225) as ONE call: nrhs <= 3
 * right-hand sides against the same matrix (the components of a momentum equation).  Every component runs the reference's
 * PBiCG loop (PBiCG.C:67-246) with its own scalars, convergence test and iteration count -- the same fma chains and reduction
 * trees as mi_pbicg_solve, hence the same bits per component -- while each pass over the matrix serves all components:
 * tile_kernel_multi stages a tile's upper / lower once for the 2 x nrhs operand vectors of a step (A pA_c and A^T pT_c; the
 * DILU pair precondition / preconditionT), sumA and 1/diag are computed once per coefficient binding, the host polls all
 * components with one copy.  diag_dev: nrhs device pointers to the components' DIAGONALS (caller order) -- solveSegregated adds
 * the boundary contribution per component, addBoundaryDiag(diag, cmpt), so the components share upper / lower but not
 * necessarily the diagonal -- or NULL: the bound diagonal for all.  psi_dev / source_dev: nrhs device pointers; perf_out: nrhs records;
 * residual_history_host: nrhs rows of history_len doubles (or NULL).  On a matrix with communicators attached (all ranks call
 * together) the shared passes stay: pA and pT of all components cross the processor patches in ONE halo exchange per pass, the
 * components' sums at a sum point travel in one all-reduce (round 4); only with cyclicAMI patches on an attached matrix are
 * the components solved one after the other.
 * mi_pbicg_solve itself uses the one-component form of the same kernel: A pA and A^T pT in one pass (MI_PBICG_PAIR=0: two).   */
int mi_pbicg_solve_multi(mi_matrix_t m, int32_t nrhs, const double *const *diag_dev_or_null, double *const *psi_dev, const double *const *source_dev,
                         const mi_solver_controls *controls, int precond, mi_solver_perf *perf_out,
                         double *residual_history_host, int32_t history_len);
int mi_pbicg_solve(mi_matrix_t m, double *psi_dev, const double *source_dev,
                   const mi_solver_controls *controls, int precond,
                   mi_solver_perf *perf_out, double *residual_history_host, int32_t history_len);
/* replicate_quirk != 0 keeps the reference's psi += omega*yA (PBiCGStab.C:263-270) */
int mi_pbicgstab_solve(mi_matrix_t m, double *psi_dev, const double *source_dev,
                       const mi_solver_controls *controls, int precond, int replicate_quirk,
                       mi_solver_perf *perf_out, double *residual_history_host, int32_t history_len);
int mi_smooth_solve(mi_matrix_t m, double *psi_dev, const double *source_dev,
                    const mi_solver_controls *controls, double omega, int32_t n_sweeps,
                    mi_solver_perf *perf_out, double *residual_history_host, int32_t history_len);

/* ---- benchmark hooks: run `iters` PCG iterations (engine order, no convergence
 * exit, same kernels as mi_pcg_solve) and `reps` Amuls back to back, timed with
 * HIP events on the context's stream; milliseconds returned.                      */
int mi_bench_amul(mi_matrix_t m, int32_t reps, float *ms_out);
/* diagnostic: resident Amul workgroups per CU as the HIP runtime computes it, LDS bytes per workgroup, block size */
int mi_debug_occupancy(mi_matrix_t m, int32_t *blocks_per_cu, int32_t *lds_bytes_out, int32_t *block_size);
/* diagnostic / test hook: the dense inversions behind GAMG's direct coarsest-level solve (GAMGSolverSolve.C:551-573: the reference
 * LU-solves the coarsest matrix on the host every cycle; the engine inverts it once per set of coefficients) on a row-major n x n
 * matrix in device memory.  which = 0 host Gauss-Jordan with partial pivoting, 1 / 2 the register-resident kernels (n <= 192; same
 * bits as 0), 3 the global-memory kernel.  *singular_out = 1: zero pivot column met.                                              */
int mi_debug_dense_invert(mi_ctx_t ctx, const double *a_dev, int32_t n, double *inv_dev, int32_t which, int32_t *singular_out);
int mi_bench_pcg_iters(mi_matrix_t m, const double *source_dev, int32_t iters, int precond,
                       float *ms_out, float *amul_ms_out);

/* ---- host-only layout inspection (no device; used by the CPU-side tests to
 * verify the tiling by interpreting the tables).  name is one of e2c, c2e,
 * tileCellStart, tileSlotStart, tileHaloStart, haloCell, tileSliceStart,
 * sliceEntryStart, entries (uint32), slotFace, extSlot, interiorTiles,
 * boundaryTiles, patchOffset, patchFaceCellsE, faceSlot (all int32 otherwise). */
int mi_layout_build_host(int32_t n_cells, int32_t n_faces, const int32_t *lower_addr_host,
                         const int32_t *upper_addr_host, int32_t n_patches, const int32_t *patch_sizes,
                         const int32_t *const *patch_face_cells_host,
                         const int32_t *const *patch_nbr_cells_host_or_null, int32_t tile_cells,
                         int32_t slot_cap, void **layout_out);
int mi_layout_array(void *layout, const char *name, const void **data, int64_t *len);
/* test hooks (round 4): the layout of a GIVEN partition (part[c] in [0, n_parts): tile of caller cell c; the partition of a
 * clustered layout gives that layout back), and the tiles a coarse GAMG level inherits from its fine level's tiles
 * (csrc/tiling.hpp inherit_tiles; part_out: n_coarse ids) */
int mi_layout_build_host_given(int32_t n_cells, int32_t n_faces, const int32_t *lower_addr_host, const int32_t *upper_addr_host,
                               int32_t n_parts, const int32_t *part_host, void **layout_out);
int mi_layout_inherit_tiles(int32_t n_fine, const int32_t *restrict_map, const int32_t *fine_tile_of_cell, int32_t n_fine_tiles,
                            int32_t n_coarse, int32_t n_coarse_faces, const int32_t *c_lower, const int32_t *c_upper,
                            int32_t cell_cap, int32_t slot_cap, int32_t *part_out, int32_t *n_parts_out);
int mi_layout_free(void *layout);

#ifdef __cplusplus
}
#endif
#endif /* MI_LDU_H */
