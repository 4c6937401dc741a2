/*
 * elag_hip.h -- C ABI of the MI355X-native AMGe V-cycle hot path (libelag_hip.so).
 *
 * This is the drop-in boundary for ParElag's solver plug-in API: a host-side SolverFactory registered
 * with SolverLibrary::AddNewSolverFactory (src/linalg/solver_core/ParELAG_SolverLibrary.hpp:199-203)
 * pulls the hypre_ParCSRMatrix pieces out of mfem::HypreParMatrix and hands them to the functions
 * below; the returned opaque handles stand behind mfem::Operator::Mult.  See INTEGRATION.md for the
 * reference-side binding.  Every entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - every function returns an int status (0 = ELAG_HIP_OK); never throws across the ABI.
 *     elag_hip_last_error() returns the message of the last failure on the calling thread
 *     (the C++ shim rethrows it like PARELAG_ASSERT_HYPRE_ERROR_FLAG, src/utilities/elagError.hpp:83-110).
 *   - indices are 32-bit (HYPRE_Int), global ids 64-bit (HYPRE_BigInt), values fp64.
 *   - *_create functions COPY the caller's host arrays to the device; the caller keeps ownership of its
 *     arrays (SURVEY 8b "Ownership").  Handles are destroyed with the matching *_destroy.
 *   - vector arguments carry a location flag `loc`: ELAG_HIP_HOST pointers are staged through device
 *     buffers (synchronous on return), ELAG_HIP_DEVICE pointers are used in place and the call is
 *     asynchronous on the library's compute stream (elag_hip_synchronize() to wait).
 *   - one calling host thread per process and one device per process (ParElag runs one thread per MPI
 *     rank: INSTALL:44-48); Hierarchy/HiptmairSmoother hold scratch exactly as the reference's mutable
 *     members do (ParELAG_Hierarchy.hpp:296-298), so apply calls on one handle are not re-entrant.
 */
#ifndef ELAG_HIP_H
#define ELAG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ELAG_HIP_API __attribute__((visibility("default")))

typedef struct elag_hip_parcsr_s *elag_hip_parcsr_t;       /* mfem::HypreParMatrix / hypre_ParCSRMatrix   */
typedef struct elag_hip_smoother_s *elag_hip_smoother_t;   /* HypreSmootherWrapper / HiptmairSmoother     */
typedef struct elag_hip_coarse_s *elag_hip_coarse_t;       /* the Level's "CoarseSolver"                  */
typedef struct elag_hip_hierarchy_s *elag_hip_hierarchy_t; /* parelag::Hierarchy                          */
typedef struct elag_hip_blockop_s *elag_hip_blockop_t;     /* parelag::MfemBlockOperator                  */
typedef struct elag_hip_krylov_s *elag_hip_krylov_t;       /* parelag::KrylovSolver / StationarySolver    */

enum { ELAG_HIP_HOST = 0, ELAG_HIP_DEVICE = 1 };

enum {
    ELAG_HIP_OK = 0,
    ELAG_HIP_ERR_ARG = 1,      /* bad argument / shape mismatch (Hierarchy.cpp:344-352 throws for these) */
    ELAG_HIP_ERR_HIP = 2,      /* a HIP runtime call failed                                            */
    ELAG_HIP_ERR_COMM = 3,     /* RCCL failure / communicator not initialised                          */
    ELAG_HIP_ERR_STATE = 4,    /* library not initialised, hierarchy not finalised, ...                */
    ELAG_HIP_ERR_NUMERIC = 5,  /* zero l1 norm, non-SPD coarse matrix, ...                             */
    ELAG_HIP_ERR_NODEVICE = 6  /* no usable GPU                                                        */
};

/* hypre relax types accepted by elag_hip_smoother_create -- the string->type table of
 * src/linalg/factories/ParELAG_HypreSmootherFactory.cpp:92-109.  The Gauss-Seidel types (2 = the reference's shipped default
 * "L1 Gauss-Seidel", 4 = "L1 Gauss-Seidel Truncated", 6 = "Gauss-Seidel") run as level-scheduled forward + backward sweeps with
 * the sequential loop's results; "Damping Factor" != 1 (2 / 4: hypre_ParCSRRelax's scaled SSOR) and, for 6, "Omega" != 1
 * (hypre_BoomerAMGRelax's weighted branch) take hypre's own branches, one launch per dependency level; 5 = "Lumped Jacobi" (mfem::HypreSmoother: l1 = A 1, then type 1); 413 = "Hypre Jacobi" (mfem::HypreDiagScale,
 * ParELAG_HypreSmootherFactory.cpp:28-31); 3 = "Kaczmarz" (hypre_ParCSRRelax with the rows' squared 2-norms: level-scheduled on the
 * rows' shared columns, the sequential loop's results; "Omega" free); 1001 = "Taubin" and 1002 = "FIR" (the polynomial smoothers of
 * mfem::HypreSmoother with mfem's default options -- lambda 0.5, mu -0.5, 40 steps; Blackman window -- which the reference's factory
 * cannot change, ParELAG_HypreSmootherFactory.cpp:38-40; "Cheby Poly Order" / "Cheby Poly Fraction" are FIR's order and pass band;
 * max_eig = hypre_ParCSRMaxEigEstimateCG(A, scale 0, 10) / 2, readable and settable through the eig-estimate entry points). */
enum {
    ELAG_HIP_SMOOTHER_JACOBI = 0, ELAG_HIP_SMOOTHER_L1_JACOBI = 1, ELAG_HIP_SMOOTHER_L1_GS = 2, ELAG_HIP_SMOOTHER_KACZMARZ = 3,
    ELAG_HIP_SMOOTHER_L1_GS_TRUNCATED = 4, ELAG_HIP_SMOOTHER_LUMPED_JACOBI = 5, ELAG_HIP_SMOOTHER_GS = 6, ELAG_HIP_SMOOTHER_CHEBYSHEV = 16,
    ELAG_HIP_SMOOTHER_HYPRE_JACOBI = 413, ELAG_HIP_SMOOTHER_TAUBIN = 1001, ELAG_HIP_SMOOTHER_FIR = 1002
};

/* --------------------------------------------------------------------------------------------------
 * Library / device / communicator
 * ------------------------------------------------------------------------------------------------ */
ELAG_HIP_API int elag_hip_init(int device);                 /* hipSetDevice + private streams          */
ELAG_HIP_API int elag_hip_finalize(void);
ELAG_HIP_API int elag_hip_synchronize(void);                /* wait for the compute stream             */
/* Device memory the library holds without using it goes back to the driver: the pool of freed blocks it keeps between calls (as hypre's
 * caching allocator does for its work arrays: the products, transposes and level operators a ParElag setup makes one by one reuse each other's
 * memory; ELAG_HIP_POOL_MB, default 1024, 0 = no pool) and the idle slabs of the setup scratch arena (elag_hip_init reserves the first one:
 * ELAG_HIP_ARENA_MB, default 2048).  The library does this by itself when the driver runs out of memory.  bytes_released may be NULL. */
ELAG_HIP_API int elag_hip_trim_memory(size_t *bytes_released);
ELAG_HIP_API const char *elag_hip_last_error(void);
ELAG_HIP_API int elag_hip_device_name(char *buf, int len);

/* One MPI rank <-> one GPU.  Replaces hypre's MPI communicator inside hypre_ParCSRCommHandleCreate
 * (call site src/hypreExtension/hypre_ParCSRMatrixMatvecBoolInt.c:189) by an RCCL communicator over xGMI.
 * Rank 0 calls elag_hip_comm_unique_id, the host broadcasts the 128 bytes (MPI_Bcast / torch.distributed),
 * every rank calls elag_hip_comm_init.  With nranks == 1 no communicator is created. */
ELAG_HIP_API int elag_hip_comm_unique_id(void *id128);
ELAG_HIP_API int elag_hip_comm_init(int rank, int nranks, const void *id128);
/* In-process communicator: `nranks` host threads of one process, each with its own library context on the same
 * device, exchange through a mailbox and device-to-device copies (same `group` id on all of them).  It lets the whole
 * multi-rank path run on a single-GPU box (tests); a production run uses elag_hip_comm_init. */
ELAG_HIP_API int elag_hip_comm_init_inprocess(int rank, int nranks, int group);
/* Diagnostic for single-GPU boxes: runs the RCCL wrappers for real on a communicator of size one (dlopen, unique id,
 * ncclCommInitRank, ncclAllReduce sum / max of {1.5,-2} / {3.25,8}, a grouped ncclSend/ncclRecv of the four values to
 * this rank, ncclCommDestroy).  Both outputs have 4 entries and must come back equal to the inputs. */
ELAG_HIP_API int elag_hip_comm_selftest(double *allreduce_out, double *sendrecv_out);
ELAG_HIP_API int elag_hip_comm_rank(int *rank, int *nranks);
/* Data-plane rounds this rank has issued since it was initialised: neighbour exchanges with at least one remote peer (the halo /
 * reverse-halo of hypre_ParCSRMatrixMatvec(T), src/hypreExtension/hypre_ParCSRMatrixMatvecBoolInt.c:178-196,296-323) and device
 * all-reduces.  Differences around a call count the messages of that call (tests pin the rounds of a V-cycle with them). */
ELAG_HIP_API int elag_hip_comm_rounds(long long *exchanges, long long *allreduces);
/* 8-byte reductions that end the norm / dot routines (MPI_Allreduce in hypre_ParCSRMatrixNorms.c:60) */
ELAG_HIP_API int elag_hip_comm_allreduce(double *value, int op /*0 sum, 1 max*/);

/* Device memory for hosts that do not bring their own allocator (the bench harness uses torch). */
ELAG_HIP_API int elag_hip_malloc(void **dptr, size_t bytes);
ELAG_HIP_API int elag_hip_free(void *dptr);
ELAG_HIP_API int elag_hip_memcpy(void *dst, const void *src, size_t bytes, int kind /*0 H2D,1 D2H,2 D2D*/);
ELAG_HIP_API int elag_hip_memset(void *dptr, int value, size_t bytes);

/* --------------------------------------------------------------------------------------------------
 * ParCSR matrix  (hypre_ParCSRMatrix as accessed in hypre_ParCSRMatrixMatvecBoolInt.c:149-185 and
 * src/linalg/factories/ParELAG_SchurComplementFactory.cpp:74-98)
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    /* diag block: local rows x local columns */
    int num_rows, num_cols_diag;
    const int *diag_i, *diag_j;
    const double *diag_a;
    /* offd block: local rows x compressed ghost columns; all NULL / 0 on a single rank */
    int num_cols_offd;
    const int *offd_i, *offd_j;
    const double *offd_a;
    const int64_t *col_map_offd;           /* ascending global column ids of the ghost columns */
    /* global layout (assumed partition: src/structures/SharingMap.cpp:103-108) */
    int64_t first_row, first_col_diag, global_num_rows, global_num_cols;
    /* comm package (hypre_ParCSRCommPkg): may be all 0/NULL when num_cols_offd == 0 */
    int num_sends;
    const int *send_procs, *send_map_starts, *send_map_elmts;
    int num_recvs;
    const int *recv_procs, *recv_vec_starts;
} elag_hip_parcsr_desc;

ELAG_HIP_API int elag_hip_parcsr_create(elag_hip_parcsr_t *out, const elag_hip_parcsr_desc *desc);
/* The same matrix from this rank's rows in GLOBAL column numbering -- what mfem::HypreParMatrix(comm, nrows, glob_nrows, glob_ncols, I, J, data,
 * rows, cols) takes ([3P] mfem linalg/hypre.cpp; hypre_ParCSRMatrixGenerateFromCSR / hypre_MatvecCommPkgCreate behind it): the library splits
 * the rows into the diag / offd blocks on the device, builds col_map_offd and negotiates the comm package over ITS communicator (collective:
 * every rank of elag_hip_comm_init calls it).  Column ids ascending within a row are not required; global sizes below 2^31. */
ELAG_HIP_API int elag_hip_parcsr_create_from_global_csr(elag_hip_parcsr_t *out, int num_rows, const int *row_ptr, const int *global_col,
                                                        const double *values, int64_t first_row, int64_t global_num_rows, int64_t first_col,
                                                        int num_cols_local, int64_t global_num_cols);
ELAG_HIP_API int elag_hip_parcsr_destroy(elag_hip_parcsr_t A);

typedef struct {
    int num_rows, num_cols_diag, num_cols_offd;
    int64_t nnz_diag, nnz_offd, first_row, first_col_diag, global_num_rows, global_num_cols;
    int num_sends, num_recvs, send_total;
} elag_hip_parcsr_info;
ELAG_HIP_API int elag_hip_parcsr_get_info(elag_hip_parcsr_t A, elag_hip_parcsr_info *info);

/* Copy a (device-built) matrix back to caller-allocated host arrays sized from get_info, e.g. to hand a
 * coarse operator to a hypre coarse solver.  diag_first != 0 stores the diagonal entry first in each
 * row of a square diag block (hypre / MFEM convention); otherwise columns ascend. */
ELAG_HIP_API int elag_hip_parcsr_export(elag_hip_parcsr_t A, int diag_first,
                                        int *diag_i, int *diag_j, double *diag_a,
                                        int *offd_i, int *offd_j, double *offd_a, int64_t *col_map_offd);

/* y = alpha*op(A)*x + beta*y.  trans=0: hypre_ParCSRMatrixMatvec (A->Mult, Hierarchy.cpp:124-125,193;
 * P->Mult :231-234); trans=1: hypre_ParCSRMatrixMatvecT (P->MultTranspose :199-203, D^T in
 * HiptmairSmoother.cpp:62-63).  Halo pattern of hypre_ParCSRMatrixMatvecBoolInt.c:148-201 / :212-345. */
ELAG_HIP_API int elag_hip_parcsr_spmv(elag_hip_parcsr_t A, int trans, double alpha, const double *x,
                                      double beta, double *y, int loc);
/* r = b - A*x   (mg_utils::ComputeResidual, src/linalg/utilities/ParELAG_MG_Utils.hpp:405-426; the
 * A->Mult + add() pair of Hierarchy.cpp:191-194), fused into the SpMV epilogue. */
ELAG_HIP_API int elag_hip_parcsr_residual(elag_hip_parcsr_t A, const double *b, const double *x, double *r, int loc);

/* Setup-side sparse algebra ------------------------------------------------------------------ */
/* hypre_ParCSRMatrixTranspose2 (src/hypreExtension/par_Tmatmul.c:41-79) */
ELAG_HIP_API int elag_hip_parcsr_transpose(elag_hip_parcsr_t A, elag_hip_parcsr_t *At);
/* hypre_ParMatmul / mfem::ParMult (MG_Utils_impl.hpp:27-88, MatrixUtils.cpp:1053-1063): C = A*B */
ELAG_HIP_API int elag_hip_parcsr_matmul(elag_hip_parcsr_t A, elag_hip_parcsr_t B, elag_hip_parcsr_t *C);
/* hypre_BoomerAMGBuildCoarseOperator(RT,A,P) = RT^T A P: mfem::RAP(A,P) at Hierarchy.cpp:365,511-513;
 * RAP(A,D) at HiptmairSmootherFactory.cpp:160-161; generic RAP/PtAP at MatrixUtils.cpp:919-948.
 * Distributed operands are supported (collective call): external rows of P are fetched along A's comm package,
 * off-rank contributions travel back along RT's, and the result carries its own negotiated comm package. */
ELAG_HIP_API int elag_hip_parcsr_rap(elag_hip_parcsr_t RT, elag_hip_parcsr_t A, elag_hip_parcsr_t P,
                                     elag_hip_parcsr_t *RAP);
/* Gather a distributed matrix onto every rank (rank-local, replicated copy; row_starts: nranks+1 entries or NULL).
 * Collective.  Used for the coarsest operators of a distributed hierarchy. */
ELAG_HIP_API int elag_hip_parcsr_replicate(elag_hip_parcsr_t A, elag_hip_parcsr_t *A_replicated, int64_t *row_starts);
/* IgnoreNonLocalRange(RT, A, P) (src/linalg/utilities/ParELAG_MatrixUtils.cpp:878-916): the same triple product
 * with RT's offd block (and comm package) set aside, i.e. contributions to rows owned by other ranks are dropped.
 * DeRhamSequence::ComputeTrueP / ComputeTrueD go through it (src/amge/DeRhamSequence.cpp:1118-1153).
 * Together with elag_hip_parcsr_rap this also is Assemble(SharingMap, SparseMatrix, SharingMap)
 * (src/structures/SharingMap.cpp:975-1011): RT = range.entity_trueEntity, A = the rank-local matrix wrapped as a
 * block-diagonal ParCSR (elag_hip_parcsr_create with num_cols_offd = 0), P = domain.entity_trueEntity. */
ELAG_HIP_API int elag_hip_parcsr_rap_ignore_nonlocal_range(elag_hip_parcsr_t RT, elag_hip_parcsr_t A, elag_hip_parcsr_t P,
                                                           elag_hip_parcsr_t *out);
/* SharingMap::IgnoreNonLocal(data, trueData) (SharingMap.cpp:726-742): y = diag(A)^T x, no exchange.  The other two
 * vector forms are plain products with entity_trueEntity: Assemble = elag_hip_parcsr_spmv(trans = 1) (:768-780),
 * Distribute = elag_hip_parcsr_spmv(trans = 0) (:664-677). */
ELAG_HIP_API int elag_hip_parcsr_spmvT_ignore_nonlocal(elag_hip_parcsr_t A, const double *x, double *y, int loc);
/* hypre_ParCSRMatrixAdd2: C = a*A + b*B (src/hypreExtension/parcsr-add.c:199-245; :27-78 is a=b=1) */
ELAG_HIP_API int elag_hip_parcsr_add(double a, elag_hip_parcsr_t A, double b, elag_hip_parcsr_t B,
                                     elag_hip_parcsr_t *C);
/* hypre_RDP: RT^T diag(d) P (par_Tmatmul.c:17-39); d is a host or device vector of local fine rows */
ELAG_HIP_API int elag_hip_parcsr_rdp(elag_hip_parcsr_t RT, const double *d, int loc, elag_hip_parcsr_t P,
                                     elag_hip_parcsr_t *RDP);
/* hypre_ParCSRMatrixFixZeroRows (Hierarchy.cpp:366-371,515-519; HiptmairSmootherFactory.cpp:162) */
ELAG_HIP_API int elag_hip_parcsr_fix_zero_rows(elag_hip_parcsr_t A, int *num_fixed);
/* hypre_ParCSRMatrixDeleteZeros (src/hypreExtension/deleteZeros.c:16-47): drop |a| <= tol */
ELAG_HIP_API int elag_hip_parcsr_delete_zeros(elag_hip_parcsr_t A, double tol);
/* hypre_ParCSRDataTransformationSign (hypre_ParCSRDataTransformationSign.c:29-33) */
ELAG_HIP_API int elag_hip_parcsr_sign_transform(elag_hip_parcsr_t A);
/* hypre_IdentityParCSRMatrix / hypre_DiagonalParCSRMatrix (hypre_CSRFactory.c:16-250); d may be NULL (=1) */
ELAG_HIP_API int elag_hip_parcsr_diagonal(elag_hip_parcsr_t *out, int num_rows, int64_t first_row,
                                          int64_t global_num_rows, const double *d_host);
/* Norms (hypre_ParCSRMatrixNorms.c:18-160): kind 0 = l1 (max col sum), 1 = linf (max row sum),
 * 2 = max |a_ij|, 3 = Frobenius.  Ends in the communicator's 8-byte all-reduce. */
ELAG_HIP_API int elag_hip_parcsr_norm(elag_hip_parcsr_t A, int kind, double *value);
/* hypre_ParCSRMatrixCompare (hypre_ParCSRMatrixCompare.c:125-195): bit mask 1..32 shape, 64 = |A-B|max>tol */
ELAG_HIP_API int elag_hip_parcsr_compare(elag_hip_parcsr_t A, elag_hip_parcsr_t B, double tol, int *mask);
/* Tuning knob: lanes of a wavefront that cooperate on one row in the SpMV kernels (power of two 1..64;
 * 0 restores the size heuristic; -2 forces the LDS-staged stream kernel, -1 its windowed x-cache form).
 * No reference counterpart. */
ELAG_HIP_API int elag_hip_parcsr_set_spmv_lanes(elag_hip_parcsr_t A, int lanes);
/* Which SpMV kernel the diag block of A is dispatched to as of now: *form = 0 row-group kernel (spmv_fused_kernel; *lanes =
 * lanes per row), 1 LDS-staged stream kernel (spmv_stream_kernel), 2 its windowed x-cache form (spmv_xc_kernel).  The x-cache
 * decision is taken on the first launch (or at hierarchy Finalize); before that a stream-kernel matrix reports 1.
 * Diagnostic (bench.py names the kernel it timed from this, not from a constant).  No reference counterpart. */
ELAG_HIP_API int elag_hip_parcsr_get_spmv_form(elag_hip_parcsr_t A, int *form, int *lanes);
/* a_ii of every local row -> d (device or host vector of num_rows) */
ELAG_HIP_API int elag_hip_parcsr_get_diagonal(elag_hip_parcsr_t A, double *d, int loc);

/* hypre_ParCSRMatrixMatvecBoolInt (src/hypreExtension/hypre_ParCSRMatrixMatvecBoolInt.c:140-202): integer
 * y = alpha*|A|_bool*x + beta*y, values of A ignored (pattern = 1), same halo exchange with comm job 11. */
ELAG_HIP_API int elag_hip_parcsr_spmv_boolint(elag_hip_parcsr_t A, int alpha, const int *x, int beta, int *y, int loc);
/* hypre_ParCSRMatrixKeepEqualEntries (src/hypreExtension/deleteZeros.c:113-200): A and B share a pattern; entries
 * of A equal to B's (|a-b| < 1e-9) become 1, the others are dropped.  Overwrites A. */
ELAG_HIP_API int elag_hip_parcsr_keep_equal_entries(elag_hip_parcsr_t A, elag_hip_parcsr_t B);

/* --------------------------------------------------------------------------------------------------
 * Agglomerate index maps.  Integer setup logic that stays on the host side of the boundary
 * (north star: "host code stays C++"); results are bit-exact with the reference's routines.  No GPU needed.
 * ------------------------------------------------------------------------------------------------ */
/* MFEMRefinedMeshPartitioner::Partition (src/partitioning/MFEMRefinedMeshPartitioner.cpp:50-95).
 * legacy_numbering = 0: MFEM >= 4.1 branch (:64-70); != 0: parent-then-children numbering (:71-91). */
ELAG_HIP_API int elag_hip_partition_refined(int num_elements, int num_parts, int num_dimensions, int legacy_numbering,
                                            int *partitioning);
/* connectedComponents (src/structures/connectedComponents.cpp:23-85): splits every part into its connected
 * components of the graph (conn_i, conn_j); new ids = offset[part] + component, components numbered by their first
 * node.  All part ids must be >= 0. */
ELAG_HIP_API int elag_hip_connected_components(int num_nodes, int *partitioning, const int *conn_i, const int *conn_j,
                                               int *num_components);
/* transposePartitioning (src/structures/transpose.cpp:27-55) and TransposeOrientation
 * (src/topology/TopologyTable.cpp:218-254; same table, entries -1 skipped, unit data): element->part array to
 * part->elements CSR (i_out: num_parts+1 entries, j_out: number of entries with part >= 0; elements ascending inside a
 * part). */
ELAG_HIP_API int elag_hip_transpose_partitioning(int num_elements, const int *partitioning, int num_parts, int *i_out,
                                                 int *j_out);

/* --------------------------------------------------------------------------------------------------
 * Smoothers  (HypreSmootherWrapper -> mfem::HypreSmoother -> hypre_ParCSRRelax;
 * src/linalg/solver_ops/ParELAG_HypreSmootherWrapper.cpp:20-35, .hpp:69-83)
 * Parameters are the ParameterList keys "Sweeps", "Damping Factor", "Omega", "Cheby Poly Order",
 * "Cheby Poly Fraction" (HypreSmootherWrapper.cpp:28-33) with the same defaults 1, 1.0, 1.0, 2, 0.3.
 * ------------------------------------------------------------------------------------------------ */
ELAG_HIP_API int elag_hip_smoother_create(elag_hip_smoother_t *out, elag_hip_parcsr_t A, int relax_type,
                                          int sweeps, double damping, double omega,
                                          int cheby_order, double cheby_fraction);
/* Chebyshev: eigenvalue bounds of D^-1/2 A D^-1/2.  create() runs hypre_ParCSRMaxEigEstimateCG's
 * algorithm (10 CG-Lanczos steps, hypre RNG seed 1); set() injects bounds (parity of the sweep).  A hierarchy composes the
 * order-2 sweep into its level operators at Finalize: after set() on a smoother that sits in a finalised hierarchy, finalise again. */
/* Hybrid Gauss-Seidel on "virtual ranks".  hypre's relax types 2 / 4 / 6 are Gauss-Seidel inside an MPI rank and Jacobi across ranks (the
 * neighbours' values stay as they were when the symmetric sweep began: hypre_ParCSRRelax / hypre_BoomerAMGRelax; the reference picks them in
 * src/linalg/factories/ParELAG_HypreSmootherFactory.cpp:92-109 and runs one rank per core), so what a sweep computes depends on how many ranks the
 * reference runs on.  nblocks = 1 (the default; also ELAG_HIP_GS_BLOCKS) is one rank: the sequential loop, bit for bit.  nblocks > 1 makes the sweep
 * smoothers created AFTER the call (elag_hip_smoother_create, elag_hip_hierarchy_build; rank-local operators) cut their rows into nblocks
 * contiguous, even blocks -- rows [b*size + min(b, rest), ...), size = n / nblocks, rest = n % nblocks -- and sweep them as that many ranks would:
 * the bits of the reference's loop on that row partition, with dependency chains nblocks times shorter.  Per host thread (like the library
 * context); 0 restores the default. */
ELAG_HIP_API int elag_hip_set_gs_hybrid_blocks(int nblocks);
ELAG_HIP_API int elag_hip_smoother_get_eig_estimate(elag_hip_smoother_t S, double *max_eig, double *min_eig);
ELAG_HIP_API int elag_hip_smoother_set_eig_estimate(elag_hip_smoother_t S, double max_eig, double min_eig);
/* HiptmairSmoother(A, D, primary, auxiliary) (ParELAG_HiptmairSmoother.cpp:22-46); A_aux = D^T A D is owned
 * by `auxiliary`.  The result is an elag_hip_smoother_t and can be used wherever a smoother can. */
ELAG_HIP_API int elag_hip_hiptmair_create(elag_hip_smoother_t *out, elag_hip_parcsr_t A, elag_hip_parcsr_t D,
                                          elag_hip_smoother_t primary, elag_hip_smoother_t auxiliary);
/* sol <- smoother(rhs, sol).  zero_initial_guess != 0 is mfem::Solver::iterative_mode == false
 * (sol is ignored on entry).  transpose != 0 calls MultTranspose (HiptmairSmoother.cpp:79-109; the hypre
 * relaxations handled here are symmetric, so it equals Mult for them). */
ELAG_HIP_API int elag_hip_smoother_apply(elag_hip_smoother_t S, const double *rhs, double *sol,
                                         int zero_initial_guess, int transpose, int loc);
ELAG_HIP_API int elag_hip_smoother_destroy(elag_hip_smoother_t S);

/* --------------------------------------------------------------------------------------------------
 * Coarse solver.  The reference puts any SolverFactory product here (AMGeSolverFactory.cpp:144-163;
 * shipped lists use PCG+BoomerAMG, Create0FormParameterList.hpp:87-136).  kind 0: dense Cholesky of the
 * (<= 4096 dof, SPD) coarsest operator, applied on device as a dense inverse; kind 1: dense LU with partial
 * pivoting (Gauss-Jordan on the device) for indefinite operators such as the blocked saddle-point systems, where the
 * reference's lists name a "Direct" solver (ParELAG_DirectSolverFactory.cpp) or a short GMRES; `iters` is unused.
 * A host that wants hypre there exports the operator instead.
 * ------------------------------------------------------------------------------------------------ */
ELAG_HIP_API int elag_hip_coarse_create(elag_hip_coarse_t *out, elag_hip_parcsr_t A, int kind, int iters);
/* Multi-rank coarse solve: every rank holds the whole (tiny) coarsest operator `A_global` as a single-rank matrix,
 * the distributed right-hand side is all-gathered over RCCL and each rank applies its own rows of the dense
 * inverse (SURVEY 8e "gather the coarsest levels onto every GPU").  row_starts has nranks+1 entries. */
ELAG_HIP_API int elag_hip_coarse_create_replicated(elag_hip_coarse_t *out, elag_hip_parcsr_t A_global,
                                                   const int64_t *row_starts, int kind);
/* A whole (rank-replicated) hierarchy as the coarse solver of a distributed one: every rank holds the same
 * sub-hierarchy of the levels that are too small to be worth a halo exchange (SURVEY 8e "gather the coarsest levels onto
 * every GPU").  With allreduce_rhs != 0 the right-hand side handed to apply() is this rank's partial sum of the
 * restricted residual (P stored with global coarse columns) and is summed over the ranks with one RCCL all-reduce
 * before the replicated cycle; the solution is complete on every rank, so the prolongation needs no exchange.
 * A distributed V-cycle then costs three RCCL rounds (two halos on the fine level + this all-reduce). */
ELAG_HIP_API int elag_hip_coarse_create_from_hierarchy(elag_hip_coarse_t *out, elag_hip_hierarchy_t sub, int allreduce_rhs);
ELAG_HIP_API int elag_hip_coarse_apply(elag_hip_coarse_t C, const double *rhs, double *sol, int loc);
ELAG_HIP_API int elag_hip_coarse_destroy(elag_hip_coarse_t C);

/* --------------------------------------------------------------------------------------------------
 * Hierarchy  (parelag::Hierarchy, src/linalg/solver_ops/ParELAG_Hierarchy.cpp:109-267)
 * Level keys follow Level::Set in the reference: level l holds "A", "PreSmoother", "PostSmoother";
 * level l >= 1 holds "P" (maps level l to level l-1); the last level holds "CoarseSolver".
 * Handles are shared, not owned (ParELAG_Level.hpp:155-176).
 * ------------------------------------------------------------------------------------------------ */
ELAG_HIP_API int elag_hip_hierarchy_create(elag_hip_hierarchy_t *out, int num_levels);
ELAG_HIP_API int elag_hip_hierarchy_set_operator(elag_hip_hierarchy_t H, int level, elag_hip_parcsr_t A);
ELAG_HIP_API int elag_hip_hierarchy_set_prolongator(elag_hip_hierarchy_t H, int coarse_level, elag_hip_parcsr_t P);
ELAG_HIP_API int elag_hip_hierarchy_set_smoothers(elag_hip_hierarchy_t H, int level,
                                                  elag_hip_smoother_t pre, elag_hip_smoother_t post);
ELAG_HIP_API int elag_hip_hierarchy_set_coarse_solver(elag_hip_hierarchy_t H, elag_hip_coarse_t C);
/* Hierarchy::Finalize (:255-267): size the per-level vectors; also builds explicit P^T and, when
 * use_graph != 0, arms hipGraph capture of the whole cycle (launch-bound small levels). */
ELAG_HIP_API int elag_hip_hierarchy_finalize(elag_hip_hierarchy_t H, int use_graph);
/* How Finalize decided to issue a cycle when use_graph was asked for: a replayed hipGraph costs the device a fixed few microseconds per
 * launch on top of its nodes and saves the host ~2.5 us per kernel, so the rule is the kernel count of one cycle -- plain stream launches
 * up to 32 kernels, a replayed graph beyond (the Gauss-Seidel chains).  ELAG_HIP_GRAPH_MODE=graph | eager forces either, =measure times
 * one cycle both ways at Finalize (host clock, queue drained) and keeps the plain launches when they are at least 3 % faster.
 * uses_graph: the outcome; kernels: graph nodes of one cycle (0: not counted); graph_us / launches_us: the two timings (0 unless measured). */
ELAG_HIP_API int elag_hip_hierarchy_launch_mode(elag_hip_hierarchy_t H, int *uses_graph, int *kernels, double *graph_us, double *launches_us);
/* Wall time of the hierarchy's setup by phase, as "name=milliseconds;name=milliseconds;..." (what the reference prints under
 * "Build Hierarchy", src/linalg/factories/ParELAG_AMGeSolverFactory.cpp:48-49,130-135, broken down): galerkin_products, renumbering,
 * smoothers, coarse_solver (elag_hip_hierarchy_build), fused_level_operators, spmv_forms, dense_tail, flat_cycle_operators, launch_mode
 * (Finalize).  Every phase ends with a drain of the compute stream. */
ELAG_HIP_API int elag_hip_hierarchy_setup_breakdown(elag_hip_hierarchy_t H, char *buf, int len);
/* Hierarchy::Mult (:109-136): is_preconditioner != 0 -> sol = cycle(rhs); else r = rhs - A sol,
 * sol += cycle(r).  mu = 0 (V-cycle) only, as in the reference. */
ELAG_HIP_API int elag_hip_hierarchy_apply(elag_hip_hierarchy_t H, const double *rhs, double *sol,
                                          int is_preconditioner, int loc);
ELAG_HIP_API int elag_hip_hierarchy_destroy(elag_hip_hierarchy_t H);

/* buildHierarchyFromDeRhamSequence + AMGeSolverFactory::_do_build_solver in one call
 * (Hierarchy.cpp:282-383, AMGeSolverFactory.cpp:26-169): A_{l+1} = P_l^T A_l P_l, optional FixZeroRows
 * (essential BCs), a smoother of `relax_type` on every non-coarsest level (pre = post), coarse solve.
 * P[l] maps level l+1 -> l.  The hierarchy owns everything it builds; level operators are retrievable.
 * The option fields are the AMGe / smoother ParameterList entries of the reference plus two device knobs. */
typedef struct {
    int fix_zero_rows;      /* hypre_ParCSRMatrixFixZeroRows on every coarse operator (Hierarchy.cpp:366-371)     */
    int relax_type;         /* "Smoother" type (HypreSmootherFactory.cpp:92-109): 0, 1 or 16                      */
    int sweeps;             /* "Sweeps"                                                                            */
    double damping;         /* "Damping Factor"                                                                    */
    int cheby_order;        /* "Cheby Poly Order"                                                                  */
    double cheby_fraction;  /* "Cheby Poly Fraction"                                                               */
    int coarse_kind;        /* "Coarse solver": 0 = dense Cholesky                                                 */
    int coarse_iters;
    int use_graph;          /* capture the cycle in a hipGraph (single rank)                                       */
    int reorder;            /* renumber the dofs of every level for cache locality of the x gathers: children are
                               numbered contiguously under their strongest coarse parent (from P), the coarsest
                               level by Cuthill-McKee.  -1 = automatic (on when level 0 has >= 2^18 rows), 0 = off,
                               1 = on.  Purely internal: vectors cross the ABI in the caller's numbering; operators
                               returned by elag_hip_hierarchy_get_operator are in the internal numbering.        */
} elag_hip_amge_options;
ELAG_HIP_API int elag_hip_amge_default_options(elag_hip_amge_options *opt);
ELAG_HIP_API int elag_hip_hierarchy_build(elag_hip_hierarchy_t *out, elag_hip_parcsr_t A, int num_P,
                                          const elag_hip_parcsr_t *P, const elag_hip_amge_options *opt);
/* Same, with a HiptmairSmoother on every non-coarsest level ("PreSmoother"/"PostSmoother" = a "Hiptmair" entry:
 * examples/testing_helpers/Create2FormParameterList.hpp:78-133).  D[l] is the true exterior derivative into the
 * form of level l (HiptmairSmootherFactory.cpp:84-103); A_aux = D^T A D + FixZeroRows is formed on the device
 * (:143-165); `opt` configures the hierarchy and the primary smoother, `aux_opt` (relax_type, sweeps, damping,
 * Chebyshev fields) the auxiliary one.  Reordering is not applied to Hiptmair hierarchies. */
ELAG_HIP_API int elag_hip_hierarchy_build_hiptmair(elag_hip_hierarchy_t *out, elag_hip_parcsr_t A, int num_P,
                                                   const elag_hip_parcsr_t *P, const elag_hip_parcsr_t *D,
                                                   const elag_hip_amge_options *opt, const elag_hip_amge_options *aux_opt);
/* new -> old row map of a level (identity when the hierarchy was not reordered); `order` has num_rows entries */
ELAG_HIP_API int elag_hip_hierarchy_get_ordering(elag_hip_hierarchy_t H, int level, int *order);
ELAG_HIP_API int elag_hip_hierarchy_get_operator(elag_hip_hierarchy_t H, int level, elag_hip_parcsr_t *A);
/* A finalized hierarchy wrapped as a solver object, for the places where a parameter list names an AMGe hierarchy as an
 * inner inverse: "A00_1 Inverse" = "AMGe-GS" of the Block LDU list (example_parameterlists/spe10_example_parameters.xml:114-127,
 * src/linalg/factories/ParELAG_Block2x2LDUSolverFactory.cpp).  Pass the handle as an entry of `inverses` to
 * elag_hip_block_smoother_create or as a ELAG_HIP_PREC_SMOOTHER preconditioner.  scale multiplies the result (-1: a hierarchy
 * built on  B D^-1 B^T  applied as the inverse of the saddle point's Schur complement  -B D^-1 B^T).  The hierarchy is
 * borrowed; destroy the wrapper with elag_hip_smoother_destroy. */
ELAG_HIP_API int elag_hip_smoother_from_hierarchy(elag_hip_smoother_t *out, elag_hip_hierarchy_t H, double scale);
ELAG_HIP_API int elag_hip_hierarchy_get_smoother(elag_hip_hierarchy_t H, int level, elag_hip_smoother_t *S);

/* --------------------------------------------------------------------------------------------------
 * Blocked (saddle-point) layer -- SURVEY 8f-2: MultigridTestDarcy / SPE10 monolithic AMGe.
 *
 * elag_hip_blockop_t mirrors parelag::MfemBlockOperator (src/linalg/utilities/ParELAG_MfemBlockOperator.hpp:30-135,
 * .cpp:93-130): a num_block_rows x num_block_cols array of ParCSR blocks (NULL = zero block) over contiguous block
 * vectors with the given LOCAL offsets (mfem::BlockVector).  On the device the blocks are additionally merged into one
 * "monolithic" ParCSR (block row i / column j at local offsets row_offsets[i] / col_offsets[j]; across ranks, rank r
 * owns the contiguous global range made of its share of every block, in block order), so that Mult / MultTranspose and
 * every V-cycle product with the operator are ONE fused SpMV launch and ONE halo exchange.  Blocks are borrowed.
 * ------------------------------------------------------------------------------------------------ */
ELAG_HIP_API int elag_hip_blockop_create(elag_hip_blockop_t *out, int num_block_rows, int num_block_cols,
                                         const int *row_offsets, const int *col_offsets);
/* MfemBlockOperator::SetBlock (.cpp:107-128): size-checked against the offsets */
ELAG_HIP_API int elag_hip_blockop_set_block(elag_hip_blockop_t B, int block_row, int block_col, elag_hip_parcsr_t A);
/* GetBlock / IsZeroBlock: *A = NULL for a zero block */
ELAG_HIP_API int elag_hip_blockop_get_block(elag_hip_blockop_t B, int block_row, int block_col, elag_hip_parcsr_t *A);
/* the merged operator (owned by B, rebuilt after SetBlock); usable wherever a ParCSR handle is (hierarchy levels ...) */
ELAG_HIP_API int elag_hip_blockop_monolithic(elag_hip_blockop_t B, elag_hip_parcsr_t *A);
/* y = alpha*op(A)*x + beta*y on block vectors (MfemBlockOperator::Mult / MultTranspose, .cpp:93-103) */
ELAG_HIP_API int elag_hip_blockop_mult(elag_hip_blockop_t B, int trans, double alpha, const double *x, double beta,
                                       double *y, int loc);
ELAG_HIP_API int elag_hip_blockop_destroy(elag_hip_blockop_t B);

/* SchurComplementFactory::BuildOperator (src/linalg/factories/ParELAG_SchurComplementFactory.cpp:36-181):
 *   S = A11 - alpha * A10 * D^-1 * A01,  D = diag(A00) ("DIAGONAL") or the absolute row sums of A00 ("ABSROWSUM");
 * negate != 0 returns -S ("Use Negative S", Block2x2JacobiSolverFactory.cpp:83-88).  "MASS" needs the DeRhamSequence
 * and stays on the host side (the caller passes ComputeTrueM(form) as A11 instead).  The caller owns *S. */
enum { ELAG_HIP_SCHUR_NONE = 0, ELAG_HIP_SCHUR_DIAGONAL = 1, ELAG_HIP_SCHUR_ABSROWSUM = 2 };
ELAG_HIP_API int elag_hip_schur_complement(elag_hip_blockop_t B, int s_type, double alpha, int negate,
                                           elag_hip_parcsr_t *S);

/* Block smoothers (all are elag_hip_smoother_t: apply / destroy / hierarchy_set_smoothers work on them):
 *   ELAG_HIP_BLOCK_JACOBI    BlockDiagonalSolver::Mult        (solver_ops/ParELAG_BlockDiagonalSolver.cpp:67-112)
 *   ELAG_HIP_BLOCK_GS_LOWER / _UPPER  BlockTriangularSolver   (solver_ops/ParELAG_BlockTriangularSolver.cpp:73-300)
 *       -- `inverses` holds one smoother per diagonal block (built on A_ii, or on S for the last block);
 *   ELAG_HIP_BLOCK_LDU       Block2x2LDUInverseOperator::Mult (solver_ops/ParELAG_Block2x2LDUInverseOperator.cpp:73-137)
 *       -- `inverses` = {A00_1, A00_2, A00_3, S} and `damping` is its "Damping Factor".
 * The inner smoothers run from the zero guess (iterative_mode == false, as the factories set them); the block smoother
 * itself honours zero_initial_guess like every other smoother.  Inverses are borrowed. */
enum { ELAG_HIP_BLOCK_JACOBI = 0, ELAG_HIP_BLOCK_GS_LOWER = 1, ELAG_HIP_BLOCK_GS_UPPER = 2, ELAG_HIP_BLOCK_LDU = 3 };
ELAG_HIP_API int elag_hip_block_smoother_create(elag_hip_smoother_t *out, elag_hip_blockop_t B, int kind,
                                                int num_inverses, const elag_hip_smoother_t *inverses, double damping);

/* buildBlockedHierarchyFromDeRhamSequence (src/linalg/solver_ops/ParELAG_Hierarchy.cpp:400-541) + the block-smoother
 * factories (factories/ParELAG_Block2x2{Jacobi,GaussSeidel,LDU}SolverFactory.cpp) on the device: per block
 * A_c[i][j] = P_i^T A[i][j] P_j for the stored blocks, FixZeroRows on the diagonal blocks flagged in fix_zero_rows
 * (NULL = none; the "label_ess[row].size() > 0" test of :515), P = blockdiag(P_i); on every level but the coarsest
 * the block smoother named in the options (A00 inverse on A00, A11 inverse on S or A11), on the coarsest a dense
 * direct solve of the merged operator.  P holds num_P * num_blocks handles, P[l*num_blocks + i] = block i of the
 * interpolator from level l+1 to level l.  The hierarchy borrows A and P and owns everything it builds. */
typedef struct {
    int block_smoother;             /* ELAG_HIP_BLOCK_*                                            */
    elag_hip_amge_options inv_a00;  /* "A00 Inverse": relax_type / sweeps / damping / cheby_* used */
    elag_hip_amge_options inv_a11;  /* "A11 Inverse" (applied to S when s_type != NONE)            */
    int s_type;                     /* "S Type": ELAG_HIP_SCHUR_*                                  */
    double alpha;                   /* "Alpha"                                                     */
    int use_negative_s;             /* "Use Negative S"                                            */
    double damping;                 /* "Damping Factor" (LDU)                                      */
    int coarse_kind;                /* 1 = dense LU (saddle point), 0 = Cholesky                   */
    int use_graph;
} elag_hip_blocked_options;
ELAG_HIP_API int elag_hip_blocked_default_options(elag_hip_blocked_options *opt);
ELAG_HIP_API int elag_hip_hierarchy_build_blocked(elag_hip_hierarchy_t *out, elag_hip_blockop_t A, int num_P,
                                                  const elag_hip_parcsr_t *P, const int *fix_zero_rows,
                                                  const elag_hip_blocked_options *opt);

/* --------------------------------------------------------------------------------------------------
 * Krylov shells around the V-cycle -- SURVEY 8f-4.
 *
 * parelag::KrylovSolver (src/linalg/solver_ops/ParELAG_KrylovSolver.cpp:23-96) wraps mfem::CGSolver / GMRESSolver /
 * FGMRESSolver chosen by "Solver name" with "Relative tolerance", "Absolute tolerance", "Maximum iterations",
 * "Restart size"; parelag::StationarySolver (solver_ops/ParELAG_StationarySolver.cpp:41-130) is the preconditioned
 * Richardson loop with "Relative Tolerance" / "Absolute Tolerance" / "Maximum Iterations".  The device versions follow
 * the MFEM 4.x iterations statement by statement (same convergence tests, final_iter / final_norm / converged), with
 * every scalar of the recurrence kept on the device: the host enqueues iterations without reading anything back and
 * polls a `done` flag every `check_every` iterations; kernels after convergence are predicated off, so iterate and
 * iteration count equal the sequential algorithm's.  The preconditioner is any object of this library and is applied
 * from the zero guess (KrylovSolver.cpp:79-83 sets iterative_mode = false on it).  A and the preconditioner are
 * borrowed.  For a blocked system pass elag_hip_blockop_monolithic(B) as A.
 * ------------------------------------------------------------------------------------------------ */
enum { ELAG_HIP_KRYLOV_CG = 0, ELAG_HIP_KRYLOV_GMRES = 1, ELAG_HIP_KRYLOV_FGMRES = 2, ELAG_HIP_KRYLOV_STATIONARY = 3,
       ELAG_HIP_KRYLOV_BICGSTAB = 4, ELAG_HIP_KRYLOV_MINRES = 5 };   /* "BICGSTAB", "MINRES": KrylovSolver.cpp:39-42 */
enum { ELAG_HIP_PREC_NONE = 0, ELAG_HIP_PREC_SMOOTHER = 1, ELAG_HIP_PREC_HIERARCHY = 2, ELAG_HIP_PREC_COARSE = 3 };
typedef struct {
    int solver;        /* ELAG_HIP_KRYLOV_*  ("Solver name")                       */
    int max_iter;      /* "Maximum iterations" (default 10, KrylovSolver.cpp:92-93) */
    double rel_tol;    /* "Relative tolerance"                                      */
    double abs_tol;    /* "Absolute tolerance"                                      */
    int restart;       /* "Restart size" (GMRES / FGMRES, default 50)               */
    int check_every;   /* iterations enqueued between two reads of the done flag    */
    int use_graph;     /* CG: replay one captured hipGraph per iteration            */
} elag_hip_krylov_options;
ELAG_HIP_API int elag_hip_krylov_default_options(elag_hip_krylov_options *opt);
/* prec: elag_hip_smoother_t / elag_hip_hierarchy_t / elag_hip_coarse_t according to prec_kind, or NULL */
ELAG_HIP_API int elag_hip_krylov_create(elag_hip_krylov_t *out, elag_hip_parcsr_t A, int prec_kind, void *prec,
                                        const elag_hip_krylov_options *opt);
/* mfem::IterativeSolver::Mult(b, x): iterative_mode != 0 starts from x, otherwise from zero */
ELAG_HIP_API int elag_hip_krylov_mult(elag_hip_krylov_t K, const double *b, double *x, int iterative_mode, int loc);
/* GetConverged / GetNumIterations / GetFinalNorm of the last Mult */
ELAG_HIP_API int elag_hip_krylov_get_stats(elag_hip_krylov_t K, int *converged, int *final_iter, double *final_norm);
/* How the last Mult issued its iterations: "graph" (a CG iteration / a GMRES inner step replays as one hipGraph), "plain" (kernel by
 * kernel: distributed operators, solvers whose apply is not pure stream work, ELAG_HIP_GMRES_GRAPH=0, use_graph == 0) or
 * "plain: <why a capture was given up>".  Diagnostic only -- no reference counterpart (mfem::IterativeSolver has no launch mode); the
 * iterates are the same either way.  The string lives as long as the solver. */
ELAG_HIP_API const char *elag_hip_krylov_launch_mode(elag_hip_krylov_t K);
ELAG_HIP_API int elag_hip_krylov_destroy(elag_hip_krylov_t K);

/* --------------------------------------------------------------------------------------------------
 * AMGe coarse-space construction: the per-agglomerate dense work of DeRhamSequence::Coarsen
 * (src/amge/DeRhamSequence.cpp:572-692) for the (H(div), L2) pair, i.e. the path taken with jformStart_ = nDim-1
 * (examples/MultigridTestSPE10.cpp:441).  The host keeps the integer tables the reference builds in
 * AgglomeratedTopology::CoarsenLocalPartitioning and DofAgglomeration (which entity lies in which agglomerate, local
 * dof numbering with interior dofs first); the device does every flop, one workgroup per agglomerated entity:
 *   elag_hip_coarsen_traces      loop body of ComputeCoarseTracesWithTargets (DeRhamSequence.cpp:1818-1926) --
 *                                local mass assembly (AssembleAgglomerateMatrix, DOFAgglomeration.cpp:533-546), PV
 *                                deflation (InnerProduct.cpp:157-169), SVD_Calculator::ComputeON in the mass inner
 *                                product (ParELAG_SVDCalculator.cpp:192-284), coarse mass, CochainProjector::
 *                                CreateDofFunctional (CochainProjector.cpp:53-137);
 *   elag_hip_coarsen_extension   loop body of hFacetExtension (DeRhamSequence.cpp:2364-2556) -- FacetSaddlePoint
 *                                (ParELAG_SaddlePointSolver.cpp:70-130) factorised once per agglomerate and solved for
 *                                the trace extensions, the RangeT bubbles and the target extensions; coarse derivative
 *                                row, null-space dofs by SVD, dof functional, coarse element mass.
 * All arrays are HOST pointers (copied in / out); dense blocks are packed one after the other.  "col-major n x m" =
 * m columns of length n.  Per-entity output slots have the maximal size (1 + nT trace dofs, nT null dofs); the int
 * outputs say how much of a slot is used.  Status ELAG_HIP_ERR_NUMERIC when a local system is singular ("bad topology,
 * such as a torus-shaped agglomerate", ParELAG_SaddlePointSolver.cpp:117-126).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    /* entity e owns items asm_ptr[e] .. asm_ptr[e+1]-1; item t adds the row-major block of order blk_n[t] stored at
     * blocks + blk_off[t] into the entity's matrix at the local indices loc_idx[loc_off[t] .. loc_off[t] + blk_n[t]) */
    const int64_t *asm_ptr;
    const int64_t *blk_off;
    const int *blk_n;
    const int64_t *loc_off;
    const int *loc_idx;
    const double *blocks;
    int64_t n_items, n_loc, n_block_values;
} elag_hip_dense_assembly;

typedef struct {
    int n_ent, nT;
    double svd_tol;               /* SVD_Tolerance_ (relative to (pv, pv)_M)                    */
    const int *ndof;              /* local dofs per entity                                       */
    elag_hip_dense_assembly mass; /* entity mass matrix                                          */
    const double *pv;             /* packed, ndof[e] values per entity                           */
    const double *T;              /* packed, col-major ndof[e] x nT per entity                   */
} elag_hip_coarsen_traces_in;
typedef struct {
    int *ncoarse;                 /* n_ent: 1 + kept target directions                           */
    double *p_loc;                /* col-major ndof[e] x (1+nT) per entity                       */
    double *cmass;                /* row-major (1+nT) x (1+nT) per entity, leading ncoarse used  */
    double *functional;           /* row-major (1+nT) x ndof[e] per entity, first ncoarse rows   */
} elag_hip_coarsen_traces_out;
ELAG_HIP_API int elag_hip_coarsen_traces(const elag_hip_coarsen_traces_in *in, const elag_hip_coarsen_traces_out *out,
                                         double *kernel_ms /* may be NULL: device time of the batch */);

typedef struct {
    int n_ent, nT;
    int mode;                     /* 0: hFacetExtension (FacetSaddlePoint, PV multiplier row);
                                     1: hRidgePeakExtension (:2629-3055; RidgePeakSaddlePoint [M B^T; B -C],
                                        ParELAG_SaddlePointSolver.cpp:150-189, -C = -(D^T W D) of the next form, GetMinusC) */
    int null_dofs;                /* build null-space dofs from the targets (0 for the "peak" stage, :2924) */
    double svd_tol;               /* absolute (DeRhamSequence.cpp:2498)                          */
    double smallest_entry;        /* SmallestEntry_ (:2403)                                      */
    const int *nu, *ni, *np;      /* u dofs (interior first), interior u dofs, p dofs per entity */
    const int *npi, *nq;          /* mode 1: interior p dofs (first in the p numbering), dofs of form j+2 */
    const int *next, *nrt;        /* coarse dofs on the boundary, RangeT bubbles per entity      */
    elag_hip_dense_assembly mass_u, mass_p;
    elag_hip_dense_assembly mass_q; /* mode 1: mass of form j+2 on the entity                    */
    const int64_t *d_ptr;         /* local derivative D (np x nu) as triplets d_ptr[e]..         */
    const int *d_row, *d_col;
    const double *d_val;
    const int64_t *d2_ptr;        /* mode 1: local derivative of form j+1 (nq x np)              */
    const int *d2_row, *d2_col;
    const double *d2_val;
    const double *pv_p;           /* packed np[e]: P_3(fine p dofs, PV dof of the agglomerate)   */
    const double *R_b;            /* packed col-major (nu-ni) x next: traces on the boundary     */
    const double *cP;             /* packed col-major npi x nrt: null-space coarse functions of form j+1 interior to the entity */
    const double *T;              /* packed col-major nu x nT: targets                           */
    const double *PDc;            /* mode 1: packed col-major np x next: (P_{j+1} D_c)(p dofs, coarse boundary dofs), :2696-2697 */
} elag_hip_coarsen_extension_in;
typedef struct {
    int *n_null;                  /* n_ent                                                       */
    double *ext;                  /* col-major ni x next                                         */
    double *dc;                   /* next: entries of the coarse derivative row of the PV dof    */
    double *rt;                   /* col-major ni x nrt                                          */
    double *nul;                  /* col-major ni x nT, first n_null columns used                */
    double *functional;           /* row-major (nrt+nT) x ni, first nrt+n_null rows used         */
    double *cmass;                /* row-major of order next+nrt+n_null, stored with ld next+nrt+nT */
} elag_hip_coarsen_extension_out;
ELAG_HIP_API int elag_hip_coarsen_extension(const elag_hip_coarsen_extension_in *in, const elag_hip_coarsen_extension_out *out,
                                            double *kernel_ms);

/* Ranks of a batch of small matrices given as triplets (matrix e: nrows[e] x ncols[e], triplets t_ptr[e] .. t_ptr[e+1]-1): the
 * DenseMatrix::Rank(1e-9) calls of AgglomeratedTopologyCheck::computeBettiNumbersAgglomeratedEntities
 * (src/topology/AgglomeratedTopologyCheck.cpp:241-318: local boundary operators of every agglomerated entity), one workgroup
 * per matrix, Gaussian elimination with row pivoting; pivots <= tol end a column.  Host arrays in, host ranks out. */
ELAG_HIP_API int elag_hip_batched_rank(int n_mat, const int *nrows, const int *ncols, const int64_t *t_ptr, const int *t_row, const int *t_col,
                                       const double *t_val, double tol, int *rank);

/* --------------------------------------------------------------------------------------------------
 * Host integer tables of the coarsening (csrc/host_tables.hip; no device work, callable before elag_hip_init).  Ragged tables
 * are (ptr[n+1], idx[ptr[n]]) with int64 entries.
 * ------------------------------------------------------------------------------------------------ */
/* Row-by-row product of two ragged tables with the columns of a row in first-touch order: mfem::Mult on patterns as
 * AgglomeratedTopology / DofAgglomeration use it; with a_sgn, b_sgn, out_sgn given it is MultOrientation
 * (src/topology/TopologyTable.cpp:130-139): signed sums, cancelled entries dropped, values reduced to their sign.
 * out_idx (and out_sgn) need room for sum over the left entries of the right rows' lengths; *out_nnz = entries written. */
ELAG_HIP_API int elag_hip_host_table_product(int64_t nrow, const int64_t *a_ptr, const int64_t *a_idx, const double *a_sgn, int64_t nmid,
                                         const int64_t *b_ptr, const int64_t *b_idx, const double *b_sgn, int64_t ncol, int64_t *out_ptr,
                                         int64_t *out_idx, double *out_sgn, int64_t *out_nnz);
/* pos[q] = position of q_id[q] within row q_row[q] of (ptr, idx) (ids in 0 .. nid), -1 when absent: the dof -> local index maps
 * of DofAgglomeration (src/amge/DOFAgglomeration.cpp:533-655) for all agglomerated entities at once. */
ELAG_HIP_API int elag_hip_host_lookup_find(int64_t nrow, const int64_t *ptr, const int64_t *idx, int64_t nid, int64_t nq, const int64_t *q_row,
                                       const int64_t *q_id, int64_t *pos);
/* mfem::SparseMatrix::GetSubMatrix for every entity of a codimension at once: block e = M[rows of e, columns of e], column-major
 * nr_e x ncol[e] at out + out_ptr[e] (out zeroed by the caller; columns of M outside the entity's list are skipped). */
ELAG_HIP_API int elag_hip_host_gather_dense(int64_t n_ent, const int64_t *row_ptr, const int64_t *rows, const int64_t *col_ptr, const int64_t *col_idx,
                                        const int64_t *ncol, int64_t m_rows, int64_t m_cols, const int32_t *m_indptr, const int32_t *m_indices,
                                        const double *m_data, const int64_t *out_ptr, double *out);
/* Triplets (exact zeros dropped) of the sparse matrix whose entity blocks are the given dense local matrices: SetSubMatrix of every
 * entity's block of P_j, the functionals or the coarse derivative (src/amge/DeRhamSequence.cpp:1971-2010, 2540-2570).  Slot e holds
 * (slot_rows ? slot_rows[e] : nrows_e) * ld[e] doubles, column-major or row-major; out_* need sum_e nrows_e * ncols_e entries. */
ELAG_HIP_API int elag_hip_host_blocks_to_coo(int64_t n_ent, const int64_t *rows_ptr, const int64_t *rows_idx, const int64_t *cols_ptr,
                                             const int64_t *cols_idx, const double *vals, const int64_t *ld, const int64_t *slot_rows, int col_major,
                                             int64_t *out_r, int64_t *out_c, double *out_v, int64_t *out_nnz);
/* Per entity the rows dofs[ptr[e] .. ptr[e+1]) of the row-major ndof x nT array T, packed column-major (local target matrices). */
ELAG_HIP_API int elag_hip_host_pack_cols(int64_t n_ent, const int64_t *ptr, const int64_t *dofs, int64_t ndof, int64_t nT, const double *T, double *out);
/* From consecutive row-major ld[e] x ld[e] slots the leading k[e] x k[e] blocks, packed (coarse entity matrices). */
ELAG_HIP_API int elag_hip_host_take_leading(int64_t n_ent, const int64_t *ld, const int64_t *k, const double *vals, double *out);
/* findMinimalIntersectionSets (src/structures/minimalIntersectionSet.cpp:43-132) of Z = AE_fc^T AE_fc (+ the boundary-attribute term for
 * facets, src/topology/Topology.cpp:757-765) from the adjacency signatures of the sub-entities, without forming Z: mis[e] = set of
 * sub-entity e (-1: on no agglomerate boundary), sets numbered by their lowest member, ori[e] = orientation relative to it. */
ELAG_HIP_API int elag_hip_host_minimal_intersection_sets(int64_t nrow, const int64_t *fp, const int64_t *fi, const double *fs, int64_t nsub,
                                                         const int32_t *attr, int64_t *mis, double *ori, int64_t *n_sets);
/* Every row of (ptr, idx) sorted in place by (key[id], id): DofAgglomeration's reorderRow (src/amge/DOFAgglomeration.cpp:33-131). */
ELAG_HIP_API int elag_hip_host_sort_rows_by_key(int64_t nrow, const int64_t *ptr, int64_t *idx, int64_t nid, const int64_t *key);
/* entity_dof[c] of one form (DofHandler's entity -> dof table, src/amge/DofHandler.cpp:105-137 / DOFAgglomeration.cpp: rows = the entities of codimension c):
 * for every layer q = the codimensions 3 ... c + 1 that carry dofs of the form, the interior dofs of the sub-entities conn_q(r, :) in connectivity order
 * (base[q] + dof_off[q][e] ... dof_off[q][e + 1]), then the entity's own interior dofs (own_base + own_off[r] ...).  Call once with out_idx == NULL for
 * the row pointers (out_ptr[nrow] = the size of out_idx), then again to fill. */
ELAG_HIP_API int elag_hip_host_entity_dof_table(int64_t nrow, int nlayers, const int64_t *const *conn_ptr, const int64_t *const *conn_idx,
                                                const int64_t *const *dof_off, const int64_t *base, const int64_t *own_off, int64_t own_base,
                                                int64_t *out_ptr, int64_t *out_idx);
/* idx = concatenation of the ranges [starts[i], starts[i] + counts[i]), owner[k] = i of entry k. */
ELAG_HIP_API int elag_hip_host_expand_ranges(int64_t n, const int64_t *starts, const int64_t *counts, int64_t total, int64_t *idx, int64_t *owner);

/* Index tables between the facet stage and the extension stage of the two-form coarsening, one agglomerate at a time on host threads
 * (csrc/host_amge_tables.hip).  Replaces the per-agglomerate loops of DofAgglomeration (src/amge/DOFAgglomeration.cpp:86-131, reorderRow),
 * of hFacetExtension's boundary bookkeeping (src/amge/DeRhamSequence.cpp:2214-2290) and of DistributeAgglomerateMatrix.
 *  - dof_agglomeration: agglomerate a = elements els[ae_ptr[a]..), element e = dofs e_idx[e_ptr[e]..); sep[d] != 0 for dofs on agglomerated
 *    facets.  Writes ae_dof_ptr (nAE+1), n_int (nAE), ae_dofs (distinct dofs, interior first, each group ascending; room for all repeated
 *    dofs) and loc (position of every repeated dof in its agglomerate's list).
 *  - boundary_traces: coarse trace columns on every agglomerate's boundary (cb_ptr/cb) and the col-major (nb x next) blocks of trace values on
 *    its boundary dofs (Rb_ptr/R_b) taken from the facet kernel's packed output p_loc; cb == NULL: sizes only.
 *  - local_derivative: triplets of D (32-bit CSR) localised to (range dofs, agglomerate dofs) of every agglomerate; d_row == NULL: d_ptr only. */
/* Allocator policy for the host table stage (glibc mallopt): on != 0 keeps the pages of freed large arrays with the process (no mmap for
 * large blocks, trim threshold 2 GiB) so that the stages do not fault their work arrays in again; on == 0 restores glibc's defaults.
 * Process-wide; the Python host layer turns it on at the first coarsening unless ELAG_HIP_HOST_KEEP_PAGES=0. */
ELAG_HIP_API int elag_hip_host_keep_pages(int on);
/* Host threads the elag_hip_host_* table functions may use (n <= 0: the default = hardware threads, at most 16, or ELAG_HIP_HOST_THREADS);
 * returns the count in force.  The tables do not depend on it. */
ELAG_HIP_API int elag_hip_host_set_threads(int n);
/* Work (about a nanosecond per unit) a thread must get before a table loop is split over threads; <= 0: the default (131072). Returns the value in force. */
ELAG_HIP_API int64_t elag_hip_host_set_min_work(int64_t min_work);
ELAG_HIP_API int elag_hip_host_amge_dof_agglomeration(int64_t nAE, const int64_t *ae_ptr, const int64_t *els, const int64_t *e_ptr, const int64_t *e_idx,
                                                      int64_t ndof, const int64_t *sep, int64_t *ae_dof_ptr, int64_t *n_int, int64_t *ae_dofs, int64_t *loc);
ELAG_HIP_API int elag_hip_host_amge_boundary_traces(int64_t nAE, const int64_t *B_ptr, const int64_t *B_idx, const int64_t *cf_ptr, const int64_t *af_of_dof,
                                                    const int64_t *pos_in_af, const int64_t *af_d_ptr, int64_t k, const double *p_loc,
                                                    const int64_t *ae_dof_ptr, const int64_t *ae_dofs, const int64_t *n_int, int64_t *cb_ptr, int64_t *Rb_ptr,
                                                    int64_t *cb, double *R_b);
ELAG_HIP_API int elag_hip_host_amge_local_derivative(int64_t nAE, const int64_t *row_ptr, const int64_t *row_dofs, const int32_t *D_indptr,
                                                     const int32_t *D_indices, const double *D_data, const int64_t *ae_dof_ptr, const int64_t *ae_dofs,
                                                     const int64_t *n_int, int64_t *d_ptr, int32_t *d_row, int32_t *d_col, double *d_val);
/* local_derivative for any pair of forms: the entity's columns come as an unordered list (local column = first position). 64-bit triplets. */
ELAG_HIP_API int elag_hip_host_local_triplets(int64_t n_ent, const int64_t *rows_ptr, const int64_t *rows, const int32_t *D_indptr, const int32_t *D_indices,
                                              const double *D_data, const int64_t *col_ptr, const int64_t *col_idx, int64_t *d_ptr, int64_t *d_row,
                                              int64_t *d_col, double *d_val);
/* Element-to-dof table of form 2 (facet dofs facet after facet, then the element's interior dofs; DofHandler::GetrDof order,
 * src/amge/DofHandler.cpp:548-640).  ptr always written; idx == NULL: sizes only. */
ELAG_HIP_API int elag_hip_host_amge_elem_dof_table(int64_t nE, const int64_t *ef_ptr, const int64_t *ef_idx, const int64_t *f_dof_ptr,
                                                   const int64_t *e_idof_ptr, int64_t nfd, int64_t *ptr, int64_t *idx);
/* A family of dense blocks of a sparse matrix (mfem SetSubMatrix over all entities of one kind): block e couples rows
 * rows_idx[rows_ptr[e]..) with columns cols_idx[cols_ptr[e]..); its values are the next (slot_rows ? slot_rows[e] : nrows_e) * ld[e]
 * entries of vals (n_vals = length of vals; col_major: entry (r,c) at c*nrows_e + r, else at r*ld[e] + c). */
typedef struct elag_hip_block_family {
    int64_t n_ent;
    const int64_t *rows_ptr, *rows_idx, *cols_ptr, *cols_idx;
    const double *vals;
    int64_t n_vals;
    const int64_t *ld, *slot_rows;        /* slot_rows may be NULL */
    int col_major;
} elag_hip_block_family;
/* All families into one CSR matrix with sorted 32-bit columns (exact zeros dropped; entries that meet are added family by family, block by
 * block, whatever the thread count).  indices == NULL: indptr (row counts before merging) only. */
ELAG_HIP_API int elag_hip_host_block_families_to_csr(int n_fam, const elag_hip_block_family *fam, int64_t nrow, int64_t ncol, int64_t *indptr,
                                                     int32_t *indices, double *data, int64_t *out_nnz);
/* One family given by its arrays (elag_hip_host_blocks_to_coo's arguments) straight to CSR (exact zeros dropped, coinciding entries added in block order).
 * indices == NULL: indptr (row counts before merging) only. */
ELAG_HIP_API int elag_hip_host_blocks_to_csr(int64_t n_ent, const int64_t *rows_ptr, const int64_t *rows_idx, const int64_t *cols_ptr,
                                             const int64_t *cols_idx, const double *vals, const int64_t *ld, const int64_t *slot_rows, int col_major,
                                             int64_t nrow, int64_t ncol, int64_t *indptr, int32_t *indices, double *data, int64_t *out_nnz);

/* --------------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py): run `reps` launches of one fine-level kernel on the compute stream
 * bracketed by HIP events created on that stream and return the mean milliseconds per launch.
 * kernel: 0 = SpMV y=Ax, 1 = residual, 2 = l1-Jacobi sweep.
 * ------------------------------------------------------------------------------------------------ */
ELAG_HIP_API int elag_hip_time_kernel(elag_hip_parcsr_t A, elag_hip_smoother_t S, int kernel, int reps,
                                      const double *x_dev, double *y_dev, double *ms_per_launch);
/* Time `reps` applications of the hierarchy (preconditioner mode, device vectors) with HIP events. */
ELAG_HIP_API int elag_hip_time_hierarchy(elag_hip_hierarchy_t H, const double *rhs_dev, double *sol_dev,
                                         int reps, double *ms_per_apply);
/* The same for the cycle alone on vectors that already are in the hierarchy's internal numbering (no permutation around it):
 * what one preconditioner application costs inside elag_hip_krylov_mult when the solver has adopted that numbering. */
ELAG_HIP_API int elag_hip_time_hierarchy_internal(elag_hip_hierarchy_t H, int reps, double *ms_per_apply);

#ifdef __cplusplus
}
#endif
#endif /* ELAG_HIP_H */
