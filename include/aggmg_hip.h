/*
 * aggmg_hip.h -- C ABI of libaggmg_hip.so: the MI355X (gfx950) V-cycle hot path of
 * AgglomerationMultigrid1D (reference: Julia, /root/reference at build time).
 *
 * The reference has no FFI; its seams are Julia multiple dispatch.  Every
 * entry point below names the reference interface it replaces (file:line in the reference tree).
 * A Julia `ccall` shim binding exactly these symbols is in julia/AggMGHip.jl and described in
 * INTEGRATION.md; the Python mirror (agglomerationmultigrid1d_amd/) binds them through ctypes.
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature; handles are opaque pointers;
 *   - every function returns an int status (0 = ok, < 0 = error class below) and records a
 *     message retrievable with aggmg_last_error();
 *   - "host" entry points take caller-owned host pointers (e.g. Julia GC-owned arrays), are
 *     synchronous on return and never modify their inputs unless documented (`*_inout`);
 *   - "_dev" entry points take device pointers valid on the context's device and enqueue on the
 *     context's stream without synchronising;
 *   - all vectors are fp64, matrices arrive as Julia SparseMatrixCSC{Float64,Int64}
 *     (colptr[n+1], rowval[nnz], nzval[nnz], 1-based when one_based != 0);
 *   - one context is not thread-safe (the reference is single-threaded).
 */
#ifndef AGGMG_HIP_H
#define AGGMG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error classes; the Julia shim maps them back to the reference's exception types. */
#define AGGMG_OK 0
#define AGGMG_ERR_ARGUMENT (-1)    /* ArgumentError      src/mesh_heirarchy.jl:33-39,142-148 */
#define AGGMG_ERR_DIMENSION (-2)   /* DimensionMismatch  src/block_diagonal.jl:138,167-169,300-302 */
#define AGGMG_ERR_SINGULAR (-3)    /* SingularException  la.lu in src/smoother.jl:160 */
#define AGGMG_ERR_HIP (-4)         /* HIP runtime failure (no reference counterpart) */
#define AGGMG_ERR_UNSUPPORTED (-5) /* ErrorException     error(...) paths */

typedef struct aggmg_ctx aggmg_ctx;
typedef struct aggmg_op aggmg_op;             /* device mirror of one SparseMatrixCSC */
typedef struct aggmg_smoother aggmg_smoother; /* device mirror of one AbstractSmoother */
typedef struct aggmg_hier aggmg_hier;         /* device mirror of one MeshHierarchy */

/* ---- context ------------------------------------------------------------------------------ */
int aggmg_create(int device_id, aggmg_ctx** out);
int aggmg_destroy(aggmg_ctx* ctx);
const char* aggmg_last_error(aggmg_ctx* ctx); /* ctx may be NULL: last error of failed create */
/* Run on a caller-provided hipStream_t (e.g. torch's current stream; NULL = the device's default
 * stream).  aggmg_reset_stream goes back to the context's own non-blocking stream. */
int aggmg_set_stream(aggmg_ctx* ctx, void* hip_stream);
int aggmg_reset_stream(aggmg_ctx* ctx);
int aggmg_synchronize(aggmg_ctx* ctx);
/* Context options (take effect for smoothers set up afterwards).
 * AGGMG_OPT_SYMMETRIC_PACKING (default 1): when a block-tridiagonal operator is symmetric to round-off
 * (|B^{-1} - B^{-T}| <= 1e-13 |B^{-1}| block by block and Sub_e == Sup_{e-1}' to the same tolerance --
 * every operator the reference builds is) the SWEEPS of the fused kernel read the packed upper
 * triangle of (B^{-1} + B^{-T}) / 2 and rebuild B^{-1} Sub_e from the neighbour's super-diagonal
 * block instead of reading B^{-1} and B^{-1} Sub_e: 35 % fewer operator bytes per sweep launch.  This
 * changes the smoother's block inverses by <= 1e-13 relative -- the smoothed iterate moves at round-off
 * level, as between any two LU implementations; residuals and restrictions always use the operator's
 * own entries.  0 switches it off (the kernels then read B^{-1} and B^{-1} Sub as factored). */
#define AGGMG_OPT_SYMMETRIC_PACKING 1
/* AGGMG_OPT_COARSE_CHUNK_LOG2 (default 12): the largest chunk, in blocks (log2), one workgroup of the device
 * coarsest solve eliminates (hierarchies created afterwards).  An element-partitioned run gives every rank only its
 * share of the chunks of the replicated coarsest system: smaller chunks keep its CUs busy (measured on one rank's
 * share of an 8-rank 2^24 job: 0.448 ms per cycle with 10, 0.474 ms with 12).  Values 1 .. 12. */
#define AGGMG_OPT_COARSE_CHUNK_LOG2 2
/* AGGMG_OPT_DETECT_CHAIN (default 1): aggmg_jacobi_setup -- the form without element lists -- looks at the operator
 * itself: when, for some p in 1 .. 8, its size is n p + 1 and every entry lies in the pattern of a CG operator of
 * degree p on n elements in the reference's vertices-first numbering (src/cg_mesh.jl:37-45,59-65: element e couples
 * vertices e, e + 1 and its own p - 1 interior nodes n + 1 + e (p - 1) ...), the level takes the fused chain kernel
 * exactly as if aggmg_jacobi_setup_elements had been given the mesh's lists (same smoother, same results: the check
 * is on the entries, set-up time only).  0: operators without lists always take the generic CSR kernels. */
#define AGGMG_OPT_DETECT_CHAIN 3
/* AGGMG_OPT_PAIR_LEVELS (default 1; environment AGGMG_PAIR=0 makes the default 0): aggmg_vcycle_dev runs two small
 * agglomerated levels per launch where it can (aggmg_hier_level_paired) -- same results bit for bit; 0 keeps one
 * launch per level (A/B timing, tests). */
#define AGGMG_OPT_PAIR_LEVELS 4
/* AGGMG_OPT_MG_CHECKPOINT (default 1; environment AGGMG_MG_CHECKPOINT=0 makes the default 0): aggmg_multigrid_dev forms
 * the residual norm (and the error norm) of a checked cycle inside the fine-level launch that post-smooths it -- the launch
 * that goes on to pre-smooth the next cycle -- instead of in a residual launch of its own; the iterates are bit for bit
 * the same, the norms equal to round-off (another summation order).  aggmg_smoother_solve_dev likewise: launches of
 * several sweeps with the norms of every checked sweep formed inside (a launch in whose middle the tolerance is met is run
 * again up to that sweep).  Fused block-tridiagonal and point-Jacobi CG-chain fine levels / smoothers; others take the
 * separate launches. */
#define AGGMG_OPT_MG_CHECKPOINT 5
int aggmg_set_option(aggmg_ctx* ctx, int option, int value);
/* Raw device memory owned by the context's device (plumbing for harnesses without torch, and the storage of the
 * Julia shim's DeviceVector).  aggmg_dev_alloc returns ZEROED memory: a fresh vector is the zero initial guess of
 * ldiv! (src/solvers.jl:66,87). */
int aggmg_dev_alloc(aggmg_ctx* ctx, int64_t nbytes, void** out);
int aggmg_dev_free(aggmg_ctx* ctx, void* ptr);
int aggmg_memcpy_h2d(aggmg_ctx* ctx, void* dst_dev, const void* src_host, int64_t nbytes);
int aggmg_memcpy_d2h(aggmg_ctx* ctx, void* dst_host, const void* src_dev, int64_t nbytes);
/* Page-locked host memory for the host-pointer entry points (aggmg_vcycle: what multigrid_v_cycle(H, x0, b) /
 * ldiv!(y, H, b) on host vectors call, src/solvers.jl:19-50,63-92).  Pageable arrays are staged through pinned chunks by
 * worker threads (24 - 28 GB/s); arrays the caller keeps for many calls -- the vectors of a Krylov loop around ldiv! --
 * can be page-locked ONCE instead: copies from and to a registered range are single asynchronous DMA transfers on the
 * context's stream.  aggmg_host_register page-locks an existing range (hipHostRegister; the range must stay valid
 * until aggmg_host_unregister or aggmg_destroy), aggmg_host_alloc hands out pinned memory (hipHostMalloc) that
 * aggmg_host_free returns.  Registering per call does not pay: the registration costs what it saves (measured). */
int aggmg_host_register(aggmg_ctx* ctx, void* ptr, int64_t nbytes);
int aggmg_host_unregister(aggmg_ctx* ctx, void* ptr);
int aggmg_host_alloc(aggmg_ctx* ctx, int64_t nbytes, void** out);
int aggmg_host_free(aggmg_ctx* ctx, void* ptr);

/* ---- operators: H.mStiffness[k], H.mInterpolation[k] (src/mesh_heirarchy.jl:20,26) ---------- */
#define AGGMG_OP_STIFFNESS 0 /* used as A*u only                    src/solvers.jl:33,36,44 */
#define AGGMG_OP_TRANSFER 1  /* used as L*v and L'*v                src/solvers.jl:36,42   */
/* Upload a SparseMatrixCSC (m x n).  The device keeps a row-gather form (CSR, int32 indices)
 * and, for transfers, the transposed orientation too (the CSC arrays as given are the CSR of
 * L').  Index maps are preserved exactly: entry order inside a row/column is ascending, explicit
 * zeros stay stored.  Rejects dimensions or nnz >= 2^31 (AGGMG_ERR_ARGUMENT). */
int aggmg_csc_upload(aggmg_ctx* ctx, int64_t m, int64_t n, const int64_t* colptr,
                     const int64_t* rowval, const double* nzval, int one_based, int kind,
                     aggmg_op** out);
int aggmg_op_free(aggmg_ctx* ctx, aggmg_op* op);
int aggmg_op_shape(aggmg_ctx* ctx, const aggmg_op* op, int64_t* m, int64_t* n, int64_t* nnz);
/* Read the device index maps back (bit-exactness tests of the transfer index maps).
 * transposed == 0: CSR of the matrix (rowptr[m+1], colind[nnz], vals[nnz]);
 * transposed != 0: CSR of its transpose (rowptr[n+1], ...), transfers only.  0-based int32. */
int aggmg_op_download(aggmg_ctx* ctx, const aggmg_op* op, int transposed, int32_t* rowptr,
                      int32_t* colind, double* vals);
/* No-op, kept for ABI compatibility: the library keeps no host copy of an operator (set-up runs on the device). */
int aggmg_op_release_host(aggmg_ctx* ctx, aggmg_op* op);

/* ---- smoothers: src/smoother.jl ------------------------------------------------------------- */
/* dg_smoother(mesh, A, :blockJac) src/smoother.jl:153-165 and cg_smoother(..., :addSchwarz /
 * :hybridSchwarz) :104-134.  blockinds is the reference's mBlockInds Matrix{Int64}(m x nb),
 * column-major, 1-based when one_based != 0.  Diagonal blocks A[inds,inds] are extracted and
 * factorised with partial pivoting (exact zero pivot -> AGGMG_ERR_SINGULAR naming the block).
 * kind: 0 = BlockJacobi (also AdditiveSchwarz: same apply loop, blocks may overlap),
 *       1 = HybridSchwarz (result divided by the per-node block count, src/smoother.jl:24-46),
 *       2 = red-black block Gauss-Seidel -- EXTENSION, the reference has no Gauss-Seidel smoother
 *           (SURVEY.md D1): one sweep = for colour in (even elements, odd elements):
 *           u_e += alpha B_e^{-1} (b - A u)_e on that colour with the newest u.  V-cycles pre-smooth
 *           in this order and post-smooth in the reverse one.  Needs contiguous blocks and a block-
 *           tridiagonal operator, or the element lists of a CG mesh (below): elements of one colour share
 *           no node (AGGMG_ERR_UNSUPPORTED otherwise); aggmg_smoother_apply on such a smoother applies
 *           its block-diagonal part like kind 0.
 * When blockinds are the element node lists of a CG mesh in mesh order ((p+1) x n, consecutive elements
 * sharing exactly one node, local order [left vertex, right vertex, interior nodes], 2 <= p+1 <= 9) the
 * SWEEPS with the smoother (aggmg_smooth*, V-cycles) run in the fused chain kernel -- the residual of a
 * tile into LDS, then every row its row of A_e \ r_e from the element inverse kept in registers;
 * aggmg_smoother_apply keeps the generic batched-block kernel.  Same results to round-off. */
int aggmg_blockjacobi_setup(aggmg_ctx* ctx, aggmg_op* A, int64_t m, int64_t nb,
                            const int64_t* blockinds, int one_based, int kind,
                            aggmg_smoother** out);
/* dg_smoother / cg_smoother (..., :jac): JacobiSmoother(Diagonal(A)) src/smoother.jl:95-102,146-152 */
int aggmg_jacobi_setup(aggmg_ctx* ctx, aggmg_op* A, aggmg_smoother** out);
/* cg_smoother(cgMesh, A, :jac) src/smoother.jl:88-102 with the element node lists the reference's
 * cg_smoother holds in cgMesh.mElements[k].mNodesInd (the (p+1) x n matrix its Schwarz variants store
 * as mBlockInds, src/smoother.jl:104-134): element_nodes is that matrix, column-major, 1-based when
 * one_based != 0, local node order [left vertex, right vertex, interior nodes] (src/cg_mesh.jl:35-45).
 * Same smoother as aggmg_jacobi_setup (Diagonal(A)); when consecutive elements share exactly one node
 * and every stored entry of A couples nodes of one element (true of every CG stiffness matrix and
 * of its Galerkin coarsenings, src/mesh_heirarchy.jl:53-59) the level additionally gets the
 * element-contiguous "chain" form and runs the LDS-tiled fused point-Jacobi kernel
 * (aggmg_smoother_is_structured reports 1).  Vectors stay in the reference's vertices-first
 * numbering (src/cg_mesh.jl:37-45,59-65) at every entry point.  Anything else falls back to the
 * generic CSR kernels with identical results. */
int aggmg_jacobi_setup_elements(aggmg_ctx* ctx, aggmg_op* A, int64_t nodes_per_element, int64_t n_elements,
                                const int64_t* element_nodes, int one_based, aggmg_smoother** out);
/* BlockDiagonal (factorize = 0: apply = `A * X`, mul! src/block_diagonal.jl:166-176) and
 * BlockDiagonalLU (factorize = 1: apply = `A \\ X`, ldiv! :299-309; `lu(A)` :276) as block objects of
 * the same kind as the block smoother: blocks = nb dense m x m blocks, each column-major, with the
 * contiguous index lists the BlockDiagonal(mBlocks) constructor (:27-41) makes.  Applied with
 * aggmg_smoother_apply(..., alpha = 1).  A singular block -> AGGMG_ERR_SINGULAR. */
int aggmg_blockdiag_setup(aggmg_ctx* ctx, int64_t m, int64_t nb, const double* blocks, int factorize,
                          aggmg_smoother** out);
int aggmg_smoother_free(aggmg_ctx* ctx, aggmg_smoother* sm);
/* apply_smoother(S, B; alpha) -> alpha * (S \ B), B is N x ncols column-major, result in Y.
 * src/smoother.jl:6-18,30-46,56-58,69-81.  Host pointers. */
int aggmg_smoother_apply(aggmg_ctx* ctx, aggmg_smoother* sm, const double* B, int64_t N,
                         int64_t ncols, double alpha, double* Y);
/* 1 when the (operator, smoother) pair was recognised as block-tridiagonal with contiguous
 * aligned element blocks and runs the LDS-tiled fused kernels; 0 = generic CSR path. */
int aggmg_smoother_is_structured(aggmg_ctx* ctx, const aggmg_smoother* sm, int* out);

/* ---- sparse set-up operations on the device ---------------------------
 * The products the reference's hierarchy constructors are made of, on operators that are already in
 * HBM; every result is an ordinary operator (aggmg_op) in CSC form.
 * aggmg_bd_sp_apply: `A * S` for A::BlockDiagonal (bd_sp_matmul / bd_sp_colmul, src/block_diagonal.jl:195-264)
 * or `A \ S` for A::BlockDiagonalLU (bd_sp_solve / bd_sp_colsolve, :314-383) with a sparse S: bd is the block
 * object of aggmg_blockdiag_setup (factorize = 0 / 1).  As in the reference, every column of the result
 * holds ALL m rows of each block its column of S touches (zeros included).  The solve applies the explicit
 * inverse of the pivoted LU (the reference runs getrs per block: equal up to round-off).
 * aggmg_sp_matmul: A * B; aggmg_sp_sub: A - B with numerically-zero results dropped (SparseArrays' `-`,
 * SURVEY.md 9.4); aggmg_op_transpose: the adjoint as an operator of its own.  With them
 *     G_c = L' * G * L,   A_c = C_c - D_c * (M_LU \ G_c)          src/mesh_heirarchy.jl:71-72,79-84,98-103
 * run on the device (agglomerationmultigrid1d_amd.api.MeshHierarchy.from_dg_operators).  One thread
 * per result column, accumulation in SparseArrays' order; columns longer than 128 rows are refused
 * (AGGMG_ERR_UNSUPPORTED).  kind: AGGMG_OP_STIFFNESS or AGGMG_OP_TRANSFER for the result. */
int aggmg_bd_sp_apply(aggmg_ctx* ctx, aggmg_smoother* bd, aggmg_op* S, int kind, aggmg_op** out);
int aggmg_sp_matmul(aggmg_ctx* ctx, aggmg_op* A, aggmg_op* B, int kind, aggmg_op** out);
int aggmg_sp_sub(aggmg_ctx* ctx, aggmg_op* A, aggmg_op* B, int kind, aggmg_op** out);
int aggmg_op_transpose(aggmg_ctx* ctx, aggmg_op* A, int kind, aggmg_op** out);
/* The CSC arrays of an operator (colptr[n+1], rowval[nnz], nzval[nnz], 0-based int32) and the dense blocks
 * of a block smoother / block object ([nb][m][m] row-major: the inverses, or the matrices of a
 * factorize = 0 object): read-back for the parity tests of the set-up products. */
int aggmg_op_download_csc(aggmg_ctx* ctx, const aggmg_op* op, int32_t* colptr, int32_t* rowval, double* nzval);
int aggmg_smoother_download_blocks(aggmg_ctx* ctx, const aggmg_smoother* sm, double* out);
/* Test aid for the size guard of the products above: runs their count -> column-pointer scan on n caller-supplied
 * int32 counts (host array).  The total is formed in 64 bits before anything is scanned or allocated; a result of
 * 2^31 or more entries is refused with AGGMG_ERR_UNSUPPORTED (int32 device indices), *total still set. */
int aggmg_debug_scan_counts(aggmg_ctx* ctx, const int32_t* counts_host, int64_t n, int64_t* total);
/* Measurement aid (tools/exp_coarse.py --calibrate; no reference counterpart): a plain 16-byte streaming kernel on
 * `workgroups` workgroups of 256 threads over device buffers of nbytes -- mode 0 copies src to dst, mode 1 only reads
 * src -- timed with HIP events on the context stream (synchronous); *ms_out = its duration.  The ceiling the
 * coarsest solve's streaming steps (src/solvers.jl:39) are held against in DESIGN.md section 5. */
int aggmg_debug_stream_copy(aggmg_ctx* ctx, void* dst, const void* src, int64_t nbytes, int workgroups, int mode,
                            double* ms_out);

/* ---- fused hot-path operations -------------------------------------------------------------- */
/* nsweeps x  `u += apply_smoother(S, b - A*u; alpha)`   src/solvers.jl:32-35,43-46, :199 */
int aggmg_smooth(aggmg_ctx* ctx, aggmg_op* A, aggmg_smoother* sm, double* u_inout,
                 const double* b, double alpha, int nsweeps);
/* r = b - A*u                                           src/solvers.jl:33,36,44 */
int aggmg_residual(aggmg_ctx* ctx, aggmg_op* A, const double* u, const double* b, double* r_out);
/* rc = L' * r                                           src/solvers.jl:36 */
int aggmg_restrict(aggmg_ctx* ctx, aggmg_op* L, const double* r, double* rc_out);
/* u += L * uc                                           src/solvers.jl:42 */
int aggmg_prolong_add(aggmg_ctx* ctx, aggmg_op* L, const double* uc, double* u_inout);
/* Device-pointer variants (asynchronous on the context stream).  u_out may equal u_in. */
int aggmg_smooth_dev(aggmg_ctx* ctx, aggmg_op* A, aggmg_smoother* sm, const double* u_in,
                     const double* b, double alpha, int nsweeps, double* u_out);
int aggmg_residual_dev(aggmg_ctx* ctx, aggmg_op* A, const double* u, const double* b,
                       double* r_out);
int aggmg_restrict_dev(aggmg_ctx* ctx, aggmg_op* L, const double* r, double* rc_out);
int aggmg_prolong_add_dev(aggmg_ctx* ctx, aggmg_op* L, const double* uc, double* u_inout);

/* ---- hierarchy: MeshHierarchy + multigrid_v_cycle ------------------------------------------- */
/* Coarsest level `u[n] = A_n \\ rhs[n]` (src/solvers.jl:39; UMFPACK re-factorises per cycle in the
 * reference, here the factorisation is done once at aggmg_hier_create). */
/* Environment variables read when the device solver is planned / launched -- measurement and test knobs, none of
 * them changes a result beyond round-off: AGGMG_CR_TAIL_ROWS, AGGMG_CR_MAX_Q (smaller tail / chunks: the tests run
 * the several-stage plan of > 2^24-row systems at small sizes with them), AGGMG_CR_FILL, AGGMG_CR_MINWG,
 * AGGMG_CR_THREADS (chunk and workgroup size sweeps, tools/exp_coarse_knobs.sh), AGGMG_CR_FUSE_TAIL=1 (boundary
 * system solved by the last-arriving workgroup of the forward launch: measured 5x slower). */
#define AGGMG_COARSE_HOST_BANDED 0 /* banded LU with partial pivoting on the host (D2H, solve, H2D) */
#define AGGMG_COARSE_DEVICE_CR 1   /* block cyclic reduction on the device; error if not applicable */
#define AGGMG_COARSE_EXTERNAL 3    /* no factorisation: the caller solves the coarsest system between
                                      aggmg_vcycle_down_dev and aggmg_vcycle_up_dev (multi-GPU driver) */
#define AGGMG_COARSE_AUTO 2        /* device cyclic reduction when the operator is block-tridiagonal
                                      with well-conditioned pivot blocks, host banded LU otherwise */
/* Mirrors the operator vectors of `struct MeshHierarchy` src/mesh_heirarchy.jl:17-28:
 * stiffness[nlevels], smoothers[nlevels-1] (the coarsest level is solved directly,
 * src/solvers.jl:39), interpolation[nlevels-1] with interpolation[k]: level k+1 -> level k.
 * Level 0 is the finest. */
int aggmg_hier_create(aggmg_ctx* ctx, int nlevels, aggmg_op* const* stiffness,
                      aggmg_smoother* const* smoothers, aggmg_op* const* interpolation,
                      int coarse_mode, aggmg_hier** out);
int aggmg_hier_free(aggmg_ctx* ctx, aggmg_hier* h);
/* multigrid_v_cycle(H, x0, b; nPre, nPost, alpha) -> x    src/solvers.jl:19-50.
 * x0 and b are not modified (src/solvers.jl:25-26); x_out may alias neither.
 * x0 == NULL: a zero initial guess -- ldiv!(H, b) / ldiv!(y, H, b), src/solvers.jl:63-92 -- without a vector of zeros
 * being sent (host form: one transfer less) or read (the finest level then starts from zeros like the others, :29-31);
 * the same bits as passing zeros. */
int aggmg_vcycle(aggmg_ctx* ctx, aggmg_hier* h, const double* x0, const double* b, int nPre,
                 int nPost, double alpha, double* x_out);
int aggmg_vcycle_dev(aggmg_ctx* ctx, aggmg_hier* h, const double* x0, const double* b, int nPre,
                     int nPost, double alpha, double* x_out);
/* ncycles V-cycles back to back, x <- multigrid_v_cycle(H, x, b): the hot loop of multigrid()
 * (src/solvers.jl:124-126), same arithmetic as ncycles aggmg_vcycle_dev calls.  On block-
 * tridiagonal fine levels the post-smoothing of one cycle and the pre-smoothing of the next run in
 * one fused launch, so the fine operator is read once per cycle instead of twice. */
int aggmg_vcycles_dev(aggmg_ctx* ctx, aggmg_hier* h, const double* x0, const double* b, int ncycles,
                      int nPre, int nPost, double alpha, double* x_out);
/* How the fused descent forms the restricted residual L'(rhs - A u) of src/solvers.jl:36.
 * AGGMG_RESTRICT_EXPLICIT (the default of every hierarchy): r = rhs - A u with the operator's own
 * entries, then L' r -- the reference's arithmetic.  AGGMG_RESTRICT_PRECONDITIONED: (L'D) w from the
 * preconditioned residual w = B^{-1} r that the sweeps already hold; equal in exact arithmetic and
 * ~20 % cheaper per V-cycle (the descent reads neither the diagonal blocks nor L), but w carries the
 * rounding of the stored block inverses.  Measured on the model problem: the factor by which one
 * V(3,3) cycle multiplies the smoothest mode grows like n^2 in every implementation -- reference-
 * order arithmetic 0.002 / 0.031 / 0.498 at 2^20 / 2^22 / 2^24 fine elements, the explicit form the
 * same to three digits, the preconditioned form 0.009 / 0.134 / 2.13: at 2^24 it AMPLIFIES the mode
 * and the multigrid iteration diverges (tests/manual/exp_smooth_mode.py).
 * aggmg_hier_set_restriction therefore returns AGGMG_ERR_UNSUPPORTED for the preconditioned form on
 * hierarchies with more than AGGMG_RESTRICT_PRECONDITIONED_MAX_ELEMS fine elements (where the
 * extrapolated factor passes ~0.05).  No environment variable selects the mode. */
#define AGGMG_RESTRICT_EXPLICIT 0
#define AGGMG_RESTRICT_PRECONDITIONED 1
#define AGGMG_RESTRICT_PRECONDITIONED_MAX_ELEMS (1 << 21)
int aggmg_hier_set_restriction(aggmg_ctx* ctx, aggmg_hier* h, int mode);
int aggmg_hier_get_restriction(aggmg_ctx* ctx, const aggmg_hier* h, int* mode);
/* The two halves of the V-cycle around the coarsest solve (src/solvers.jl:28-37 and :41-47), for
 * callers that solve the coarsest system themselves (element-partitioned multi-GPU runs gather it
 * across ranks).  After _down the coarsest right-hand side is in the buffer reported by
 * aggmg_hier_coarse_buffers(); _up expects the coarsest solution in the solution buffer. */
int aggmg_vcycle_down_dev(aggmg_ctx* ctx, aggmg_hier* h, const double* x0, const double* b, int nPre,
                          double alpha);
int aggmg_vcycle_up_dev(aggmg_ctx* ctx, aggmg_hier* h, const double* b, int nPost, double alpha,
                        double* x_out);
/* The ascent in three calls so that a partitioned run can exchange the interface elements of
 * x_out while the rest of the fine level is still being smoothed: part 0 = the coarser levels,
 * part 1 = the fine-level tiles holding elements [0, head_elems) and [tail_elem, ne), part 2 = the
 * remaining fine-level tiles.  Parts 1 and 2 do not depend on each other (they may be issued on
 * different streams, aggmg_set_stream); together they are bitwise aggmg_vcycle_up_dev.
 * AGGMG_ERR_UNSUPPORTED unless the fine level runs the fused block-tridiagonal kernel.  part 3: the finest level alone,
 * every tile, whatever kernel runs it (after the coarser levels went through part 0 or aggmg_vcycle_up_coarse_dev). */
int aggmg_vcycle_up_split_dev(aggmg_ctx* ctx, aggmg_hier* h, const double* b, int nPost, double alpha,
                              double* x_out, int64_t head_elems, int64_t tail_elem, int part);
/* The ascent below the finest level in two parts around the exchange of the coarsest solution's ghost blocks
 * (element-partitioned runs; src/solvers.jl:41-47 for the levels between the coarsest and the finest): part 2 runs the
 * tiles of the first launch that read none of the first ghosts_lo / last ghosts_hi elements of the coarsest level -- they
 * can go while the neighbours' blocks are still travelling --, part 1 the remaining tiles at the two ends and the rest of
 * those levels.  Both parts together = aggmg_vcycle_up_split_dev(part 0), bit for bit.  AGGMG_ERR_UNSUPPORTED unless
 * the levels below the finest are one two-level launch next to the coarsest level (aggmg_hier_level_paired). */
int aggmg_vcycle_up_coarse_dev(aggmg_ctx* ctx, aggmg_hier* h, const double* b, int nPost, double alpha, int part,
                               int64_t ghosts_lo, int64_t ghosts_hi);
int aggmg_hier_coarse_buffers(aggmg_ctx* ctx, aggmg_hier* h, void** rhs_dev, void** sol_dev,
                              int64_t* n);
/* The device coarsest solve phase by phase, for element-partitioned runs: block cyclic reduction
 * eliminates the first `chunk_log2` levels independently per chunk of 2^chunk_log2 block rows, so
 * every rank runs `chunk_forward` on the chunks of its own block range [blk_lo, blk_hi) (aligned to
 * the chunk size) from its owned right-hand side, the ranks exchange their slices of the two
 * boundary vectors partR / partL (n_boundary * block_size values each, written at global positions),
 * every rank solves the small boundary system, and `chunk_backward` produces the owned solution.
 * aggmg_coarse_plan reports chunk_log2 = -1 when the hierarchy's coarsest solver has no such plan. */
int aggmg_coarse_plan(aggmg_ctx* ctx, const aggmg_hier* h, int* chunk_log2, int64_t* n_boundary,
                      int* block_size, int64_t* n_blocks);
int aggmg_coarse_chunk_forward_dev(aggmg_ctx* ctx, aggmg_hier* h, const double* rhs_owned, int64_t blk_lo,
                                   int64_t blk_hi, double* partR, double* partL);
int aggmg_coarse_boundary_solve_dev(aggmg_ctx* ctx, aggmg_hier* h, const double* partR, const double* partL,
                                    double* xq);
int aggmg_coarse_chunk_backward_dev(aggmg_ctx* ctx, aggmg_hier* h, const double* rhs_owned, int64_t blk_lo,
                                    int64_t blk_hi, const double* xq, double* x_owned);
/* Which kernels smooth a level (level 0 = finest): the fused block-tridiagonal kernel (DG /
 * agglomerated levels), the fused chain kernel (CG levels set up with aggmg_jacobi_setup_elements),
 * the generic CSR kernels, or -- last level -- the coarsest direct solve. */
#define AGGMG_LEVEL_GENERIC 0
#define AGGMG_LEVEL_FUSED_BTD 1
#define AGGMG_LEVEL_FUSED_CHAIN 2
#define AGGMG_LEVEL_COARSEST 3
int aggmg_hier_level_kind(aggmg_ctx* ctx, const aggmg_hier* h, int level, int* kind);
/* Whether aggmg_vcycle_dev runs levels `level` and `level + 1` in ONE launch each way (nsweeps sweeps per level):
 * small agglomerated levels -- block size 2, dense off-diagonal blocks, one agglomeration ratio per level -- below
 * the finest one, both halves of src/solvers.jl:28-37 / :41-47 for two levels with the hand-over in LDS; bit for bit
 * the separate launches.  The launch is then attributed to `level` (profile tags, aggmg_hier_launch_bytes of both
 * levels apply).  AGGMG_OPT_PAIR_LEVELS switches the pairing off. */
int aggmg_hier_level_paired(aggmg_ctx* ctx, const aggmg_hier* h, int level, int nsweeps, int* paired);
/* Which coarsest solver a hierarchy uses: on_device (1 = cyclic reduction), its block size and the
 * largest pivot-block condition estimate met while factoring (0 for the host solver). */
int aggmg_hier_coarse_info(aggmg_ctx* ctx, const aggmg_hier* h, int* on_device, int* block_size,
                           double* cond_est);
/* The cyclic reduction pivots inside its m x m blocks only, the reference's `A_n \ rhs_n` (UMFPACK,
 * src/solvers.jl:39) across the whole matrix.  aggmg_hier_create therefore accepts a device factorisation on
 * evidence: it solves one probe system d = A w (w hash-random) and keeps the factorisation only when
 * ||d - A x|| / ||d|| < 1e-10; otherwise AGGMG_COARSE_AUTO falls back to the host banded LU with partial
 * pivoting (AGGMG_COARSE_DEVICE_CR: AGGMG_ERR_UNSUPPORTED).  backward_error receives the probe's figure
 * (-1 when no device factorisation was attempted). */
int aggmg_hier_coarse_probe(aggmg_ctx* ctx, const aggmg_hier* h, double* backward_error);
/* How the device solve ends: the boundary system its chunk stages leave (or the whole system when it is small) goes to
 * ONE workgroup -- *kind = 2: parallel cyclic reduction (block sizes 1 and 2, up to 1024 blocks -- above 512 with one
 * ordinary reduction level around it; kept only where
 * aggmg_hier_create measured it as accurate as kind 1 on a system with a large smooth solution), 1: the register-blocked
 * cyclic reduction of the stages, 0: no device factorisation; *blocks = block rows of that system. */
int aggmg_hier_coarse_tail(aggmg_ctx* ctx, const aggmg_hier* h, int* kind, int64_t* blocks);
/* Milliseconds the last aggmg_vcycle* call spent in the coarsest direct solve (host path:
 * D2H + solve + H2D, measured with the host clock). */
int aggmg_hier_last_coarse_ms(aggmg_ctx* ctx, const aggmg_hier* h, double* ms);

/* Up to four strided 2-D copies of device doubles in ONE launch:
 * dst[g][i * dst_ld[g] + j] = src[g][i * src_ld[g] + j], i < rows[g], j < cols[g].  Packs / unpacks
 * the interface DoFs around the all-gathers of an element-partitioned run (the reference has no
 * counterpart: it is single-process). */
int aggmg_copy_segments_dev(aggmg_ctx* ctx, int nseg, const double* const* src, double* const* dst,
                            const int64_t* rows, const int64_t* cols, const int64_t* src_ld,
                            const int64_t* dst_ld);

/* ---- element-partitioned V-cycle inside the library (BASELINE config 4) -----------
 * The reference is single-process; this is one rank's share of multigrid_v_cycle (src/solvers.jl:19-50)
 * on a LOCAL hierarchy (owned elements + W_k ghost elements per side and level, created with
 * AGGMG_COARSE_EXTERNAL) plus a one-level hierarchy holding the GLOBAL coarsest operator, with the
 * interface exchanges issued by the library on the context's stream: per cycle one all-gather of the
 * interface elements of x0, and the coarsest solve across ranks (chunk elimination on the rank's own
 * block range, all-gather of the chunk-boundary system, replicated boundary solve, back substitution,
 * all-gather of the coarse ghost blocks; small coarsest levels: all-gather of the owned right-hand
 * side + replicated solve).  Layout arrays have nlevels entries (level 0 finest): owned element range
 * [own_lo, own_hi), local range [loc_lo, loc_hi) (owned + ghosts, clipped to the domain), global
 * element counts ne, DoFs per element m, ghost widths W.  Vectors are LOCAL (loc range) device vectors.
 * Owned results are bitwise those of the single-GPU cycle (tests/test_distributed_gpu.py). */
typedef struct aggmg_dist aggmg_dist;
int aggmg_dist_create(aggmg_ctx* ctx, aggmg_hier* local, aggmg_hier* coarse_global, int world, int rank, int nlevels,
                      const int64_t* own_lo, const int64_t* own_hi, const int64_t* loc_lo, const int64_t* loc_hi,
                      const int64_t* ne, const int32_t* m, const int32_t* W, aggmg_dist** out);
int aggmg_dist_free(aggmg_ctx* ctx, aggmg_dist* d);
/* Collectives, one of:
 *  - RCCL inside the library: rank 0 calls aggmg_rccl_unique_id, the AGGMG_RCCL_ID_BYTES bytes travel to
 *    every rank by any means (they are not secret and small), every rank calls aggmg_dist_init_rccl
 *    (ncclCommInitRank; collective).  librccl.so.1 is loaded with dlopen on first use -- the copy already
 *    in the process if there is one.  nranks_out receives what the communicator reports.  Handing over TWO ids back
 *    to back (nbytes = 2 * AGGMG_RCCL_ID_BYTES, two calls of aggmg_rccl_unique_id) gives the second stream -- the
 *    interface exchange issued under the fine-level ascent -- a communicator of its own: operations of one
 *    communicator in flight on two streams may start in a different order on different ranks and wait for each other.
 *  - a caller-supplied all-gather: recv_dev[r * count + i] = rank r's send_dev[i], ordered on hip_stream
 *    after everything enqueued there so far; returns 0 on success.  ALIASING: the chunk-boundary system is gathered
 *    IN PLACE -- send_dev == recv_dev + rank * count (the caller's own slice of the receive buffer, already in its
 *    final position) -- and the callback has to treat that as an in-place all-gather (ncclAllGather does by
 *    definition; MPI needs MPI_IN_PLACE; a copy-based implementation must not clobber or re-copy that slice).  The
 *    other gathers pass disjoint buffers.
 *  - a device-local loop-back (every slot receives the caller's own data): rehearsals of one rank's
 *    share on a single GPU, measurement only. */
#define AGGMG_RCCL_ID_BYTES 128
typedef int (*aggmg_allgather_fn)(void* user, const double* send_dev, double* recv_dev, int64_t count, void* hip_stream);
int aggmg_rccl_unique_id(aggmg_ctx* ctx, void* id_out, int nbytes);
/* Whether RCCL can be loaded in this process (no context, no device needed): AGGMG_OK, or AGGMG_ERR_UNSUPPORTED
 * with the loader's message copied into why (NUL-terminated, at most nbytes; may be NULL) -- the same outcome
 * aggmg_rccl_unique_id / aggmg_dist_init_rccl report, so that all ranks can agree on a fallback before the first
 * collective.  The environment variable AGGMG_RCCL_LIB names another library file to load (tests). */
int aggmg_rccl_available(char* why, int nbytes);
int aggmg_dist_init_rccl(aggmg_ctx* ctx, aggmg_dist* d, const void* id, int nbytes, int* nranks_out);
int aggmg_dist_set_allgather(aggmg_ctx* ctx, aggmg_dist* d, aggmg_allgather_fn fn, void* user);
/* Neighbour messages.  An interface exchange moves a rank's first / last owned elements into the ghost elements of
 * its left / right neighbour; as grouped ncclSend / ncclRecv between the vectors themselves that is ONE launch
 * with no pack, all-gather or unpack around it (SURVEY.md 8e names it as the equivalent of the interface
 * all-gather).  RCCL and the loop-back do so by default; a caller-supplied collective backend does when it also
 * supplies aggmg_dist_set_sendrecv: fn posts nops operations -- is_send[i] ? send count[i] doubles at dev_ptr[i]
 * to rank peer[i] : receive them from it -- ordered on hip_stream, messages between one pair of ranks matched in
 * order, returns 0 on success (`user` is the pointer given to aggmg_dist_set_allgather).  The environment variable
 * AGGMG_DIST_P2P=0 keeps every exchange on pack -> all-gather -> unpack.
 * aggmg_dist_set_neighbor_layout: the slices [off, off + len) of a local level-`level` vector (finest or
 * coarsest) that go to / are filled from the left and right neighbour, at most two per direction; rank r's
 * to_right must match rank r + 1's from_left slice by slice, to_left rank r - 1's from_right.  Levels with
 * element-contiguous DoFs have it by default; a level given an explicit aggmg_dist_set_exchange_layout loses the
 * default and exchanges by all-gather until this is called. */
typedef int (*aggmg_sendrecv_fn)(void* user, int nops, const int* peer, const int* is_send, double* const* dev_ptr,
                                 const int64_t* count, void* hip_stream);
int aggmg_dist_set_sendrecv(aggmg_ctx* ctx, aggmg_dist* d, aggmg_sendrecv_fn fn);
int aggmg_dist_set_neighbor_layout(aggmg_ctx* ctx, aggmg_dist* d, int level, int nto_left, const int64_t* to_left_off,
                                   const int64_t* to_left_len, int nto_right, const int64_t* to_right_off,
                                   const int64_t* to_right_len, int nfrom_left, const int64_t* from_left_off,
                                   const int64_t* from_left_len, int nfrom_right, const int64_t* from_right_off,
                                   const int64_t* from_right_len);
int aggmg_dist_set_loopback(aggmg_ctx* ctx, aggmg_dist* d);
/* Interface layout of a level whose DoFs are not contiguous per element (a CG level in the reference's
 * vertices-first numbering, src/cg_mesh.jl:37-45,59-65: the interface is a slice of the vertex part plus
 * a slice of the interior part).  count doubles per rank travel; send_*: slices [src, src + len) of the local
 * vector packed at dst of this rank's part; left_* / right_*: slices of the left / right neighbour's part
 * unpacked at dst of the local vector.  level: 0 (finest) or nlevels - 1 (coarsest).  Levels with contiguous
 * elements need no call (first / last W elements, the default). */
int aggmg_dist_set_exchange_layout(aggmg_ctx* ctx, aggmg_dist* d, int level, int64_t count, int nsend,
                                   const int64_t* send_src, const int64_t* send_dst, const int64_t* send_len,
                                   int nleft, const int64_t* left_src, const int64_t* left_dst, const int64_t* left_len,
                                   int nright, const int64_t* right_src, const int64_t* right_dst,
                                   const int64_t* right_len);
/* One all-gather through the configured backend (count doubles per rank); also the smoke test of it. */
int aggmg_dist_allgather_dev(aggmg_ctx* ctx, aggmg_dist* d, const double* send_dev, double* recv_dev, int64_t count);
/* Fill the ghost entries of a local finest- or coarsest-level vector from the neighbours' owned
 * interface elements (needed once for the right-hand side; the cycle does it for x0 itself). */
int aggmg_dist_exchange_ghosts_dev(aggmg_ctx* ctx, aggmg_dist* d, double* x_local, int level);
/* One V-cycle.  x0's ghost entries are overwritten with the neighbours' values unless
 * AGGMG_DIST_X0_GHOSTS_VALID; b must be valid on the whole local range; on return the owned part of
 * x_out is the result (its ghosts are not).  AGGMG_DIST_OVERLAP_NEXT: the caller will pass x_out as the
 * next cycle's x0 (the loop of multigrid, src/solvers.jl:124-126) -- the fine-level ascent then produces
 * the interface elements first and their all-gather runs on a second stream under the rest of that
 * launch; the next call only waits for it.  Same arithmetic either way. */
#define AGGMG_DIST_X0_GHOSTS_VALID 1
#define AGGMG_DIST_OVERLAP_NEXT 2
/* AGGMG_DIST_GRAPH: replay the cycle as a hipGraph.  The first call with a given argument tuple is
 * issued launch by launch, the second is captured (both streams, the collective included) and every
 * later one is a single hipGraphLaunch.  Falls back to launch-by-launch issue for good when the capture
 * fails (a collective backend that cannot be captured), with caller-supplied collectives, the host
 * coarsest solver, or while the event profiler is on.  aggmg_dist_graph_info reports what happened. */
#define AGGMG_DIST_GRAPH 4
int aggmg_dist_vcycle_dev(aggmg_ctx* ctx, aggmg_dist* d, double* x0, const double* b, double* x_out, int nPre,
                          int nPost, double alpha, int flags);
/* exchanges: all-gathers issued so far; chunked: 1 when the coarsest solve follows the partition;
 * backend: 0 none, 1 caller-supplied, 2 RCCL, 3 loop-back */
/* Whether the exchange of the coarsest solution's ghost blocks runs on the side stream under the middle tiles of the
 * two-level ascent (aggmg_vcycle_up_coarse_dev) instead of in front of it.  Off by default (environment
 * AGGMG_DIST_COARSE_OVERLAP=1 makes it the default): it trades an extra launch and two stream joins for the latency of one
 * neighbour exchange, which pays only where that latency is real -- callers time both (bench.py does, in its warm-up) and
 * must choose the same on every rank.  Same results bit for bit either way. */
int aggmg_dist_set_coarse_overlap(aggmg_ctx* ctx, aggmg_dist* d, int on);
int aggmg_dist_info(aggmg_ctx* ctx, const aggmg_dist* d, int64_t* exchanges, int* chunked, int* backend);
int aggmg_dist_graph_info(aggmg_ctx* ctx, const aggmg_dist* d, int64_t* replays, int* captured, int* broken);

/* ---- outer solver loops, device-resident -------------------------------------- */
/* x . y and ||x||_2 of device vectors (fixed reduction tree: reproducible run to run). */
int aggmg_dot_dev(aggmg_ctx* ctx, const double* x, const double* y, int64_t n, double* out);
int aggmg_norm2_dev(aggmg_ctx* ctx, const double* x, int64_t n, double* out);
/* ||b - A x||_2: the `la.norm( A * x - b, 2 )` of src/solvers.jl:129 and :204. */
int aggmg_residual_norm_dev(aggmg_ctx* ctx, aggmg_op* A, const double* x, const double* b, double* out);
/* multigrid(H, x0, b, maxiter, tol) -> x, iter, res, err       src/solvers.jl:116-139:
 * x <- multigrid_v_cycle(H, x, b) until ||A x - b|| < tol ||b|| (:131) or maxiter cycles.  The residual is checked
 * every `check_every` cycles (1 = the reference's loop; c > 1 runs c cycles per check through aggmg_vcycles_dev);
 * res_hist (host, >= ceil(maxiter / check_every) entries) receives one norm per check.
 * The reference's `err` history -- err[i] = ||x_i - u_exact||_2 with u_exact = H.mStiffness[1] \ b (:120, :128) --
 * is formed on the device when u_exact (device vector) and err_hist (host, as long as res_hist) are given: the
 * caller solves the fine system ONCE (a one-level hierarchy of the fine operator is the direct solve: block cyclic
 * reduction when the operator is block-tridiagonal, host banded LU otherwise -- aggmg_hier_create with nlevels = 1)
 * and no iterate crosses PCIe.  Both NULL: no error history.  All vectors are device pointers; x_out may alias
 * neither x0 nor b. */
int aggmg_multigrid_dev(aggmg_ctx* ctx, aggmg_hier* h, const double* x0, const double* b, int maxiter,
                        double tol, int check_every, int nPre, int nPost, double alpha, double* x_out,
                        double* res_hist, int* n_cycles, int* n_checks, const double* u_exact, double* err_hist);
/* iterative_smoother_solve(A, smoother, x0, b; maxiter, tol, alpha) -> x, iter, res, err
 * src/solvers.jl:189-213; same conventions as above (u_exact = A \ b of :194, err[i] of :202). */
int aggmg_smoother_solve_dev(aggmg_ctx* ctx, aggmg_op* A, aggmg_smoother* sm, const double* x0,
                             const double* b, int maxiter, double tol, double alpha, int check_every,
                             double* x_out, double* res_hist, int* n_iters, int* n_checks, const double* u_exact,
                             double* err_hist);
/* Conjugate gradients on A x = b with ldiv!(y, H, r) (one V-cycle from a zero guess,
 * src/solvers.jl:84-92) as the preconditioner; x_inout holds the initial guess and the result.
 * EXTENSION: the reference provides ldiv! so that a hierarchy can be used as a preconditioner but
 * has no Krylov loop of its own.  Needs A symmetric positive definite and nPre == nPost.
 * res_hist (host, >= maxiter entries): ||r|| of the recurrence after every iteration. */
int aggmg_pcg_dev(aggmg_ctx* ctx, aggmg_hier* h, const double* b, double* x_inout, int maxiter, double tol,
                  int nPre, int nPost, double alpha, double* res_hist, int* n_iters);

/* ---- measurement ---------------------------------------------------------------------------- */
/* HIP-event timing of kernel launches on the context stream, by tag = kind * 16 + level
 * (level < 16; level 0 for the stand-alone fused ops).  While enabled every launch is bracketed
 * by two events; aggmg_profile_collect synchronises and returns per-tag total ms and counts
 * (arrays of AGGMG_PROFILE_NTAGS), then resets. */
#define AGGMG_PROFILE_NTAGS 256
#define AGGMG_KIND_FUSED_DOWN 0 /* [u=0|x0] -> nPre sweeps -> residual -> restriction */
#define AGGMG_KIND_FUSED_UP 1   /* prolong-add -> nPost sweeps */
#define AGGMG_KIND_SMOOTH 2     /* structured multi-sweep smoother kernel */
#define AGGMG_KIND_RESIDUAL 3
#define AGGMG_KIND_RESTRICT 4
#define AGGMG_KIND_PROLONG 5
#define AGGMG_KIND_JACOBI 6      /* generic CSR fused point-Jacobi sweep */
#define AGGMG_KIND_BLOCK_APPLY 7 /* generic gather block apply */
#define AGGMG_KIND_COARSE 8      /* device coarsest solve (all its launches) */
#define AGGMG_KIND_FUSED_MID 9   /* prolong-add -> nPost + nPre sweeps -> restriction (between cycles) */
/* on: 0 = off, 1 = every launch, 2 = only the fine-level fused-down launch (the dominant kernel:
 * one event pair per cycle, so that timing it does not disturb the cycle being timed -- an event pair
 * costs ~7 us of stream time on MI355X). */
int aggmg_profile_enable(aggmg_ctx* ctx, int on);
int aggmg_profile_collect(aggmg_ctx* ctx, double* total_ms, int64_t* counts);
/* Compulsory HBM bytes of one launch: the sizes of the arrays the launch has to read and to write, each counted
 * once, in the format the level stores them (index-free block rows, packed symmetric inverses, transfer rows) --
 * no halo re-reads, no cache effects, no CSR model.  bytes / launch duration / 8 TB/s is the launch's roofline
 * fraction (<= 1 by construction; measurement aid, no reference counterpart).
 * aggmg_hier_launch_bytes: the fused launch of `level` (not the coarsest) that aggmg_vcycle_dev enqueues, kind =
 * AGGMG_KIND_FUSED_DOWN (src/solvers.jl:28-37), _UP (:41-47) or _MID (both, between the cycles of aggmg_vcycles_dev);
 * has_x0: the descent reads an initial guess (level 0).  AGGMG_ERR_UNSUPPORTED on a level that runs the generic kernels.
 * aggmg_smoother_launch_bytes: what = 0 one launch of aggmg_smooth_dev (any number of sweeps that fit one launch),
 * what = 1 aggmg_residual_dev (sm may be NULL). */
int aggmg_hier_launch_bytes(aggmg_ctx* ctx, const aggmg_hier* h, int level, int kind, int has_x0, int64_t* read_bytes,
                            int64_t* write_bytes);
int aggmg_smoother_launch_bytes(aggmg_ctx* ctx, aggmg_op* A, aggmg_smoother* sm, int what, int64_t* read_bytes,
                                int64_t* write_bytes);

/* Library / build identification ("aggmg_hip gfx950 ..."). */
const char* aggmg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* AGGMG_HIP_H */
