/*
 * libdedalus_hip.so -- C ABI of the MI355X (gfx950) implementation of the Dedalus v3 IMEX hot path.
 *
 * Conventions:
 *   - plain C symbols, int status return: 0 = ok, negative = error; ddh_last_error() gives the text
 *   - every pointer is a DEVICE pointer unless its name ends in _h (host)
 *   - opaque uint64_t handles; the library owns plan/workspace memory, the caller owns field buffers
 *   - asynchronous on the caller-supplied hipStream_t (passed as void*; NULL = default stream)
 *   - thread-compatible, not thread-safe: one host thread per GPU
 *   - all floating point data is IEEE double (the reference computes in float64 throughout)
 *
 * Each entry point names the reference interface it replaces (file:line under /root/reference).
 *
 * Data layout ("z-major pencil layout"): a field's coefficient array is
 *   [component][coupled axis (Chebyshev) index][separable axis 0][separable axis 1]
 * with the last axis contiguous.  A transform acts on one axis of the 3-index view
 *   [outer][n][inner]            (n = axis length, inner = product of the faster axes).
 */
#ifndef DEDALUS_HIP_H
#define DEDALUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t ddh_handle;

/* ---- runtime ------------------------------------------------------------------------------- */
int ddh_init(int device);                       /* replaces fftw_mpi_init, libraries/fftw/fftw_wrappers.pyx:23-25 */
const char *ddh_last_error(void);
int ddh_device_count(int *count);
int ddh_alloc(void **ptr, size_t bytes);        /* replaces create_buffer, fftw_wrappers.pyx:28-58 (Field._create_buffer core/field.py:476-485) */
int ddh_free(void *ptr);
int ddh_memset(void *ptr, int value, size_t bytes, void *stream);
int ddh_memcpy_h2d(void *dst, const void *src_h, size_t bytes, void *stream);
int ddh_memcpy_d2h(void *dst_h, const void *src, size_t bytes, void *stream);
int ddh_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream);
int ddh_stream_sync(void *stream);
int ddh_destroy(ddh_handle h);                  /* any plan / pack handle */

/* ---- spectral transforms -------------------------------------------- */
/* RealFourier: replaces FFTWRealFFT.forward/backward core/transforms.py:537-565 with the
 * unpack_rescale / repack_rescale passes (:469-509) fused into the FFT's load/store.
 * Coefficient axis is interleaved [a0,b0,a1,b1,...] (cos, -sin); n_coeff must be even.        */
int ddh_plan_rfft(ddh_handle *plan, int n_grid, int n_coeff);
int ddh_rfft_forward(ddh_handle plan, const double *g, double *c, long outer, long inner, void *stream);
int ddh_rfft_backward(ddh_handle plan, const double *c, double *g, long outer, long inner, void *stream);
/* backward transform of d/dx of the data: DifferentiateRealFourier (core/basis.py:1233-1260) applied
 * while the coefficients are loaded, mode k (cos, msin) -> (-k dscale msin, k dscale cos) with
 * dscale = 2 pi / L; saves transforming the derivative as a separate field.                       */
int ddh_rfft_backward_deriv(ddh_handle plan, const double *c, double *g, long outer, long inner, double dscale,
                            void *stream);

/* Both at once: g = backward transform of c, g_deriv = backward transform of d/dx c, from ONE read of the coefficients
 * (the nonlinear terms need a field and its derivative along the same axis: core/operators.py gradient components).
 * g, g_deriv: distinct buffers that do not overlap c.                                                              */
int ddh_rfft_backward_dual(ddh_handle plan, const double *c, double *g, double *g_deriv, long outer, long inner,
                           double dscale, void *stream);

/* Chebyshev counterpart: g = backward transform of the coefficients c (in the family's own basis, no conversion),
 * g_deriv = backward transform of D c, where D is a one-superdiagonal operator (D c)[k] = dvec[k] c[k+1] into the basis
 * of `plan` -- the Jacobi derivative d/dz T_n -> (a0+1, b0+1) (DifferentiateJacobi, core/basis.py:806-840; the matrix is
 * tools/jacobi.py differentiation_matrix scaled by 1/stretch) -- undone by the plan's conversion solve exactly like
 * ddh_cheb_backward on the stored derivative.  `plan` must be the DERIVATIVE basis's plan (nbands > 0) with the same
 * (n_grid, n_coeff) as the field's.  dvec: device [n_coeff].  Replaces a sparse mat-vec + a second transform.      */
int ddh_cheb_backward_dual(ddh_handle plan, const double *c, double *g, double *g_deriv, const double *dvec, long outer,
                           long inner, void *stream);

/* ComplexFourier: replaces FFTWComplexFFT core/transforms.py:292-330 (resize_coeffs :243-267 fused).
 * Arrays are complex128 stored as interleaved doubles; `inner` counts complex elements.        */
int ddh_plan_cfft(ddh_handle *plan, int n_grid, int n_coeff);
int ddh_cfft_forward(ddh_handle plan, const double *g, double *c, long outer, long inner, void *stream);
int ddh_cfft_backward(ddh_handle plan, const double *c, double *g, long outer, long inner, void *stream);

/* Fused grid stage along the contiguous RealFourier axis: for every line
 *   out[ic] = rfft_forward( sum_t coef[t] * rfft_backward(D a[ia[t]]) * rfft_backward(D b[ib[t]]) )
 * a_h / b_h / out_h are host arrays of device pointers to [nlines][n_coeff] coefficient-line arrays
 * (na <= 3, nb <= 12, nc <= 4, nterms <= 32).  a_dscale_h / b_dscale_h (nullable) select per operand
 * D = d/dy applied while the line is unpacked: a value kappa0 = 2 pi / L != 0 maps mode k
 * (cos, msin) -> (-k kappa0 msin, k kappa0 cos), i.e. DifferentiateRealFourier (core/basis.py:1233-1260);
 * 0 means D = identity.  One launch replaces the last backward transform of every operand, the
 * product (DotProduct / MultiplyFields.operate, core/arithmetic.py:666-674, 855-866) and the first
 * forward transform of the result; the dealiased grid data of that axis never reaches HBM. */
int ddh_rfft_bilinear_fused(ddh_handle plan, int na, const double *const *a_h, const double *a_dscale_h,
                            int nb, const double *const *b_h, const double *b_dscale_h, int nc,
                            double *const *out_h, long nlines, int nterms, const int *ic_h,
                            const int *ia_h, const int *ib_h, const double *coef_h, void *stream);

/* Chebyshev-T grid (a0=b0=-1/2) with optional ultraspherical output basis: replaces
 * FFTWFastChebyshevTransform core/transforms.py:801-902 (FastCosineTransform :715-746, FFTWDCT
 * :771-798, conversion apply :862-874, solve_upper_sparse :876-890).
 * The conversion matrix (n_coeff x n_coeff, upper banded) is passed as `nbands` diagonals with
 * offsets band_offsets_h[d] >= 0 (offset 0 first): bands_h[d*n_coeff + k] = C[k, k+offset_d].
 * nbands = 0 means output basis == grid basis (no conversion).
 * Only dealias_before_converting = True (the reference default, dedalus.cfg:41) is implemented. */
int ddh_plan_cheb(ddh_handle *plan, int n_grid, int n_coeff, int nbands,
                  const int *band_offsets_h, const double *bands_h);
int ddh_cheb_forward(ddh_handle plan, const double *g, double *c, long outer, long inner, void *stream);
/* The same transform with the coefficient rows written TILE-MAJOR: a row of `inner` = nx * row_len doubles is stored as
 * [kx / 8][ky / 8][kx % 8][ky % 8] (64-byte segments of a tile of 8 storage rows x 8 doubles contiguous) instead of
 * [kx][ky].  It is the layout the pencil sweeps read with two contiguous 512-byte runs per wavefront and row
 * (ddh_pencil_solve_recombined_tiled); the forward transform of a right-hand-side product writes the equation rows of the
 * solver's F vector directly in it.  Values are those of ddh_cheb_forward (core/transforms.py:801-902), only their
 * addresses differ.  nx and row_len multiples of 8; strided-axis wave kernel sizes only (N = 384, M = 256), an error
 * otherwise; not in place.                                                                                            */
int ddh_cheb_forward_tiled(ddh_handle plan, const double *g, double *c, long outer, long inner, long row_len, void *stream);
/* x-blocked STAGE layout.  On a three-axis problem the array between the z (Chebyshev) and the x (real Fourier) transforms
 * is [comp][z][kx][ky] in the natural layout: consecutive rows of a z line are a whole [kx][ky] plane (2 MiB at 512 x 512)
 * apart, one page per 64-byte row segment.  In the x-blocked layout [comp][kx / 64][z][kx % 64][ky] they are 64 ky-rows
 * (256 KiB) apart, and the rows of an x line are 64 contiguous rows per block.  It exists only between two transforms:
 *   Chebyshev plan,    value = ny (row length in doubles): the GRID side of every strided transform of the plan
 *                      (output of ddh_cheb_backward / _dual, input of ddh_cheb_forward / _tiled) is x-blocked
 *   real-Fourier plan, value = gz (z planes per component): the COEFFICIENT side of every strided transform of the plan
 *                      (input of ddh_rfft_backward / _deriv / _dual, output of ddh_rfft_forward) is x-blocked;
 *                      outer = components * gz
 *   value = 0 restores the natural layout.  The setting stays until changed and applies to STRIDED-axis transforms only
 * (a contiguous-axis transform of the same plan -- the y axis of a cubic box shares the x axis' plan -- ignores it);
 * strided transforms that cannot honour it (sizes the wave kernels are not instantiated for) return an error instead of
 * using another layout.  Values are
 * those of the plain transforms (core/transforms.py:469-565, 801-902), only addresses differ.                        */
int ddh_fft_set_stage_layout(ddh_handle plan, long value);
/* Rows per block of the blocked layout on the coefficient side of a real-Fourier plan: 64 (default, 0), 128 or 256.  In a
 * sharded run the all-to-all delivers a component as [p][z][nx / P][ky] (core/transposes.pyx:359-445 would now unpack it to
 * [z][nx][ky]): with rows = nx / P that IS this layout, so the x transforms read what arrived and write what leaves -- no
 * unpack / pack pass (ddh_a2a_unpack / ddh_a2a_pack) around the exchange. */
int ddh_fft_set_stage_block(ddh_handle plan, int rows);
/* Diagnostic: the number of Chebyshev / real-Fourier launches of this process that ran on the wave-per-four-line-pairs
 * kernels (csrc/ddh_fftwave.hip; the sizes of its tables) rather than on the workgroup-per-tile kernel every size can
 * take.  The reference's plans are size-generic (core/transforms.py:537-565, 801-902); here the kernel is chosen per
 * (grid, coefficient) size, and tests assert which one a size took.                                                   */
int ddh_fft_wave_launches(long *count);
int ddh_cheb_backward(ddh_handle plan, const double *c, double *g, long outer, long inner, void *stream);

/* Dense matrix-multiply transform along an axis (JacobiMMT core/transforms.py:114-158 via
 * apply_dense tools/array.py:104-129): out[o, i, :] = sum_j mat[i, j] * in[o, j, :].
 * mat_h is row-major (n_out x n_in) float64 on the host; copied to the device by the plan.     */
int ddh_plan_mmt(ddh_handle *plan, int n_out, int n_in, const double *mat_h);
int ddh_mmt_apply(ddh_handle plan, const double *in, double *out, long outer, long inner, void *stream);

/* ---- grouped dense transforms ---------------------------------------------
 * Spin-weighted spherical harmonic colatitude transform: replaces the Python loop over the local
 * azimuthal wavenumbers in SWSHColatitudeTransform.forward_reduced / backward_reduced
 * (core/transforms.py:1258-1288), each iteration an apply_matrix of one dense (Lmax+1-|m|) x Ntheta
 * matrix on a slice of the data, by one launch over all groups.
 * Data are the reference's reduced 4-D views: grid side g[n0][n1g][n_grid][n3], coefficient side
 * c[n0][n1c][n2c][n3] (C order).  A group is one entry of SphereBasis.m_maps (core/basis.py:2939-2970):
 * grid slice [g_start, g_start+count) and coefficient slice [c_start, c_start+count) of axis 1, and the
 * coefficient rows ell_start + r*ell_step (r = 0..n_ell-1; ell_step = -1 for folded modes) of axis 2.
 * mat = index of the group's matrix pair, or -1 for |m| > Lmax (forward skips the group, backward
 * writes zeros, :1268-1283).  fwd matrices are [n_ell][n_grid], bwd matrices [n_grid][n_ell].            */
typedef struct {
    int mat;
    int g_start, c_start, count;
    int ell_start, ell_step, n_ell;
} ddh_mmt_group;
int ddh_plan_grouped_mmt(ddh_handle *plan, int n_grid, int ngroups, const ddh_mmt_group *groups_h, int nmats,
                         const int *mat_rows_h, const double *const *fwd_h, const double *const *bwd_h);
/* Second right-hand-side set per group served by the group's own matrices (few-column / GEMV path): pair_mode 0 none,
 * 1 plain (another tensor component with the same spin weight), 2 mirrored -- the component of opposite spin weight,
 * whose matrices are the colatitude-reversed ones with signs: F_{-s}[l, j] = (-1)^(l + m) F_{+s}[l, N-1-j]
 * (parity = m & 1).  The reference builds and applies the +s and -s matrices separately
 * (core/transforms.py:1251-1340); here only one of them is stored and streamed. */
int ddh_grouped_mmt_set_pairs(ddh_handle plan, int ngroups, const int *pair_g_h, const int *pair_c_h,
                              const int *pair_mode_h, const int *parity_h);
int ddh_grouped_mmt_forward(ddh_handle plan, const double *g, double *c, long n0, long n1g, long n1c, long n2c,
                            long n3, void *stream);
int ddh_grouped_mmt_backward(ddh_handle plan, const double *c, double *g, long n0, long n1g, long n1c, long n2c,
                             long n3, void *stream);

/* ---- regularity recombination of shell / ball tensor fields ------------------
 * In place on data[ncomp][n1][n2][n3] (tensor components, m slots, ell slots, radius): the components of
 * every slot (i1, i2) with k = slot_map_d[i1*n2 + i2] >= 0 are multiplied by the 3^rank x 3^rank matrix
 * mats_d[k] (row major), replacing the per-ell Python loop of forward/backward_regularity_recombination
 * (core/basis.py:3595-3626).  The host composes mats from the intertwiners Q(ell): Q(ell)^T forward,
 * Q(ell) backward, and -- because the reference's ell_maps are bounding-box slices that may overlap --
 * the ordered product of the matrices of all entries that cover a slot
 * (dedalus_amd/core/curvilinear.py::recombination_tables).
 * radial_factor_d (nullable, length n3) multiplies every point: the (dR/r)^(-+k) factor of
 * ShellBasis.forward/backward_transform_radius (core/basis.py:4474-4508).
 * The radial transform itself is the Jacobi transform of row a8 (ddh_cheb_* / ddh_mmt_apply).           */
int ddh_regularity_recombine(double *data, int ncomp, long n1, long n2, long n3, const int *slot_map_d, int nmats,
                             const double *mats_d, const double *radial_factor_d, void *stream);

/* ---- sphere coefficient-space kernels (csrc/ddh_sphere.hip) -------------------
 * Tensor fields on S2 are stored as real arrays [component][2 m + part][n] with part 0/1 = cos/msin of the
 * azimuthal mode m (the reference's real-dtype azimuth layout, core/basis.py:1595-1663), n = ell
 * (coefficient space) or theta (colatitude grid).
 * ddh_spin_recombine: out[(c', p')] = sum mat[(c', p'), (c, p)] in[(c, p)] at every (m, n): coordinate <-> spin
 * components, mat_h = spin_recombination_matrix (core/basis.py:1576-1593), [2 ncomp][2 ncomp] row major.    */
int ddh_spin_recombine(const double *in, double *out, int ncomp, long npairs, long inner, const double *mat_h,
                       void *stream);
/* Linear sphere operators as ell-local term lists: y[co][m][ell] = sum_t coef_t[m][ell] * x[ci_t][m][ell + d_t]
 * (complex numbers cos + i msin).  Covers the SeparableSphereOperator symbols (core/operators.py:2725-2866;
 * grad/div/lap/average/convert symbols core/basis.py:3279-3420, 5296-5320), MulCosine (:2995-3046), SpinSkew
 * (:2125-2147) and their compositions; replaces the per-m CSR products of the subproblem matrices
 * (core/subsystems.py:497-596; timesteppers.py:588-591).  Terms sorted by co; coef_h complex
 * [nterms][nm][nl] (re, im interleaved).                                                              */
int ddh_sphere_terms_create(ddh_handle *h, int nm, int nl, int ncomp_out, int nterms, const int *co_h, const int *ci_h,
                            const int *d_h, const double *coef_h);
int ddh_sphere_terms_apply(ddh_handle h, const double *x, double *y, void *stream);
/* Per-m dense complex systems, unknown j = comp * (nl - m) + (ell - m): y_m = A_m x_m for all m in one launch
 * (the LHS inverses of the per-m subproblems; replaces the per-m SuperLU solves, libraries/matsolvers.py:
 * 126-149 / timesteppers.py:630-643).  mats_h: the nm row-major complex matrices concatenated.          */
int ddh_cgemv_batch_create(ddh_handle *h, int nm, int nl, int ncomp, const double *mats_h);
int ddh_cgemv_batch_apply(ddh_handle h, const double *x, double *y, void *stream);
/* Shell fields [component][2 m + part][ell][n]: y[co][i1][ell][:] = sum_t A_t[id] x[ci_t][i1][ell][:] with real
 * radial matrices (nr x nr, row major in mats_h [nterms][nmat][nr][nr]) selected per slot by
 * id = slot_map_h[i1 * nl + ell] (-1: the slot carries no mode and is zeroed; normally id = ell).  Replaces
 * SphericalEllOperator.operate / subproblem_matrix (core/operators.py:3108-3222) and, with the per-ell LHS inverses
 * as matrices, the per-ell subproblem solves.  Terms sorted by co.                                              */
int ddh_ell_terms_create(ddh_handle *h, int nm, int nl, int nr, int ncomp_out, int nterms, const int *co_h,
                         const int *ci_h, int nmat, const double *mats_h, const int *slot_map_h);
int ddh_ell_terms_apply(ddh_handle h, const double *x, double *y, void *stream);
/* accumulate != 0: y += A x (dense term lists only: the FP64 MFMA per-ell GEMM path); used to apply a term list
 * split into its banded part (first, writes y) and its dense blocks (second, accumulates).              */
int ddh_ell_terms_apply_acc(ddh_handle h, const double *x, double *y, int accumulate, void *stream);

/* ---- device factorization of the curvilinear subproblems (SURVEY 8a row a9 for configs S and H) ------------------
 * (a M + b L)^-1 for a batch of small dense systems (n <= 1024), real or complex, formed and inverted ON THE DEVICE
 * (in-place Gauss-Jordan with partial pivoting, one workgroup per system): replaces the per-subproblem SuperLU
 * factorizations the reference repeats whenever a0 / b0 or k H_ii change (core/timesteppers.py:172-181, 630-640,
 * libraries/matsolvers.py:126-149).  M_h / L_h: the nsys row-major n_s x n_s matrices concatenated (complex:
 * interleaved re, im); row_valid_h / col_valid_h: concatenated per-system 0/1 masks of the equation / variable modes a
 * system really has (valid-mode filtering, core/subsystems.py:540-556) -- the result is the inverse of the valid block
 * embedded in zeros.  compute() writes all inverses, concatenated like the inputs, to out_d (device);
 * nsingular_h (nullable; forces a stream sync) counts systems that hit a zero pivot.                            */
int ddh_dense_inverse_create(ddh_handle *h, int nsys, const int *n_h, int is_complex, const double *M_h,
                             const double *L_h, const unsigned char *row_valid_h, const unsigned char *col_valid_h);
int ddh_dense_inverse_elements(ddh_handle h, long *count_doubles);
int ddh_dense_inverse_compute(ddh_handle h, double a, double b, double *out_d, int *nsingular_h, void *stream);
/* Consumers of those inverses.  ddh_cgemv_batch_create accepts mats_h = NULL (storage only); ddh_cgemv_batch_mats gives
 * the device address of its matrices, the layout ddh_dense_inverse_compute writes for systems ordered by m.
 * ddh_ell_terms_create_dense makes a term list of all ncomp x ncomp dense blocks (FP64 MFMA GEMM path) without host
 * data; ddh_ell_blocks_from_dense fills its storage (ddh_ell_terms_mats) from per-ell inverses [nl][ncomp nr][ncomp nr]. */
int ddh_cgemv_batch_mats(ddh_handle h, double **mats_d);
int ddh_ell_terms_create_dense(ddh_handle *h, int nm, int nl, int nr, int ncomp);
int ddh_ell_terms_mats(ddh_handle h, double **mats_d);
int ddh_ell_blocks_from_dense(const double *inv_d, double *mats_d, int nl, int ncomp, int nr, void *stream);
int ddh_ell_terms_prune(ddh_handle h, void *stream);      /* drop blocks that are zero for every ell from the GEMM */

/* ---- band LU of the curvilinear subproblems (SURVEY 8a row a9 for config H; VERDICT r3 "block-banded LU") ---------
 * The per-group (shell: per-ell) matrices a M + b L, permuted and column-recombined once on the host
 * (dedalus_amd/core/ellband.py) into bands of kl sub / ku super diagonals, are formed, factorized (partial pivoting,
 * gbtrf-shaped) and swept on the device: replaces the reference's per-subproblem sparse LU + solve
 * (core/timesteppers.py:172-181, 630-640; libraries/matsolvers.py:129-160) at O(n kl (kl + ku)) per factorization and
 * O(n (kl + ku)) per right-hand side instead of the dense inverse's O(n^3) / O(n^2).
 * create: nl groups of n_h[g] <= nmax unknowns (0: skipped); rowoff_h / coloff_h [nl][nmax]: element offset of permuted
 * row / column i inside one slot of the [component][slot][group][n] system vectors (slot_stride elements apart);
 * slot_limit_h[g]: leading slots that can hold modes of group g; nbc_h[g] <= 8 boundary rows with their combination
 * T_h [nl][max(nbc,1)]^2; P_h [nl][nmax][max(mp,1)]: super diagonals of the recombination in the permuted order
 * (mp <= 16); MB_h / LB_h [nl][nmax][kl + ku + 1]: row i holds columns i - kl .. i + ku.
 * factor: factorization number `index` (== count so far: a new one) of a M + b L; nsingular_h (nullable; syncs) counts
 * zero pivots.  solve: x = (a M + b L)^-1 rhs on the valid modes of the banded groups (x cleared by the caller). */
int ddh_ellband_create(ddh_handle *h, int nl, int nmax, int kl, int ku, int mp, int nbc, int nslots, long slot_stride,
                       const int *n_h, const int *nbc_h, const int *slot_limit_h, const long *rowoff_h,
                       const long *coloff_h, const double *T_h, const double *P_h, const double *MB_h,
                       const double *LB_h);
int ddh_ellband_factor(ddh_handle h, int index, double a, double b, int *nsingular_h, void *stream);
int ddh_ellband_solve(ddh_handle h, int index, const double *rhs_d, double *x_d, void *stream);
int ddh_ellband_info(ddh_handle h, int *nw, int *wt, long *factor_bytes);
/* Complex per-m inverses (the layout of ddh_dense_inverse_compute / ddh_cgemv_batch_mats, complex offsets off_d[m]) from
 * unit solves of the real-form transposed systems: x_d [2 ncomp][nslots][nm][nl], components 2c / 2c + 1 = real / imaginary
 * part, slot (nl - 1 - ell) ncomp + c = the solve with the unit right-hand side of unknown (c, ell), i.e. row (c, ell) of
 * the inverse.  The sphere's timestep change: O(n b^2 + n^2 b) per m instead of the O(n^3) Gauss-Jordan inversion. */
int ddh_ellband_gather_complex_inverse(const double *x_d, double *out_d, const long *off_d, int ncomp, int nl, int nm,
                                       int nslots, void *stream);

/* Inverse of a bordered pencil whose band block B (n x n, permuted order) is singular only through ONE vanishing column
 * j0 -- the k = 0 subproblem of a Cartesian problem with a pressure gauge (tau_p, "integ(p) = 0"; the reference factors
 * it like every other subproblem, libraries/matsolvers.py:126-149, core/timesteppers.py:630-640) -- from X_d = B2^-1
 * ([n][n] row-major: n unit solves of the band LU of B2 = B with column j0 replaced by the gauge variable's column):
 * wM_d / wL_d [n] = the M / L parts of the gauge row over the columns of B2 (its corner entry at j0), dM / dL = its entry
 * in the free mode's column.  out_d: the (n + 1)^2 inverse, row-major (rows = unknowns, the gauge variable last).
 * Replaces the O(n^3) dense inversion of that one pencil at every change of the timestep. */
int ddh_ellband_bordered_inverse(const double *X_d, int n, int j0, const double *wM_d, const double *wL_d, double dM,
                                 double dL, double a, double b, double *out_d, void *stream);

/* ---- grid-space and vector kernels -------------------------------- */
/* y[idx[i]] += vals[i] for n distinct indices (device arrays): the constant right-hand-side entries
 * (e.g. "b(z=0) = Lz", gathered into F by gather_outputs core/timesteppers.py:611-614) touch a handful of
 * rows of the k = 0 pencil only. */
int ddh_scatter_add(double *y, const long *idx_d, const double *vals_d, long n, void *stream);
/* y[idx[i]] = vals[i]: the same entries when the right-hand-side rows are written directly by the forward transforms
 * (the constant rows of F are then set, not accumulated; core/timesteppers.py:611-614). */
int ddh_scatter_set(double *y, const long *idx_d, const double *vals_d, long n, void *stream);

/* Round 6: the STATE vector of a pack kept tile-major.  The reference keeps one layout for every system vector
 * (core/subsystems.py:497-596 gather / scatter into per-pencil vectors; core/timesteppers.py:588-643); here the state X --
 * written by every solve, read by every mat-vec and by the backward z transforms -- may be stored like the right-hand-side
 * vectors, [kx / 8][ky / 8][kx % 8][ky % 8] within a row, so that a wavefront's stores of a solution row are two 512-byte
 * runs.  ddh_pencil_set_state_tiled switches the solves' output / the mat-vecs' input of a pack, ddh_fft_set_coeff_tiled
 * the coefficient side of a Chebyshev plan's next strided transforms, ddh_tile_rows converts rows between the layouts
 * (user access to a state field, output, generic operators).  Mode 2 / band_rows = R goes one step further: the rows of
 * a band of 8 storage rows together, [kx / 8][R][ky / 8][kx % 8][ky % 8] -- the rows of a pencil are 8 ny doubles apart
 * instead of one nx x ny plane (32 KiB instead of 2 MiB at 512 x 512), for the sweeps' stores and the z transforms' loads. */
int ddh_pencil_set_state_tiled(ddh_handle pack, int on);     /* 0 natural, 1 tile-major rows, 2 kx-band-major (below) */
int ddh_fft_set_coeff_tiled(ddh_handle plan, long row_len, long band_rows);
int ddh_tile_rows(const double *src, double *dst, long nrows, long nx, long ny, int to_tiled, long band_rows, void *stream);

/* y = sum_t alpha[t] * x_t  (RHS assembly timesteppers.py:617-623 / :156-166; BLAS axpy chain).
 * xs_h: host array of nterms device pointers; y may alias one of them only if it is xs_h[0].   */
int ddh_lincomb(double *y, int nterms, const double *const *xs_h, const double *alpha_h,
                long n, void *stream);
/* out[c] = sum_t coef[t] * a[ia[t]] * b[ib[t]] over `n` grid points per component: covers
 * MultiplyFields / DotProduct / CrossProduct operate() (core/arithmetic.py:666-674,708-728,855-866). */
int ddh_grid_bilinear(double *out, int ncomp_out, const double *a, const double *b, long n,
                      int nterms, const int *ic_h, const int *ia_h, const int *ib_h,
                      const double *coef_h, void *stream);
/* max over grid points of sum_c |u_c| / dx_c  (AdvectiveCFL core/operators.py:4342-4419 with the
 * spacings of CartesianAdvectiveCFL core/basis.py:6078-6111).  u is [ncomp][grid, naxes storage
 * axes]; inv_spacing_comp[c] is a device array of 1/dx along storage axis comp_axis_h[c].        */
int ddh_grid_cfl(double *result_d, const double *u, int ncomp, long n,
                 const double *const *inv_spacing_comp, const int *comp_axis_h,
                 const long *axis_len_h, int naxes, void *stream);

/* Spherical shells: max over the grid of sqrt(u_phi^2 + u_theta^2) * inv_h[r] + |u_r| * inv_dr[r] with
 * u = [3][n_ang][nr] coordinate components (Spherical3DAdvectiveCFL, core/basis.py:6183-6204: horizontal spacing
 * r / sqrt(Lmax (Lmax + 1)), radial spacing of the dealiased Gauss grid).                                 */
int ddh_grid_cfl_spherical(double *result_d, const double *u, long n_ang, int nr, const double *inv_h_d,
                           const double *inv_dr_d, void *stream);

/* out3_d = {min, max, sum} over n grid values (device), the local reductions of GlobalArrayReducer / GlobalFlowProperty
 * (extras/flow_tools.py:9-47, 49-111: np.min / np.max / np.sum of the grid data followed by an 8-byte Allreduce).
 * work_d: 3 * 1024 doubles of scratch.  Fixed reduction tree: the sum is reproducible from run to run.           */
#define DDH_REDUCE_WORK_DOUBLES 3072
int ddh_grid_reduce(double *out3_d, const double *x, long n, double *work_d, void *stream);

/* ---- pencil systems ------------------------------------------ */
/* A "pencil pack" describes all pencils of a problem at once.  System vectors are real arrays
 * [nrows][nx][ny] (cell index fastest).  With nfourier real-Fourier separable axes a cell holds
 * 2^nfourier real parts which are combined on the fly into complex systems.
 *
 * Matrices are "polynomial term lists": entry (row,col) += coef * kx^ex * ky^ey * [mx==0]^dx * [my==0]^dy.
 * This replaces the per-pencil scipy CSR matrices of Subproblem.build_matrices
 * (core/subsystems.py:497-596) and the per-pencil Python loops of timesteppers.py:588-591,630-643. */
typedef struct {
    int nfourier;          /* 0, 1 or 2 separable real-Fourier axes                                */
    int nrows;             /* rows of the system vector (all variables x coupled-axis modes)       */
    long nx, ny;           /* real storage extents of the two cell axes (ny = 1 ... if unused)     */
    const double *kx_h;    /* physical wavenumber per x mode index (nx/2 entries), host            */
    const double *ky_h;    /* physical wavenumber per y mode index (ny/2 entries), host            */
    long mx_offset;        /* global x mode index of the first local pencil (0 on one GPU): pencils
                              are sharded over ranks along x like Layout.local_chunks,
                              core/distributor.py:357-385                                          */
} ddh_pencil_geom;

typedef struct {
    int nterms;
    const int *row_h, *col_h;       /* nterms each                                                  */
    const double *coef_re_h, *coef_im_h;
    const signed char *ex_h, *ey_h; /* monomial exponents                                           */
    const signed char *dx_h, *dy_h; /* 1 -> term only present where mx==0 / my==0                   */
} ddh_polymat;

int ddh_pencil_create(ddh_handle *pack, const ddh_pencil_geom *geom);
/* register a matrix; returns its id in *mat_id */
int ddh_pencil_add_matrix(ddh_handle pack, const ddh_polymat *mat, int nrows_out, int *mat_id);
/* y[nrows_out][cells] = A x  (apply_sparse tools/array.py:171-203 over all pencils at once;
 * gather/scatter subsystems.py:340-380 are the identity in this layout)                           */
int ddh_pencil_matvec(ddh_handle pack, int mat_id, const double *x, double *y, void *stream);
/* The same for a buffer the caller keeps for this product alone (the M.X vectors of a timestepper, core/timesteppers.py:588-591
 * with its CoeffSystem buffers core/system.py:39-60, zero-initialised): rows of
 * y whose matrix row has no terms MAY be left untouched instead of being overwritten with zeros (the window-form kernel
 * skips those stores; a fifth of M.X for an incompressible flow).  Every other row is written as by ddh_pencil_matvec.  */
int ddh_pencil_matvec_update(ddh_handle pack, int mat_id, const double *x, double *y, void *stream);
/* Mat-vec fused with an upper-banded back-substitution along the coupled index of every output
 * component (rows = (component, kz), nz per component): y <- C^-1 (A x).  With C the ultraspherical
 * conversion matrix this yields the coefficients directly in the grid (Chebyshev-T) basis, i.e. the
 * solve_upper_sparse step of FastChebyshevTransform.backward (core/transforms.py:876-890) is done
 * here, per pencil thread, instead of inside the transform.  Bands as in ddh_plan_cheb.           */
int ddh_pencil_add_upper_bands(ddh_handle pack, int nz, int nbands, const int *offsets_h,
                               const double *bands_h, int *bands_id);
int ddh_pencil_matvec_solve(ddh_handle pack, int mat_id, int bands_id, const double *x, double *y,
                            void *stream);
/* Bordered-banded LU of (a*M + b*L) for every pencil (matsolvers.py:126-149 SuperLU replaced;
 * LHS formation timesteppers.py:172-181, 630-640).
 * perm_h: logical->physical row/col permutations, n_interior leading logical rows/cols form the
 * banded block with lower/upper bandwidth kl/ku, the remaining nrows-n_interior are the border.
 * valid masks: row_axes_h/col_axes_h[r] bit0: row exists for mx>0, bit1: exists for my>0.
 * Entries outside the declared band send the affected pencil to the dense fallback below.
 * reuse_lu_id >= 0 re-factors into the storage of an existing LU (dt change), -1 allocates.
 * Returns an LU id in *lu_id.                                                                      */
int ddh_pencil_factor(ddh_handle pack, int matM_id, int matL_id, double a, double b,
                      const int *row_perm_h, const int *col_perm_h, int n_interior, int kl, int ku,
                      const unsigned char *row_axes_h, const unsigned char *col_axes_h,
                      int reuse_lu_id, int *lu_id, void *stream);
/* Real graded variant.  For real differential operators the complex pencil symbol factors as
 *   lambda(kx, ky) = D_r A D_c^-1,   lambda(-kx, ky) = D_r S_r A S_c D_c^-1
 * with a REAL matrix A(kx, ky), D = diag(i^rot) and S = diag((-1)^sgn) (a Z2 grading of rows and
 * columns by derivative parity; the host checks that it exists).  One real factorization then serves
 * both systems of a cell and both their real/imaginary parts: 1/4 of the LU bytes of the complex
 * path.  matM / matL must be the real-graded term lists (imaginary parts ignored);
 * row_code_h / col_code_h (logical order): bit0 = rot, bit1 = sgn.                                 */
int ddh_pencil_factor_real(ddh_handle pack, int matM_id, int matL_id, double a, double b,
                           const int *row_perm_h, const int *col_perm_h, int n_interior, int kl, int ku,
                           const unsigned char *row_axes_h, const unsigned char *col_axes_h,
                           const unsigned char *row_code_h, const unsigned char *col_code_h,
                           int reuse_lu_id, int *lu_id, void *stream);
int ddh_pencil_solve(ddh_handle pack, int lu_id, const double *rhs, double *x, void *stream);
/* Solve with the right-hand side given as a linear combination rhs = sum_t alpha[t] * xs[t] of stored system vectors
 * (1 <= nterms <= 8): the RHS assembly of the IMEX schemes (timesteppers.py:156-166, 617-623: an axpy chain into a RHS
 * buffer, then LHS_solver.solve) fused into the forward sweep -- the combined vector is never written to or re-read
 * from HBM.  xs_h: host array of device pointers; none may alias x.                                              */
int ddh_pencil_solve_lincomb(ddh_handle pack, int lu_id, int nterms, const double *const *xs_h, const double *alpha_h,
                             double *x, void *stream);
/* x = P (a M + b L P)^-1 rhs: the solve of the column-recombined system (X = P Y makes the
 * boundary rows sparse) including the recombination.  p_mat_id: the registered constant matrix P.  When P is unit upper
 * banded in the LU's ordering (the Dirichlet / Neumann recombinations are) and the one-thread-per-system backward
 * sweep runs, x_j = y_j + sum_d P[j, j+d] y_(j+d) is formed from the sweep's register window and written directly --
 * neither y nor a separate P.y mat-vec touch HBM; otherwise y goes to `work` and ddh_pencil_matvec(P) follows.
 * work: a system vector of scratch, distinct from x and the right-hand-side terms.                                */
int ddh_pencil_solve_recombined(ddh_handle pack, int lu_id, int nterms, const double *const *xs_h,
                                const double *alpha_h, int p_mat_id, double *work, double *x, void *stream);
/* The same with a row mask: zero_rows[r] != 0 (device array, one byte per row of the system vectors) promises that row r is
 * zero in EVERY right-hand-side term -- rows of equations without time derivative and without right-hand side (the
 * continuity equation of core/timesteppers.py:588-623's M.X / F vectors).  The forward sweep then does not read them
 * (a fifth of the right-hand-side traffic of 3-D Rayleigh-Benard).  Kernels without the shortcut ignore the mask, which
 * is always correct.  skip_rows[r] != 0 (same shape, rows of the SOLUTION): the caller does not need unknown r of x -- an
 * intermediate Runge-Kutta stage never reads the pressure and the tau variables (no mass-matrix column, not an operand of F)
 * -- and the backward sweep does not store it; x keeps whatever it held there.  null masks = ddh_pencil_solve_recombined. */
int ddh_pencil_solve_recombined_sparse(ddh_handle pack, int lu_id, int nterms, const double *const *xs_h,
                                       const double *alpha_h, int p_mat_id, double *work, double *x,
                                       const unsigned char *zero_rows, const unsigned char *skip_rows, void *stream);
/* Sweep variant used by ddh_pencil_solve (all variants compute the same factorization's solution; they differ in how
 * many lanes share one system).  mode 1 (default): chosen by the number of systems; 0: one
 * thread per system; 2: cooperative (16 lanes) in both sweeps.  fwd = 0 / 1 and backward_lanes = 0 / 4 / 16 override
 * the forward and backward kernel individually, -1 leaves the choice to `mode`.  The defaults can also be preset with
 * the environment variables DDH_SOLVE_COOP / DDH_COOP_FWD / DDH_COOP_CB, which are read ONCE, by ddh_pencil_create. */
int ddh_pencil_set_solve_variant(ddh_handle pack, int mode, int fwd, int backward_lanes);
/* Partner pencils.  For a problem that is symmetric under the exchange of its two Fourier axes (same box length and
 * size along x and y, isotropic equations: 3-D Rayleigh-Benard is), the pencil matrices obey
 *     lambda(ky, kx) = Pi_r lambda(kx, ky) Pi_c
 * with Pi_r / Pi_c the row / column involutions that swap the x and y components of every vector (tensor) equation /
 * variable.  The pencil (my, mx) is then solved with the factorization of (mx, my): permuted right-hand side in,
 * permuted solution out (its -kx system through lambda(kx, -ky) = conj lambda(-kx, ky)).  Factorizations made after this
 * call with at least min_systems systems store one factorization per PAIR of cells (cells on the axes and on the
 * diagonal keep their own): factor memory, factor time and the solve's factor stream -- 7/8 of its HBM bytes -- halve;
 * the four systems of a pair sit in adjacent lanes, so the shared loads coalesce in the wavefront.  The reference has no
 * counterpart (it factors every pencil, core/subsystems.py:497-596); the solutions are those of the same linear systems.
 * row_swap_h / col_swap_h: PHYSICAL row / column -> its image, involutions of 0..nrows-1 (the caller has verified the
 * symmetry on its term lists); NULL, NULL switches pairing off.  Needs the real-graded factorization
 * (ddh_pencil_factor_real), two Fourier axes, a square unsharded cell grid with kx_h == ky_h; otherwise ignored.   */
int ddh_pencil_set_pairing(ddh_handle pack, const int *row_swap_h, const int *col_swap_h, long min_systems);
/* Tile-major right-hand-side vectors (two Fourier axes, nx and ny multiples of 8).  The sweeps hand a wavefront the 16
 * cells of a 4 x 4 tile and of its transposed partner tile; in the natural row layout [nx][ny] its 64 accesses to a row
 * of a system vector are sixteen 64-byte runs in eight storage rows, in the tile-major layout [kx/8][ky/8][kx%8][ky%8]
 * two contiguous 512-byte runs (measured: -1.4 ms of a 6.8 ms solve at 512 x 512 x 256 with contiguous term reads).
 *   ddh_pencil_matvec_update_tiled      y = A x like ddh_pencil_matvec_update, y written tile-major (window-form
 *                                       matrices only: the timesteppers' M.X products); x in the natural layout
 *   ddh_pencil_solve_recombined_tiled   ddh_pencil_solve_recombined_sparse whose term vectors xs_h are ALL tile-major;
 *                                       x (the state) and work keep the natural layout.  Needs the lean forward sweep
 *                                       (ddh_pencil_lu_info), an error otherwise.
 * The reference gathers every pencil's right-hand side into a dense buffer (core/subsystems.py:340-380); the values and
 * the linear systems are the same, only the addresses of the solver-internal vectors differ.                          */
int ddh_pencil_matvec_update_tiled(ddh_handle pack, int mat_id, const double *x, double *y, void *stream);
int ddh_pencil_solve_recombined_tiled(ddh_handle pack, int lu_id, int nterms, const double *const *xs_h,
                                      const double *alpha_h, int p_mat_id, double *work, double *x,
                                      const unsigned char *zero_rows, const unsigned char *skip_rows, void *stream);
/* Independent diagonal blocks.  When the band block of the ordered pencil matrix (rows / columns < n_interior of
 * ddh_pencil_factor*) is block diagonal with `nblocks` blocks of EQUAL size -- the connected components of the matrix,
 * e.g. the two reflection parities of constant-coefficient equations between two plates once the boundary rows are
 * decoupled (dedalus_amd/core/solvers.py::_decouple_boundary_rows) -- the caller orders it block after block (its
 * bandwidth is then that of one block) and announces the count here before factoring.  The factorization is the plain
 * band LU of that matrix (partial pivoting never leaves a block: the candidates of another block are exact zeros); the
 * one-thread-per-system sweeps of the real-graded two-axis path then run one thread per (system, block), so the chain
 * of dependent rows of a sweep is n / nblocks long.  Every other sweep variant treats the matrix as the band matrix it
 * is.  The caller guarantees the structure (no entry couples two blocks).  The reference factors the coupled sparse
 * matrix per pencil (libraries/matsolvers.py:126-149, core/subsystems.py:497-596); the solutions are those of the same
 * linear systems.  nblocks = 1 switches it off.                                                                      */
int ddh_pencil_set_row_blocks(ddh_handle pack, int nblocks);
/* Pencils whose band block is singular (e.g. the kx=ky=0 pressure-gauge pencil) are flagged by
 * ddh_pencil_factor and solved with an explicit dense inverse the host supplies: query the flagged
 * cell ids, then upload inverses in logical (permuted) ordering, complex row-major N x N per
 * system, systems ordered (flagged cell, s).                                                      */
int ddh_pencil_flagged(ddh_handle pack, int lu_id, int *count, long *cells_h, int max_cells);
int ddh_pencil_set_dense_inverse(ddh_handle pack, int lu_id, const double *inv_h);
/* the same from device memory, one system (index f * S + s) at a time, real or interleaved complex N x N row-major:
 * formed by ddh_dense_inverse_compute, so a timestep change costs no host linear algebra for flagged pencils either */
int ddh_pencil_set_dense_inverse_dev(ddh_handle pack, int lu_id, int sys, const double *inv_d, int is_complex, void *stream);
/* Few systems with one Fourier axis (2-D problems): explicit inverses of the independent diagonal blocks of a real-graded
 * factorization (ddh_pencil_set_row_blocks; nblocks = 1: the whole band block), TRANSPOSED -- binv_d [cell][block][k][i] =
 * (block^-1)[i][k], caller-owned device memory, formed e.g. by unit solves of ddh_ellband_* -- make ddh_pencil_solve_recombined*
 * apply them as streaming GEMVs instead of running the sweeps (a chain of n / nblocks dependent rows that a few hundred
 * systems cannot hide).  The solutions are those of the same systems the reference factors per subproblem
 * (libraries/matsolvers.py:126-149).  Border unknowns must exist for flagged pencils only.  binv_d = NULL: back to the sweeps. */
int ddh_pencil_set_block_inverse(ddh_handle pack, int lu_id, const double *binv_d);
int ddh_pencil_lu_bytes(ddh_handle pack, int lu_id, size_t *bytes);
/* Shape of a factorization and the sweep kernels ddh_pencil_solve* will launch for it (diagnostics, byte accounting,
 * tests): info_h[12] = { n (band rows), nb (border), kl, ku, W = kl + ku, BW (stored entries per band row), nsplit
 * (independent diagonal blocks swept by separate threads, ddh_pencil_set_row_blocks), rows per block, forward sweep
 * (0 general one-thread-per-system, 1 lean one-thread-per-system -- the one that honours zero_rows --, 2 cooperative),
 * backward lanes per system (0 = one thread per system -- the one that honours skip_rows and fuses x = P y --, 4, 16),
 * partner pencils (0 / 1), real-graded (0 / 1) }.  No reference counterpart (SuperLU objects are opaque,
 * libraries/matsolvers.py:126-149).                                                                                  */
int ddh_pencil_lu_info(ddh_handle pack, int lu_id, int *info_h);
/* wrow_h[j] (n_interior ints) = max over all factorizations of the last non-zero super-diagonal offset of U row j:
 * how much of the partial-pivoting fill space (kl extra super-diagonals, LAPACK gbtrf storage) is really used.  */
int ddh_pencil_lu_row_widths(ddh_handle pack, int lu_id, int *wrow_h);

/* ---- distributed transposes (boundary B3) ------------------------------------ */
/* Communicator: one process per GPU, RCCL over xGMI, owned by the library (replaces the mpi4py communicator the
 * reference plans its transposes on, core/distributor.py:696-768).  Rank 0 obtains an id and hands it to the other
 * ranks through any out-of-band channel (the launcher's store, a file, MPI_Bcast); then every rank calls
 * ddh_comm_create, collectively.  RCCL is bound at run time (librccl.so.1).  Destroy plans before their communicator. */
#define DDH_COMM_ID_BYTES 128
int ddh_comm_probe(void);                                          /* 0 when RCCL can be bound in this process: lets the
                                                                    * ranks agree BEFORE the collective ddh_comm_create */
int ddh_comm_unique_id(unsigned char *id_h);                       /* DDH_COMM_ID_BYTES bytes */
int ddh_comm_create(ddh_handle *comm, int rank, int nranks, const unsigned char *id_h);
/* Rank emulation on ONE GPU (no counterpart in the reference, whose transposes need a real MPI run,
 * core/transposes.pyx:359-445): a communicator that IS rank `rank` of `nranks` but has no peers -- every exchange hands
 * back, as the block "received" from peer p, the block this rank sends to p (a device copy on the same stream).  Pack /
 * unpack kernels, buffer sizes, stream ordering and the rank's kernel shapes are those of the P-rank run; the values
 * are not.  For timing one rank's share of a sharded problem (tools/rank_emulation.py), never for results.            */
int ddh_comm_create_loopback(ddh_handle *comm, int rank, int nranks);
int ddh_comm_info(ddh_handle comm, int *rank, int *nranks);
/* in-place all-reduce of `count` doubles: op 0 sum, 1 max, 2 min (the MPI Allreduce of GlobalArrayReducer,
 * extras/flow_tools.py:9-47, and of the CFL frequency) */
int ddh_comm_allreduce(ddh_handle comm, double *buf, long count, int op, void *stream);
/* equal-split all-to-all on caller-owned device buffers (the MPI Alltoall of the transposes, core/transposes.pyx:329-358,
 * for callers that pipeline pack / exchange / unpack themselves, e.g. one field component at a time on a side stream):
 * block p (`chunk` doubles) of `send` goes to rank p, block q of `recv` comes from rank q; grouped ncclSend / ncclRecv,
 * the rank's own block as a device copy; asynchronous on `stream` */
int ddh_comm_alltoall(ddh_handle comm, const double *send, double *recv, long chunk, void *stream);
/* a PART of such an exchange: `count` doubles per peer, the peers' blocks `peer_stride` doubles apart; send / recv point at
 * the part inside block 0.  One window of the planes of a component stored [p][z][rows][ky] (the reference exchanges whole
 * fields, core/transposes.pyx:329-358; windows let the grid stage of one window run while the next is on the wire). */
int ddh_comm_alltoall_part(ddh_handle comm, const double *send, double *recv, long count, long peer_stride, int nbatch,
                           long batch_stride, void *stream);      /* nbatch such exchanges batch_stride apart (the components of a field) as ONE group */
/* ... and the x transforms of one window: planes z0 .. z0 + nplanes of every component of the blocked stage array
 * (ddh_fft_set_stage_layout / _block), the other side holding nplanes planes per component; nplanes = 0: all planes */
int ddh_fft_set_stage_window(ddh_handle plan, int z0, int nplanes);

/* Transpose plan = FFTWTranspose / AlltoallvTranspose (core/transposes.pyx:22-445, planner interface
 * core/distributor.py:696-768): (n0, n1, n2, n3) is the reference's reduced GLOBAL shape (N0, N1, N2, N3) around the
 * transposed axis pair (axis, axis + 1); axes that are not divisible by the number of ranks are dealt out in blocks of ceil(n / P) (uneven
 * blocks).  Local layouts, C order:
 *     column-local CL [n0][n1][n2 / P][n3]      row-local RL [n0][n1 / P][n2][n3]
 * ddh_a2a_localize_rows(plan, CL, RL)     = plan.localize_rows(CL, RL)     (FFTWTranspose :211-220, AlltoallvTranspose :329-342; Transpose.decrement, forward)
 * ddh_a2a_localize_columns(plan, RL, CL)  = plan.localize_columns(RL, CL)  (:237-246 / :344-358; Transpose.increment, backward)
 * ddh_a2a_forward / ddh_a2a_backward are the same two calls under the transform-direction names.
 * Each is pack kernel -> grouped ncclSend/ncclRecv to every peer -> unpack kernel, asynchronous on `stream`; source
 * and destination must be different buffers (the reference's CL/RL views alias one FFTW buffer and are transposed
 * through an internal copy; here the plan owns the two staging buffers). */
int ddh_a2a_plan(ddh_handle *plan, ddh_handle comm, long n0, long n1, long n2, long n3);
/* the same with explicit block sizes along the two transposed axes (0 = ceil(n / P)) */
int ddh_a2a_plan_blocks(ddh_handle *plan, ddh_handle comm, long n0, long n1, long n2, long n3, long block1, long block2);
int ddh_a2a_localize_rows(ddh_handle plan, const double *cl, double *rl, void *stream);
int ddh_a2a_localize_columns(ddh_handle plan, const double *rl, double *cl, void *stream);
int ddh_a2a_forward(ddh_handle plan, const double *cl, double *rl, void *stream);
int ddh_a2a_backward(ddh_handle plan, const double *rl, double *cl, void *stream);
/* The two local re-ordering kernels of a transpose on their own (split_rows / split_columns and their inverses,
 * core/transposes.pyx:359-445), for callers that issue the exchange themselves (e.g. torch.distributed):
 * pack splits axis `a` (length na) of [outer][na][nb][inner] into P blocks and writes [P][outer][na/P][nb][inner];
 * unpack reads [P][outer][na][nb/P][inner] and gathers along nb into [outer][na][nb][inner].                    */
int ddh_a2a_pack(const double *src, double *dst, long outer, long na, long nb, long inner,
                 int nparts, void *stream);
int ddh_a2a_unpack(const double *src, double *dst, long outer, long na, long nb, long inner,
                   int nparts, void *stream);
/* the same with UNEVEN blocks of ceil(n / nparts) along the split axis (the last ranks own less, possibly nothing):
 * pack [outer][na][row] -> blocks p = [outer][na_p][row] back to back; unpack blocks p = [outer_na][nb_p][inner] ->
 * [outer_na][nb][inner].  ddh_a2a_plan uses them when an axis is not divisible by the number of ranks (the reference's
 * Alltoallv transposes, core/transposes.pyx:287-445). */
int ddh_a2av_pack(const double *src, double *dst, long outer, long na, long row, int nparts, void *stream);
int ddh_a2av_unpack(const double *src, double *dst, long outer_na, long nb, long inner, int nparts, void *stream);
/* ... with an explicit block size (0 = ceil(n / nparts)): the reference's blocks are whole chunks,
 * chunk * ceil(ceil(n / chunk) / P) (core/distributor.py Layout blocks) */
int ddh_a2av_pack_b(const double *src, double *dst, long outer, long na, long row, int nparts, long block, void *stream);
int ddh_a2av_unpack_b(const double *src, double *dst, long outer_na, long nb, long inner, int nparts, long block, void *stream);

#ifdef __cplusplus
}
#endif
#endif
